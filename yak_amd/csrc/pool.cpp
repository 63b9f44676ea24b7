/*
 * pool.cpp -- what the engine's translation units share below the counting passes: the wall clock, the run-time settings (public environment
 * names + the test hook) and the device memory pool (superblocks for small buffers, physical chunks behind virtual ranges for large ones).
 * Cut out of engine.cpp in round 6; nothing here knows about tables or passes.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <map>
#include <string>
#include <mutex>
#include <algorithm>
#include <chrono>
#include "yk_device.h"
#include "engine.h"

double yk_now_ms(void)
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static inline double now_ms() { return yk_now_ms(); }

/* Run-time settings.  The environment reaches the few a user has a reason to touch; every other name is a test
 * switch -- it forces a code path that sizes or shapes of input would otherwise select -- and only yakamd_test_set() (tests/conftest.py's
 * `knobs`, `yak-amd -X name=value`, `bench.py --knob name=value`) reaches those.  A value set through the hook wins over the environment. */
static std::mutex g_knob_mu;
static std::map<std::string, int64_t> g_knob;
static bool knob_is_public(const char *name)
{
	static const char *const pub[] = { "YAKAMD_VERBOSE", "YAKAMD_DEVICE", "YAKAMD_GPUS", "YAKAMD_GPU_LIST", "YAKAMD_AUTO_SWEEP_GB", "YAKAMD_MGPU_CHUNK", "YAKAMD_MGPU_NO_RCCL",
	                                   "YAKAMD_BATCH", "YAKAMD_FAST_BUDGET", "YAKAMD_NO_RETAIN", "YAKAMD_RETAIN_GB", "YAKAMD_PARSE_THREADS", "YAKAMD_PARSE_WINDOW",
	                                   "YAKAMD_NO_LIBDEFLATE", "YAKAMD_NO_PGZ", "YAKAMD_NO_HOST_PACK" };
	for (const char *q : pub) if (strcmp(q, name) == 0) return true;
	return false;
}
int64_t yk_knob(const char *name, int64_t dflt)
{
	{
		std::lock_guard<std::mutex> lk(g_knob_mu);
		if (!g_knob.empty()) { auto it = g_knob.find(name); if (it != g_knob.end()) return it->second; }
	}
	if (!knob_is_public(name)) return dflt;
	const char *s = getenv(name);
	return s && *s ? atoll(s) : dflt;
}
extern "C" void yakamd_test_set(const char *name, int64_t value) { std::lock_guard<std::mutex> lk(g_knob_mu); g_knob[name] = value; }
extern "C" void yakamd_test_reset(void) { std::lock_guard<std::mutex> lk(g_knob_mu); g_knob.clear(); }

/* Device memory pool.  A counting job allocates and frees tens of GB per pass (bloom filters, partition buffers, table arenas); on ROCm a
 * hipMalloc costs ~27 ms per GB (measured: 97 GB in 2.7 s) and a hipFree of that size about as much, so freed memory is kept and handed out
 * again.  The driver's allocations are the pool's SUPERBLOCKS; a freed range joins the free ranges next to it inside its superblock, and a
 * request is carved from the best-fitting free range: the 97 GB a slice's first partition sweep used serve, once freed, the two 48 GB arrays of
 * the counting stage, and the next, shorter slice finds its buffers inside the longer one's.  Small requests (< 64 MB) only take ranges of about
 * their own size, so that they never pin a large superblock.  Beyond the cap (three quarters of the HBM idle) the superblocks that are
 * entirely free go back to the driver, least recently used first.  YAKAMD_POOL=0 disables it; a failed allocation drops every free
 * superblock and retries. */
#include <map>
#include <mutex>
#include <unordered_map>
struct DevPool {
	std::mutex mu;
	struct Super { size_t size; u64 stamp; };
	std::map<char*, Super> supers;                     /* by base address */
	std::map<char*, size_t> free_at;                   /* free ranges by start address (never spanning two superblocks) */
	std::multimap<size_t, char*> free_sz;              /* the same ranges by length */
	std::unordered_map<void*, size_t> live;            /* ranges handed out */
	size_t cached = 0;                                 /* bytes in free ranges */
	size_t in_use = 0, peak_in_use = 0;                /* bytes handed out, and their high-water mark (yakamd_peak_bytes) */
	u64 clock = 0;
	/* what the driver was asked for (YAKAMD_VERBOSE prints it: a job whose buffers do not come out of the pool pays ~27 ms per GB) */
	u64 n_malloc = 0, n_release = 0, n_trim = 0; double gb_malloc = 0, ms_malloc = 0, ms_release = 0;
	/* the tier of large buffers (>= VM_MIN): physical chunks of VM_CH bytes behind virtual ranges, see vm_alloc */
	struct VmRange { size_t size; std::vector<hipMemGenericAllocationHandle_t> h; u64 stamp; };
	std::map<char*, VmRange> vm_live, vm_idle;         /* ranges handed out / mapped and idle, by address */
	std::multimap<size_t, char*> vm_idle_sz;           /* the idle ranges by length */
	std::vector<hipMemGenericAllocationHandle_t> vm_spare;   /* chunks mapped nowhere */
	size_t vm_cached = 0, vm_phys = 0, vm_dead_va = 0; /* bytes idle (ranges + spare chunks); bytes of chunks obtained from the driver and not given back; bytes of address space left behind by unmapped ranges */
	size_t total_mem = 0;                              /* of the device (hipMemGetInfo, once) */
	bool vm_off = false;                               /* a virtual-memory call failed: large buffers come from superblocks like the small ones */
	u64 n_vm_new = 0, n_vm_reuse = 0, n_vm_remap = 0, n_vm_create = 0; double ms_vm_map = 0, ms_vm_create = 0;
};
static DevPool g_pool[16];
static DevPool &pool_here() { int d = 0; (void)hipGetDevice(&d); return g_pool[d & 15]; }
static const size_t POOL_SPLIT_MIN = (size_t)64 << 20;

static std::map<char*, DevPool::Super>::iterator pool_super_of(DevPool &P, char *p)
{
	auto it = P.supers.upper_bound(p);
	return it == P.supers.begin() ? P.supers.end() : std::prev(it);
}
static void pool_range_drop(DevPool &P, std::map<char*, size_t>::iterator it)
{
	auto r = P.free_sz.equal_range(it->second);
	for (auto q = r.first; q != r.second; ++q) if (q->second == it->first) { P.free_sz.erase(q); break; }
	P.cached -= it->second;
	P.free_at.erase(it);
}
static void pool_range_add(DevPool &P, char *p, size_t n)
{
	/* join the neighbours inside the same superblock */
	auto su = pool_super_of(P, p);
	char *lo = su->first, *hi = su->first + su->second.size;
	auto nx = P.free_at.lower_bound(p);
	if (nx != P.free_at.end() && nx->first == p + n && nx->first < hi) { n += nx->second; pool_range_drop(P, nx); }
	auto pv = P.free_at.lower_bound(p);
	if (pv != P.free_at.begin()) { --pv; if (pv->first + pv->second == p && pv->first >= lo) { p = pv->first; n += pv->second; pool_range_drop(P, pv); } }
	P.free_at[p] = n; P.free_sz.insert({ n, p }); P.cached += n;
	su->second.stamp = ++P.clock;
}
/* give the entirely free superblocks back to the driver, least recently used first, until at most `keep` bytes stay idle */
static void pool_release(DevPool &P, size_t keep)
{
	while (P.cached > keep) {
		std::map<char*, DevPool::Super>::iterator best = P.supers.end();
		for (auto it = P.supers.begin(); it != P.supers.end(); ++it) {
			auto f = P.free_at.find(it->first);
			if (f == P.free_at.end() || f->second != it->second.size) continue;
			if (best == P.supers.end() || it->second.stamp < best->second.stamp) best = it;
		}
		if (best == P.supers.end()) break;
		pool_range_drop(P, P.free_at.find(best->first));
		{ const double t0 = now_ms(); (void)hipFree(best->first); P.ms_release += now_ms() - t0; ++P.n_release; }
		P.supers.erase(best);
	}
}
static void pool_trim(DevPool &P) { ++P.n_trim; pool_release(P, 0); }

/* Large buffers: fungible memory.  A superblock that is free but of the wrong size is useless to the next request -- the 5 Gb assembly obtained 329 GB
 * from the driver for 188 GB in use (buffers of 22-90 GB that do not fit the ranges earlier ones left; one out-of-memory trim on the way gave 115 GB
 * back that were then obtained again), and the driver charges ~30 ms per GB beyond the first ~112 GB of a process (tests/tools/mb/mb_malloc.hip).  So a
 * request of VM_MIN (1 GiB) bytes or more gets a virtual range of its own, backed by physical chunks of VM_CH bytes (hipMemCreate / hipMemMap).  A freed range
 * stays mapped (the next request of that size takes it as it is: the steady state of a benchmark's steps costs nothing); a request that finds no idle
 * range of its size takes the chunks of idle ranges, least recently used first, and maps them into a fresh range -- 0.1 ms per GB
 * (tests/tools/mb/mb_vmm.hip: map 5 us per chunk, access 0.02 ms per GB, unmap 0.06 ms per GB; fills and random probes run as on hipMalloc memory) -- and
 * only what is still missing is created.  The driver is asked for the high-water mark of the buffers in use, rounded to chunks, and never for the
 * same memory twice.  YAKAMD_POOL_VM=0 (test switch), or any failing virtual-memory call, sends large buffers to the superblocks instead. */
static const size_t VM_MIN = (size_t)1 << 30;        /* (2 GiB serves a cfg3 rank 3 % better -- its ~1.1 GB per-round buffers split superblocks at no cost -- and two ranks' jobs on one device twice as badly: near a full device the superblocks of that class are released and obtained over and over, 573 -> 1107 ms; r06_experiments.txt e14) */
/* the chunk size is fixed with the first large buffer of the process (test switch YAKAMD_POOL_VM_CH: a multiple of 2 MiB) */
static size_t vm_ch()
{
	static const size_t v = [] { const int64_t e = yk_knob("YAKAMD_POOL_VM_CH", 0); return e >= (2 << 20) ? (size_t)e / (2u << 20) * (2u << 20) : (size_t)256 << 20; }();
	return v;
}
#define VM_CH (vm_ch())
static void vm_idle_drop(DevPool &P, std::map<char*, DevPool::VmRange>::iterator it)
{
	auto r = P.vm_idle_sz.equal_range(it->second.size);
	for (auto q = r.first; q != r.second; ++q) if (q->second == it->first) { P.vm_idle_sz.erase(q); break; }
	P.vm_idle.erase(it);
}
/* an idle range gives its chunks up.  A buffer may be freed while the last kernel that uses it is still queued (the next user comes behind it on the
 * stream, as with any stream-ordered allocator); memory that is about to leave its addresses must not have such work pending: `synced` makes the first
 * unmap of a call wait for the device -- requests that find their range idle never get here */
static void vm_unmap_idle(DevPool &P, std::map<char*, DevPool::VmRange>::iterator it, bool *synced)
{
	if (!*synced) { (void)hipDeviceSynchronize(); *synced = true; }
	(void)hipMemUnmap(it->first, it->second.size);
	/* The addresses are NOT given back (hipMemAddressFree): with this runtime (ROCm 7.2.0) kernels that touch a range whose addresses had been reserved, mapped,
	 * unmapped and freed before fault or hang -- a 1 Gb assembly counted twice in one process died in k_r2_place of the second job, on ranges that were alive
	 * and whose contents did not matter (every buffer pre-filled: same fault), while the same job with the addresses kept runs and writes the reference's
	 * bytes (profiles/r06_experiments.txt e7).  Address space is plentiful (47 bits); only requests that find no idle range of their size consume any */
	P.vm_dead_va += it->second.size;
	for (auto h : it->second.h) P.vm_spare.push_back(h);
	vm_idle_drop(P, it);
}
static std::map<char*, DevPool::VmRange>::iterator vm_lru(DevPool &P)
{
	auto best = P.vm_idle.end();
	for (auto it = P.vm_idle.begin(); it != P.vm_idle.end(); ++it) if (best == P.vm_idle.end() || it->second.stamp < best->second.stamp) best = it;
	return best;
}
/* idle memory of the tier back to the driver until at most `keep` bytes of it stay: spare chunks first, then idle ranges, least recently used first */
static void vm_release(DevPool &P, size_t keep)
{
	const double t0 = now_ms();
	bool synced = false;
	while (P.vm_cached > keep) {
		if (P.vm_spare.empty()) { auto it = vm_lru(P); if (it == P.vm_idle.end()) break; vm_unmap_idle(P, it, &synced); }
		(void)hipMemRelease(P.vm_spare.back()); P.vm_spare.pop_back();
		P.vm_cached -= VM_CH; P.vm_phys -= VM_CH; ++P.n_release;
	}
	P.ms_release += now_ms() - t0;
}
static void pool_trim(DevPool &P);
static void *vm_alloc(DevPool &P, size_t bytes)
{
	const size_t need = (bytes + VM_CH - 1) / VM_CH * VM_CH, m = need / VM_CH;
	/* While the pool holds less than YAKAMD_POOL_VM_ROOMY per cent (60) of the device, ranges are not taken apart: a request takes an idle range of up to
	 * 1.5 x its size as it is, or gets new chunks -- the ranges a repeated job needs settle after its first run and every later request finds one (taking a
	 * 16 GB range apart for a 12 GB request, and building the 16 GB range again a moment later, cost the default step 5.5 of its 50.7 ms: every unmap
	 * waits for the device first).  Beyond that the memory is worth more than the mapping: tight fits only, and chunks come from idle ranges */
	if (P.total_mem == 0) { size_t fr = 0, tt = 0; if (hipMemGetInfo(&fr, &tt) == hipSuccess) P.total_mem = tt; else (void)hipGetLastError(); }
	size_t held = P.vm_phys;
	for (auto &kv : P.supers) held += kv.second.size;
	const bool roomy = P.total_mem && (double)(held + need) <= (double)P.total_mem * (double)yk_knob("YAKAMD_POOL_VM_ROOMY", 60) / 100.0;
	auto fit = P.vm_idle_sz.lower_bound(need);
	if (fit != P.vm_idle_sz.end() && fit->first <= need + (roomy ? std::max(need / 2, VM_CH) : std::max(need / 8, VM_CH))) {      /* an idle range of about this size, as it is */
		char *p = fit->second;
		auto it = P.vm_idle.find(p);
		DevPool::VmRange r = std::move(it->second);
		vm_idle_drop(P, it);
		P.vm_cached -= r.size; P.in_use += r.size; P.peak_in_use = std::max(P.peak_in_use, P.in_use);
		r.stamp = ++P.clock;
		P.vm_live[p] = std::move(r);
		++P.n_vm_reuse;
		return p;
	}
	const double t0 = now_ms();
	bool remapped = false;
	if (!roomy) while (P.vm_spare.size() < m) { auto it = vm_lru(P); if (it == P.vm_idle.end()) break; vm_unmap_idle(P, it, &remapped); }
	int dev = 0;
	(void)hipGetDevice(&dev);
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
	bool trimmed = false;
	const double tc0 = now_ms();
	while (P.vm_spare.size() < m) {
		hipMemGenericAllocationHandle_t h;
		if (hipMemCreate(&h, VM_CH, &prop, 0) != hipSuccess) {
			(void)hipGetLastError();
			{ auto it = vm_lru(P); if (it != P.vm_idle.end()) { vm_unmap_idle(P, it, &remapped); continue; } }   /* the device is full: idle ranges give their chunks up after all */
			if (trimmed) return 0;                                  /* nothing left to take: the chunks gathered so far stay spare */
			pool_trim(P); trimmed = true;                           /* the superblocks' idle memory goes back to the driver first */
			continue;
		}
		P.vm_spare.push_back(h); P.vm_cached += VM_CH; P.vm_phys += VM_CH; ++P.n_vm_create;
		P.gb_malloc += (double)VM_CH / 1e9;
	}
	P.ms_vm_create += now_ms() - tc0;
	void *va = 0;
	if (hipMemAddressReserve(&va, need, 0, 0, 0) != hipSuccess) { (void)hipGetLastError(); P.vm_off = true; return 0; }
	DevPool::VmRange r;
	r.size = need; r.stamp = ++P.clock;
	size_t mapped = 0;
	bool ok = true;
	for (; mapped < m && ok; ++mapped) {
		hipMemGenericAllocationHandle_t h = P.vm_spare.back();
		ok = hipMemMap((char*)va + mapped * VM_CH, VM_CH, 0, h, 0) == hipSuccess;
		if (ok) { P.vm_spare.pop_back(); r.h.push_back(h); } else --mapped;
	}
	if (ok) {
		/* this device, and every peer that can reach it (a sharded table's image is read across devices by the set operations and the peer-copy exchange) */
		std::vector<hipMemAccessDesc> acc;
		int nd = 1;
		(void)hipGetDeviceCount(&nd);
		for (int d = 0; d < nd; ++d) {
			int can = d == dev;
			if (!can && hipDeviceCanAccessPeer(&can, d, dev) != hipSuccess) { can = 0; (void)hipGetLastError(); }
			if (!can) continue;
			hipMemAccessDesc a = {};
			a.location.type = hipMemLocationTypeDevice; a.location.id = d; a.flags = hipMemAccessFlagsProtReadWrite;
			if (d == dev) acc.insert(acc.begin(), a); else acc.push_back(a);
		}
		ok = hipMemSetAccess(va, need, acc.data(), acc.size()) == hipSuccess;
		if (!ok && acc.size() > 1) { (void)hipGetLastError(); ok = hipMemSetAccess(va, need, acc.data(), 1) == hipSuccess; }
	}
	if (!ok) {
		(void)hipGetLastError();
		if (!r.h.empty()) (void)hipMemUnmap(va, r.h.size() * VM_CH);
		(void)hipMemAddressFree(va, need);
		for (auto h : r.h) P.vm_spare.push_back(h);
		P.vm_off = true;
		fprintf(stderr, "[W::yak_amd] the virtual-memory tier of the device pool failed to map a range: large buffers come from hipMalloc from now on\n");
		return 0;
	}
	P.vm_cached -= need; P.in_use += need; P.peak_in_use = std::max(P.peak_in_use, P.in_use);
	P.vm_live[(char*)va] = std::move(r);
	P.ms_vm_map += now_ms() - t0;
	if (remapped) ++P.n_vm_remap; else ++P.n_vm_new;
	return va;
}

static void *pool_alloc_(size_t bytes, bool plain);
/* test switch YAKAMD_POOL_FILL = v + 1: every buffer is filled with byte v before it is handed out (the null stream: behind everything queued) -- no result may
 * depend on what a buffer held when it was obtained, be it the driver's zeros or its last user's data */
void *yk_pool_alloc(size_t bytes, bool plain)
{
	void *p = pool_alloc_(bytes, plain);
	const int64_t f = yk_knob("YAKAMD_POOL_FILL", 0);
	if (p && f > 0) { (void)hipMemset(p, (int)(f - 1) & 0xff, bytes); }
	return p;
}
static void *pool_alloc_(size_t bytes, bool plain)
{
	static const bool on = true;
	DevPool &P = pool_here();
	const size_t gran = bytes >= (1u << 20) ? (2u << 20) : 256;
	bytes = (bytes + gran - 1) / gran * gran;
	std::lock_guard<std::mutex> lk(P.mu);
	if (bytes >= (size_t)yk_knob("YAKAMD_POOL_VM_MIN", (int64_t)VM_MIN) && !plain && !P.vm_off && yk_knob("YAKAMD_POOL_VM", 1) != 0) {   /* (YAKAMD_POOL_VM_MIN: tests send small buffers through the tier) */
		void *p = vm_alloc(P, bytes);
		if (p) return p;
	}
	if (on) {
		auto it = P.free_sz.lower_bound(bytes);
		if (it != P.free_sz.end() && (it->first <= bytes + bytes / 4 || bytes >= POOL_SPLIT_MIN)) {
			char *p = it->second;
			const size_t have = it->first;
			pool_range_drop(P, P.free_at.find(p));
			size_t take = bytes;
			if (have - take < (2u << 20) || bytes < POOL_SPLIT_MIN) take = have;      /* no crumbs; small requests never split */
			if (have > take) { P.free_at[p + take] = have - take; P.free_sz.insert({ have - take, p + take }); P.cached += have - take; }
			P.live[p] = take;
			P.in_use += take; P.peak_in_use = std::max(P.peak_in_use, P.in_use);
			pool_super_of(P, p)->second.stamp = ++P.clock;
			return p;
		}
	}
	void *p = 0;
	const double t0 = now_ms();
	if (hipMalloc(&p, bytes) != hipSuccess) {
		(void)hipGetLastError();
		pool_trim(P);
		vm_release(P, 0);
		if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
	}
	P.ms_malloc += now_ms() - t0; ++P.n_malloc; P.gb_malloc += (double)bytes / 1e9;
	P.supers[(char*)p] = DevPool::Super{ bytes, ++P.clock };
	P.live[p] = bytes;
	P.in_use += bytes; P.peak_in_use = std::max(P.peak_in_use, P.in_use);
	return p;
}

void yk_pool_free(void *p)
{
	static const bool on = true;
	/* idle bytes kept per device: three quarters of the HBM.  A step of the larger configurations
	 * (1 Gb assembly, 30 M reads) turns over > 100 GB; a cap below the turnover makes every step pay the driver for its buffers again
	 * (measured with 96 GB: 2.3 s instead of 0.27 s per cfg4 pass).  An allocation that fails drops the whole cache and retries */
	static const size_t cap = []() -> size_t {
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return (size_t)96 << 30; }
		return tot / 4 * 3;
	}();
	/* the block's owner is the pool that handed it out, whatever device is current now */
	DevPool *Pp = &pool_here();
	{
		bool mine;
		{ std::lock_guard<std::mutex> lk(Pp->mu); mine = Pp->live.count(p) != 0 || Pp->vm_live.count((char*)p) != 0; }
		for (int d = 0; d < 16 && !mine; ++d) { std::lock_guard<std::mutex> lk(g_pool[d].mu); if (g_pool[d].live.count(p) || g_pool[d].vm_live.count((char*)p)) { Pp = &g_pool[d]; mine = true; } }
	}
	DevPool &P = *Pp;
	std::lock_guard<std::mutex> lk(P.mu);
	{
		auto vt = P.vm_live.find((char*)p);
		if (vt != P.vm_live.end()) {                                  /* stays mapped: the next request of this size takes it as it is */
			DevPool::VmRange r = std::move(vt->second);
			P.vm_live.erase(vt);
			P.in_use -= r.size; P.vm_cached += r.size;
			r.stamp = ++P.clock;
			P.vm_idle_sz.insert({ r.size, (char*)p });
			P.vm_idle[(char*)p] = std::move(r);
			if (P.cached + P.vm_cached > cap) { vm_release(P, cap > P.cached ? cap - P.cached : 0); }
			return;
		}
	}
	auto it = P.live.find(p);
	if (it == P.live.end()) { (void)hipFree(p); return; }          /* not the pool's */
	const size_t n = it->second;
	P.live.erase(it);
	P.in_use -= n;
	pool_range_add(P, (char*)p, n);
	pool_release(P, on ? (cap > P.vm_cached ? cap - P.vm_cached : 0) : 0);
}

/* device memory the pool of device `dev` has obtained from the driver and not given back (in use or idle): what a process has paid the driver for already */
size_t yk_pool_held_bytes(int dev)
{
	DevPool &P = g_pool[dev & 15];
	std::lock_guard<std::mutex> lk(P.mu);
	size_t n = P.vm_phys;
	for (auto &kv : P.supers) n += kv.second.size;
	return n;
}
size_t yk_pool_cached_bytes(void) { DevPool &P = pool_here(); std::lock_guard<std::mutex> lk(P.mu); return P.cached + P.vm_cached; }

/* high-water mark of the device memory the library had IN USE on device `dev` (what a job needs; the idle ranges the pool keeps are not in it);
 * reset != 0 starts a new measurement from what is in use now */
extern "C" int64_t yakamd_peak_bytes(int dev, int reset)
{
	DevPool &P = g_pool[dev & 15];
	std::lock_guard<std::mutex> lk(P.mu);
	const int64_t v = (int64_t)P.peak_in_use;
	if (reset) P.peak_in_use = P.in_use;
	return v;
}
extern "C" void yakamd_trim(void) { DevPool &P = pool_here(); std::lock_guard<std::mutex> lk(P.mu); pool_trim(P); vm_release(P, 0); }
void yk_pool_report(const char *what);
extern "C" void yakamd_pool_report(const char *what) { yk_pool_report(what ? what : "the call"); }
void yk_pool_report(const char *what)
{
	DevPool &P = pool_here();
	std::lock_guard<std::mutex> lk(P.mu);
	size_t live = 0, sup = 0;
	for (auto &kv : P.live) live += kv.second;
	for (auto &kv : P.supers) sup += kv.second.size;
	fprintf(stderr, "[yak_amd] pool after %s: %llu hipMalloc (%.1f GB, %.0f ms), %llu hipFree (%.0f ms), %llu trims; %zu superblocks of %.1f GB hold %.1f GB in use and %.1f GB free in %zu ranges\n", what,
	        (unsigned long long)P.n_malloc, P.gb_malloc, P.ms_malloc, (unsigned long long)P.n_release, P.ms_release, (unsigned long long)P.n_trim, P.supers.size(), (double)sup / 1e9, (double)live / 1e9, (double)P.cached / 1e9, P.free_at.size());
	size_t vlive = 0, vidle = 0;
	for (auto &kv : P.vm_live) vlive += kv.second.size;
	for (auto &kv : P.vm_idle) vidle += kv.second.size;
	fprintf(stderr, "[yak_amd] pool after %s, large buffers: %.1f GB of %zu MiB chunks obtained (%llu created in %.0f ms) hold %.1f GB in use in %zu ranges, %.1f GB idle in %zu mapped ranges and %zu spare chunks; %llu ranges new, %llu taken as they were, %llu built from other ranges' chunks (%.1f ms of mapping, %.1f GB of address space left behind); in use at the peak (both tiers) %.1f GB%s\n", what,
	        (double)P.vm_phys / 1e9, VM_CH >> 20, (unsigned long long)P.n_vm_create, P.ms_vm_create, (double)vlive / 1e9, P.vm_live.size(), (double)vidle / 1e9, P.vm_idle.size(), P.vm_spare.size(),
	        (unsigned long long)P.n_vm_new, (unsigned long long)P.n_vm_reuse, (unsigned long long)P.n_vm_remap, P.ms_vm_map, (double)P.vm_dead_va / 1e9, (double)P.peak_in_use / 1e9, P.vm_off ? " (tier switched off after a failed call)" : "");
}

void yk_pool_release(void *p) { if (p) yk_pool_free(p); }
void *yk_pool_get(size_t bytes) { return yk_pool_alloc(bytes ? bytes : 1, true); }   /* (plain hipMalloc memory: what the collective library's transports are used with) */   /* the current device's pool (multi-GPU chunk and exchange buffers: a job's second call finds the first one's) */
