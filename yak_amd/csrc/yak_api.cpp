/*
 * yak_api.cpp -- the drop-in C surface (include/yak.h) on top of the HBM-resident engine.
 *
 * Every function keeps the reference's name, argument meaning, return convention and messages
 * (reference file:line cited per function).  Host work here is orchestration only: parsing the
 * input file, staging bases, writing the .yak file from the host mirror.  All counting, bloom
 * gating, table layout, clearing and shrinking run in the HIP kernels of kernels.hip.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <assert.h>
#include <zlib.h>
#include <sys/time.h>
#include <sys/resource.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdarg>
#include <thread>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <vector>
#include <map>
#include <algorithm>
#include <string>
#include <mutex>
#include <condition_variable>
#include <cmath>
#include "engine.h"
#include <dlfcn.h>
#include <immintrin.h>
#include "pgz.h"                                            /* parallel inflate of ordinary gzip files */
#include <rccl/rccl.h>                                   /* types and prototypes only: the library is opened when a job asks for several GPUs */

struct yak_ht_t { uint32_t bits, count; uint32_t *used; uint64_t *keys; };
struct yak_ch_ext { yak_ch_t pub; yakamd_ctx *ctx; uint32_t magic; int n_sub; yak_ch_t **sub; };   /* n_sub > 1: sharded over several GPUs, sub[r] owns prefixes [r P / n_sub, (r + 1) P / n_sub) */
#define EXT_MAGIC 0x59414b41u

extern "C" {

int yak_verbose = 3;                                         /* reference sys.c:5 */

unsigned char seq_nt4_table[256] = {                         /* reference misc.c:4-21 */
#define R4(v) v, v, v, v
#define R16(v) R4(v), R4(v), R4(v), R4(v)
	0, 1, 2, 3, R4(4), R4(4), R4(4),
	R16(4), R16(4), R16(4),
	4, 0, 4, 1, 4, 4, 4, 2, R4(4), R4(4),
	4, 4, 4, 4, 3, 3, 4, 4, R4(4), R4(4),
	4, 0, 4, 1, 4, 4, 4, 2, R4(4), R4(4),
	4, 4, 4, 4, 3, 3, 4, 4, R4(4), R4(4),
	R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4)
#undef R16
#undef R4
};

static double yk_realtime0 = -1;
static double yk_realtime(void)
{
	struct timeval tp;
	gettimeofday(&tp, 0);
	const double t = tp.tv_sec + tp.tv_usec * 1e-6;
	if (yk_realtime0 < 0) yk_realtime0 = t;
	return t - yk_realtime0;
}
static double yk_cputime(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

void yak_copt_init(yak_copt_t *o)                            /* reference misc.c:23-32 */
{
	memset(o, 0, sizeof(*o));
	o->bf_shift = 0; o->bf_n_hash = 4; o->k = 31; o->pre = 10; o->n_thread = 4;
	o->chunk_size = 10000000;
}

/* ---- stand-alone host bloom filter: API completeness only (reference bbf.c); the counting path
 * keeps its filters in HBM and never calls these ---- */
yak_bf_t *yak_bf_init(int n_shift, int n_hashes)
{
	if (n_shift + YAK_BLK_SHIFT > 64 || n_shift < YAK_BLK_SHIFT) return 0;
	yak_bf_t *b = (yak_bf_t*)calloc(1, sizeof(*b));
	void *p = 0;
	b->n_shift = n_shift; b->n_hashes = n_hashes;
	if (posix_memalign(&p, 64, (size_t)1 << (n_shift - 3)) != 0) { free(b); return 0; }
	memset(p, 0, (size_t)1 << (n_shift - 3));
	b->b = (uint8_t*)p;
	return b;
}

void yak_bf_destroy(yak_bf_t *b) { if (b) { free(b->b); free(b); } }

int yak_bf_insert(yak_bf_t *b, uint64_t hash)
{
	const int lgblk = b->n_shift - YAK_BLK_SHIFT;
	uint8_t *blk = b->b + ((hash & ((1ULL << lgblk) - 1)) << 6);
	int z = (int)(hash >> lgblk) & YAK_BLK_MASK, step = (int)(hash >> b->n_shift) & YAK_BLK_MASK, hits = 0;
	if ((step & 31) == 0) step = (step + 1) & YAK_BLK_MASK;
	for (int i = 0; i < b->n_hashes; ++i, z = (z + step) & YAK_BLK_MASK) {
		hits += blk[z >> 3] >> (z & 7) & 1;
		blk[z >> 3] |= (uint8_t)(1 << (z & 7));
	}
	return hits;
}

/* ---- table life cycle (reference htab.c:13-49) ---- */
yak_ch_t *yak_ch_init(int k, int pre, int n_hash, int n_shift)
{
	if (pre < YAK_COUNTER_BITS) return 0;
	yakamd_ctx *ctx = yk_ctx_create(k, pre, n_hash, n_shift);
	if (!ctx) return 0;                                      /* no GPU: fail, never count on the CPU */
	yak_ch_ext *e = (yak_ch_ext*)calloc(1, sizeof(*e));
	e->ctx = ctx; e->magic = EXT_MAGIC;
	yak_ch_t *h = &e->pub;
	h->k = k; h->pre = pre;
	h->h = (yak_ch1_t*)calloc((size_t)1 << pre, sizeof(yak_ch1_t));
	if (n_hash > 0 && n_shift > pre) {
		h->n_hash = n_hash; h->n_shift = n_shift;
		if (n_shift - pre >= YAK_BLK_SHIFT && n_shift - pre + YAK_BLK_SHIFT <= 64) {
			/* descriptors only: the bits live in HBM (b == NULL on the host side) */
			yak_bf_t *bf = (yak_bf_t*)calloc((size_t)1 << pre, sizeof(yak_bf_t));
			for (int i = 0; i < 1 << pre; ++i) { bf[i].n_shift = n_shift - pre; bf[i].n_hashes = n_hash; h->h[i].b = &bf[i]; }
		}
	}
	yk_ctx_sync_host(ctx, h);
	return h;
}

#define YK_MULTI(e) ((e)->n_sub > 1)
static void multi_tot(yak_ch_t *h) { yak_ch_ext *e = (yak_ch_ext*)h; uint64_t t = 0; for (int r = 0; r < e->n_sub; ++r) t += e->sub[r]->tot; h->tot = t; }
static int multi_refuse(const yak_ch_t *h, const char *what)
{
	if (!YK_MULTI((const yak_ch_ext*)h)) return 0;
	fprintf(stderr, "[E::%s] not available on a table sharded over prefix ranges (several GPUs, or a large unfiltered count taken in sweeps: YAKAMD_GPUS / YAKAMD_AUTO_SWEEP_GB)\n", what);
	return 1;
}

void yak_ch_destroy_bf(yak_ch_t *h)
{
	yak_ch_ext *e = (yak_ch_ext*)h;
	if (YK_MULTI(e)) { for (int r = 0; r < e->n_sub; ++r) yak_ch_destroy_bf(e->sub[r]); for (int i = 0; i < 1 << h->pre; ++i) h->h[i].b = 0; return; }
	if (h->h[0].b) free(h->h[0].b);                          /* one block of descriptors */
	for (int i = 0; i < 1 << h->pre; ++i) h->h[i].b = 0;
	yk_ctx_destroy_bf(e->ctx);
}

void yak_ch_destroy(yak_ch_t *h)
{
	if (h == 0) return;
	yak_ch_ext *e = (yak_ch_ext*)h;
	if (YK_MULTI(e)) { for (int r = 0; r < e->n_sub; ++r) yak_ch_destroy(e->sub[r]); free(e->sub); free(h->h); free(e); return; }
	yak_ch_destroy_bf(h);
	yk_ctx_destroy(e->ctx);
	free(h->h); free(e);
}

/* ---- reference htab.c:51-78.  The list is one bucket of hashed k-mers sharing a prefix; list
 * order is stream order.  Runs as a one-batch device pass. ---- */
int yak_ch_insert_list(yak_ch_t *h, int create_new, int n, const uint64_t *a)
{
	yak_ch_ext *e = (yak_ch_ext*)h;
	if (n <= 0) return 0;
	if (YK_MULTI(e)) return yak_ch_insert_list(e->sub[(a[0] & ((1ULL << h->pre) - 1)) * e->n_sub >> h->pre], create_new, n, a);   /* the owner of the list's prefix */
	const uint64_t pm = (1ULL << h->pre) - 1;
	std::vector<uint64_t> hv; std::vector<uint32_t> tv;
	hv.reserve(n); tv.reserve(n);
	for (int j = 0; j < n; ++j)
		if ((a[j] & pm) == (a[0] & pm)) { hv.push_back(a[j]); tv.push_back((uint32_t)j); }   /* htab.c:61 */
	/* the reference's callers run this from kt_for workers, one sub-table each (count.c:129-143); here a call
	 * is a whole-table device pass, so concurrent callers take turns.  Failures are reported, never silent. */
	struct Lock { yakamd_ctx *c; Lock(yakamd_ctx *c_) : c(c_) { yk_ctx_lock(c); } ~Lock() { yk_ctx_unlock(c); } } lock(e->ctx);
	const size_t nb = hv.size() * 8, need = ((nb + 15) & ~(size_t)15) + hv.size() * 4;
	uint8_t *d = (uint8_t*)yk_ctx_scratch(e->ctx, need);
	int64_t n_ins = -1;
	bool ok = d != 0 && hipSetDevice(yk_ctx_device(e->ctx)) == hipSuccess;
	uint64_t *d_h = (uint64_t*)d; uint32_t *d_t = (uint32_t*)(d + ((nb + 15) & ~(size_t)15));
	ok = ok && hipMemcpy(d_h, hv.data(), nb, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_t, tv.data(), tv.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
	if (ok) {
		const uint64_t t0 = yk_ctx_list_time(e->ctx, (uint64_t)n);
		ok = yakamd_pass_begin(h, create_new) == 0;
		if (ok) {
			ok = yakamd_feed_hashed_dev(h, d_h, d_t, (int64_t)hv.size(), t0, (uint64_t)n) == 0;
			n_ins = yakamd_pass_end(h);                          /* closes the pass whatever the feed did */
			ok = ok && n_ins >= 0;
		}
	}
	if (!ok) { fprintf(stderr, "[E::%s] %s\n", __func__, *yakamd_last_error() ? yakamd_last_error() : "device buffer allocation or copy failed"); return -1; }
	return (int)n_ins;
}

static inline uint32_t ht_cap(const yak_ht_t *g) { return g->keys ? 1U << g->bits : 0U; }

static uint32_t ht_get(const yak_ht_t *g, uint64_t key)      /* khashl.h:137-150 on the host mirror */
{
	if (g->keys == 0) return 0;
	const uint32_t n = 1U << g->bits, mask = n - 1;
	uint32_t i = (uint32_t)((uint32_t)(key >> YAK_COUNTER_BITS) * 2654435769U) >> (32 - g->bits), first = i;
	while ((g->used[i >> 5] >> (i & 31) & 1) && g->keys[i] >> YAK_COUNTER_BITS != key >> YAK_COUNTER_BITS) {
		i = (i + 1) & mask;
		if (i == first) return n;
	}
	return (g->used[i >> 5] >> (i & 31) & 1) ? i : n;
}

int yak_ch_get(const yak_ch_t *h, uint64_t x)                /* reference htab.c:93-100 */
{
	yak_ch_ext *e = (yak_ch_ext*)h;
	if (YK_MULTI(e)) return yak_ch_get(e->sub[(x & ((1ULL << h->pre) - 1)) * e->n_sub >> h->pre], x);
	if (yk_ctx_sync_host(e->ctx, (yak_ch_t*)h)) return -1;
	const yak_ht_t *g = h->h[x & ((1ULL << h->pre) - 1)].h;
	const uint32_t i = ht_get(g, x >> h->pre << YAK_COUNTER_BITS);
	return i == ht_cap(g) ? -1 : (int)(g->keys[i] & YAK_MAX_COUNT);
}

int yak_ch_inc(yak_ch_t *h, uint64_t x)                      /* reference htab.c:80-91 */
{
	if (YK_MULTI((yak_ch_ext*)h)) { yak_ch_ext *e = (yak_ch_ext*)h; return yak_ch_inc(e->sub[(x & ((1ULL << h->pre) - 1)) * e->n_sub >> h->pre], x); }
	int c = -1;                                              /* one single-lane kernel on the table image; a valid host mirror is patched in place */
	yakamd_ctx *ctx = ((yak_ch_ext*)h)->ctx;
	struct Lock { yakamd_ctx *c; Lock(yakamd_ctx *c_) : c(c_) { yk_ctx_lock(c); } ~Lock() { yk_ctx_unlock(c); } } lock(ctx);   /* shares the stream and the counters with yak_ch_insert_list */
	if (yk_ctx_inc(ctx, x, &c) != 0) { fprintf(stderr, "[E::%s] %s\n", __func__, yakamd_last_error()); return -1; }
	return c;
}

void yak_ch_clear(yak_ch_t *h, int n_thread)                 /* reference htab.c:127-130 */
{
	(void)n_thread;
	if (YK_MULTI((yak_ch_ext*)h)) { yak_ch_ext *e = (yak_ch_ext*)h; for (int r = 0; r < e->n_sub; ++r) yak_ch_clear(e->sub[r], n_thread); return; }
	yk_ctx_clear(((yak_ch_ext*)h)->ctx);
}

void yak_ch_shrink(yak_ch_t *h, int min, int max, int n_thread) /* reference htab.c:199-208 */
{
	(void)n_thread;
	if (YK_MULTI((yak_ch_ext*)h)) {                           /* every GPU shrinks its own sub-tables, side by side */
		yak_ch_ext *e = (yak_ch_ext*)h;
		std::vector<std::thread> th;
		for (int r = 0; r < e->n_sub; ++r) th.emplace_back([=]() { yak_ch_shrink(e->sub[r], min, max, n_thread); });
		for (auto &t : th) t.join();
		multi_tot(h);
		return;
	}
	unsigned long long tot = 0;
	const int hi = (max >= min && max <= YAK_MAX_COUNT) ? max : YAK_MAX_COUNT;
	if (yk_ctx_shrink(((yak_ch_ext*)h)->ctx, min, hi, &tot) == 0) h->tot = tot;
}

/* reference htab.c:102-110: resize every sub-table filled to less than a third */
void yak_ch_tighten(yak_ch_t *h)
{
	if (YK_MULTI((yak_ch_ext*)h)) { yak_ch_ext *e = (yak_ch_ext*)h; for (int r = 0; r < e->n_sub; ++r) yak_ch_tighten(e->sub[r]); return; }
	if (yk_ctx_tighten(((yak_ch_ext*)h)->ctx)) fprintf(stderr, "[E::yak_ch_tighten] %s\n", yakamd_last_error());
}

/* ---- operations on two tables.  Either operand may be sharded over prefix ranges (yak_count() shards by itself: several GPUs, or an
 * unfiltered count of a large plain file taken in sweeps); every sub-table belongs to exactly one shard of each operand, and the
 * reference's operations are per sub-table (kt_for over 1 << pre, htab.c:246-347), so they are carried out shard by shard. ---- */
int64_t yakamd_dump_mem(yak_ch_t *h, uint8_t **out);
int64_t yakamd_dump_range_mem(yak_ch_t *h, int lo, int hi, uint8_t **out);
static std::vector<yak_ch_t*> shards_of(const yak_ch_t *h)
{
	const yak_ch_ext *e = (const yak_ch_ext*)h;
	return YK_MULTI(e) ? std::vector<yak_ch_t*>(e->sub, e->sub + e->n_sub) : std::vector<yak_ch_t*>(1, (yak_ch_t*)h);
}
static inline yakamd_ctx *shard_ctx(yak_ch_t *s) { return ((yak_ch_ext*)s)->ctx; }
static void set_tot(yak_ch_t *h) { if (YK_MULTI((yak_ch_ext*)h)) multi_tot(h); }

static bool parse_yak_image(const uint8_t *img, size_t sz, uint32_t hdr[3], std::vector<uint32_t> &caps, std::vector<uint32_t> &sizes, std::vector<uint64_t> &keys)
{
	if (sz < 16 || memcmp(img, YAK_MAGIC, 4) != 0) return false;
	memcpy(hdr, img + 4, 12);
	const int P = 1 << hdr[1];
	caps.assign(P, 0); sizes.assign(P, 0); keys.clear();
	size_t off = 16;
	for (int p = 0; p < P; ++p) {
		if (off + 8 > sz) return false;
		uint32_t u[2]; memcpy(u, img + off, 8); off += 8;
		if (off + (size_t)8 * u[1] > sz) return false;
		caps[p] = u[0]; sizes[p] = u[1];
		const size_t at = keys.size();
		keys.resize(at + u[1]);
		if (u[1]) memcpy(&keys[at], img + off, (size_t)8 * u[1]);
		off += (size_t)8 * u[1];
	}
	return true;
}

/* an unsharded copy of `h` on device `dev`, by way of its .yak image (same keys and counts; the slot layout is the restored one, which
 * membership tests do not look at).  Only for the second operand of subtract / isec when it is sharded or lives on another device */
static yak_ch_t *unsharded_copy(const yak_ch_t *h, int dev)
{
	uint8_t *img = 0;
	const int64_t sz = yakamd_dump_mem((yak_ch_t*)h, &img);
	if (sz < 0) return 0;
	uint32_t hdr[3];
	std::vector<uint32_t> caps, sizes;
	std::vector<uint64_t> keys;
	const bool ok = parse_yak_image(img, (size_t)sz, hdr, caps, sizes, keys);
	free(img);
	if (!ok) return 0;
	yk_ctx_next_device(dev);
	yak_ch_t *c = yak_ch_init(h->k, h->pre, 0, 0);
	if (c && yk_ctx_load(shard_ctx(c), caps.data(), sizes.data(), keys.data()) != 0) { yak_ch_destroy(c); c = 0; }
	return c;
}

/* reference htab.c:287-316 / 318-347: keep the k-mers of h0 that are absent from (which = 1) / present in (2) h1 */
static void keep_by_membership(yak_ch_t *h0, const yak_ch_t *h1, int which, const char *fn_name)
{
	if (h0->k != h1->k || h0->pre != h1->pre) { fprintf(stderr, "[E::%s] tables of different k / prefix length\n", fn_name); return; }
	std::map<int, yak_ch_t*> copy_on;                             /* device -> unsharded copy of h1 made for it */
	bool ok = true;
	for (yak_ch_t *s0 : shards_of(h0)) {
		yakamd_ctx *c0 = shard_ctx(s0);
		const int dev = yk_ctx_device(c0);
		const yak_ch_t *other = h1;
		if (YK_MULTI((const yak_ch_ext*)h1) || yk_ctx_device(shard_ctx((yak_ch_t*)h1)) != dev) {
			if (!copy_on.count(dev)) copy_on[dev] = unsharded_copy(h1, dev);
			other = copy_on[dev];
			if (!other) { ok = false; break; }
		}
		unsigned long long tot = 0;
		if ((which == 1 ? yk_ctx_subtract(c0, shard_ctx((yak_ch_t*)other), &tot) : yk_ctx_isec(c0, shard_ctx((yak_ch_t*)other), &tot)) != 0) { ok = false; break; }
		s0->tot = tot;
	}
	for (auto &kv : copy_on) if (kv.second) yak_ch_destroy(kv.second);
	set_tot(h0);
	if (!ok) fprintf(stderr, "[E::%s] %s\n", fn_name, *yakamd_last_error() ? yakamd_last_error() : "the second table could not be brought to the device of the first");
}

void yak_ch_subtract(yak_ch_t *h0, const yak_ch_t *h1, int n_thread) { (void)n_thread; keep_by_membership(h0, h1, 1, __func__); }
void yak_ch_isec(yak_ch_t *h0, const yak_ch_t *h1, int n_thread) { (void)n_thread; keep_by_membership(h0, h1, 2, __func__); }

/* reference htab.c:246-285: every k-mer of h1 with min <= count <= max is put into h0 (its count in
 * h0 goes up by one, saturating; new k-mers start at 1), sub-table by sub-table in h1's slot order;
 * h1 is destroyed.  One counting pass per shard of h0: the list positions are the stream times; a shard of
 * h0 takes the lists of the shards of h1 one after the other (the feed keeps the k-mers of its own prefix range). */
void yak_ch_merge(yak_ch_t *h0, yak_ch_t *h1, int min, int max, int n_thread, int pre_resize)
{
	(void)n_thread;
	const int hi = (max >= min && max <= YAK_MAX_COUNT) ? max : YAK_MAX_COUNT;
	bool ok = h0->k == h1->k && h0->pre == h1->pre;
	if (!ok) fprintf(stderr, "[E::yak_ch_merge] tables of different k / prefix length\n");
	for (yak_ch_t *s0 : shards_of(h0)) {
		if (!ok) break;
		yakamd_ctx *c0 = shard_ctx(s0);
		int lo0, hi0;
		yk_ctx_range(c0, &lo0, &hi0);
		std::vector<yak_ch_t*> from;
		for (yak_ch_t *s1 : shards_of(h1)) { int lo1, hi1; yk_ctx_range(shard_ctx(s1), &lo1, &hi1); if (std::max(lo0, lo1) < std::min(hi0, hi1)) from.push_back(s1); }
		if (pre_resize) for (yak_ch_t *s1 : from) ok = ok && yk_ctx_merge_presize(c0, shard_ctx(s1)) == 0;
		if (!ok) break;
		yk_ctx_gate(c0, false);
		ok = yakamd_pass_begin(s0, 1) == 0;
		uint64_t t0 = 0;
		for (yak_ch_t *s1 : from) {
			u64 *d_hash = 0, n = 0; u32 *d_t = 0;
			ok = ok && hipSetDevice(yk_ctx_device(shard_ctx(s1))) == hipSuccess && yk_ctx_list_hashes(shard_ctx(s1), min, hi, &d_hash, &d_t, &n) == 0;
			const int dv0 = yk_ctx_device(c0), dv1 = yk_ctx_device(shard_ctx(s1));
			if (ok && n && dv0 != dv1) {
				/* the list lies on s1's device and the feed's kernels run on c0's: peer access (enabled here: nothing else in this process may have
				 * done it), or a staged copy on c0's device when the two cannot reach each other */
				int can = 0;
				ok = hipSetDevice(dv0) == hipSuccess;
				if (ok && hipDeviceCanAccessPeer(&can, dv0, dv1) == hipSuccess && can) {
					const hipError_t pe = hipDeviceEnablePeerAccess(dv1, 0);
					if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) can = 0;
					(void)hipGetLastError();
				}
				if (ok && !can) {
					u64 *h2 = (u64*)yakamd_dev_alloc(n * 8); u32 *t2 = (u32*)yakamd_dev_alloc(n * 4);
					ok = h2 && t2 && hipMemcpyPeer(h2, dv0, d_hash, dv1, n * 8) == hipSuccess && hipMemcpyPeer(t2, dv0, d_t, dv1, n * 4) == hipSuccess;
					if (ok) ok = yakamd_feed_hashed_dev(s0, h2, t2, (int64_t)n, t0, n) == 0;
					if (ok) ok = hipStreamSynchronize(yk_ctx_stream(c0)) == hipSuccess;
					yakamd_dev_free(h2); yakamd_dev_free(t2);
					t0 += n;
					yk_pool_release(d_hash); yk_pool_release(d_t);
					continue;
				}
			}
			if (ok && n) ok = yakamd_feed_hashed_dev(s0, d_hash, d_t, (int64_t)n, t0, n) == 0;
			t0 += n;
			yk_pool_release(d_hash); yk_pool_release(d_t);
		}
		if (yakamd_pass_end(s0) < 0) ok = false;                 /* closes the pass whatever the feeds did */
		yk_ctx_gate(c0, true);
		if (ok) s0->tot = yk_ctx_keys_total(c0);                /* htab.c:284: tot = sum of the sub-table sizes */
	}
	set_tot(h0);
	if (!ok) fprintf(stderr, "[E::yak_ch_merge] %s\n", yakamd_last_error());
	yak_ch_destroy(h1);                                          /* htab.c:283: h1 is consumed whatever happened */
}

void yak_ch_hist(const yak_ch_t *h, int64_t cnt[YAK_N_COUNTS], int n_thread) /* reference htab.c:156-169 */
{
	(void)n_thread;
	memset(cnt, 0, YAK_N_COUNTS * sizeof(int64_t));
	if (YK_MULTI((const yak_ch_ext*)h)) {
		const yak_ch_ext *e = (const yak_ch_ext*)h;
		std::vector<int64_t> part(YAK_N_COUNTS);
		for (int r = 0; r < e->n_sub; ++r) { yak_ch_hist(e->sub[r], part.data(), n_thread); for (int i = 0; i < YAK_N_COUNTS; ++i) cnt[i] += part[i]; }
		return;
	}
	if (yk_ctx_hist(((yak_ch_ext*)h)->ctx, cnt)) fprintf(stderr, "[E::yak_ch_hist] %s\n", yakamd_last_error());
}

void yak_ch_setcnt(yak_ch_t *h, int cnt, int n_thread)        /* reference htab.c:219-235: every stored k-mer gets count `cnt` */
{
	(void)n_thread;
	if (YK_MULTI((yak_ch_ext*)h)) { yak_ch_ext *e = (yak_ch_ext*)h; for (int r = 0; r < e->n_sub; ++r) yak_ch_setcnt(e->sub[r], cnt, n_thread); return; }
	if (yk_ctx_setcnt(((yak_ch_ext*)h)->ctx, cnt)) fprintf(stderr, "[E::yak_ch_setcnt] %s\n", yakamd_last_error());
}

static uint64_t hash64_inv(uint64_t x, uint64_t m)           /* reference yak-priv.h:41-68 */
{
	uint64_t t;
	t = x - (x << 31); x = (x - (t << 31)) & m;
	t = x ^ x >> 28; x = x ^ t >> 28;
	x = (x * 14933078535860113213ULL) & m;
	t = x ^ x >> 14; t = x ^ t >> 14; t = x ^ t >> 14; x = x ^ t >> 14;
	x = (x * 15244667743933553977ULL) & m;
	t = x ^ x >> 24; x = x ^ t >> 24;
	t = ~x; t = ~(x - (t << 21)); t = ~(x - (t << 21)); x = ~(x - (t << 21)) & m;
	return x;
}

yak_knt_t *yak_ch_getseq(const yak_ch_t *h, int w, uint32_t *n) /* reference htab.c:353-367 */
{
	assert(h->k < 32 && w < 1 << h->pre);
	if (YK_MULTI((const yak_ch_ext*)h)) { const yak_ch_ext *e = (const yak_ch_ext*)h; return yak_ch_getseq(e->sub[(uint64_t)w * e->n_sub >> h->pre], w, n); }
	*n = 0;
	if (yk_ctx_sync_host(((yak_ch_ext*)h)->ctx, (yak_ch_t*)h)) return 0;
	const yak_ht_t *g = h->h[w].h;
	const uint64_t mask = (1ULL << h->k * 2) - 1;
	yak_knt_t *a = (yak_knt_t*)calloc(g->count ? g->count : 1, sizeof(*a));
	uint32_t j = 0;
	for (uint32_t i = 0, cap = ht_cap(g); i < cap; ++i)
		if (g->used[i >> 5] >> (i & 31) & 1) {
			a[j].x = hash64_inv(g->keys[i] >> YAK_COUNTER_BITS << h->pre | (uint64_t)w, mask);
			a[j++].c = (int)(g->keys[i] & YAK_MAX_COUNT);
		}
	*n = g->count;
	return a;
}

/* ---- .yak serialisation (reference htab.c:373-394): header, then per sub-table capacity, size and
 * the keys in ascending slot order.  Every shard puts the bytes of its own sub-tables together on its device
 * (yk_ctx_dump_image_dev); they come back in one copy (yakamd_dump_mem), or go to the file in 8 MiB pieces through a
 * few page-locked buffers that writer threads pwrite() while the next pieces are on the bus (yak_ch_dump) ---- */
struct DumpSink {
	uint8_t *mem; int fd;                                      /* one of the two */
	bool put(int dev, hipStream_t st, const uint8_t *d_src, size_t bytes, size_t off);
};
bool DumpSink::put(int dev, hipStream_t st, const uint8_t *d_src, size_t bytes, size_t off)
{
	if (bytes == 0) return true;
	if (mem) return hipMemcpyAsync(mem + off, d_src, bytes, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
	enum { NB = 6, W = 3 };
	const size_t CH = (size_t)8 << 20, n_ch = (bytes + CH - 1) / CH;
	static std::mutex mu;                                         /* the staging buffers are the process's: one dump at a time uses them */
	static void *stage[NB] = { 0 };
	std::lock_guard<std::mutex> lk(mu);
	for (int i = 0; i < NB; ++i) if (!stage[i] && hipHostMalloc(&stage[i], CH, hipHostMallocPortable) != hipSuccess) { stage[i] = 0; return false; }
	hipEvent_t ev[NB];
	for (int i = 0; i < NB; ++i) if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return false;
	std::vector<int> issued_v(n_ch, 0), written_v(n_ch, 0);
	volatile int *issued = issued_v.data(), *written = written_v.data();
	bool good = true;
	std::vector<std::thread> th;
	for (int w = 0; w < W; ++w) th.emplace_back([&, w]() {
		(void)hipSetDevice(dev);
		for (size_t j = (size_t)w; j < n_ch; j += W) {
			while (!__atomic_load_n(&issued[j], __ATOMIC_ACQUIRE)) std::this_thread::yield();
			bool ok = __atomic_load_n(&issued[j], __ATOMIC_ACQUIRE) == 1 && hipEventSynchronize(ev[j % NB]) == hipSuccess;
			const size_t n = std::min(CH, bytes - j * CH);
			for (size_t done = 0; ok && done < n; ) {
				const ssize_t r = ::pwrite(fd, (const char*)stage[j % NB] + done, n - done, (off_t)(off + j * CH + done));
				if (r <= 0) ok = false; else done += (size_t)r;
			}
			if (!ok) __atomic_store_n(&good, false, __ATOMIC_RELAXED);
			__atomic_store_n(&written[j], 1, __ATOMIC_RELEASE);
		}
	});
	for (size_t i = 0; i < n_ch; ++i) {
		if (i >= NB) while (!__atomic_load_n(&written[i - NB], __ATOMIC_ACQUIRE)) std::this_thread::yield();
		const size_t n = std::min(CH, bytes - i * CH);
		const bool ok = hipMemcpyAsync(stage[i % NB], d_src + i * CH, n, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(ev[i % NB], st) == hipSuccess;
		__atomic_store_n(&issued[i], ok ? 1 : 2, __ATOMIC_RELEASE);
	}
	for (auto &t : th) t.join();
	(void)hipStreamSynchronize(st);
	for (int i = 0; i < NB; ++i) (void)hipEventDestroy(ev[i]);
	return good;
}

/* the whole .yak image through `sink`; its size, or -1 */
static int64_t dump_through(yak_ch_t *h, DumpSink *sink, bool size_only)
{
	yak_ch_ext *e = (yak_ch_ext*)h;
	const int P = 1 << h->pre, n_sub = YK_MULTI(e) ? e->n_sub : 1;
	size_t sz = 16 + (size_t)8 * P;
	uint32_t cap, cnt;
	for (int r = 0; r < n_sub; ++r) {
		yak_ch_t *hs = YK_MULTI(e) ? e->sub[r] : h;
		const int lo = YK_MULTI(e) ? (int)(((int64_t)r << h->pre) / n_sub) : 0, hi = YK_MULTI(e) ? (int)(((int64_t)(r + 1) << h->pre) / n_sub) : P;
		for (int p = lo; p < hi; ++p) { if (yakamd_subtable(hs, p, &cap, &cnt) != 0) return -1; sz += (size_t)8 * cnt; }
	}
	if (size_only) return (int64_t)sz;
	uint8_t head[16];
	const uint32_t t[3] = { (uint32_t)h->k, (uint32_t)h->pre, YAK_COUNTER_BITS };
	memcpy(head, YAK_MAGIC, 4); memcpy(head + 4, t, 12);
	if (sink->mem) memcpy(sink->mem, head, 16);
	else if (::pwrite(sink->fd, head, 16, 0) != 16) return -1;
	size_t off = 16;
	for (int r = 0; r < n_sub; ++r) {
		yak_ch_t *hs = YK_MULTI(e) ? e->sub[r] : h;
		const int lo = YK_MULTI(e) ? (int)(((int64_t)r << h->pre) / n_sub) : 0, hi = YK_MULTI(e) ? (int)(((int64_t)(r + 1) << h->pre) / n_sub) : P;
		yakamd_ctx *c = ((yak_ch_ext*)hs)->ctx;
		u64 *d_img = 0, n_words = 0;
		if (yk_ctx_dump_image_dev(c, lo, hi, &d_img, &n_words) != 0) return -1;
		const bool ok = sink->put(yk_ctx_device(c), yk_ctx_stream(c), (const uint8_t*)d_img, (size_t)n_words * 8, off);
		(void)hipSetDevice(yk_ctx_device(c));
		yk_pool_release(d_img);
		if (!ok) return -1;
		off += (size_t)n_words * 8;
	}
	return off == sz ? (int64_t)sz : -1;
}

int64_t yakamd_dump_mem(yak_ch_t *h, uint8_t **out)
{
	*out = 0;
	const int64_t sz = dump_through(h, 0, true);
	if (sz < 0) return -1;
	DumpSink sink; sink.mem = (uint8_t*)malloc((size_t)sz); sink.fd = -1;
	if (!sink.mem) return -1;
	if (dump_through(h, &sink, false) != sz) { free(sink.mem); return -1; }
	*out = sink.mem;
	return sz;
}

/* the bytes of sub-tables [lo, hi) alone -- {capacity, size, keys in slot order} each, no header: what one rank of a prefix-sharded job owns of
 * the .yak file (tests and bench.py compare a rank's share with the oracle's without serialising the whole table) */
int64_t yakamd_dump_range_mem(yak_ch_t *h, int lo, int hi, uint8_t **out)
{
	*out = 0;
	yak_ch_ext *e = (yak_ch_ext*)h;
	const int P = 1 << h->pre, n_sub = YK_MULTI(e) ? e->n_sub : 1;
	if (lo < 0 || hi > P || lo >= hi) return -1;
	size_t sz = (size_t)8 * (hi - lo);
	uint32_t cap, cnt;
	auto owner_range = [&](int r, int *a, int *b) {
		*a = YK_MULTI(e) ? (int)(((int64_t)r << h->pre) / n_sub) : 0; *b = YK_MULTI(e) ? (int)(((int64_t)(r + 1) << h->pre) / n_sub) : P;
		*a = std::max(*a, lo); *b = std::min(*b, hi);
	};
	for (int r = 0; r < n_sub; ++r) {
		int a, b;
		owner_range(r, &a, &b);
		yak_ch_t *hs = YK_MULTI(e) ? e->sub[r] : h;
		for (int p = a; p < b; ++p) { if (yakamd_subtable(hs, p, &cap, &cnt) != 0) return -1; sz += (size_t)8 * cnt; }
	}
	DumpSink sink; sink.mem = (uint8_t*)malloc(sz); sink.fd = -1;
	if (!sink.mem) return -1;
	size_t off = 0;
	for (int r = 0; r < n_sub; ++r) {
		int a, b;
		owner_range(r, &a, &b);
		if (a >= b) continue;
		yakamd_ctx *c = ((yak_ch_ext*)(YK_MULTI(e) ? e->sub[r] : h))->ctx;
		u64 *d_img = 0, n_words = 0;
		bool ok = yk_ctx_dump_image_dev(c, a, b, &d_img, &n_words) == 0 && off + (size_t)n_words * 8 <= sz;
		ok = ok && sink.put(yk_ctx_device(c), yk_ctx_stream(c), (const uint8_t*)d_img, (size_t)n_words * 8, off);
		(void)hipSetDevice(yk_ctx_device(c));
		if (d_img) yk_pool_release(d_img);
		if (!ok) { free(sink.mem); return -1; }
		off += (size_t)n_words * 8;
	}
	if (off != sz) { free(sink.mem); return -1; }
	*out = sink.mem;
	return (int64_t)sz;
}

int yak_ch_dump(const yak_ch_t *h, const char *fn)
{
	struct stat sb;
	const bool to_stdout = strcmp(fn, "-") == 0;
	if (to_stdout || (stat(fn, &sb) == 0 && !S_ISREG(sb.st_mode))) {   /* a pipe (or any name that is no regular file) takes the image in one piece */
		FILE *fp = to_stdout ? stdout : fopen(fn, "wb");
		if (fp == 0) return -1;
		uint8_t *buf = 0;
		const int64_t sz = yakamd_dump_mem((yak_ch_t*)h, &buf);
		if (sz < 0) { if (!to_stdout) fclose(fp); return -1; }
		const bool ok = fwrite(buf, 1, (size_t)sz, fp) == (size_t)sz;
		free(buf);
		if ((to_stdout ? fflush(fp) : fclose(fp)) != 0 || !ok) return -1;
	} else {
		const double t0 = yk_realtime();
		DumpSink sink; sink.mem = 0;
		/* fopen(fn, "wb"), htab.c:377 -- except that a file that is there is cut to the new size AFTER it was written over: its pages are reused
		 * instead of being given back and asked for again (0.05 s instead of 0.12 s for 420 MB) */
		sink.fd = ::open(fn, O_WRONLY | O_CREAT, 0666);
		if (sink.fd < 0) return -1;
		const int64_t sz = dump_through((yak_ch_t*)h, &sink, false);
		const bool cut = sz >= 0 && ::ftruncate(sink.fd, (off_t)sz) == 0;
		if (sz < 0) (void)::ftruncate(sink.fd, 0);                  /* a dump that failed midway leaves an empty file, not a header in front of old bytes (fopen "wb", htab.c:377, never keeps any) */
		if (::close(sink.fd) != 0 || !cut) return -1;
		if (getenv("YAKAMD_VERBOSE")) fprintf(stderr, "[yak_amd] dump: %.1f MB in %.3f s\n", sz / 1e6, yk_realtime() - t0);
	}
	fprintf(stderr, "[M::%s] dumpped the hash table to file '%s'.\n", __func__, fn);
	return 0;
}

/* reference htab.c:396-481 (mode YAK_LOAD_ALL): every sub-table is pre-sized to the saved
 * capacity and the keys are put back in file order -- the same staged FCFS replay as shrink */
/* reads a .yak file (htab.c:413-433): header checks, then per sub-table capacity, size and keys */
static bool read_yak(const char *fn, uint32_t hdr[3], std::vector<uint32_t> &caps, std::vector<uint32_t> &sizes, std::vector<uint64_t> &keys)
{
	FILE *fp = fopen(fn, "rb");
	char magic[4];
	if (fp == 0) return false;
	if (fread(magic, 1, 4, fp) != 4) { fclose(fp); return false; }
	if (strncmp(magic, YAK_MAGIC, 4) != 0) { fprintf(stderr, "ERROR: wrong file magic.\n"); fclose(fp); return false; }
	if (fread(hdr, 4, 3, fp) != 3) { fclose(fp); return false; }
	if (hdr[2] != YAK_COUNTER_BITS) {
		fprintf(stderr, "ERROR: saved counter bits: %d; compile-time counter bits: %d\n", hdr[2], YAK_COUNTER_BITS);
		fclose(fp); return false;
	}
	if (hdr[1] < YAK_COUNTER_BITS || hdr[1] > 24) { fclose(fp); return false; }
	const int P = 1 << hdr[1];
	caps.assign(P, 0); sizes.assign(P, 0); keys.clear();
	for (int p = 0; p < P; ++p) {
		uint32_t u[2];
		if (fread(u, 4, 2, fp) != 2) break;
		/* a capacity is 0 or a power of two <= 2^31 (khashl.h:155-158), and it holds its keys at <= 75 % load after the
		 * resize-before-put rule: anything else is a corrupt file */
		if (u[0] > (1u << 31) || (u[0] & (u[0] - 1)) != 0 || u[1] > u[0]) { fprintf(stderr, "ERROR: corrupt sub-table header in '%s' (capacity %u, size %u)\n", fn, u[0], u[1]); fclose(fp); return false; }
		caps[p] = u[0]; sizes[p] = u[1];
		const size_t at = keys.size();
		keys.resize(at + u[1]);
		if (u[1] && fread(&keys[at], 8, u[1], fp) != u[1]) break;
	}
	fclose(fp);
	return true;
}

/* reference htab.c:396-476.  YAK_LOAD_ALL builds the table from the file; the flag modes (trio binning
 * 2 / 3 with min_cnt, mid_cnt; sex chromosomes 4 / 5 / 6) put every selected key with a flag in its low
 * bits, ORing the flag into keys already present -- one counting pass on the device whose records
 * carry the flag in the low 4 bits of their list position. */
yak_ch_t *yak_ch_restore_core(yak_ch_t *ch0, const char *fn, int mode, ...)
{
	int min_cnt = 0, mid_cnt = 0;
	va_list ap;
	va_start(ap, mode);
	if (mode == YAK_LOAD_TRIOBIN1 || mode == YAK_LOAD_TRIOBIN2) { min_cnt = va_arg(ap, int); mid_cnt = va_arg(ap, int); }
	va_end(ap);
	if (mode < YAK_LOAD_ALL || mode > YAK_LOAD_SEXCHR3) return 0;
	if (ch0 == 0 && (mode == YAK_LOAD_TRIOBIN2 || mode == YAK_LOAD_SEXCHR2 || mode == YAK_LOAD_SEXCHR3)) return 0;   /* htab.c:413-420 */
	uint32_t hdr[3];
	std::vector<uint32_t> caps, sizes;
	std::vector<uint64_t> keys;
	if (!read_yak(fn, hdr, caps, sizes, keys)) return 0;
	if (mode == YAK_LOAD_ALL && ch0 == 0) {
		yak_ch_t *h = yak_ch_init((int)hdr[0], (int)hdr[1], 0, 0);
		if (h == 0) return 0;
		if (yk_ctx_load(((yak_ch_ext*)h)->ctx, caps.data(), sizes.data(), keys.data()) != 0) { yak_ch_destroy(h); return 0; }
		fprintf(stderr, "[M::%s] inserted %ld k-mers, of which %ld are new\n", __func__, (long)keys.size(), (long)keys.size());
		return h;
	}
	/* every other case puts the selected keys, in file order, into a table that may already hold some of them
	 * (htab.c:441-470): a flag mode ORs a flag into keys already present, YAK_LOAD_ALL leaves them alone, and a
	 * new key keeps the flag / its saved count.  That is a counting pass whose records carry the payload in the low
	 * bits of their list position (4 bits for a flag, 10 for a count), run in as many passes as the 32-bit
	 * position field needs: a pass meets the keys of the earlier ones as existing state, exactly as the
	 * reference's sequential puts do. */
	if (ch0 && multi_refuse(ch0, __func__)) return 0;           /* a load into a table sharded over prefix ranges: not carried out shard by shard (yet) */
	yak_ch_t *h = ch0 ? ch0 : yak_ch_init((int)hdr[0], (int)hdr[1], 0, 0);
	if (h == 0) return 0;
	assert((int)hdr[0] == h->k && (int)hdr[1] == h->pre);       /* htab.c:437 */
	yakamd_ctx *c = ((yak_ch_ext*)h)->ctx;
	const int P = 1 << h->pre;
	const uint64_t mask = (1ULL << YAK_COUNTER_BITS) - 1;
	const int pbits = mode == YAK_LOAD_ALL ? YAK_COUNTER_BITS : 4;
	size_t per_pass = ((size_t)1 << (32 - pbits)) - 16;
	if (yk_knob("YAKAMD_LOAD_SLICE", 0) > 0) per_pass = std::min<size_t>(per_pass, (size_t)yk_knob("YAKAMD_LOAD_SLICE", 0));   /* tests */
	std::vector<uint64_t> hashes;
	std::vector<uint32_t> times;
	long n_tot = 0, n_new = 0;
	bool ok = yk_ctx_resize_to(c, caps.data()) == 0;             /* htab.c:441 */
	void *d_h = 0, *d_t = 0;
	auto flush = [&]() {
		const size_t n = hashes.size();
		if (n == 0 || !ok) return;
		if (!d_h) { d_h = yakamd_dev_alloc(std::min(per_pass, keys.size()) * 8); d_t = yakamd_dev_alloc(std::min(per_pass, keys.size()) * 4); }
		ok = d_h && d_t && yakamd_memcpy_h2d(d_h, hashes.data(), n * 8) == 0 && yakamd_memcpy_h2d(d_t, times.data(), n * 4) == 0;
		if (ok) {
			yk_ctx_gate(c, false); yk_ctx_or_mode(c, mode == YAK_LOAD_ALL ? 2 : 1);
			ok = yakamd_pass_begin(h, 1) == 0;
			if (ok) {
				ok = yakamd_feed_hashed_dev(h, d_h, d_t, (int64_t)n, 0, (uint64_t)n << pbits) == 0;
				const int64_t r = yakamd_pass_end(h);
				ok = ok && r >= 0;
				if (ok) n_new += (long)r;
			}
			yk_ctx_gate(c, true); yk_ctx_or_mode(c, 0);
		}
		n_tot += (long)n;
		hashes.clear(); times.clear();
	};
	size_t at = 0;
	for (int p = 0; p < P && ok; ++p)
		for (uint32_t j = 0; j < sizes[p] && ok; ++j, ++at) {
			const uint64_t key = keys[at];
			int x;
			if (mode == YAK_LOAD_ALL) x = (int)(key & mask);
			else if (mode == YAK_LOAD_TRIOBIN1 || mode == YAK_LOAD_TRIOBIN2) {
				const int cnt = (int)(key & mask), shift = mode == YAK_LOAD_TRIOBIN1 ? 0 : 2;
				x = cnt >= mid_cnt ? 2 << shift : cnt >= min_cnt ? 1 << shift : -1;
			} else x = 1 << (mode - YAK_LOAD_SEXCHR1);
			if (x < 0) continue;
			hashes.push_back((key >> YAK_COUNTER_BITS) << h->pre | (uint64_t)p);
			times.push_back((uint32_t)(hashes.size() - 1) << pbits | (uint32_t)x);
			if (hashes.size() >= per_pass) flush();
		}
	flush();
	yakamd_dev_free(d_h); yakamd_dev_free(d_t);
	if (!ok) { fprintf(stderr, "[E::%s] %s\n", __func__, yakamd_last_error()); if (!ch0) yak_ch_destroy(h); return 0; }
	fprintf(stderr, "[M::%s] inserted %ld k-mers, of which %ld are new\n", __func__, n_tot, n_new);
	return h;
}

yak_ch_t *yak_ch_restore(const char *fn) { return yak_ch_restore_core(0, fn, YAK_LOAD_ALL); }   /* reference htab.c:478 */

/* ------------------------------------------------------------------------------------------
 * FASTA/FASTQ record reader with the observable behaviour of the reference's parser as driven by
 * count.c:88-110 (record grammar of kseq.h:192-232): header lines start with '>' or '@', the
 * sequence is every following line up to one starting with '>', '@' or '+', a '+' line introduces
 * quality lines covering at least the sequence length; a truncated quality ends the input.
 * ------------------------------------------------------------------------------------------ */
extern "C++" {
/* What the parallel parser reads: a plain file, or the uncompressed stream of a BGZF file (block gzip: every member carries its
 * compressed size in a 'BC' extra field and holds <= 64 KiB of data, so members can be found without inflating and inflated
 * independently).  A read at any offset inflates just the blocks it touches, on the calling thread -- the parser's threads each
 * read their own segment, so inflation is spread over them by itself.  libdeflate is used when the image has it, else zlib. */
struct ByteSource {
	struct Blk { int64_t foff, uoff; uint32_t csize, usize; };   /* offset of the deflate payload, offset in the uncompressed stream, bytes of both */
	int fd; int64_t size; bool bgzf; std::vector<Blk> blk;
	uint64_t gen;                                                 /* identity of this source for the per-thread block cache (an address can be reused by the next job's source) */
	const unsigned char *map; size_t map_len;                     /* a plain file, mapped: the body of a long FASTA record is stripped of its line ends by several threads straight from here */
	bool pack;                                                    /* the parser threads also pack what they parsed (yakamd_pack_bases_host) */
	bool in_memory, partial;                                      /* bytes in memory (a batch of an inflated gzip stream; map is not ours); more of the stream follows them: a record that touches their end is not finished */
	static uint64_t next_gen() { static uint64_t g = 0; return __atomic_add_fetch(&g, 1, __ATOMIC_RELAXED); }
	ByteSource() : fd(-1), size(0), bgzf(false), gen(next_gen()), map(0), map_len(0), pack(false), in_memory(false), partial(false) {}
	~ByteSource() { if (map && !in_memory) munmap((void*)map, map_len); }
	void set_memory(const unsigned char *p, size_t n, bool more_follows) { fd = -1; bgzf = false; map = p; map_len = n; size = (int64_t)n; in_memory = true; partial = more_follows; }
	ByteSource(const ByteSource&) = delete; ByteSource &operator=(const ByteSource&) = delete;
	void map_plain() {
		if (bgzf || fd < 0 || size <= 0 || map) return;
		void *m = mmap(0, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
		if (m != MAP_FAILED) { map = (const unsigned char*)m; map_len = (size_t)size; (void)madvise(m, map_len, MADV_SEQUENTIAL); }
	}
	typedef void *(*ld_alloc_t)(void); typedef int (*ld_dec_t)(void*, const void*, size_t, void*, size_t, size_t*); typedef void (*ld_free_t)(void*);
	static void ld_api(ld_alloc_t *al, ld_dec_t *de, ld_free_t *fr = 0) {
		static ld_alloc_t a = 0; static ld_dec_t d = 0; static ld_free_t f = 0; static bool tried = false;
		if (!tried) {                                              /* benign race: every thread resolves the same pointers */
			void *l = yk_knob("YAKAMD_NO_LIBDEFLATE", 0) ? 0 : dlopen("libdeflate.so.0", RTLD_NOW);
			if (l) { a = (ld_alloc_t)dlsym(l, "libdeflate_alloc_decompressor"); d = (ld_dec_t)dlsym(l, "libdeflate_deflate_decompress"); f = (ld_free_t)dlsym(l, "libdeflate_free_decompressor"); }
			if (!a || !d) { a = 0; d = 0; f = 0; }
			tried = true;
		}
		*al = a; *de = d; if (fr) *fr = f;
	}
	/* per-thread inflate state, released when the thread ends (the parser starts fresh threads for every window) */
	struct LdState { void *dec; ld_free_t fr; LdState() : dec(0), fr(0) {} ~LdState() { if (dec && fr) fr(dec); } };
	struct ZState { z_stream zs; bool init; ZState() : init(false) { memset(&zs, 0, sizeof(zs)); } ~ZState() { if (init) inflateEnd(&zs); } };
	/* index the members of an open file; false if it is not BGZF from the first byte to the last */
	bool index_bgzf(int f) {
		struct stat sb;
		if (fstat(f, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
		int64_t off = 0, uoff = 0;
		unsigned char h[18], t[4];
		blk.clear();
		while (off < sb.st_size) {
			if (::pread(f, h, 18, off) != 18) return false;
			if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
			const uint32_t xlen = h[10] | h[11] << 8;
			/* the 'BC' subfield is the first one in every BGZF writer; anything else is not indexed */
			if (xlen < 6 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0 || (h[3] & ~4)) return false;
			const uint32_t bsize = (h[16] | h[17] << 8) + 1u;
			if (bsize < 12 + xlen + 8 || off + bsize > sb.st_size) return false;
			if (::pread(f, t, 4, off + bsize - 4) != 4) return false;
			const uint32_t isize = t[0] | t[1] << 8 | t[2] << 16 | (uint32_t)t[3] << 24;
			if (isize > 65536) return false;
			Blk b; b.foff = off + 12 + xlen; b.csize = bsize - 12 - xlen - 8; b.uoff = uoff; b.usize = isize;
			if (isize) blk.push_back(b);
			off += bsize; uoff += isize;
		}
		fd = f; size = uoff; bgzf = true;
		return true;
	}
	bool inflate_block(const Blk &b, unsigned char *out, std::vector<unsigned char> &cbuf) const {
		cbuf.resize(b.csize + 8);
		size_t got = 0;
		while (got < b.csize + 8) { const ssize_t r = ::pread(fd, cbuf.data() + got, b.csize + 8 - got, b.foff + got); if (r <= 0) return false; got += r; }
		ld_alloc_t al; ld_dec_t de; ld_api(&al, &de);
		bool ok = false;
		if (al) {
			static thread_local LdState st;
			if (!st.dec) { st.dec = al(); ld_free_t fr = 0; ld_api(&al, &de, &fr); st.fr = fr; }
			size_t n = 0;
			ok = st.dec && de(st.dec, cbuf.data(), b.csize, out, b.usize, &n) == 0 && n == b.usize;
		} else {
			static thread_local ZState st;
			z_stream &zs = st.zs;
			if (!st.init) { if (inflateInit2(&zs, -15) != Z_OK) return false; st.init = true; } else inflateReset(&zs);
			zs.next_in = cbuf.data(); zs.avail_in = b.csize; zs.next_out = out; zs.avail_out = b.usize;
			ok = inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
		}
		if (!ok) return false;
		const unsigned char *t = cbuf.data() + b.csize;            /* CRC32 of the uncompressed data, as gzread would check it */
		const uint32_t crc = t[0] | t[1] << 8 | t[2] << 16 | (uint32_t)t[3] << 24;
		return (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, b.usize) == crc;
	}
	/* pread(2) semantics on the uncompressed stream; -1 on a corrupt block */
	ssize_t pread_at(void *dst, size_t n, int64_t off) const {
		if (in_memory) { if (off >= size || n == 0) return 0; const size_t take = std::min<size_t>(n, (size_t)(size - off)); memcpy(dst, map + off, take); return (ssize_t)take; }
		if (!bgzf) return ::pread(fd, dst, n, off);
		if (off >= size || n == 0) return 0;
		static thread_local std::vector<unsigned char> ub, cb;
		static thread_local uint64_t who = 0; static thread_local size_t which = (size_t)-1;
		size_t lo = 0, hi = blk.size();                            /* the block that holds `off` */
		while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (blk[mid].uoff <= off) lo = mid; else hi = mid; }
		size_t done = 0;
		for (size_t bi = lo; bi < blk.size() && done < n; ++bi) {
			const Blk &b = blk[bi];
			if (who != gen || which != bi) {
				ub.resize(65536);
				if (!inflate_block(b, ub.data(), cb)) { who = 0; return -1; }
				who = gen; which = bi;
			}
			const size_t skip = (size_t)(off + (int64_t)done - b.uoff), take = std::min<size_t>(b.usize - skip, n - done);
			memcpy((char*)dst + done, ub.data() + skip, take);
			done += take;
		}
		return (ssize_t)done;
	}
};

struct FxReader {
	gzFile fp; int fd; unsigned char *buf; int beg, end, eof, last;
	std::vector<char> seq, name; size_t qlen; int qlast;
	enum { BUF = 1 << 20, NOT_FAST = -3 };
	FxReader() : fp(0), fd(-1), buf(0), beg(0), end(0), eof(0), last(0), qlen(0), qlast(0), mem(false), psrc(0), poff(0), pos0(0) {}
	/* open `fn` (NULL or "-": stdin); a plain (not gzip) regular file is then read with read(2), skipping zlib's copy */
	bool open_file(const char *fn) {
		const bool is_stdin = fn == 0 || strcmp(fn, "-") == 0;
		fp = is_stdin ? gzdopen(0, "r") : gzopen(fn, "r");
		if (fp == 0) return false;
		gzbuffer(fp, 1 << 20);                               /* zlib's default 8 KB means a read() per 8 KB */
		if (!is_stdin && gzdirect(fp)) fd = ::open(fn, O_RDONLY);
		buf = (unsigned char*)malloc(BUF);
		return true;
	}
	void close_file() { if (fd >= 0) ::close(fd); if (fp) gzclose(fp); if (!mem) free(buf); fp = 0; fd = -1; buf = 0; }
	/* positional mode for the parallel parser: read a shared source (plain file or BGZF stream) from offset `from` */
	bool mem; const ByteSource *psrc; int64_t poff, pos0;
	void open_at(const ByteSource *src, int64_t from) { psrc = src; poff = pos0 = from; buf = (unsigned char*)malloc(BUF); beg = end = 0; eof = 0; last = 0; }
	void close_at() { free(buf); buf = 0; }
	/* consume up to the next record marker ('>' or '@', kseq.h:196-199) so that `last` holds it; false at EOF */
	bool seek_marker() {
		if (last != 0) return true;
		int c;
		while ((c = getc()) != -1 && c != '>' && c != '@') {}
		if (c == -1) return false;
		last = c;
		return true;
	}
	int64_t marker_pos() const { return pos0 + (last != 0 ? beg - 1 : end); }   /* file offset of the marker `last` was read from (positional mode) */
	bool fill() {
		if (beg < end) return true;
		if (eof) return false;
		beg = 0;
		if (psrc) {
			pos0 = poff; end = 0;
			while (end < BUF) { const ssize_t r = psrc->pread_at(buf + end, BUF - end, poff); if (r <= 0) break; end += (int)r; poff += r; }
		} else if (fd >= 0) {                                /* read(2) may return short counts before EOF */
			end = 0;
			while (end < BUF) { const ssize_t r = ::read(fd, buf + end, BUF - end); if (r <= 0) break; end += (int)r; }
		} else end = gzread(fp, buf, BUF);
		if (end < BUF) eof = 1;
		if (end <= 0) { end = 0; return false; }
		return true;
	}
	int getc() { return fill() ? buf[beg++] : -1; }
	/* consume through the next delimiter; what: 0 discard, 1 append to seq, 2 count quality bytes, 3 append to name */
	int until(bool line, int what, int *dret) {
		if (dret) *dret = 0;
		if (beg >= end && eof) return -1;
		while (fill()) {
			int i = beg;
			if (line) { const unsigned char *q = (const unsigned char*)memchr(buf + beg, '\n', end - beg); i = q ? (int)(q - buf) : end; }
			else while (i < end && !isspace(buf[i])) ++i;
			if (what == 1) seq.insert(seq.end(), buf + beg, buf + i);
			else if (what == 3) name.insert(name.end(), buf + beg, buf + i);
			else if (what == 2 && i > beg) { qlen += i - beg; qlast = buf[i - 1]; }
			const bool hit = i < end;
			if (hit && dret) *dret = buf[i];
			beg = i + 1;
			if (hit) break;
		}
		if (line && what == 1 && seq.size() > 1 && seq.back() == '\r') seq.pop_back();
		if (line && what == 2 && qlen > 1 && qlast == '\r') { --qlen; qlast = 0; }
		return 0;
	}
	/* Fast path for the record shapes real files are made of -- header line, ONE sequence line, and
	 * either the next record's marker (FASTA) or a '+' line and ONE quality line of the same length
	 * (FASTQ) -- when all of it, plus the byte after it, already sits in the buffer: located with
	 * memchr and appended to `out` (sequence + '\n') straight from the buffer if it has >= min_len
	 * bases.  The reader state it leaves is exactly what next() would leave; anything else (blank or
	 * wrapped lines, CR, a record cut by the buffer end, EOF) returns NOT_FAST without touching the
	 * state, and the caller takes next(). */
	template <class V> int64_t fast(V &out, int64_t min_len) {
		const unsigned char *b = buf;
		int p = beg;
		if (p >= end) return NOT_FAST;
		if (last == 0) { if (b[p] != '@' && b[p] != '>') return NOT_FAST; ++p; }
		const unsigned char *q = (const unsigned char*)memchr(b + p, '\n', end - p);
		if (!q) return NOT_FAST;
		const int s0 = (int)(q - b) + 1;
		if (s0 >= end) return NOT_FAST;
		const int c0 = b[s0];
		if (c0 == '\n' || c0 == '>' || c0 == '+' || c0 == '@') return NOT_FAST;
		q = (const unsigned char*)memchr(b + s0, '\n', end - s0);
		if (!q) return NOT_FAST;
		const int s1 = (int)(q - b), slen = s1 - s0, n0 = s1 + 1;
		if (b[s1 - 1] == '\r' || n0 >= end) return NOT_FAST;
		int nbeg, nlast;
		if (b[n0] == '>' || b[n0] == '@') { nlast = b[n0]; nbeg = n0 + 1; }
		else if (b[n0] == '+') {
			q = (const unsigned char*)memchr(b + n0, '\n', end - n0);
			if (!q) return NOT_FAST;
			const int q0 = (int)(q - b) + 1;
			if ((int64_t)q0 + slen + 1 >= end) return NOT_FAST;
			if (b[q0 + slen] != '\n' || memchr(b + q0, '\n', slen)) return NOT_FAST;
			if (slen > 1 && b[q0 + slen - 1] == '\r') return NOT_FAST;
			nlast = 0; nbeg = q0 + slen + 1;
		} else return NOT_FAST;
		if (slen >= min_len) { out.insert(out.end(), b + s0, b + s1); out.push_back('\n'); }
		beg = nbeg; last = nlast;
		return slen;
	}
	/* The body of a long FASTA record (a chromosome: 1.7 M lines), from a mapped plain file: `n_thr` threads each take a range of the bytes from
	 * `from` on, walk the lines that START in their range -- a line that begins with '>', '@' or '+' ends the body (kseq.h:209) -- and count the
	 * bytes the lines contribute (kseq.h:145: a '\r' before the line end is dropped; the sequence is longer than one byte here); then every thread
	 * copies its lines to their place in `out`.  Returns the offset where the body scan stopped (a line start: the marker line, or the end of
	 * the span / file); out grows by the body's bytes.  `from` must be a line start */
	template <class V> int64_t bulk_body(V &out, int64_t from, int n_thr) {
		const unsigned char *m = psrc->map;
		if (n_thr > 64) n_thr = 64;
		/* a span of 4 MB per thread: a body of 100 MB then keeps every thread busy for several spans (with one span of 1 GB cut into n_thr parts the
		 * first three parts held the whole body and the other threads walked the records behind it for nothing) */
		const int64_t fend = (int64_t)psrc->map_len, span_end = std::min<int64_t>(fend, from + std::max<int64_t>((int64_t)8 << 20, (int64_t)n_thr << 22));
		const int64_t step = (span_end - from + n_thr - 1) / n_thr;
		struct Part { int64_t a, stop, bytes; bool hit; };
		std::vector<Part> part(n_thr);
		int64_t first_hit = INT64_MAX;                                  /* where the body was seen to end: the threads behind it stop counting (their part is not the body's) */
		auto walk = [&](int t, char *dst) {                             /* dst == 0: count; else copy */
			Part &P = part[t];
			int64_t p = P.a;
			const int64_t lim = dst ? P.stop : std::min<int64_t>(span_end, from + (int64_t)(t + 1) * step);
			int64_t nb = 0;
			bool hit = false;
			unsigned n_line = 0;
			while (p < lim) {                                          /* p is a line start */
				const unsigned char c = m[p];
				if (c == '>' || c == '@' || c == '+') {
					hit = true;
					if (!dst) { int64_t cur = __atomic_load_n(&first_hit, __ATOMIC_RELAXED); while (p < cur && !__atomic_compare_exchange_n(&first_hit, &cur, p, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
					break;
				}
				if (!dst && (++n_line & 255) == 0 && P.a > __atomic_load_n(&first_hit, __ATOMIC_RELAXED)) break;
				const unsigned char *q = (const unsigned char*)memchr(m + p, '\n', (size_t)(fend - p));
				const int64_t e = q ? (int64_t)(q - m) : fend;
				int64_t len = e - p;
				if (len > 0 && m[e - 1] == '\r') --len;
				if (dst) memcpy(dst + nb, m + p, (size_t)len);
				nb += len;
				p = q ? e + 1 : fend;
			}
			if (!dst) { P.stop = p; P.bytes = nb; P.hit = hit; }
		};
		std::vector<std::thread> th;
		for (int t = 0; t < n_thr; ++t) {
			int64_t a = from + (int64_t)t * step;
			if (t > 0 && a < span_end) {                                /* first line start at or behind the cut */
				const unsigned char *q = (const unsigned char*)memchr(m + a - 1, '\n', (size_t)(fend - (a - 1)));
				a = q ? (int64_t)(q - m) + 1 : fend;
			}
			part[t].a = std::min(a, span_end); part[t].stop = part[t].a; part[t].bytes = 0; part[t].hit = false;
		}
		for (int t = 1; t < n_thr; ++t) th.emplace_back(walk, t, (char*)0);
		walk(0, 0);
		for (auto &x : th) x.join();
		th.clear();
		/* a part whose first line lies beyond its range walked nothing; the body ends at the first marker.  Parts tile the span: part t stops
		 * where part t + 1 starts, unless a marker stopped it */
		int n_use = 0;
		int64_t total = 0, stop = part[0].a;
		std::vector<int64_t> off(n_thr, 0);
		for (int t = 0; t < n_thr; ++t) {
			if (part[t].a != stop) break;                               /* (a long line swallowed this part's range) */
			off[t] = total; total += part[t].bytes; stop = part[t].stop; ++n_use;
			if (part[t].hit) break;
		}
		const size_t at = out.size();
		out.resize(at + (size_t)total);
		char *base = &out[0] + at;
		for (int t = 1; t < n_use; ++t) th.emplace_back(walk, t, base + off[t]);
		if (n_use > 0) walk(0, base + off[0]);
		for (auto &x : th) x.join();
		return stop;
	}
	/* next(), with the sequence appended to `out` (+ '\n') when it has >= min_len bytes, and long FASTA bodies of a mapped file stripped by
	 * bulk_threads threads.  Same return values and reader state as next() */
	template <class V> int64_t next_to(V &out, int64_t min_len, int bulk_threads) {
		if (!(psrc && psrc->map && bulk_threads > 1)) {
			const int64_t l = next();
			if (l >= min_len) { out.insert(out.end(), seq.begin(), seq.end()); out.push_back('\n'); }
			return l;
		}
		int c, d;
		if (last == 0) {
			while ((c = getc()) != -1 && c != '>' && c != '@') {}
			if (c == -1) return -1;
			last = c;
		}
		seq.clear(); name.clear(); qlen = 0; qlast = 0;
		if (until(false, 3, &d) < 0) return -1;
		if (d != '\n') until(true, 0, 0);
		const size_t at0 = out.size();
		int64_t bulked = 0;
		while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;
			seq.push_back((char)c);
			until(true, 1, 0);
			if (seq.size() >= ((size_t)1 << 20)) {                       /* a long body: what is read so far goes out, the rest in parallel from the map */
				out.insert(out.end(), seq.begin(), seq.end());
				bulked += (int64_t)seq.size();
				seq.clear();
				int64_t p = pos0 + beg;                                 /* the reader stands at a line start (or at the end of the file) */
				for (;;) {
					const size_t before = out.size();
					/* room for the spans to come in one step (a vector that grows span by span copies the whole body again and again, on one thread:
					 * a quarter of the time of a 100 Mb record); address space only until it is written */
					const size_t ahead = (size_t)std::min<int64_t>((int64_t)psrc->map_len - p, (int64_t)1 << 30) + ((size_t)1 << 20);
					if (out.capacity() - before < std::min<size_t>(ahead, (size_t)bulk_threads << 22)) { try { out.reserve(before + ahead); } catch (const std::bad_alloc&) {} }
					const int64_t q = bulk_body(out, p, bulk_threads);
					bulked += (int64_t)(out.size() - before);
					const bool more = q > p && q < (int64_t)psrc->map_len && psrc->map[q] != '>' && psrc->map[q] != '@' && psrc->map[q] != '+';   /* the span ended before the body did */
					p = q;
					if (!more) break;
				}
				beg = end = 0; eof = 0; poff = p;                       /* the buffered reader goes on from there */
				seq.push_back('x'); seq.push_back('x');                 /* (kseq.h:145 looks at the sequence's length: "more than one byte" stays true) */
			}
		}
		const int64_t slen = bulked ? bulked + (int64_t)seq.size() - 2 : (int64_t)seq.size();
		if (bulked) { out.insert(out.end(), seq.begin() + 2, seq.end()); }
		else if ((int64_t)seq.size() >= min_len) out.insert(out.end(), seq.begin(), seq.end());
		if (c == '>' || c == '@') last = c;
		int64_t ret = slen;
		if (c == '+') {
			while ((c = getc()) != -1 && c != '\n') {}
			if (c == -1) ret = -2;
			else {
				while (until(true, 2, 0) >= 0 && (int64_t)qlen < slen) {}
				last = 0;
				if ((int64_t)qlen != slen) ret = -2;
			}
		}
		if (ret >= min_len) out.push_back('\n'); else out.resize(at0);
		return ret;
	}
	int64_t next() {
		int c, d;
		if (last == 0) {
			while ((c = getc()) != -1 && c != '>' && c != '@') {}
			if (c == -1) return -1;
			last = c;
		}
		seq.clear(); name.clear(); qlen = 0; qlast = 0;
		if (until(false, 3, &d) < 0) return -1;
		if (d != '\n') until(true, 0, 0);
		while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;
			seq.push_back((char)c);
			until(true, 1, 0);
		}
		if (c == '>' || c == '@') last = c;
		if (c != '+') return (int64_t)seq.size();
		while ((c = getc()) != -1 && c != '\n') {}
		if (c == -1) return -2;
		while (until(true, 2, 0) >= 0 && qlen < seq.size()) {}
		last = 0;
		return qlen == seq.size() ? (int64_t)seq.size() : -2;
	}
};
} /* extern "C++" */

/* ------------------------------------------------------------------------------------------
 * Parallel parsing of a plain (uncompressed, mapped) file.  A window of the file is cut into one
 * segment per thread.  Segment 0 starts at a verified record boundary; the others start at a GUESS
 * (first line after the cut that begins with '>' or, for '@', whose third line begins with '+').
 * Every thread parses records with the ordinary reader until the next record would start at or
 * beyond its segment's end and reports where that is.  A segment's output is accepted only if the
 * previous accepted segment stopped exactly at its start -- so the accepted stream is, by induction,
 * what the single reader would have produced; the next window starts where the last accepted
 * segment stopped.  A wrong guess costs time, never correctness.
 * ------------------------------------------------------------------------------------------ */
static int64_t env_threads_window() { const int64_t w = yk_knob("YAKAMD_PARSE_WINDOW", 0); return w > 0 ? w : 0; }
static int parse_threads(int n_thread)
{
	const char *e = getenv("YAKAMD_PARSE_THREADS");
	int n = e ? atoi(e) : n_thread;
	const int hw = (int)std::thread::hardware_concurrency();
	if (hw > 0 && n > hw) n = hw;
	return n < 1 ? 1 : n > 32 ? 32 : n;
}

/* The parsed base images' buffers (reused from window to window; pageable: page-locking them cost more than the runtime's staged copies of
 * pageable memory -- CLI run 1.44 s against 2.2 s -- and the multi-GPU reader stages through its own two pinned buffers) */
extern "C++" {
template <class T> struct PinAlloc {
	typedef T value_type;
	PinAlloc() {}
	template <class U> PinAlloc(const PinAlloc<U>&) {}
	T *allocate(size_t n) {
		void *p = malloc(n * sizeof(T) + 16);
		if (!p) throw std::bad_alloc();
		return (T*)p;
	}
	void deallocate(T *p, size_t) { free((void*)p); }
	template <class U> void construct(U*) {}                        /* resize() leaves new bytes alone: they are written right away (no zero fill of a 100 MB sequence) */
	template <class U, class A0> void construct(U *p, const A0 &a) { ::new ((void*)p) U(a); }
	template <class U> bool operator==(const PinAlloc<U>&) const { return true; }
	template <class U> bool operator!=(const PinAlloc<U>&) const { return false; }
};
typedef std::vector<char, PinAlloc<char> > PinVec;
} /* extern "C++" */

/* ---- the base image packed on the host (include/yak_amd.h: yakamd_feed_packed_dev's format): 2-bit codes, 16 bases per 32-bit word, and one
 * validity bit per base, by the table the kernels use (seq_nt4_table, reference yak.h / count.c:28-31: ACGT, acgt, U, u and the bytes 0..3 are
 * bases, everything else -- N, the '\n' between two records -- is not) ---- */
static const uint8_t yk_nt4[256] = {
#define R4(v) v, v, v, v
#define R16(v) R4(v), R4(v), R4(v), R4(v)
	0, 1, 2, 3, R4(4), R4(4), R4(4),
	R16(4), R16(4), R16(4),
	4, 0, 4, 1, 4, 4, 4, 2, R4(4), R4(4),
	4, 4, 4, 4, 3, 3, 4, 4, R4(4), R4(4),
	4, 0, 4, 1, 4, 4, 4, 2, R4(4), R4(4),
	4, 4, 4, 4, 3, 3, 4, 4, R4(4), R4(4),
	R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4)
#undef R16
#undef R4
};
static inline void pack32_scalar(const uint8_t *a, int64_t left, uint32_t *c0, uint32_t *c1, uint32_t *v)
{
	uint32_t x0 = 0, x1 = 0, m = 0;
	const int n = left < 32 ? (int)left : 32;
	for (int j = 0; j < n; ++j) {
		const uint32_t c = yk_nt4[a[j]];
		if (c < 4) { m |= 1u << j; if (j < 16) x0 |= c << (2 * j); else x1 |= c << (2 * (j - 16)); }
	}
	*c0 = x0; *c1 = x1; *v = m;
}
/* 32 bases per step with AVX2 + BMI2: A / C / G / T of either case by four compares (the validity word is their movemask), the code of such a
 * byte is bits 1..2 of it with bit 1 flipped when bit 2 is set (A 0x41 -> 0, C 0x43 -> 1, G 0x47 -> 2, T 0x54 -> 3), gathered by pext; a group
 * that holds one of the rare other bases (U, u, a raw 0..3) goes through the table */
__attribute__((target("avx2,bmi2")))
static void pack_words_avx2(const uint8_t *a, int64_t n_words, uint32_t *codes, uint32_t *valid)
{
	const __m256i up = _mm256_set1_epi8((char)0xDF), A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'), G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T'),
	              U = _mm256_set1_epi8('U'), four = _mm256_set1_epi8(4), b4 = _mm256_set1_epi8(0x04);
	for (int64_t w = 0; w < n_words; ++w, a += 32) {
		const __m256i x = _mm256_loadu_si256((const __m256i*)a), u = _mm256_and_si256(x, up);
		const __m256i acgt = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, A), _mm256_cmpeq_epi8(u, C)), _mm256_or_si256(_mm256_cmpeq_epi8(u, G), _mm256_cmpeq_epi8(u, T)));
		const __m256i rare = _mm256_or_si256(_mm256_cmpeq_epi8(u, U), _mm256_cmpeq_epi8(_mm256_min_epu8(x, four), x) /* x <= 4 */);
		const __m256i rare4 = _mm256_andnot_si256(_mm256_cmpeq_epi8(x, four), rare);   /* x < 4, or U / u */
		if (_mm256_movemask_epi8(rare4)) { pack32_scalar(a, 32, &codes[2 * w], &codes[2 * w + 1], &valid[w]); continue; }
		const uint32_t m = (uint32_t)_mm256_movemask_epi8(acgt);
		const __m256i y = _mm256_xor_si256(x, _mm256_srli_epi16(_mm256_and_si256(x, b4), 1));
		uint64_t q[4];
		_mm256_storeu_si256((__m256i*)q, y);
		const uint64_t sel = 0x0606060606060606ull;
		const uint64_t code = _pext_u64(q[0], sel) | _pext_u64(q[1], sel) << 16 | _pext_u64(q[2], sel) << 32 | _pext_u64(q[3], sel) << 48;
		const uint64_t keep = _pdep_u64((uint64_t)m, 0x5555555555555555ull) * 3;
		const uint64_t cv = code & keep;
		codes[2 * w] = (uint32_t)cv; codes[2 * w + 1] = (uint32_t)(cv >> 32); valid[w] = m;
	}
}
/* n bases -> (n + 31) / 32 words of validity bits and twice as many of codes; the bits behind base n - 1 in the last words are zero */
static void pack_into(const uint8_t *a, int64_t n, uint32_t *codes, uint32_t *valid)
{
	const int64_t nw = (n + 31) / 32, whole = n / 32;
	static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !yk_knob("YAKAMD_NO_AVX2", 0);
	int64_t w = 0;
	if (wide) { pack_words_avx2(a, whole, codes, valid); w = whole; }
	for (; w < nw; ++w) pack32_scalar(a + 32 * w, n - 32 * w, &codes[2 * w], &codes[2 * w + 1], &valid[w]);
}
void yakamd_pack_bases_host(const void *ascii, int64_t n, void *h_packed)
{
	if (n <= 0) return;
	const int64_t nw = (n + 31) / 32;
	uint32_t *codes = (uint32_t*)h_packed, *valid = (uint32_t*)((char*)h_packed + ((nw * 8 + 15) & ~(int64_t)15));
	pack_into((const uint8_t*)ascii, n, codes, valid);
	const int64_t pad = (((nw * 8 + 15) & ~(int64_t)15) - nw * 8) / 4;
	for (int64_t i = 0; i < pad; ++i) codes[2 * nw + i] = 0;
}

typedef std::vector<uint32_t, PinAlloc<uint32_t> > PinWords;
/* a parsed segment: its base image -- or, when the source asks for packed pieces, that image packed as it grows (the ASCII bytes then only pass
 * through a buffer of ~1 MB that stays in the thread's cache): code words, validity words, 32 stream positions per validity word.  The image
 * of a segment is taken to end at a multiple of 32 positions: the up to 31 positions behind it are no bases (stream positions only order the
 * k-mers; a few of them unused change nothing), so a window's segments can be laid one behind the other word by word */
struct ParSeg { int64_t start, end, stop; PinVec img; PinWords codes, valid; int64_t n_seq, sum_len; bool hard_end; };
/* the accepted segments of a window as packed pieces, in stream order (yakamd_feed_packed_pieces_host lays them out on the device: one feed) */
struct WinPack { std::vector<const void*> codes, valid; std::vector<int64_t> n_words; int64_t n_pos, n_seq; WinPack() : n_pos(0), n_seq(0) {} };
/* what takes the parsed pieces, in stream order: the base image (sequences, each followed by '\n'), its bytes, its sequences, and -- when the
 * source asked for it (ByteSource::pack) -- no ASCII image but the packed pieces of a whole window (n = its stream positions), else 0 */
typedef std::function<bool(const char*, size_t, int64_t, const WinPack*)> ImgSink;

static int64_t guess_record_start(const ByteSource *src, int64_t from, int64_t limit)
{
	std::vector<unsigned char> tmp((size_t)(limit - from));
	int64_t got = 0;
	while (got < (int64_t)tmp.size()) { const ssize_t r = src->pread_at(tmp.data() + got, tmp.size() - got, from + got); if (r <= 0) break; got += r; }
	const unsigned char *base = tmp.data(), *p = base, *e = base + got;
	p = (const unsigned char*)memchr(p, '\n', e - p);
	if (!p) return -1;
	for (++p; p < e; ) {
		const unsigned char *l1 = (const unsigned char*)memchr(p, '\n', e - p);
		if (*p == '>') return from + (p - base);
		if (*p == '@' && l1) {
			const unsigned char *l2 = l1 + 1 < e ? (const unsigned char*)memchr(l1 + 1, '\n', e - (l1 + 1)) : 0;
			if (l2 && l2 + 1 < e && l2[1] == '+') return from + (p - base);
		}
		if (!l1) return -1;
		p = l1 + 1;
	}
	return -1;
}

static void parse_segment(const ByteSource *src, int64_t file_end, ParSeg *sg, int min_len, int bulk_threads)
{
	FxReader r;
	r.open_at(src, sg->start);
	sg->n_seq = sg->sum_len = 0; sg->hard_end = false;
	sg->img.clear(); sg->codes.clear(); sg->valid.clear();
	const bool pack = src->pack;
	if (!pack && sg->img.capacity() < (size_t)(sg->end - sg->start)) sg->img.reserve((size_t)(sg->end - sg->start) + (1 << 16));   /* the sequences are a part of the segment's bytes */
	if (pack) { const size_t w = (size_t)(sg->end - sg->start) / 32 + 64; if (sg->valid.capacity() < w) { sg->valid.reserve(w); sg->codes.reserve(2 * w); } }
	auto flush = [&](bool all) {                                  /* whole words of the staged bases go to the packed image; at the end the rest too, padded */
		const size_t n = all ? sg->img.size() : sg->img.size() & ~(size_t)31;
		if (n == 0) return;
		const size_t w0 = sg->valid.size(), nw = (n + 31) / 32;
		sg->valid.resize(w0 + nw); sg->codes.resize(2 * (w0 + nw));
		pack_into((const uint8_t*)sg->img.data(), (int64_t)n, &sg->codes[2 * w0], &sg->valid[w0]);
		const size_t rest = sg->img.size() - n;
		if (rest) memmove(&sg->img[0], &sg->img[n], rest);
		sg->img.resize(rest);
	};
	int64_t l;
	for (;;) {
		if (pack && sg->img.size() >= ((size_t)1 << 20)) flush(false);
		if (!r.seek_marker()) { sg->stop = file_end; sg->hard_end = !src->partial; break; }
		const int64_t mp = r.marker_pos();
		if (mp >= sg->end) { sg->stop = mp; break; }
		const size_t img0 = sg->img.size();
		if ((l = r.fast(sg->img, min_len)) == FxReader::NOT_FAST) l = r.next_to(sg->img, min_len, bulk_threads);
		/* more of the stream follows these bytes and the reader has used them up: the record may go on there (`last` still holds the
		 * marker it started with, kseq.h:186-190, so it cannot tell) -- it is left, from its marker on, for the next batch */
		if (src->partial && !r.fill()) { sg->img.resize(img0); sg->stop = mp; break; }
		if (l < 0) { sg->stop = file_end; sg->hard_end = true; break; }   /* EOF inside a record, or a truncated FASTQ record: the stream ends (count.c:93) */
		if (l >= min_len) { ++sg->n_seq; sg->sum_len += l; }
	}
	r.close_at();
	if (pack) flush(true);
}

/* one window: cut [pos, wend) into segments, parse them on n_thr threads, accept the verified prefix.  Returns the
 * number of accepted segments; *next = where the following window starts; *done = the stream has ended */
static int parse_window(const ByteSource *fd, int64_t size, int64_t pos, int64_t WIN, int min_len, int n_thr, std::vector<ParSeg> &seg, int64_t *next, bool *done, WinPack *wp)
{
	const int64_t wend = std::min(size, pos + WIN), step = (wend - pos + n_thr - 1) / n_thr;
	int n_seg = 0;
	for (int i = 0; i < n_thr; ++i) {
		const int64_t cut = pos + i * step;
		if (cut >= wend) break;
		const int64_t st = i == 0 ? pos : guess_record_start(fd, cut, std::min(size, cut + ((int64_t)1 << 18)));
		if (i && (st < 0 || st >= wend)) continue;
		if (n_seg && st <= seg[n_seg - 1].start) continue;
		seg[n_seg].start = st; ++n_seg;
	}
	for (int i = 0; i < n_seg; ++i) seg[i].end = i + 1 < n_seg ? seg[i + 1].start : wend;
	std::vector<std::thread> th;
	/* few segments (long records: a cut finds no record start nearby): their threads' share of the parser threads strips the long bodies */
	const int bulk_threads = std::max(1, std::min(n_thr, (int)std::thread::hardware_concurrency()) / std::max(1, n_seg));
	for (int i = 1; i < n_seg; ++i) th.emplace_back(parse_segment, fd, size, &seg[i], min_len, bulk_threads);
	parse_segment(fd, size, &seg[0], min_len, bulk_threads);
	for (auto &t : th) t.join();
	int64_t at = pos;
	int n_ok = 0;
	for (int i = 0; i < n_seg; ++i) {
		if (seg[i].start != at) break;                           /* wrong guess: the rest of the window is parsed again */
		++n_ok;
		at = seg[i].stop;
		if (seg[i].hard_end) { *done = true; break; }
	}
	*next = at;
	if (fd->pack) {
		wp->codes.clear(); wp->valid.clear(); wp->n_words.clear(); wp->n_pos = wp->n_seq = 0;
		for (int i = 0; i < n_ok; ++i) {
			wp->n_seq += seg[i].n_seq;
			if (seg[i].valid.empty()) continue;
			wp->codes.push_back(seg[i].codes.data()); wp->valid.push_back(seg[i].valid.data()); wp->n_words.push_back((int64_t)seg[i].valid.size());
			wp->n_pos += 32 * (int64_t)seg[i].valid.size();
		}
	}
	return n_ok;
}

/* the source the parallel parser can take for `fn`, if any: a plain regular file (fx.fd) or a BGZF file, larger than min_size.
 * *own_fd (>= 0) is a descriptor the caller closes afterwards */
static bool parallel_source(const char *fn, const FxReader &fx, int n_thr, int64_t min_size, ByteSource *src, int *own_fd)
{
	*own_fd = -1;
	if (n_thr <= 1) return false;
	struct stat sb;
	if (fx.fd >= 0) {
		if (fstat(fx.fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= min_size) return false;
		src->fd = fx.fd; src->size = sb.st_size; src->bgzf = false;
		src->map_plain();
		return true;
	}
	if (fn == 0 || strcmp(fn, "-") == 0) return false;
	const int f = ::open(fn, O_RDONLY);
	if (f < 0) return false;
	if (!src->index_bgzf(f) || src->size <= min_size) { ::close(f); src->fd = -1; src->bgzf = false; return false; }
	*own_fd = f;
	return true;
}

/* calls sink(image bytes, n_bytes, n_seq) for consecutive pieces of the input, in order; false if sink failed.
 * Two sets of segment buffers: while the sink consumes one window (copy to the device + kernels), the parser
 * threads already work on the next one. */
static double g_t_parse_windows = 0, g_t_first_window = 0;    /* YAKAMD_VERBOSE: wall time of the window parses (they overlap the sink), of the first one */
/* A parser thread fills a ring of window sets while the caller's thread hands the finished windows to the sink, in order: two sets for
 * ASCII pieces (the sink copies a window to the device while the next one is parsed), four when the windows are packed -- a new table's
 * first feed waits ~0.25 s for the runtime to come up, time in which the parser gets through 2 GB of file instead of standing still */
static bool parse_parallel(const ByteSource *fd, int min_len, int n_thr, const ImgSink &sink, int64_t *stopped_at = 0, bool *stream_ended = 0)
{
	if (stopped_at) *stopped_at = 0;
	if (stream_ended) *stream_ended = false;
	const int64_t size = fd->size;
	if (size <= 0) return true;
	/* the windows grow from 128 MiB to 1 GiB: the device has its first piece after an eighth of the time a full window takes to parse */
	const int64_t win_set = env_threads_window();
	struct WinSet { std::vector<ParSeg> seg; WinPack wp; int n_ok; int64_t next; bool done; };
	const int NSET = fd->pack ? 4 : 2;
	std::vector<WinSet> *ring_p = new std::vector<WinSet>(NSET);
	std::vector<WinSet> &ring = *ring_p;
	for (auto &w : ring) { w.seg.resize(n_thr); w.n_ok = 0; w.next = 0; w.done = false; }
	std::mutex mu; std::condition_variable cv;
	int produced = 0, consumed = 0;
	bool prod_end = false, abort = false;
	std::thread producer([&]() {
		int64_t pos = 0;
		for (int k = 0; ; ++k) {
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return abort || k - consumed < NSET; }); if (abort) break; }
			WinSet &w = ring[k % NSET];
			const int64_t WIN = win_set ? win_set : std::min<int64_t>((int64_t)1 << 30, (int64_t)128 << 20 << std::min(k, 3));
			const double t = yk_realtime();
			w.done = false;
			w.n_ok = parse_window(fd, size, pos, WIN, min_len, n_thr, w.seg, &w.next, &w.done, &w.wp);
			const double dt = yk_realtime() - t;
			g_t_parse_windows += dt; if (k == 0) g_t_first_window = dt;
			/* (a partial source: a window that gets nowhere stands at a record that wants the bytes still to come) */
			const bool more = !w.done && w.next < size && !(fd->partial && w.next == pos);
			pos = w.next;
			{ std::lock_guard<std::mutex> lk(mu); ++produced; if (!more) prod_end = true; }
			cv.notify_all();
			if (!more) break;
		}
		{ std::lock_guard<std::mutex> lk(mu); prod_end = true; }
		cv.notify_all();
	});
	bool ok = true;
	for (int k = 0; ; ++k) {
		{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return produced > k || prod_end; }); if (produced <= k) break; }
		WinSet &w = ring[k % NSET];
		if (stopped_at) *stopped_at = w.next;
		if (stream_ended) *stream_ended = w.done;
		if (fd->pack) { if (w.n_ok > 0 && (w.wp.n_pos > 0 || w.wp.n_seq > 0)) ok = sink(0, (size_t)w.wp.n_pos, w.wp.n_seq, &w.wp); }
		else for (int i = 0; i < w.n_ok && ok; ++i) if (!w.seg[i].img.empty() || w.seg[i].n_seq > 0) ok = sink(w.seg[i].img.data(), w.seg[i].img.size(), w.seg[i].n_seq, 0);
		{ std::lock_guard<std::mutex> lk(mu); ++consumed; if (!ok) abort = true; }
		cv.notify_all();
		if (!ok) break;
	}
	producer.join();
	std::thread([ring_p]() { delete ring_p; }).detach();          /* (giving some GB of images back to the system takes ~0.1 s: not in the caller's way) */
	return ok;
}

/* an ordinary gzip file: batches of it are inflated by several threads (pgz.h) while the batch before is parsed, by the same window
 * parser, from memory; the record a batch ends in is carried to the front of the next one */
static bool gz_source(const char *fn, const FxReader &fx, int n_thr, pgz::Reader *z)
{
	if (n_thr <= 1 || fx.fd >= 0 || fn == 0 || strcmp(fn, "-") == 0 || yk_knob("YAKAMD_NO_PGZ", 0)) return false;
	pgz::tune().no_simd = yk_knob("YAKAMD_NO_AVX2", 0) != 0;
	return z->open(fn, n_thr);
}
static bool parse_gz(pgz::Reader *z, int min_len, int n_thr, const ImgSink &sink, bool pack = false)
{
	size_t keep = 0;
	for (bool last = false; !last; ) {
		uint8_t *p = 0; size_t n = 0;
		if (!z->next(keep, &p, &n, &last)) { yk_set_error("%s", z->why.c_str()); return false; }
		ByteSource src;
		src.set_memory(p, n, !last);
		src.pack = pack;
		int64_t stop = 0; bool ended = false;
		if (!parse_parallel(&src, min_len, n_thr, sink, &stop, &ended)) return false;
		if (ended) break;                                         /* a truncated record ended the stream (count.c:93) */
		keep = (size_t)stop;
	}
	if (getenv("YAKAMD_VERBOSE")) fprintf(stderr, "[yak_amd] gzip: %d threads inflated %lu chunks from a searched block start (%lu searched starts not used, %.1f MB decoded by the stitch)\n",
	                                      z->n_thr, (unsigned long)z->n_search_ok, (unsigned long)z->n_search_bad, z->n_gap_bits / 8e6);
	return true;
}

/* ------------------------------------------------------------------------------------------
 * Several GPUs behind yak_count() (replaces the kt_for over prefixes, count.c:129-143).
 * YAKAMD_GPUS = N (a divisor of 1 << pre): GPU r owns the contiguous prefixes [r P / N, (r + 1) P / N) --
 * table, filters and all.  The input is dealt to the GPUs in chunks of YAKAMD_MGPU_CHUNK bytes of sequence:
 * chunk j goes to GPU j % N, which extracts and groups its k-mers by prefix (yakamd_partition_dev); one
 * exchange per round of N chunks then moves every record to the owner of its prefix -- RCCL
 * (ncclGroupStart + ncclSend / ncclRecv pairs over xGMI, one communicator per GPU from ncclCommInitAll),
 * or plain device copies when two ranks share a GPU (YAKAMD_GPU_LIST=0,0: one-GPU test rigs); the owner
 * feeds the slices in chunk order, which is the stream order of the file, so the N-GPU bytes are the
 * 1-GPU bytes.  librccl is opened only when a job asks for several distinct GPUs.
 * ------------------------------------------------------------------------------------------ */
static bool env_fast_default() { return yk_knob("YAKAMD_FAST", 1) != 0; }   /* the exclusive-ownership path (the only one that takes tagged records) is on */

struct RcclApi {
	void *lib;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*GroupStart)(void);
	ncclResult_t (*GroupEnd)(void);
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	const char *(*GetErrorString)(ncclResult_t);
};
static bool rccl_open(RcclApi *R)
{
	memset(R, 0, sizeof(*R));
	R->lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!R->lib) R->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!R->lib) return false;
#define YK_SYM(f, n) *(void**)&R->f = dlsym(R->lib, n)
	YK_SYM(CommInitAll, "ncclCommInitAll"); YK_SYM(CommDestroy, "ncclCommDestroy"); YK_SYM(GroupStart, "ncclGroupStart"); YK_SYM(GroupEnd, "ncclGroupEnd");
	YK_SYM(Send, "ncclSend"); YK_SYM(Recv, "ncclRecv"); YK_SYM(GetErrorString, "ncclGetErrorString");
#undef YK_SYM
	return R->CommInitAll && R->CommDestroy && R->GroupStart && R->GroupEnd && R->Send && R->Recv;
}

/* Ranks own prefix ranges; chunks of the input live in SLOTS, one per distinct device (ranks that share a device -- the sweeps of one device
 * posing as several -- share its chunk, its partition and its buffers: the owner's slice of a chunk on its own device is fed where it lies).
 * Two sets of slot buffers: set x is being partitioned, exchanged and fed by a worker thread while the reader fills set 1 - x. */
struct MultiJob {
	int N, P, S;                                               /* ranks, sub-tables, slots */
	std::vector<int> dev, sdev, slot_of;                       /* device of rank r; device of slot s; slot of rank r */
	std::vector<hipStream_t> st, cp;                           /* per slot: exchange stream; copy stream of the reader (non-blocking: the fill of the next set must not wait for the kernels of this one) */
	bool use_rccl;
	RcclApi R;
	std::vector<ncclComm_t> comm;                              /* per slot */
	std::vector<uint8_t*> d_base[2];                           /* [set][slot]: chunk of sequence */
	std::vector<uint64_t*> d_send[2], d_recv[2];               /* [set][slot]: records grouped by prefix / slices received from the other slots */
	int64_t chunk, send_words, recv_words;
	bool ext_base;                                             /* d_base points at the caller's device buffers (yakamd_count_multi_dev) */
};

/* no filter + a plain file of more than YAKAMD_AUTO_SWEEP_GB (2.5) GB: nearly every k-mer instance may be a key of its own (an assembly),
 * and one pass holds ~70 bytes per selected key at its peak -- such inputs are counted as N ranks on one device, i.e. in N sweeps over
 * prefix ranges (N so that a sweep sees at most ~2.8 G positions of its own: 5 Gb in 2 sweeps, 2.4 s on a device whose memory has been in use
 * before, 2.4 s in 4; round 3 needed 4 -- a rank of 2 held a third copy of its table and 4 bytes of pending counts per slot while its layout
 * was replayed).  YAKAMD_GPUS set to anything switches the rule off */
static int auto_sweeps(const yak_copt_t *opt, const char *fn)
{
	if (fn == 0 || strcmp(fn, "-") == 0 || opt->bf_shift > opt->pre) return 1;
	const char *g = getenv("YAKAMD_AUTO_SWEEP_GB");
	const double lim = (g ? atof(g) : 2.5) * 1e9;
	if (lim <= 0) return 1;
	struct stat sb;
	if (stat(fn, &sb) != 0 || !S_ISREG(sb.st_mode) || (double)sb.st_size <= lim) return 1;
	unsigned char m[2] = { 0, 0 };
	const int f = ::open(fn, O_RDONLY);
	if (f < 0) return 1;
	const bool gz = ::read(f, m, 2) == 2 && m[0] == 0x1f && m[1] == 0x8b;
	::close(f);
	if (gz) return 1;                                           /* compressed: the size says little; the knob is there */
	int N = 2;
	while (N < 16 && (double)sb.st_size / N > 2.8e9) N <<= 1;
	return (1 << opt->pre) % N ? 1 : N;
}

static int multi_gpus(const yak_copt_t *opt, std::vector<int> *dev, const char *fn = 0)
{
	const char *e = getenv("YAKAMD_GPUS");
	if (!e) {
		const int S = auto_sweeps(opt, fn);
		if (S <= 1) return 1;
		int nd = 0;
		if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return 1;
		const char *dv = getenv("YAKAMD_DEVICE"), *lr = getenv("LOCAL_RANK");
		const int d = (dv ? atoi(dv) : lr ? atoi(lr) : 0) % nd;
		dev->assign(S, d);
		fprintf(stderr, "[M::yak_count] %s: no filter and a large plain file: counting in %d sweeps over prefix ranges on device %d (YAKAMD_GPUS / YAKAMD_AUTO_SWEEP_GB change that)\n", fn, S, d);
		return S;
	}
	const int N = atoi(e);
	if (N <= 1) return 1;
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return 1;
	if ((1 << opt->pre) % N) { fprintf(stderr, "[W::yak_count] YAKAMD_GPUS=%d does not divide the %d sub-tables: counting on one GPU\n", N, 1 << opt->pre); return 1; }
	dev->clear();
	if (const char *l = getenv("YAKAMD_GPU_LIST")) { for (const char *q = l; *q; ) { dev->push_back(atoi(q) % nd); while (*q && *q != ',') ++q; if (*q) ++q; } }
	for (int r = (int)dev->size(); r < N; ++r) dev->push_back(r % nd);
	dev->resize(N);
	return N;
}

static bool multi_open(MultiJob *J, int N, int P, const std::vector<int> &dev, int64_t chunk_dev = 0, bool tagged_only = false)   /* chunk_dev > 0: the chunks are the caller's device buffers of at most that many bytes */
{
	J->N = N; J->P = P; J->dev = dev;
	J->sdev.clear(); J->slot_of.assign(N, 0);
	for (int r = 0; r < N; ++r) {
		int s = -1;
		for (size_t q = 0; q < J->sdev.size(); ++q) if (J->sdev[q] == dev[r]) s = (int)q;
		if (s < 0) { s = (int)J->sdev.size(); J->sdev.push_back(dev[r]); }
		J->slot_of[r] = s;
	}
	const int S = J->S = (int)J->sdev.size();
	J->st.assign(S, 0); J->cp.assign(S, 0);
	for (int x = 0; x < 2; ++x) { J->d_base[x].assign(S, 0); J->d_send[x].assign(S, 0); J->d_recv[x].assign(S, 0); }
	const char *c = getenv("YAKAMD_MGPU_CHUNK");
	J->chunk = c && atoll(c) > 0 ? atoll(c) : (int64_t)1 << 28;
	if (chunk_dev > 0) J->chunk = chunk_dev;
	J->chunk = (J->chunk + 4095) & ~(int64_t)4095;
	J->ext_base = chunk_dev > 0;
	J->send_words = (tagged_only ? 1 : 2) * J->chunk;          /* one 16-byte record per position at most (8-byte tagged records use half of it) */
	/* a slot receives, for the ranks it hosts, their share of the S - 1 other chunks: (ranks here / N) each on average; refused beyond 1.5 x that */
	int most = 0;
	for (int s = 0; s < S; ++s) { int n_here = 0; for (int r = 0; r < N; ++r) n_here += J->slot_of[r] == s; most = std::max(most, n_here); }
	J->recv_words = S > 1 ? (int64_t)((double)J->send_words * (S - 1) * most / N * 1.5) + 4096 * N : 0;
	J->use_rccl = S > 1 && !yk_knob("YAKAMD_MGPU_NO_RCCL", 0);
	if (S < N) fprintf(stderr, "[M::yak_count] %d ranks on %d device%s: ranks that share a device share its chunks and take turns (their slices are fed where they lie)%s\n",
	                   N, S, S > 1 ? "s" : "", S == 1 ? "; nothing is exchanged" : "");
	if (J->use_rccl) {
		J->comm.assign(S, 0);
		if (!rccl_open(&J->R)) { fprintf(stderr, "[W::yak_count] librccl.so not found: exchanging with peer copies\n"); J->use_rccl = false; }
		else { const ncclResult_t r = J->R.CommInitAll(J->comm.data(), S, J->sdev.data()); if (r != ncclSuccess) { fprintf(stderr, "[W::yak_count] ncclCommInitAll: %s; exchanging with peer copies\n", J->R.GetErrorString ? J->R.GetErrorString(r) : "error"); J->use_rccl = false; } }
	}
	for (int s = 0; s < S; ++s) {
		if (hipSetDevice(J->sdev[s]) != hipSuccess || hipStreamCreate(&J->st[s]) != hipSuccess || hipStreamCreateWithFlags(&J->cp[s], hipStreamNonBlocking) != hipSuccess) return false;
		if (!J->use_rccl) for (int q = 0; q < S; ++q) if (q != s) (void)hipDeviceEnablePeerAccess(J->sdev[q], 0);
		for (int x = 0; x < 2; ++x) {
			/* (from the engine's pool: a process that counts again -- a benchmark's steps, the second pass of the filtered protocol -- finds these buffers there
			 * instead of asking the driver while the pool holds most of the device) */
			J->d_base[x][s] = J->ext_base ? 0 : (uint8_t*)yk_pool_get((size_t)J->chunk + 4096);
			J->d_send[x][s] = (uint64_t*)yk_pool_get((size_t)J->send_words * 8);
			J->d_recv[x][s] = J->recv_words ? (uint64_t*)yk_pool_get((size_t)J->recv_words * 8) : 0;
			if ((!J->ext_base && !J->d_base[x][s]) || !J->d_send[x][s] || (J->recv_words && !J->d_recv[x][s])) return false;
		}
	}
	(void)hipGetLastError();
	return true;
}

static void multi_close(MultiJob *J)
{
	for (int s = 0; s < J->S; ++s) {
		hipSetDevice(J->sdev[s]);
		for (int x = 0; x < 2; ++x) { if (!J->ext_base) yk_pool_release(J->d_base[x][s]); yk_pool_release(J->d_send[x][s]); yk_pool_release(J->d_recv[x][s]); J->d_base[x][s] = 0; J->d_send[x][s] = 0; J->d_recv[x][s] = 0; }
		if (J->st[s]) { hipStreamDestroy(J->st[s]); J->st[s] = 0; }
		if (J->cp[s]) { hipStreamDestroy(J->cp[s]); J->cp[s] = 0; }
		if (J->use_rccl && J->comm[s]) { J->R.CommDestroy(J->comm[s]); J->comm[s] = 0; }
	}
}

/* one round on buffer set x: chunk s (fill[s] bytes, stream offset t0[s]) sits in slot s.  Partition, exchange, feed. */
static bool multi_round(MultiJob *J, int x, yak_ch_ext *e, int k, int pre, int create_new, const std::vector<int64_t> &fill, const std::vector<uint64_t> &t0, std::string *why)
{
	std::mutex why_mu;
	auto note = [&]() { std::lock_guard<std::mutex> lk(why_mu); if (why && why->empty()) *why = yakamd_last_error(); };   /* called on the thread that failed */
	bool tagged = create_new && yakamd_tagged_ok(k, pre) && !yk_knob("YAKAMD_MGPU_REC16", 0);   /* 8-byte tagged records: half the exchange; every owner must still be on the exclusive-ownership path */
	for (int r = 0; r < J->N && tagged; ++r) tagged = e->sub[r] && yakamd_pass_fast(e->sub[r]);
	const int N = J->N, P = J->P, S = J->S, W = create_new && !tagged ? 2 : 1;      /* words per record: {hash, position}, or one (tagged record / bare hash) */
	for (int s = 0; s < S; ++s) if (fill[s] * W > J->send_words) { fprintf(stderr, "[E::yak_count] a chunk of %lld positions does not fit the send buffer (%lld words): 16-byte records were not planned for\n", (long long)fill[s], (long long)J->send_words); return false; }
	std::vector<std::vector<uint64_t> > bst(S, std::vector<uint64_t>(P + 1, 0));
	std::vector<int64_t> n_rec(S, 0);
	std::vector<char> ok(std::max(N, S), 1);
	{	/* every slot groups the k-mers of its chunk by prefix */
		std::vector<std::thread> th;
		for (int s = 0; s < S; ++s) th.emplace_back([&, s]() {
			if (fill[s] <= 0) return;
			hipSetDevice(J->sdev[s]);
			n_rec[s] = tagged ? yakamd_partition_tagged_dev(k, pre, J->d_base[x][s], fill[s], J->d_send[x][s], bst[s].data())
			         : create_new ? yakamd_partition_dev(k, pre, J->d_base[x][s], fill[s], J->d_send[x][s], bst[s].data())
			                      : yakamd_partition_hashes_dev(k, pre, J->d_base[x][s], fill[s], J->d_send[x][s], bst[s].data());
			if (n_rec[s] < 0) { ok[s] = 0; note(); }
		});
		for (auto &t : th) t.join();
	}
	for (int s = 0; s < S; ++s) if (!ok[s]) return false;
	/* receive layout of a slot: for each rank it hosts (rank order), the slices of the other slots' chunks (slot order) */
	std::vector<std::vector<uint64_t> > roff(N, std::vector<uint64_t>(S, 0));   /* roff[d][s]: where owner d's slice of chunk s lies in its slot's receive buffer (records) */
	std::vector<uint64_t> used(S, 0);
	for (int d = 0; d < N; ++d) {
		const int lo = d * (P / N), hi = (d + 1) * (P / N), sd = J->slot_of[d];
		for (int s = 0; s < S; ++s) { if (s == sd) continue; roff[d][s] = used[sd]; used[sd] += bst[s][hi] - bst[s][lo]; }
	}
	for (int s = 0; s < S; ++s) if ((int64_t)(used[s] * W) > J->recv_words) { fprintf(stderr, "[E::yak_count] device %d would receive %llu records in one round: prefixes too unevenly filled for YAKAMD_MGPU_CHUNK\n", J->sdev[s], (unsigned long long)used[s]); return false; }
	if (S > 1) {
		if (J->use_rccl) J->R.GroupStart();
		for (int s = 0; s < S; ++s)
			for (int d = 0; d < N; ++d) {
				const int lo = d * (P / N), hi = (d + 1) * (P / N), sd = J->slot_of[d];
				const uint64_t cnt = (bst[s][hi] - bst[s][lo]) * W;
				if (cnt == 0 || s == sd) continue;
				const uint64_t *src = J->d_send[x][s] + bst[s][lo] * W;
				uint64_t *dst = J->d_recv[x][sd] + roff[d][s] * W;
				if (J->use_rccl) {                                      /* (the current device matches the communicator of every call, as the library's own examples do it) */
					hipSetDevice(J->sdev[s]);
					if (J->R.Send(src, cnt, ncclUint64, sd, J->comm[s], J->st[s]) != ncclSuccess) ok[0] = 0;
					hipSetDevice(J->sdev[sd]);
					if (J->R.Recv(dst, cnt, ncclUint64, s, J->comm[sd], J->st[sd]) != ncclSuccess) ok[0] = 0;
				} else {
					hipSetDevice(J->sdev[sd]);
					if (hipMemcpyPeerAsync(dst, J->sdev[sd], src, J->sdev[s], cnt * 8, J->st[sd]) != hipSuccess) ok[0] = 0;
				}
			}
		if (J->use_rccl && J->R.GroupEnd() != ncclSuccess) ok[0] = 0;
		for (int s = 0; s < S; ++s) { hipSetDevice(J->sdev[s]); if (hipStreamSynchronize(J->st[s]) != hipSuccess) ok[0] = 0; }
		if (!ok[0] && J->use_rccl) {
			/* the collective library let the round down: the same slices as plain peer copies, from here on */
			fprintf(stderr, "[W::yak_count] RCCL exchange failed (%s): peer copies from now on\n", hipGetErrorString(hipGetLastError()));
			J->use_rccl = false; ok[0] = 1;
			for (int s = 0; s < S; ++s) for (int q = 0; q < S; ++q) if (q != s) { hipSetDevice(J->sdev[s]); (void)hipDeviceEnablePeerAccess(J->sdev[q], 0); }
			(void)hipGetLastError();
			for (int s = 0; s < S; ++s)
				for (int d = 0; d < N; ++d) {
					const int lo = d * (P / N), hi = (d + 1) * (P / N), sd = J->slot_of[d];
					const uint64_t cnt = (bst[s][hi] - bst[s][lo]) * W;
					if (cnt == 0 || s == sd) continue;
					hipSetDevice(J->sdev[sd]);
					if (hipMemcpyPeerAsync(J->d_recv[x][sd] + roff[d][s] * W, J->sdev[sd], J->d_send[x][s] + bst[s][lo] * W, J->sdev[s], cnt * 8, J->st[sd]) != hipSuccess) ok[0] = 0;
				}
			for (int s = 0; s < S; ++s) { hipSetDevice(J->sdev[s]); if (hipStreamSynchronize(J->st[s]) != hipSuccess) ok[0] = 0; }
		}
		if (!ok[0]) { fprintf(stderr, "[E::yak_count] exchange between the GPUs failed\n"); return false; }
	}
	{	/* every owner takes its slices, in chunk order = stream order; owners that share a device take turns (a feed may count a whole slice of the pass) */
		std::vector<std::thread> th;
		for (int sd = 0; sd < S; ++sd) th.emplace_back([&, sd]() { for (int d = 0; d < N; ++d) if (J->slot_of[d] == sd) {
			hipSetDevice(J->dev[d]);
			const int lo = d * (P / N), hi = (d + 1) * (P / N);
			std::vector<uint64_t> ob(P + 1);
			for (int s = 0; s < S; ++s) {
				const uint64_t cnt = bst[s][hi] - bst[s][lo];
				if (cnt == 0) continue;
				for (int p = 0; p <= P; ++p) { const int q = p < lo ? lo : p > hi ? hi : p; ob[p] = bst[s][q] - bst[s][lo]; }
				const uint64_t *rec = s == sd ? J->d_send[x][s] + bst[s][lo] * W : J->d_recv[x][sd] + roff[d][s] * W;   /* the slice of the device's own chunk is fed where the partition left it */
				const int rc = tagged ? yakamd_feed_partitioned_tagged_dev(e->sub[d], rec, (int64_t)cnt, ob.data(), t0[s], (uint64_t)fill[s], 0)
				             : create_new ? yakamd_feed_partitioned_dev(e->sub[d], rec, (int64_t)cnt, ob.data(), t0[s], (uint64_t)fill[s])
				                          : yakamd_count_partitioned_dev(e->sub[d], rec, (int64_t)cnt, ob.data());
				if (rc != 0) { ok[d] = 0; note(); }
			}
			if (hipStreamSynchronize(yk_ctx_stream(((yak_ch_ext*)e->sub[d])->ctx)) != hipSuccess) ok[d] = 0;   /* the copies out of this set's buffers are done before the set is filled again */
		} });
		for (auto &t : th) t.join();
	}
	for (int r = 0; r < N; ++r) if (!ok[r]) return false;
	return true;
}

static yak_ch_t *multi_table_new(const yak_copt_t *opt, int N, const std::vector<int> &dev);
static yak_ch_t *yak_count_multi(const char *fn, const yak_copt_t *opt, yak_ch_t *h0, int N, const std::vector<int> &dev)
{
	FxReader fx;
	if (!fx.open_file(fn)) return 0;
	const int P = 1 << opt->pre;
	yak_ch_t *h = h0;
	const int create_new = h0 ? 0 : 1;
	if (h0 == 0) {                                             /* N tables, one per rank, each owning its prefix range */
		h = multi_table_new(opt, N, dev);
		if (!h) { fx.close_file(); return 0; }
	}
	yak_ch_ext *e = (yak_ch_ext*)h;
	MultiJob J;
	bool ok = multi_open(&J, N, P, dev);
	const int S = J.S;
	const double t_job0 = yk_realtime();
	for (int r = 0; r < N && ok; ++r) ok = yakamd_pass_begin(e->sub[r], create_new) == 0;
	/* the reader fills the chunks of set `cur` while a worker thread partitions, exchanges and feeds the set before it */
	std::vector<int64_t> fill[2] = { std::vector<int64_t>(S, 0), std::vector<int64_t>(S, 0) };
	std::vector<uint64_t> t0[2] = { std::vector<uint64_t>(S, 0), std::vector<uint64_t>(S, 0) };
	std::thread worker;
	bool worker_ok = true;
	std::string worker_why;                                    /* yakamd_last_error() is per thread: the round's text comes back with it */
	int cur = 0;
	uint64_t t_stream = 0;
	int64_t n_seq_tot = 0;
	int g = 0;                                                 /* the slot whose chunk is being filled */
	auto wait_worker = [&]() { if (worker.joinable()) worker.join(); if (!worker_ok) ok = false; };
	double t_sink = 0, t_round_wait = 0;                        /* YAKAMD_VERBOSE: where the reader's time goes */
	/* host -> device through two pinned staging buffers: while one is on its way over the bus the reader copies the next piece into the other (a
	 * copy from pageable memory is staged by the runtime anyway, but behind a synchronise per piece) */
	const size_t STG = (size_t)32 << 20;
	uint8_t *stg[2] = { 0, 0 };
	/* an event belongs to the device that was current when it was made and can only be recorded on a stream of that device: one per staging
	 * buffer AND slot, made with the slot's device current; stg_on[i] = the slot whose copy stream holds buffer i's last copy (-1: idle) */
	std::vector<hipEvent_t> stg_ev[2];
	int stg_on[2] = { -1, -1 };
	int stg_i = 0;
	for (int i = 0; i < 2 && ok; ++i) {
		ok = hipHostMalloc((void**)&stg[i], STG) == hipSuccess;
		stg_ev[i].assign(S, (hipEvent_t)0);
		for (int s = 0; s < S && ok; ++s) { hipSetDevice(J.sdev[s]); ok = hipEventCreateWithFlags(&stg_ev[i][s], hipEventDisableTiming) == hipSuccess; }
	}
	auto to_device = [&](int gdev, uint8_t *dst, const char *src, size_t n) -> bool {
		hipSetDevice(J.sdev[gdev]);
		for (size_t o = 0; o < n; o += STG) {
			const size_t m = std::min(STG, n - o);
			if (stg_on[stg_i] >= 0 && hipEventSynchronize(stg_ev[stg_i][stg_on[stg_i]]) != hipSuccess) return false;
			memcpy(stg[stg_i], src + o, m);
			if (hipMemcpyAsync(dst + o, stg[stg_i], m, hipMemcpyHostToDevice, J.cp[gdev]) != hipSuccess || hipEventRecord(stg_ev[stg_i][gdev], J.cp[gdev]) != hipSuccess) return false;
			stg_on[stg_i] = gdev; stg_i ^= 1;
		}
		return true;
	};
	auto copies_done = [&]() { for (int s = 0; s < S && ok; ++s) { hipSetDevice(J.sdev[s]); ok = hipStreamSynchronize(J.cp[s]) == hipSuccess; } };
	auto round = [&]() {
		const double tw0 = yk_realtime();
		copies_done();                                          /* the chunks of this set are on their devices */
		wait_worker();                                          /* at most one round in flight: its set becomes the one to fill next */
		t_round_wait += yk_realtime() - tw0;
		if (ok) {
			const int x = cur;
			worker = std::thread([&, x]() { std::string why; worker_ok = multi_round(&J, x, e, opt->k, opt->pre, create_new, fill[x], t0[x], &why); if (!worker_ok) worker_why = why; });
		}
		cur ^= 1;
		std::fill(fill[cur].begin(), fill[cur].end(), 0); g = 0;
	};
	/* a piece (whole sequences, each followed by '\n') goes to the chunk being filled; a chunk is closed between two
	 * sequences, or inside one that is longer than a whole chunk */
	auto take_piece_body = [&](const char *img, size_t n, int64_t ns) -> bool {
		n_seq_tot += ns;
		while (n > 0 && ok) {
			const size_t room = (size_t)(J.chunk - fill[cur][g]);
			size_t m = n, back = 0;
			if (n > room) {
				const void *nl = room ? memrchr(img, '\n', room) : 0;
				if (nl) m = (size_t)((const char*)nl - img) + 1;
				else if (fill[cur][g] > 0) { if (++g == S) round(); continue; }
				else {
					/* one sequence longer than a whole chunk (a chromosome beyond YAKAMD_MGPU_CHUNK bases): the chunk ends inside it and the
					 * next one starts k - 1 bases earlier -- the k-mers that end in this chunk are counted here, those that end behind it
					 * there (a chunk's first k - 1 positions complete no k-mer), and stream positions simply continue */
					m = room; back = (size_t)opt->k - 1;
				}
			}
			if (fill[cur][g] == 0) t0[cur][g] = t_stream;
			ok = to_device(g, J.d_base[cur][g] + fill[cur][g], img, m);
			fill[cur][g] += (int64_t)m;
			t_stream += m - back; img += m - back; n -= m - back;
			if (fill[cur][g] == J.chunk || n > 0) { if (++g == S) round(); }
		}
		return ok;
	};
	auto take_piece = [&](const char *img, size_t n, int64_t ns, const WinPack*) -> bool { const double t0 = yk_realtime(); const bool r = take_piece_body(img, n, ns); t_sink += yk_realtime() - t0; return r; };
	const int n_thr = parse_threads(opt->n_thread);
	ByteSource psrc; int psrc_fd = -1;
	bool par = parallel_source(fn, fx, n_thr, 1 << 20, &psrc, &psrc_fd);
	pgz::Reader gz;
	if (ok && par) ok = parse_parallel(&psrc, opt->k, n_thr, take_piece) && ok;
	else if (ok && gz_source(fn, fx, n_thr, &gz)) { par = true; ok = parse_gz(&gz, opt->k, n_thr, take_piece) && ok; }
	else if (ok) {
		std::vector<char> piece;
		int64_t l, ns = 0;
		for (;;) {
			if ((l = fx.fast(piece, opt->k)) == FxReader::NOT_FAST) {
				if ((l = fx.next()) < 0) break;
				if (l >= opt->k) { piece.insert(piece.end(), fx.seq.begin(), fx.seq.end()); piece.push_back('\n'); }
			}
			if (l >= opt->k) ++ns;
			if (piece.size() >= ((size_t)1 << 24)) { if (!take_piece(piece.data(), piece.size(), ns, 0)) break; piece.clear(); ns = 0; }
		}
		if (ok && !piece.empty()) take_piece(piece.data(), piece.size(), ns, 0);
	}
	if (ok) { bool any = false; for (int s = 0; s < S; ++s) any = any || fill[cur][s] > 0; if (any) round(); }
	wait_worker();
	const double t_fed = yk_realtime() - t_job0;
	for (int i = 0; i < 2; ++i) {
		if (stg_on[i] >= 0) (void)hipEventSynchronize(stg_ev[i][stg_on[i]]);
		for (hipEvent_t e_ : stg_ev[i]) if (e_) (void)hipEventDestroy(e_);
		if (stg[i]) (void)hipHostFree(stg[i]);
	}
	multi_close(&J);                                           /* the chunk and exchange buffers go before the passes finish: memory is tightest there */
	{	/* every rank finishes its pass: partitions, counting, layout -- side by side; ranks that share a device take turns, so that the
		 * scratch of only one of them is alive at a time (one device posing as N = the pass in N sweeps over prefix ranges: what lets a
		 * 5 Gb assembly through 288 GB) */
		std::vector<int64_t> n_ins(N, 0);
		std::vector<std::thread> th;
		std::vector<std::string> why(N);                       /* the error text is per thread: bring it back */
		for (int sd = 0; sd < S; ++sd) th.emplace_back([&, sd]() { for (int r = 0; r < N; ++r) if (J.slot_of[r] == sd) { n_ins[r] = yakamd_pass_end(e->sub[r]); if (n_ins[r] < 0) why[r] = yakamd_last_error(); } });
		for (auto &t : th) t.join();
		for (int r = 0; r < N; ++r) if (n_ins[r] < 0) fprintf(stderr, "[E::yak_count] rank %d of %d (device %d): %s\n", r, N, dev[r], why[r].c_str());
		for (int r = 0; r < N; ++r) { if (n_ins[r] < 0) ok = false; else e->sub[r]->tot += (uint64_t)n_ins[r]; }
	}
	multi_tot(h);
	if (getenv("YAKAMD_VERBOSE") && atoi(getenv("YAKAMD_VERBOSE")) > 0) {
		fprintf(stderr, "[yak_amd] %d ranks: input read, dealt and fed by %.3f s (%d parser threads; %.3f s inside the sink that copies the pieces to the devices, %.3f s of it waiting for copies and the round before), the ranks' passes finished by %.3f s\n",
		        N, t_fed, n_thr, t_sink, t_round_wait, yk_realtime() - t_job0);
		for (int s = 0; s < S; ++s) { hipSetDevice(J.sdev[s]); yk_pool_report("the job"); }
	}
	fprintf(stderr, "[M::%s::%.3f*%.2f] %ld sequences in total; %ld distinct k-mers in the hash table (%d GPUs, %s)\n", "yak_count",
	        yk_realtime(), yk_cputime() / (yk_realtime() + 1e-9), (long)n_seq_tot, (long)h->tot, N, S == 1 ? "one device: nothing exchanged" : J.use_rccl ? "RCCL exchange" : "peer copies");
	if (psrc_fd >= 0) ::close(psrc_fd);
	fx.close_file();
	if (!ok) { fprintf(stderr, "[E::yak_count] %s\n", !worker_why.empty() ? worker_why.c_str() : yakamd_last_error()); if (!h0) yak_ch_destroy(h); return 0; }
	return h;
}

/* a table sharded over N ranks (dev[r] = device of rank r), every rank owning its prefix range */
static yak_ch_t *multi_table_new(const yak_copt_t *opt, int N, const std::vector<int> &dev)
{
	const int P = 1 << opt->pre;
	yak_ch_ext *e = (yak_ch_ext*)calloc(1, sizeof(*e));
	e->magic = EXT_MAGIC; e->n_sub = N; e->sub = (yak_ch_t**)calloc(N, sizeof(yak_ch_t*));
	yak_ch_t *h = &e->pub;
	h->k = opt->k; h->pre = opt->pre;
	h->h = (yak_ch1_t*)calloc((size_t)P, sizeof(yak_ch1_t));
	bool ok = true;
	for (int r = 0; r < N && ok; ++r) {
		yk_ctx_next_device(dev[r]);
		e->sub[r] = yak_ch_init(opt->k, opt->pre, opt->bf_n_hash, opt->bf_shift);
		ok = e->sub[r] && yakamd_set_shard(e->sub[r], r * (P / N), (r + 1) * (P / N)) == 0;
	}
	if (!ok) { for (int r = 0; r < N; ++r) if (e->sub[r]) yak_ch_destroy(e->sub[r]); free(e->sub); free(h->h); free(e); return 0; }
	e->ctx = 0;                                               /* no context of its own: every yakamd_* entry point refuses a sharded table instead of working on one shard */
	h->n_hash = e->sub[0]->n_hash; h->n_shift = e->sub[0]->n_shift;
	for (int p = 0; p < P; ++p) h->h[p].b = e->sub[0]->h[p].b;   /* descriptors only: "has a filter" for callers that look */
	return h;
}

/* The same job with its input already on the devices (the benchmark's N-GPU mode; a caller with its own reader): the stream is cut into rounds of
 * one chunk per DEVICE -- chunk s of round b lies at d_chunk[b * S + s] on the s-th distinct device of `dev` (n_bytes[b * S + s] bytes of the base
 * image, at most 2^31 - 4096; 0 = none), and the stream order is round by round, device by device, exactly as yak_count() deals a file.  h0 == 0:
 * a new table sharded over the n_rank ranks (dev[r] = device of rank r; several ranks may share a device) comes back; h0 != 0: its k-mers are counted
 * (count.c:155-157).  exchange_out (may be 0): 1 = RCCL grouped send / recv, 2 = peer copies, 0 = one device, nothing exchanged.  The caller keeps
 * the chunks alive until the call returns */
extern "C" yak_ch_t *yakamd_count_multi_dev(const yak_copt_t *opt, yak_ch_t *h0, int n_rank, const int *dev_of_rank, int n_rounds,
                                            const void *const *d_chunk, const int64_t *n_bytes, int *exchange_out)
{
	const int P = 1 << opt->pre, N = n_rank;
	if (N < 1 || P % N) { fprintf(stderr, "[E::yakamd_count_multi_dev] %d ranks do not divide the %d sub-tables\n", N, P); return 0; }
	std::vector<int> dev(dev_of_rank, dev_of_rank + N);
	if (h0) {
		yak_ch_ext *e0 = (yak_ch_ext*)h0;
		if (e0->n_sub != N) { fprintf(stderr, "[E::yakamd_count_multi_dev] the table is sharded over %d ranks, not %d\n", e0->n_sub > 0 ? e0->n_sub : 1, N); return 0; }
		assert(h0->k == opt->k && h0->pre == opt->pre);
	}
	const int create_new = h0 ? 0 : 1;
	yak_ch_t *h = h0 ? h0 : multi_table_new(opt, N, dev);
	if (!h) return 0;
	yak_ch_ext *e = (yak_ch_ext*)h;
	MultiJob J;
	J.S = 0;
	int64_t cmax = 4096;
	{	/* the distinct devices, in rank order: that is the order of the chunks inside a round */
		std::vector<int> sd;
		for (int r = 0; r < N; ++r) if (std::find(sd.begin(), sd.end(), dev[r]) == sd.end()) sd.push_back(dev[r]);
		for (int i = 0; i < n_rounds * (int)sd.size(); ++i) cmax = std::max<int64_t>(cmax, n_bytes[i]);
	}
	if (cmax > ((int64_t)1 << 31) - 4096) { fprintf(stderr, "[E::yakamd_count_multi_dev] a chunk holds at most 2^31 - 4096 stream positions\n"); if (!h0) yak_ch_destroy(h); return 0; }
	const bool tagged_only = create_new && yakamd_tagged_ok(opt->k, opt->pre) && !yk_knob("YAKAMD_MGPU_REC16", 0) && env_fast_default();
	bool ok = multi_open(&J, N, P, dev, cmax, tagged_only || !create_new);
	const int S = J.S;
	yk_realtime();
	for (int r = 0; r < N && ok; ++r) ok = yakamd_pass_begin(e->sub[r], create_new) == 0;
	std::string why;
	uint64_t t_stream = 0;
	for (int b = 0; b < n_rounds && ok; ++b) {
		std::vector<int64_t> fill(S, 0);
		std::vector<uint64_t> t0(S, 0);
		for (int s = 0; s < S; ++s) {
			fill[s] = n_bytes[(size_t)b * S + s];
			t0[s] = t_stream; t_stream += (uint64_t)fill[s];
			J.d_base[b & 1][s] = (uint8_t*)d_chunk[(size_t)b * S + s];
		}
		ok = multi_round(&J, b & 1, e, opt->k, opt->pre, create_new, fill, t0, &why);
	}
	const int exch = S == 1 ? 0 : J.use_rccl ? 1 : 2;
	for (int x = 0; x < 2; ++x) for (int s = 0; s < S; ++s) J.d_base[x][s] = 0;
	multi_close(&J);
	{
		std::vector<int64_t> n_ins(N, 0);
		std::vector<std::thread> th;
		std::vector<std::string> whyr(N);
		for (int sd = 0; sd < S; ++sd) th.emplace_back([&, sd]() { for (int r = 0; r < N; ++r) if (J.slot_of[r] == sd) { n_ins[r] = yakamd_pass_end(e->sub[r]); if (n_ins[r] < 0) whyr[r] = yakamd_last_error(); } });
		for (auto &t : th) t.join();
		for (int r = 0; r < N; ++r) if (n_ins[r] < 0) { fprintf(stderr, "[E::yakamd_count_multi_dev] rank %d of %d (device %d): %s\n", r, N, dev[r], whyr[r].c_str()); ok = false; }
		for (int r = 0; r < N; ++r) if (n_ins[r] >= 0) e->sub[r]->tot += (uint64_t)n_ins[r];
	}
	multi_tot(h);
	if (exchange_out) *exchange_out = exch;
	fprintf(stderr, "[M::%s::%.3f*%.2f] %d rounds of device-resident chunks; %ld distinct k-mers in the hash table (%d ranks, %s)\n", "yakamd_count_multi_dev",
	        yk_realtime(), yk_cputime() / (yk_realtime() + 1e-9), n_rounds, (long)h->tot, N, exch == 0 ? "one device: nothing exchanged" : exch == 1 ? "RCCL exchange" : "peer copies");
	if (!ok) { fprintf(stderr, "[E::yakamd_count_multi_dev] %s\n", !why.empty() ? why.c_str() : yakamd_last_error()); if (!h0) yak_ch_destroy(h); return 0; }
	return h;
}

/* reference count.c:147-166 */
yak_ch_t *yak_count(const char *fn, const yak_copt_t *opt, yak_ch_t *h0)
{
	{
		std::vector<int> dev;
		const int N = h0 ? ((yak_ch_ext*)h0)->n_sub : multi_gpus(opt, &dev, fn);
		if (N > 1) {
			if (h0) { dev.clear(); for (int r = 0; r < N; ++r) dev.push_back(yk_ctx_device(((yak_ch_ext*)((yak_ch_ext*)h0)->sub[r])->ctx)); }
			return yak_count_multi(fn, opt, h0, N, dev);
		}
	}
	/* the file's identity: the filtered protocol counts the same file twice (main.c:53-57); the first call then keeps its hashed k-mers on
	 * the device (yakamd_retain_input) and the second counts those instead of parsing, copying and hashing the file again */
	uint64_t sid[4] = { 0, 0, 0, 0 };
	bool have_sid = false;
	{
		struct stat sb;
		if (fn && strcmp(fn, "-") != 0 && stat(fn, &sb) == 0 && S_ISREG(sb.st_mode) && !yk_knob("YAKAMD_NO_RETAIN", 0)) {
			sid[0] = (uint64_t)sb.st_dev; sid[1] = (uint64_t)sb.st_ino; sid[2] = (uint64_t)sb.st_size;
			sid[3] = (uint64_t)sb.st_mtim.tv_sec * 1000000000ull + (uint64_t)sb.st_mtim.tv_nsec;
			have_sid = true;
		}
	}
	bool pass_open = false;
	if (h0) {
		assert(h0->k == opt->k && h0->pre == opt->pre);         /* count.c:157 */
		int64_t n_seq_kept = 0;
		if (have_sid && yk_ctx_same_source(((yak_ch_ext*)h0)->ctx, sid, &n_seq_kept)) {
			yk_realtime();
			if (yakamd_pass_begin(h0, 0) != 0) return 0;
			const int r = yakamd_count_retained(h0);
			if (r < 0) { yakamd_pass_end(h0); return 0; }
			if (r == 0) {
				const int64_t n_ins = yakamd_pass_end(h0);
				if (n_ins < 0) return 0;
				h0->tot += (uint64_t)n_ins;
				fprintf(stderr, "[M::%s::%.3f*%.2f] %ld sequences in total (the k-mers of the first pass over this file, kept on the device); %ld distinct k-mers in the hash table\n", "yak_count",
				        yk_realtime(), yk_cputime() / (yk_realtime() + 1e-9), (long)n_seq_kept, (long)h0->tot);
				return h0;
			}
			pass_open = true;                                     /* nothing usable was kept: the pass goes on with the file */
		}
	}
	FxReader fx;
	if (!fx.open_file(fn)) { if (pass_open) yakamd_pass_end(h0); return 0; }   /* count.c:152 */
	yak_ch_t *h = h0;
	const int create_new = h0 ? 0 : 1;
	yk_realtime();
	/* a plain regular file is mapped and parsed by several threads; anything else (gzip, a pipe) streams through the reader */
	const int n_thr = parse_threads(opt->n_thread);
	ByteSource psrc; int psrc_fd = -1;
	int64_t par_size = parallel_source(fn, fx, n_thr, 1 << 20, &psrc, &psrc_fd) ? psrc.size : -1;   /* plain or block-gzipped regular file */
	pgz::Reader *gz_p = new pgz::Reader;
	pgz::Reader &gz = *gz_p;
	struct GzDrop { pgz::Reader *p; ~GzDrop() { pgz::Reader *q = p; std::thread([q]() { delete q; }).detach(); } } gz_drop{ gz_p };   /* (its buffers go back to the system behind the caller's back) */
	const bool use_gz = par_size < 0 && gz_source(fn, fx, n_thr, &gz);   /* an ordinary gzip file */
	if (use_gz) par_size = 0;
	int ok = 0;
	auto open_table = [&]() {                                /* a new table: runtime start-up, the filter's 2^bf_shift bits, the pass */
		if (!h0) h = yak_ch_init(opt->k, opt->pre, opt->bf_n_hash, opt->bf_shift);
		if (!h0 && h && have_sid && opt->bf_shift > opt->pre) yakamd_retain_input(h, 1);   /* a filtered count: a second pass over this file is to be expected */
		ok = h != 0 && (pass_open || yakamd_pass_begin(h, create_new) == 0);
	};
	std::thread opener;                                      /* ... happen while the first window of the file is being parsed */
	if (par_size >= 0 && !h0) opener = std::thread(open_table); else open_table();
	if (!opener.joinable() && h == 0) { if (psrc_fd >= 0) ::close(psrc_fd); fx.close_file(); return 0; }
	std::vector<char> chunk;
	if (par_size < 0) chunk.reserve((size_t)std::min<int64_t>(opt->chunk_size + (opt->chunk_size >> 3) + 65536, (int64_t)1 << 31));
	uint64_t t0 = 0;
	int64_t l, sum_len = 0, n_seq = 0, n_seq_tot = 0;
	if (par_size >= 0) {
		const bool pack = !yk_knob("YAKAMD_NO_HOST_PACK", 0);          /* the stream crosses the bus at 0.375 B per base, packed by the threads that parsed it */
		psrc.pack = pack;
		double t_sink = 0, t_open_wait = 0;
		g_t_parse_windows = 0;
		const ImgSink sink = [&](const char *img, size_t img_n, int64_t ns, const WinPack *packed) {
			const double ts0 = yk_realtime();
			if (opener.joinable()) { opener.join(); t_open_wait = yk_realtime() - ts0; }
			if (!ok) return false;
			bool good = img_n == 0 || (pack ? yakamd_feed_packed_pieces_host(h, (int)packed->n_words.size(), packed->codes.data(), packed->valid.data(), packed->n_words.data(), t0)
			                                : yakamd_feed_bases_host(h, img, (int64_t)img_n, t0)) == 0;
			t_sink += yk_realtime() - ts0;
			t0 += img_n; n_seq_tot += ns;
			fprintf(stderr, "[M::%s::%.3f*%.2f] processed %ld sequences\n", "yak_count", yk_realtime(), yk_cputime() / (yk_realtime() + 1e-9), (long)ns);
			return good;
		};
		const double tp0 = yk_realtime();
		const bool parsed = use_gz ? parse_gz(&gz, opt->k, n_thr, sink, pack) : parse_parallel(&psrc, opt->k, n_thr, sink);
		if (getenv("YAKAMD_VERBOSE")) fprintf(stderr, "[yak_amd] reader: %.3f s from the first window to the last piece fed; the windows took %.3f s to parse (the first %.3f s), the feeds %.3f s (%.3f s of it waiting for the new table)\n",
		                                      yk_realtime() - tp0, g_t_parse_windows, g_t_first_window, t_sink, t_open_wait);
		if (opener.joinable()) opener.join();
		if (h == 0) { if (psrc_fd >= 0) ::close(psrc_fd); fx.close_file(); return 0; }
		ok = ok && parsed;
	}
	auto flush = [&]() {
		if (!chunk.empty() && ok) ok = yakamd_feed_bases_host(h, chunk.data(), (int64_t)chunk.size(), t0) == 0;
		t0 += chunk.size();
		n_seq_tot += n_seq;
		fprintf(stderr, "[M::%s::%.3f*%.2f] processed %ld sequences\n", "yak_count", yk_realtime(), yk_cputime() / (yk_realtime() + 1e-9), (long)n_seq);
		chunk.clear(); sum_len = 0; n_seq = 0;
	};
	while (ok && par_size < 0) {                             /* count.c:93 */
		if ((l = fx.fast(chunk, opt->k)) == FxReader::NOT_FAST) {
			if ((l = fx.next()) < 0) break;
			if (l >= opt->k) { chunk.insert(chunk.end(), fx.seq.begin(), fx.seq.end()); chunk.push_back('\n'); }   /* a non-ACGT byte ends the read (count.c:41) */
		}
		if (l < opt->k) continue;                            /* count.c:95 */
		sum_len += l; ++n_seq;
		if (sum_len >= opt->chunk_size || chunk.size() > ((size_t)1 << 31)) flush();   /* count.c:106 */
	}
	if (n_seq) flush();
	if (ok) {
		const int64_t n_ins = yakamd_pass_end(h);
		if (n_ins < 0) ok = 0; else h->tot += (uint64_t)n_ins;   /* count.c:138 */
		if (ok && create_new && have_sid && yakamd_retained_instances(h) > 0) yk_ctx_set_source(((yak_ch_ext*)h)->ctx, sid, n_seq_tot);
	}
	fprintf(stderr, "[M::%s::%.3f*%.2f] %ld sequences in total; %ld distinct k-mers in the hash table\n", "yak_count",
	        yk_realtime(), yk_cputime() / (yk_realtime() + 1e-9), (long)n_seq_tot, (long)h->tot);
	if (getenv("YAKAMD_VERBOSE") && h) { (void)hipSetDevice(yk_ctx_device(((yak_ch_ext*)h)->ctx)); yk_pool_report("the pass"); }
	if (psrc_fd >= 0) ::close(psrc_fd);
	fx.close_file();
	if (!ok) { if (!h0) yak_ch_destroy(h); return 0; }
	return h;
}

/* reference count.c:168-193: clear the counts, then count the k-mers of fn that are in the table --
 * one count-existing pass on the device (sequences shorter than k have no k-mer either way) */
void yak_recount(const char *fn, yak_ch_t *h)
{
	yak_copt_t o;
	yak_copt_init(&o);
	o.k = h->k; o.pre = h->pre;
	{                                                         /* count.c:172-173: an unreadable file leaves the table untouched */
		gzFile fp = (fn == 0 || strcmp(fn, "-") == 0) ? 0 : gzopen(fn, "r");
		if (fn != 0 && strcmp(fn, "-") != 0) { if (fp == 0) return; gzclose(fp); }
	}
	yak_ch_clear(h, 1);
	if (yak_count(fn, &o, h) == 0) fprintf(stderr, "[E::yak_recount] %s\n", yakamd_last_error());
}

/* host-only hook for tests: the base image yak_count() would hand to the device for `fn` (sequences
 * of at least min_len bases, each followed by '\n'); caller frees *out with free().  -1 if unreadable. */
int64_t yakamd_host_image(const char *fn, int min_len, int use_fast_path, char **out)
{
	FxReader fx;
	if (!fx.open_file(fn)) return -1;
	std::vector<char> img;
	int64_t l;
	const double t_ = yk_realtime();
	const int n_thr = use_fast_path ? parse_threads(1) : 1;    /* tests set YAKAMD_PARSE_THREADS */
	{
		ByteSource psrc; int psrc_fd = -1;
		pgz::Reader gz;
		const bool plain = parallel_source(fn, fx, n_thr, 0, &psrc, &psrc_fd);
		if (plain || gz_source(fn, fx, n_thr, &gz)) {
			size_t total = 0;
			const bool keep = !(use_fast_path & 2);                   /* (2: the image is only measured, for timing the reader) */
			const ImgSink sink = [&](const char *part, size_t part_n, int64_t, const WinPack*) { total += part_n; if (keep) img.insert(img.end(), part, part + part_n); return true; };
			if (plain) parse_parallel(&psrc, min_len, n_thr, sink);
			else if (!parse_gz(&gz, min_len, n_thr, sink)) { fx.close_file(); return -1; }
			if (getenv("YAKAMD_VERBOSE")) fprintf(stderr, "[yak_amd] host_image: %.3f s, %d threads, %zu bytes%s\n", yk_realtime() - t_, n_thr, total, psrc.bgzf ? " (BGZF blocks inflated by the parser threads)" : "");
			if (psrc_fd >= 0) ::close(psrc_fd);
			fx.close_file();
			*out = (char*)malloc(img.size() + 1);
			memcpy(*out, img.data(), img.size());
			return (int64_t)img.size();
		}
	}
	for (;;) {
		if (!use_fast_path || (l = fx.fast(img, min_len)) == FxReader::NOT_FAST) {
			if ((l = fx.next()) < 0) break;
			if (l >= min_len) { img.insert(img.end(), fx.seq.begin(), fx.seq.end()); img.push_back('\n'); }
		}
	}
	if (getenv("YAKAMD_VERBOSE")) fprintf(stderr, "[yak_amd] host_image: %.3f s in the reader loop\n", yk_realtime() - t_);
	fx.close_file();
	*out = (char*)malloc(img.size() + 1);
	memcpy(*out, img.data(), img.size());
	return (int64_t)img.size();
}

/* host-only hook for tests: what yak_count() hands to the device when its parser threads pack (one packed image per window), unpacked again --
 * 'A' 'C' 'G' 'T' for a base, '\n' for a position that is none (N, a record's end, the positions that pad a segment to a multiple of 32) */
int64_t yakamd_host_image_packed(const char *fn, int min_len, char **out)
{
	FxReader fx;
	*out = 0;
	if (!fx.open_file(fn)) return -1;
	const int n_thr = parse_threads(1);
	std::vector<char> img;
	ByteSource psrc; int psrc_fd = -1;
	pgz::Reader gz;
	const ImgSink sink = [&](const char*, size_t, int64_t, const WinPack *wp) {
		for (size_t p = 0; p < wp->n_words.size(); ++p) {
			const uint32_t *codes = (const uint32_t*)wp->codes[p], *valid = (const uint32_t*)wp->valid[p];
			for (size_t j = 0; j < (size_t)wp->n_words[p] * 32; ++j) img.push_back((valid[j >> 5] >> (j & 31) & 1) ? "ACGT"[codes[j >> 4] >> (2 * (j & 15)) & 3] : '\n');
		}
		return true;
	};
	bool ok = true;
	if (parallel_source(fn, fx, n_thr, 0, &psrc, &psrc_fd)) { psrc.pack = true; ok = parse_parallel(&psrc, min_len, n_thr, sink); }
	else if (gz_source(fn, fx, n_thr, &gz)) ok = parse_gz(&gz, min_len, n_thr, sink, true);
	else ok = false;
	if (psrc_fd >= 0) ::close(psrc_fd);
	fx.close_file();
	if (!ok) return -1;
	*out = (char*)malloc(img.size() + 1);
	memcpy(*out, img.data(), img.size());
	return (int64_t)img.size();
}

/* host-only hooks for tests of the gzip reader: its chunk size, the smallest file it takes and the room in front of a batch (0 / < 0: as is); the whole inflated stream of
 * `fn` through the batch interface, every batch handing a tail back as the parser does (-1: not a file the reader takes; -2: it failed) */
void yakamd_gz_tune(int64_t chunk_bytes, int64_t min_file_bytes, int64_t front_bytes)
{
	if (chunk_bytes > 0) pgz::tune().chunk = (size_t)chunk_bytes;
	if (min_file_bytes >= 0) pgz::tune().min_size = (size_t)min_file_bytes;
	if (front_bytes >= 0) pgz::tune().front = (size_t)front_bytes;
}
int64_t yakamd_gz_inflate(const char *fn, int n_threads, char **out)
{
	pgz::Reader z;
	*out = 0;
	pgz::tune().no_simd = yk_knob("YAKAMD_NO_AVX2", 0) != 0;
	if (!z.open(fn, n_threads, true)) return -1;
	std::vector<char> all;
	size_t keep = 0;
	for (bool last = false; !last; ) {
		uint8_t *p = 0; size_t n = 0;
		if (!z.next(keep, &p, &n, &last)) { yk_set_error("%s", z.why.c_str()); return -2; }
		keep = last ? n : n - std::min<size_t>(n, (all.size() * 7 + 13) % 5000);
		all.insert(all.end(), (const char*)p, (const char*)p + keep);
	}
	*out = (char*)malloc(all.size() + 1);
	memcpy(*out, all.data(), all.size());
	return (int64_t)all.size();
}

/* reference qv.c:137-144 */
void yak_qopt_init(yak_qopt_t *opt)
{
	memset(opt, 0, sizeof(yak_qopt_t));
	opt->chunk_size = 1000000000;
	opt->n_threads = 4;
	opt->min_frac = 0.5;
	opt->fpr = 0.00004;
}

/* reference qv.c:34-135.  The table is already resident on the device; every chunk of sequences is
 * looked up there (k_lookup), reduced per sequence and binned (k_qv_reduce).  The EK / SQ lines of
 * -E / -p are printed from the values copied back, in input order (the reference prints them in a
 * thread-dependent order).  On a device error the function prints a message and leaves cnt zeroed. */
void yak_qv(const yak_qopt_t *opt, const char *fn, const yak_ch_t *ch, int64_t *cnt)
{
	const int n_cnt = 1 << YAK_COUNTER_BITS;
	memset(cnt, 0, n_cnt * sizeof(int64_t));
	yak_ch_t *h = (yak_ch_t*)ch;
	if (ch->k >= 32) { fprintf(stderr, "[E::yak_qv] k must be below 32\n"); return; }   /* qv.c:44 asserts */
	if (multi_refuse(ch, __func__)) return;                      /* the lookup kernel reads one table image: restore the .yak file for qv */
	FxReader fx;
	if (!fx.open_file(fn)) return;
	uint64_t *d_hist = (uint64_t*)yakamd_dev_alloc(n_cnt * 8);
	std::vector<uint64_t> zero(n_cnt, 0), h_off;
	std::vector<uint32_t> h_len, h_tot, h_non0;
	std::vector<std::string> names;
	std::vector<char> chunk;
	std::vector<unsigned short> h_t;
	bool ok = d_hist && yakamd_memcpy_h2d(d_hist, zero.data(), n_cnt * 8) == 0;
	int64_t l, sum_len = 0;
	auto flush = [&]() {
		const size_t nb = chunk.size(), ns = h_len.size();
		if (ns == 0) return;
		chunk.resize((nb + 15) & ~(size_t)15, '\n');
		void *d_b = yakamd_dev_alloc(chunk.size()), *d_t = yakamd_dev_alloc(chunk.size() * 2);
		uint64_t *d_off = (uint64_t*)yakamd_dev_alloc(ns * 8);
		uint32_t *d_len = (uint32_t*)yakamd_dev_alloc(ns * 4), *d_tot = (uint32_t*)yakamd_dev_alloc(ns * 4), *d_non0 = (uint32_t*)yakamd_dev_alloc(ns * 4);
		ok = ok && d_b && d_t && d_off && d_len && d_tot && d_non0
		     && yakamd_memcpy_h2d(d_b, chunk.data(), chunk.size()) == 0 && yakamd_memcpy_h2d(d_off, h_off.data(), ns * 8) == 0
		     && yakamd_memcpy_h2d(d_len, h_len.data(), ns * 4) == 0
		     && yakamd_lookup_dev(h, d_b, (int64_t)nb, d_t) == 0
		     && yakamd_qv_reduce_dev(h, d_t, d_off, d_len, (int64_t)ns, opt->min_len, opt->min_frac, d_tot, d_non0, d_hist) == 0;
		if (ok && (opt->print_each || opt->print_err_kmer)) {
			h_tot.resize(ns); h_non0.resize(ns);
			ok = yakamd_memcpy_d2h(h_tot.data(), d_tot, ns * 4) == 0 && yakamd_memcpy_d2h(h_non0.data(), d_non0, ns * 4) == 0;
			if (ok && opt->print_err_kmer) { h_t.resize(chunk.size()); ok = yakamd_memcpy_d2h(h_t.data(), d_t, chunk.size() * 2) == 0; }
			for (size_t j = 0; ok && j < ns; ++j) {
				if (h_tot[j] == 0xffffffffu) continue;                          /* below min_len: qv.c:45 */
				if (opt->print_err_kmer)
					for (uint32_t i = 0; i < h_len[j]; ++i)
						if (h_t[h_off[j] + i] == 0) printf("EK\t%s\t%d\n", names[j].c_str(), (int)(i + 1 - ch->k));
				if (opt->print_each) {
					const int tot = (int)h_tot[j], non0 = (int)h_non0[j];
					double qv = -1.0;
					if (tot > 0) {
						if (non0 > 0) {
							if (tot > non0) { qv = log((double)tot / non0) / ch->k; qv = -4.3429448190325175 * log(qv); }
							else qv = 99.0;
						} else qv = 0.0;
					}
					printf("SQ\t%s\t%d\t%d\t%d\t%.2f\n", names[j].c_str(), (int)h_len[j], tot, non0, qv);
				}
			}
		}
		yakamd_dev_free(d_b); yakamd_dev_free(d_t); yakamd_dev_free(d_off); yakamd_dev_free(d_len); yakamd_dev_free(d_tot); yakamd_dev_free(d_non0);
		fprintf(stderr, "[M::%s] processed %ld sequences\n", "yak_qv", (long)ns);
		chunk.clear(); h_off.clear(); h_len.clear(); names.clear(); sum_len = 0;
	};
	/* without -p / -E nothing of a record but its bases is needed: a plain, block-gzipped or gzip file then goes through the parallel reader
	 * (every record's sequence + '\n', in order: bseq.c:40 keeps records of any length, qv.c:45 skips the short ones later) */
	const int n_thr = parse_threads(opt->n_threads);
	ByteSource psrc; int psrc_fd = -1;
	pgz::Reader *gz_p = new pgz::Reader;
	struct GzDrop { pgz::Reader *p; ~GzDrop() { pgz::Reader *q = p; std::thread([q]() { delete q; }).detach(); } } gz_drop{ gz_p };
	bool parallel = false;
	if (ok && !opt->print_each && !opt->print_err_kmer) {
		const ImgSink sink = [&](const char *img, size_t n, int64_t, const WinPack*) {
			const size_t base = chunk.size();
			for (const char *p = img, *e = img + n; p < e; ) {
				const char *q = (const char*)memchr(p, '\n', (size_t)(e - p));
				if (!q) q = e;
				h_off.push_back(base + (size_t)(p - img)); h_len.push_back((uint32_t)(q - p));
				sum_len += q - p;
				p = q + 1;
			}
			chunk.insert(chunk.end(), img, img + n);
			if (sum_len >= opt->chunk_size || chunk.size() > ((size_t)1 << 31)) flush();   /* bseq.c:54 */
			return ok;
		};
		if (parallel_source(fn, fx, n_thr, 1 << 20, &psrc, &psrc_fd)) { parallel = true; ok = parse_parallel(&psrc, 0, n_thr, sink) && ok; }
		else if (gz_source(fn, fx, n_thr, gz_p)) { parallel = true; ok = parse_gz(gz_p, 0, n_thr, sink) && ok; }
		if (psrc_fd >= 0) ::close(psrc_fd);
	}
	while (ok && !parallel && (l = fx.next()) >= 0) {         /* bseq.c:40 */
		h_off.push_back(chunk.size()); h_len.push_back((uint32_t)l);
		if (opt->print_each || opt->print_err_kmer) names.emplace_back(fx.name.begin(), fx.name.end());
		chunk.insert(chunk.end(), fx.seq.begin(), fx.seq.end());
		chunk.push_back('\n');
		sum_len += l;
		if (sum_len >= opt->chunk_size || chunk.size() > ((size_t)1 << 31)) flush();   /* bseq.c:54 */
	}
	if (ok) flush();
	std::vector<uint64_t> hh(n_cnt, 0);
	ok = ok && yakamd_memcpy_d2h(hh.data(), d_hist, n_cnt * 8) == 0;
	if (ok) for (int i = 0; i < n_cnt; ++i) cnt[i] = (int64_t)hh[i];
	else fprintf(stderr, "[E::yak_qv] %s\n", yakamd_last_error());
	yakamd_dev_free(d_hist);
	fx.close_file();
}

/* n x n linear system a x = b by Gauss-Jordan elimination with full pivoting (the solver the
 * reference links as 6gjdn.c); the solution replaces b.  false if the matrix is singular. */
static bool solve_full_pivot(double *a, double *b, int n)
{
	std::vector<int> col_of(n);
	for (int k = 0; k < n; ++k) {
		int pr = k, pc = k;
		double big = 0.0;
		for (int i = k; i < n; ++i)
			for (int j = k; j < n; ++j)
				if (fabs(a[i * n + j]) > big) { big = fabs(a[i * n + j]); pr = i; pc = j; }
		if (big + 1.0 == 1.0) return false;
		col_of[k] = pc;
		if (pc != k) for (int i = 0; i < n; ++i) std::swap(a[i * n + k], a[i * n + pc]);
		if (pr != k) { for (int j = k; j < n; ++j) std::swap(a[k * n + j], a[pr * n + j]); std::swap(b[k], b[pr]); }
		const double piv = a[k * n + k];
		for (int j = k + 1; j < n; ++j) a[k * n + j] /= piv;
		b[k] /= piv;
		for (int j = k + 1; j < n; ++j)
			for (int i = 0; i < n; ++i)
				if (i != k) a[i * n + j] -= a[i * n + k] * a[k * n + j];
		for (int i = 0; i < n; ++i)
			if (i != k) b[i] -= a[i * n + k] * b[k];
	}
	for (int k = n - 1; k >= 0; --k)                          /* undo the column exchanges */
		if (col_of[k] != k) std::swap(b[k], b[col_of[k]]);
	return true;
}

/* yak_qv_solve (reference qv.c:146-244): host arithmetic on two 1024-bin histograms -- in_table[c] = stored k-mers
 * that occur c times in the short reads, in_seqs[c] = k-mers of the assembly found with count c.  The statistics are
 * the reference's (raw QV from the share of absent k-mers; coverage at the histogram peak; bounds on the false-positive
 * rate of "absent"; corrected counts between the trough and the peak; successive ratios fitted by a parabola and
 * extrapolated to count 0; adjusted QV) and the printed digits must equal the reference's, so every floating-point
 * expression keeps the reference's operand order and association (sums run over ascending k, powers are built by
 * repeated multiplication).  That is the only thing shared with qv.c: the stages below are this file's own cut. */
} /* extern "C" */
namespace {
struct QvShape { int peak = -1, trough = -1; };

/* the mode of in_seqs over counts [2, 1023) and the lowest point in front of it */
QvShape qv_shape(const int64_t *in_seqs)
{
	QvShape s;
	int32_t top = 0;
	for (int c = 2; c < YAK_N_COUNTS - 1; ++c) if (top < in_seqs[c]) { top = (int32_t)in_seqs[c]; s.peak = c; }
	int32_t low = top;
	for (int c = 2; c < s.peak; ++c) if (low > in_seqs[c]) { low = (int32_t)in_seqs[c]; s.trough = c; }
	return s;
}

/* brackets the false-positive rate from the counts below the peak and clamps the caller's estimate into them */
double qv_fpr_bounds(const int64_t *in_table, const int64_t *in_seqs, const QvShape &s, double fpr, yak_qstat_t *qs)
{
	qs->fpr_upper = 1.0;
	for (int c = 2; c < s.peak; ++c) {
		const double e = in_seqs[c] / (qs->cov * in_table[c]);
		if (qs->fpr_upper > e) qs->fpr_upper = e;
	}
	if (fpr > qs->fpr_upper) fpr = qs->fpr_upper * 0.5;
	qs->fpr_lower = 0.0;
	if (s.trough > 2 && in_table[2] > in_table[s.trough]) {
		const double e = (in_seqs[2] - in_seqs[s.trough]) / (qs->cov * (in_table[2] - in_table[s.trough]));
		if (qs->fpr_lower < e) qs->fpr_lower = e;
	}
	if (fpr < qs->fpr_lower) fpr = qs->fpr_lower;
	if (qs->fpr_lower >= qs->fpr_upper)
		fprintf(stderr, "Warning: the FPR upper bound is smaller than the lower bound. Trust the lower bound.\n");
	return fpr;
}

/* least-squares polynomial of degree DEG through (x[k], y[k]), k < n, by the normal equations: coef[i] multiplies x^i.
 * pw[m][k] = x[k]^m by repeated multiplication; entry (i, j) of the matrix is the sum over k of pw[i + j][k] */
template <int DEG>
bool fit_polynomial(const double *x, const double *y, int n, double *coef)
{
	std::vector<double> pw((size_t)(2 * DEG + 1) * n);
	for (int k = 0; k < n; ++k) {
		double t = 1.0;
		for (int m = 0; m <= 2 * DEG; ++m) { pw[(size_t)m * n + k] = t; t *= x[k]; }
	}
	double M[(DEG + 1) * (DEG + 1)];
	for (int i = 0; i <= DEG; ++i) {
		for (int j = 0; j <= i; ++j) {
			double acc = 0.0;
			for (int k = 0; k < n; ++k) acc += pw[(size_t)(i + j) * n + k];
			M[i * (DEG + 1) + j] = M[j * (DEG + 1) + i] = acc;
		}
		double acc = 0.0;
		for (int k = 0; k < n; ++k) acc += pw[(size_t)i * n + k] * y[k];
		coef[i] = acc;
	}
	return solve_full_pivot(M, coef, DEG + 1);
}

template <int DEG> double eval_polynomial(const double *coef, double at)
{
	double r = 0.0, t = 1.0;
	for (int i = 0; i <= DEG; ++i) { r += coef[i] * t; t *= at; }
	return r;
}
} // namespace

extern "C" int yak_qv_solve(const int64_t *in_table, const int64_t *in_seqs, int kmer, double fpr, yak_qstat_t *qs)
{
	constexpr int DEG = 2, MAX_FIT = 8;
	const double db_per_ln = 4.3429448190325175;             /* 10 / ln 10 */
	memset(qs, 0, sizeof(*qs));
	for (int c = 0; c < YAK_N_COUNTS; ++c) { qs->tot += in_seqs[c]; qs->adj_cnt[c] = (double)in_seqs[c]; }
	qs->err = (double)in_seqs[0];
	qs->qv = -1.0;
	qs->qv_raw = (qs->tot > 0 && qs->tot > in_seqs[0]) ? -db_per_ln * log(log((double)qs->tot / (qs->tot - in_seqs[0])) / kmer) : -1.0;

	const QvShape s = qv_shape(in_seqs);
	if (s.peak < 0) return -1;                               /* nothing beyond count 1 */
	qs->cov = (double)in_seqs[s.peak] / in_table[s.peak];
	fpr = qv_fpr_bounds(in_table, in_seqs, s, fpr, qs);

	const int n_fit = std::min(MAX_FIT, s.peak - s.trough + 1);
	if (s.peak <= 4 || n_fit < 3) return -1;                 /* not high-coverage data: no adjustment */

	for (int c = s.peak - 1; c >= s.trough; --c) {           /* take the expected false "present" calls out */
		const double wrong = (in_table[c] - in_seqs[c] / qs->cov) / (1.0 - fpr);
		qs->adj_cnt[c] = in_seqs[c] - wrong * qs->cov * fpr;
		if (qs->adj_cnt[c] < 0.0) qs->adj_cnt[c] = 0.0;
	}

	double at[MAX_FIT], ratio[MAX_FIT], coef[DEG + 1];
	for (int k = 0; k < n_fit; ++k) { at[k] = s.trough + k; ratio[k] = qs->adj_cnt[s.trough + k + 1] / qs->adj_cnt[s.trough + k]; }
	if (!fit_polynomial<DEG>(at, ratio, n_fit, coef)) fprintf(stderr, "ERROR: fail\n");
	for (int c = s.trough - 1; c >= 0; --c) {                /* below the trough: divide down by the fitted ratio, at least 1.01 */
		double r = eval_polynomial<DEG>(coef, c);
		if (r < 1.01) r = 1.01;
		qs->adj_cnt[c] = qs->adj_cnt[c + 1] / r;
	}

	double adj_sum = 0.0;
	for (int c = 0; c < YAK_N_COUNTS; ++c) adj_sum += qs->adj_cnt[c];
	if (adj_sum <= (double)qs->tot) {
		qs->err = qs->tot - adj_sum;
		qs->qv = -db_per_ln * log(log(qs->tot / adj_sum) / kmer);
	} else {
		fprintf(stderr, "WARNING: failed to estimate the calibrated QV\n");
		qs->err = 0;
		qs->qv = qs->qv_raw;
	}
	return 0;
}


