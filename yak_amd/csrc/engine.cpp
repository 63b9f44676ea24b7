/*
 * engine.cpp -- host orchestration of the MI355X counting engine behind yak.h.
 *
 * State model.  The authoritative copy of a counting table lives in HBM as an exact image of the
 * reference's 1<<pre khashl sets (one arena of 64-bit slots + one "used" bitmap, reference
 * khashl.h:104-109 / htab.c:9-11) together with the blocked bloom filters (bbf.c).  The slot arrays
 * reachable from the caller-visible yak_ch_t are a host mirror that is refreshed on demand
 * (yak_ch_dump, yak_ch_get, ...).  There is NO CPU counting path: without a usable GPU every
 * entry point fails loudly.
 *
 * A counting pass (reference count.c:147-166) is: pass_begin, feed..., pass_end.  For
 * create_new = 1 the pass accumulates, per distinct hashed k-mer, its count and the stream times
 * of its first and second occurrence (kernels.hip K1/K3), applies the order-exact bloom gate
 * (K2), and pass_end turns that into the reference's slot layout (select -> sort -> K5 replay).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <map>
#include <string>
#include <mutex>
#include <algorithm>
#include <chrono>
#include <functional>
#include "engine_int.h"

static thread_local char g_err[512] = "";


extern "C" const char *yakamd_last_error(void) { return g_err; }
int yk_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); fprintf(stderr, "[E::yak_amd] %s\n", g_err); return -1; }

extern "C" int yakamd_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	int ok = 0;
	for (int i = 0; i < n; ++i) {
		hipDeviceProp_t pr;
		if (hipGetDeviceProperties(&pr, i) == hipSuccess && strncmp(pr.gcnArchName, "gfx950", 6) == 0) ++ok;
	}
	return ok;
}


/* ------------------------------------------------------------------------------------------ */


static yakamd_ctx *ctx_of(const yak_ch_t *h)
{
	const yak_ch_ext *e = (const yak_ch_ext*)h;
	return (h && e->magic == EXT_MAGIC) ? e->ctx : 0;
}
extern "C" yakamd_ctx *yakamd_ctx_of(yak_ch_t *h) { return ctx_of(h); }

static ImgView img_view(yakamd_ctx *c)
{
	ImgView v;
	v.bits = c->d_bits; v.off = c->d_off; v.keys = c->d_keys; v.used = c->d_used; v.delta = c->d_delta;
	v.pre = c->pre; v.k = c->k;
	return v;
}

static BloomView bloom_view(yakamd_ctx *c)
{
	BloomView b;
	b.bits32 = c->d_bf; b.nb = c->nb; b.n_hash = c->n_hash;
	const int nd = c->n_hash < 512 ? c->n_hash : 512;
	b.mw = (nd + 63) / 64; if (b.mw < 1) b.mw = 1;
	return b;
}

/* install a fresh, empty image: every sub-table has capacity 0 but owns 32 arena slots */
static int img_reset_empty(yakamd_ctx *c)
{
	const int P = c->P;
	dfree(c->d_keys); dfree(c->d_used); dfree(c->d_delta);
	c->n_slots = (u64)P * 32;
	if (dmalloc(&c->d_keys, c->n_slots) || dmalloc(&c->d_used, c->n_slots / 32)) return -1;   /* d_delta: see delta_ensure */
	HIPCK(hipMemsetAsync(c->d_keys, 0xff, c->n_slots * 8, c->st));
	HIPCK(hipMemsetAsync(c->d_used, 0, c->n_slots / 8, c->st));
	c->h_bits.assign(P, YK_NOCAP); c->h_count.assign(P, 0); c->h_off.resize(P);
	for (int p = 0; p < P; ++p) c->h_off[p] = (u64)p * 32;
	HIPCK(hipMemcpyAsync(c->d_bits, c->h_bits.data(), P * 4, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(c->d_off, c->h_off.data(), P * 8, hipMemcpyHostToDevice, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	c->img_keys_total = 0;
	c->host_valid = false;
	return 0;
}

/* the per-slot pending increments (4 bytes per slot of the arena: 8.6 GB beside a 2 Gb assembly's table) exist only while a pass can add to the
 * counts of stored keys -- a count-existing pass, or puts of the accumulator path on a table that holds keys -- and are folded in and released when
 * that pass ends */
static int delta_ensure(yakamd_ctx *c)
{
	if (c->d_delta) return 0;
	if (dmalloc(&c->d_delta, c->n_slots)) return -1;
	HIPCK(hipMemsetAsync(c->d_delta, 0, c->n_slots * 4, c->st));
	return 0;
}

static hipStream_t stream_get(int dev);
static void stream_put(int dev, hipStream_t x);
static thread_local int g_next_device = -1;          /* device of the next context created on this thread (multi-GPU tables) */
void yk_ctx_next_device(int dev) { g_next_device = dev; }

yakamd_ctx *yk_ctx_create(int k, int pre, int n_hash, int n_shift)
{
	if (yakamd_device_count() < 1) { fail("no gfx950 GPU visible: the counting engine has no CPU fallback"); return 0; }
	yakamd_ctx *c = new yakamd_ctx();
	memset(&c->st_cur, 0, sizeof(c->st_cur)); memset(&c->st_last, 0, sizeof(c->st_last));
	c->k = k; c->pre = pre; c->P = 1 << pre; c->plo = 0; c->phi = c->P;
	c->n_hash = 0; c->bf_shift = 0; c->nb = 0; c->has_bloom = false;
	c->d_bits = 0; c->d_used = 0; c->d_delta = 0; c->d_off = 0; c->d_keys = 0; c->n_slots = 0;
	c->d_bf = 0; c->bf_words = 0; c->d_multi = 0; c->multi_bits = 0; c->bf_virgin = false; c->bf_deferred = false;
	c->in_pass = false; c->gate_off = false; c->or_mode = 0; c->acc.s = 0; c->acc_count = 0;
	c->d_counters = 0; c->d_lastput = 0; c->d_lpbatch = 0; c->d_missing = 0; c->d_nmissing = 0;
	c->d_rec = 0; c->rec_cap = 0; c->d_newlist = 0; c->d_miss = 0; c->d_cand = 0; c->new_cap = 0;
	c->d_stage = 0; c->stage_cap = 0; c->t_end = 0; c->list_t = 0; c->d_scratch = 0; c->scratch_bytes = 0;
	c->d_rows = 0; c->d_partial = 0; c->d_bstart = 0; c->rows_blk = 0;
	c->nb_bits = (int)std::min<int64_t>(pre, env_i64("YAKAMD_PART_BITS", 13));
	if (c->nb_bits < 3) c->nb_bits = 3;
	if (c->nb_bits > 13) c->nb_bits = 13;
	c->fast = false; c->kept_bytes = 0; c->fast_budget = 0; c->t_pass0 = 0; c->t_pass0_set = false;
	c->host_valid = false; c->hm_keys = 0; c->hm_used = 0; c->hm_slots = 0; c->hts = 0;
	c->retain_on = false; c->retain_broken = false; c->retained_bytes = 0; c->src_set = false;
	memset(&c->ret2, 0, sizeof(c->ret2)); c->n_slices = 0;
	c->dev = (int)env_i64("YAKAMD_DEVICE", 0);
	{
		const char *lr = getenv("LOCAL_RANK");
		int nd = 0;
		if (!getenv("YAKAMD_DEVICE") && lr && hipGetDeviceCount(&nd) == hipSuccess && nd > 0) c->dev = atoi(lr) % nd;
	}
	if (g_next_device >= 0) { c->dev = g_next_device; g_next_device = -1; }
	if (hipSetDevice(c->dev) != hipSuccess || (c->st = stream_get(c->dev)) == 0) { fail("cannot open device %d", c->dev); delete c; return 0; }
	if (n_hash > 0 && n_shift > pre) {                       /* reference htab.c:23-27 */
		c->n_hash = n_hash; c->bf_shift = n_shift; c->nb = n_shift - pre;
		c->has_bloom = c->nb >= 9 && c->nb + 9 <= 64;        /* yak_bf_init returns NULL otherwise (bbf.c:9) */
	}
	if (dmalloc(&c->d_bits, c->P) || dmalloc(&c->d_off, c->P) || dmalloc(&c->d_counters, YKC_N) ||
	    dmalloc(&c->d_lastput, c->P) || dmalloc(&c->d_lpbatch, c->P) || dmalloc(&c->d_missing, (c->P + 31) / 32 + 1) ||
	    dmalloc(&c->d_nmissing, 1) || img_reset_empty(c)) { yk_ctx_destroy(c); return 0; }
	if (c->has_bloom) {
		c->bf_words = (size_t)c->P << (c->nb - 5);
		if (dmalloc(&c->d_bf, c->bf_words)) { yk_ctx_destroy(c); return 0; }
		c->bf_virgin = true;               /* zero-filled lazily: the first counting pass writes every block anyway */
		c->multi_bits = (int)env_i64("YAKAMD_MULTI_BITS", 30);
		if (c->multi_bits < 10) c->multi_bits = 10;
		if (dmalloc(&c->d_multi, (size_t)1 << (c->multi_bits - 5))) { yk_ctx_destroy(c); return 0; }
		hipMemsetAsync(c->d_multi, 0, (size_t)1 << (c->multi_bits - 3), c->st);
		hipStreamSynchronize(c->st);
	}
	return c;
}

/* a filter whose write-back was skipped becomes real before the records it can be rebuilt from go away */
static void bloom_undefer(yakamd_ctx *c)
{
	if (!c->bf_deferred) return;
	c->bf_deferred = false;
	if (!c->d_bf || !c->ret2.valid) return;
	(void)hipMemsetAsync(c->d_bf, 0, c->bf_words * 4, c->st);
	yk_launch_bf_rebuild(c->ret2.fp, c->ret2.d_sbstart, c->ret2.d_r2, c->d_bf, c->st);
	(void)hipStreamSynchronize(c->st);
	if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] the filter of the last pass rebuilt from its retained records (its write-back had been skipped)\n");
}

static void retained_drop(yakamd_ctx *c)
{
	bloom_undefer(c);
	for (auto &r : c->retained) dfree(r.d_rec);
	c->retained.clear(); c->retained_bytes = 0; c->src_set = false;
	dfree(c->ret2.d_r2); dfree(c->ret2.d_sbstart); dfree(c->ret2.d_koff); dfree(c->ret2.d_kkc); dfree(c->ret2.d_segbase);
	c->ret2.valid = false; c->ret2.n_total = 0;
}

static void pass_free(yakamd_ctx *c)
{
	for (auto &k : c->kept) if (k.owned) dfree(k.d_rec);
	c->kept.clear(); c->kept_bytes = 0;
	dfree(c->acc.s); c->acc_count = 0;
	dfree(c->d_rec); c->rec_cap = 0;
	dfree(c->d_newlist); dfree(c->d_miss); dfree(c->d_cand); c->new_cap = 0;
	c->in_pass = false;
}

void yk_ctx_destroy(yakamd_ctx *c)
{
	if (!c) return;
	hipSetDevice(c->dev);
	pass_free(c);
	c->bf_deferred = false;                                    /* nobody will read it */
	retained_drop(c);
	dfree(c->d_stage); dfree(c->d_rows); dfree(c->d_partial); dfree(c->d_bstart);
	{ uint8_t *q = (uint8_t*)c->d_scratch; dfree(q); c->d_scratch = 0; }
	dfree(c->d_bits); dfree(c->d_used); dfree(c->d_delta); dfree(c->d_off); dfree(c->d_keys);
	dfree(c->d_bf); dfree(c->d_multi);
	dfree(c->d_counters); dfree(c->d_lastput); dfree(c->d_lpbatch); dfree(c->d_missing); dfree(c->d_nmissing);
	if (c->hm_keys) hipHostFree(c->hm_keys);
	if (c->hm_used) hipHostFree(c->hm_used);
	free(c->hts);
	if (c->st) stream_put(c->dev, c->st);
	delete c;
}

int yk_ctx_destroy_bf(yakamd_ctx *c)
{
	hipSetDevice(c->dev);
	dfree(c->d_bf); dfree(c->d_multi);
	c->has_bloom = false; c->bf_deferred = false;
	return 0;
}

/* ------------------------------------------------------------------------------------------ */

static int acc_alloc(yakamd_ctx *c, AccTab *t, int bits)
{
	t->bits = bits; t->pre = c->pre; t->mask = (1ull << bits) - 1;
	if (dmalloc(&t->s, (size_t)1 << bits)) return -1;
	yk_launch_acc_init(t->s, 1ull << bits, c->st);
	return 0;
}

/* make room for `incoming` more distinct keys at a load factor <= 0.6 */
static int acc_reserve(yakamd_ctx *c, u64 incoming)
{
	const u64 need = c->acc_count + incoming;
	int bits = c->acc.s ? c->acc.bits : 0;
	int want = std::max(c->pre + 2, ceil_log2_u64((u64)(need / 0.6) + 1));
	if (want < 12) want = 12;
	if (c->acc.s && bits >= want) return 0;
	AccTab nt;
	if (acc_alloc(c, &nt, want)) return -1;
	if (c->acc.s) {
		yk_launch_acc_rehash(c->acc, nt, c->st);
		HIPCK(hipStreamSynchronize(c->st));
		dfree(c->acc.s);
	}
	c->acc = nt;
	return 0;
}

extern "C" int yakamd_set_shard(yak_ch_t *h, int lo, int hi)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || lo < 0 || hi > c->P || lo >= hi) return fail("bad shard range");
	if (c->in_pass) return fail("shard change inside a pass");
	if (c->has_bloom && !c->bf_virgin && (lo != c->plo || hi != c->phi))
		return fail("the shard cannot move once its bloom filters have been written (only the shard's part is initialised)");
	c->plo = lo; c->phi = hi;
	return 0;
}

extern "C" int yakamd_pass_begin(yak_ch_t *h, int create_new)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c) return fail("not an engine table");
	if (c->in_pass) return fail("pass already open");
	HIPCK(hipSetDevice(c->dev));
	(void)hipGetLastError();                                   /* whatever other users of the runtime left behind is not this pass's (see yakamd_pass_end) */
	c->create_new = create_new;
	c->delta_dirty = false;
	if (!create_new && delta_ensure(c)) return -1;
	if (create_new) { retained_drop(c); c->retain_broken = false; }
	c->n_slices = 0;
	c->bloom_mode = create_new && c->has_bloom && !c->gate_off;
	c->in_pass = true;
	c->t_end = 0;
	memset(&c->st_cur, 0, sizeof(c->st_cur));
	HIPCK(hipMemsetAsync(c->d_counters, 0, YKC_N * 8, c->st));
	HIPCK(hipMemsetAsync(c->d_lastput, 0, c->P * 8, c->st));
	c->st_cur.ms_total = now_ms();
	/* exclusive-ownership LDS counting needs level-1 buckets == sub-tables and 2-bit k-mers */
	c->fast = create_new && env_i64("YAKAMD_FAST", 1) != 0 && c->nb_bits == c->pre && (!c->has_bloom || c->nb <= 42);
	c->kept_bytes = 0; c->t_pass0_set = false; c->ms_part2 = c->ms_lds = 0; c->keys_at_begin = c->img_keys_total;
	if (c->fast) {
		size_t fr = 0, tot = 0;
		HIPCK(hipMemGetInfo(&fr, &tot));
		c->fast_budget = (u64)env_i64("YAKAMD_FAST_BUDGET", (int64_t)((fr + yk_pool_cached_bytes()) / 4));
	}
	return 0;
}

static int bloom_materialise(yakamd_ctx *c);

/* events and streams are kept and handed out again: creating and destroying them costs a runtime call each (a millisecond per stream with the
 * ROCm 7.2 runtime), and a counting job of 60 ms opens a table per step and times a dozen stages */
struct HipCache {
	std::mutex mu;
	std::vector<hipEvent_t> ev[16];
	std::vector<hipStream_t> st[16];
};
static HipCache g_hc;
static hipEvent_t ev_get()
{
	int d = 0; (void)hipGetDevice(&d); d &= 15;
	{ std::lock_guard<std::mutex> lk(g_hc.mu); if (!g_hc.ev[d].empty()) { hipEvent_t e = g_hc.ev[d].back(); g_hc.ev[d].pop_back(); return e; } }
	hipEvent_t e = 0; (void)hipEventCreate(&e); return e;
}
static void ev_put(hipEvent_t e) { int d = 0; (void)hipGetDevice(&d); d &= 15; std::lock_guard<std::mutex> lk(g_hc.mu); g_hc.ev[d].push_back(e); }
static hipStream_t stream_get(int dev)
{
	{ std::lock_guard<std::mutex> lk(g_hc.mu); auto &v = g_hc.st[dev & 15]; if (!v.empty()) { hipStream_t x = v.back(); v.pop_back(); return x; } }
	hipStream_t x = 0;
	return hipStreamCreate(&x) == hipSuccess ? x : 0;
}
static void stream_put(int dev, hipStream_t x) { (void)hipStreamSynchronize(x); std::lock_guard<std::mutex> lk(g_hc.mu); g_hc.st[dev & 15].push_back(x); }

struct EvTimer {
	hipEvent_t a, b; hipStream_t st;
	EvTimer(hipStream_t s) : st(s) { a = ev_get(); b = ev_get(); hipEventRecord(a, st); }
	double stop() { float ms = 0; hipEventRecord(b, st); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); return ms; }
	~EvTimer() { ev_put(a); ev_put(b); }
};

/* bloom gate over the keys first seen in the batch just inserted (kernels.hip K2) */
static int bloom_phases(yakamd_ctx *c, u64 n_new)
{
	if (n_new == 0) return 0;
	const BloomView bf = bloom_view(c);
	u64 h_cnt[YKC_N];
	EvTimer tm(c->st);
	HIPCK(hipMemsetAsync(c->d_counters + YKC_ANYMULTI, 0, 3 * 8, c->st));   /* ANYMULTI, NCAND, NMARKED */
	yk_launch_bf_test(c->acc, c->d_newlist, n_new, bf, c->d_miss, c->st);
	yk_launch_bf_set(c->acc, c->d_newlist, n_new, bf, c->d_miss, c->d_multi, c->multi_bits, c->d_counters, c->st);
	HIPCK(hipMemcpyAsync(h_cnt, c->d_counters, sizeof(h_cnt), hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	if (h_cnt[YKC_ANYMULTI]) {
		yk_launch_bf_check(c->acc, c->d_newlist, n_new, bf, c->d_miss, c->d_multi, c->multi_bits, c->d_cand, c->d_counters, c->st);
		HIPCK(hipMemcpyAsync(h_cnt, c->d_counters, sizeof(h_cnt), hipMemcpyDeviceToHost, c->st));
		HIPCK(hipStreamSynchronize(c->st));
		const u64 n_cand = h_cnt[YKC_NCAND], n_marked = h_cnt[YKC_NMARKED];
		if (n_cand) {
			const int map_bits = std::max(10, ceil_log2_u64(2 * n_marked + 2));
			u64 *d_map = 0;
			if (dmalloc(&d_map, (size_t)2 << map_bits)) return -1;
			HIPCK(hipMemsetAsync(d_map, 0, (size_t)16 << map_bits, c->st));
			yk_launch_bf_mapfill(c->acc, c->d_newlist, n_new, bf, c->d_miss, c->d_multi, c->multi_bits, d_map, map_bits, c->st);
			yk_launch_bf_resolve(c->acc, c->d_newlist, c->d_cand, n_cand, bf, c->d_miss, d_map, map_bits, c->st);
			HIPCK(hipStreamSynchronize(c->st));
			dfree(d_map);
			c->st_cur.n_bloom_candidates += (int64_t)n_cand;
		}
		HIPCK(hipMemsetAsync(c->d_multi, 0, (size_t)1 << (c->multi_bits - 3), c->st));
	}
	c->st_cur.ms_bloom += tm.stop();
	return 0;
}

/* per sub-table time of the last put-call of this batch (see k_lastput) */
static int lastput_phase(yakamd_ctx *c, int64_t n_rec, u64 t0, u64 batch_lo, u64 batch_hi, int img_nonempty)
{
	const int64_t tail = env_i64("YAKAMD_LASTPUT_TAIL", 4 << 20);
	const u64 t_from = (batch_hi - batch_lo > (u64)tail) ? batch_hi - (u64)tail : batch_lo;
	const ImgView img = img_view(c);
	u32 n_missing = 0;
	HIPCK(hipMemsetAsync(c->d_lpbatch, 0, c->P * 8, c->st));
	HIPCK(hipMemsetAsync(c->d_missing, 0, ((c->P + 31) / 32) * 4, c->st));
	HIPCK(hipMemsetAsync(c->d_nmissing, 0, 4, c->st));
	yk_launch_lastput(c->d_rec, n_rec, t0, t_from, c->acc, img, img_nonempty, c->bloom_mode, 0, c->d_lpbatch, c->st);
	yk_launch_lastput_merge(c->d_lastput, c->d_lpbatch, c->d_missing, c->d_nmissing, c->P, c->plo, c->phi, c->st);
	if (t_from > batch_lo) {
		HIPCK(hipMemcpyAsync(&n_missing, c->d_nmissing, 4, hipMemcpyDeviceToHost, c->st));
		HIPCK(hipStreamSynchronize(c->st));
		if (n_missing) {    /* the tail did not reach every sub-table: scan the whole batch for those */
			yk_launch_lastput(c->d_rec, n_rec, t0, batch_lo, c->acc, img, img_nonempty, c->bloom_mode, c->d_missing, c->d_lpbatch, c->st);
			HIPCK(hipMemsetAsync(c->d_nmissing, 0, 4, c->st));
			yk_launch_lastput_merge(c->d_lastput, c->d_lpbatch, c->d_missing, c->d_nmissing, c->P, c->plo, c->phi, c->st);
		}
	}
	return 0;
}

static int rec_reserve(yakamd_ctx *c, int64_t n)
{
	if (n <= c->rec_cap) return 0;
	dfree(c->d_rec);
	c->rec_cap = n;
	return dmalloc(&c->d_rec, (size_t)n);
}

static int part_reserve(yakamd_ctx *c, int n_blk)
{
	const size_t NB = (size_t)1 << c->nb_bits;
	if (!c->d_partial && (dmalloc(&c->d_partial, NB * yk_part_groups()) || dmalloc(&c->d_bstart, NB + 1))) return -1;
	if (n_blk <= c->rows_blk) return 0;
	dfree(c->d_rows);
	c->rows_blk = n_blk;
	return dmalloc(&c->d_rows, NB * (size_t)n_blk);
}

static int new_reserve(yakamd_ctx *c, int64_t n)
{
	if (!c->bloom_mode || n <= c->new_cap) return 0;
	dfree(c->d_newlist); dfree(c->d_miss); dfree(c->d_cand);
	c->new_cap = n;
	const BloomView bf = bloom_view(c);
	return dmalloc(&c->d_newlist, (size_t)n) || dmalloc(&c->d_miss, (size_t)n * bf.mw) || dmalloc(&c->d_cand, (size_t)n) ? -1 : 0;
}

/* records [0, n_rec) of d_rh/d_rt (times = t0 + d_rt[i], all within [batch_lo, batch_hi)) -> table */
/* LDS bytes of the exclusive-ownership pass-2 counter for the current table image (0: not eligible) */
static size_t count_lds_bytes(yakamd_ctx *c)
{
	size_t lds = 0;
	if (c->nb_bits != c->pre || env_i64("YAKAMD_COUNT_LDS", 1) == 0) return 0;
	for (int p = c->plo; p < c->phi; ++p)
		if (c->h_bits[p] != YK_NOCAP) lds = std::max(lds, yk_img_count_lds_bytes(1u << c->h_bits[p], c->h_count[p]));
	return lds > 150 * 1024 ? 0 : lds;
}

/* count-existing pass, records = 8-byte hashes grouped by prefix (d_bstart), tables beyond the LDS
 * kernel: level-2 partition by home-slot range (k_hpart2) + one workgroup per range (k_img_count_rng).
 * Returns 0 done, -1 error, 1 not applicable (caller falls back to the device-atomics kernel). */
static int count_by_ranges(yakamd_ctx *c, int64_t n_rec, const u64 *d_bstart)
{
	const int P = c->P;
	u32 bmax = 0;
	for (int p = c->plo; p < c->phi; ++p) if (c->h_bits[p] != YK_NOCAP) bmax = std::max(bmax, c->h_bits[p]);
	const int rb = (int)bmax > yk_rng_log() ? (int)bmax - yk_rng_log() : 0;
	if (bmax == 0 || rb > 8) return 1;
	const size_t NB = (size_t)1 << c->nb_bits, S2 = (size_t)1 << rb, n_sb = (size_t)P << rb;
	std::vector<u64> bst(NB + 1);
	HIPCK(hipMemcpyAsync(bst.data(), d_bstart, (NB + 1) * 8, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	const u64 ch = (u64)yk_hpart2_chunk();
	std::vector<Chunk2> chunks;
	std::vector<u32> chunk_first(P + 1, 0);
	std::vector<u64> bbase(P + 1, 0);
	for (int p = 0; p < P; ++p) {
		chunk_first[p] = (u32)chunks.size();
		const u64 a = bst[p], b = bst[p + 1];
		for (u64 o = a; o < b; o += ch) {
			Chunk2 k;
			k.rec = (const Rec*)((const u64*)c->d_rec + o); k.spare = 0;
			k.n = (u32)std::min<u64>(ch, b - o); k.bucket = (u32)p; k.tbase = 0; k.pad = 0;
			chunks.push_back(k);
		}
		if (chunks.size() > chunk_first[p]) chunks.back().spare = 1;
		bbase[p + 1] = bbase[p] + (b - a);
	}
	chunk_first[P] = (u32)chunks.size();
	Chunk2 *d_chunks = 0; u32 *d_cf = 0, *d_rows2 = 0, *d_ln = 0; u64 *d_bbase = 0, *d_sbstart = 0, *d_h2 = 0, *d_list = 0;
	struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ [&]() { dfree(d_chunks); dfree(d_cf); dfree(d_bbase); dfree(d_rows2); dfree(d_sbstart); dfree(d_h2); dfree(d_list); dfree(d_ln); } };
	const u32 list_cap = (u32)std::min<int64_t>(env_i64("YAKAMD_XLIST_CAP", 1 << 22), 1 << 22);   /* the knob is for tests */
	if (dmalloc(&d_chunks, chunks.size()) || dmalloc(&d_cf, P + 1) || dmalloc(&d_bbase, P + 1) || dmalloc(&d_rows2, (chunks.size() + 1) * S2) ||
	    dmalloc(&d_sbstart, n_sb + 1) || dmalloc(&d_h2, (size_t)n_rec) || dmalloc(&d_list, (size_t)1 << 22) || dmalloc(&d_ln, 2)) return -1;
	HIPCK(hipMemcpyAsync(d_chunks, chunks.data(), chunks.size() * sizeof(Chunk2), hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(d_cf, chunk_first.data(), (P + 1) * 4, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(d_bbase, bbase.data(), (P + 1) * 8, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemsetAsync(d_ln, 0, 8, c->st));
	const ImgView img = img_view(c);
	yk_launch_hpart2(d_chunks, (int)chunks.size(), d_cf, d_bbase, img, rb, P, d_rows2, d_sbstart, d_h2, c->st);
	const u32 max_len = bmax > (u32)yk_rng_log() ? 1u << yk_rng_log() : 1u << bmax;
	int r = 0;
	if (yk_launch_img_count_rng(d_h2, 0, d_sbstart, img, c->plo, c->phi, rb, max_len, d_list, d_ln, list_cap, c->st)) r = fail("range count kernel could not be configured");
	u32 xn[2] = { 0, 0 };
	if (!r) {
		HIPCK(hipMemcpyAsync(xn, d_ln, 8, hipMemcpyDeviceToHost, c->st));
		HIPCK(hipStreamSynchronize(c->st));
		if (xn[1]) yk_launch_img_count_rng(d_h2, 1, d_sbstart, img, c->plo, c->phi, rb, max_len, d_list, d_ln, list_cap, c->st);   /* list too small: second sweep */
		else if (xn[0]) yk_launch_img_count_h(d_list, (int64_t)xn[0], img, c->st);
		if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] range count: 2^%d ranges per sub-table, %u boundary-crossing instances%s\n", rb, xn[0], xn[1] ? " (list overflow: second sweep)" : "");
		HIPCK(hipStreamSynchronize(c->st));
	}
	return r;
}

/* count-existing pass, records grouped by prefix: one workgroup per slot range with the range's keys in LDS
 * (k_img_count_own).  Returns 0 done, -1 error, 1 not applicable. */
/* the ranges of k_img_count_own for the table as it stands: 1 = not applicable, 0 = go (rb == -1: no sub-table has a slot) */
static int count_own_plan(yakamd_ctx *c, int *rng_log_out, int *rb_out, u32 *kmax_out)
{
	if (env_i64("YAKAMD_COUNT_OWN", 1) == 0 || c->nb_bits != c->pre) return 1;
	const size_t budget = (size_t)env_i64("YAKAMD_OWN_LDS", 156000);     /* one workgroup per CU with as few ranges as possible: every range re-reads the sub-table's records (measured: 8 ranges x 1 WG/CU 16.9 ms, 16 ranges x 2 WG/CU 24.8 ms) */
	u32 bmax = 0;
	for (int p = c->plo; p < c->phi; ++p) if (c->h_bits[p] != YK_NOCAP) bmax = std::max(bmax, c->h_bits[p]);
	*rng_log_out = 0; *rb_out = -1; *kmax_out = 0;
	if (bmax == 0) return 0;                                         /* no sub-table has a slot: nothing can be found */
	int rng_log = -1; u32 kmax = 0;
	for (int rl = (int)bmax; rl >= 5 && rng_log < 0; --rl) {
		u32 km = 1;
		for (int p = c->plo; p < c->phi; ++p) {
			if (c->h_bits[p] == YK_NOCAP) continue;
			const int rbp = (int)c->h_bits[p] > rl ? (int)c->h_bits[p] - rl : 0;
			const u32 e = (c->h_count[p] + (1u << rbp) - 1) >> rbp;
			km = std::max(km, rbp ? e + e / 8 + 192 : c->h_count[p]);   /* a range's share of the keys + 12 % + 6 sigma at small counts */
		}
		if (yk_count_own_lds(1u << rl, km) <= budget) { rng_log = rl; kmax = km; }
	}
	const int rb = rng_log < 0 ? 99 : (int)bmax - rng_log;
	if (rb > (int)env_i64("YAKAMD_OWN_MAXRB", 4)) return 1;        /* every range's workgroup streams the whole sub-table: too redundant beyond 16 ranges */
	*rng_log_out = rng_log; *rb_out = rb; *kmax_out = kmax;
	return 0;
}

/* ytag: the records are k_xpart_wcs<false>'s with the top bits of the home-slot product in place of the sub-table's own bits */
static int count_own(yakamd_ctx *c, int64_t n_rec, const u64 *d_bstart, int hash_only, int ytag)
{
	int rng_log, rb; u32 kmax;
	const int pl = count_own_plan(c, &rng_log, &rb, &kmax);
	if (pl) return ytag ? fail("the table changed between the extraction and the count of a pass") : 1;
	if (rb < 0) return 0;
	u64 *d_list = 0; u32 *d_ln = 0;
	const u32 list_cap = (u32)std::min<int64_t>(env_i64("YAKAMD_XLIST_CAP", 1 << 22), 1 << 22);   /* the knob is for tests */
	if (dmalloc(&d_list, (size_t)1 << 22) || dmalloc(&d_ln, 2)) { dfree(d_list); return -1; }
	int r = 0;
	const ImgView img = img_view(c);
	const size_t lds = yk_count_own_lds(1u << rng_log, kmax);
	if (hipMemsetAsync(d_ln, 0, 8, c->st) != hipSuccess) r = fail("memset");
	if (!r && yk_launch_img_count_own(c->d_rec, hash_only, 0, ytag, d_bstart, img, c->plo, c->phi, rb, rng_log, kmax, lds, d_list, d_ln, list_cap, c->st)) r = fail("key-owning count kernel could not be configured");
	u32 xn[2] = { 0, 0 };
	if (!r && (hipMemcpyAsync(xn, d_ln, 8, hipMemcpyDeviceToHost, c->st) != hipSuccess || hipStreamSynchronize(c->st) != hipSuccess)) r = fail("key-owning count kernel failed: %s", hipGetErrorString(hipGetLastError()));
	if (!r) {
		if (xn[1]) yk_launch_img_count_own(c->d_rec, hash_only, 1, ytag, d_bstart, img, c->plo, c->phi, rb, rng_log, kmax, lds, d_list, d_ln, list_cap, c->st);   /* list too small: second sweep */
		else if (xn[0]) yk_launch_img_count_h(d_list, (int64_t)xn[0], img, c->st);
		if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] key-owning count: 2^%d slots per range (up to 2^%d ranges per sub-table), %u keys of LDS room, %zu B of LDS, %u boundary-crossing instances%s\n",
		                                          rng_log, rb, kmax, lds, xn[0], xn[1] ? " (list overflow: second sweep)" : "");
		if (hipStreamSynchronize(c->st) != hipSuccess) r = fail("count sweep failed");
	}
	(void)n_rec;
	dfree(d_list); dfree(d_ln);
	return r;
}

static int consume_records(yakamd_ctx *c, int64_t n_rec, u64 t0, u64 batch_lo, u64 batch_hi, const u64 *d_bstart, int hash_only, int ytag)
{
	if (n_rec <= 0) return 0;
	const ImgView img = img_view(c);
	c->st_cur.n_instances += n_rec;
	if (batch_hi > c->t_end) c->t_end = batch_hi;
	if (!c->create_new) {
		c->delta_dirty = true;
		EvTimer tm(c->st);
		/* records grouped by sub-table and every sub-table small enough for LDS rank counters:
		 * exclusive-ownership counting, no global atomics */
		if (d_bstart) {
			const int r = count_own(c, n_rec, d_bstart, hash_only, ytag);
			if (r <= 0 || ytag) {
				const double ms = tm.stop();
				c->st_cur.ms_insert += ms; c->st_cur.ms_dominant_kernel += ms; c->st_cur.n_dominant_launches += 1;
				return r;
			}
		}
		const size_t lds = d_bstart ? count_lds_bytes(c) : 0;
		if (!lds && d_bstart && hash_only && c->nb_bits == c->pre && env_i64("YAKAMD_COUNT_RNG", 1) != 0) {
			/* sub-tables too large for one workgroup's LDS: split each one's hashes by home-slot range first */
			const int r = count_by_ranges(c, n_rec, d_bstart);
			if (r <= 0) {                                        /* done (0) or failed (-1); 1 = not applicable */
				const double ms = tm.stop();
				c->st_cur.ms_insert += ms; c->st_cur.ms_dominant_kernel += ms; c->st_cur.n_dominant_launches += 1;
				return r;
			}
		}
		u64 *d_ck = 0; u32 ck_stride = 1;
		if (lds) {
			for (int p = c->plo; p < c->phi; ++p) ck_stride = std::max(ck_stride, c->h_count[p]);
			if (dmalloc(&d_ck, (size_t)(c->phi - c->plo) * ck_stride)) return -1;
		}
		const bool lds_done = lds && yk_launch_img_count_lds(c->d_rec, hash_only, d_bstart, img, c->plo, c->phi, lds, d_ck, ck_stride, c->st) == 0;
		if (d_ck) { HIPCK(hipStreamSynchronize(c->st)); dfree(d_ck); }
		if (!lds_done) {
			if (hash_only) yk_launch_img_count_h((const u64*)c->d_rec, n_rec, img, c->st);
			else yk_launch_img_count(c->d_rec, n_rec, img, c->st);
		}
		const double ms = tm.stop();
		c->st_cur.ms_insert += ms; c->st_cur.ms_dominant_kernel += ms; c->st_cur.n_dominant_launches += 1;
		return 0;
	}
	const int img_nonempty = c->img_keys_total > 0;
	if (img_nonempty) { if (delta_ensure(c)) return -1; c->delta_dirty = true; }
	if (c->bloom_mode && bloom_materialise(c)) return -1;
	if (acc_reserve(c, (u64)n_rec) || new_reserve(c, n_rec)) return -1;
	u64 h_cnt[YKC_N];
	{
		EvTimer tm(c->st);
		yk_launch_acc_insert(c->d_rec, n_rec, t0, c->acc, img, img_nonempty, c->bloom_mode,
		                     c->bloom_mode ? c->d_newlist : 0, c->d_counters, c->st);
		const double ms = tm.stop();
		c->st_cur.ms_insert += ms; c->st_cur.ms_dominant_kernel += ms; c->st_cur.n_dominant_launches += 1;
	}
	HIPCK(hipMemcpyAsync(h_cnt, c->d_counters, sizeof(h_cnt), hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	const u64 n_new = h_cnt[YKC_NEW];
	c->acc_count += n_new;
	HIPCK(hipMemsetAsync(c->d_counters + YKC_NEW, 0, 8, c->st));
	if (c->bloom_mode && bloom_phases(c, n_new)) return -1;
	return lastput_phase(c, n_rec, t0, batch_lo, batch_hi, img_nonempty);
}

/* give a never-written bloom array its zeros (paths that read-modify-write it in place) */
static int bloom_materialise(yakamd_ctx *c)
{
	if (c->d_bf && c->bf_virgin) {
		HIPCK(hipMemsetAsync(c->d_bf, 0, c->bf_words * 4, c->st));
		c->bf_virgin = false;
	}
	return 0;
}

/* ---- fast path bookkeeping ---- */
static int consume_records(yakamd_ctx *c, int64_t n_rec, u64 t0, u64 batch_lo, u64 batch_hi, const u64 *d_bstart = 0, int hash_only = 0, int ytag = 0);

/* leave the fast path: push every kept batch through the accumulator path, in stream order */
static int fast_abandon(yakamd_ctx *c)
{
	for (auto &k : c->kept) if (k.fmt) return fail("a batch was fed out of stream order after tagged 8-byte batches were kept: feed in order, or set YAKAMD_REC8=0");
	c->fast = false;
	c->retain_broken = true; retained_drop(c);                 /* the pass leaves the path whose records can be kept */
	Rec *keep = c->d_rec;
	int r = 0;
	for (auto &k : c->kept) {
		c->d_rec = k.d_rec;
		if (!r) r = consume_records(c, (int64_t)k.n, k.t0, k.t0, k.t0 + k.span);
		if (k.owned) dfree(k.d_rec);
	}
	c->d_rec = keep;
	c->kept.clear(); c->kept_bytes = 0;
	return r;
}

static int fast_finish(yakamd_ctx *c, bool last = false);
static int fast_flush_slice(yakamd_ctx *c)
{
	if (fast_finish(c)) return -1;
	++c->n_slices;
	HIPCK(hipMemsetAsync(c->d_lastput, 0, c->P * 8, c->st));      /* the put-calls of the slice are accounted for */
	HIPCK(hipMemsetAsync(c->d_counters, 0, YKC_N * 8, c->st));
	if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] slice of the pass counted early (budget / 2^32-position limit): %llu keys in the table\n", (unsigned long long)c->img_keys_total);
	return 0;
}

/* may this batch (n_pos stream positions starting at time t) stay on the fast path?  If so,
 * allocate its level-1 output buffers */
static int fast_admit(yakamd_ctx *c, u64 t, u64 n_pos, u64 n_cap, Rec **out, Rec *borrowed = 0, int fmt = 0)
{
	if (!c->t_pass0_set) { c->t_pass0 = t; c->t_pass0_set = true; }
	/* what a kept record costs against the budget (a quarter of the free memory): its 16 bytes, or for an 8-byte tagged record 12 -- the count of
	 * the slice holds 24-32 bytes per record at its peak (level-2 records, 8 or 16 bytes when the ranks outgrow their field, + 16 of key / time
	 * output), so that a slice admitted by records x 12 <= free / 4 peaks at two thirds of the free memory.  (Tagged slices used to be cut by the
	 * 2^32-position limit long before memory mattered; a rank of an 8-GPU job feeds 9 G records.)  A borrowed buffer is the caller's memory */
	const u64 cost = borrowed ? 0 : n_cap * (fmt ? 12 : 16);
	/* {hash, position} records hold 32-bit positions relative to the slice; tagged records hold no position at all (their times are ranks inside a
	 * sub-table's stream, checked when the slice is counted), so a slice of them may span more than 2^32 stream positions */
	const bool span_ok = fmt != 0 || t + n_pos - c->t_pass0 < 0xfffffff0ull;
	bool fits = t >= c->t_pass0 && span_ok && c->kept_bytes + cost <= c->fast_budget;
	if (fits && !c->kept.empty()) {
		/* the count of a slice cuts a sub-table into at most 2^13 sub-buckets, each meant to hold what one workgroup's LDS table takes (fast_finish):
		 * a slice stops growing where its sub-tables would outgrow that.  Unsharded that is > 5 G records; an 8-GPU rank (an eighth of the
		 * sub-tables, 9 G records of 600 M reads) reaches it every ~700 M records -- the 2^32-position limit used to cut there by accident */
		u64 kept_n = 0;
		for (auto &k : c->kept) kept_n += k.n;
		const u64 per_sb = (u64)yk_lc2_per_sb(c->bloom_mode);
		if (kept_n + n_cap > (u64)(c->phi - c->plo) * (u64)env_i64("YAKAMD_SLICE_SB", env_i64("YAKAMD_P3_MIN", 13) < 18 ? (int64_t)1 << 18 : 8192) * per_sb) fits = false;
	}
	if (!c->kept.empty() && c->kept.back().fmt != fmt) fits = false;   /* tagged records carry ranks, Rec records stream positions: never in one slice */
	/* (a chunk that continues a sequence cut by the caller starts k - 1 positions before the end of the previous one, yak_api.cpp take_piece:
	 * its first k - 1 positions complete no k-mer, so it is still "behind" everything kept) */
	if (!fits && !c->kept.empty() && t + (u64)(c->k - 1) >= c->t_end && (fmt != 0 || n_pos < 0xfffffff0ull) && cost <= c->fast_budget && !c->acc.s) {
		/* the kept batches are a complete prefix of the stream: count them now, exactly as if the pass
		 * ended here (table, filter and counts carry over; the next slice meets them as existing state --
		 * what the reference does chunk after chunk), and start a new slice with times relative to t */
		if (fast_flush_slice(c)) return -1;
		c->t_pass0 = t;
		fits = true;
	}
	if (!fits) { if (fast_abandon(c)) return -1; return 0; }
	yakamd_ctx::Kept k;
	k.d_rec = borrowed; k.n = 0; k.t0 = t; k.span = n_pos; k.owned = !borrowed; k.fmt = fmt;
	if (!borrowed && dmalloc(&k.d_rec, (size_t)(fmt ? (n_cap + 1) / 2 : n_cap))) return -1;
	c->kept_bytes += cost;
	c->kept.push_back(k);
	*out = k.d_rec;
	return 0;
}

static int fast_keep(yakamd_ctx *c, u64 n_rec)
{
	yakamd_ctx::Kept &k = c->kept.back();
	const size_t NB = (size_t)1 << c->nb_bits;
	k.bstart.resize(NB + 1);
	HIPCK(hipMemcpyAsync(k.bstart.data(), c->d_bstart, (NB + 1) * 8, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	k.n = n_rec;
	c->st_cur.n_instances += (int64_t)n_rec;
	return 0;
}

static int feed_image(yak_ch_t *h, const void *d_bases, const u32 *d_valid, int64_t n_bytes, uint64_t t0);
extern "C" int yakamd_feed_bases_dev(yak_ch_t *h, const void *d_bases, int64_t n_bytes, uint64_t t0) { return feed_image(h, d_bases, 0, n_bytes, t0); }
/* the same stream at 0.375 bytes per base: 2-bit codes, 16 bases per 32-bit word (base j at bits 2 (j % 16)), and a validity bit
 * per base (bit j % 32 of word j / 32; 0 = N, any other non-ACGT byte or a record separator).  Position j of the stream is base j */
extern "C" int yakamd_feed_packed_dev(yak_ch_t *h, const void *d_codes, const void *d_valid, int64_t n_bases, uint64_t t0)
{
	if (!d_valid) return fail("packed feed without a validity mask");
	if (((uintptr_t)d_valid & 3) != 0) return fail("validity mask must be 4-byte aligned");
	return feed_image(h, d_codes, (const u32*)d_valid, n_bases, t0);
}
extern "C" int yakamd_pack_bases_dev(const void *d_ascii, int64_t n, void *d_codes, void *d_valid, void *stream)
{
	yk_launch_pack((const uint8_t*)d_ascii, n, (u32*)d_codes, (u32*)d_valid, (hipStream_t)stream);
	return hipGetLastError() == hipSuccess ? 0 : fail("pack kernel launch failed");
}
static int feed_image(yak_ch_t *h, const void *d_bases, const u32 *d_valid, int64_t n_bytes, uint64_t t0)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass) return fail("feed outside a pass");
	if (c->k >= 64 || c->k < 1) return fail("k must be in [1, 63]");
	if (((uintptr_t)d_bases & 15) != 0) return fail("device base image must be 16-byte aligned");
	HIPCK(hipSetDevice(c->dev));
	int64_t batch = env_i64("YAKAMD_BATCH", (int64_t)1 << 31);
	batch = std::min<int64_t>((int64_t)1 << 31, std::max<int64_t>(4096, batch & ~(int64_t)4095));   /* run starts are 31-bit (tagged records: bit 31 is the toggle) */
	int hash_only = !c->create_new;                        /* counting existing keys needs no stream positions */
	int ytag = 0;
	if (hash_only && c->k < 32 && c->nb_bits <= 10 && env_i64("YAKAMD_YTAG", 1) != 0) {   /* the key-owning count will run: its range test is prepared by the extraction */
		int rl, rb; u32 km;
		if (count_own_plan(c, &rl, &rb, &km) == 0 && rb >= 0) { ytag = 1; hash_only = 3; }
	}
	const int64_t bmax = std::min(batch, (n_bytes + 4095) & ~(int64_t)4095);
	if ((!c->fast && rec_reserve(c, bmax)) || part_reserve(c, yk_xpart_blocks(bmax))) return -1;   /* the fast path keeps every batch in a buffer of its own */
	for (int64_t pos = 0; pos < n_bytes; pos += batch) {
		const int64_t end = std::min(n_bytes, pos + batch);
		u64 n_rec = 0;
		Rec *out = c->d_rec;
		if (c->fast) {                                   /* the batch stays resident until pass_end */
			/* tagged 8-byte records (half the partition traffic) when the hash leaves room for the tag: (hash >> pre) < 2^52 */
			const int fmt = c->k < 32 && 2 * c->k - c->pre <= 64 - YK_R8_TAG_BITS && c->nb_bits == c->pre && c->nb_bits <= 10 && !c->or_mode && env_i64("YAKAMD_REC8", 1) != 0;
			if (fast_admit(c, t0 + (u64)pos, (u64)(end - pos), (u64)(end - pos), &out, 0, fmt)) return -1;
			if (!c->fast) { if (rec_reserve(c, bmax)) return -1; out = c->d_rec; }   /* the pass has just left the fast path */
		}
		{
			EvTimer tm(c->st);
			yk_launch_xpart((const uint8_t*)d_bases, pos, end, pos, c->k, c->pre, c->plo, c->phi, c->nb_bits,
			                c->d_rows, c->d_partial, c->d_bstart, out, (c->fast && c->kept.back().fmt) ? 2 : hash_only, c->st, d_valid);
			HIPCK(hipMemcpyAsync(&n_rec, c->d_bstart + ((size_t)1 << c->nb_bits), 8, hipMemcpyDeviceToHost, c->st));
			c->st_cur.ms_extract += tm.stop();
		}
		if (c->fast) { if (fast_keep(c, n_rec)) return -1; if (t0 + (u64)end > c->t_end) c->t_end = t0 + (u64)end; continue; }
		if (consume_records(c, (int64_t)n_rec, t0 + (u64)pos, t0 + (u64)pos, t0 + (u64)end, c->d_bstart, hash_only ? 1 : 0, ytag)) return -1;
	}
	return 0;
}

extern "C" int yakamd_feed_bases_host(yak_ch_t *h, const void *h_bases, int64_t n_bytes, uint64_t t0)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass) return fail("feed outside a pass");
	HIPCK(hipSetDevice(c->dev));
	if (n_bytes > c->stage_cap) {
		dfree(c->d_stage);
		c->stage_cap = n_bytes + (n_bytes >> 3) + 4096;
		if (dmalloc(&c->d_stage, (size_t)c->stage_cap)) { c->stage_cap = 0; return -1; }
	}
	HIPCK(hipMemcpyAsync(c->d_stage, h_bases, (size_t)n_bytes, hipMemcpyHostToDevice, c->st));
	return yakamd_feed_bases_dev(h, c->d_stage, n_bytes, t0);
}

/* the packed image made on the host (yakamd_pack_bases_host: the code words, then -- 16-byte aligned -- the validity words): one copy of
 * 0.375 bytes per base over the bus, then the packed feed */
extern "C" int64_t yakamd_packed_bytes(int64_t n_bases) { const int64_t nw = (n_bases + 31) / 32; return ((nw * 8 + 15) & ~(int64_t)15) + nw * 4; }
extern "C" int yakamd_feed_packed_host(yak_ch_t *h, const void *h_packed, int64_t n_bases, uint64_t t0)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass) return fail("feed outside a pass");
	if (n_bases <= 0) return 0;
	HIPCK(hipSetDevice(c->dev));
	const int64_t need = yakamd_packed_bytes(n_bases), valid_at = need - (n_bases + 31) / 32 * 4;
	if (need > c->stage_cap) {
		dfree(c->d_stage);
		c->stage_cap = need + (need >> 3) + 4096;
		if (dmalloc(&c->d_stage, (size_t)c->stage_cap)) { c->stage_cap = 0; return -1; }
	}
	HIPCK(hipMemcpyAsync(c->d_stage, h_packed, (size_t)need, hipMemcpyHostToDevice, c->st));
	return yakamd_feed_packed_dev(h, c->d_stage, c->d_stage + valid_at, n_bases, t0);
}

/* packed pieces of the stream, each a whole number of 32-position words (its code words and its validity words apart), laid one behind the
 * other on the device -- two copies per piece -- and fed as one image: what yak_count() does with the segments its parser threads packed */
extern "C" int yakamd_feed_packed_pieces_host(yak_ch_t *h, int n_pieces, const void *const *codes, const void *const *valid, const int64_t *n_words, uint64_t t0)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass) return fail("feed outside a pass");
	int64_t nw = 0;
	for (int i = 0; i < n_pieces; ++i) { if (n_words[i] < 0) return fail("packed piece of negative length"); nw += n_words[i]; }
	if (nw == 0) return 0;
	HIPCK(hipSetDevice(c->dev));
	const int64_t n_bases = nw * 32, need = yakamd_packed_bytes(n_bases), valid_at = need - nw * 4;
	if (need > c->stage_cap) {
		dfree(c->d_stage);
		c->stage_cap = need + (need >> 3) + 4096;
		if (dmalloc(&c->d_stage, (size_t)c->stage_cap)) { c->stage_cap = 0; return -1; }
	}
	int64_t w = 0;
	for (int i = 0; i < n_pieces; ++i) {
		if (n_words[i] == 0) continue;
		HIPCK(hipMemcpyAsync(c->d_stage + w * 8, codes[i], (size_t)n_words[i] * 8, hipMemcpyHostToDevice, c->st));
		HIPCK(hipMemcpyAsync(c->d_stage + valid_at + w * 4, valid[i], (size_t)n_words[i] * 4, hipMemcpyHostToDevice, c->st));
		w += n_words[i];
	}
	return yakamd_feed_packed_dev(h, c->d_stage, c->d_stage + valid_at, n_bases, t0);
}

extern "C" int yakamd_feed_hashed_dev(yak_ch_t *h, const void *d_hash, const void *d_t, int64_t n, uint64_t t0, uint64_t t_span)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass) return fail("feed outside a pass");
	HIPCK(hipSetDevice(c->dev));
	/* group the records by sub-table prefix first (same locality as the extraction path) */
	if (n <= 0) return 0;
	if (rec_reserve(c, n) || part_reserve(c, yk_rpart_blocks(n))) return -1;
	u64 n_rec = 0;
	Rec *out = c->d_rec;
	if (c->fast && fast_admit(c, t0, t_span, (u64)n, &out)) return -1;
	yk_launch_rpart((const u64*)d_hash, (const u32*)d_t, n, c->pre, c->plo, c->phi, c->nb_bits,
	                c->d_rows, c->d_partial, c->d_bstart, out, c->st);
	HIPCK(hipMemcpyAsync(&n_rec, c->d_bstart + ((size_t)1 << c->nb_bits), 8, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	if (c->fast) { if (t0 + t_span > c->t_end) c->t_end = t0 + t_span; return fast_keep(c, n_rec); }
	return consume_records(c, (int64_t)n_rec, t0, t0, t0 + t_span, c->d_bstart);
}

static int64_t partition_dev(int k, int pre, const void *d_bases, int64_t n_bytes, void *d_out, uint64_t *h_bstart, int hash_only)
{
	if (k < 1 || k > 63 || pre < 3 || pre > 13) { fail("partition: unsupported k / pre"); return -1; }
	if (((uintptr_t)d_bases & 15) != 0 || n_bytes >= ((int64_t)1 << 32)) { fail("partition: base image must be 16-byte aligned and < 4 GiB"); return -1; }
	const size_t NB = (size_t)1 << pre;
	const int n_blk = yk_xpart_blocks(n_bytes);
	u32 *d_rows = 0; u64 *d_partial = 0, *d_bstart = 0;
	if (dmalloc(&d_rows, NB * (size_t)n_blk) || dmalloc(&d_partial, NB * yk_part_groups()) || dmalloc(&d_bstart, NB + 1)) return -1;
	/* on a stream of its own, not the null stream (which waits for, and holds up, every other stream of the device): a multi-GPU job partitions the
	 * chunks of its next round while the owners' streams take in the round before */
	int dev = 0;
	(void)hipGetDevice(&dev);
	hipStream_t ps = stream_get(dev);
	yk_launch_xpart((const uint8_t*)d_bases, 0, n_bytes, 0, k, pre, 0, 1 << pre, pre, d_rows, d_partial, d_bstart, (Rec*)d_out, hash_only, ps);
	hipError_t e = hipMemcpyAsync(h_bstart, d_bstart, (NB + 1) * 8, hipMemcpyDeviceToHost, ps);
	if (e == hipSuccess) e = hipStreamSynchronize(ps);
	stream_put(dev, ps);
	dfree(d_rows); dfree(d_partial); dfree(d_bstart);
	if (e != hipSuccess) { fail("partition: %s", hipGetErrorString(e)); return -1; }
	return (int64_t)h_bstart[NB];
}

extern "C" int64_t yakamd_partition_dev(int k, int pre, const void *d_bases, int64_t n_bytes, void *d_rec_out, uint64_t *h_bstart)
{
	return partition_dev(k, pre, d_bases, n_bytes, d_rec_out, h_bstart, 0);
}

/* the same partition as 8-byte tagged records (yk_device.h YK_R8_*: half the exchange payload of a sharded pass); for
 * yakamd_feed_partitioned_tagged_dev on the owner.  yakamd_tagged_ok says whether k / pre allow the format */
extern "C" int yakamd_tagged_ok(int k, int pre) { return k >= 1 && k < 32 && 2 * k - pre <= 64 - YK_R8_TAG_BITS && pre >= 3 && pre <= 10 && env_i64("YAKAMD_REC8", 1) != 0; }
extern "C" int64_t yakamd_partition_tagged_dev(int k, int pre, const void *d_bases, int64_t n_bytes, void *d_rec8_out, uint64_t *h_bstart)
{
	if (!yakamd_tagged_ok(k, pre)) { fail("partition: tagged records need k < 32, 2k - pre <= 52, pre <= 10"); return -1; }
	if (n_bytes > ((int64_t)1 << 31)) { fail("partition: at most 2^31 stream positions per call with tagged records (31-bit run starts)"); return -1; }
	return partition_dev(k, pre, d_bases, n_bytes, d_rec8_out, h_bstart, 2);
}

extern "C" int64_t yakamd_partition_hashes_dev(int k, int pre, const void *d_bases, int64_t n_bytes, void *d_hash_out, uint64_t *h_bstart)
{
	return partition_dev(k, pre, d_bases, n_bytes, d_hash_out, h_bstart, 1);
}

static int feed_partitioned(yak_ch_t *h, const void *d_rec, int64_t n, const uint64_t *h_bstart, uint64_t t0, uint64_t t_span, bool borrow, int fmt = 0)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass) return fail("feed outside a pass");
	if (c->nb_bits != c->pre) return fail("feed_partitioned needs pre <= 13");
	HIPCK(hipSetDevice(c->dev));
	if (n <= 0) return 0;
	const size_t NB = (size_t)1 << c->nb_bits;
	if (c->fast) {
		Rec *out = 0;
		if (fmt && !(yakamd_tagged_ok(c->k, c->pre) && !c->or_mode)) return fail("tagged records do not fit this table (k, pre)");
		if (fast_admit(c, t0, t_span, (u64)n, &out, borrow ? (Rec*)d_rec : 0, fmt)) return -1;
		if (c->fast) {                                       /* admitted: already grouped by prefix; keep a private copy unless lent */
			if (!borrow) HIPCK(hipMemcpyAsync(out, d_rec, (size_t)n * (fmt ? 8 : sizeof(Rec)), hipMemcpyDeviceToDevice, c->st));
			yakamd_ctx::Kept &k = c->kept.back();
			k.bstart.assign(h_bstart, h_bstart + NB + 1);
			k.n = (u64)n;
			c->st_cur.n_instances += n;
			if (t0 + t_span > c->t_end) c->t_end = t0 + t_span;
			return 0;
		}
	}
	if (fmt) return fail("tagged records need the exclusive-ownership path (the pass left it: input beyond the device budget, or YAKAMD_FAST=0)");
	Rec *keep = c->d_rec;                                    /* general path / pass 2: consume in place */
	c->d_rec = (Rec*)d_rec;
	const int r = consume_records(c, n, t0, t0, t0 + t_span);
	c->d_rec = keep;
	return r;
}

extern "C" int yakamd_feed_partitioned_dev(yak_ch_t *h, const void *d_rec, int64_t n, const uint64_t *h_bstart, uint64_t t0, uint64_t t_span)
{
	return feed_partitioned(h, d_rec, n, h_bstart, t0, t_span, false);
}

extern "C" int yakamd_feed_partitioned_lent_dev(yak_ch_t *h, const void *d_rec, int64_t n, const uint64_t *h_bstart, uint64_t t0, uint64_t t_span)
{
	return feed_partitioned(h, d_rec, n, h_bstart, t0, t_span, true);
}

/* 1 while the open pass of h runs on the exclusive-ownership path (the only one that takes tagged records) */
extern "C" int yakamd_pass_fast(yak_ch_t *h) { yakamd_ctx *c = ctx_of(h); return c && c->in_pass && c->fast && !c->or_mode; }

extern "C" int yakamd_feed_partitioned_tagged_dev(yak_ch_t *h, const void *d_rec8, int64_t n, const uint64_t *h_bstart, uint64_t t0, uint64_t t_span, int lent)
{
	return feed_partitioned(h, d_rec8, n, h_bstart, t0, t_span, lent != 0, 1);
}

extern "C" int yakamd_count_partitioned_dev(yak_ch_t *h, const void *d_hash_u64, int64_t n, const uint64_t *h_bstart)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass || c->create_new) return fail("count_partitioned needs an open create_new = 0 pass");
	if (c->nb_bits != c->pre) return fail("count_partitioned needs pre <= 13");
	HIPCK(hipSetDevice(c->dev));
	if (n <= 0) return 0;
	const size_t NB = (size_t)1 << c->nb_bits;
	if (part_reserve(c, 1)) return -1;
	HIPCK(hipMemcpyAsync(c->d_bstart, h_bstart, (NB + 1) * 8, hipMemcpyHostToDevice, c->st));
	Rec *keep = c->d_rec;
	c->d_rec = (Rec*)d_hash_u64;
	const int r = consume_records(c, n, 0, 0, 0, c->d_bstart, 1);
	c->d_rec = keep;
	HIPCK(hipStreamSynchronize(c->st));                       /* h_bstart may be a temporary of the caller */
	return r;
}

extern "C" int yakamd_count_hashes_dev(yak_ch_t *h, const void *d_hash_u64, int64_t n)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass || c->create_new) return fail("count_hashes needs an open create_new = 0 pass");
	HIPCK(hipSetDevice(c->dev));
	if (n <= 0) return 0;
	c->delta_dirty = true;
	EvTimer tm(c->st);
	yk_launch_img_count_h((const u64*)d_hash_u64, n, img_view(c), c->st);
	const double ms = tm.stop();
	c->st_cur.n_instances += n;
	c->st_cur.ms_insert += ms; c->st_cur.ms_dominant_kernel += ms; c->st_cur.n_dominant_launches += 1;
	return 0;
}

/* ---- the count pass over the input of the pass before (reference main.c:53-57: both passes read the same file) ---- */
extern "C" int yakamd_retain_input(yak_ch_t *h, int on)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c) return fail("not an engine table");
	if (c->in_pass) return fail("yakamd_retain_input inside a pass");
	HIPCK(hipSetDevice(c->dev));
	c->retain_on = on != 0;
	if (!on) retained_drop(c);
	return 0;
}

extern "C" int64_t yakamd_retained_instances(yak_ch_t *h)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || c->retain_broken) return 0;
	u64 n = c->ret2.valid ? c->ret2.n_total : 0;
	for (auto &r : c->retained) n += r.n;
	return (int64_t)n;
}

/* inside an open create_new = 0 pass: count every instance of the retained records (k_img_count_own reads the tagged level-1 records
 * as they are).  0 = counted, the records are released; 1 = nothing usable was retained (the caller feeds the input again); -1 = error */
extern "C" int yakamd_count_retained(yak_ch_t *h)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass || c->create_new) return fail("yakamd_count_retained needs an open create_new = 0 pass");
	HIPCK(hipSetDevice(c->dev));
	if (c->ret2.valid && !c->retain_broken) {
		u32 *d_kcnt = 0;
		if (dmalloc(&d_kcnt, c->ret2.n_keys)) return -1;
		HIPCK(hipMemsetAsync(c->d_nmissing, 0, 4, c->st));          /* (a spare word of the context: "some instance went through the pending counts") */
		EvTimer tm(c->st);
		yk_launch_cnt2(c->ret2.fp, c->ret2.d_sbstart, c->ret2.d_r2, c->ret2.d_koff, c->ret2.d_kkc, c->ret2.d_segbase, d_kcnt, img_view(c), c->ret2.n_keys, c->d_nmissing, c->st);
		u32 used = 1;
		(void)hipMemcpyAsync(&used, c->d_nmissing, 4, hipMemcpyDeviceToHost, c->st);
		const double ms = tm.stop();
		if (used) c->delta_dirty = true;
		dfree(d_kcnt);
		const bool bad = hipGetLastError() != hipSuccess;
		c->st_cur.n_instances += (int64_t)c->ret2.n_total;
		c->st_cur.ms_insert += ms; c->st_cur.ms_dominant_kernel += ms; c->st_cur.n_dominant_launches += 1;
		retained_drop(c);
		return bad ? fail("the count over the retained sub-bucket records failed") : 0;
	}
	if (c->retained.empty() || c->retain_broken) { retained_drop(c); return 1; }
	{
		int rl, rb; u32 km;
		if (count_own_plan(c, &rl, &rb, &km) != 0) { retained_drop(c); return 1; }   /* tables beyond the key-owning count kernel: the general count path wants plain hashes */
	}
	const size_t NB = (size_t)1 << c->nb_bits;
	if (part_reserve(c, 1)) return -1;
	Rec *keep = c->d_rec;
	int r = 0;
	for (auto &b : c->retained) {
		if (r) break;
		if (hipMemcpyAsync(c->d_bstart, b.bstart.data(), (NB + 1) * 8, hipMemcpyHostToDevice, c->st) != hipSuccess) { r = fail("memcpy"); break; }
		c->d_rec = (Rec*)b.d_rec;
		r = consume_records(c, (int64_t)b.n, 0, 0, 0, c->d_bstart, 1, 2);
		if (!r && hipStreamSynchronize(c->st) != hipSuccess) r = fail("count of the retained records failed");
	}
	c->d_rec = keep;
	retained_drop(c);
	return r ? -1 : 0;
}

/* yak_count(): the file whose records are retained (device, inode, size, mtime) and its sequence count, for the log line */
void yk_ctx_set_source(yakamd_ctx *c, const uint64_t id[4], int64_t n_seq) { memcpy(c->src_id, id, 32); c->src_id[4] = (u64)n_seq; c->src_set = true; }
/* (The retained key lists name the keys pass 1 put into the table.  Nothing can add a key between the two passes without a create_new pass -- which
 * drops what is retained (yakamd_pass_begin) -- and yak_ch_inc / setcnt / clear only touch counts; keys removed in between (shrink, subtract) are
 * simply not found by the count) */
bool yk_ctx_same_source(yakamd_ctx *c, const uint64_t id[4], int64_t *n_seq)
{
	if (!c->src_set || (c->retained.empty() && !c->ret2.valid) || c->retain_broken || memcmp(c->src_id, id, 32) != 0) return false;
	*n_seq = (int64_t)c->src_id[4];
	return true;
}

/* ---- lookup-only path (yak qv) ---- */
extern "C" int yakamd_lookup_dev(yak_ch_t *h, const void *d_bases, int64_t n_bytes, void *d_out_u16)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c) return fail("not an engine table");
	if (c->in_pass) return fail("lookup during an open pass");
	if (c->k < 1 || c->k >= 32) return fail("lookup: k must be below 32 (reference qv.c:44)");
	if (((uintptr_t)d_bases & 15) != 0) return fail("device base image must be 16-byte aligned");
	HIPCK(hipSetDevice(c->dev));
	yk_launch_lookup((const uint8_t*)d_bases, n_bytes, c->k, img_view(c), (unsigned short*)d_out_u16, c->st);
	HIPCK(hipStreamSynchronize(c->st));
	return 0;
}

extern "C" int yakamd_qv_reduce_dev(yak_ch_t *h, const void *d_t_u16, const uint64_t *d_seq_off, const uint32_t *d_seq_len, int64_t n_seq,
                                    int min_len, double min_frac, uint32_t *d_tot, uint32_t *d_non0, uint64_t *d_hist1024)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c) return fail("not an engine table");
	HIPCK(hipSetDevice(c->dev));
	yk_launch_qv_reduce((const unsigned short*)d_t_u16, (const u64*)d_seq_off, d_seq_len, n_seq, min_len, min_frac, d_tot, d_non0, (u64*)d_hist1024, c->st);
	HIPCK(hipStreamSynchronize(c->st));
	return 0;
}

extern "C" int64_t yakamd_extract_dev(int k, const void *d_bases, int64_t n_bytes, void *d_hash, void *d_t,
                                      int pre, int plo, int phi, void *stream)
{
	if (k >= 64 || k < 1) { fail("extract: unsupported k"); return -1; }
	u64 *d_cur = 0, n = 0;
	hipStream_t st = (hipStream_t)stream;
	if (hipMalloc((void**)&d_cur, 8) != hipSuccess) { fail("hipMalloc"); return -1; }
	hipMemsetAsync(d_cur, 0, 8, st);
	yk_launch_extract((const uint8_t*)d_bases, 0, n_bytes, 0, k, pre, plo, phi, (u64*)d_hash, (u32*)d_t, d_cur, st);
	hipMemcpyAsync(&n, d_cur, 8, hipMemcpyDeviceToHost, st);
	hipStreamSynchronize(st);
	hipFree(d_cur);
	return (int64_t)n;
}

/* pass_end of the fast path: level-2 partition -> exclusive LDS counting (+ bloom gate) ->
 * per sub-table sort by insertion time -> exact layout replay */
static int fast_finish(yakamd_ctx *c, bool last)
{
	const int P = c->P;
	u64 n_total = 0;
	for (auto &k : c->kept) n_total += k.n;
	const int fmt_in = c->kept.empty() ? 0 : c->kept.front().fmt;
	u64 sort_tmax = c->t_end;
	FastParams fp;
	fp.pre = c->pre; fp.k = c->k; fp.bloom_mode = c->bloom_mode; fp.nb = c->nb; fp.n_hash = c->n_hash;
	fp.img_nonempty = c->img_keys_total > 0; fp.plo = c->plo; fp.phi = c->phi; fp.t_pass0 = c->t_pass0;
	fp.dbg = (int)env_i64("YAKAMD_DBG", 0);
	fp.or_mode = c->or_mode;
	fp.bf_nowb = 0;
	/* YAKAMD_VERBOSE > 1: wall-clock laps of the stages (each behind a stream synchronise: allocation stalls show up where they happen) */
	const bool laps = env_i64("YAKAMD_VERBOSE", 0) > 1;
	double lap_t = now_ms();
	auto lap = [&](const char *what) {
		if (!laps) return;
		(void)hipStreamSynchronize(c->st);
		const double t = now_ms();
		size_t fr = 0, tt_ = 0;
		(void)hipMemGetInfo(&fr, &tt_);
		fprintf(stderr, "[yak_amd] slice stage %-28s %9.2f ms   (device memory in use %.1f GB, of it idle in the pool %.1f GB)\n", what, t - lap_t, (double)(tt_ - fr) / 1e9, (double)yk_pool_cached_bytes() / 1e9);
		lap_t = now_ms();
	};
	/* mean sub-bucket <= ~600 instances: even if all are distinct the 1024-slot LDS table holds them.  With a filter the input is reads with
	 * coverage (a filtered count of all-distinct k-mers keeps nothing): three times as many instances per sub-bucket still leave the distinct
	 * k-mers far below the table's 624 (30 x coverage: ~100 distinct per 560 instances), and 2048 sub-buckets per sub-table are what the
	 * level-2 scatter takes in one sweep -- 30 M reads: 13 -> 11 bits, its partition 124 -> ~40 ms.  A sub-bucket that does overflow goes to
	 * the tiers behind k_lc2, as always */
	const u64 per_sb = (u64)yk_lc2_per_sb(c->bloom_mode);         /* 1800 with a filter; without one 1200 (k_lc2's 2048-slot table) or 600 (the older tiers) */
	/* the largest stream of one sub-table in this slice: decides whether the level-2 records take 8 bytes (below), which the plan needs */
	u64 np_max = 0;
	for (int p = 0; p < P; ++p) { u64 np = 0; for (auto &k : c->kept) np += k.bstart[p + 1] - k.bstart[p]; np_max = std::max(np_max, np); }
	/* Known before the partition: will every record of this pass stay on the device for the count pass over the same input (keep2 below) while
	 * the filter is still untouched?  Then nobody reads the filter's bits before the records can rebuild them (bf_nowb), k_lc2 needs no stage of
	 * the filter in LDS, and a sub-bucket may own 256 blocks instead of 128: half as many, twice as large */
	bool nowb_plan = false;
	{
		size_t fr = 0, tot = 0;
		const int64_t cap_gb = env_i64("YAKAMD_RETAIN_GB", -1);
		u64 budget = cap_gb >= 0 ? (u64)cap_gb << 30 : 0;
		if (cap_gb < 0 && hipMemGetInfo(&fr, &tot) == hipSuccess) budget = tot / 8;
		nowb_plan = c->bloom_mode && c->bf_virgin && last && c->n_slices == 0 && c->img_keys_total == 0 && c->retain_on && !c->retain_broken && fmt_in == 1 &&
		            !c->or_mode && c->retained.empty() && c->retained_bytes + n_total * 8 <= budget && c->n_hash <= 32 &&
		            env_i64("YAKAMD_RETAIN2", 1) != 0 && env_i64("YAKAMD_BF_DEFER", 1) != 0 && env_i64("YAKAMD_LC2", 1) != 0 && env_i64("YAKAMD_LC2_NOSTAGE", 1) != 0;
		for (auto &k : c->kept) nowb_plan = nowb_plan && k.owned;
	}
	/* one sweep of the level-2 scatter takes up to 2^11 sub-buckets (2^13 in sweeps over the chunk); beyond that -- the share of an N-GPU job's rank:
	 * 128 sub-tables of 69 M instances each -- the partition takes two sweeps (p3): the high bits first into {hash, rank} records, then 2^p3_low
	 * sub-buckets inside every group.  2^18 sub-buckets per sub-table bound the per-sub-bucket arrays */
	const int p3_min = (int)env_i64("YAKAMD_P3_MIN", 13), p3_low = (int)std::min<int64_t>(11, std::max<int64_t>(1, env_i64("YAKAMD_P3_LOW", 11)));
	int s2 = 0, s2a = 0, s2b = 0;
	bool three = false;
	for (int attempt = 0; attempt < 2; ++attempt) {
		const int lb_max = nowb_plan ? 8 : 7;                      /* log2 blocks a sub-bucket may own: k_lc2 stages at most 128; without a stage its table gives every block >= 4 home slots */
		s2 = n_total ? ceil_log2_u64((n_total / (u64)(c->phi - c->plo) + per_sb - 1) / per_sb) : 0;
		if (c->bloom_mode && s2 < c->nb - 9 - lb_max && n_total / (u64)(c->phi - c->plo) > 600) s2 = c->nb - 9 - lb_max;
		s2 = (int)env_i64("YAKAMD_S2_BITS", s2);
		if (s2 > 18) s2 = 18;
		if (s2 < 0) s2 = 0;
		if (c->bloom_mode) {
			if (s2 > c->nb - 9) s2 = c->nb - 9;              /* a sub-bucket owns whole 512-bit blocks ... */
			if (s2 < c->nb - 9 - 20) s2 = c->nb - 9 - 20;    /* ... and at most 2^20 of them (sort-key packing) */
		}
		three = s2 > p3_min;
		if (!three && s2 > 13) s2 = 13;
		s2a = three ? std::min(11, std::max(std::min(4, s2 - 1), s2 - p3_low)) : 0; s2b = s2 - s2a;   /* the first sweep keeps >= 16 groups: the write-combining scatter wants >= 4 bits */
		/* the plan without a filter stage falls when the pass cannot keep its level-2 records after all (16-byte records are not kept; ranges beyond 256 blocks
		 * go to the tier behind k_lc2).  s2 was sized for 256 blocks per sub-bucket then: it is chosen again for the 128 a stage holds (the
		 * first choice stayed, k_lc2 refused the 256-block ranges and the whole pass went through the global-scratch tier) */
		if (nowb_plan && c->bloom_mode && (YK_R8_TAG_BITS + s2 >= 64 || np_max >= (1ull << (YK_R8_TAG_BITS + s2)) || env_i64("YAKAMD_REC8_OUT", 1) == 0 || c->nb - 9 - s2 > 8 || c->nb - 9 - s2 < 0)) { nowb_plan = false; continue; }
		break;
	}
	if (three && s2b > 13) return fail("level-2 partition: 2^%d sub-buckets per sub-table cannot be split into two sweeps", s2);
	fp.s2_bits = s2; fp.s2_tot = s2;
	fp.sw = c->bloom_mode ? c->nb - 9 : s2; fp.ssh = fp.sw - s2;
	fp.bf_virgin = 0;
	fp.rec8_in = fmt_in; fp.tb = YK_R8_TAG_BITS + s2;
	fp.rec8_out = 0;                                         /* set below, once the largest sub-table stream is known */
	if (c->bloom_mode) {
		const bool lc2_runs = env_i64("YAKAMD_LC2", 1) != 0 && c->n_hash <= 32;   /* (with the block range below: yk_lc2_ok) -- the tier behind k_lc2 works on the filter in memory and needs real zeros */
		if (c->bf_virgin && lc2_runs && c->nb - 9 - s2 <= (nowb_plan ? 8 : 7)) fp.bf_virgin = 1;   /* LDS-staged ranges: skip the read, write every block of the shard (yakamd_set_shard refuses to move the shard afterwards) */
		else if (bloom_materialise(c)) return -1;
		c->bf_virgin = false;
	}
	const int s2_first = three ? s2a : s2;                       /* bits of the sweep that reads the level-1 records */
	const u64 ch2 = std::max<u64>((u64)env_i64("YAKAMD_CH2", YK_CH2), (u64)32 << s2_first);    /* keep >= 32 records per sub-bucket run */
	/* chunk table: runs of one sub-table's records, grouped by sub-table */
	std::vector<Chunk2> chunks;
	std::vector<u32> chunk_first(P + 1, 0);
	std::vector<u64> bbase(P + 1, 0);
	for (int p = 0; p < P; ++p) {
		chunk_first[p] = (u32)chunks.size();
		u64 np = 0;
		for (auto &k : c->kept) {
			const u64 a = k.bstart[p], b = k.bstart[p + 1];
			for (u64 o = a; o < b; o += ch2) {
				Chunk2 ch;
				ch.rec = fmt_in ? (const Rec*)((const u64*)k.d_rec + o) : k.d_rec + o; ch.spare = 0;
				ch.n = (u32)std::min<u64>(ch2, b - o); ch.bucket = (u32)p;
				ch.tbase = fmt_in ? (u32)np : (u32)(k.t0 - c->t_pass0); ch.pad = (u32)fmt_in;   /* tagged: ranks continue from the earlier batches of this sub-table */
				ch.before = (u32)(o - a); ch.after = (u32)(b - (o + ch.n));
				chunks.push_back(ch);
			}
			np += b - a;
		}
		bbase[p + 1] = bbase[p] + np;
		if (chunks.size() > chunk_first[p]) chunks.back().spare = 1;   /* last chunk of its sub-table */
	}
	chunk_first[P] = (u32)chunks.size();
	const size_t n_sb = (size_t)P << s2, S2F = (size_t)1 << s2_first;
	if (fmt_in) {
		if (np_max >= (1ull << 32)) return fail("more than 2^32 k-mer instances of one sub-table in one slice");
		fp.rec8_out = np_max < (1ull << fp.tb) && env_i64("YAKAMD_REC8_OUT", 1) != 0;   /* the rank must fit below the hash bits; else 16-byte records {hash, rank} */
		fp.t_pass0 = 0;                                       /* times are ranks inside the sub-table's stream of this slice */
		sort_tmax = np_max;
	}

	Chunk2 *d_chunks = 0; u32 *d_cf = 0, *d_rows2 = 0, *d_segcur = 0, *d_ovf2 = 0, *d_ndist = 0; u64 *d_bbase = 0, *d_sbstart = 0, *d_segbase = 0, *d_sba = 0, *d_koff = 0; Rec *d_r2 = 0, *d_ra = 0;
	u64 *kc[2] = { 0, 0 }, *tt[2] = { 0, 0 };
	LcOut lo; lo.kc = 0; lo.T = 0; lo.nsel = 0; lo.lp = 0; lo.nd = 0;
	u64 *d_scr = 0, *d_scroff = 0;
	/* every device buffer of this function is released here, whichever way it is left */
	struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ [&]() {
		dfree(d_chunks); dfree(d_cf); dfree(d_rows2); dfree(d_segcur); dfree(d_ovf2); dfree(d_ndist); dfree(d_bbase); dfree(d_sbstart);
		dfree(d_segbase); dfree(d_r2); dfree(kc[0]); dfree(kc[1]); dfree(tt[0]); dfree(tt[1]); dfree(lo.kc); dfree(lo.T); dfree(lo.nsel); dfree(lo.lp); dfree(lo.nd);
		dfree(d_scr); dfree(d_scroff); dfree(d_sba); dfree(d_ra); dfree(d_koff);
	} };
	if (dmalloc(&d_chunks, chunks.size()) || dmalloc(&d_cf, P + 1) || dmalloc(&d_bbase, P + 1) || dmalloc(&d_rows2, chunks.size() * S2F) ||
	    dmalloc(&d_sbstart, n_sb + 1) || dmalloc(&d_segcur, P) || dmalloc(&d_ovf2, n_sb)) return -1;
	if (three ? (dmalloc(&d_ra, n_total) || dmalloc(&d_sba, ((size_t)P << s2a) + 1)) : dmalloc(&d_r2, fp.rec8_out ? (n_total + 1) / 2 : n_total)) return -1;
	HIPCK(hipMemcpyAsync(d_chunks, chunks.data(), chunks.size() * sizeof(Chunk2), hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(d_cf, chunk_first.data(), (P + 1) * 4, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(d_bbase, bbase.data(), (P + 1) * 8, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemsetAsync(c->d_counters + YKC_NOVF2, 0, 8, c->st));
	std::vector<u64> h_sba;
	{
		EvTimer tm(c->st);
		if (!three) yk_launch_part2(d_chunks, (int)chunks.size(), d_cf, d_bbase, fp, P, d_rows2, d_sbstart, d_r2, c->st);
		else {
			/* first sweep: the high s2a bits of the sub-bucket, {hash, rank} records out (the rank no longer fits beside the hash while only
			 * s2a of its bits are implied by the place) */
			FastParams fa = fp;
			fa.s2_bits = s2a; fa.ssh = fp.sw - s2a; fa.rec8_out = 0;
			yk_launch_part2(d_chunks, (int)chunks.size(), d_cf, d_bbase, fa, P, d_rows2, d_sba, d_ra, c->st);
			h_sba.resize(((size_t)P << s2a) + 1);
			HIPCK(hipMemcpyAsync(h_sba.data(), d_sba, h_sba.size() * 8, hipMemcpyDeviceToHost, c->st));
		}
		c->ms_part2 = tm.stop();
		c->st_cur.ms_extract += c->ms_part2; c->st_cur.ms_part2 += c->ms_part2;
	}
	lap("level-2 partition (first sweep)");
	bool keep2 = false;
	{
		/* the level-1 records stay for the count pass over the same input when the caller asked for that and they fit the budget
		 * (an eighth of the device memory unless YAKAMD_RETAIN_GB says otherwise): all of the pass's records, or none */
		size_t fr = 0, tot = 0;
		const int64_t cap_gb = env_i64("YAKAMD_RETAIN_GB", -1);
		u64 budget = cap_gb >= 0 ? (u64)cap_gb << 30 : 0;
		if (cap_gb < 0 && hipMemGetInfo(&fr, &tot) == hipSuccess) budget = tot / 8;
		bool keep = c->retain_on && !c->retain_broken && fmt_in == 1 && c->bloom_mode && !c->or_mode && c->retained_bytes + n_total * 8 <= budget;
		for (auto &k : c->kept) keep = keep && k.owned;
		/* the whole pass in this one slice, into an empty table: the level-2 records and the sub-buckets' key lists serve the count pass better (k_cnt2) */
		keep2 = keep && last && c->n_slices == 0 && c->img_keys_total == 0 && fp.rec8_out && c->retained.empty() && env_i64("YAKAMD_RETAIN2", 1) != 0;
		if (keep2) keep = false;
		if (c->retain_on && !keep && !keep2 && !c->kept.empty()) { c->retain_broken = true; retained_drop(c); }
		for (auto &k : c->kept) {
			if (keep && k.n) { yakamd_ctx::Retained r; r.d_rec = (u64*)k.d_rec; r.n = k.n; r.bstart.swap(k.bstart); c->retained.push_back(std::move(r)); c->retained_bytes += k.n * 8; }
			else if (k.owned) dfree(k.d_rec);
		}
	}
	c->kept.clear(); c->kept_bytes = 0;
	dfree(d_chunks); dfree(d_cf); dfree(d_rows2); dfree(d_bbase);
	if (three) {
		/* second sweep: every group of the first is a bucket of its own (sub-table << s2a | group); its 2^s2b sub-buckets take the group's place in
		 * the final numbering sub-table << s2 | sub-bucket, so the offsets of the first sweep are the bucket bases of the second */
		const size_t PB = (size_t)P << s2a, S2B = (size_t)1 << s2b;
		const u64 chb = std::max<u64>((u64)env_i64("YAKAMD_CH2", YK_CH2), (u64)32 << s2b);
		std::vector<Chunk2> cb;
		std::vector<u32> cfb(PB + 1, 0);
		for (size_t q = 0; q < PB; ++q) {
			cfb[q] = (u32)cb.size();
			for (u64 o = h_sba[q]; o < h_sba[q + 1]; o += chb) {
				Chunk2 ch;
				ch.rec = d_ra + o; ch.spare = 0; ch.n = (u32)std::min<u64>(chb, h_sba[q + 1] - o); ch.bucket = (u32)q; ch.tbase = 0; ch.pad = 0; ch.before = 0; ch.after = 0;
				cb.push_back(ch);
			}
			if (cb.size() > cfb[q]) cb.back().spare = 1;
		}
		cfb[PB] = (u32)cb.size();
		FastParams fb = fp;
		fb.s2_bits = s2b; fb.rec8_in = 0;                          /* routes by the low s2b bits, packs with all of them (s2_tot) */
		if (dmalloc(&d_chunks, cb.size()) || dmalloc(&d_cf, PB + 1) || dmalloc(&d_rows2, cb.size() * S2B) || dmalloc(&d_r2, fp.rec8_out ? (n_total + 1) / 2 : n_total)) return -1;
		HIPCK(hipMemcpyAsync(d_chunks, cb.data(), cb.size() * sizeof(Chunk2), hipMemcpyHostToDevice, c->st));
		HIPCK(hipMemcpyAsync(d_cf, cfb.data(), (PB + 1) * 4, hipMemcpyHostToDevice, c->st));
		EvTimer tm(c->st);
		yk_launch_part2(d_chunks, (int)cb.size(), d_cf, d_sba, fb, (int)PB, d_rows2, d_sbstart, d_r2, c->st);
		const double ms = tm.stop();                                /* (the host vectors above outlive the copies) */
		c->ms_part2 += ms; c->st_cur.ms_extract += ms; c->st_cur.ms_part2 += ms;
		dfree(d_chunks); dfree(d_cf); dfree(d_rows2); dfree(d_ra); dfree(d_sba);
		if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] level-2 partition in two sweeps: 2^%d groups, then 2^%d sub-buckets each (%llu records, %.2f ms)\n", s2a, s2b, (unsigned long long)n_total, c->ms_part2);
	}
	lap("release / second sweep");
	/* the keys a sub-bucket selects are written over the front of its own record range in lo.kc / lo.T */
	/* 8-byte level-2 records that nobody keeps: the keys go over the records themselves (a workgroup has read all of its sub-bucket's records
	 * before it writes its first key, and nobody else reads them) -- 8 bytes per record less at the peak of a slice (a cfg3 rank: 71 GB) */
	const bool kc_inplace = fp.rec8_out && !keep2 && env_i64("YAKAMD_KC_INPLACE", 1) != 0;
	if (kc_inplace) { lo.kc = (u64*)d_r2; d_r2 = 0; }
	else if (dmalloc(&lo.kc, n_total)) return -1;
	if (dmalloc(&lo.T, n_total) || dmalloc(&lo.nsel, n_sb) || dmalloc(&lo.lp, n_sb) || dmalloc(&lo.nd, n_sb) || dmalloc(&d_ndist, P)) return -1;
	const Rec *lc_rec_in = kc_inplace ? (const Rec*)lo.kc : d_r2;
	if (c->plo > 0 || c->phi < P) {                              /* sub-buckets outside the shard are never visited */
		HIPCK(hipMemsetAsync(lo.nsel, 0, n_sb * 4, c->st)); HIPCK(hipMemsetAsync(lo.lp, 0, n_sb * 4, c->st)); HIPCK(hipMemsetAsync(lo.nd, 0, n_sb * 4, c->st));
	}
	HIPCK(hipMemsetAsync(d_ndist, 0, P * 4, c->st));
	HIPCK(hipMemsetAsync(d_segcur, 0, P * 4, c->st));
	u64 h_cnt[YKC_N];
	/* every record of the pass stays on the device (keep2) and the filter has never been written: the 2^bf_shift bits are not written at all --
	 * yak_ch_destroy_bf usually comes next (main.c:55); whatever reads the filter first rebuilds it (bloom_undefer).  nowb_plan said so before the
	 * partition (sub-buckets of up to 256 blocks, no stage in LDS); with the stage (a plan that was off) the bits are kept in LDS and dropped */
	fp.bf_nowb = keep2 && fp.bf_virgin && env_i64("YAKAMD_BF_DEFER", 1) != 0 && env_i64("YAKAMD_LC2", 1) != 0 && c->n_hash <= 32;
	if (nowb_plan && !fp.bf_nowb) {
		/* the two predicates (nowb_plan before the partition, keep2 behind it) disagree: the sub-buckets own up to 256 filter blocks, which only the
		 * kernel without a stage takes.  Not a reason to fail the pass: the filter gets its real zeros and the tier behind k_lc2 counts on it in memory */
		if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] a pass planned without a filter stage does not keep its records after all: filter materialised, sub-buckets to the global-scratch tier\n");
		if (fp.bf_virgin) { if (c->d_bf) HIPCK(hipMemsetAsync(c->d_bf, 0, c->bf_words * 4, c->st)); fp.bf_virgin = 0; }   /* (c->bf_virgin is off already: bloom_materialise would do nothing) */
	}
	const bool lc2 = yk_lc2_ok(fp) != 0;
	if (!lc2) fp.bf_nowb = 0;
	{
		EvTimer tm(c->st);
		if (lc2) yk_launch_lc2(fp, d_sbstart, lc_rec_in, c->d_bf, img_view(c), lo, c->d_counters, d_ovf2, c->st);
		c->ms_lds = tm.stop();
		c->st_cur.ms_insert += c->ms_lds; c->st_cur.ms_dominant_kernel += c->ms_lds; c->st_cur.n_dominant_launches += 1;
	}
	HIPCK(hipMemcpyAsync(h_cnt, c->d_counters, sizeof(h_cnt), hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	if (!lc2) {
		/* k_lc2 does not run this pass (n_hash > 32, sub-buckets that own more filter blocks than it stages, YAKAMD_LC2=0): every sub-bucket of the shard
		 * goes to the tier behind it */
		const u32 first = (u32)c->plo << s2, n_all = (u32)(c->phi - c->plo) << s2;
		std::vector<u32> all(n_all);
		for (u32 i = 0; i < n_all; ++i) all[i] = first + i;
		HIPCK(hipMemcpy(d_ovf2, all.data(), (size_t)n_all * 4, hipMemcpyHostToDevice));
		h_cnt[YKC_NOVF2] = n_all;
	}
	if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] %s: %.2f ms, %llu of %zu sub-buckets passed on\n", lc2 ? "k_lc2" : "k_lc2 not run", c->ms_lds, (unsigned long long)h_cnt[YKC_NOVF2], n_sb);
	if (h_cnt[YKC_NOVF2]) {     /* the same algorithm on tables in global scratch */
		const u32 n_ovf = (u32)h_cnt[YKC_NOVF2];
		std::vector<u32> ovf(n_ovf);
		std::vector<u64> sbs(n_sb + 1), off(n_ovf);
		HIPCK(hipMemcpy(ovf.data(), d_ovf2, n_ovf * 4, hipMemcpyDeviceToHost));
		HIPCK(hipMemcpy(sbs.data(), d_sbstart, (n_sb + 1) * 8, hipMemcpyDeviceToHost));
		/* scratch tables of 40 B per slot (5 u64), in groups of sub-buckets that stay within a quarter of the free memory */
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) != hipSuccess) return fail("hipMemGetInfo failed");
		const u64 max_words = (u64)env_i64("YAKAMD_OVF_SCRATCH_WORDS", (int64_t)(std::max<u64>(fr / 4, (u64)1 << 28) / 8));
		if (dmalloc(&d_scroff, n_ovf)) return -1;
		for (u32 i0 = 0; i0 < n_ovf;) {
			u64 words = 0;
			u32 i1 = i0;
			for (; i1 < n_ovf; ++i1) {
				const u64 n = sbs[ovf[i1] + 1] - sbs[ovf[i1]];
				u64 cap = 4096; while (cap < 2 * n) cap <<= 1;
				if (i1 > i0 && words + 5 * cap > max_words) break;
				off[i1] = words; words += 5 * cap;
			}
			if (dmalloc(&d_scr, words)) return -1;
			HIPCK(hipMemcpyAsync(d_scroff + i0, off.data() + i0, (size_t)(i1 - i0) * 8, hipMemcpyHostToDevice, c->st));
			EvTimer tm(c->st);
			yk_launch_lds_count_ovf(fp, d_sbstart, lc_rec_in, c->d_bf, img_view(c), lo, d_ovf2 + i0, i1 - i0, d_scroff + i0, d_scr, c->st);
			c->st_cur.ms_insert += tm.stop();
			HIPCK(hipStreamSynchronize(c->st));
			dfree(d_scr); d_scr = 0;
			i0 = i1;
		}
		dfree(d_scroff); d_scroff = 0;
	}
	lap("insert (k_lc2 + tiers)");
	if (keep2) { c->ret2.d_r2 = d_r2; d_r2 = 0; c->ret2.n_total = n_total; c->ret2.fp = fp; c->ret2.fp.bf_nowb = 0; }
	dfree(d_r2); dfree(d_ovf2);
	/* gather the fragments: keys per sub-table, then one contiguous list each */
	std::vector<u32> m(P, 0);
	std::vector<u64> ro(P + 1, 0);
	/* many sub-buckets per sub-table, or few sub-tables: the gather spread over the whole chip (k_lc_sum3 / k_nsel_scan / k_lc_gather) */
	const bool flat = env_i64("YAKAMD_LC_FLAT", (s2 > 11 || c->phi - c->plo < 512) ? 1 : 0) != 0;
	if (flat) yk_launch_lc_sum3(lo, s2, c->plo, c->phi, fp.t_pass0, d_segcur, c->d_lastput, d_ndist, c->st);
	else yk_launch_lc_sum(lo.nsel, s2, c->plo, c->phi, d_segcur, c->st);
	HIPCK(hipMemcpyAsync(m.data(), d_segcur, P * 4, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	for (int p = 0; p < P; ++p) ro[p + 1] = ro[p] + m[p];
	const u64 n_sel = ro[P];
	/* The sort by insertion time T.  Times are unique inside a sub-table and < 2^tbits, so the order needs no comparisons: one partition sweep by
	 * the top ts_b bits of T (the level-2 partition's kernels on {key, T} pairs), then every bin ranks its keys with a bitmap of its 2^ts_w possible
	 * times (k_ts_rank).  ts_w = 11 wherever that leaves at most 2^13 bins: a bin then never holds more keys than the ranking kernel's registers and
	 * stage (the times of first occurrences are dense at the start of a stream of reads and sparse later), and the kernel takes 2^ts_j sparse
	 * neighbours in one step (~1000 keys).  Times beyond 32 bits or bins wider than 2^18 times take the stable 8-bit passes as before */
	const int tbits = std::max(1, ceil_log2_u64(sort_tmax + 1));
	u32 m_max = 0;
	for (int p = 0; p < P; ++p) m_max = std::max(m_max, m[p]);
	int ts_b = std::min(13, std::max(0, tbits - 11));
	ts_b = (int)std::min<int64_t>(13, std::max<int64_t>(0, env_i64("YAKAMD_TS_BITS", ts_b)));
	const int ts_w = std::max(5, tbits - ts_b);
	int ts_j = 0;
	while (ts_j < ts_b && ts_w + ts_j < 15 && ((u64)m_max >> (ts_b - ts_j - 1)) <= 256) ++ts_j;   /* a mean of 256: the head of a stream of reads is several times denser */
	ts_j = (int)std::min<int64_t>(ts_b, std::max<int64_t>(0, env_i64("YAKAMD_TS_JOIN", ts_j)));
	/* short lists in many sub-tables (a filtered count of reads: ~50 K keys per sub-table) are done sooner by three launches of the stable pass */
	const bool tsort = env_i64("YAKAMD_TSORT", (m_max >= 100000 || c->phi - c->plo < 512) ? 1 : 0) != 0 && tbits <= 32 && ts_w + ts_j <= 18 && n_sel > 0;
	Rec *d_kt = 0, *d_kt2 = 0; u32 *d_tsfail = 0; u64 *d_binstart = 0;
	struct Guard2 { std::function<void()> f; ~Guard2() { f(); } } guard2{ [&]() { dfree(d_kt); dfree(d_kt2); dfree(d_tsfail); dfree(d_binstart); } };
	if (dmalloc(&kc[0], n_sel) || dmalloc(&tt[0], n_sel) || dmalloc(&d_segbase, P + 1)) return -1;
	if (tsort && (dmalloc(&d_kt, n_sel) || dmalloc(&d_tsfail, 1))) return -1;
	HIPCK(hipMemcpyAsync(d_segbase, ro.data(), (P + 1) * 8, hipMemcpyHostToDevice, c->st));
	{
		EvTimer tm(c->st);
		if (flat) {
			if (dmalloc(&d_koff, n_sb + 1)) return -1;
			yk_launch_nsel_scan(lo.nsel, s2, c->plo, c->phi, P, d_segbase, d_koff, c->st);
			yk_launch_lc_gather(lo, d_sbstart, d_koff, s2, c->plo, c->phi, kc[0], tt[0], d_kt, c->st);
		} else yk_launch_lc_compact(lo, d_sbstart, s2, c->plo, c->phi, fp.t_pass0, d_segbase, kc[0], tt[0], c->d_lastput, d_ndist, d_kt, c->st);
		c->st_cur.ms_select += tm.stop();
	}
	{
		std::vector<u32> nd(P);
		HIPCK(hipMemcpy(nd.data(), d_ndist, P * 4, hipMemcpyDeviceToHost));
		u64 tot_d = 0;
		for (int p = 0; p < P; ++p) tot_d += nd[p];
		c->st_cur.n_distinct_seen += (int64_t)tot_d;
	}
	if (keep2) {
		/* the gathered list is grouped by sub-bucket (k_lc_compact walks them in order): a copy of it + the first key of every sub-bucket */
		if ((!d_koff && dmalloc(&c->ret2.d_koff, n_sb + 1)) || dmalloc(&c->ret2.d_kkc, n_sel) || dmalloc(&c->ret2.d_segbase, P + 1)) {
			if (fp.bf_nowb) {                                         /* the records go after all: the filter they stood for is written now */
				FastParams fr = fp;
				fr.bf_nowb = 0;
				HIPCK(hipMemsetAsync(c->d_bf, 0, c->bf_words * 4, c->st));
				yk_launch_bf_rebuild(fr, d_sbstart, c->ret2.d_r2, c->d_bf, c->st);
				HIPCK(hipStreamSynchronize(c->st));
			}
			retained_drop(c); c->retain_broken = true; keep2 = false;
		}
		else {
			if (d_koff) { c->ret2.d_koff = d_koff; d_koff = 0; }        /* the flat gather has them already */
			else yk_launch_nsel_scan(lo.nsel, s2, c->plo, c->phi, P, d_segbase, c->ret2.d_koff, c->st);
			if (d_kt) yk_launch_kt_split(d_kt, n_sel, c->ret2.d_kkc, 0, c->st);
			else HIPCK(hipMemcpyAsync(c->ret2.d_kkc, kc[0], n_sel * 8, hipMemcpyDeviceToDevice, c->st));
			c->ret2.d_sbstart = d_sbstart; d_sbstart = 0;
			HIPCK(hipMemcpyAsync(c->ret2.d_segbase, d_segbase, (P + 1) * 8, hipMemcpyDeviceToDevice, c->st));
			c->ret2.n_keys = n_sel;
			c->ret2.valid = true;
			c->bf_deferred = fp.bf_nowb != 0;
		}
	}
	dfree(lo.kc); dfree(lo.T); dfree(lo.nsel); dfree(lo.lp); dfree(lo.nd); dfree(d_sbstart); dfree(d_ndist); dfree(d_koff);
	lap("gather of the selected keys");
	int cur = 0;
	bool sorted = false;
	if (tsort) {
		EvTimer tm(c->st);
		HIPCK(hipMemsetAsync(d_tsfail, 0, 4, c->st));
		const Rec *rank_in = d_kt;
		const u64 *rank_start = d_segbase;                        /* ts_b == 0: a sub-table is its one bin */
		if (ts_b > 0) {
			const u64 chs = std::max<u64>((u64)env_i64("YAKAMD_CH2", YK_CH2), (u64)32 << ts_b);
			std::vector<Chunk2> cs;
			std::vector<u32> cfs(P + 1, 0);
			for (int p = 0; p < P; ++p) {
				cfs[p] = (u32)cs.size();
				for (u64 o = ro[p]; o < ro[p + 1]; o += chs) {
					Chunk2 ch;
					ch.rec = d_kt + o; ch.spare = 0; ch.n = (u32)std::min<u64>(chs, ro[p + 1] - o); ch.bucket = (u32)p; ch.tbase = 0; ch.pad = 0; ch.before = 0; ch.after = 0;
					cs.push_back(ch);
				}
				if (cs.size() > cfs[p]) cs.back().spare = 1;
			}
			cfs[P] = (u32)cs.size();
			Chunk2 *d_cs = 0; u32 *d_cfs = 0, *d_rows = 0;
			struct Guard3 { std::function<void()> f; ~Guard3() { f(); } } guard3{ [&]() { dfree(d_cs); dfree(d_cfs); dfree(d_rows); } };
			if (dmalloc(&d_cs, cs.size()) || dmalloc(&d_cfs, P + 1) || dmalloc(&d_rows, cs.size() << ts_b) || dmalloc(&d_binstart, ((size_t)P << ts_b) + 1) || dmalloc(&d_kt2, n_sel)) return -1;
			HIPCK(hipMemcpyAsync(d_cs, cs.data(), cs.size() * sizeof(Chunk2), hipMemcpyHostToDevice, c->st));
			HIPCK(hipMemcpyAsync(d_cfs, cfs.data(), (P + 1) * 4, hipMemcpyHostToDevice, c->st));
			FastParams ft = fp;
			ft.s2_bits = ts_b; ft.ssh = ts_w; ft.rec8_in = 0; ft.rec8_out = 0;
			yk_launch_part2_ts(d_cs, (int)cs.size(), d_cfs, d_segbase, ft, P, d_rows, d_binstart, d_kt2, c->st);
			HIPCK(hipStreamSynchronize(c->st));                  /* the host tables above go out of scope */
			rank_in = d_kt2; rank_start = d_binstart;
			dfree(d_kt);                                            /* the pairs as gathered: no longer needed (a refused ranking re-splits the partitioned ones: same keys per sub-table) */
		}
		u32 h_fail = 1;
		if (yk_launch_ts_rank(rank_start, rank_in, ts_w, ts_j, (u32)c->plo << ts_b, (u32)(c->phi - c->plo) << ts_b, kc[0], tt[0], d_tsfail, c->st) == 0) {
			HIPCK(hipMemcpyAsync(&h_fail, d_tsfail, 4, hipMemcpyDeviceToHost, c->st));
			HIPCK(hipStreamSynchronize(c->st));
		}
		sorted = h_fail == 0;
		if (!sorted) {                                            /* a time seen twice, or the kernel could not be configured: the stable passes on the gathered pairs */
			if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] sort by bitmap ranks refused: stable radix passes instead\n");
			yk_launch_kt_split(rank_in, n_sel, kc[0], tt[0], c->st);
		}
		dfree(d_kt2); dfree(d_binstart);
		c->st_cur.ms_sort += tm.stop();
	}
	dfree(d_kt);
	if (!sorted) {
		if (dmalloc(&kc[1], n_sel) || dmalloc(&tt[1], n_sel)) return -1;
		EvTimer tm(c->st);
		const int sort_big = n_sel / (u64)std::max(1, c->phi - c->plo) >= 30000;
		for (int shift = 0; shift < tbits; shift += 8) {
			yk_launch_seg_sort_pass2(d_segbase, d_segcur, P, kc[cur], tt[cur], kc[cur ^ 1], tt[cur ^ 1], shift, c->st, sort_big);
			cur ^= 1;
		}
		c->st_cur.ms_sort += tm.stop();
	}
	dfree(kc[cur ^ 1]); dfree(tt[cur ^ 1]);                     /* the sort's other buffer pair: 16 bytes per key the layout stage can use */
	lap("sort by insertion time");
	{
		EvTimer tm(c->st);
		ro.resize(P);
		if (yk_run_replay(c, m, 0, kc[cur], tt[cur], c->d_lastput, 0, false, &ro)) return -1;
		c->st_cur.ms_replay += tm.stop();
	}
	lap("exact layout");
	return 0;
}

static int64_t pass_end_body(yakamd_ctx *c);

extern "C" int64_t yakamd_pass_end(yak_ch_t *h)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || !c->in_pass) return fail("no pass open");
	int64_t r = pass_end_body(c);
	if (r >= 0) {                                              /* a launch that was refused (bad configuration) reports nothing by itself */
		const hipError_t e = hipGetLastError();
		if (e != hipSuccess) r = fail("a HIP call of this pass was refused: %s", hipGetErrorString(e));
	}
	if (r < 0) pass_free(c);                                 /* a failed pass is closed too: the table stays usable (the pass's k-mers are lost, the error is reported) */
	return r;
}

static int64_t pass_end_body(yakamd_ctx *c)
{
	HIPCK(hipSetDevice(c->dev));
	const int P = c->P;
	int64_t n_ins = 0;
	if (c->k >= 32 && yk_bad_hash_seen(c->st)) { pass_free(c); return fail("a 64-bit k-mer hash equals the empty-slot pattern: unsupported input for k >= 32"); }
	if (!c->create_new) {
		if (c->d_delta && c->delta_dirty) yk_launch_img_fold(img_view(c), c->n_slots, c->st);   /* (k_cnt2_apply writes its counts straight into the keys) */
		HIPCK(hipStreamSynchronize(c->st));
		dfree(c->d_delta);
		c->host_valid = false;
		retained_drop(c);                                        /* used or not: the next pass is another input's */
	} else if (c->or_mode && !(c->fast && !c->acc.s)) {
		pass_free(c);
		return fail("flag-mode loads need the exclusive-ownership path (prefix length <= 13, input within the device budget)");
	} else if (c->fast && !c->acc.s) {
		if (fast_finish(c, true)) return -1;
		n_ins = (int64_t)(c->img_keys_total - c->keys_at_begin);   /* earlier slices of the pass included */
		c->st_cur.n_new_keys = n_ins;
	} else {
		if (c->fast && fast_abandon(c)) return -1;          /* cannot happen today; keeps the invariant explicit */
		const u64 before = c->keys_at_begin;                 /* slices counted earlier in this pass included */
		if (c->img_keys_total && c->d_delta) yk_launch_img_fold(img_view(c), c->n_slots, c->st);   /* put-calls that hit existing keys */
		std::vector<u32> m(P, 0);
		std::vector<u64> seg_off(P + 1, 0);
		u32 *d_segcnt = 0, *d_segcur = 0; u64 *d_segoff = 0, *kc[2] = { 0, 0 }, *tt[2] = { 0, 0 };
		struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ [&]() { dfree(d_segcnt); dfree(d_segcur); dfree(d_segoff); dfree(kc[0]); dfree(kc[1]); dfree(tt[0]); dfree(tt[1]); } };
		if (dmalloc(&d_segcnt, P) || dmalloc(&d_segcur, P) || dmalloc(&d_segoff, P + 1)) return -1;
		HIPCK(hipMemsetAsync(d_segcnt, 0, P * 4, c->st));
		HIPCK(hipMemsetAsync(d_segcur, 0, P * 4, c->st));
		int cur = 0;
		if (c->acc.s) {
			{
				EvTimer tm(c->st);
				yk_launch_select_count(c->acc, c->bloom_mode, P, d_segcnt, c->st);
				HIPCK(hipMemcpyAsync(m.data(), d_segcnt, P * 4, hipMemcpyDeviceToHost, c->st));
				HIPCK(hipStreamSynchronize(c->st));
				for (int p = 0; p < P; ++p) seg_off[p + 1] = seg_off[p] + m[p];
				const u64 tot = seg_off[P];
				if (dmalloc(&kc[0], tot) || dmalloc(&kc[1], tot) || dmalloc(&tt[0], tot) || dmalloc(&tt[1], tot)) return -1;
				HIPCK(hipMemcpyAsync(d_segoff, seg_off.data(), (P + 1) * 8, hipMemcpyHostToDevice, c->st));
				yk_launch_select_scatter(c->acc, c->bloom_mode, P, d_segoff, d_segcur, kc[0], tt[0], c->st);
				c->st_cur.ms_select += tm.stop();
			}
			dfree(c->acc.s);                              /* the accumulator is no longer needed */
			{
				EvTimer tm(c->st);
				const int tbits = std::max(1, ceil_log2_u64(c->t_end + 1));
				for (int shift = 0; shift < tbits; shift += 8) {
					yk_launch_seg_sort_pass(d_segoff, P, kc[cur], tt[cur], kc[cur ^ 1], tt[cur ^ 1], shift, c->st);
					cur ^= 1;
				}
				c->st_cur.ms_sort += tm.stop();
			}
		} else {
			HIPCK(hipMemcpyAsync(d_segoff, seg_off.data(), (P + 1) * 8, hipMemcpyHostToDevice, c->st));
			if (dmalloc(&kc[0], 1) || dmalloc(&tt[0], 1)) return -1;
		}
		{
			EvTimer tm(c->st);
			if (yk_run_replay(c, m, d_segoff, kc[cur], tt[cur], c->d_lastput, 0, false)) return -1;
			c->st_cur.ms_replay += tm.stop();
		}
		n_ins = (int64_t)(c->img_keys_total - before);
		c->st_cur.n_distinct_seen = (int64_t)c->acc_count;
		c->st_cur.n_new_keys = n_ins;
	}
	pass_free(c);
	c->st_cur.ms_total = now_ms() - c->st_cur.ms_total;
	c->st_last = c->st_cur;
	return n_ins;
}

extern "C" void yakamd_debug_counters(uint32_t *out4)
{
	out4[0] = out4[1] = 0;
	yk_par_counters(&out4[0], &out4[1]);
	yk_replay_counters(&out4[2], &out4[3]);
}

/* small runtime services for callers that hold no HIP runtime of their own (bench.py's single-GPU mode, the tests): page-locked host memory and
 * "everything queued on the current device is done" */
extern "C" void *yakamd_host_alloc(size_t bytes) { void *p = 0; return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : 0; }
extern "C" void yakamd_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int yakamd_device_sync(void) { return hipDeviceSynchronize() == hipSuccess ? 0 : fail("hipDeviceSynchronize: %s", hipGetErrorString(hipGetLastError())); }
extern "C" int yakamd_mem_info(size_t *free_bytes, size_t *total_bytes) { return hipMemGetInfo(free_bytes, total_bytes) == hipSuccess ? 0 : -1; }
extern "C" void *yakamd_dev_alloc(size_t bytes) { void *p = 0; return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? p : 0; }
extern "C" void yakamd_dev_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int yakamd_memcpy_h2d(void *d, const void *s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
extern "C" int yakamd_memcpy_d2h(void *d, const void *s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }

extern "C" int yakamd_get_stats(yak_ch_t *h, yakamd_stats_t *st)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c) return -1;
	*st = c->st_last;
	return 0;
}

int yk_ctx_hist(yakamd_ctx *c, int64_t *cnt1024)
{
	HIPCK(hipSetDevice(c->dev));
	u64 *d_h = 0;
	if (dmalloc(&d_h, 1024)) return -1;
	HIPCK(hipMemsetAsync(d_h, 0, 1024 * 8, c->st));
	yk_launch_img_hist(img_view(c), c->n_slots, d_h, c->st);
	const hipError_t e = hipMemcpyAsync(cnt1024, d_h, 1024 * 8, hipMemcpyDeviceToHost, c->st);
	const hipError_t e2 = hipStreamSynchronize(c->st);
	dfree(d_h);
	return e == hipSuccess && e2 == hipSuccess ? 0 : fail("hist: %s", hipGetErrorString(e != hipSuccess ? e : e2));
}

int yk_ctx_setcnt(yakamd_ctx *c, int cnt)
{
	HIPCK(hipSetDevice(c->dev));
	yk_launch_img_setcnt(img_view(c), c->n_slots, (u32)cnt & 1023u, c->st);
	HIPCK(hipStreamSynchronize(c->st));
	c->host_valid = false;
	return 0;
}

int yk_ctx_clear(yakamd_ctx *c)
{
	HIPCK(hipSetDevice(c->dev));
	yk_launch_img_clear(img_view(c), c->n_slots, c->st);
	HIPCK(hipStreamSynchronize(c->st));
	c->host_valid = false;
	return 0;
}

/* reference htab.c:180-208 */
/* htab.c:171-197 (shrink), 287-347 (subtract / isec): per sub-table, a new set resized for the old
 * key count receives, in old slot order, the keys that pass the test.  which: 0 count range only,
 * 1 and absent from `other`, 2 and present in `other` */
static int rebuild(yakamd_ctx *c, int cmin, int cmax, int which, yakamd_ctx *other, u64 *tot)
{
	HIPCK(hipSetDevice(c->dev));
	const int P = c->P;
	if (other && (other->pre != c->pre || other->k != c->k)) return fail("tables of different k / prefix length");
	const ImgView ov = other ? img_view(other) : img_view(c);
	std::vector<u32> m(P), init(P);
	std::vector<u64> seg_off(P + 1, 0);
	u32 *d_segcnt = 0; u64 *d_segoff = 0, *d_kc = 0;
	struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ [&]() { dfree(d_segcnt); dfree(d_segoff); dfree(d_kc); } };
	const int RG = yk_shrink_shares();                            /* workgroups per sub-table */
	std::vector<u32> mr((size_t)P * RG);
	if (dmalloc(&d_segcnt, (size_t)P * RG) || dmalloc(&d_segoff, P + 1)) return -1;
	yk_launch_shrink_count(img_view(c), P, cmin, cmax, which, ov, d_segcnt, c->st, 1);
	HIPCK(hipMemcpyAsync(mr.data(), d_segcnt, mr.size() * 4, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	for (int p = 0; p < P; ++p) {
		m[p] = 0;
		for (int g = 0; g < RG; ++g) m[p] += mr[(size_t)p * RG + g];
		seg_off[p + 1] = seg_off[p] + m[p]; init[p] = kh_bits_for(c->h_count[p]);
	}
	if (dmalloc(&d_kc, seg_off[P])) return -1;
	HIPCK(hipMemcpyAsync(d_segoff, seg_off.data(), (P + 1) * 8, hipMemcpyHostToDevice, c->st));
	yk_launch_shrink_scatter(img_view(c), P, cmin, cmax, which, ov, d_segoff, d_kc, c->st, d_segcnt);
	EvTimer tm(c->st);
	const int r = yk_run_replay(c, m, d_segoff, d_kc, 0, 0, &init, true);
	c->st_last.ms_shrink = tm.stop();
	if (r) return r;
	*tot = c->img_keys_total;
	return 0;
}

int yk_ctx_shrink(yakamd_ctx *c, int cmin, int cmax, u64 *tot) { return rebuild(c, cmin, cmax, 0, 0, tot); }
int yk_ctx_subtract(yakamd_ctx *c, yakamd_ctx *other, u64 *tot) { return rebuild(c, 0, 1023, 1, other, tot); }
int yk_ctx_isec(yakamd_ctx *c, yakamd_ctx *other, u64 *tot) { return rebuild(c, 0, 1023, 2, other, tot); }

/* khashl resize of every sub-table whose entry in new_bits differs from YK_NOCAP-as-"leave alone"
 * (new_bits[p] == 0xfffffffe); the table image moves to a fresh arena */
#define YK_LEAVE 0xfffffffeu
static int resize_tables(yakamd_ctx *c, const std::vector<u32> &new_bits)
{
	HIPCK(hipSetDevice(c->dev));
	const int P = c->P;
	std::vector<ResizeTask> tasks(P);
	std::vector<u64> new_off(P);
	std::vector<u32> bits_after(P);
	u64 tot = 0;
	for (int p = 0; p < P; ++p) {
		ResizeTask &t = tasks[p];
		t.old_bits = c->h_bits[p]; t.old_off = c->h_off[p]; t.pad = 0;
		t.rehash = new_bits[p] != YK_LEAVE;
		t.new_bits = t.rehash ? new_bits[p] : c->h_bits[p];
		bits_after[p] = t.new_bits;
		const u64 n = t.old_bits == YK_NOCAP ? 0 : (u64)1 << t.old_bits, N = t.new_bits == YK_NOCAP ? 0 : (u64)1 << t.new_bits;
		t.new_off = new_off[p] = tot;
		tot += std::max<u64>(32, std::max(n, N));
	}
	u64 *nk = 0; u32 *nu = 0, *su = 0; ResizeTask *d_tasks = 0;
	if (dmalloc(&nk, tot) || dmalloc(&nu, tot / 32) || dmalloc(&su, tot / 32) || dmalloc(&d_tasks, P)) return -1;
	HIPCK(hipMemsetAsync(nk, 0xff, tot * 8, c->st));
	HIPCK(hipMemsetAsync(nu, 0, tot / 8, c->st));
	HIPCK(hipMemcpyAsync(d_tasks, tasks.data(), P * sizeof(ResizeTask), hipMemcpyHostToDevice, c->st));
	yk_launch_resize(d_tasks, P, c->d_keys, c->d_used, nk, nu, su, c->st);
	HIPCK(hipStreamSynchronize(c->st));
	dfree(su); dfree(d_tasks);
	dfree(c->d_keys); dfree(c->d_used); dfree(c->d_delta);
	c->d_keys = nk; c->d_used = nu; c->n_slots = tot;
	c->h_off = new_off; c->h_bits = bits_after;
	HIPCK(hipMemcpyAsync(c->d_bits, c->h_bits.data(), P * 4, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(c->d_off, c->h_off.data(), P * 8, hipMemcpyHostToDevice, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	c->host_valid = false;
	return 0;
}

/* new_bits of khashl's resize(want) on a set of (cap, count), or YK_LEAVE when it refuses (khashl.h:155-160) */
static u32 resize_target(u32 count, u32 want)
{
	const u32 nb = kh_bits_for(want);
	const u64 N = (u64)1 << nb;
	return count > (N >> 1) + (N >> 2) ? YK_LEAVE : nb;
}

/* htab.c:102-110: sub-tables filled to less than a third are resized to 3 x their key count */
int yk_ctx_tighten(yakamd_ctx *c)
{
	const int P = c->P;
	std::vector<u32> nb(P, YK_LEAVE);
	bool any = false;
	for (int p = 0; p < P; ++p) {
		const u64 cap = c->h_bits[p] == YK_NOCAP ? 0 : (u64)1 << c->h_bits[p];
		if ((u64)c->h_count[p] * 3 < cap) { nb[p] = resize_target(c->h_count[p], c->h_count[p] * 3); any = any || nb[p] != YK_LEAVE; }
	}
	return any ? resize_tables(c, nb) : 0;
}

/* htab.c:262-266: before a merge, grow sub-table p for count0 + count1 keys at 75 % load */
int yk_ctx_merge_presize(yakamd_ctx *c, yakamd_ctx *other)
{
	const int P = c->P;
	std::vector<u32> nb(P, YK_LEAVE);
	bool any = false;
	/* only the sub-tables both sides own: a shard of a table sharded over prefix ranges says nothing about the others (each sub-table
	 * must meet its one pre-resize with the true key count of the other side) */
	for (int p = std::max(c->plo, other->plo); p < std::min(c->phi, other->phi); ++p) {
		const u64 cap = c->h_bits[p] == YK_NOCAP ? 0 : (u64)1 << c->h_bits[p];
		const u64 want = ((u64)c->h_count[p] + other->h_count[p]) * 4 / 3 + 1;
		if (want > cap) { nb[p] = resize_target(c->h_count[p], (u32)want); any = any || nb[p] != YK_LEAVE; }
	}
	return any ? resize_tables(c, nb) : 0;
}

/* keys of `c` with cmin <= count <= cmax, sub-table by sub-table in slot order, as full hashes +
 * list positions on the device (caller frees both with yakamd_dev_free-compatible pool_free) */
int yk_ctx_list_hashes(yakamd_ctx *c, int cmin, int cmax, u64 **d_hash, u32 **d_t, u64 *n)
{
	HIPCK(hipSetDevice(c->dev));
	const int P = c->P;
	std::vector<u32> m(P);
	std::vector<u64> seg_off(P + 1, 0);
	u32 *d_segcnt = 0; u64 *d_segoff = 0, *d_kc = 0;
	*d_hash = 0; *d_t = 0;
	bool done = false;
	struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ [&]() { dfree(d_segcnt); dfree(d_segoff); dfree(d_kc); if (!done) { dfree(*d_hash); dfree(*d_t); } } };
	if (dmalloc(&d_segcnt, P) || dmalloc(&d_segoff, P + 1)) return -1;
	yk_launch_shrink_count(img_view(c), P, cmin, cmax, 0, img_view(c), d_segcnt, c->st);
	HIPCK(hipMemcpyAsync(m.data(), d_segcnt, P * 4, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	for (int p = 0; p < P; ++p) seg_off[p + 1] = seg_off[p] + m[p];
	*n = seg_off[P];
	if (dmalloc(&d_kc, seg_off[P]) || dmalloc(d_hash, seg_off[P]) || dmalloc(d_t, seg_off[P])) return -1;
	HIPCK(hipMemcpyAsync(d_segoff, seg_off.data(), (P + 1) * 8, hipMemcpyHostToDevice, c->st));
	yk_launch_shrink_scatter(img_view(c), P, cmin, cmax, 0, img_view(c), d_segoff, d_kc, c->st);
	yk_launch_keys_to_hashes(d_kc, d_segoff, P, c->pre, *d_hash, *d_t, c->st);
	HIPCK(hipStreamSynchronize(c->st));
	done = true;
	return 0;
}

/* the .yak bytes of sub-tables [lo, hi) -- per sub-table capacity and size (4 bytes each) and the stored keys in ascending slot order,
 * htab.c:385-389 -- put together on the device: *d_img (pool memory: yk_pool_release) holds *n_words 8-byte words, ready on the table's stream */
int yk_ctx_dump_image_dev(yakamd_ctx *c, int lo, int hi, u64 **d_img, u64 *n_words)
{
	HIPCK(hipSetDevice(c->dev));
	const int P = c->P, n = hi - lo;
	*d_img = 0; *n_words = 0;
	if (lo < 0 || hi > P || n <= 0) return fail("yk_ctx_dump_image_dev: sub-tables [%d, %d) of %d", lo, hi, P);
	std::vector<u64> seg_off(P + 1, ~0ull), head(2 * (size_t)n);     /* ~0: a sub-table outside [lo, hi) is skipped */
	u64 at = 0;
	for (int p = lo; p < hi; ++p) {
		head[p - lo] = at;
		head[n + p - lo] = (u64)(c->h_bits[p] == YK_NOCAP ? 0 : 1u << c->h_bits[p]) | (u64)c->h_count[p] << 32;
		seg_off[p] = ++at;
		at += c->h_count[p];
	}
	u64 *d_segoff = 0, *d_head = 0, *img = 0;
	struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ [&]() { dfree(d_segoff); dfree(d_head); } };
	if (dmalloc(&d_segoff, P + 1) || dmalloc(&d_head, 2 * (size_t)n) || dmalloc(&img, at)) { dfree(img); return -1; }
	HIPCK(hipMemcpyAsync(d_segoff, seg_off.data(), (P + 1) * 8, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(d_head, head.data(), head.size() * 8, hipMemcpyHostToDevice, c->st));
	yk_launch_shrink_scatter(img_view(c), P, 0, 1023, 0, img_view(c), d_segoff, img, c->st);
	yk_launch_put_u64(d_head, d_head + n, (u32)n, img, c->st);
	HIPCK(hipStreamSynchronize(c->st));                            /* (the two small host arrays and the offsets go away with this call) */
	*d_img = img; *n_words = at;
	return 0;
}
void yk_ctx_gate(yakamd_ctx *c, bool on) { c->gate_off = !on; }
void yk_ctx_or_mode(yakamd_ctx *c, int mode) { c->or_mode = mode; }

/* khashl resize(want[p]) on every sub-table (htab.c:441 on an existing table): same size re-places in place */
int yk_ctx_resize_to(yakamd_ctx *c, const uint32_t *want)
{
	const int P = c->P;
	std::vector<u32> nb(P, YK_LEAVE);
	bool any = false;
	for (int p = 0; p < P; ++p) { nb[p] = resize_target(c->h_count[p], want[p]); any = any || nb[p] != YK_LEAVE; }
	return any ? resize_tables(c, nb) : 0;
}
u64 yk_ctx_keys_total(yakamd_ctx *c) { return c->img_keys_total; }
void yk_ctx_range(yakamd_ctx *c, int *lo, int *hi) { *lo = c->plo; *hi = c->phi; }

/* reference htab.c:441-447: resize each sub-table to its saved capacity, then put in file order */
int yk_ctx_load(yakamd_ctx *c, const uint32_t *caps, const uint32_t *sizes, const uint64_t *keys)
{
	HIPCK(hipSetDevice(c->dev));
	const int P = c->P;
	std::vector<u32> m(P), init(P);
	u64 tot = 0;
	for (int p = 0; p < P; ++p) { m[p] = sizes[p]; init[p] = kh_bits_for(caps[p]); tot += sizes[p]; }
	u64 *d_kc = 0;
	if (dmalloc(&d_kc, tot)) return -1;
	HIPCK(hipMemcpyAsync(d_kc, keys, tot * 8, hipMemcpyHostToDevice, c->st));
	const int r = yk_run_replay(c, m, 0, d_kc, 0, 0, &init, true);
	dfree(d_kc);
	return r;
}

/* ------------------------------------------------------------------------------------------
 * host mirror
 * ------------------------------------------------------------------------------------------ */
struct yak_ht_t { uint32_t bits, count; uint32_t *used; uint64_t *keys; };   /* same shape as khashl.h:104-109 */

int yk_ctx_sync_host(yakamd_ctx *c, yak_ch_t *h)
{
	/* yak_ch_get() is called concurrently by the reference's kt_for workers (qv.c:59, triobin.c): the
	 * first callers serialise on the refresh, later ones only read the flag */
	if (__atomic_load_n(&c->host_valid, __ATOMIC_ACQUIRE)) return 0;
	static std::mutex mu;
	std::lock_guard<std::mutex> lk(mu);
	if (c->host_valid) return 0;
	HIPCK(hipSetDevice(c->dev));
	if (c->hm_slots < c->n_slots) {
		if (c->hm_keys) hipHostFree(c->hm_keys);
		if (c->hm_used) hipHostFree(c->hm_used);
		c->hm_keys = 0; c->hm_used = 0;
		HIPCK(hipHostMalloc((void**)&c->hm_keys, c->n_slots * 8));
		HIPCK(hipHostMalloc((void**)&c->hm_used, c->n_slots / 8));
		c->hm_slots = c->n_slots;
	}
	HIPCK(hipMemcpyAsync(c->hm_keys, c->d_keys, c->n_slots * 8, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipMemcpyAsync(c->hm_used, c->d_used, c->n_slots / 8, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	if (!c->hts) c->hts = (yak_ht_t*)calloc(c->P, sizeof(yak_ht_t));
	for (int p = 0; p < c->P; ++p) {
		yak_ht_t *g = &c->hts[p];
		const bool has = c->h_bits[p] != YK_NOCAP;
		g->bits = has ? c->h_bits[p] : 0;
		g->count = c->h_count[p];
		g->keys = has ? (uint64_t*)(c->hm_keys + c->h_off[p]) : 0;
		g->used = c->hm_used + c->h_off[p] / 32;
		if (h) h->h[p].h = g;
	}
	__atomic_store_n(&c->host_valid, true, __ATOMIC_RELEASE);
	return 0;
}

extern "C" int yakamd_sync_host(yak_ch_t *h)
{
	yakamd_ctx *c = ctx_of(h);
	return c ? yk_ctx_sync_host(c, h) : fail("not an engine table");
}

extern "C" int yakamd_subtable(yak_ch_t *h, int i, uint32_t *capacity, uint32_t *size)
{
	yakamd_ctx *c = ctx_of(h);
	if (!c || i < 0 || i >= c->P) return -1;
	*capacity = c->h_bits[i] == YK_NOCAP ? 0 : 1u << c->h_bits[i];
	*size = c->h_count[i];
	return 0;
}

u64 yk_ctx_list_time(yakamd_ctx *c, u64 n) { const u64 t = c->list_t; c->list_t += n; return t; }
void yk_ctx_lock(yakamd_ctx *c) { c->api_mu.lock(); }
void yk_ctx_unlock(yakamd_ctx *c) { c->api_mu.unlock(); }

/* a device buffer of at least `bytes` that lives as long as the context (small repeated calls: yak_ch_insert_list) */
void *yk_ctx_scratch(yakamd_ctx *c, size_t bytes)
{
	if (bytes <= c->scratch_bytes) return c->d_scratch;
	if (hipSetDevice(c->dev) != hipSuccess) return 0;
	uint8_t *q = (uint8_t*)c->d_scratch;
	dfree(q);
	c->d_scratch = 0; c->scratch_bytes = 0;
	const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 16);
	if (dmalloc(&q, want)) return 0;
	c->d_scratch = q; c->scratch_bytes = want;
	return q;
}

/* reference htab.c:80-91: saturating ++ of one stored k-mer; -1 if absent.  One single-lane kernel; a valid host
 * mirror is patched in place instead of being refreshed */
int yk_ctx_inc(yakamd_ctx *c, u64 hash, int *count)
{
	if (c->in_pass) return fail("yak_ch_inc during an open pass");
	HIPCK(hipSetDevice(c->dev));
	u64 out[2];
	yk_launch_img_inc(img_view(c), hash, c->d_counters, c->st);
	HIPCK(hipMemcpyAsync(out, c->d_counters, 16, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	if (out[0] == ~0ull) { *count = -1; return 0; }
	*count = (int)out[1];
	if (c->host_valid && c->hm_keys) c->hm_keys[out[0]] = (c->hm_keys[out[0]] & ~1023ull) | out[1];
	return 0;
}
int yk_ctx_device(yakamd_ctx *c) { return c->dev; }
hipStream_t yk_ctx_stream(yakamd_ctx *c) { return c->st; }
const yak_ht_t *yk_ctx_ht(yakamd_ctx *c, int p) { return c->hts ? &c->hts[p] : 0; }
