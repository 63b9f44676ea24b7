/*
 * yak_api.cpp -- the drop-in C surface (include/yak.h) on top of the HBM-resident engine.
 *
 * Every function keeps the reference's name, argument meaning, return convention and messages
 * (reference file:line cited per function).  Host work here is orchestration only: parsing the
 * input file, staging bases, writing the .yak file from the host mirror.  All counting, bloom
 * gating, table layout, clearing and shrinking run in the HIP kernels of kernels.hip.
 */
#include "yak_host.h"
#include <unistd.h>

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

} /* extern "C" */
static double yk_realtime0 = -1;
double yk_realtime(void)                                     /* (C++ linkage, as yak_host.h declares them) */
{
	struct timeval tp;
	gettimeofday(&tp, 0);
	const double t = tp.tv_sec + tp.tv_usec * 1e-6;
	if (yk_realtime0 < 0) yk_realtime0 = t;
	return t - yk_realtime0;
}
double yk_cputime(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}
extern "C" {

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
	if (hipSetDevice(dev) != hipSuccess) return false;            /* the events below are recorded on `st`: they must be this device's */
	if (mem) return hipMemcpyAsync(mem + off, d_src, bytes, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
	enum { NB = 6, W = 3 };
	const size_t CH = (size_t)8 << 20, n_ch = (bytes + CH - 1) / CH;
	static std::mutex mu;                                         /* the staging buffers are the process's: one dump at a time uses them */
	static void *stage[NB] = { 0 };
	std::lock_guard<std::mutex> lk(mu);
	for (int i = 0; i < NB; ++i) if (!stage[i] && hipHostMalloc(&stage[i], CH, hipHostMallocPortable) != hipSuccess) { stage[i] = 0; return false; }
	hipEvent_t ev[NB];
	for (int i = 0; i < NB; ++i) if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { while (i-- > 0) (void)hipEventDestroy(ev[i]); return false; }
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
	struct GzDrop { pgz::Reader *p; ~GzDrop() { pgz::Reader *q = p; yk_reap_later([q]() { delete q; }); } } gz_drop{ gz_p };   /* (its buffers go back to the system behind the caller's back) */
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
		g_t_parse_windows.store(0, std::memory_order_relaxed);
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
		                                      yk_realtime() - tp0, g_t_parse_windows.load(), g_t_first_window.load(), t_sink, t_open_wait);
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
	/* count.c:172-173: an unreadable file leaves the table untouched.  (Asked, not tried: opening and closing a named pipe -- `yak recount <(zcat ...)` --
	 * would take the stream away from the count behind it) */
	if (fn != 0 && strcmp(fn, "-") != 0 && access(fn, R_OK) != 0) return;
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
	struct GzDrop { pgz::Reader *p; ~GzDrop() { pgz::Reader *q = p; yk_reap_later([q]() { delete q; }); } } gz_drop{ gz_p };
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


