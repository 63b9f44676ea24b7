/*
 * yko_core.c -- ORACLE (test infrastructure only).  CPU restatement of the hashing, the khashl
 * slot set, the blocked bloom filter and the yak_ch_* table operations of lh3/yak.
 * Written from the behaviour documented in SURVEY.md section 8(a); reference lines are cited per
 * function so a reviewer can check parity.  Nothing here is used by the shipped GPU path.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "yko.h"

/* ------------------------------------------------------------------ options (misc.c:23-32) */
void yko_copt_init(yko_copt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->bf_shift = 0; o->bf_n_hash = 4; o->k = 31; o->pre = 10; o->n_thread = 4;
	o->chunk_size = 10000000;
}

/* ------------------------------------------------------------------ base code (misc.c:4-21)
 * A/a->0 C/c->1 G/g->2 T/t/U/u->3, raw bytes 0..3 map to themselves, all else 4. */
const unsigned char yko_nt4[256] = {
#define R4(v) v, v, v, v
#define R16(v) R4(v), R4(v), R4(v), R4(v)
	0, 1, 2, 3, R4(4), R4(4), R4(4),               /* 0x00-0x0f */
	R16(4), R16(4), R16(4),                          /* 0x10-0x3f */
	4, 0, 4, 1, 4, 4, 4, 2, R4(4), R4(4),            /* 0x40-0x4f : A C G */
	4, 4, 4, 4, 3, 3, 4, 4, R4(4), R4(4),            /* 0x50-0x5f : T U   */
	4, 0, 4, 1, 4, 4, 4, 2, R4(4), R4(4),            /* 0x60-0x6f : a c g */
	4, 4, 4, 4, 3, 3, 4, 4, R4(4), R4(4),            /* 0x70-0x7f : t u   */
	R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4)
#undef R16
#undef R4
};

/* ------------------------------------------------------------------ hashes (yak-priv.h) */
uint64_t yko_hash64(uint64_t x, uint64_t m)                 /* yak-priv.h:11-21 */
{
	x = (~x + (x << 21)) & m;
	x ^= x >> 24;
	x = (x + (x << 3) + (x << 8)) & m;
	x ^= x >> 14;
	x = (x + (x << 2) + (x << 4)) & m;
	x ^= x >> 28;
	x = (x + (x << 31)) & m;
	return x;
}

uint64_t yko_hash64_64(uint64_t x)                          /* yak-priv.h:23-33 */
{
	x = ~x + (x << 21);
	x ^= x >> 24;
	x = x + (x << 3) + (x << 8);
	x ^= x >> 14;
	x = x + (x << 2) + (x << 4);
	x ^= x >> 28;
	x = x + (x << 31);
	return x;
}

uint64_t yko_hash_long(const uint64_t x[4])                  /* yak-priv.h:35-39 */
{
	int j = x[1] < x[3] ? 0 : 1;   /* strand chosen on the high bit-planes only */
	return yko_hash64_64(x[2 * j]) + yko_hash64_64(x[2 * j + 1]);
}

/* inverse of yko_hash64 on [0, mask]  (yak-priv.h:41-68).  Each forward step is undone in
 * reverse order: x + (x<<s) is undone by fixed-point iteration, x ^ x>>s by repeated xor,
 * the odd multipliers 21 and 265 by their inverses modulo 2^64. */
uint64_t yko_hash64_inv(uint64_t x, uint64_t m)
{
	uint64_t t;
	t = x - (x << 31); x = (x - (t << 31)) & m;             /* undo x + (x<<31) */
	t = x ^ x >> 28; x = x ^ t >> 28;                       /* undo x ^ x>>28  */
	x = (x * 14933078535860113213ULL) & m;                  /* undo *21        */
	t = x ^ x >> 14; t = x ^ t >> 14; t = x ^ t >> 14; x = x ^ t >> 14;
	x = (x * 15244667743933553977ULL) & m;                  /* undo *265       */
	t = x ^ x >> 24; x = x ^ t >> 24;                       /* undo x ^ x>>24  */
	t = ~x; t = ~(x - (t << 21)); t = ~(x - (t << 21)); x = ~(x - (t << 21)) & m;
	return x;
}

uint32_t yko_h2b(uint32_t hash, uint32_t bits)               /* khashl.h:98 (Fibonacci) */
{
	return (uint32_t)(hash * 2654435769U) >> (32 - bits);
}

/* ------------------------------------------------------------------ slot set (khashl.h) */
#define KEY_ID(x)   ((x) >> YKO_COUNTER_BITS)                /* htab.c:9-10: eq/hash drop the count */
#define USED(u, i)  ((u)[(i) >> 5] >> ((i) & 31) & 1U)
#define MARK(u, i)  ((u)[(i) >> 5] |= 1U << ((i) & 31))
#define UNMARK(u, i) ((u)[(i) >> 5] &= ~(1U << ((i) & 31)))
static inline uint32_t fwords(uint32_t n) { return n < 32 ? 1 : n >> 5; }   /* khashl.h:96 */
static inline uint32_t home_of(uint64_t key, uint32_t bits) { return yko_h2b((uint32_t)KEY_ID(key), bits); }

static yko_set_t *set_new(void) { return (yko_set_t*)calloc(1, sizeof(yko_set_t)); }
static void set_free(yko_set_t *s) { if (s) { free(s->keys); free(s->used); free(s); } }

uint32_t yko_set_capacity(const yko_set_t *s) { return s->keys ? 1U << s->bits : 0U; }  /* khashl.h:309 */

uint32_t yko_set_get(const yko_set_t *s, uint64_t key)       /* khashl.h:137-150 */
{
	uint32_t n, mask, i, first;
	if (s->keys == 0) return 0;
	n = 1U << s->bits; mask = n - 1;
	i = first = home_of(key, s->bits);
	while (USED(s->used, i) && KEY_ID(s->keys[i]) != KEY_ID(key)) {
		i = (i + 1) & mask;
		if (i == first) return n;
	}
	return USED(s->used, i) ? i : n;
}

int yko_set_resize(yko_set_t *s, uint32_t want)               /* khashl.h:152-195 */
{
	uint32_t lg = 0, x = want, old_n, new_n, new_bits, new_mask, j, *nu;
	while ((x >>= 1) != 0) ++lg;
	if (want & (want - 1)) ++lg;                              /* round up to a power of two */
	new_bits = lg > 2 ? lg : 2;                               /* at least 4 slots */
	new_n = 1U << new_bits;
	if (s->count > (new_n >> 1) + (new_n >> 2)) return 0;     /* would exceed 75 %: refuse */
	nu = (uint32_t*)calloc(fwords(new_n), sizeof(uint32_t));
	old_n = yko_set_capacity(s);
	if (old_n < new_n) s->keys = (uint64_t*)realloc(s->keys, (size_t)new_n * 8);
	new_mask = new_n - 1;
	/* in-place re-placement: every live slot is lifted and dropped at its new position; a
	 * still-unmoved occupant of that position is lifted in turn (the "kick-out" chain) */
	for (j = 0; j != old_n; ++j) {
		uint64_t key;
		if (!USED(s->used, j)) continue;
		key = s->keys[j];
		UNMARK(s->used, j);
		for (;;) {
			uint32_t i = home_of(key, new_bits);
			while (USED(nu, i)) i = (i + 1) & new_mask;
			MARK(nu, i);
			if (i < old_n && USED(s->used, i)) {
				uint64_t t = s->keys[i]; s->keys[i] = key; key = t;
				UNMARK(s->used, i);
			} else { s->keys[i] = key; break; }
		}
	}
	if (old_n > new_n) s->keys = (uint64_t*)realloc(s->keys, (size_t)new_n * 8);
	free(s->used);
	s->used = nu; s->bits = new_bits;
	return 0;
}

uint32_t yko_set_put(yko_set_t *s, uint64_t key, int *absent) /* khashl.h:197-221 */
{
	uint32_t n = yko_set_capacity(s), mask, i, first;
	*absent = -1;
	if (s->count >= (n >> 1) + (n >> 2)) {                    /* grow BEFORE looking the key up */
		if (yko_set_resize(s, n + 1) < 0) return n;
		n = 1U << s->bits;
	}
	mask = n - 1;
	i = first = home_of(key, s->bits);
	while (USED(s->used, i) && KEY_ID(s->keys[i]) != KEY_ID(key)) {
		i = (i + 1) & mask;
		if (i == first) break;
	}
	if (!USED(s->used, i)) {
		s->keys[i] = key; MARK(s->used, i); ++s->count; *absent = 1;
	} else *absent = 0;
	return i;
}

/* ------------------------------------------------------------------ bloom filter (bbf.c) */
yko_bf_t *yko_bf_init(int n_shift, int n_hashes)             /* bbf.c:5-17 */
{
	yko_bf_t *b;
	void *p = 0;
	if (n_shift + YKO_BLK_SHIFT > 64 || n_shift < YKO_BLK_SHIFT) return 0;
	b = (yko_bf_t*)calloc(1, sizeof(*b));
	b->n_shift = n_shift; b->n_hashes = n_hashes;
	if (posix_memalign(&p, 64, (size_t)1 << (n_shift - 3)) != 0) { free(b); return 0; }
	memset(p, 0, (size_t)1 << (n_shift - 3));
	b->b = (uint8_t*)p;
	return b;
}

void yko_bf_destroy(yko_bf_t *b) { if (b) { free(b->b); free(b); } }

int yko_bf_insert(yko_bf_t *b, uint64_t hash)                /* bbf.c:25-42 */
{
	int x = b->n_shift - YKO_BLK_SHIFT;                      /* log2(#blocks) */
	uint64_t blk = hash & ((1ULL << x) - 1);
	int z = (int)(hash >> x & 511), step = (int)(hash >> b->n_shift & 511);
	uint8_t *p = b->b + (blk << 6);
	int i, hits = 0;
	if ((step & 31) == 0) step = (step + 1) & 511;
	for (i = 0; i < b->n_hashes; ++i, z = (z + step) & 511) {
		uint8_t bit = (uint8_t)(1u << (z & 7));
		hits += (p[z >> 3] & bit) != 0;
		p[z >> 3] |= bit;
	}
	return hits;
}

/* ------------------------------------------------------------------ counting table (htab.c) */
yko_ch_t *yko_ch_init(int k, int pre, int n_hash, int n_shift) /* htab.c:13-29 */
{
	yko_ch_t *h;
	int i, P;
	if (pre < YKO_COUNTER_BITS) return 0;
	h = (yko_ch_t*)calloc(1, sizeof(*h));
	h->k = k; h->pre = pre; P = 1 << pre;
	h->h = (yko_ch1_t*)calloc(P, sizeof(yko_ch1_t));
	for (i = 0; i < P; ++i) h->h[i].h = set_new();
	if (n_hash > 0 && n_shift > pre) {
		h->n_hash = n_hash; h->n_shift = n_shift;
		for (i = 0; i < P; ++i) h->h[i].b = yko_bf_init(n_shift - pre, n_hash);
	}
	return h;
}

void yko_ch_destroy_bf(yko_ch_t *h)                          /* htab.c:31-39 */
{
	int i;
	for (i = 0; i < 1 << h->pre; ++i) { yko_bf_destroy(h->h[i].b); h->h[i].b = 0; }
}

void yko_ch_destroy(yko_ch_t *h)                             /* htab.c:41-49 */
{
	int i;
	if (!h) return;
	yko_ch_destroy_bf(h);
	for (i = 0; i < 1 << h->pre; ++i) set_free(h->h[i].h);
	free(h->h); free(h);
}

static inline void bump(uint64_t *slot) { if ((*slot & YKO_MAX_COUNT) < YKO_MAX_COUNT) ++*slot; }

int yko_ch_insert_list(yko_ch_t *h, int create_new, int n, const uint64_t *a) /* htab.c:51-78 */
{
	uint64_t pm = (1ULL << h->pre) - 1;
	yko_ch1_t *g;
	int j, n_new = 0;
	if (n == 0) return 0;
	g = &h->h[a[0] & pm];
	for (j = 0; j < n; ++j) {
		uint64_t x = a[j] >> h->pre;
		if ((a[j] & pm) != (a[0] & pm)) continue;            /* foreign prefix: silently skipped */
		if (create_new) {
			int absent, pass = 1;
			if (g->b) pass = (yko_bf_insert(g->b, x) == h->n_hash);
			if (pass) {
				uint32_t i = yko_set_put(g->h, x << YKO_COUNTER_BITS, &absent);
				if (absent) ++n_new;
				bump(&g->h->keys[i]);
			}
		} else {
			uint32_t i = yko_set_get(g->h, x << YKO_COUNTER_BITS);
			if (i != yko_set_capacity(g->h)) bump(&g->h->keys[i]);
		}
	}
	return n_new;
}

int yko_ch_get(const yko_ch_t *h, uint64_t x)                /* htab.c:93-100 */
{
	const yko_set_t *g = h->h[x & ((1ULL << h->pre) - 1)].h;
	uint32_t i = yko_set_get(g, x >> h->pre << YKO_COUNTER_BITS);
	return i == yko_set_capacity(g) ? -1 : (int)(g->keys[i] & YKO_MAX_COUNT);
}

int yko_ch_inc(yko_ch_t *h, uint64_t x)                      /* htab.c:80-91 */
{
	yko_set_t *g = h->h[x & ((1ULL << h->pre) - 1)].h;
	uint32_t i = yko_set_get(g, x >> h->pre << YKO_COUNTER_BITS);
	if (i == yko_set_capacity(g)) return -1;
	bump(&g->keys[i]);
	return (int)(g->keys[i] & YKO_MAX_COUNT);
}

void yko_ch_clear(yko_ch_t *h)                               /* htab.c:116-130 */
{
	int p;
	for (p = 0; p < 1 << h->pre; ++p) {
		yko_set_t *g = h->h[p].h;
		uint32_t i, n = yko_set_capacity(g);
		for (i = 0; i < n; ++i)
			if (USED(g->used, i)) g->keys[i] &= ~(uint64_t)YKO_MAX_COUNT;
	}
}

/* rebuild sub-table g into a fresh set pre-sized to the OLD size, visiting old slots in
 * ascending order and keeping what `keep` accepts (htab.c:180-197 / 287-347 share this shape) */
typedef int (*keep_fn)(uint64_t key, const void *aux);
static yko_set_t *rebuild(yko_set_t *g, keep_fn keep, const void *aux)
{
	yko_set_t *f = set_new();
	uint32_t i, n = yko_set_capacity(g);
	int absent;
	yko_set_resize(f, g->count);
	for (i = 0; i < n; ++i)
		if (USED(g->used, i) && keep(g->keys[i], aux)) yko_set_put(f, g->keys[i], &absent);
	set_free(g);
	return f;
}

static void recount_tot(yko_ch_t *h)
{
	int p;
	for (p = 0, h->tot = 0; p < 1 << h->pre; ++p) h->tot += h->h[p].h->count;
}

static int keep_range(uint64_t key, const void *aux)
{
	const int *r = (const int*)aux;
	int c = (int)(key & YKO_MAX_COUNT);
	return c >= r[0] && c <= r[1];
}

void yko_ch_shrink(yko_ch_t *h, int min, int max)            /* htab.c:180-208 */
{
	int p, r[2];
	r[0] = min; r[1] = (max >= min && max <= YKO_MAX_COUNT) ? max : YKO_MAX_COUNT;
	for (p = 0; p < 1 << h->pre; ++p) h->h[p].h = rebuild(h->h[p].h, keep_range, r);
	recount_tot(h);
}

void yko_ch_tighten(yko_ch_t *h)                             /* htab.c:102-110 */
{
	int p;
	for (p = 0; p < 1 << h->pre; ++p) {
		yko_set_t *g = h->h[p].h;
		if (g->count * 3 < yko_set_capacity(g)) yko_set_resize(g, g->count * 3);
	}
}

void yko_ch_setcnt(yko_ch_t *h, int cnt)                     /* htab.c:219-235 */
{
	int p;
	for (p = 0; p < 1 << h->pre; ++p) {
		yko_set_t *g = h->h[p].h;
		uint32_t i, n = yko_set_capacity(g);
		for (i = 0; i < n; ++i)
			if (USED(g->used, i)) g->keys[i] = (g->keys[i] & ~(uint64_t)YKO_MAX_COUNT) | (uint64_t)cnt;
	}
}

void yko_ch_hist(const yko_ch_t *h, int64_t cnt[1 << YKO_COUNTER_BITS]) /* htab.c:145-169 */
{
	int p;
	memset(cnt, 0, sizeof(int64_t) << YKO_COUNTER_BITS);
	for (p = 0; p < 1 << h->pre; ++p) {
		const yko_set_t *g = h->h[p].h;
		uint32_t i, n = yko_set_capacity(g);
		for (i = 0; i < n; ++i)
			if (USED(g->used, i)) ++cnt[g->keys[i] & YKO_MAX_COUNT];
	}
}

void yko_ch_merge(yko_ch_t *h0, yko_ch_t *h1, int min, int max, int pre_resize) /* htab.c:246-285 */
{
	int p, hi = (max >= min && max <= YKO_MAX_COUNT) ? max : YKO_MAX_COUNT;
	for (p = 0; p < 1 << h0->pre; ++p) {
		yko_set_t *g0 = h0->h[p].h, *g1 = h1->h[p].h;
		uint32_t i, n1 = yko_set_capacity(g1);
		if (pre_resize) {
			uint32_t want = (g0->count + g1->count) * 4 / 3 + 1;
			if (want > yko_set_capacity(g0)) yko_set_resize(g0, want);
		}
		for (i = 0; i < n1; ++i) {
			int c, absent;
			if (!USED(g1->used, i)) continue;
			c = (int)(g1->keys[i] & YKO_MAX_COUNT);
			if (c >= min && c <= hi) {
				uint32_t l = yko_set_put(g0, g1->keys[i] & ~(uint64_t)YKO_MAX_COUNT, &absent);
				bump(&g0->keys[l]);
			}
		}
		set_free(g1);
		yko_bf_destroy(h1->h[p].b);
	}
	free(h1->h); free(h1);
	recount_tot(h0);
}

static int keep_absent_in(uint64_t key, const void *aux)
{
	const yko_set_t *g1 = (const yko_set_t*)aux;
	return yko_set_get(g1, key) == yko_set_capacity(g1);
}
static int keep_present_in(uint64_t key, const void *aux) { return !keep_absent_in(key, aux); }

void yko_ch_subtract(yko_ch_t *h0, const yko_ch_t *h1)       /* htab.c:287-316 */
{
	int p;
	for (p = 0; p < 1 << h0->pre; ++p) h0->h[p].h = rebuild(h0->h[p].h, keep_absent_in, h1->h[p].h);
	recount_tot(h0);
}

void yko_ch_isec(yko_ch_t *h0, const yko_ch_t *h1)           /* htab.c:318-347 */
{
	int p;
	for (p = 0; p < 1 << h0->pre; ++p) h0->h[p].h = rebuild(h0->h[p].h, keep_present_in, h1->h[p].h);
	recount_tot(h0);
}

yko_knt_t *yko_ch_getseq(const yko_ch_t *h, int w, uint32_t *n) /* htab.c:353-367 */
{
	const yko_set_t *g = h->h[w].h;
	uint64_t mask = (1ULL << h->k * 2) - 1;
	uint32_t i, j = 0, cap = yko_set_capacity(g);
	yko_knt_t *a = (yko_knt_t*)calloc(g->count ? g->count : 1, sizeof(*a));
	*n = g->count;
	for (i = 0; i < cap; ++i)
		if (USED(g->used, i)) {
			a[j].x = yko_hash64_inv(KEY_ID(g->keys[i]) << h->pre | (uint64_t)w, mask);
			a[j++].c = (int)(g->keys[i] & YKO_MAX_COUNT);
		}
	return a;
}

void yko_ch_subtable(const yko_ch_t *h, int i, uint32_t *cap, uint32_t *size)
{
	*cap = yko_set_capacity(h->h[i].h); *size = h->h[i].h->count;
}

/* ------------------------------------------------------------------ .yak I/O (htab.c:373-481)
 * "YAK\2" | u32 k | u32 pre | u32 10 | per sub-table: u32 capacity | u32 size | size x u64
 * keys in ascending slot order; native little-endian. */
size_t yko_ch_dump_mem(const yko_ch_t *h, uint8_t **out)
{
	int p, P = 1 << h->pre;
	size_t sz = 16 + (size_t)8 * P, off;
	uint8_t *o;
	uint32_t t[3];
	for (p = 0; p < P; ++p) sz += (size_t)8 * h->h[p].h->count;
	o = (uint8_t*)malloc(sz);
	memcpy(o, "YAK\2", 4);
	t[0] = h->k; t[1] = h->pre; t[2] = YKO_COUNTER_BITS;
	memcpy(o + 4, t, 12);
	off = 16;
	for (p = 0; p < P; ++p) {
		const yko_set_t *g = h->h[p].h;
		uint32_t i, cap = yko_set_capacity(g);
		t[0] = cap; t[1] = g->count;
		memcpy(o + off, t, 8); off += 8;
		for (i = 0; i < cap; ++i)
			if (USED(g->used, i)) { memcpy(o + off, &g->keys[i], 8); off += 8; }
	}
	*out = o;
	return sz;
}

int yko_ch_dump_range(const yko_ch_t *h, const char *fn, int lo, int hi)   /* htab.c:373-394, sub-tables [lo, hi) only; header iff lo == 0 */
{
	int p;
	uint32_t t[3];
	FILE *fp = strcmp(fn, "-") ? fopen(fn, "wb") : stdout;
	if (!fp) return -1;
	if (lo == 0) {
		fwrite("YAK\2", 1, 4, fp);
		t[0] = h->k; t[1] = h->pre; t[2] = YKO_COUNTER_BITS;
		fwrite(t, 4, 3, fp);
	}
	for (p = lo; p < hi; ++p) {
		const yko_set_t *g = h->h[p].h;
		uint32_t i, cap = yko_set_capacity(g);
		t[0] = cap; t[1] = g->count;
		fwrite(t, 4, 2, fp);
		for (i = 0; i < cap; ++i)
			if (USED(g->used, i)) fwrite(&g->keys[i], 8, 1, fp);
	}
	if (fp != stdout) fclose(fp);
	return 0;
}

int yko_ch_dump(const yko_ch_t *h, const char *fn)
{
	uint8_t *buf;
	size_t sz = yko_ch_dump_mem(h, &buf);
	FILE *fp = strcmp(fn, "-") ? fopen(fn, "wb") : stdout;
	if (!fp) { free(buf); return -1; }
	fwrite(buf, 1, sz, fp);
	if (fp != stdout) fclose(fp);
	free(buf);
	return 0;
}

/* htab.c:396-476.  mode 1 = all keys as stored; 2 / 3 = trio binning flags (counts >= mid_cnt -> 2,
 * >= min_cnt -> 1, below -> dropped; shifted left by 2 in mode 3); 4 / 5 / 6 = sex chromosome flags
 * 1 / 2 / 4.  In the flag modes the low 10 bits of a stored key are a flag set: a key already in the
 * table gets the new flag ORed in.  Modes 3, 5, 6 need an existing table. */
yko_ch_t *yko_ch_restore_core(yko_ch_t *ch0, const char *fn, int mode, int min_cnt, int mid_cnt)
{
	FILE *fp;
	char magic[4];
	uint32_t t[3];
	yko_ch_t *h;
	int p;
	const uint64_t mask = (1ULL << YKO_COUNTER_BITS) - 1;
	if (mode < 1 || mode > 6) return 0;
	if (ch0 == 0 && (mode == 3 || mode == 5 || mode == 6)) return 0;
	fp = fopen(fn, "rb");
	if (!fp) return 0;
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "YAK\2", 4) != 0) { fclose(fp); return 0; }
	if (fread(t, 4, 3, fp) != 3 || t[2] != YKO_COUNTER_BITS) { fclose(fp); return 0; }
	h = ch0 ? ch0 : yko_ch_init((int)t[0], (int)t[1], 0, 0);
	for (p = 0; p < 1 << h->pre; ++p) {
		uint32_t j, u[2];
		if (fread(u, 4, 2, fp) != 2) break;
		yko_set_resize(h->h[p].h, u[0]);                      /* to the saved capacity (htab.c:441) */
		for (j = 0; j < u[1]; ++j) {
			uint64_t key; int absent, x = -1;
			uint32_t k;
			if (fread(&key, 8, 1, fp) != 1) break;
			if (mode == 1) { yko_set_put(h->h[p].h, key, &absent); continue; }   /* file order (htab.c:447) */
			if (mode == 2 || mode == 3) {
				const int cnt = (int)(key & mask), shift = mode == 2 ? 0 : 2;
				if (cnt >= mid_cnt) x = 2 << shift;
				else if (cnt >= min_cnt) x = 1 << shift;
			} else x = 1 << (mode - 4);
			if (x < 0) continue;
			key = (key & ~mask) | (uint64_t)x;
			k = yko_set_put(h->h[p].h, key, &absent);
			if (!absent) h->h[p].h->keys[k] |= (uint64_t)x;
		}
	}
	fclose(fp);
	if (!ch0) h->tot = 0;          /* the reference leaves tot = 0 after restore: it is not serialised */
	return h;
}

yko_ch_t *yko_ch_restore(const char *fn) { return yko_ch_restore_core(0, fn, 1, 0, 0); }   /* htab.c:478 */
