/*
 * yko.h -- ORACLE (test infrastructure only, never shipped, never measured as the product).
 *
 * A from-scratch, single-threaded CPU restatement of the k-mer counting path of lh3/yak
 * (reference: count.c, htab.c, khashl.h, bbf.c, yak-priv.h, misc.c).  Every function cites
 * the reference file:line whose behaviour it follows.  Parity status: PINNED -- the oracle is
 * checked (tests/test_oracle_vs_ref.py, oracle/Makefile target `_ref`) byte-for-byte against
 * the reference itself compiled from /root/reference into oracle/_ref/, and against the golden
 * .yak fixtures under tests/golden/ that the reference produced.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 */
#ifndef YKO_H
#define YKO_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YKO_COUNTER_BITS 10            /* yak.h:9  */
#define YKO_MAX_COUNT    1023          /* yak.h:11 */
#define YKO_BLK_SHIFT    9             /* yak.h:13 : 512-bit bloom blocks */

/* ---- options: same fields/defaults as yak_copt_t (yak.h:25-31, misc.c:23-32) ---- */
typedef struct {
	int32_t bf_shift, bf_n_hash, k, pre, n_thread;
	int64_t chunk_size;
} yko_copt_t;
void yko_copt_init(yko_copt_t *o);

/* ---- hashing (yak-priv.h:11-68) and base encoding (misc.c:4-21) ---- */
extern const unsigned char yko_nt4[256];
uint64_t yko_hash64(uint64_t key, uint64_t mask);
uint64_t yko_hash64_64(uint64_t key);
uint64_t yko_hash_long(const uint64_t x[4]);
uint64_t yko_hash64_inv(uint64_t key, uint64_t mask);
uint32_t yko_h2b(uint32_t hash, uint32_t bits);      /* khashl.h:98 */

/* ---- slot set with khashl's exact layout behaviour (khashl.h:104-221) ---- */
typedef struct {
	uint32_t bits, count;
	uint32_t *used;      /* 1 bit per slot, >= 1 word */
	uint64_t *keys;      /* NULL until the first resize */
} yko_set_t;
uint32_t yko_set_capacity(const yko_set_t *s);
uint32_t yko_set_get(const yko_set_t *s, uint64_t key);          /* returns capacity if absent */
uint32_t yko_set_put(yko_set_t *s, uint64_t key, int *absent);
int      yko_set_resize(yko_set_t *s, uint32_t want);

/* ---- blocked bloom filter (bbf.c) ---- */
typedef struct { int n_shift, n_hashes; uint8_t *b; } yko_bf_t;
yko_bf_t *yko_bf_init(int n_shift, int n_hashes);
void      yko_bf_destroy(yko_bf_t *b);
int       yko_bf_insert(yko_bf_t *b, uint64_t hash);

/* ---- counting table = 1<<pre sub-tables (htab.c) ---- */
typedef struct { yko_set_t *h; yko_bf_t *b; } yko_ch1_t;
typedef struct {
	int k, pre, n_hash, n_shift;
	uint64_t tot;
	yko_ch1_t *h;
} yko_ch_t;
typedef struct { uint64_t x; int c; } yko_knt_t;

yko_ch_t *yko_ch_init(int k, int pre, int n_hash, int n_shift);
void      yko_ch_destroy(yko_ch_t *h);
void      yko_ch_destroy_bf(yko_ch_t *h);
int       yko_ch_insert_list(yko_ch_t *h, int create_new, int n, const uint64_t *a);
int       yko_ch_get(const yko_ch_t *h, uint64_t x);
int       yko_ch_inc(yko_ch_t *h, uint64_t x);
void      yko_ch_clear(yko_ch_t *h);
void      yko_ch_shrink(yko_ch_t *h, int min, int max);
void      yko_ch_tighten(yko_ch_t *h);
void      yko_ch_setcnt(yko_ch_t *h, int cnt);
void      yko_ch_hist(const yko_ch_t *h, int64_t cnt[1 << YKO_COUNTER_BITS]);
void      yko_ch_merge(yko_ch_t *h0, yko_ch_t *h1, int min, int max, int pre_resize);
void      yko_ch_subtract(yko_ch_t *h0, const yko_ch_t *h1);
void      yko_ch_isec(yko_ch_t *h0, const yko_ch_t *h1);
yko_knt_t *yko_ch_getseq(const yko_ch_t *h, int w, uint32_t *n);
int       yko_ch_dump(const yko_ch_t *h, const char *fn);
yko_ch_t *yko_ch_restore(const char *fn);
/* htab.c:396-476; mode as YAK_LOAD_* (yak.h:16-21); min_cnt / mid_cnt only in modes 2, 3 */
yko_ch_t *yko_ch_restore_core(yko_ch_t *ch0, const char *fn, int mode, int min_cnt, int mid_cnt);
/* Prefix-range mode (sizes whose tables do not fit the host at once: the 5 Gb assembly of BASELINE configs[3] needs ~80 GB): with a range set,
 * the counting drivers below insert only the k-mers of sub-tables [lo, hi) -- every sub-table is a function of its own k-mers alone (htab.c:51-78,
 * count.c:133 kt_for over prefixes), so the sub-tables of the range come out exactly as in a full run -- and yko_ch_dump_range writes the bytes
 * of those sub-tables (behind the 16-byte header when lo == 0): the files of consecutive ranges, concatenated, are the full .yak file */
void      yko_set_prefix_range(int lo, int hi);               /* lo < 0: all (the default) */
/* want[p] != 0 for the sub-tables to count, one byte per prefix (NULL: all): yko_extract then lists the k-mers of those alone (oracle/yko_synth.c) */
void      yko_set_prefix_mask(const unsigned char *want);
int       yko_ch_dump_range(const yko_ch_t *h, const char *fn, int lo, int hi);
/* serialise to memory in .yak format; caller frees *out */
size_t    yko_ch_dump_mem(const yko_ch_t *h, uint8_t **out);
/* sub-table introspection for tests */
void      yko_ch_subtable(const yko_ch_t *h, int i, uint32_t *cap, uint32_t *size);

/* ---- counting driver (count.c) ---- */
/* k-mers of one sequence appended, in position order, to per-prefix lists (count.c:28-60) */
typedef struct { int64_t n, m; uint64_t *a; } yko_kbuf_t;
void yko_extract(yko_kbuf_t *buf, int k, int pre, int64_t len, const char *seq);

int64_t yko_extract_pos(int k, const uint8_t *bases, int64_t n, uint64_t *out_hash, uint32_t *out_t);

yko_ch_t *yko_count_file(const char *fn, const yko_copt_t *opt, yko_ch_t *h0);
/* same, but the "file" is a memory image of sequences separated by any non-ACGT byte */
yko_ch_t *yko_count_mem(const uint8_t *bases, int64_t n, const yko_copt_t *opt, yko_ch_t *h0);
/* the whole `yak count` protocol of main.c:53-60 on a memory image; returns the final table */
yko_ch_t *yko_count_protocol_mem(const uint8_t *b1, int64_t n1, const uint8_t *b2, int64_t n2,
                                 const yko_copt_t *opt);
yko_ch_t *yko_count_protocol_file(const char *fn1, const char *fn2, const yko_copt_t *opt);

/* test helper: the sequences of fn that count.c:93-96 would process, each followed by '\n'; free(*out) */
int64_t   yko_read_image(const char *fn, int min_len, char **out);

/* ---- `yak qv` counting step (qv.c:34-135); the statistics of yak_qv_solve are host math outside the path ---- */
typedef struct {                                              /* yak.h:33-40 */
	int32_t print_each, print_err_kmer;
	int32_t min_len;
	int32_t n_threads;
	double min_frac;
	double fpr;
	int64_t chunk_size;
} yko_qopt_t;
void yko_qopt_init(yko_qopt_t *o);
/* cnt: 1 << YKO_COUNTER_BITS bins; `out` (FILE*, may be NULL) receives the EK / SQ lines */
int yko_qv(const yko_qopt_t *opt, const char *fn, const yko_ch_t *ch, int64_t *cnt, void *out);

#ifdef __cplusplus
}
#endif
#endif
