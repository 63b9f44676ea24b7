/*
 * yak.h -- drop-in C boundary of the MI355X k-mer counting engine.
 *
 * This header declares, with identical names, argument meaning, struct layouts and error
 * behaviour, the part of lh3/yak's public C API (reference yak.h:25-107) that the `yak count`
 * hot path is made of.  A program written against the reference header links against
 * libyak_amd.so unchanged; the counting itself runs in hand-written gfx950 HIP kernels.
 * Each declaration cites the reference definition it replaces.
 *
 * Constants that are part of the .yak file format are fixed exactly as in the reference
 * (yak.h:7-22): 10 counter bits, 512-bit bloom blocks, magic "YAK\2".
 */
#ifndef YAK_H
#define YAK_H

#include <stdint.h>

#define YAKS_VERSION     "0.1-amd-r1"

#define YAK_MAX_KMER     31
#define YAK_COUNTER_BITS 10
#define YAK_N_COUNTS     (1<<YAK_COUNTER_BITS)

#define YAK_LOAD_ALL       1      /* reference yak.h:16-21 */
#define YAK_LOAD_TRIOBIN1  2
#define YAK_LOAD_TRIOBIN2  3
#define YAK_LOAD_SEXCHR1   4
#define YAK_LOAD_SEXCHR2   5
#define YAK_LOAD_SEXCHR3   6
#define YAK_MAX_COUNT    ((1<<YAK_COUNTER_BITS)-1)
#define YAK_BLK_SHIFT    9
#define YAK_BLK_MASK     ((1<<(YAK_BLK_SHIFT)) - 1)
#define YAK_MAGIC        "YAK\2"

#ifdef __cplusplus
extern "C" {
#endif

/* counting options; caller-allocated, fixed layout (reference yak.h:25-31) */
typedef struct {
	int32_t bf_shift, bf_n_hash;
	int32_t k;
	int32_t pre;
	int32_t n_thread;
	int64_t chunk_size;
} yak_copt_t;

/* blocked bloom filter of one sub-table (reference yak.h:49-52).  In this implementation the
 * authoritative bits live in HBM; `b` is a host mirror that is brought up to date on demand. */
typedef struct {                    /* reference yak.h:33-40 */
	int32_t print_each, print_err_kmer;
	int32_t min_len;
	int32_t n_threads;
	double min_frac;
	double fpr;
	int64_t chunk_size;
} yak_qopt_t;

typedef struct {                    /* reference yak.h:42-47 */
	int64_t tot;
	double qv_raw, qv, cov, err;
	double fpr_lower, fpr_upper;
	double adj_cnt[1<<YAK_COUNTER_BITS];
} yak_qstat_t;

typedef struct {
	int n_shift, n_hashes;
	uint8_t *b;
} yak_bf_t;

/* one sub-table: slot array in khashl's exact layout (opaque to callers, reference htab.c:11) */
struct yak_ht_t;
typedef struct {
	struct yak_ht_t *h;
	yak_bf_t *b;
} yak_ch1_t;

/* the counting table: 1<<pre sub-tables selected by the low `pre` bits of the hashed k-mer
 * (reference yak.h:61-65).  Callers read k, pre and tot directly. */
typedef struct {
	int k, pre, n_hash, n_shift;
	uint64_t tot;
	yak_ch1_t *h;
} yak_ch_t;

typedef struct {
	uint64_t x;
	int c;
} yak_knt_t;

extern int yak_verbose;                              /* reference sys.c:5  */
extern unsigned char seq_nt4_table[256];             /* reference misc.c:4 */

void yak_copt_init(yak_copt_t *opt);                 /* reference misc.c:23 */

/* reference bbf.c:5-42 */
yak_bf_t *yak_bf_init(int n_shift, int n_hashes);
void yak_bf_destroy(yak_bf_t *b);
int yak_bf_insert(yak_bf_t *b, uint64_t hash);

/* reference htab.c:13-49 */
yak_ch_t *yak_ch_init(int k, int pre, int n_hash, int n_shift);
void yak_ch_destroy(yak_ch_t *h);
void yak_ch_destroy_bf(yak_ch_t *h);
/* reference htab.c:51-78: all a[j] must share one prefix; returns the number of new keys */
int yak_ch_insert_list(yak_ch_t *h, int create_new, int n, const uint64_t *a);
int yak_ch_get(const yak_ch_t *h, uint64_t x);       /* reference htab.c:93  */
int yak_ch_inc(yak_ch_t *h, uint64_t x);             /* reference htab.c:80  */
yak_knt_t *yak_ch_getseq(const yak_ch_t *h, int w, uint32_t *n);   /* reference htab.c:353 */

void yak_ch_clear(yak_ch_t *h, int n_thread);                     /* reference htab.c:127 */
void yak_ch_hist(const yak_ch_t *h, int64_t cnt[YAK_N_COUNTS], int n_thread); /* htab.c:156 */
void yak_ch_setcnt(yak_ch_t *h, int cnt, int n_thread);            /* htab.c:225: set every stored count */
void yak_ch_tighten(yak_ch_t *h);                                 /* htab.c:102 */
void yak_ch_merge(yak_ch_t *h0, yak_ch_t *h1, int min, int max, int n_thread, int pre_resize); /* htab.c:272; destroys h1 */
void yak_ch_subtract(yak_ch_t *h0, const yak_ch_t *h1, int n_thread); /* htab.c:309 */
void yak_ch_isec(yak_ch_t *h0, const yak_ch_t *h1, int n_thread);     /* htab.c:340 */
void yak_ch_shrink(yak_ch_t *h, int min, int max, int n_thread);  /* reference htab.c:199 */

int yak_ch_dump(const yak_ch_t *h, const char *fn);               /* reference htab.c:373 */
yak_ch_t *yak_ch_restore(const char *fn);                         /* reference htab.c:478 */
/* reference htab.c:396: mode = YAK_LOAD_*; the two trio-binning modes take (int min_cnt, int mid_cnt) */
yak_ch_t *yak_ch_restore_core(yak_ch_t *ch0, const char *fn, int mode, ...);

/* reference count.c:147: count the k-mers of a FASTA/FASTQ(.gz) file ("-"/NULL = stdin).
 * h0 == NULL: create a table (bloom-gated if opt->bf_shift > pre) and return it;
 * h0 != NULL: only increment k-mers already present in h0 (k and pre must match) and return h0.
 * NULL when the file cannot be opened or no usable GPU is present (a message goes to stderr). */
yak_ch_t *yak_count(const char *fn, const yak_copt_t *opt, yak_ch_t *h0);

void yak_recount(const char *fn, yak_ch_t *h);                  /* reference count.c:168: clear + count existing k-mers of fn */

/* `yak qv` counting step (reference qv.c:34-135, yak.h:33-40,105-106): per sequence of fn with at
 * least min_len bases, the table count of every k-mer; sequences whose fraction of present k-mers is
 * >= min_frac add their counts to cnt[YAK_N_COUNTS].  Runs on the device-resident table. */
void yak_qopt_init(yak_qopt_t *opt);
int yak_qv_solve(const int64_t *hist, const int64_t *cnt, int kmer, double fpr, yak_qstat_t *qs); /* qv.c:146: host arithmetic */
void yak_qv(const yak_qopt_t *opt, const char *fn, const yak_ch_t *ch, int64_t *cnt);

#ifdef __cplusplus
}
#endif
#endif
