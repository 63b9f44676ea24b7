/*
 * yk_device.h -- shared declarations between the HIP kernels (kernels.hip) and the host engine
 * (engine.cpp).  Internal; the public surface is include/yak.h + include/yak_amd.h.
 */
#ifndef YK_DEVICE_H
#define YK_DEVICE_H

#include <stdint.h>
#include <hip/hip_runtime.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef ulonglong2 Rec;                  /* one k-mer instance: .x = yak_hash64 value, .y = stream position (32 bits used) */

#define YK_EMPTY   0xFFFFFFFFFFFFFFFFull     /* unclaimed slot (accumulator and table image) */
#define YK_TINF    0xFFFFFFFFFFFFFFFFull     /* "never" */
#define YK_NOCAP   0xFFFFFFFFu               /* sub-table without a slot array (kh_capacity == 0) */
#define YK_FLAG_FP 1u                        /* first occurrence passed the bloom gate */

/* accumulator slot: one per distinct hashed k-mer seen by the current pass (32 B, one sector) */
struct __attribute__((aligned(32))) AccSlot {
	u64 key;        /* full yak_hash64 value, YK_EMPTY if free */
	u64 t1;         /* stream position of the first occurrence  (atomicMin) */
	u64 t2;         /* stream position of the second occurrence (atomicMin of the losers) */
	u32 cnt;        /* occurrences (exact up to 2^32-1, clamped to 1023 on output) */
	u32 flags;
};

struct AccTab {
	AccSlot *s;
	u64 mask;       /* capacity - 1 */
	int bits;       /* log2 capacity, >= pre + 1 */
	int pre;
};

/* device-resident image of the 1<<pre khashl tables (exact layout) */
struct ImgView {
	const u32 *bits;    /* [P] log2 capacity or YK_NOCAP */
	const u64 *off;     /* [P] first slot in the arena (multiple of 32) */
	u64 *keys;          /* arena: (hash>>pre)<<10 | count, YK_EMPTY in unused slots */
	u32 *used;          /* arena bitmap, bit i <-> arena slot i */
	u32 *delta;         /* per-slot pending count increments of the running pass (may be 0) */
	int pre;
	int k;
};

/* bloom geometry (reference bbf.c): per sub-table 1<<nb bits in 512-bit blocks */
struct BloomView {
	u32 *bits32;        /* P << (nb-5) words */
	int nb;             /* log2 bits per sub-table (= bf_shift - pre) */
	int n_hash;
	int mw;             /* u64 words of a "missing probes" mask */
};

/* parameters of one replay task (one sub-table) */
struct ReplayTask {
	u32 old_bits, old_count;
	u64 old_off;        /* slot offset in the old arena */
	u64 new_off;        /* slot offset in the new arena (multiple of 32) */
	u64 rec_off;        /* first record of this sub-table's sorted new keys */
	u32 m;              /* number of new keys */
	u32 init_bits;      /* pre-sized empty table (shrink), YK_NOCAP otherwise */
	u32 cap_max_bits;   /* room reserved in the new arena */
	u32 dbg;
};

#ifdef __cplusplus
extern "C" {
#endif

/* ---- launch wrappers implemented in kernels.hip (all asynchronous on `st`) ---- */
void yk_launch_extract(const uint8_t *bases, int64_t pos0, int64_t n, int64_t t_sub, int k, int pre, int plo, int phi,
                       u64 *out_hash, u32 *out_t, u64 *cursor, hipStream_t st);
void yk_launch_pack(const uint8_t *a, int64_t n, u32 *codes, u32 *valid, hipStream_t st);
void yk_launch_xpart(const uint8_t *bases, int64_t pos0, int64_t n, int64_t t_sub, int k, int pre, int plo, int phi,
                     int nb_bits, u32 *rows, u64 *partial, u64 *bstart, Rec *out, int hash_only, hipStream_t st, const u32 *valid = 0);   /* valid != 0: `bases` is the packed 2-bit image (yakamd_feed_packed_dev) */   /* hash_only: 0 = Rec, 1 = bare hashes, 2 = tagged 8-byte records */
void yk_launch_rpart(const u64 *in_hash, const u32 *in_t, int64_t n, int pre, int plo, int phi,
                     int nb_bits, u32 *rows, u64 *partial, u64 *bstart, Rec *out, hipStream_t st);
int yk_xpart_blocks(int64_t n_pos);
int yk_part_groups(void);
int yk_rpart_blocks(int64_t n_rec);
int yk_bad_hash_seen(hipStream_t st);
void yk_par_counters(u32 *ok, u32 *fail);
void yk_replay_prof(u64 *out8);
void yk_launch_acc_init(AccSlot *s, u64 n, hipStream_t st);
void yk_launch_acc_insert(const Rec *rec, int64_t n, u64 t0, AccTab tab, ImgView img,
                          int img_nonempty, int bloom_mode, u64 *newlist, u64 *counters, hipStream_t st);
void yk_launch_acc_rehash(AccTab oldt, AccTab newt, hipStream_t st);
void yk_launch_img_count(const Rec *rec, int64_t n, ImgView img, hipStream_t st);
size_t yk_img_count_lds_bytes(u32 cap, u32 count);
void yk_launch_lookup(const uint8_t *bases, int64_t n, int k, ImgView img, unsigned short *out, hipStream_t st);
void yk_launch_qv_reduce(const unsigned short *t, const u64 *roff, const u32 *rlen, int64_t n_reads, int min_len, double min_frac,
                         u32 *tot_out, u32 *non0_out, u64 *hist, hipStream_t st);
int yk_launch_img_count_lds(const void *rec, int hash_only, const u64 *bstart, ImgView img, int plo, int phi, size_t lds, u64 *compact, u32 stride, hipStream_t st);
void yk_launch_img_count_h(const u64 *hash, int64_t n, ImgView img, hipStream_t st);
void yk_launch_img_inc(ImgView img, u64 hash, u64 *out2, hipStream_t st);
void yk_launch_img_fold(ImgView img, u64 n_slots, hipStream_t st);
void yk_launch_img_hist(ImgView img, u64 n_slots, u64 *hist, hipStream_t st);
void yk_launch_img_setcnt(ImgView img, u64 n_slots, u32 cnt, hipStream_t st);
void yk_launch_img_clear(ImgView img, u64 n_slots, hipStream_t st);
void yk_launch_lastput(const Rec *rec, int64_t n, u64 t0, u64 t_from, AccTab tab, ImgView img,
                       int img_nonempty, int bloom_mode, const u32 *only_missing, u64 *lp_batch, hipStream_t st);
void yk_launch_lastput_merge(u64 *lastput, const u64 *lp_batch, u32 *missing, u32 *n_missing, int P, int plo, int phi, hipStream_t st);

void yk_launch_bf_test(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, u64 *miss, hipStream_t st);
void yk_launch_bf_set(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
                      u32 *multi, int multi_bits, u64 *counters, hipStream_t st);
void yk_launch_bf_check(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
                        const u32 *multi, int multi_bits, u64 *cand, u64 *counters, hipStream_t st);
void yk_launch_bf_mapfill(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
                          const u32 *multi, int multi_bits, u64 *map, int map_bits, hipStream_t st);
void yk_launch_bf_resolve(AccTab tab, const u64 *newlist, const u64 *cand, u64 n_cand, BloomView bf,
                          const u64 *miss, const u64 *map, int map_bits, hipStream_t st);

void yk_launch_select_count(AccTab tab, int bloom_mode, int P, u32 *seg_cnt, hipStream_t st);
void yk_launch_select_scatter(AccTab tab, int bloom_mode, int P, const u64 *seg_off, u32 *seg_cur,
                              u64 *rec_kc, u64 *rec_t, hipStream_t st);
void yk_launch_seg_sort_pass(const u64 *seg_off, int P, const u64 *src_kc, const u64 *src_t,
                             u64 *dst_kc, u64 *dst_t, int shift, hipStream_t st);
void yk_launch_replay(const ReplayTask *tasks, int n_tasks, int n_threads, const u64 *old_keys, const u32 *old_used,
                      u64 *new_keys, u32 *new_used, u32 *scr_used, u32 *scr_owner, u64 *scr_par,
                      const u64 *rec_kc, const u64 *rec_t, const u64 *lastput,
                      u32 *out_bits, u32 *out_count, u32 lds_words, hipStream_t st);
int yk_shrink_shares(void);
void yk_launch_shrink_count(ImgView img, int P, int cmin, int cmax, int which, ImgView other, u32 *seg_cnt, hipStream_t st, int rng = 0);
void yk_launch_shrink_scatter(ImgView img, int P, int cmin, int cmax, int which, ImgView other, const u64 *seg_off, u64 *rec_kc, hipStream_t st, const u32 *rng_cnt = 0);
struct ResizeTask { u32 old_bits, new_bits, rehash, pad; u64 old_off, new_off; };
void yk_launch_resize(const ResizeTask *tasks, int P, const u64 *old_keys, const u32 *old_used, u64 *new_keys, u32 *new_used, u32 *scr_used, hipStream_t st);
void yk_launch_keys_to_hashes(const u64 *kc, const u64 *seg_off, int P, int pre, u64 *hash, u32 *t, hipStream_t st);
long yk_knob(const char *name, long dflt);                   /* engine.cpp: run-time settings (test switches come through yakamd_test_set only) */
void yk_launch_fill_u64(u64 *p, u64 v, u64 n, hipStream_t st);
void yk_launch_put_u64(const u64 *pos, const u64 *val, u32 n, u64 *out, hipStream_t st);

#ifdef __cplusplus
}
#endif

/* counters[] layout (u64 each) */
enum { YKC_NEW = 0, YKC_INST = 1, YKC_ANYMULTI = 2, YKC_NCAND = 3, YKC_NMARKED = 4, YKC_EXIST = 5, YKC_N = 8 };


/* ---------------- fast path (exclusive-ownership LDS counting) ---------------- */
#define YK_CH2     32768            /* records per level-2 partition chunk */
#define YK_LDS_C   1024             /* slots of the LDS counting table (overflow beyond 768 distinct k-mers) */

struct Chunk2 {                     /* one level-2 partition work item: a run of one level-1 bucket */
	const Rec *rec;
	u64 spare;
	u32 n, bucket;
	u32 tbase;                      /* added to tlo: time relative to the start of the pass */
	u32 pad;                        /* record format of the chunk: 0 = Rec {hash, position}; 1 = tagged 8-byte records (YK_R8_*), tbase = records of this bucket in earlier batches */
	u32 before, after;              /* format 1: records of the same bucket of the same batch lying before / behind the chunk in memory */
};

/* Tagged 8-byte level-1 record (k < 32, 2k - pre <= 52): (hash >> pre) << 12 | toggle << 10 | position inside the 1024-position
 * round of the partitioning workgroup.  Inside a bucket the records of one round are contiguous and the rounds follow each other in
 * stream order (the write combining is stable at round granularity); the toggle bit flips between consecutive rounds that
 * contributed to the bucket, across workgroups too (the histogram sweep counts them), so the level-2 partition recovers every
 * record's rank in the sub-table's stream: the "time" that orders put-calls.  Stream positions are not stored at all. */
#define YK_R8_TAG_BITS 12
#define YK_R8_TOGGLE   (1u << 10)

struct FastParams {
	int pre, k, s2_bits;            /* sub-buckets per sub-table = 1 << s2_bits */
	int bloom_mode, nb, n_hash;     /* nb = log2 bits per sub-table filter */
	int img_nonempty;
	int plo, phi;
	int dbg, bf_virgin;             /* dbg: timing ablations only (YAKAMD_DBG); bf_virgin: filter never written (all zero) */
	int rec8_in, rec8_out, tb;      /* level-2 input is tagged 8-byte records; its output (the counting kernels' input) is 8 bytes: (hash >> pre minus the sub-bucket bits) << tb | rank, tb = 12 + s2_bits */
	int or_mode;                    /* loads from a .yak file (htab.c:436-470) instead of counting: 1 = the low 4 bits of a record's time are a flag, ORed into the key's low bits; 2 = the low 10 bits are the saved count, kept by new keys only */
	/* which bits of x = hash >> pre name the sub-bucket: the top s2_tot bits of x's low `sw` bits (sw = log2 bloom blocks per sub-table with a filter -- a
	 * sub-bucket owns a contiguous range of blocks -- else sw = s2_tot).  A partition sweep routes by s2_bits of them: ((x mod 2^sw) >> ssh) mod 2^s2_bits.
	 * One sweep does them all (ssh = sw - s2_tot, s2_bits = s2_tot: the counting kernels always see this form); beyond 2^13 sub-buckets per sub-table a
	 * first sweep takes the high s2_tot - 11 bits (ssh = sw - s2_bits) and a second one, on the groups of the first, the low 11 (ssh = sw - s2_tot) */
	int sw, ssh, s2_tot;
	int bf_nowb;                    /* k_lc2: the staged filter ranges are not written back (the caller keeps every record of the pass and rebuilds the filter from them if anything ever reads it: k_bf_rebuild) */
	u64 t_pass0;
};

/* outputs of the exclusive-ownership counting kernels: the keys a sub-bucket adds to its sub-table go to the
 * front of the sub-bucket's own record range in kc / T (fragments, gathered by k_lc_compact); per sub-bucket
 * their number, the time (relative to the slice, + 1; 0 = none) of its last put-call, its distinct k-mers */
struct LcOut { u64 *kc, *T; u32 *nsel, *lp, *nd; };

#ifdef __cplusplus
extern "C" {
#endif
int yk_rng_log(void);
int yk_hpart2_chunk(void);
void yk_launch_hpart2(const Chunk2 *chunks, int n_chunks, const u32 *chunk_first, const u64 *bbase, ImgView img, int rb, int P,
                      u32 *rows2, u64 *sbstart, u64 *out, hipStream_t st);
int yk_launch_img_count_rng(const u64 *rec, int cross, const u64 *sbstart, ImgView img, int plo, int phi, int rb, u32 max_len,
                            u64 *list, u32 *list_n, u32 list_cap, hipStream_t st);
void yk_launch_part2(const Chunk2 *chunks, int n_chunks, const u32 *chunk_first /*[P+1]*/, const u64 *bbase, FastParams fp, int P,
                     u32 *rows2, u64 *sbstart, Rec *out, hipStream_t st);
void yk_launch_lds_count_ovf(FastParams fp, const u64 *sbstart, const Rec *rec,
                             u32 *bloom32, ImgView img, LcOut O, const u32 *ovf_list, u32 n_ovf, const u64 *scr_off,
                             u64 *scr, hipStream_t st);
/* replay2 (layout replay of large sub-tables, kernels.hip) */
struct R2Tab { u64 off, rec_off; };
struct R2Act { u32 kind, bits, i0, batch, src, seg0, pad0, pad1; };
struct R2Load { u64 src_off; u32 bits, from_src, dst, pad; };
struct R2Pub { u64 new_off; u32 bits, src; };
void yk_r2_binit(const R2Tab *tabs, const R2Act *acts, int P, u32 bmax, u32 *OCC, u32 *USED, hipStream_t st);
void yk_r2_dsmall(const R2Tab *tabs, const R2Act *acts, int P, u64 *K0, u64 *K1, u32 *TAG, u32 *OCC, u32 *USED, u32 *Fcur, u32 *Gcur, u32 *fail, hipStream_t st);
int yk_r2_double(const R2Tab *tabs, const R2Act *acts, int P, int n_dbl, u64 *K0, u64 *K1, u32 *TAG, u32 *OCC, u32 *USED, const u32 *Fin, u32 *Fout, u32 *fail, hipStream_t st);
void yk_r2_place(const R2Tab *tabs, const R2Act *acts, int P, int p0, int np, u32 bmax, u64 *K0, u64 *K1, const u64 *kc, u64 *pk, u32 *pr, u32 *seg_start,
                 u32 *head, u64 *spill, u32 *spill_n, u32 spill_cap, u32 *fail, u32 *img_u, const u32 *USED, u32 *pcnt, int G, hipStream_t st);
void yk_r2_load(const R2Tab *tabs, const R2Load *ld, int P, u32 bmax, const u64 *src1, const u64 *src2, u64 *K0, u64 *K1, u32 *USED, hipStream_t st);
void yk_r2_trail(const u64 *lastput, const u64 *rec_t, const u64 *rec_off, const u32 *m, int P, u32 *out, hipStream_t st);
void yk_r2_publish(const R2Tab *tabs, const R2Pub *pub, int P, u32 bmax, const u64 *K0, const u64 *K1, u64 *nk, u32 *nu, hipStream_t st);
int yk_r2_small_f(void);
int yk_r2_seg_log(void);
int yk_r2_head(void);
size_t yk_count_own_lds(u32 range_len, u32 kmax);
int yk_launch_img_count_own(const void *rec, int hash_only, int cross, int ytag, const u64 *bstart, ImgView img, int plo, int phi, int rb, int rng_log, u32 kmax,
                            size_t lds, u64 *list, u32 *list_n, u32 list_cap, hipStream_t st);
int yk_lc2_ok(FastParams fp);
int yk_lc2_per_sb(int bloom_mode);
void yk_launch_lc2(FastParams fp, const u64 *sbstart, const Rec *rec, u32 *bloom32, ImgView img, LcOut O, u64 *counters, u32 *ovf_list, hipStream_t st);
void yk_launch_lc_sum(const u32 *nsel, int s2_bits, int plo, int phi, u32 *seg_cnt, hipStream_t st);
void yk_launch_lc_compact(LcOut O, const u64 *sbstart, int s2_bits, int plo, int phi, u64 t_pass0, const u64 *seg_base,
                          u64 *out_kc, u64 *out_T, u64 *lastput, u32 *ndist_p, Rec *out_kt, hipStream_t st);
void yk_launch_part2_ts(const Chunk2 *chunks, int n_chunks, const u32 *chunk_first, const u64 *bbase, FastParams fp, int P, u32 *rows2, u64 *sbstart, Rec *out, hipStream_t st);
int yk_launch_ts_rank(const u64 *binstart, const Rec *in, int w, int j, u32 bin_lo, u32 n_bins, u64 *out_kc, u64 *out_t, u32 *fail, hipStream_t st);
void yk_launch_kt_split(const Rec *in, u64 n, u64 *out_kc, u64 *out_t, hipStream_t st);
void yk_launch_bf_rebuild(FastParams fp, const u64 *sbstart, const Rec *rec, u32 *bloom32, hipStream_t st);
void yk_launch_lc_sum3(LcOut O, int s2_bits, int plo, int phi, u64 t_pass0, u32 *seg_cnt, u64 *lastput, u32 *ndist_p, hipStream_t st);
void yk_launch_lc_gather(LcOut O, const u64 *sbstart, const u64 *key_off, int s2_bits, int plo, int phi, u64 *out_kc, u64 *out_T, Rec *out_kt, hipStream_t st);
void yk_launch_cnt2(FastParams fp, const u64 *sbstart, const Rec *rec, const u64 *key_off, const u64 *key_kc, const u64 *seg_base, u32 *key_cnt, ImgView img, u64 n_keys, u32 *used_delta, hipStream_t st);
void yk_launch_nsel_scan(const u32 *nsel, int s2_bits, int plo, int phi, int P, const u64 *seg_base, u64 *key_off, hipStream_t st);
void yk_launch_seg_sort_pass2(const u64 *seg_base, const u32 *seg_cnt, int P, const u64 *src_kc, const u64 *src_t,
                              u64 *dst_kc, u64 *dst_t, int shift, hipStream_t st, int big);
#ifdef __cplusplus
}
#endif
enum { YKC_NOVF2 = 7 };               /* sub-buckets k_lc2 passed on to the tier behind it */

#endif
