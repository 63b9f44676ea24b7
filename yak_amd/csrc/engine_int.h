/* engine_int.h -- what the engine's own translation units (engine.cpp: passes; layout.cpp: exact slot layout) share and nobody else sees:
 * the context of a table, the device-buffer helpers on top of the pool, the error macro. */
#ifndef YK_ENGINE_INT_H
#define YK_ENGINE_INT_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <map>
#include <string>
#include <mutex>
#include <algorithm>
#include <functional>
#include "yk_device.h"
#include "engine.h"

#define fail(...) yk_set_error(__VA_ARGS__)                  /* this thread's yakamd_last_error() text + a line on stderr; -1 */
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s:%d: %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); } while (0)

static inline double now_ms() { return yk_now_ms(); }
static inline void *pool_alloc(size_t bytes, bool plain = false) { return yk_pool_alloc(bytes, plain); }
static inline void pool_free(void *p) { yk_pool_free(p); }

static inline int64_t env_i64(const char *name, int64_t dflt) { return yk_knob(name, dflt); }

static inline int ceil_log2_u64(u64 x) { int b = 0; while ((1ull << b) < x) ++b; return b; }

/* khashl resize target (reference khashl.h:155-158): bits for a requested slot count */
static inline u32 kh_bits_for(u32 want)
{
	u32 lg = 0, x = want;
	while ((x >>= 1) != 0) ++lg;
	if (want & (want - 1)) ++lg;
	return lg > 2 ? lg : 2;
}


template <class T> static int dmalloc(T **p, size_t n)
{
	if (n == 0) n = 1;
	*p = (T*)pool_alloc(n * sizeof(T));
	if (!*p) return fail("device allocation of %zu bytes failed", n * sizeof(T));
	return 0;
}
template <class T> static void dfree(T *&p) { if (p) { pool_free((void*)p); p = 0; } }

struct yakamd_ctx {
	int k, pre, P, n_hash, bf_shift, nb;
	bool has_bloom;
	int dev, plo, phi;
	hipStream_t st;

	/* table image */
	u32 *d_bits, *d_used, *d_delta;
	u64 *d_off, *d_keys;
	u64 n_slots;                       /* arena size (multiple of 32) */
	std::vector<u32> h_bits, h_count;
	std::vector<u64> h_off;
	u64 img_keys_total;                /* sum of counts */

	/* bloom */
	u32 *d_bf; size_t bf_words;
	bool bf_virgin;                    /* allocated but never written: logically all zero */
	bool bf_deferred;                  /* the last pass left its bits in LDS only (FastParams.bf_nowb): the filter is whatever k_bf_rebuild makes of the retained records (ret2) */
	u32 *d_multi; int multi_bits;

	/* running pass */
	bool delta_dirty;                  /* the running pass left pending counts in d_delta (k_img_fold at its end) */
	bool in_pass; int create_new; bool bloom_mode; bool gate_off; int or_mode;   /* gate_off: puts of a merge never consult the filter */
	AccTab acc; u64 acc_count;
	u64 *d_counters, *d_lastput, *d_lpbatch;
	u32 *d_missing, *d_nmissing;
	Rec *d_rec; int64_t rec_cap;
	u64 *d_newlist, *d_miss, *d_cand; int64_t new_cap;
	uint8_t *d_stage; int64_t stage_cap;
	u32 *d_rows; u64 *d_partial, *d_bstart; int rows_blk; int nb_bits;
	/* fast path: level-1 partitioned batches kept until pass_end */
	struct Kept { Rec *d_rec; u64 n; u64 t0, span; std::vector<u64> bstart; bool owned; int fmt; };   /* fmt 1: tagged 8-byte records (yk_device.h YK_R8_*) */
	std::vector<Kept> kept;
	bool fast; u64 kept_bytes, fast_budget; u64 t_pass0; bool t_pass0_set; u64 keys_at_begin;
	double ms_part2, ms_lds;
	u64 t_end;
	u64 list_t;                        /* running stream time of yak_ch_insert_list calls */
	yakamd_stats_t st_cur, st_last;

	/* level-1 records of the last create_new pass, kept for a count pass over the SAME input (yakamd_retain_input / yakamd_count_retained):
	 * the second pass of the bloom protocol (reference main.c:53-57) then neither reads nor hashes the input again */
	struct Retained { u64 *d_rec; u64 n; std::vector<u64> bstart; };
	std::vector<Retained> retained; bool retain_on, retain_broken; u64 retained_bytes;
	/* ... or, when the whole pass was one slice into an empty table, its level-2 records (grouped by sub-bucket) + the keys every sub-bucket put
	 * into the table: the count pass then owns each key's counter in LDS (k_cnt2) */
	struct Ret2 { Rec *d_r2; u64 *d_sbstart, *d_koff, *d_kkc, *d_segbase; FastParams fp; u64 n_total, n_keys; bool valid; } ret2;
	int n_slices;                      /* slices of the running pass counted so far (fast_flush_slice) */
	u64 src_id[5]; bool src_set;       /* identity of the file the retained records came from + its sequence count (yak_count) */

	std::mutex api_mu;                 /* serialises whole-table entry points that callers may reach from several threads (yak_ch_insert_list) */
	void *d_scratch; size_t scratch_bytes;

	/* host mirror */
	bool host_valid;
	u64 *hm_keys; u32 *hm_used; u64 hm_slots;
	struct yak_ht_t *hts;
};

struct yak_ch_ext { yak_ch_t pub; yakamd_ctx *ctx; u32 magic; int n_sub; yak_ch_t **sub; };   /* n_sub > 1: a table sharded over several GPUs (yak_api.cpp) */
#define EXT_MAGIC 0x59414b41u

/* layout.cpp: the exact khashl slot layout (khashl.h:152-221) of `m[p]` new keys per sub-table, sorted by insertion time, on top of the table image */
int yk_run_replay(yakamd_ctx *c, const std::vector<u32> &m, const u64 *d_seg_off, const u64 *d_rec_kc, const u64 *d_rec_t,
                  const u64 *d_lastput, const std::vector<u32> *init_bits, bool from_empty, const std::vector<u64> *rec_off = 0);
void yk_replay_counters(u32 *used, u32 *refused);            /* debug: replays done by the streaming kernels / handed back to k_replay */
#endif
