/* engine.h -- internal interface between pool.cpp, engine.cpp and the yak.h surface (yak_api.cpp, yak_reader.cpp, yak_multi.cpp) */
#ifndef YK_ENGINE_H
#define YK_ENGINE_H
#include "../../include/yak_amd.h"
#include "yk_device.h"

struct yak_ht_t;
yakamd_ctx *yk_ctx_create(int k, int pre, int n_hash, int n_shift);
void yk_ctx_next_device(int dev);
void yk_ctx_destroy(yakamd_ctx *c);
int  yk_ctx_destroy_bf(yakamd_ctx *c);
int  yk_ctx_clear(yakamd_ctx *c);
int  yk_ctx_hist(yakamd_ctx *c, int64_t *cnt1024);
int  yk_ctx_setcnt(yakamd_ctx *c, int cnt);
int  yk_ctx_shrink(yakamd_ctx *c, int cmin, int cmax, u64 *tot);
int  yk_ctx_subtract(yakamd_ctx *c, yakamd_ctx *other, u64 *tot);
int  yk_ctx_isec(yakamd_ctx *c, yakamd_ctx *other, u64 *tot);
int  yk_ctx_tighten(yakamd_ctx *c);
int  yk_ctx_merge_presize(yakamd_ctx *c, yakamd_ctx *other);
int  yk_ctx_list_hashes(yakamd_ctx *c, int cmin, int cmax, u64 **d_hash, u32 **d_t, u64 *n);
void yk_pool_release(void *p);
int yk_set_error(const char *fmt, ...);                      /* this thread's yakamd_last_error() text (+ a line on stderr); returns -1 */
int64_t yk_knob(const char *name, int64_t dflt);             /* a run-time setting: the test hook's value, else (public names only) the environment's, else dflt */
void *yk_pool_get(size_t bytes);
void *yk_pool_alloc(size_t bytes, bool plain);              /* pool.cpp: a device buffer from the current device's pool (plain: never from the virtual-memory tier); 0 when the device is full */
void yk_pool_free(void *p);
double yk_now_ms(void);                                      /* monotonic wall clock */
int yk_ctx_dump_image_dev(yakamd_ctx *c, int lo, int hi, u64 **d_img, u64 *n_words);   /* the .yak bytes of sub-tables [lo, hi), in pool memory */
void yk_ctx_gate(yakamd_ctx *c, bool on);
void yk_ctx_or_mode(yakamd_ctx *c, int mode);   /* 0 counting; 1 flag loads; 2 saved-count loads (yk_device.h FastParams.or_mode) */
void yk_ctx_lock(yakamd_ctx *c);
void yk_ctx_unlock(yakamd_ctx *c);
void *yk_ctx_scratch(yakamd_ctx *c, size_t bytes);
int  yk_ctx_inc(yakamd_ctx *c, u64 hash, int *count);
int  yk_ctx_resize_to(yakamd_ctx *c, const uint32_t *want);
u64  yk_ctx_keys_total(yakamd_ctx *c);
void yk_ctx_range(yakamd_ctx *c, int *lo, int *hi);   /* the prefix range [lo, hi) this context owns (yakamd_set_shard) */
int  yk_ctx_load(yakamd_ctx *c, const uint32_t *caps, const uint32_t *sizes, const uint64_t *keys);
int  yk_ctx_sync_host(yakamd_ctx *c, yak_ch_t *h);
u64  yk_ctx_list_time(yakamd_ctx *c, u64 n);
int  yk_ctx_device(yakamd_ctx *c);
size_t yk_pool_cached_bytes(void);
size_t yk_pool_held_bytes(int dev);
void yk_pool_report(const char *what);
hipStream_t yk_ctx_stream(yakamd_ctx *c);
void yk_ctx_set_source(yakamd_ctx *c, const uint64_t id[4], int64_t n_seq);
bool yk_ctx_same_source(yakamd_ctx *c, const uint64_t id[4], int64_t *n_seq);
#endif
