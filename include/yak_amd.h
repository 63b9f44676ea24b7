/*
 * yak_amd.h -- device-level entry points of the MI355X counting engine (C ABI, plain pointers).
 *
 * yak.h is the drop-in surface; the functions below are what yak_count()/yak_ch_*() are built
 * from, exported so that a harness can (a) hand over bases that are ALREADY resident in HBM,
 * (b) time the device path without host parsing / PCIe, and (c) shard the sub-tables over several
 * GPUs.  No torch types, no C++ types: `void *` device pointers and sizes only.
 *
 * Input format ("base image"): ASCII sequence bytes; every byte that is not one of
 * A C G T U a c g t u or a raw 0..3 (reference misc.c:4-21) breaks the rolling k-mer exactly like
 * an 'N' or a record boundary does in reference count.c:41, so sequences are simply separated by
 * at least one such byte (e.g. '\n').
 */
#ifndef YAK_AMD_H
#define YAK_AMD_H

#include <stdint.h>
#include <stddef.h>
#include "yak.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct yakamd_ctx yakamd_ctx;        /* device state behind one yak_ch_t */

/* number of usable gfx950 devices (0 if none); never falls back to the CPU */
int yakamd_device_count(void);
/* last error message of the calling thread ("" if none) */
const char *yakamd_last_error(void);

/* engine bound to an existing table (the yak_ch_t returned by yak_ch_init) */
yakamd_ctx *yakamd_ctx_of(yak_ch_t *h);
/* restrict the engine to the sub-tables [lo, hi): k-mers of other prefixes are dropped at
 * insertion (multi-GPU prefix sharding, replaces the kt_for over prefixes of count.c:133) */
int yakamd_set_shard(yak_ch_t *h, int prefix_lo, int prefix_hi);

/* One counting pass = begin, any number of feeds, end  (replaces count.c:85-166).
 * create_new as in yak_ch_insert_list (htab.c:51).  `t0` = position of the first byte of this
 * buffer in the logical input stream (stream order decides the table layout). */
int yakamd_pass_begin(yak_ch_t *h, int create_new);
int yakamd_feed_bases_dev(yak_ch_t *h, const void *d_bases, int64_t n_bytes, uint64_t t0);
int yakamd_feed_bases_host(yak_ch_t *h, const void *h_bases, int64_t n_bytes, uint64_t t0);
/* the same stream packed to 0.375 bytes per base (what count.c:28-43 keeps of a base: its 2-bit code, or "not ACGT"): d_codes =
 * 32-bit words of 16 bases, base j of the stream at bits 2 (j % 16) of word j / 16, code = seq_nt4_table (A 0, C 1, G 2, T 3);
 * d_valid = one bit per base, bit j % 32 of word j / 32, 0 for N / any other byte / the separator between two records.
 * Both device pointers, the codes 16-byte aligned; words past n_bases are not read beyond the last partial one.  The counting
 * passes then read 0.375 B per position instead of 1 */
int yakamd_feed_packed_dev(yak_ch_t *h, const void *d_codes, const void *d_valid, int64_t n_bases, uint64_t t0);
/* the packed image from host memory: `h_packed` = yakamd_packed_bytes(n_bases) bytes, the code words followed (at the next multiple of 16
 * bytes) by the validity words -- what yakamd_pack_bases_host() writes (any thread; yak_count()'s parser threads pack what they parsed,
 * so the stream crosses the bus at 0.375 B per base, count.c:88-110's reader loop being the host side of this) */
int64_t yakamd_packed_bytes(int64_t n_bases);
void yakamd_pack_bases_host(const void *ascii, int64_t n_bases, void *h_packed);
int yakamd_feed_packed_host(yak_ch_t *h, const void *h_packed, int64_t n_bases, uint64_t t0);
/* ... or as pieces, each a whole number of 32-position words (n_words[i] validity words at valid[i], twice as many code words at codes[i]; a
 * piece whose bases end inside its last word leaves the rest of it invalid): laid one behind the other on the device, fed as ONE image of
 * 32 * sum(n_words) stream positions.  yak_count() hands over a window of its parser's segments this way (stream positions only order the
 * k-mers: the up to 31 unused ones behind a segment change nothing) */
int yakamd_feed_packed_pieces_host(yak_ch_t *h, int n_pieces, const void *const *codes, const void *const *valid, const int64_t *n_words, uint64_t t0);
/* device-side packer: ASCII image -> d_codes ((n + 31) / 32 * 8 bytes) and d_valid ((n + 31) / 32 * 4 bytes); `stream` = a hipStream_t or 0 */
int yakamd_pack_bases_dev(const void *d_ascii, int64_t n, void *d_codes, void *d_valid, void *stream);
/* already hashed k-mers (yak_hash64 output) with their stream positions t0 + t[i], t[i] < t_span;
 * device pointers */
int yakamd_feed_hashed_dev(yak_ch_t *h, const void *d_hash_u64, const void *d_t_u32, int64_t n,
                           uint64_t t0, uint64_t t_span);
/* returns the number of keys added to the table by this pass (-1 on error).  Like
 * yak_ch_insert_list it does not touch h->tot: the caller adds (count.c:138) */
int64_t yakamd_pass_end(yak_ch_t *h);

/* extraction only (count.c:28-43): hashed canonical k-mers, and their positions, of a
 * device-resident base image, restricted to prefixes [prefix_lo, prefix_hi) of a 1<<pre split --
 * one call per destination GPU gives the send buffers of the prefix exchange.  Outputs must hold
 * n_bytes entries; returns the count or -1. */
int64_t yakamd_extract_dev(int k, const void *d_bases, int64_t n_bytes,
                           void *d_hash_u64_out, void *d_t_u32_out,
                           int pre, int prefix_lo, int prefix_hi, void *stream);

/* Sharded exchange without re-extraction.  yakamd_partition_dev() turns a device-resident base
 * image into 16-byte records {yak_hash64, position} grouped by sub-table prefix (ascending), and
 * returns the 1<<pre + 1 group offsets in h_bstart; prefixes owned by one rank are contiguous, so a
 * rank's send buffer per destination is a slice.  The receiver hands every source's slice to
 * yakamd_feed_partitioned_dev() together with that slice's own offsets (h_bstart[p] = first record
 * of prefix p inside d_rec, entries outside the shard equal their neighbours). */
int64_t yakamd_partition_dev(int k, int pre, const void *d_bases, int64_t n_bytes, void *d_rec_out, uint64_t *h_bstart);
int yakamd_feed_partitioned_dev(yak_ch_t *h, const void *d_rec, int64_t n, const uint64_t *h_bstart,
                                uint64_t t0, uint64_t t_span);
/* same, but d_rec is LENT: the caller keeps it valid and unmodified until yakamd_pass_end() returns,
 * and the engine works on it in place instead of taking a copy */
int yakamd_feed_partitioned_lent_dev(yak_ch_t *h, const void *d_rec, int64_t n, const uint64_t *h_bstart,
                                     uint64_t t0, uint64_t t_span);
/* The same exchange at 8 bytes per k-mer instance: TAGGED records (hash >> pre) << 12 | toggle << 10 | position in the
 * 1024-position round -- the stream order of a prefix's records is implied by the order the (round-stable) partition
 * wrote them in, so no position travels.  yakamd_tagged_ok(k, pre) != 0 iff the format applies (k < 32,
 * 2k - pre <= 52, pre <= 10); the slice of a source must arrive whole and unpermuted.  lent != 0 as above */
int yakamd_tagged_ok(int k, int pre);
int yakamd_pass_fast(yak_ch_t *h);        /* 1 while h's open pass can take tagged records (it runs on the exclusive-ownership path) */
int64_t yakamd_partition_tagged_dev(int k, int pre, const void *d_bases, int64_t n_bytes, void *d_rec8_out, uint64_t *h_bstart);
int yakamd_feed_partitioned_tagged_dev(yak_ch_t *h, const void *d_rec8, int64_t n, const uint64_t *h_bstart,
                                       uint64_t t0, uint64_t t_span, int lent);

/* The same for passes that only count existing keys (create_new = 0; main.c:57): 8-byte records,
 * just the yak_hash64 values, grouped by prefix the same way. */
int64_t yakamd_partition_hashes_dev(int k, int pre, const void *d_bases, int64_t n_bytes, void *d_hash_out, uint64_t *h_bstart);
int yakamd_count_partitioned_dev(yak_ch_t *h, const void *d_hash_u64, int64_t n, const uint64_t *h_bstart);

/* create_new = 0 pass on bare yak_hash64 values (any order): count the ones present in the table */
int yakamd_count_hashes_dev(yak_ch_t *h, const void *d_hash_u64, int64_t n);

/* The second pass of the filtered protocol reads the SAME input as the first (reference main.c:53-57: `yak count -b`
 * with one file).  yakamd_retain_input(h, 1) before the create_new = 1 pass keeps that pass's hashed, prefix-grouped
 * k-mers (8-byte tagged records) in device memory -- all of them or, beyond an eighth of the device memory
 * (YAKAMD_RETAIN_GB), none; inside the following create_new = 0 pass yakamd_count_retained(h) counts them instead of
 * a feed: 0 = every instance counted (nothing else must be fed), 1 = nothing usable was kept: feed the input as
 * usual, < 0 = error.  The records are released by the count, by the end of any count pass, by the next create_new
 * pass and by yakamd_retain_input(h, 0).  The caller vouches that both passes see the same input; yak_count() does it
 * by itself when the second call names the file of the first (same device, inode, size and modification time). */
int yakamd_retain_input(yak_ch_t *h, int on);
int yakamd_count_retained(yak_ch_t *h);
int64_t yakamd_retained_instances(yak_ch_t *h);

/* Lookup-only path (`yak qv`, reference qv.c:34-86, k < 32).  yakamd_lookup_dev(): d_out_u16[i] =
 * max(0, yak_ch_get()) of the canonical k-mer ENDING at byte i of the base image, 0xffff where no
 * k-mer ends (window shorter than k or holding a non-ACGT byte).  yakamd_qv_reduce_dev(): sequence j
 * is bytes [d_seq_off[j], d_seq_off[j] + d_seq_len[j]) of that image; d_tot / d_non0 receive its
 * number of k-mers / of k-mers present in the table (d_tot = 0xffffffff for a sequence shorter than
 * min_len), and every sequence with non0 >= tot * min_frac adds its values to d_hist1024 (uint64
 * bins, accumulated: zero them first). */
int yakamd_lookup_dev(yak_ch_t *h, const void *d_bases, int64_t n_bytes, void *d_out_u16);
int yakamd_qv_reduce_dev(yak_ch_t *h, const void *d_t_u16, const uint64_t *d_seq_off, const uint32_t *d_seq_len, int64_t n_seq,
                         int min_len, double min_frac, uint32_t *d_tot, uint32_t *d_non0, uint64_t *d_hist1024);

/* Host-only test hook (no device needed): the base image yak_count() hands to the device for a
 * FASTA/FASTQ(.gz) file -- sequences of >= min_len bases, each followed by '\n'.  use_fast_path = 0
 * forces the general record reader for every record.  *out is malloc()ed; returns its length or -1. */
int64_t yakamd_host_image(const char *fn, int min_len, int use_fast_path, char **out);
/* the same stream as yak_count() really hands it over: packed by the parser threads, one image per window -- here unpacked again, a base as
 * 'A' 'C' 'G' 'T', any other position (N, the end of a record, the positions that pad a parsed segment to a multiple of 32) as '\n'.
 * -1 if the file is not one the parallel parser takes (a pipe, a file read by one thread). */
int64_t yakamd_host_image_packed(const char *fn, int min_len, char **out);
/* Host-only test hooks of the reader for ordinary gzip files (csrc/pgz.h; replaces gzread() behind kseq.h:80-96 for yak_count()):
 * the compressed bytes one thread takes per batch, the smallest file the reader takes and the room it keeps in front of a batch for the
 * record the parser carries over (0 / negative: unchanged; defaults 1 MiB, 4 MiB, 64 MiB);
 * the inflated stream of `fn` (malloc()ed *out; -1: not taken, -2: the stream is invalid -- yakamd_last_error()). */
/* Test switches: names that force a code path which the size or shape of the input would otherwise select (INTEGRATION.md section 4 lists
 * them).  No environment variable reaches them; a value set here also overrides the environment for the public knobs.  reset: forget all. */
void yakamd_test_set(const char *name, int64_t value);
void yakamd_test_reset(void);
void yakamd_gz_tune(int64_t chunk_bytes, int64_t min_file_bytes, int64_t front_bytes);
int64_t yakamd_gz_inflate(const char *fn, int n_threads, char **out);

/* device buffers for harnesses that do not bring their own allocator (tests; bench.py uses torch) */
void *yakamd_dev_alloc(size_t bytes);
void yakamd_dev_free(void *p);
int yakamd_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int yakamd_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);

/* diagnostics: [0] doublings done by the parallel replay routine, [1] sent back to the serial one */
void yakamd_debug_counters(uint32_t *out4);

/* yak_count() on several GPUs with the input already in HBM (the multi-GPU driver of reference count.c:129-143's kt_for over prefixes, without
 * the reader): the stream is cut into rounds of one chunk per DISTINCT device of dev_of_rank (in the order the devices first appear there); chunk s
 * of round b lies at d_chunk[b * S + s] on that device, n_bytes[b * S + s] bytes of the base image (ASCII, sequences separated by a non-ACGT byte;
 * at most 2^31 - 4096; 0 = none); stream order = round by round, device by device, exactly as yak_count() deals a file under YAKAMD_GPUS.  Every
 * device groups its chunk's k-mers by prefix, one RCCL grouped send / recv per round (or peer copies, or nothing on one device) moves them to their
 * owners, each rank counts its prefix range.  h0 == 0: returns a new table sharded over n_rank ranks (every yak_ch_* entry point takes it);
 * h0 != 0 (a table this call made): counts the chunks' k-mers that are in it (count.c:155-157) and returns h0.  NULL on failure.
 * *exchange_out (may be NULL): 0 nothing exchanged (one device), 1 RCCL, 2 peer copies, 3 the library's in-process test rig (test switch YAKAMD_MGPU_LOOPBACK) */
yak_ch_t *yakamd_count_multi_dev(const yak_copt_t *opt, yak_ch_t *h0, int n_rank, const int *dev_of_rank, int n_rounds,
                                 const void *const *d_chunk, const int64_t *n_bytes, int *exchange_out);

/* runtime services for callers without a HIP runtime of their own: page-locked host memory, a device-wide synchronise, free / total device memory */
void *yakamd_host_alloc(size_t bytes);
void yakamd_host_free(void *p);
int yakamd_device_sync(void);
int yakamd_mem_info(size_t *free_bytes, size_t *total_bytes);

/* high-water mark of the device memory the library had in use on device `dev` (buffers handed out by its pool; the idle ranges it keeps for the next
 * pass do not count); reset != 0 starts a new measurement from what is in use now */
int64_t yakamd_peak_bytes(int dev, int reset);
/* release the device-memory cache kept between passes (memory pool) */
void yakamd_trim(void);
/* one line per tier on stderr: what the current device's pool has obtained from the driver, holds in use and idle, and how its large buffers came about */
void yakamd_pool_report(const char *what);
/* how many ranks this process's last yak_count() ran as (> 1 on one device: the pass went in that many sweeps over prefix ranges -- YAKAMD_GPUS, or the
 * library's own rule for large unfiltered inputs, which takes more sweeps while the process does not own the device memory of fewer: INTEGRATION.md section 4) */
int yakamd_last_sweeps(void);

/* bring the host view (slot arrays reachable from yak_ch_t) up to date with HBM */
int yakamd_sync_host(yak_ch_t *h);
/* serialise the table in .yak format straight from the host view into memory (malloc'ed) */
int64_t yakamd_dump_mem(yak_ch_t *h, uint8_t **out);
/* the bytes of sub-tables [lo, hi) alone ({u32 capacity, u32 size, keys in slot order} each, htab.c:385-389; no header): one rank's share of the
 * .yak file of a prefix-sharded job; malloc'ed, returns the size or -1 */
int64_t yakamd_dump_range_mem(yak_ch_t *h, int lo, int hi, uint8_t **out);

/* sub-table shape, for tests: capacity and size of sub-table i (device-authoritative) */
int yakamd_subtable(yak_ch_t *h, int i, uint32_t *capacity, uint32_t *size);

/* timing / traffic counters of the last pass (filled by pass_end) */
typedef struct {
	double ms_extract, ms_insert, ms_bloom, ms_select, ms_sort, ms_replay, ms_total;
	double ms_dominant_kernel;      /* accumulated duration of the insert kernel launches */
	int64_t n_dominant_launches;
	int64_t n_instances;            /* valid k-mer windows consumed */
	int64_t n_distinct_seen;        /* distinct k-mers observed by the pass (create_new only) */
	int64_t n_new_keys;             /* keys that entered the table */
	int64_t n_bloom_candidates;     /* keys that needed exact in-batch bloom resolution */
	double ms_part2;                /* level-2 partition (part of ms_extract) */
	double ms_shrink;               /* last yak_ch_shrink on this table: compaction + layout replay */
} yakamd_stats_t;
int yakamd_get_stats(yak_ch_t *h, yakamd_stats_t *st);

#ifdef __cplusplus
}
#endif
#endif
