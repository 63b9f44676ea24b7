/*
 * kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the k-mer counting engine.
 *
 * Pipeline of one counting pass (replaces reference count.c:85-166 + htab.c:51-78):
 *   K1  k_extract        bases -> (yak_hash64 of canonical k-mer, stream position)   [count.c:28-43]
 *   K3  k_acc_insert     records -> accumulator table (first/second occurrence time, count)
 *   K2  k_bf_*           order-exact blocked-bloom gate on first occurrences        [bbf.c:25-42]
 *   K4  k_img_count      create_new == 0: increment keys of the existing table     [htab.c:71-75]
 *   sel k_select_*       keys that enter the table, grouped by sub-table
 *   srt k_seg_sort_pass  per sub-table LSD radix sort by insertion time
 *   K5  k_replay         exact khashl layout: staged FCFS placement + in-place doubling [khashl.h:152-221]
 * All work is 64-bit integer arithmetic; the bound is HBM / L2-atomic traffic, never MFMA.
 */
#include "yk_device.h"
#include <algorithm>

#define WAVE 64

/* ------------------------------------------------------------------------------------------
 * small device helpers
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ u64 yk_hash64(u64 x, u64 m)           /* reference yak-priv.h:11-21 */
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

__device__ __forceinline__ u32 yk_h2b(u32 h, u32 bits) { return (u32)(h * 2654435769u) >> (32 - bits); } /* khashl.h:98 */

__device__ __forceinline__ u64 yk_mix64(u64 z)
{
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return z ^ (z >> 31);
}

/* reverse the order of the 32 two-bit groups of a 64-bit word */
__device__ __forceinline__ u64 yk_rev2(u64 w)
{
	u64 r = __brevll(w);
	return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}

__device__ __forceinline__ u64 lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1; }

__device__ __forceinline__ void block_sync_global()   /* make global writes/atomics of the block visible to the block */
{
	__threadfence();
	__syncthreads();
}

/* base code table (reference misc.c:4-21) */
__device__ const unsigned char d_nt4[256] = {
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

/* ------------------------------------------------------------------------------------------
 * K1: extraction.  One workgroup = one tile of XT_TILE stream positions.
 *   phase 1: 16 B/lane coalesced loads, nt4 translation, 2-bit packing into LDS (+32-base halo)
 *   phase 2: lane <-> position; the k-mer ending at the position is a 2k-bit window of the packed
 *            stream; reverse strand = complement of the window, forward = 2-bit-group reversal;
 *            canonical = min; yak_hash64.  A window is emitted iff its k validity bits are all set,
 *            which is exactly the `l` run counter of count.c:33-42.
 *   phase 3: wave ballot + prefix compaction, one global cursor bump per workgroup, coalesced
 *            stores of (hash, position).
 * ------------------------------------------------------------------------------------------ */
#define XT_TILE    4096
#define XT_THREADS 256
#define XT_ROUNDS  (XT_TILE / XT_THREADS)
#define XT_HALO    64          /* left context: k - 1 <= 62 bases */

struct XtTile {
	u32 code[(XT_TILE + XT_HALO) / 16 + 4];      /* 2 bits per base, 16 bases per word, base j of the stream at bits 2(j%16) */
	u32 valid[(XT_TILE + XT_HALO) / 32 + 4];     /* 1 bit per base: ACGT or not */
	unsigned char lut[256];
};

__device__ __forceinline__ void xt_init(XtTile &S)
{
	const int tid = threadIdx.x;
	if (tid < 64) ((u32*)S.lut)[tid] = ((const u32*)d_nt4)[tid];
	if (tid < 4) { S.code[(XT_TILE + XT_HALO) / 16 + tid] = 0; S.valid[(XT_TILE + XT_HALO) / 32 + tid] = 0; }
	__syncthreads();
}

/* phase 1: translate + pack the tile [tile0 - HALO, tile0 + XT_TILE) into LDS; ends with a barrier */
/* `valid` != 0: the stream comes packed (yakamd_feed_packed_dev) -- `bases` is then the array of 2-bit codes, 16 bases per 32-bit
 * word in the tile's own layout, `valid` one bit per base; tiles start at multiples of 16 positions, so a word is a word */
__device__ __forceinline__ void xt_load(XtTile &S, const uint8_t *__restrict__ bases, int64_t tile0, int64_t n, const u32 *__restrict__ valid = 0)
{
	const int64_t origin = tile0 - XT_HALO;
	for (int w = threadIdx.x; w < (XT_TILE + XT_HALO) / 16; w += blockDim.x) {
		const int64_t pos = origin + 16 * (int64_t)w;
		u32 code = 0, val = 0;
		if (valid) {
			if (pos >= 0 && pos < n) {
				code = ((const u32*)bases)[pos >> 4];
				val = (valid[pos >> 5] >> (pos & 16)) & 0xffffu;
				if (pos + 16 > n) val &= (1u << (n - pos)) - 1;
				u32 m = 0;                                                    /* invalid positions carry code 0, as the ASCII path leaves them */
				for (int j = 0; j < 16; ++j) m |= (val >> j & 1u) * (3u << (2 * j));
				code &= m;
			}
		} else if (pos >= 0 && pos + 16 <= n) {
			const uint4 v = *(const uint4*)(bases + pos);
			const u32 q[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				const u32 c = S.lut[(q[j >> 2] >> (8 * (j & 3))) & 0xff];
				if (c < 4) { code |= c << (2 * j); val |= 1u << j; }
			}
		} else {
			for (int j = 0; j < 16; ++j) {
				const int64_t x = pos + j;
				if (x >= 0 && x < n) {
					const u32 c = S.lut[bases[x]];
					if (c < 4) { code |= c << (2 * j); val |= 1u << j; }
				}
			}
		}
		S.code[w] = code;
		((unsigned short*)S.valid)[w] = (unsigned short)val;
	}
	__syncthreads();
}

__device__ __forceinline__ u64 yk_hash64_64(u64 x)                   /* reference yak-priv.h:23-33 */
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

/* even bits of a 64-bit word packed into the low 32 bits */
__device__ __forceinline__ u64 yk_even_bits(u64 x)
{
	x &= 0x5555555555555555ull;
	x = (x | x >> 1) & 0x3333333333333333ull;
	x = (x | x >> 2) & 0x0f0f0f0f0f0f0f0full;
	x = (x | x >> 4) & 0x00ff00ff00ff00ffull;
	x = (x | x >> 8) & 0x0000ffff0000ffffull;
	return (x | x >> 16) & 0xffffffffull;
}

__device__ u32 d_bad_hash;       /* k >= 32 only: a 64-bit hash equal to a table sentinel was met */

/* k in [32, 63] (reference count.c:45-60 + yak-priv.h:35-39): the four k-bit planes are the low / high
 * bits of the forward strand (first base most significant) and the complemented low / high bits of
 * the reverse strand; the strand is chosen on the high planes, the hash is the sum of two 64-bit mixes */
__device__ __forceinline__ bool xt_kmer_long(const XtTile &S, int q, int k, int pre, int64_t tile0, int64_t n, u64 *h)
{
	const int e = XT_HALO + q;                 /* q = position inside the tile of the k-mer's last base */
	const int s = e - k + 1;
	const int v0 = s >> 5, vo = s & 31;
	const u64 va = (u64)S.valid[v0] | (u64)S.valid[v0 + 1] << 32, vb = S.valid[v0 + 2];
	u64 V = va >> vo;
	if (vo) V |= vb << (64 - vo);
	const u64 kones = (1ull << k) - 1;
	const int w0 = s >> 4, o = 2 * (s & 15);
	const u64 a = (u64)S.code[w0] | (u64)S.code[w0 + 1] << 32, b = (u64)S.code[w0 + 2] | (u64)S.code[w0 + 3] << 32;
	u64 lo = a >> o, hi = b >> o;
	if (o) { lo |= b << (64 - o); hi |= (u64)S.code[w0 + 4] << (64 - o); }
	const u64 L = (yk_even_bits(lo) | yk_even_bits(hi) << 32) & kones;
	const u64 H = (yk_even_bits(lo >> 1) | yk_even_bits(hi >> 1) << 32) & kones;
	const u64 x0 = __brevll(L) >> (64 - k), x1 = __brevll(H) >> (64 - k), x2 = ~L & kones, x3 = ~H & kones;
	const u64 hv = x1 < x3 ? yk_hash64_64(x0) + yk_hash64_64(x1) : yk_hash64_64(x2) + yk_hash64_64(x3);
	*h = hv;
	const bool ok = (V & kones) == kones && tile0 + q < n;
	if (ok && (hv >> pre) == (~0ull >> pre)) d_bad_hash = 1;       /* would collide with the EMPTY slot pattern */
	return ok;
}

/* phase 2: hashed canonical k-mer ending at tile position q; false if the window holds a non-ACGT
 * byte or lies beyond n */
__device__ __forceinline__ bool xt_kmer(const XtTile &S, int q, int k, u64 mask, u64 kones, int64_t tile0, int64_t n, u64 *h)
{
	const int e = XT_HALO + q;                                      /* LDS base index of the k-mer's last base */
	const int s = e - k + 1;
	const u64 V = ((u64)S.valid[s >> 5] | (u64)S.valid[(s >> 5) + 1] << 32) >> (s & 31);
	const int w0 = s >> 4, o = 2 * (s & 15);
	const u64 lo = (u64)S.code[w0] | (u64)S.code[w0 + 1] << 32;
	u64 W = lo >> o;
	if (o) W |= (u64)S.code[w0 + 2] << (64 - o);
	W &= mask;
	const u64 rv = ~W & mask;                          /* count.c:37: base j of the window at bits 2j, complemented */
	const u64 fw = yk_rev2(W) >> (64 - 2 * k);         /* count.c:36: first base most significant */
	*h = yk_hash64(fw < rv ? fw : rv, mask);
	return (V & kones) == kones && tile0 + q < n;
}

/* compacting extraction (no partition): used for the explicit extract entry point */
__global__ __launch_bounds__(XT_THREADS)
void k_extract(const uint8_t *__restrict__ bases, int64_t pos0, int64_t n, int64_t t_sub, int k, int pre, int plo, int phi,
               u64 *__restrict__ out_hash, u32 *__restrict__ out_t, u64 *cursor)
{
	__shared__ XtTile S;
	__shared__ u32 s_cnt[XT_ROUNDS * 4];
	__shared__ u64 s_base;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int64_t tile0 = pos0 + (int64_t)blockIdx.x * XT_TILE;   /* pos0 is a multiple of 16 */
	xt_init(S);
	xt_load(S, bases, tile0, n);
	const u64 mask = k < 32 ? (1ull << (2 * k)) - 1 : ~0ull, kones = (1ull << k) - 1;
	const u32 pmask = (1u << pre) - 1;
	u64 hv[XT_ROUNDS];
	u32 okm = 0;
#pragma unroll
	for (int r = 0; r < XT_ROUNDS; ++r) {
		u64 h;
		const int q = r * XT_THREADS + (int)threadIdx.x;
			bool ok = k < 32 ? xt_kmer(S, q, k, mask, kones, tile0, n, &h) : xt_kmer_long(S, q, k, pre, tile0, n, &h);
		const u32 p = (u32)h & pmask;
		ok = ok && (int)p >= plo && (int)p < phi;
		hv[r] = h;
		okm |= (u32)ok << r;
		const u64 b = __ballot(ok);
		if (lane == 0) s_cnt[r * 4 + wave] = __popcll(b);
	}
	__syncthreads();
	if (tid == 0) {
		u32 acc = 0;
		for (int i = 0; i < XT_ROUNDS * 4; ++i) { const u32 c = s_cnt[i]; s_cnt[i] = acc; acc += c; }
		s_base = acc ? atomicAdd(cursor, (u64)acc) : 0;
	}
	__syncthreads();
	const u64 base = s_base;
#pragma unroll
	for (int r = 0; r < XT_ROUNDS; ++r) {
		const bool ok = okm >> r & 1;
		const u64 b = __ballot(ok);
		if (ok) {
			const u64 d = base + s_cnt[r * 4 + wave] + __popcll(b & lanemask_lt());
			out_hash[d] = hv[r];
			out_t[d] = (u32)(tile0 + r * XT_THREADS + tid - t_sub);
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * K1 with radix partition: the hashed k-mers of a batch are grouped by the top `nb_bits` of their
 * sub-table prefix, so that every later pass walks the accumulator / bloom / table image one
 * contiguous region at a time (the regions being worked on at any moment stay cache resident).
 * Two sweeps over the bases, no global atomics:
 *   hist   : one workgroup = XP_T tiles; bucket counts in an LDS histogram -> one row of `rows`
 *   scan   : rows -> exclusive offsets (per bucket across workgroups) + bucket starts
 *   scatter: same workgroups recompute their k-mers and place them with LDS cursors
 * The same two kernels partition already-hashed records (SRC = 1: exchanged k-mers of the
 * multi-GPU path, yak_ch_insert_list).
 * ------------------------------------------------------------------------------------------ */
#define XP_T 16                                   /* tiles per workgroup: 65536 stream positions */

__device__ __forceinline__ u32 bucket_of(u64 h, int pre, int nb_bits)
{
	const u32 p = (u32)h & ((1u << pre) - 1);
	return nb_bits <= pre ? p >> (pre - nb_bits) : p;   /* nb_bits is clamped to pre by the host */
}

template <int MODE>   /* 0 = histogram, 1 = scatter {hash, position}, 2 = scatter hash only (count-existing passes), 3 = histogram that also counts, per bucket, the 1024-position rounds that contribute to it (count | rounds << 24) */
__global__ __launch_bounds__(XT_THREADS)
void k_xpart(const uint8_t *__restrict__ bases, int64_t pos0, int64_t n, int64_t t_sub, int k, int pre, int plo, int phi,
             int nb_bits, u32 *rows, Rec *__restrict__ out, const u32 *__restrict__ valid)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_bkt[];
	__shared__ XtTile S;
	const int NB = 1 << nb_bits;
	u32 *row = rows + (size_t)blockIdx.x * NB;
	u32 *s_lastr = s_bkt + NB, *s_nr = s_bkt + 2 * NB;                /* MODE 3 only */
	if (MODE == 3) for (int j = threadIdx.x; j < NB; j += XT_THREADS) { s_lastr[j] = 0; s_nr[j] = 0; }
	for (int j = threadIdx.x; j < NB; j += XT_THREADS) s_bkt[j] = (MODE == 1 || MODE == 2) ? row[j] : 0;
	xt_init(S);
	const u64 mask = k < 32 ? (1ull << (2 * k)) - 1 : ~0ull, kones = (1ull << k) - 1;
	const u32 pmask = (1u << pre) - 1;
	for (int t = 0; t < XP_T; ++t) {
		const int64_t tile0 = pos0 + ((int64_t)blockIdx.x * XP_T + t) * XT_TILE;
		if (tile0 >= n) break;
		xt_load(S, bases, tile0, n, valid);
#pragma unroll 4
		for (int r = 0; r < XT_ROUNDS; ++r) {
			u64 h;
			const int q = r * XT_THREADS + (int)threadIdx.x;
			bool ok = k < 32 ? xt_kmer(S, q, k, mask, kones, tile0, n, &h) : xt_kmer_long(S, q, k, pre, tile0, n, &h);
			const u32 p = (u32)h & pmask;
			ok = ok && (int)p >= plo && (int)p < phi;
			if (ok) {
				const u32 bk = bucket_of(h, pre, nb_bits);
				const u32 d = atomicAdd(&s_bkt[bk], 1u);
				if (MODE == 1) out[d] = make_ulonglong2(h, (u64)(u32)(tile0 + r * XT_THREADS + threadIdx.x - t_sub));
				if (MODE == 2) ((u64*)out)[d] = h;
				if (MODE == 3) {                                        /* waves drift apart inside a tile: a bit per round, counted at the end */
					const u32 R = (u32)t * (XT_TILE / 1024) + (u32)r * XT_THREADS / 1024;
					static_assert(XP_T * (XT_TILE / 1024) <= 64, "one bit per round in two words");
					atomicOr(R < 32 ? &s_lastr[bk] : &s_nr[bk], 1u << (R & 31));
				}
			}
		}
		__syncthreads();
	}
	if (MODE == 0) { __syncthreads(); for (int j = threadIdx.x; j < NB; j += XT_THREADS) row[j] = s_bkt[j]; }
	if (MODE == 3) { __syncthreads(); for (int j = threadIdx.x; j < NB; j += XT_THREADS) row[j] = s_bkt[j] | (u32)(__popc(s_lastr[j]) + __popc(s_nr[j])) << 24; }
}

/* ------------------------------------------------------------------------------------------
 * Software write combining for the partition scatters.  A lone 8- or 16-byte store to a random
 * address costs a whole memory transaction (stores are not merged behind the CU: ~21 G records/s
 * measured on gfx950, whatever the record size), while GS neighbouring lanes storing one aligned
 * 64-byte group run at 110-220 G records/s.  So each bucket gets a CAP-record stack in LDS; a round
 * places one record per thread, and every stack that then holds enough to reach the next 64-byte
 * boundary of its output run is flushed as one aligned group by GS lanes.  The order of records
 * inside a bucket is free (later stages order by stream position themselves), so a stack is all
 * the bookkeeping needed.  A record that finds its stack full (a burst of one bucket inside a
 * round) is stored alone, taken from the END of the workgroup's run of that bucket, so the groups
 * stay aligned.  HAS_T: {hash, position} records of 16 bytes (GS = 4), else bare hashes (GS = 8).
 * ------------------------------------------------------------------------------------------ */
struct WcView { u64 *h; u32 *t, *cnt, *head, *tail, *task, *ntask; };

template <int GS, int CAP, bool HAS_T>
__device__ __forceinline__ size_t wc_carve(WcView &w, u32 *lds, int NB, int NT)
{
	w.h = (u64*)lds;
	u32 *p = lds + 2 * (size_t)NB * CAP;
	w.t = p; if (HAS_T) p += (size_t)NB * CAP;
	w.cnt = p; w.head = p + NB; w.tail = p + 2 * NB; w.task = p + 3 * NB; w.ntask = w.task + NT;
	return 0;
}
template <int GS, int CAP, bool HAS_T> __host__ __device__ constexpr size_t wc_lds_bytes(int NB, int NT)
{
	return (size_t)NB * CAP * (HAS_T ? 12 : 8) + (size_t)NB * 12 + (size_t)NT * 4 + 16;
}

template <bool HAS_T> __device__ __forceinline__ void wc_store(void *out, u32 d, u64 h, u32 t)
{
	if (HAS_T) ((Rec*)out)[d] = make_ulonglong2(h, (u64)t); else ((u64*)out)[d] = h;
}

template <int GS, int CAP, bool HAS_T>
__device__ __forceinline__ void wc_place(const WcView &w, u32 b, u64 h, u32 t, u32 par, void *out)
{
	const u32 pos = atomicAdd(&w.cnt[b], 1u);
	if (pos < (u32)CAP) { w.h[b * CAP + pos] = h; if (HAS_T) w.t[b * CAP + pos] = t; }
	else wc_store<HAS_T>(out, atomicSub(&w.tail[b], 1u) - 1, h, t);
	if (pos == (u32)(GS - 1) - (w.head[b] & (GS - 1))) w.task[atomicAdd(&w.ntask[par], 1u)] = b;   /* this record completes a group */
}

/* between two barriers: flush every stack listed this round; GS lanes of one wave per stack */
template <int GS, int CAP, bool HAS_T>
__device__ __forceinline__ void wc_flush(const WcView &w, u32 par, void *out)
{
	const u32 nt = w.ntask[par], q = threadIdx.x & (GS - 1);
	for (u32 ti = threadIdx.x / GS; ti < nt; ti += blockDim.x / GS) {
		const u32 b = w.task[ti];
		const u32 cn = w.cnt[b], stored = cn < (u32)CAP ? cn : (u32)CAP, h0 = w.head[b], need = GS - (h0 & (GS - 1));
		const u32 two = stored - need >= (u32)GS;                 /* a second whole group behind the first */
		const u32 flushed = need + GS * two, rem = stored - flushed;
		if (q < need) wc_store<HAS_T>(out, h0 + q, w.h[b * CAP + q], HAS_T ? w.t[b * CAP + q] : 0);
		if (two) wc_store<HAS_T>(out, h0 + need + q, w.h[b * CAP + need + q], HAS_T ? w.t[b * CAP + need + q] : 0);
		u64 mh = 0; u32 mt = 0;
		if (q < rem) { mh = w.h[b * CAP + flushed + q]; if (HAS_T) mt = w.t[b * CAP + flushed + q]; }
		__builtin_amdgcn_wave_barrier();
		if (q < rem) { w.h[b * CAP + q] = mh; if (HAS_T) w.t[b * CAP + q] = mt; }
		if (q == 0) { w.head[b] = h0 + flushed; w.cnt[b] = rem; }
	}
	if (threadIdx.x == 0) w.ntask[par ^ 1] = 0;
}

/* after the last round: what is left in the stacks (fewer records than reach the next boundary) */
template <int GS, int CAP, bool HAS_T>
__device__ __forceinline__ void wc_drain(const WcView &w, int NB, void *out)
{
	const u32 q = threadIdx.x & (GS - 1);
	for (u32 b = threadIdx.x / GS; b < (u32)NB; b += blockDim.x / GS) {
		const u32 cn = w.cnt[b], stored = cn < (u32)CAP ? cn : (u32)CAP;
		if (q < stored) wc_store<HAS_T>(out, w.head[b] + q, w.h[b * CAP + q], HAS_T ? w.t[b * CAP + q] : 0);
	}
}

/* The same write combining, STABLE at round granularity (tagged 8-byte records, groups of 8): a record that finds its stack
 * full is stored at once at its final place, head + position in the stack's count, and the flush of that round then empties the
 * whole stack in front of it -- so inside a bucket the records of a round stay together and the rounds stay in order.  Uses
 * w.tail as the per-bucket toggle. */
template <int CAP>
__device__ __forceinline__ void wcs_place(const WcView &w, u32 b, u64 v, u32 par, u64 *out)
{
	const u32 pos = atomicAdd(&w.cnt[b], 1u), h0 = w.head[b];
	if (pos < (u32)CAP) w.h[b * CAP + pos] = v; else out[h0 + pos] = v;
	if (pos == 7u - (h0 & 7u)) w.task[atomicAdd(&w.ntask[par], 1u)] = b;
}
template <int CAP>
__device__ __forceinline__ void wcs_flush(const WcView &w, u32 par, u64 *out)
{
	const u32 nt = w.ntask[par], q = threadIdx.x & 7;
	for (u32 ti = threadIdx.x / 8; ti < nt; ti += blockDim.x / 8) {
		const u32 b = w.task[ti];
		const u32 cn = w.cnt[b], h0 = w.head[b];
		if (cn > (u32)CAP) {                                            /* the round overflowed the stack: everything goes, the direct stores sit behind it */
			for (u32 i = q; i < (u32)CAP; i += 8) out[h0 + i] = w.h[b * CAP + i];
			__builtin_amdgcn_wave_barrier();
			if (q == 0) { w.head[b] = h0 + cn; w.cnt[b] = 0; }
			continue;
		}
		const u32 need = 8 - (h0 & 7);
		const u32 two = cn - need >= 8u;
		const u32 flushed = need + 8 * two, rem = cn - flushed;
		if (q < need) out[h0 + q] = w.h[b * CAP + q];
		if (two) out[h0 + need + q] = w.h[b * CAP + need + q];
		u64 mh = 0;
		if (q < rem) mh = w.h[b * CAP + flushed + q];
		__builtin_amdgcn_wave_barrier();
		if (q < rem) w.h[b * CAP + q] = mh;
		if (q == 0) { w.head[b] = h0 + flushed; w.cnt[b] = rem; }
	}
	if (threadIdx.x == 0) w.ntask[par ^ 1] = 0;
}

#define XW_CAP_S 7                                     /* 2 workgroups per CU; the 8th record of a group is stored directly and takes the stack with it */
template <bool TAG>   /* TAG: tagged records (rows carry the toggle); else bare hashes for the count-existing passes (plain rows) */
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))    /* two workgroups per CU: <= 64 VGPRs and <= 80 SGPRs */
void k_xpart_wcs(const uint8_t *__restrict__ bases, int64_t pos0, int64_t n, int k, int pre, int plo, int phi,
                 int nb_bits, const u32 *__restrict__ rows, u64 *__restrict__ out, int ytag, const u32 *__restrict__ valid)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
	__shared__ XtTile S;
	const int NB = 1 << nb_bits, tid = threadIdx.x;
	WcView w;
	wc_carve<8, XW_CAP_S, false>(w, s_dyn, NB, 1024);
	const u32 *row = rows + (size_t)blockIdx.x * NB;
	for (int b = tid; b < NB; b += 1024) { const u32 v = row[b]; w.cnt[b] = 0; w.head[b] = TAG ? v & 0x7fffffffu : v; w.tail[b] = v >> 31; }
	if (tid < 2) w.ntask[tid] = 0;
	xt_init(S);
	const u64 mask = (1ull << (2 * k)) - 1, kones = (1ull << k) - 1;
	const u32 pmask = (1u << pre) - 1;
	u32 par = 0;
	for (int t = 0; t < XP_T; ++t) {
		const int64_t tile0 = pos0 + ((int64_t)blockIdx.x * XP_T + t) * XT_TILE;
		if (tile0 >= n) break;
		xt_load(S, bases, tile0, n, valid);
		for (int r = 0; r < XT_TILE / 1024; ++r, par ^= 1) {
			u64 h;
			bool ok = xt_kmer(S, r * 1024 + tid, k, mask, kones, tile0, n, &h);
			const u32 p = (u32)h & pmask;
			ok = ok && (int)p >= plo && (int)p < phi;
			u32 bk = 0, tg = 0;
			if (ok) {
				bk = bucket_of(h, pre, nb_bits);
				if (TAG) tg = w.tail[bk];
				/* bare hashes, ytag: the sub-table's own bits (the bucket says them) make room for the top bits of the home-slot product
				 * (khashl.h __kh_h2b), so that k_img_count_own's range test is a shift and a compare */
				const u64 hv = ytag ? (h & ~(u64)pmask) | (((u32)(h >> pre) * 2654435769u) >> (32 - pre)) : h;
				wcs_place<XW_CAP_S>(w, bk, TAG ? (h >> pre) << YK_R8_TAG_BITS | (u64)(tg << 10) | (u32)tid : hv, par, out);
			}
			__syncthreads();
			if (TAG && ok) w.tail[bk] = tg ^ 1;                           /* every lane of the bucket writes the same value: the next contributing round gets the other toggle */
			wcs_flush<XW_CAP_S>(w, par, out);
			__syncthreads();
		}
	}
	wc_drain<8, XW_CAP_S, false>(w, NB, out);
}

/* k_xpart's scatter with write combining: 1024 threads, one round = one quarter tile */
#define XW_NT 1024
#define XW_CAP_T 6          /* {hash, position}: groups of 4 */
#define XW_CAP_H 11         /* hash only: groups of 8 */
template <int MODE>   /* 1 = {hash, position}, 2 = hash only */
__global__ __launch_bounds__(XW_NT)
void k_xpart_wc(const uint8_t *__restrict__ bases, int64_t pos0, int64_t n, int64_t t_sub, int k, int pre, int plo, int phi,
                int nb_bits, const u32 *__restrict__ rows, const u64 *__restrict__ bstart, void *__restrict__ out, const u32 *__restrict__ valid)
{
	constexpr bool HAS_T = MODE == 1;
	constexpr int GS = HAS_T ? 4 : 8, CAP = HAS_T ? XW_CAP_T : XW_CAP_H;
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
	__shared__ XtTile S;
	const int NB = 1 << nb_bits, tid = threadIdx.x;
	WcView w;
	wc_carve<GS, CAP, HAS_T>(w, s_dyn, NB, XW_NT);
	const u32 *row = rows + (size_t)blockIdx.x * NB;
	const bool last = blockIdx.x + 1 == gridDim.x;
	for (int b = tid; b < NB; b += XW_NT) {
		w.cnt[b] = 0; w.head[b] = row[b];
		w.tail[b] = last ? (u32)bstart[b + 1] : row[NB + b];      /* the next workgroup's run starts where mine ends */
	}
	if (tid < 2) w.ntask[tid] = 0;
	xt_init(S);
	const u64 mask = k < 32 ? (1ull << (2 * k)) - 1 : ~0ull, kones = (1ull << k) - 1;
	const u32 pmask = (1u << pre) - 1;
	u32 par = 0;
	for (int t = 0; t < XP_T; ++t) {
		const int64_t tile0 = pos0 + ((int64_t)blockIdx.x * XP_T + t) * XT_TILE;
		if (tile0 >= n) break;
		xt_load(S, bases, tile0, n, valid);
		for (int r = 0; r < XT_TILE / XW_NT; ++r, par ^= 1) {
			const int q = r * XW_NT + tid;
			u64 h;
			bool ok = k < 32 ? xt_kmer(S, q, k, mask, kones, tile0, n, &h) : xt_kmer_long(S, q, k, pre, tile0, n, &h);
			const u32 p = (u32)h & pmask;
			ok = ok && (int)p >= plo && (int)p < phi;
			if (ok) wc_place<GS, CAP, HAS_T>(w, bucket_of(h, pre, nb_bits), h, (u32)(tile0 + q - t_sub), par, out);
			__syncthreads();
			wc_flush<GS, CAP, HAS_T>(w, par, out);
			__syncthreads();
		}
	}
	wc_drain<GS, CAP, HAS_T>(w, NB, out);
}

#define RP_CHUNK 65536
template <int MODE>
__global__ __launch_bounds__(256)
void k_rpart(const u64 *__restrict__ in_hash, const u32 *__restrict__ in_t, int64_t n, int pre, int plo, int phi,
             int nb_bits, u32 *rows, Rec *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_bkt[];
	const int NB = 1 << nb_bits;
	u32 *row = rows + (size_t)blockIdx.x * NB;
	for (int j = threadIdx.x; j < NB; j += 256) s_bkt[j] = MODE ? row[j] : 0;
	__syncthreads();
	const int64_t lo = (int64_t)blockIdx.x * RP_CHUNK, hi = lo + RP_CHUNK < n ? lo + RP_CHUNK : n;
	const u32 pmask = (1u << pre) - 1;
	for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
		const u64 h = in_hash[i];
		const u32 p = (u32)h & pmask;
		if ((int)p >= plo && (int)p < phi) {
			const u32 d = atomicAdd(&s_bkt[bucket_of(h, pre, nb_bits)], 1u);
			if (MODE) out[d] = make_ulonglong2(h, (u64)in_t[i]);
		}
	}
	if (!MODE) { __syncthreads(); for (int j = threadIdx.x; j < NB; j += 256) row[j] = s_bkt[j]; }
}

/* rows[blk][b] (counts) -> rows[blk][b] (absolute start of that workgroup's run in bucket b);
 * bstart[b] = first record of bucket b, bstart[NB] = total.  Rows are summed in PS_G groups so the
 * scan is three short, wide kernels instead of one long serial one. */
#define PS_G 64
#define PS_PAR (1ull << 63)
template <bool PAR>   /* PAR: rows hold count | contributing rounds << 24; bit 63 of the sums carries the parity of the rounds */
__global__ __launch_bounds__(256)
void k_part_sum(const u32 *rows, int n_blk, int NB, u64 *partial)
{
	const int b = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
	if (b >= NB) return;
	const int per = (n_blk + PS_G - 1) / PS_G, lo = g * per, hi = lo + per < n_blk ? lo + per : n_blk;
	u64 acc = 0;
	for (int i = lo; i < hi; ++i) { const u32 v = rows[(size_t)i * NB + b]; if (PAR) acc = (acc + (v & 0xffffffu)) ^ ((u64)(v >> 24 & 1) << 63); else acc += v; }
	partial[(size_t)g * NB + b] = acc;
}

__global__ __launch_bounds__(256)
void k_part_mid(u64 *partial, int NB, u64 *bstart)
{
	__shared__ u64 s_tot[256];
	__shared__ u64 s_carry;
	if (threadIdx.x == 0) s_carry = 0;
	__syncthreads();
	for (int b0 = 0; b0 < NB; b0 += 256) {
		const int b = b0 + threadIdx.x;
		u64 run = 0;
		if (b < NB) for (int g = 0; g < PS_G; ++g) { const u64 c = partial[(size_t)g * NB + b]; partial[(size_t)g * NB + b] = run; run = (run + (c & ~PS_PAR)) ^ (c & PS_PAR); }   /* bit 63: parity of the rounds so far (always 0 without PAR) */
		run &= ~PS_PAR;
		s_tot[threadIdx.x] = run;
		__syncthreads();
		if (threadIdx.x == 0) {
			u64 acc = s_carry;
			for (int j = 0; j < 256; ++j) { const u64 c = s_tot[j]; s_tot[j] = acc; acc += c; }
			s_carry = acc;
		}
		__syncthreads();
		if (b < NB) bstart[b] = s_tot[threadIdx.x];
		__syncthreads();
	}
	if (threadIdx.x == 0) bstart[NB] = s_carry;
}

template <bool PAR>
__global__ __launch_bounds__(256)
void k_part_fin(u32 *rows, int n_blk, int NB, const u64 *partial, const u64 *bstart)
{
	const int b = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
	if (b >= NB) return;
	const int per = (n_blk + PS_G - 1) / PS_G, lo = g * per, hi = lo + per < n_blk ? lo + per : n_blk;
	const u64 pg = partial[(size_t)g * NB + b];
	u64 run = bstart[b] + (pg & ~PS_PAR);
	u32 par = (u32)(pg >> 63);
	for (int i = lo; i < hi; ++i) {
		const u32 c = rows[(size_t)i * NB + b];
		if (PAR) { rows[(size_t)i * NB + b] = (u32)run | par << 31; run += c & 0xffffffu; par ^= c >> 24 & 1; }   /* start (< 2^31) | toggle of the workgroup's first contributing round */
		else { rows[(size_t)i * NB + b] = (u32)run; run += c; }
	}
}

static void launch_part_scan(u32 *rows, int n_blk, int nb_bits, u64 *partial, u64 *bstart, hipStream_t st, bool par = false)
{
	const int NB = 1 << nb_bits;
	if (par) hipLaunchKernelGGL(k_part_sum<true>, dim3((NB + 255) / 256, PS_G), dim3(256), 0, st, rows, n_blk, NB, partial);
	else hipLaunchKernelGGL(k_part_sum<false>, dim3((NB + 255) / 256, PS_G), dim3(256), 0, st, rows, n_blk, NB, partial);
	hipLaunchKernelGGL(k_part_mid, dim3(1), dim3(256), 0, st, partial, NB, bstart);
	if (par) hipLaunchKernelGGL(k_part_fin<true>, dim3((NB + 255) / 256, PS_G), dim3(256), 0, st, rows, n_blk, NB, partial, bstart);
	else hipLaunchKernelGGL(k_part_fin<false>, dim3((NB + 255) / 256, PS_G), dim3(256), 0, st, rows, n_blk, NB, partial, bstart);
}

/* ------------------------------------------------------------------------------------------
 * table image probing (khashl get, reference khashl.h:137-150 / htab.c:93-100)
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ int64_t img_find(const ImgView &img, u64 key)
{
	const u32 p = (u32)key & ((1u << img.pre) - 1);
	const u32 bits = img.bits[p];
	if (bits == YK_NOCAP) return -1;
	const u64 kid = key >> img.pre;
	const u64 off = img.off[p];
	const u32 nmask = (1u << bits) - 1;
	u32 i = yk_h2b((u32)kid, bits);
	const u32 first = i;
	for (;;) {
		const u64 a = off + i;
		if (!(img.used[a >> 5] >> (a & 31) & 1)) return -1;
		if (img.keys[a] >> 10 == kid) return (int64_t)a;
		i = (i + 1) & nmask;
		if (i == first) return -1;
	}
}

/* ------------------------------------------------------------------------------------------
 * lookup-only path (`yak qv`, reference qv.c:34-86): t = max(0, yak_ch_get()) of the k-mer ENDING
 * at every position of a base image (QV_NOKMER where none ends: window shorter than k or holding a
 * non-ACGT byte); then per sequence tot = k-mers, non0 = present ones, and the sequences with
 * non0 >= tot * min_frac add all their t values to the 1024-bin histogram.
 * ------------------------------------------------------------------------------------------ */
#define QV_NOKMER 0xffffu
/* yak_ch_get() clamped at 0 (qv.c:59-60), read-only and on the key array alone: the image keeps every
 * unused slot at YK_EMPTY (k_replay publishes it so), which no 2k < 64-bit key can equal, so the
 * `used` bitmap -- a second random 64-byte read per probe -- is not needed here */
__device__ __forceinline__ u32 img_get_count(const ImgView &img, u64 key)
{
	const u32 p = (u32)key & ((1u << img.pre) - 1);
	const u32 bits = img.bits[p];
	if (bits == YK_NOCAP) return 0;
	const u64 kid = key >> img.pre;
	const u64 *keys = img.keys + img.off[p];
	const u32 nmask = (1u << bits) - 1;
	u32 i = yk_h2b((u32)kid, bits);
	const u32 first = i;
	for (;;) {
		const u64 kc = keys[i];
		if (kc == YK_EMPTY) return 0;
		if (kc >> 10 == kid) return (u32)(kc & 1023u);
		i = (i + 1) & nmask;
		if (i == first) return 0;
	}
}
__global__ __launch_bounds__(XT_THREADS)
void k_lookup(const uint8_t *__restrict__ bases, int64_t n, int k, ImgView img, unsigned short *__restrict__ out)
{
	__shared__ XtTile S;
	xt_init(S);
	const u64 mask = (1ull << (2 * k)) - 1, kones = (1ull << k) - 1;
	for (int t = 0; t < XP_T; ++t) {
		const int64_t tile0 = ((int64_t)blockIdx.x * XP_T + t) * XT_TILE;
		if (tile0 >= n) break;
		xt_load(S, bases, tile0, n);
#pragma unroll 4
		for (int r = 0; r < XT_ROUNDS; ++r) {
			const int q = r * XT_THREADS + (int)threadIdx.x;
			u64 h;
			const bool ok = xt_kmer(S, q, k, mask, kones, tile0, n, &h);
			if (tile0 + q < n) {
				u32 v = QV_NOKMER;
				if (ok) v = img_get_count(img, h);
				out[tile0 + q] = (unsigned short)v;
			}
		}
		__syncthreads();
	}
}

/* one wave per sequence; tot = 0xffffffff marks a sequence below min_len (qv.c:45) */
__global__ __launch_bounds__(256)
void k_qv_reduce(const unsigned short *__restrict__ t, const u64 *__restrict__ roff, const u32 *__restrict__ rlen, int64_t n_reads,
                 int min_len, double min_frac, u32 *tot_out, u32 *non0_out, unsigned long long *hist)
{
	__shared__ u32 s_hist[1024];
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (u32 i = threadIdx.x; i < 1024; i += 256) s_hist[i] = 0;
	__syncthreads();
	for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n_reads; r += (int64_t)gridDim.x * 4) {
		const u64 off = roff[r];
		const u32 len = rlen[r];
		if ((int64_t)len < (int64_t)min_len) { if (lane == 0) { tot_out[r] = 0xffffffffu; non0_out[r] = 0; } continue; }
		u32 tot = 0, non0 = 0;
		for (u32 i = lane; i < len; i += 64) { const u32 v = t[off + i]; if (v != QV_NOKMER) { ++tot; non0 += v > 0; } }
		for (int o = 32; o > 0; o >>= 1) { tot += __shfl_xor(tot, o); non0 += __shfl_xor(non0, o); }
		if (lane == 0) { tot_out[r] = tot; non0_out[r] = non0; }
		if ((double)non0 < (double)tot * min_frac) continue;                /* qv.c:83 */
		for (u32 i = lane; i < len; i += 64) { const u32 v = t[off + i]; if (v != QV_NOKMER) atomicAdd(&s_hist[v], 1u); }
	}
	__syncthreads();
	for (u32 i = threadIdx.x; i < 1024; i += 256) if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
}

void yk_launch_lookup(const uint8_t *bases, int64_t n, int k, ImgView img, unsigned short *out, hipStream_t st)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(k_lookup, dim3(yk_xpart_blocks(n)), dim3(XT_THREADS), 0, st, bases, n, k, img, out);
}

void yk_launch_qv_reduce(const unsigned short *t, const u64 *roff, const u32 *rlen, int64_t n_reads, int min_len, double min_frac,
                         u32 *tot_out, u32 *non0_out, u64 *hist, hipStream_t st)
{
	if (n_reads <= 0) return;
	const int64_t want = (n_reads + 3) / 4;
	hipLaunchKernelGGL(k_qv_reduce, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, st, t, roff, rlen, n_reads, min_len, min_frac,
	                   tot_out, non0_out, (unsigned long long*)hist);
}

/* ------------------------------------------------------------------------------------------
 * accumulator table (our own layout: one 32-B slot per distinct k-mer, linear probing).
 * Slot index = prefix-major: the high bits select the sub-table region so that a later
 * prefix-partitioned pass touches one contiguous region per sub-table.
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ u64 acc_home(const AccTab &t, u64 key)
{
	const int rb = t.bits - t.pre;
	const u64 p = key & ((1ull << t.pre) - 1);
	const u64 r = ((key >> t.pre) * 0x9E3779B97F4A7C15ull) >> (64 - rb);
	return p << rb | r;
}

/* find-or-claim; returns the slot, *created = 1 if this call claimed it */
__device__ __forceinline__ AccSlot *acc_claim(const AccTab &t, u64 key, bool *created)
{
	u64 i = acc_home(t, key);
	*created = false;
	for (;;) {
		AccSlot *s = t.s + i;
		u64 cur = s->key;
		if (cur == key) return s;
		if (cur == YK_EMPTY) {
			cur = atomicCAS(&s->key, YK_EMPTY, key);
			if (cur == YK_EMPTY) { *created = true; return s; }
			if (cur == key) return s;
		}
		i = (i + 1) & t.mask;
	}
}

__device__ __forceinline__ AccSlot *acc_find(const AccTab &t, u64 key)
{
	u64 i = acc_home(t, key);
	for (;;) {
		AccSlot *s = t.s + i;
		const u64 cur = s->key;
		if (cur == key) return s;
		if (cur == YK_EMPTY) return 0;
		i = (i + 1) & t.mask;
	}
}

__global__ __launch_bounds__(256)
void k_acc_init(AccSlot *s, u64 n)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		ulonglong2 *q = (ulonglong2*)(s + i);
		q[0] = make_ulonglong2(YK_EMPTY, YK_TINF);
		q[1] = make_ulonglong2(YK_TINF, 0ull);
	}
}

/* K3: every k-mer instance -> accumulator.  Per instance: one 32-B slot read, a CAS only for a new
 * key, atomicMin on t1/t2 only when the instance can still lower them (both only ever decrease,
 * so a stale read errs on the safe side), one fire-and-forget count increment. */
__global__ __launch_bounds__(256)
void k_acc_insert(const Rec *__restrict__ rec, int64_t n, u64 t0,
                  AccTab tab, ImgView img, int img_nonempty, int bloom_mode, u64 *newlist, u64 *counters)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	const int lane = threadIdx.x & 63;
	u32 n_exist = 0;
	for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += stride) {
		const int64_t i = i0 + threadIdx.x;
		bool created = false;
		AccSlot *s = 0;
		if (i < n) {
			const Rec rc = rec[i];
			const u64 key = rc.x, t = t0 + (u32)rc.y;
			int64_t hit = -1;
			if (img_nonempty) hit = img_find(img, key);
			if (hit >= 0) {
				atomicAdd(&img.delta[hit], 1u);       /* a put-call on an existing key: count only */
				++n_exist;
			} else {
				s = acc_claim(tab, key, &created);
				const ulonglong2 tt = *(const ulonglong2*)&s->t1;      /* {t1, t2}, possibly stale-high */
				const u32 c = s->cnt;
				if (c < 4096u) atomicAdd(&s->cnt, 1u);        /* only min(cnt, 1023[+1]) is ever used */
				if (bloom_mode) {
					u64 loser = t;
					if (t < tt.x) {
						const u64 old = atomicMin(&s->t1, t);
						loser = old < t ? t : old;            /* the larger of the two leaves t1 */
					}
					if (loser != YK_TINF && loser < tt.y) atomicMin(&s->t2, loser);
				} else if (t < tt.x) atomicMin(&s->t1, t);
			}
		}
		const u64 b = __ballot(created);
		if (b) {
			u64 base = 0;
			const int leader = __ffsll((long long)b) - 1;
			if (lane == leader) base = atomicAdd(&counters[YKC_NEW], (u64)__popcll(b));
			base = __shfl(base, leader);
			if (created && newlist) newlist[base + __popcll(b & lanemask_lt())] = (u64)(s - tab.s);
		}
	}
	if (img_nonempty) {
		for (int o = 32; o; o >>= 1) n_exist += __shfl_down(n_exist, o);
		if (lane == 0 && n_exist) atomicAdd(&counters[YKC_EXIST], (u64)n_exist);
	}
}

__global__ __launch_bounds__(256)
void k_acc_rehash(AccTab oldt, AccTab newt)
{
	const u64 n = oldt.mask + 1, stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const AccSlot o = oldt.s[i];
		if (o.key == YK_EMPTY) continue;
		bool created;
		AccSlot *s = acc_claim(newt, o.key, &created);
		s->t1 = o.t1; s->t2 = o.t2; s->cnt = o.cnt; s->flags = o.flags;
	}
}

/* K4: create_new == 0 (reference htab.c:71-75): look the key up in the existing table image and
 * count the hit; the saturating fold into the 10 count bits happens once at the end of the pass */
__global__ __launch_bounds__(256)
void k_img_count(const Rec *__restrict__ rec, int64_t n, ImgView img)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const int64_t hit = img_find(img, rec[i].x);
		if (hit >= 0) atomicAdd(&img.delta[hit], 1u);
	}
}

/* K4 with exclusive ownership (records grouped by sub-table prefix): ONE workgroup counts all
 * instances of one sub-table.  The sub-table's `used` bitmap and a per-word rank table live in LDS,
 * so a hit is turned into the rank of its slot among the used slots and counted in a 16-bit LDS
 * counter -- no global atomics; only the key compare reads HBM/L2.  The counters are flushed into
 * the per-slot delta array with plain read-modify-writes (nobody else touches this sub-table). */
template <int W>   /* record width in u64 words: 2 = {hash, position}, 1 = hash only */
__global__ __launch_bounds__(1024)
void k_img_count_lds(const u64 *__restrict__ rec, const u64 *__restrict__ bstart, ImgView img, int plo, u64 *__restrict__ compact, u32 stride)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
	__shared__ u32 s_wsum[16];
	const u32 p = (u32)plo + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const u32 bits = img.bits[p];
	const u64 lo = bstart[p], hi = bstart[p + 1];
	if (bits == YK_NOCAP || lo == hi) return;
	const u32 cap = 1u << bits, nw = (cap + 31) / 32;
	const u64 off = img.off[p];
	u32 *s_bm = s_dyn, *s_rk = s_dyn + nw, *s_ct = s_dyn + 2 * nw;      /* bitmap | rank of each word's first slot | 16-bit counters */
	/* bitmap + exclusive popcount scan (every thread owns a contiguous run of words) */
	const u32 per = (nw + 1023) / 1024;
	u32 mine = 0;
	for (u32 j = 0; j < per; ++j) {
		const u32 w = tid * per + j;
		if (w < nw) { const u32 x = img.used[(off >> 5) + w]; s_bm[w] = x; mine += __popc(x); }
	}
	u32 incl = mine;
	for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o); if (lane >= (u32)o) incl += t; }
	if (lane == 63) s_wsum[wave] = incl;
	__syncthreads();
	u32 base = incl - mine;
	for (u32 w2 = 0; w2 < wave; ++w2) base += s_wsum[w2];
	u32 total = 0;
	for (u32 w2 = 0; w2 < 16; ++w2) total += s_wsum[w2];
	for (u32 j = 0; j < per; ++j) {
		const u32 w = tid * per + j;
		if (w < nw) { s_rk[w] = base; base += __popc(s_bm[w]); }
	}
	for (u32 i = tid; i < (total + 1) / 2; i += 1024) s_ct[i] = 0;
	/* the keys of the used slots, packed in rank order: the random key reads of the probe loop then
	 * fall into count x 8 bytes instead of capacity x 8 -- 2.5x denser, so far more of the 256
	 * sub-tables in flight stay in the 256 MB Infinity Cache */
	u64 *ck = compact + (size_t)blockIdx.x * stride;
	{
		u32 r = incl - mine;
		for (u32 w2 = 0; w2 < wave; ++w2) r += s_wsum[w2];
		for (u32 j = 0; j < per; ++j) {
			const u32 w = tid * per + j;
			if (w >= nw) break;
			u32 x = s_bm[w];
			while (x) { const u32 b = __ffs((int)x) - 1; x &= x - 1; ck[r++] = img.keys[off + w * 32 + b]; }
		}
	}
	__syncthreads();
	const u32 nmask = cap - 1;
	for (u64 i = lo + tid; i < hi; i += 1024) {
		const u64 kid = __builtin_nontemporal_load(&rec[W * i]) >> img.pre;   /* streamed once: keep it out of the caches the keys live in */
		u32 s = yk_h2b((u32)kid, bits);
		const u32 first = s;
		for (;;) {
			const u32 word = s_bm[s >> 5];
			if (!(word >> (s & 31) & 1)) break;                           /* khashl get: stop at the first unused slot */
			const u32 r = s_rk[s >> 5] + __popc(word & ((1u << (s & 31)) - 1));
			if (ck[r] >> 10 == kid) {
				const u32 sh = 16 * (r & 1);
				if ((s_ct[r >> 1] >> sh & 0xffffu) < 4096u) atomicAdd(&s_ct[r >> 1], 1u << sh);   /* only min(count, 1023) matters */
				break;
			}
			s = (s + 1) & nmask;
			if (s == first) break;
		}
	}
	__syncthreads();
	for (u32 j = 0; j < per; ++j) {
		const u32 w = tid * per + j;
		if (w >= nw) break;
		u32 x = s_bm[w], r = s_rk[w];
		while (x) {
			const u32 b = __ffs((int)x) - 1;
			x &= x - 1;
			const u32 c = s_ct[r >> 1] >> (16 * (r & 1)) & 0xffffu;
			if (c) img.delta[off + w * 32 + b] += c;
			++r;
		}
	}
}

/* same on bare hash values (pass 2 of the sharded path ships 8 bytes per instance) */
__global__ __launch_bounds__(256)
void k_img_count_h(const u64 *__restrict__ hash, int64_t n, ImgView img)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const int64_t hit = img_find(img, hash[i]);
		if (hit >= 0) atomicAdd(&img.delta[hit], 1u);
	}
}

/* reference htab.c:80-91 (yak_ch_inc): one key, saturating ++; out[0] = arena slot or ~0, out[1] = new count */
__global__ void k_img_inc(ImgView img, u64 hash, u64 *out)
{
	const int64_t a = img_find(img, hash);
	if (a < 0) { out[0] = ~0ull; out[1] = 0; return; }
	u64 kc = img.keys[a];
	if ((kc & 1023) < 1023) img.keys[a] = ++kc;
	out[0] = (u64)a; out[1] = kc & 1023;
}

__global__ __launch_bounds__(256)
void k_img_fold(ImgView img, u64 n_slots)                    /* htab.c:68-69,73-74: saturate at 1023 */
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride) {
		const u32 d = img.delta[i];
		if (d) {
			const u64 kc = img.keys[i];
			const u64 c = (kc & 1023) + d;
			img.keys[i] = (kc & ~1023ull) | (c > 1023 ? 1023 : c);
			img.delta[i] = 0;
		}
	}
}

/* reference htab.c:145-169 (histogram of counts over all stored k-mers) and htab.c:219-235 (setcnt) */
__global__ __launch_bounds__(256)
void k_img_hist(ImgView img, u64 n_slots, unsigned long long *hist)
{
	__shared__ u32 s_h[1024];
	for (u32 i = threadIdx.x; i < 1024; i += 256) s_h[i] = 0;
	__syncthreads();
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride)
		if (img.used[i >> 5] >> (i & 31) & 1) atomicAdd(&s_h[img.keys[i] & 1023u], 1u);
	__syncthreads();
	for (u32 i = threadIdx.x; i < 1024; i += 256) if (s_h[i]) atomicAdd(&hist[i], (unsigned long long)s_h[i]);
}

__global__ __launch_bounds__(256)
void k_img_setcnt(ImgView img, u64 n_slots, u32 cnt)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride)
		if (img.used[i >> 5] >> (i & 31) & 1) img.keys[i] = (img.keys[i] & ~1023ull) | cnt;
}

__global__ __launch_bounds__(256)
void k_img_clear(ImgView img, u64 n_slots)                   /* reference htab.c:116-125 */
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride)
		if (img.used[i >> 5] >> (i & 31) & 1) img.keys[i] &= ~1023ull;
}

/* ------------------------------------------------------------------------------------------
 * last put-call per sub-table.  khashl grows at the NEXT put-call after the load factor reaches
 * 0.75 (khashl.h:202), even when that call finds its key.  So the layout needs, per sub-table, the
 * stream time of the last put-call of the pass.  Only the tail of the stream can matter; the host
 * scans the tail first and the whole batch only for sub-tables the tail did not reach.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256)
void k_lastput(const Rec *__restrict__ rec, int64_t n, u64 t0, u64 t_from,
               AccTab tab, ImgView img, int img_nonempty, int bloom_mode, const u32 *only_missing, u64 *lp_batch)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	const u32 pmask = (1u << tab.pre) - 1;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const Rec rc = rec[i];
		const u64 t = t0 + (u32)rc.y;
		if (t < t_from) continue;
		const u64 key = rc.x;
		const u32 p = (u32)key & pmask;
		if (only_missing && !(only_missing[p >> 5] >> (p & 31) & 1)) continue;
		bool put = true;
		if (bloom_mode && !(img_nonempty && img_find(img, key) >= 0)) {
			const AccSlot *s = acc_find(tab, key);
			/* the first occurrence is a put-call only if it passed the gate (htab.c:63-65) */
			if (s && s->t1 == t && !(s->flags & YK_FLAG_FP)) put = false;
		}
		if (put) atomicMax(&lp_batch[p], t + 1);
	}
}

__global__ void k_lastput_merge(u64 *lastput, const u64 *lp_batch, u32 *missing, u32 *n_missing, int P, int plo, int phi)
{
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= P) return;
	const u64 v = lp_batch[p];
	if (v) { if (v > lastput[p]) lastput[p] = v; }
	else if (p >= plo && p < phi) { atomicOr(&missing[p >> 5], 1u << (p & 31)); atomicAdd(n_missing, 1u); }
}

/* ------------------------------------------------------------------------------------------
 * K2: order-exact blocked bloom gate (reference bbf.c:25-42 + htab.c:63-65).
 * Only the FIRST occurrence of a k-mer can be rejected by the gate, and it is accepted iff every
 * one of its probe bits was set by the first occurrence of some other k-mer earlier in the stream.
 * Per time-ordered batch, over the keys first seen in the batch:
 *   test   : probe bits against the pre-batch filter -> all set: accepted (FP); else missing mask
 *   set    : OR the missing bits in; a bit found already set here was set by another key of the
 *            same batch -> note it in the (hashed) `multi` filter
 *   check  : a key whose missing bits are ALL noted is a candidate (its fate depends on order)
 *   mapfill/resolve : exact order for candidates through a min-time map over the noted bits
 * ------------------------------------------------------------------------------------------ */
struct BfProbe { u64 bit_base; u32 h1, h2, nd; };

__device__ __forceinline__ BfProbe bf_probe(const BloomView &bf, u64 key, int pre)
{
	BfProbe q;
	const u64 p = key & ((1ull << pre) - 1), x = key >> pre;
	const int xb = bf.nb - 9;
	const u64 blk = x & ((1ull << xb) - 1);
	q.h1 = (u32)(x >> xb) & 511;
	q.h2 = bf.nb < 64 ? (u32)(x >> bf.nb) & 511 : 0;
	if ((q.h2 & 31) == 0) q.h2 = (q.h2 + 1) & 511;
	const u32 cyc = 512u / (q.h2 & (0u - q.h2));         /* probes repeat after 512/gcd(h2,512) steps */
	q.nd = (u32)bf.n_hash < cyc ? (u32)bf.n_hash : cyc;
	q.bit_base = p << bf.nb | blk << 9;
	return q;
}

__global__ __launch_bounds__(256)
void k_bf_test(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, u64 *miss)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_new; i += stride) {
		AccSlot *s = tab.s + newlist[i];
		const BfProbe q = bf_probe(bf, s->key, tab.pre);
		const u32 *blk = bf.bits32 + (q.bit_base >> 5);
		bool any = false;
		for (int w = 0; w < bf.mw; ++w) {
			u64 m = 0;
			for (u32 j = w * 64; j < q.nd && j < (u32)(w + 1) * 64; ++j) {
				const u32 z = (q.h1 + j * q.h2) & 511;
				if (!(blk[z >> 5] >> (z & 31) & 1)) m |= 1ull << (j & 63);
			}
			miss[i * bf.mw + w] = m;
			any |= m != 0;
		}
		if (!any) s->flags |= YK_FLAG_FP;
	}
}

__device__ __forceinline__ u32 multi_idx(u64 bitid, int multi_bits) { return (u32)(yk_mix64(bitid) >> (64 - multi_bits)); }

__global__ __launch_bounds__(256)
void k_bf_set(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
              u32 *multi, int multi_bits, u64 *counters)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_new; i += stride) {
		const AccSlot *s = tab.s + newlist[i];
		if (s->flags & YK_FLAG_FP) continue;
		const BfProbe q = bf_probe(bf, s->key, tab.pre);
		u32 *blk = bf.bits32 + (q.bit_base >> 5);
		for (int w = 0; w < bf.mw; ++w) {
			u64 m = miss[i * bf.mw + w];
			while (m) {
				const u32 j = w * 64 + (__ffsll((long long)m) - 1);
				m &= m - 1;
				const u32 z = (q.h1 + j * q.h2) & 511, bit = 1u << (z & 31);
				const u32 old = atomicOr(&blk[z >> 5], bit);
				if (old & bit) {
					const u32 x = multi_idx(q.bit_base + z, multi_bits);
					atomicOr(&multi[x >> 5], 1u << (x & 31));
					counters[YKC_ANYMULTI] = 1;
				}
			}
		}
	}
}

__global__ __launch_bounds__(256)
void k_bf_check(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
                const u32 *multi, int multi_bits, u64 *cand, u64 *counters)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_new; i += stride) {
		const AccSlot *s = tab.s + newlist[i];
		if (s->flags & YK_FLAG_FP) continue;
		const BfProbe q = bf_probe(bf, s->key, tab.pre);
		bool all = true;
		u32 n_noted = 0;
		for (int w = 0; w < bf.mw; ++w) {
			u64 m = miss[i * bf.mw + w];
			while (m) {
				const u32 j = w * 64 + (__ffsll((long long)m) - 1);
				m &= m - 1;
				const u32 z = (q.h1 + j * q.h2) & 511;
				const u32 x = multi_idx(q.bit_base + z, multi_bits);
				if (multi[x >> 5] >> (x & 31) & 1) ++n_noted; else all = false;
			}
		}
		if (n_noted) atomicAdd(&counters[YKC_NMARKED], (u64)n_noted);
		if (all) cand[atomicAdd(&counters[YKC_NCAND], 1ull)] = i;
	}
}

/* min-time map: open addressing, key = bit id + 1 (0 = free), value = ~(earliest first-occurrence
 * time) so that a zero-filled map means "never" and atomicMax implements the minimum */
__device__ __forceinline__ void map_min(u64 *map, int map_bits, u64 bitid, u64 t)
{
	const u64 mask = (1ull << map_bits) - 1, k1 = bitid + 1;
	u64 i = yk_mix64(bitid) >> (64 - map_bits);
	for (;;) {
		u64 cur = map[2 * i];
		if (cur == 0) cur = atomicCAS(&map[2 * i], 0ull, k1);
		if (cur == 0 || cur == k1) { atomicMax(&map[2 * i + 1], ~t); return; }
		i = (i + 1) & mask;
	}
}

__device__ __forceinline__ u64 map_get(const u64 *map, int map_bits, u64 bitid)
{
	const u64 mask = (1ull << map_bits) - 1, k1 = bitid + 1;
	u64 i = yk_mix64(bitid) >> (64 - map_bits);
	for (;;) {
		const u64 cur = map[2 * i];
		if (cur == k1) return ~map[2 * i + 1];
		if (cur == 0) return YK_TINF;
		i = (i + 1) & mask;
	}
}

__global__ __launch_bounds__(256)
void k_bf_mapfill(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
                  const u32 *multi, int multi_bits, u64 *map, int map_bits)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_new; i += stride) {
		const AccSlot *s = tab.s + newlist[i];
		if (s->flags & YK_FLAG_FP) continue;
		const BfProbe q = bf_probe(bf, s->key, tab.pre);
		const u64 t1 = s->t1;
		for (int w = 0; w < bf.mw; ++w) {
			u64 m = miss[i * bf.mw + w];
			while (m) {
				const u32 j = w * 64 + (__ffsll((long long)m) - 1);
				m &= m - 1;
				const u32 z = (q.h1 + j * q.h2) & 511;
				const u32 x = multi_idx(q.bit_base + z, multi_bits);
				if (multi[x >> 5] >> (x & 31) & 1) map_min(map, map_bits, q.bit_base + z, t1);
			}
		}
	}
}

__global__ __launch_bounds__(256)
void k_bf_resolve(AccTab tab, const u64 *newlist, const u64 *cand, u64 n_cand, BloomView bf,
                  const u64 *miss, const u64 *map, int map_bits)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < n_cand; c += stride) {
		const u64 i = cand[c];
		AccSlot *s = tab.s + newlist[i];
		const BfProbe q = bf_probe(bf, s->key, tab.pre);
		const u64 t1 = s->t1;
		bool all = true;
		for (int w = 0; w < bf.mw && all; ++w) {
			u64 m = miss[i * bf.mw + w];
			while (m) {
				const u32 j = w * 64 + (__ffsll((long long)m) - 1);
				m &= m - 1;
				const u32 z = (q.h1 + j * q.h2) & 511;
				if (!(map_get(map, map_bits, q.bit_base + z) < t1)) { all = false; break; }
			}
		}
		if (all) s->flags |= YK_FLAG_FP;
	}
}

/* ------------------------------------------------------------------------------------------
 * selection: accumulator slots that enter the table -> (insertion time T, key<<10|count) records
 * grouped by sub-table.  No bloom: T = t1, count = occurrences (htab.c:66-69).  Bloom: accepted at
 * the first occurrence (T = t1, every occurrence counted) or at the second (T = t2, one less).
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ bool acc_select(const AccSlot &s, int bloom_mode, int pre, u32 *p, u64 *T, u64 *kc)
{
	if (s.key == YK_EMPTY) return false;
	u64 c = s.cnt;
	if (!bloom_mode || (s.flags & YK_FLAG_FP)) *T = s.t1;
	else if (s.t2 != YK_TINF) { *T = s.t2; c -= 1; }
	else return false;
	if (c > 1023) c = 1023;
	*p = (u32)s.key & ((1u << pre) - 1);
	*kc = (s.key >> pre) << 10 | c;
	return true;
}

/* One workgroup owns SEL_CHUNK consecutive accumulator slots.  The accumulator is prefix-major,
 * so a chunk holds keys of one or two sub-tables only: counts are aggregated in an LDS histogram
 * and flushed with a handful of global atomics per workgroup instead of one per key. */
#define SEL_CHUNK 8192
#define SEL_MAXP  8192

__global__ __launch_bounds__(256)
void k_select_count(AccTab tab, int bloom_mode, u32 *seg_cnt)
{
	__shared__ u32 s_hist[SEL_MAXP];
	const u32 P = 1u << tab.pre;
	const bool lds = P <= SEL_MAXP;
	const u64 n = tab.mask + 1, lo = (u64)blockIdx.x * SEL_CHUNK, hi = lo + SEL_CHUNK < n ? lo + SEL_CHUNK : n;
	if (lds) { for (u32 p = threadIdx.x; p < P; p += 256) s_hist[p] = 0; __syncthreads(); }
	for (u64 i = lo + threadIdx.x; i < hi; i += 256) {
		const AccSlot s = tab.s[i];
		u32 p; u64 T, kc;
		if (acc_select(s, bloom_mode, tab.pre, &p, &T, &kc)) atomicAdd(lds ? &s_hist[p] : &seg_cnt[p], 1u);
	}
	if (lds) {
		__syncthreads();
		for (u32 p = threadIdx.x; p < P; p += 256) if (s_hist[p]) atomicAdd(&seg_cnt[p], s_hist[p]);
	}
}

__global__ __launch_bounds__(256)
void k_select_scatter(AccTab tab, int bloom_mode, const u64 *seg_off, u32 *seg_cur, u64 *rec_kc, u64 *rec_t)
{
	__shared__ u32 s_hist[SEL_MAXP];
	const u32 P = 1u << tab.pre;
	const bool lds = P <= SEL_MAXP;
	const u64 n = tab.mask + 1, lo = (u64)blockIdx.x * SEL_CHUNK, hi = lo + SEL_CHUNK < n ? lo + SEL_CHUNK : n;
	if (lds) {
		for (u32 p = threadIdx.x; p < P; p += 256) s_hist[p] = 0;
		__syncthreads();
		for (u64 i = lo + threadIdx.x; i < hi; i += 256) {
			const AccSlot s = tab.s[i];
			u32 p; u64 T, kc;
			if (acc_select(s, bloom_mode, tab.pre, &p, &T, &kc)) atomicAdd(&s_hist[p], 1u);
		}
		__syncthreads();
		/* reserve this workgroup's range of every sub-table it met; s_hist becomes the running cursor */
		for (u32 p = threadIdx.x; p < P; p += 256) { const u32 c = s_hist[p]; if (c) s_hist[p] = atomicAdd(&seg_cur[p], c); }
		__syncthreads();
	}
	for (u64 i = lo + threadIdx.x; i < hi; i += 256) {      /* second read of the chunk hits L2 */
		const AccSlot s = tab.s[i];
		u32 p; u64 T, kc;
		if (acc_select(s, bloom_mode, tab.pre, &p, &T, &kc)) {
			const u64 d = seg_off[p] + atomicAdd(lds ? &s_hist[p] : &seg_cur[p], 1u);
			rec_kc[d] = kc; rec_t[d] = T;
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * per sub-table stable LSD radix sort pass (8-bit digit of T), one workgroup per sub-table.
 * Stable ranking: the tile is split wave-major, each wave ranks its 64 elements per round with
 * eight ballots (peers with an equal digit), per-wave digit counters live in LDS.
 * ------------------------------------------------------------------------------------------ */
#define SS_E 8
template <int BITS, int NT>   /* digit width; threads per workgroup (256, or 1024 for long segments: four times the loads in flight per sub-table).
                               * 8 bits: a tile of NT x 8 keys leaves >= 8 neighbours per digit, i.e. 64-byte runs in the output; 11-bit digits (two passes for 22-bit
                               * ranks, the first one fused with the gather of the fragments) were measured and lost: one key per digit and tile means lone 8-byte
                               * stores, and the saved pass does not pay for them */
__global__ __launch_bounds__(NT)
void k_seg_sort_pass(const u64 *__restrict__ seg_off, const u32 *__restrict__ seg_len, const u64 *__restrict__ src_kc, const u64 *__restrict__ src_t,
                     u64 *__restrict__ dst_kc, u64 *__restrict__ dst_t, int shift)
{
	constexpr int NBIN = 1 << BITS, NWV = NT / 64;
	static_assert(NBIN == 256, "the scans below give four digits to each lane of one wave");
	__shared__ u32 s_hist[NBIN];
	__shared__ u32 s_wc[NWV][NBIN];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const u64 a = seg_off[blockIdx.x], len = seg_len ? (u64)seg_len[blockIdx.x] : seg_off[blockIdx.x + 1] - a;
	if (len == 0) return;
	if (len == 1) { if (tid == 0) { dst_kc[a] = src_kc[a]; dst_t[a] = src_t[a]; } return; }
	for (int q = tid; q < NBIN; q += NT) s_hist[q] = 0;
	__syncthreads();
	for (u64 i = tid; i < len; i += NT) atomicAdd(&s_hist[(src_t[a + i] >> shift) & (NBIN - 1)], 1u);
	__syncthreads();
	if (tid < 64) {                                              /* exclusive scan of the 256 digit counts by one wave */
		u32 v[4], t = 0;
		for (int q = 0; q < 4; ++q) { v[q] = s_hist[4 * tid + q]; t += v[q]; }
		u32 incl = t;
		for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(incl, o); if (lane >= o) incl += x; }
		u32 e = incl - t;
		for (int q = 0; q < 4; ++q) { s_hist[4 * tid + q] = e; e += v[q]; }
	}
	__syncthreads();
	for (u64 tile = 0; tile < len; tile += (u64)NT * SS_E) {
		for (int w = 0; w < NWV; ++w) for (int q = tid; q < NBIN; q += NT) s_wc[w][q] = 0;
		__syncthreads();
		u64 et[SS_E], ek[SS_E];
		u32 erk[SS_E];
#pragma unroll
		for (int r = 0; r < SS_E; ++r) {
			const u64 idx = tile + (u64)wave * (64 * SS_E) + r * 64 + lane;
			const bool valid = idx < len;
			u64 t = 0, kc = 0;
			if (valid) { t = src_t[a + idx]; kc = src_kc[a + idx]; }
			const u32 d = (u32)(t >> shift) & (NBIN - 1);
			u64 peers = __ballot(valid);
#pragma unroll
			for (int b = 0; b < BITS; ++b) {
				const u64 vb = __ballot(valid && (d >> b & 1));
				peers &= (d >> b & 1) ? vb : ~vb;
			}
			u32 old = 0;
			const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
			if (valid && lane == leader) { old = s_wc[wave][d]; s_wc[wave][d] = old + __popcll(peers); }
			old = __shfl(old, leader);
			et[r] = t; ek[r] = kc;
			erk[r] = valid ? (d << 24 | (old + __popcll(peers & lanemask_lt()))) : 0xffffffffu;
		}
		__syncthreads();
		if (tid < NBIN) {	/* digit `tid`: turn per-wave counts into start offsets, advance the running offset */
			u32 run = s_hist[tid];
			for (int w = 0; w < NWV; ++w) { const u32 c = s_wc[w][tid]; s_wc[w][tid] = run; run += c; }
			s_hist[tid] = run;
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < SS_E; ++r) {
			if (erk[r] != 0xffffffffu) {
				const u64 d = a + s_wc[wave][erk[r] >> 24] + (erk[r] & 0xffffff);
				dst_kc[d] = ek[r]; dst_t[d] = et[r];
			}
		}
		__syncthreads();
	}
}

/* ------------------------------------------------------------------------------------------
 * K5: exact khashl layout replay, one workgroup per sub-table.
 *   * between two doublings every put is a first-come-first-served linear-probing placement of a
 *     NEW key; FCFS in order sigma == ordered probing with priority = rank in sigma, which is
 *     order-free and therefore done in parallel with atomicMin on a per-slot owner rank;
 *   * a doubling is khashl's in-place kick-out rehash (khashl.h:171-189), whose placement order is
 *     data dependent: executed literally by one lane;
 *   * growth happens at the next put-call once count >= 0.75 capacity (khashl.h:202), including a
 *     trailing put-call on an existing key (lastput vs. time of the last new key).
 * ------------------------------------------------------------------------------------------ */
#define RP_U 1
#define RP_PAR_MIN  2048                                 /* doublings from this size on try the exact parallel routine */
#define RP_LDS_WORDS 4096                                 /* doublings up to 131072 slots keep their bitmaps in LDS */
__device__ __forceinline__ bool bm_get(const u32 *u, u32 i) { return u[i >> 5] >> (i & 31) & 1; }

/* khashl's in-place doubling (khashl.h:171-189), executed literally by ONE lane because the order in
 * which keys are re-placed is data dependent.  `cur`/`oth` are the old/new "used" bitmaps (LDS when
 * they fit, else global).  The lane publishes its scan position so that a helper wave can run ahead
 * and pull the cache lines the kick-out chain is about to touch (replay_prefetch). */
__device__ void replay_double(u64 *keys, u32 *cur, u32 *oth, u32 n, u32 N, u32 nbits_new, volatile u32 *progress)
{
	const u32 Nmask = N - 1;
	for (u32 jw = 0; jw < (n + 31) / 32; ++jw) {
		*progress = jw * 32;
		while (cur[jw]) {                                  /* next still-unmoved old slot of this word */
			const u32 j = jw * 32 + (__ffs((int)cur[jw]) - 1);
			if (j >= n) { cur[jw] = 0; break; }
			u64 key = keys[j];
			cur[jw] &= cur[jw] - 1;
			for (;;) {
				u32 i = yk_h2b((u32)(key >> 10), nbits_new);
				u32 wo = oth[i >> 5], wc = i < n ? cur[i >> 5] : 0;     /* both words in flight together */
				while (wo >> (i & 31) & 1) {
					i = (i + 1) & Nmask;
					if ((i & 31) == 0) { wo = oth[i >> 5]; }
					if ((i & 31) == 0 || i == n) wc = i < n ? cur[i >> 5] : 0;
				}
				oth[i >> 5] = wo | 1u << (i & 31);
				if (i < n && (wc >> (i & 31) & 1)) {
					const u64 tmp = keys[i]; keys[i] = key; key = tmp;
					cur[i >> 5] = wc & ~(1u << (i & 31));
				} else { keys[i] = key; break; }
			}
		}
	}
	*progress = 0xffffffffu;
}

/* The same doubling, wave-cooperative.  The ORDER of placements is untouched (lane L of a round
 * commits only after lanes 0..L-1, rounds follow the scan order), but the global-memory latency is
 * taken off the serial path: a round picks the next 64 unmoved slots from the bitmap, all 64 lanes
 * load their key and the 4 slots at its new home at once (an unmoved slot's key never changes before
 * it is moved, so the loads stay valid), then the lanes commit one after the other against the
 * authoritative LDS bitmaps, falling back to a real load only for kicks beyond the first. */
__device__ void replay_double_wave(u64 *keys, u32 *cur, u32 *oth, u32 n, u32 N, u32 nbits_new, volatile u32 *progress, u32 *s_sel)
{
	const u32 lane = threadIdx.x & 63, Nmask = N - 1, nwords = (n + 31) / 32;
	u32 pos = 0;
	while (pos < nwords) {
		if (lane == 0) *progress = pos * 32;
		const u32 w = pos + lane;
		const u32 cw = w < nwords ? cur[w] : 0;
		const u32 c = __popc(cw);
		u32 incl = c;
		for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o); if (lane >= (u32)o) incl += t; }
		const u32 total = __shfl(incl, 63), excl = incl - c;
		if (total == 0) { pos += 64; continue; }
		s_sel[lane] = 0xffffffffu;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (excl < 64) {
			u32 bits = cw, r = excl;
			while (bits && r < 64) { s_sel[r++] = w * 32 + (__ffs((int)bits) - 1); bits &= bits - 1; }
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		const u32 j = s_sel[lane];
		const u32 nsel = total < 64 ? total : 64;
		const u32 last = s_sel[nsel - 1];
		pos = total <= 64 ? pos + 64 : last / 32;          /* leftovers of that word are rescanned next round */
		/* speculative loads: own key, then the old keys sitting where it is going to land */
		u64 key0 = 0, win[4] = { 0, 0, 0, 0 };
		u32 h0 = 0;
		if (j != 0xffffffffu) {
			key0 = keys[j];
			h0 = yk_h2b((u32)(key0 >> 10), nbits_new);
#pragma unroll
			for (int q = 0; q < 4; ++q) { const u32 x = (h0 + q) & Nmask; if (x < n) win[q] = keys[x]; }
		}
		for (u32 L = 0; L < nsel; ++L) {
			if (lane == L && (cur[j >> 5] >> (j & 31) & 1)) {      /* still unmoved: its turn in the scan */
				cur[j >> 5] &= ~(1u << (j & 31));
				u64 key = key0;
				bool first = true;
				for (;;) {
					u32 i = yk_h2b((u32)(key >> 10), nbits_new);
					u32 wo = oth[i >> 5], wc = i < n ? cur[i >> 5] : 0;
					while (wo >> (i & 31) & 1) {
						i = (i + 1) & Nmask;
						if ((i & 31) == 0) wo = oth[i >> 5];
						if ((i & 31) == 0 || i == n) wc = i < n ? cur[i >> 5] : 0;
					}
					oth[i >> 5] = wo | 1u << (i & 31);
					if (i < n && (wc >> (i & 31) & 1)) {
						const u32 d = (i - h0) & Nmask;
						u64 kicked;
						if (first && d < 4) kicked = d == 0 ? win[0] : d == 1 ? win[1] : d == 2 ? win[2] : win[3];
						else kicked = keys[i];
						keys[i] = key; key = kicked;
						cur[i >> 5] = wc & ~(1u << (i & 31));
						first = false;
					} else { keys[i] = key; break; }
				}
			}
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
	}
	if (lane == 0) *progress = 0xffffffffu;
}

/* LDS variant with the two bitmaps interleaved: bm[w] = old "used" word w (high half) | new "used"
 * word w (low half), so that a placement costs one LDS read and one LDS write on its serial path */
__device__ void replay_double_wave64(u64 *keys, u64 *bm, u32 n, u32 N, u32 nbits_new, volatile u32 *progress, u32 *s_sel)
{
	const u32 lane = threadIdx.x & 63, Nmask = N - 1, nwords = (n + 31) / 32;
	u32 pos = 0;
	while (pos < nwords) {
		if (lane == 0) *progress = pos * 32;
		const u32 w = pos + lane;
		const u32 cw = w < nwords ? (u32)(bm[w] >> 32) : 0;
		const u32 c = __popc(cw);
		u32 incl = c;
		for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o); if (lane >= (u32)o) incl += t; }
		const u32 total = __shfl(incl, 63), excl = incl - c;
		if (total == 0) { pos += 64; continue; }
		s_sel[lane] = 0xffffffffu;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (excl < 64) {
			u32 bits = cw, r = excl;
			while (bits && r < 64) { s_sel[r++] = w * 32 + (__ffs((int)bits) - 1); bits &= bits - 1; }
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		const u32 j = s_sel[lane];
		const u32 nsel = total < 64 ? total : 64;
		const u32 last = s_sel[nsel - 1];
		pos = total <= 64 ? pos + 64 : last / 32;
		u64 key0 = 0, win[4] = { 0, 0, 0, 0 };
		u32 h0 = 0;
		if (j != 0xffffffffu) {
			key0 = keys[j];
			h0 = yk_h2b((u32)(key0 >> 10), nbits_new);
#pragma unroll
			for (int q = 0; q < 4; ++q) { const u32 x = (h0 + q) & Nmask; if (x < n) win[q] = keys[x]; }
		}
		for (u32 L = 0; L < nsel; ++L) {
			if (lane == L) {
				u64 b = bm[j >> 5];
				if (b >> (32 + (j & 31)) & 1) {                  /* still unmoved: its turn in the scan */
					u64 key = key0;
					bool first = true;
					u32 i = h0, iw = i >> 5;
					if (iw == (j >> 5)) b &= ~(1ull << (32 + (j & 31)));
					else { bm[j >> 5] = b & ~(1ull << (32 + (j & 31))); b = bm[iw]; }
					for (;;) {
						while (b >> (i & 31) & 1) {                /* linear probing on the new bitmap */
							i = (i + 1) & Nmask;
							if ((i & 31) == 0) { bm[iw] = b; iw = i >> 5; b = bm[iw]; }
						}
						b |= 1ull << (i & 31);
						if (b >> (32 + (i & 31)) & 1) {           /* an unmoved old key sits there: kick it out */
							const u32 d = (i - h0) & Nmask;
							u64 kicked;
							if (first && d < 4) kicked = d == 0 ? win[0] : d == 1 ? win[1] : d == 2 ? win[2] : win[3];
							else kicked = keys[i];
							keys[i] = key; key = kicked;
							b &= ~(1ull << (32 + (i & 31)));
							first = false;
							const u32 i2 = yk_h2b((u32)(key >> 10), nbits_new);
							if ((i2 >> 5) != iw) { bm[iw] = b; iw = i2 >> 5; b = bm[iw]; }
							i = i2;
						} else { keys[i] = key; bm[iw] = b; break; }
					}
				}
			}
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
	}
	if (lane == 0) *progress = 0xffffffffu;
}

/* helper wave of the doubling: for old slots a little ahead of the serial lane, touch the line the
 * key will land on and, one level deeper, the line its kicked-out victim will land on */
__device__ void replay_prefetch(const u64 *keys, u32 n, u32 nbits_new, volatile u32 *progress, u32 *sink)
{
	const u32 lane = threadIdx.x & 63;
	u32 acc = 0, last = 0xfffffffeu;
	for (;;) {
		const u32 p = *progress;
		if (p == 0xffffffffu) break;
		if (p == last) { __builtin_amdgcn_s_sleep(8); continue; }
		last = p;
		const u32 j = p + 48 + lane;                       /* window [p+48, p+112) */
		if (j < n) {
			const u64 key = keys[j];
			if (key != YK_EMPTY) {
				const u32 i = yk_h2b((u32)(key >> 10), nbits_new);
				const u64 v = *(volatile const u64*)&keys[i];
				acc ^= (u32)v;
				if (i < n && v != YK_EMPTY) {
					const u32 i2 = yk_h2b((u32)(v >> 10), nbits_new);
					acc ^= (u32)*(volatile const u64*)&keys[i2];
				}
			}
		}
	}
	if (acc == 0x5a5a5a5au) *sink = acc;                   /* keeps the touches alive */
}

/* ------------------------------------------------------------------------------------------
 * Exact PARALLEL doubling.
 * The serial loop above defines, for every key, a processing time sigma = (c, d): c = old slot at
 * which the scan started the kick-out chain that moved the key, d = its depth in that chain; the
 * final layout is first-come-first-served linear probing in sigma order.  Two facts make it
 * parallel:  (1) FCFS in sigma order == ordered probing with priority sigma (atomicMin per slot);
 * (2) the key at old slot s is kicked by the key x that lands on physical slot s iff c(x) < s, and
 * x comes from an old slot <= s/2 + D1 (new home = 2 * old home + 1 bit, D1 = largest old
 * displacement).  So sigma and landing slot become final front to back: once all old slots < F are
 * final, sigma is known up to ~2F and landings up to ~2F - 4 D1.  Each round is a handful of
 * workgroup-wide passes; the first ~8 D1 slots are done by the literal serial rule.
 * The result is VERIFIED against the defining fixed point (sigma of every key follows from its
 * lander; every key sits at the first slot from its home not held by an earlier key).  That fixed
 * point is unique, so a verified layout is the reference's; on any doubt the caller falls back to
 * the serial routine (nothing has been modified until the commit).
 * ------------------------------------------------------------------------------------------ */
__device__ u32 d_par_ok, d_par_fail;      /* doublings done by the parallel routine / sent back to the serial one */
__device__ u64 d_rp_prof[8];              /* debug (dbg & 32): wall-clock ticks of block 0 per phase */
#define RP_TICK(slot) if ((T.dbg & 32) && blockIdx.x == 0 && tid == 0) { const u64 now_ = wall_clock64(); d_rp_prof[slot] += now_ - tick_; tick_ = now_; }
#define PD_U 4
#define PD_EMPTY  0xffffffffffffffffull
#define PD_FINAL  0x8000000000000000ull
#define PD_PACK(c, d, s) ((u64)(c) << 40 | (u64)(d) << 24 | (u64)(s))
#define PD_C(o) ((u32)((o) >> 40) & 0x7fffffu)
#define PD_D(o) ((u32)((o) >> 24) & 0xffffu)
#define PD_S(o) ((u32)(o) & 0xffffffu)

/* one workgroup on one CU: its waves share the L1, so a workgroup barrier (which carries the
 * workgroup-scope fence) orders its own global traffic; the agent-scope write-back + invalidate of
 * block_sync_global() is only needed once, before the result is handed to the rest of the kernel */
__device__ __forceinline__ void pd_sync() { __syncthreads(); }

__device__ bool replay_double_par(u64 *keys, const u32 *cur, u32 *oth, u32 n, u32 N, u32 nb_new,
                                  u64 *OWN, u64 *SIG, u64 *TMP, u32 *s_par /* LDS [8] */, u64 *lds, size_t lds_bytes,
                                  u64 *xtra /* global, N / 2 entries */, bool prof)
{
	u64 tk_ = wall_clock64();
#define PD_TICK(slot) if (prof && threadIdx.x == 0) { const u64 now_ = wall_clock64(); d_rp_prof[slot] += now_ - tk_; tk_ = now_; }
	const u32 tid = threadIdx.x, Nmask = N - 1, nmask = n - 1;
	u32 *s_fail = s_par, *s_d1 = s_par + 1, *s_cnt = s_par + 2;
	if (tid < 8) s_par[tid] = 0;
	__syncthreads();
	/* 0. copies, largest old displacement */
	u32 d1 = 0, nused = 0;
	for (u32 s = tid; s < n; s += blockDim.x) {
		const u64 k = keys[s];
		TMP[s] = k; SIG[s] = PD_EMPTY;
		if (bm_get(cur, s)) { const u32 h = yk_h2b((u32)(k >> 10), nb_new - 1); const u32 d = (s - h) & nmask; d1 = d > d1 ? d : d1; ++nused; }
	}
	for (u32 i = tid; i < N; i += blockDim.x) OWN[i] = PD_EMPTY;
	atomicMax(s_d1, d1); atomicAdd(s_cnt, nused);
	pd_sync();
	const u32 D1 = *s_d1, n_used = *s_cnt;
	PD_TICK(4)
	const u32 F0 = 5 * D1 + 16, B0 = F0 + 2 * D1 + 4;               /* final after the base phase / simulated by it */
	if (B0 * 4 > n || n > (1u << 23)) { if (tid == 0) s_par[7] = 1; return false; }   /* too clustered / too large: serial */
	/* 1. the literal rule for scan positions below B0; a chain is followed only while it stays below
	 * B0 (what it kicks further up lands beyond anything the first F0 slots can reach, and gets its
	 * sigma from the lander rule later); only slots < F0 are kept, the rest is margin.  One lane does
	 * it, on LDS copies of the few hundred entries involved (the chain is a string of dependent
	 * accesses: ~10x faster than on the global arrays). */
	{
		const u32 WA = 2 * B0 + 4 * D1 + 16, WB = 4 * D1 + 16;           /* landings: [0, WA) and, wrapped, [N - WB, N) */
		if (WA + WB > N) { if (tid == 0) s_par[7] = 2; return false; }
		const bool in_lds = (size_t)(2 * B0 + WA + WB) * 8 <= lds_bytes;      /* else the same walk on global scratch */
		if (!in_lds && 2 * B0 + WA + WB > N / 2) { if (tid == 0) s_par[7] = 2; return false; }
		u64 *w_tmp, *w_sig, *w_own;
		if (in_lds) { w_tmp = lds; w_sig = lds + B0; w_own = lds + 2 * B0; }
		else { w_tmp = xtra; w_sig = xtra + B0; w_own = xtra + 2 * B0; }
		for (u32 j = tid; j < B0; j += blockDim.x) { w_tmp[j] = bm_get(cur, j) ? TMP[j] : PD_EMPTY; w_sig[j] = PD_EMPTY; }
		for (u32 i = tid; i < WA + WB; i += blockDim.x) w_own[i] = PD_EMPTY;
		pd_sync();
		if (tid == 0) {
			bool lost = false;
			for (u32 j = 0; j < B0 && !lost; ++j) {
				if (w_tmp[j] == PD_EMPTY || w_sig[j] != PD_EMPTY) continue;
				u32 s = j, d = 0;
				for (;;) {
					const u64 me = PD_PACK(j, d, s);
					w_sig[s] = me | PD_FINAL;
					u32 i = yk_h2b((u32)(w_tmp[s] >> 10), nb_new);
					for (;;) {
						const u32 li = i < WA ? i : i >= N - WB ? WA + (i - (N - WB)) : 0xffffffffu;
						if (li == 0xffffffffu) { lost = true; break; }
						if (w_own[li] == PD_EMPTY) { w_own[li] = me; break; }
						i = (i + 1) & Nmask;
					}
					if (lost) break;
					if (i < B0 && w_tmp[i] != PD_EMPTY && w_sig[i] == PD_EMPTY) { s = i; ++d; } else break;
				}
			}
			if (lost) { *s_fail = 1; s_par[7] = 3; }
		}
		pd_sync();
		if (*s_fail) return false;
		/* keep what concerns old slots < F0 */
		for (u32 j = tid; j < F0; j += blockDim.x) SIG[j] = w_sig[j];
		for (u32 i = tid; i < WA + WB; i += blockDim.x) {
			const u64 o = w_own[i];
			if (o != PD_EMPTY && PD_S(o) < F0) OWN[i < WA ? i : N - WB + (i - WA)] = o;
		}
	}
	pd_sync();
	PD_TICK(5)
	/* 2. rounds */
	u32 F = F0;
	while (F < n) {
		u32 S1 = 2 * (F - D1) - 1;
		if (S1 > n) S1 = n;
		const u32 S2 = S1 == n ? n : S1 - (2 * D1 + 3);
		if (S2 <= F) { if (tid == 0) s_par[7] = 4; return false; }
		/* A: sigma of the keys whose possible landers are all final */
		for (u32 s0 = F + tid; s0 < S1; s0 += PD_U * blockDim.x) {          /* PD_U independent slots per lane in flight */
			u64 g[PD_U], o[PD_U]; bool ok[PD_U];
#pragma unroll
			for (int u = 0; u < PD_U; ++u) {
				const u32 s = s0 + u * blockDim.x;
				ok[u] = s < S1 && bm_get(cur, s);
				g[u] = ok[u] ? SIG[s] : 0; o[u] = ok[u] ? OWN[s] : 0;
			}
#pragma unroll
			for (int u = 0; u < PD_U; ++u) {
				const u32 s = s0 + u * blockDim.x;
				if (ok[u] && g[u] == PD_EMPTY)
					SIG[s] = (o[u] != PD_EMPTY && PD_S(o[u]) != s && PD_C(o[u]) < s) ? PD_PACK(PD_C(o[u]), PD_D(o[u]) + 1, s) : PD_PACK(s, 0, s);
			}
		}
		pd_sync();
		/* B: ordered probing of those keys on top of the final ones */
		for (u32 s0 = F + tid; s0 < S1; s0 += PD_U * blockDim.x) {
			u64 cand[PD_U], old[PD_U]; u32 i[PD_U], act = 0;
#pragma unroll
			for (int u = 0; u < PD_U; ++u) {
				const u32 s = s0 + u * blockDim.x;
				if (s < S1 && bm_get(cur, s)) { cand[u] = SIG[s]; i[u] = yk_h2b((u32)(TMP[s] >> 10), nb_new); if (!(cand[u] & PD_FINAL)) act |= 1u << u; }
			}
			for (u32 guard = 0; act && guard < N; ++guard) {
#pragma unroll
				for (int u = 0; u < PD_U; ++u) if (act >> u & 1) old[u] = atomicMin(&OWN[i[u]], cand[u]);
#pragma unroll
				for (int u = 0; u < PD_U; ++u)
					if (act >> u & 1) {
						if (old[u] == PD_EMPTY) { act &= ~(1u << u); continue; }
						if (old[u] > cand[u]) {
							if (__hip_atomic_load(&SIG[PD_S(old[u])], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & PD_FINAL) { *s_fail = 1; act &= ~(1u << u); continue; }
							cand[u] = old[u];
						}
						i[u] = (i[u] + 1) & Nmask;
					}
			}
		}
		pd_sync();
		if (*s_fail) { if (tid == 0) s_par[7] = 5; return false; }
		/* C: finalise [F, S2); take the not yet final participants [S2, S1) out again */
		for (u32 s = F + tid; s < S2; s += blockDim.x) if (bm_get(cur, s) && !(SIG[s] & PD_FINAL)) SIG[s] |= PD_FINAL;
		if (S2 < S1) {
			const u32 lo = 2 * (S2 > D1 ? S2 - D1 : 0), hi0 = 2 * S1 + 4 * D1 + 16, hi = hi0 < N ? hi0 : N;
			for (u32 i = (lo > 2 ? lo - 2 : 0) + tid; i < hi; i += blockDim.x) {
				const u64 o = OWN[i];
				if (o != PD_EMPTY && PD_S(o) >= S2 && PD_S(o) < S1 && !(SIG[PD_S(o)] & PD_FINAL)) OWN[i] = PD_EMPTY;
			}
			pd_sync();
			for (u32 s = S2 + tid; s < S1; s += blockDim.x) if (bm_get(cur, s) && !(SIG[s] & PD_FINAL)) SIG[s] = PD_EMPTY;
		}
		pd_sync();
		F = S2;
	}
	PD_TICK(6)
	/* 3. verification of the fixed point */
	if (tid < 8 && tid >= 2) s_par[tid] = 0;
	__syncthreads();
	u32 placed = 0, bad = 0;
	for (u32 i0 = tid; i0 < N; i0 += PD_U * blockDim.x) {
		u64 o[PD_U], g[PD_U], k[PD_U];
#pragma unroll
		for (int u = 0; u < PD_U; ++u) { const u32 i = i0 + u * blockDim.x; o[u] = i < N ? OWN[i] : PD_EMPTY; }
#pragma unroll
		for (int u = 0; u < PD_U; ++u) if (o[u] != PD_EMPTY) { g[u] = SIG[PD_S(o[u])]; k[u] = TMP[PD_S(o[u])]; }
#pragma unroll
		for (int u = 0; u < PD_U; ++u) {
			if (o[u] == PD_EMPTY) continue;
			const u32 i = i0 + u * blockDim.x;
			++placed;
			if ((g[u] & ~PD_FINAL) != o[u]) { bad = 1; continue; }
			u32 q = yk_h2b((u32)(k[u] >> 10), nb_new), steps = 0;
			while (q != i) {                                          /* every slot before it holds an earlier key */
				const u64 p = OWN[q];
				if (p == PD_EMPTY || p > o[u] || ++steps > 4 * D1 + 64) { bad = 1; break; }
				q = (q + 1) & Nmask;
			}
		}
	}
	for (u32 s = tid; s < n; s += blockDim.x) {
		if (!bm_get(cur, s)) continue;
		const u64 g = SIG[s], o = OWN[s];
		if (g == PD_EMPTY || !(g & PD_FINAL)) { bad = 1; continue; }
		const u64 want = (o != PD_EMPTY && PD_S(o) != s && PD_C(o) < s) ? PD_PACK(PD_C(o), PD_D(o) + 1, s) : PD_PACK(s, 0, s);
		if ((g & ~PD_FINAL) != want) bad = 1;
	}
	atomicAdd(s_par + 3, placed);
	if (bad) *s_fail = 1;
	__syncthreads();
	if (*s_fail || s_par[3] != n_used) { if (tid == 0) s_par[7] = 6; return false; }
	/* 4. commit: move the keys, publish the new bitmap (one slot per lane: a wave's ballot is two bitmap words) */
	for (u32 i0 = 0; i0 < N; i0 += blockDim.x) {
		const u32 i = i0 + tid;
		const u64 o = i < N ? OWN[i] : PD_EMPTY;
		if (o != PD_EMPTY) keys[i] = TMP[PD_S(o)];
		const u64 b = __ballot(o != PD_EMPTY);
		if ((tid & 63) == 0 && i < N) { oth[i >> 5] = (u32)b; if (i + 32 < N) oth[(i >> 5) + 1] = (u32)(b >> 32); }
	}
	__syncthreads();
	PD_TICK(7)
	return true;
}

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_replay(const ReplayTask *tasks, const u64 *old_keys, const u32 *old_used, u64 *new_keys, u32 *new_used,
              u32 *scr_used, u32 *scr_owner, u64 *scr_par, const u64 *rec_kc, const u64 *rec_t, const u64 *lastput,
              u32 *out_bits, u32 *out_count, u32 lds_words)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];   /* >= 2 * RP_LDS_WORDS words: placement owner ranks, or ... */
	u64 *s_bm = (u64*)s_dyn;                               /* ... doubling: old (high) and new (low) bitmap words, interleaved */
	__shared__ u32 s_progress;
	__shared__ u32 s_sel[64];
	__shared__ u32 s_par[8];
	const ReplayTask T = tasks[blockIdx.x];
	const int tid = threadIdx.x;
	u64 *const gkeys = new_keys + T.new_off;
	/* while the table has <= 8192 slots its keys live in LDS (second half of s_dyn): the doublings of
	 * that phase are the serial routine, whose cost is the latency of its dependent key accesses */
	u64 *const lkeys = (u64*)(s_dyn + 16384);
	u64 *keys = (lds_words >= 32768 && (T.old_bits == YK_NOCAP || T.old_bits <= 12) && (T.init_bits == YK_NOCAP || T.init_bits <= 12) && !(T.dbg & 64)) ? lkeys : gkeys;
	u32 *UA = new_used + (T.new_off >> 5), *UB = scr_used + (T.new_off >> 5);
	u32 *owner = scr_owner + T.new_off;
	u32 *cur = UA, *oth = UB;
	u32 n = T.old_bits == YK_NOCAP ? 0 : 1u << T.old_bits, bits = T.old_bits == YK_NOCAP ? 0 : T.old_bits;
	u32 cnt = T.old_count;

	for (u32 i = tid; i < n; i += blockDim.x) keys[i] = old_keys[T.old_off + i];
	for (u32 w = tid; w < (n + 31) / 32; w += blockDim.x) UA[w] = old_used[(T.old_off >> 5) + w];
	if (n == 0 && T.init_bits != YK_NOCAP) {                 /* yak_ht_resize(f, size) on an empty set (htab.c:186) */
		bits = T.init_bits; n = 1u << bits;
		for (u32 w = tid; w < (n + 31) / 32; w += blockDim.x) UA[w] = 0;
	}
	__syncthreads();

	u32 i0 = 0;
	u64 tick_ = wall_clock64();
	for (;;) {
		const u32 thr = (n >> 1) + (n >> 2);
		bool grow = false;
		if (i0 < T.m) grow = cnt >= thr;
		else {
			/* all new keys are in; one more doubling if a put-call follows the moment the load hit 75 % */
			const u64 lp = lastput ? lastput[blockIdx.x] : 0;
			const bool later_put = lp != 0 && (T.m == 0 || lp - 1 > rec_t[T.rec_off + T.m - 1]);
			grow = later_put && cnt >= thr && i0 != 0xffffffffu;
			if (!grow) break;
			i0 = 0xffffffffu;                              /* at most one trailing doubling */
		}
		if (grow) {
			const u32 N = n ? n << 1 : 4, nb = n ? bits + 1 : 2;
			bool done = false;
			if (scr_par && n >= RP_PAR_MIN && !(T.dbg & 16)) {
				u64 *pb = scr_par + 2 * T.new_off;
				done = replay_double_par(keys, cur, oth, n, N, nb, pb, pb + N, pb + N + n, s_par, (u64*)s_dyn, (size_t)(lds_words < 16384 ? lds_words : 16384) * 4,
				                         (u64*)(scr_owner + T.new_off), (T.dbg & 32) && blockIdx.x == 0);
				if (tid == 0) atomicAdd(done ? &d_par_ok : &d_par_fail, 1u);
				if (!done && (T.dbg & 128) && tid == 0) printf("parfail n=%u code=%u\n", n, s_par[7]);
			}
			if (!done) {
			const bool in_lds = (N + 31) / 32 <= RP_LDS_WORDS && !(T.dbg & 8);
			if (in_lds) {
				for (u32 w = tid; w < (N + 31) / 32; w += blockDim.x) s_bm[w] = w < (n + 31) / 32 ? (u64)cur[w] << 32 : 0ull;
			} else {
				for (u32 w = tid; w < (N + 31) / 32; w += blockDim.x) oth[w] = 0;
			}
			if (tid == 0) s_progress = 0;
			__syncthreads();
			if (T.dbg & 1) { if (tid == 0) s_progress = 0xffffffffu; }
			else if (tid < 64) {
				if (in_lds) replay_double_wave64(keys, s_bm, n, N, nb, &s_progress, s_sel);
				else replay_double_wave(keys, cur, oth, n, N, nb, &s_progress, s_sel);
			} else if (tid < 128 && !(T.dbg & 4) && keys == gkeys) replay_prefetch(keys, n, nb, &s_progress, scr_owner + T.new_off);
			__syncthreads();
			if (in_lds) { for (u32 w = tid; w < (N + 31) / 32; w += blockDim.x) oth[w] = (u32)s_bm[w]; __syncthreads(); }
			}
			u32 *t = cur; cur = oth; oth = t;
			n = N; bits = nb;
			if (keys == lkeys && n >= 8192) {                 /* grown out of LDS */
				for (u32 i = tid; i < n; i += blockDim.x) gkeys[i] = lkeys[i];
				__syncthreads();
				keys = gkeys;
			}
			RP_TICK(n >= 32768 ? 1 : 0)
			if (i0 == 0xffffffffu) break;
			continue;
		}
		/* FCFS placement of the next keys, up to the growth threshold: every key walks from its home
		 * slot and claims a slot with min(rank); a displaced later key is carried on by the claimer.
		 * The owner ranks live in LDS while the table has <= 2 x lds_words slots (32-bit ranks, or
		 * 16-bit ones updated by compare-and-swap), else in global scratch (memory-side atomics). */
		const u32 batch = (T.m - i0 < thr - cnt) ? T.m - i0 : thr - cnt;
		const u32 nmask = n - 1;
		if (T.dbg & 2) { cnt += batch; i0 += batch; continue; }
		const u64 *src = rec_kc + T.rec_off + i0;
#define RP_PLACE(INIT, AMIN, LOAD)                                                                      \
		for (u32 i = tid; i < n; i += blockDim.x) { INIT(i, bm_get(cur, i)); }                              \
		__syncthreads();                                                                                    \
		for (u32 q0 = tid; q0 < batch; q0 += RP_U * blockDim.x) {       /* RP_U independent keys per lane in flight */ \
			u32 r[RP_U], slot[RP_U], old[RP_U], act = 0;                                                    \
			_Pragma("unroll")                                                                               \
			for (int u = 0; u < RP_U; ++u) {                                                                \
				const u32 q = q0 + u * blockDim.x;                                                          \
				if (q < batch) { r[u] = q + 1; slot[u] = yk_h2b((u32)(src[q] >> 10), bits); act |= 1u << u; } \
			}                                                                                               \
			while (act) {                                                                                   \
				_Pragma("unroll")                                                                           \
				for (int u = 0; u < RP_U; ++u) if (act >> u & 1) { AMIN(old[u], slot[u], r[u]); }           \
				_Pragma("unroll")                                                                           \
				for (int u = 0; u < RP_U; ++u)                                                              \
					if (act >> u & 1) {                                                                     \
						if (old[u] == EMPTYV) { act &= ~(1u << u); continue; }                              \
						if (old[u] > r[u]) r[u] = old[u];   /* we took the slot; carry the displaced later key on */ \
						slot[u] = (slot[u] + 1) & nmask;                                                    \
					}                                                                                       \
			}                                                                                               \
		}                                                                                                   \
		__syncthreads();                                                                                    \
		for (u32 i0 = 0; i0 < n; i0 += blockDim.x) {          /* one slot per lane; a wave's ballot is two bitmap words */ \
			const u32 i = i0 + tid;                                                                         \
			u32 o = 0;                                                                                      \
			if (i < n) { LOAD(o, i); }                                                                      \
			const bool fresh = i < n && o != 0 && o != EMPTYV;                                              \
			if (fresh) keys[i] = src[o - 1];                                                                \
			const u64 b = __ballot(fresh);                                                                  \
			if ((tid & 63) == 0 && i < n) { cur[i >> 5] |= (u32)b; if (i + 32 < n) cur[(i >> 5) + 1] |= (u32)(b >> 32); } \
		}                                                                                                   \
		__syncthreads();
		if (n <= lds_words) {
#define EMPTYV 0xffffffffu
#define INIT32(i, used) s_dyn[i] = (used) ? 0u : EMPTYV
#define AMIN32(old, slot, r) old = atomicMin(&s_dyn[slot], r)
#define LOAD32(o, i) o = s_dyn[i]
			RP_PLACE(INIT32, AMIN32, LOAD32)
#undef EMPTYV
		} else if (n <= 2 * lds_words && batch < 0xffffu) {
#define EMPTYV 0xffffu
#define INIT16(i, used) ((unsigned short*)s_dyn)[i] = (used) ? (unsigned short)0 : (unsigned short)0xffff
#define AMIN16(old, slot, r) { u32 *wp = &s_dyn[(slot) >> 1]; const u32 sh = 16 * ((slot) & 1); u32 seen = *(volatile u32*)wp;        \
				for (;;) { old = seen >> sh & 0xffffu; if (old <= (r)) break;                                                            \
				           const u32 got = atomicCAS(wp, seen, (seen & ~(0xffffu << sh)) | (r) << sh); if (got == seen) break; seen = got; } }
#define LOAD16(o, i) o = ((unsigned short*)s_dyn)[i]
			RP_PLACE(INIT16, AMIN16, LOAD16)
#undef EMPTYV
		} else if (scr_par && lds_words >= 1024 && (lds_words & (lds_words - 1)) == 0 && !(T.dbg & 256)) {
			/* larger tables: the same ordered probing, one segment of lds_words slots at a time with the
			 * ranks in LDS; a walk that reaches the end of its segment is parked as (rank, next slot) and
			 * finished afterwards on the written-back global array -- the fixed point of min-rank
			 * probing does not depend on the order in which the walks are made */
			const u32 SEG = lds_words, nseg = n / SEG;
			u64 *spill = scr_par + 2 * T.new_off;                    /* free between doublings; 2 x capacity entries */
			if (tid == 0) s_par[6] = 0;
			for (u32 seg = 0; seg < nseg; ++seg) {
				const u32 base = seg * SEG;
				for (u32 i = tid; i < SEG; i += blockDim.x) s_dyn[i] = bm_get(cur, base + i) ? 0u : 0xffffffffu;
				__syncthreads();
				for (u32 q = tid; q < batch; q += blockDim.x) {
					const u32 home = yk_h2b((u32)(src[q] >> 10), bits);
					if (home / SEG != seg) continue;
					u32 r = q + 1, li = home - base;
					for (;;) {
						const u32 old = atomicMin(&s_dyn[li], r);
						if (old == 0xffffffffu) break;
						if (old > r) r = old;
						if (++li == SEG) { spill[atomicAdd(&s_par[6], 1u)] = (u64)r << 32 | ((base + SEG) & nmask); break; }
					}
				}
				__syncthreads();
				for (u32 i = tid; i < SEG; i += blockDim.x) owner[base + i] = s_dyn[i];
				__syncthreads();
			}
			const u32 ns = s_par[6];
			for (u32 j = tid; j < ns; j += blockDim.x) {
				u32 r = (u32)(spill[j] >> 32), slot = (u32)spill[j];
				for (;;) {
					const u32 old = atomicMin(&owner[slot], r);
					if (old == 0xffffffffu) break;
					if (old > r) r = old;
					slot = (slot + 1) & nmask;
				}
			}
			__syncthreads();
			for (u32 i0 = 0; i0 < n; i0 += blockDim.x) {
				const u32 i = i0 + tid;
				const u32 o = __hip_atomic_load(&owner[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const bool fresh = o != 0 && o != 0xffffffffu;
				if (fresh) keys[i] = src[o - 1];
				const u64 b = __ballot(fresh);
				if ((tid & 63) == 0) { cur[i >> 5] |= (u32)b; cur[(i >> 5) + 1] |= (u32)(b >> 32); }
			}
			__syncthreads();
		} else {
#define EMPTYV 0xffffffffu
#define INITG(i, used) owner[i] = (used) ? 0u : EMPTYV
#define AMING(old, slot, r) old = atomicMin(&owner[slot], r)
#define LOADG(o, i) o = __hip_atomic_load(&owner[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
			RP_PLACE(INITG, AMING, LOADG)
#undef EMPTYV
		}
		cnt += batch; i0 += batch;
		RP_TICK(n >= 32768 ? 3 : 2)
	}
	__syncthreads();
	if (keys == lkeys) { for (u32 i = tid; i < n; i += blockDim.x) gkeys[i] = lkeys[i]; __syncthreads(); keys = gkeys; }
	/* publish: bitmap in the arena, unused slots normalised to YK_EMPTY */
	if (cur != UA) for (u32 w = tid; w < (n + 31) / 32; w += blockDim.x) UA[w] = cur[w];
	for (u32 i = tid; i < n; i += blockDim.x) if (!bm_get(cur, i)) keys[i] = YK_EMPTY;
	RP_TICK(4)
	if (tid == 0) { out_bits[blockIdx.x] = n ? bits : YK_NOCAP; out_count[blockIdx.x] = cnt; }
}

/* ==========================================================================================
 * K5 for LARGE sub-tables ("replay2").  k_replay gives one sub-table to one workgroup and walks global
 * arrays with random accesses and device atomics; that is fine up to a few thousand slots (everything in
 * LDS) and hopeless at a million (measured 2.7 s for 1024 x 2 Mi slots).  Two facts about khashl make
 * both steps of the replay LOCAL, streaming and free of global atomics:
 *
 * DOUBLING (khashl.h:171-189, n -> 2n slots).  A maximal run of used old slots (e, e') -- slots e and e'
 * unused -- holds keys whose old home lies inside the run, so their new homes lie in [2e + 2, 2e' - 1], and
 * for every h the run has at most e' - h keys with new home >= 2h: the run's keys end up inside the new
 * region [2e + 2, 2e' + 1] whatever the order, and no key of another run ever enters it.  Runs interact
 * only through the ORDER of re-insertion: the key at old slot s is kicked out early iff a key lands on
 * physical slot s before the scan reaches s.  Its processing time is therefore
 *     sigma(s) = (c, d + 1) if the final occupant x of new slot s has sigma(x) = (c, d) with c < s
 *                (s, 0)     otherwise,
 * and the final content of a region is first-come-first-served linear probing of its run's keys in sigma
 * order.  New slot s < 2F belongs to the region of a run below F (when slot F - 1 is unused), so once all
 * runs below F are placed, every run inside [F, 2F) can be placed at once, one lane per run, with plain
 * loads and stores on its own region; `tag` keeps sigma of the occupant of every new slot < n.  The first
 * few slots are done by the literal rule by one lane (chains followed while they stay in that prefix).
 *
 * PLACEMENT of the new keys of a stage (first come first served in rank order, khashl.h:197-221) is
 * ordered probing with the smallest rank winning a slot (order-free), done per segment of R2_SEG slots
 * with the ranks in LDS; the keys of the stage are first grouped by the segment of their home slot.  A walk
 * that leaves its segment is finished afterwards on the first R2_HEAD slots of the next segment, whose
 * ranks are kept in a global array for that purpose (device atomics, but only for these few walks).
 *
 * All sub-tables advance together, one kernel per step; the table of a sub-table alternates between two
 * buffers.  Anything unexpected (a run longer than the round, a walk longer than a head) raises `fail`
 * and the host falls back to k_replay.
 * ========================================================================================== */
#define R2_SEG_LOG 14
#define R2_SEG  (1u << R2_SEG_LOG)
#define R2_HEAD 2048u
#define R2_NONE 0xffffffffu
#define R2_SMALL_F 256u                       /* the single-workgroup kernel runs the doubling rounds up to here */
#define R2_BASE_MAX 4096u
#define R2_LONG 8u                             /* runs longer than this are placed by a whole wave */
#define R2_LMAX 512u
#define R2_RMAX (2 * R2_LMAX + 2 + 256)
#define R2_WS 512u                              /* k_r2_dsmall's LDS windows: old slots [0, R2_WS), new slots [0, R2_WD) */
#define R2_WD (2 * R2_WS + 64)
#define R2_MOVED (YK_EMPTY - 1)                 /* old slot whose key a chain of the prefix has re-inserted: still part of its run, no longer a key */

__device__ __forceinline__ u32 r2_home(u64 key, u32 bits) { return yk_h2b((u32)(key >> 10), bits); }

/* largest g in (F, x] such that old slot g - 1 is unused (g == n is always a boundary); 0 if there is none */
__device__ __forceinline__ u32 r2_boundary(const u64 *S, u32 F, u32 x, u32 n)
{
	if (x >= n) return n;
	u32 g = x;
	while (g > F && S[g - 1] != YK_EMPTY) --g;
	return g > F ? g : 0;
}

/* put `key`, whose processing time is (c, d), into the new table; while it lands on an unmoved key of the run [a, b)
 * that key is kicked out and follows at once with (c, d + 1) (khashl.h:183-187).  A key landing on an unmoved key of
 * ANOTHER run only leaves its tag: that run reads it when its round comes. */
__device__ __forceinline__ void r2_chain(u64 *S, u64 *D, u32 *TG, u64 key, u32 c, u32 d, u32 a, u32 b, u32 n, u32 nb)
{
	const u32 Nmask = 2 * n - 1;
	for (;;) {
		u32 p = r2_home(key, nb);
		while (D[p] != YK_EMPTY) p = (p + 1) & Nmask;
		D[p] = key;
		if (p >= n) return;
		TG[p] = c << 6 | (d < 63 ? d : 63);
		if (p < a || p >= b) return;
		const u64 v = S[p];
		if (v == YK_EMPTY || v == R2_MOVED) return;
		key = v; S[p] = R2_MOVED; ++d;
	}
}

/* place the run of used old slots starting at a (slot a - 1 is unused).  First the keys kicked out by keys that are
 * already in place (tag with c < slot), in (c, d) order -- c lies below the run, so they all come before the run's own
 * scan positions; then the scan.  COH: the tags were written by other waves of this kernel (in-kernel rounds). */
template <bool COH>
__device__ __forceinline__ bool r2_run(u64 *S, u64 *D, u32 *TG, u32 a, u32 n, u32 nb)      /* false: a long run, left to a wave */
{
	u32 b = a, nk = 0;
	for (; b < n && S[b] != YK_EMPTY; ++b) {
		if (b - a >= R2_LONG) return false;
		const u32 t = COH ? __hip_atomic_load(&TG[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : TG[b];
		nk += t != R2_NONE && (t >> 6) < b && S[b] != R2_MOVED;
	}
	u32 last = 0;                                                          /* tag + 1 of the last key taken */
	for (u32 it = 0; it < nk; ++it) {
		u32 best = R2_NONE, bs = a;
		for (u32 s = a; s < b; ++s) {
			if (S[s] == R2_MOVED) continue;
			const u32 t = COH ? __hip_atomic_load(&TG[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : TG[s];
			if (t == R2_NONE || (t >> 6) >= s) continue;
			if (t + 1 > last && t < best) { best = t; bs = s; }               /* tags order like (c, d) */
		}
		if (best == R2_NONE) break;                                           /* the others left with a chain */
		last = best + 1;
		const u64 key = S[bs];
		S[bs] = R2_MOVED;
		r2_chain(S, D, TG, key, best >> 6, (best & 63u) + 1, a, b, n, nb);
	}
	for (u32 s = a; s < b; ++s) {
		const u64 key = S[s];
		if (key == R2_MOVED) continue;
		S[s] = R2_MOVED;
		r2_chain(S, D, TG, key, s, 0, a, b, n, nb);
	}
	return true;
}

/* A long run is placed by a whole wave on LDS copies.  In a round all tags a run reads are final, its kicked keys
 * all precede its scan positions, and no chain continues inside it (its region starts at 2a >= b): first come first
 * served in sigma order is then ordered probing with priority = rank in sigma order, all keys at once.  DYN: the run
 * covers [F, 2F] and feeds its own slots (only below R2_SMALL_F): one lane walks it with the literal rule, on LDS. */
template <u32 LM> struct R2WaveT {                     /* LM: longest run the buffers hold; its region + the shared wrap-around head */
	static constexpr u32 LMAX = LM, RMAX = 2 * LM + 2 + 256;
	u64 keys[LM]; u32 sig[LM]; u32 own[2 * LM + 2 + 256]; unsigned short byrank[LM], rk[LM]; u32 mv[LM / 32];
};
typedef R2WaveT<R2_LMAX> R2Wave;
#define R2_MMAX 160u                                  /* "medium" runs: placed by the wave that met them, in the LDS of its chunk (k_r2_double) */
typedef R2WaveT<R2_MMAX> R2WaveM;

__device__ __forceinline__ void r2_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

template <bool COH, bool DYN, class WT>
__device__ __forceinline__ void r2_wave_run(WT &W, u64 *S, u64 *D, u32 *TG, u32 a, u32 n, u32 nb, u32 *fail)
{
	constexpr u32 R2_LMAX_ = WT::LMAX, R2_RMAX_ = WT::RMAX;
	const u32 lane = threadIdx.x & 63, Nmask = 2 * n - 1;
	u32 L = 0;
	for (;;) {                                                              /* length of the run */
		const u32 idx = a + L + lane;
		const u64 m = __ballot(idx < n && S[idx] != YK_EMPTY);
		if (m == ~0ull) { L += 64; if (L > R2_LMAX_) break; continue; }
		L += (u32)__ffsll((long long)~m) - 1;
		break;
	}
	u32 wrap = 0;
	if (a + L >= n) {                                                       /* the run that reaches the end of the table shares its region with the table's first run */
		u32 j0 = 0;
		while (j0 < 128 && S[j0] != YK_EMPTY) ++j0;
		wrap = 2 * j0 + 2;
	}
	const u32 R = 2 * L + 2 + wrap;
	if (L > R2_LMAX_ || R > R2_RMAX_) { if (lane == 0) *fail = 6; return; }
	for (u32 i = lane; i < L; i += 64) {
		const u32 sl = a + i;
		const u64 k = S[sl];
		const u32 t = COH ? __hip_atomic_load(&TG[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : TG[sl];
		W.keys[i] = k;
		W.sig[i] = k == R2_MOVED ? R2_NONE : (t != R2_NONE && (t >> 6) < sl) ? ((t & ~63u) | ((t & 63u) < 62 ? (t & 63u) + 1 : 63u)) : sl << 6;
	}
	for (u32 i = lane; i < R; i += 64) W.own[i] = D[(2 * a + i) & Nmask] != YK_EMPTY ? 0u : 0xffffffffu;
	if (lane < R2_LMAX_ / 32) W.mv[lane] = 0;
	r2_wave_sync();
	for (u32 i = lane; i < L; i += 64) {                                    /* rank in sigma order */
		const u32 g = W.sig[i];
		if (g == R2_NONE) continue;
		u32 r = 0;
		for (u32 j = 0; j < L; ++j) r += W.sig[j] < g;
		W.rk[i] = (unsigned short)r; W.byrank[r] = (unsigned short)i;
	}
	r2_wave_sync();
	if (!DYN) {
		for (u32 i = lane; i < L; i += 64) {
			if (W.sig[i] == R2_NONE) continue;
			u32 r = (u32)W.rk[i] + 1, li = (r2_home(W.keys[i], nb) - 2 * a) & Nmask;
			for (;;) {
				if (li >= R) { *fail = 7; break; }
				const u32 old = atomicMin(&W.own[li], r);
				if (old == 0xffffffffu) break;
				if (old > r) r = old;
				++li;
			}
		}
		r2_wave_sync();
		for (u32 i = lane; i < R; i += 64) { const u32 o = W.own[i]; if (o != 0 && o != 0xffffffffu) W.own[i] = (u32)W.byrank[o - 1] + 1; }
	} else if (lane == 0) {
		u32 nv = 0;
		for (u32 i = 0; i < L; ++i) nv += W.sig[i] != R2_NONE;
		auto chain = [&](u32 i, u32 c, u32 d) {
			for (;;) {
				W.mv[i >> 5] |= 1u << (i & 31);
				W.sig[i] = c << 6 | (d < 63 ? d : 63);
				u32 li = (r2_home(W.keys[i], nb) - 2 * a) & Nmask;
				while (li < R && W.own[li] != 0xffffffffu) ++li;
				if (li >= R) { *fail = 7; return; }
				W.own[li] = i + 1;
				const u32 slot = (2 * a + li) & Nmask;
				if (slot < a || slot >= a + L) return;
				const u32 v = slot - a;
				if (W.sig[v] == R2_NONE || (W.mv[v >> 5] >> (v & 31) & 1)) return;
				i = v; ++d;                                                  /* an unmoved key of the run sits there: it follows at once */
			}
		};
		for (u32 r = 0; r < nv; ++r) {                                       /* the keys kicked out from below come first, in sigma order */
			const u32 i = W.byrank[r];
			if (W.mv[i >> 5] >> (i & 31) & 1) continue;                     /* left with a chain already (its time has been overwritten) */
			const u32 g = W.sig[i];
			if ((g & 63u) == 0) break;
			chain(i, g >> 6, g & 63u);
		}
		for (u32 i = 0; i < L; ++i)
			if (W.sig[i] != R2_NONE && !(W.mv[i >> 5] >> (i & 31) & 1)) chain(i, a + i, 0);
	}
	r2_wave_sync();
	for (u32 i = lane; i < R; i += 64) {
		const u32 o = W.own[i];
		if (o == 0 || o == 0xffffffffu) continue;
		const u32 slot = (2 * a + i) & Nmask;
		D[slot] = W.keys[o - 1];
		if (slot < n) TG[slot] = W.sig[o - 1];
	}
	if (DYN) for (u32 i = lane; i < L; i += 64) if (W.sig[i] != R2_NONE) S[a + i] = R2_MOVED;
}

/* R2Tab: first slot in the two buffers, first sorted new key.  R2Act: kind 0 nothing, 1 place, 2 double; bits = log2 capacity before
 * the action; src = buffer (0/1) holding the table; seg0 = first entry of this sub-table in the per-segment arrays (yk_device.h) */

/* empty destination + no tags, for the sub-tables that double in this step */
__global__ __launch_bounds__(256)
void k_r2_dinit(const R2Tab *tabs, const R2Act *acts, u64 *K0, u64 *K1, u32 *TAG, u32 *OCC)
{
	const R2Act A = acts[blockIdx.y];
	if (A.kind != 2) return;
	const u64 off = tabs[blockIdx.y].off;
	const u64 n = 1ull << A.bits, N = 2 * n;
	u64 *D = (A.src ? K0 : K1) + off;
	u32 *TG = TAG + (off >> 1);
	u32 *OC = OCC + (off >> 4);                                       /* one bit per new slot: taken by a chain of the prefix (k_r2_dsmall) */
	for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < N; i += (u64)gridDim.x * 256) { D[i] = YK_EMPTY; if (i < n) TG[i] = R2_NONE; if (i < (N + 31) / 32) OC[i] = 0; }
}

/* the doubling rounds [F, G), G <= 2F, of ONE sub-table by one workgroup, until F reaches small_f.  COH: S / D / TG are the global
 * arrays (cross-wave data through L2: fences and agent-scope loads); else they are LDS windows holding slots [0, ws_lim) of the
 * old table and [0, 2 ws_lim + 64) of the new one -- every access of a round stays below 2G + 2 -- and the function returns early
 * (*s_dyn = 2) if a self-feeding run would leave the window */
template <bool COH>
__device__ u32 r2_small_rounds(u64 *S, u64 *D, u32 *TG, u32 F, const u32 n, const u32 nb, const u32 small_f, const u32 ws_lim,
                               R2Wave *waves, const u32 n_waves, u32 *s_long, u32 *s_F, u32 *s_dyn, u32 *s_nlong, u32 *fail)
{
	const u32 tid = threadIdx.x;
	while (F < n && F < small_f) {
		if (tid == 0) {
			u32 g = r2_boundary(S, F, 2 * F < n ? 2 * F : n, n);
			*s_dyn = 0; *s_nlong = 0;
			if (g == 0) {
				/* one run covers [F, 2F]: the keys landing on its slots come from the run itself; it is placed alone, by the
				 * literal rule in sigma order (r2_wave_run<.., true>), up to its own end */
				*s_dyn = 1;
				g = 2 * F;
				while (g < n && S[g] != YK_EMPTY) { ++g; if (!COH && g + 1 >= ws_lim) { *s_dyn = 2; break; } }
				g = g < n ? g + 1 : n;
			}
			*s_F = g;
		}
		__syncthreads();
		const u32 G = *s_F;
		if (*s_dyn == 2) break;                                              /* (uniform) */
		if (*s_dyn) { if (tid < 64) r2_wave_run<COH, true>(waves[0], S, D, TG, F, n, nb, fail); }
		else {
			for (u32 s = F + tid; s < G; s += blockDim.x)
				if (S[s] != YK_EMPTY && (s == F || S[s - 1] == YK_EMPTY) && !r2_run<COH>(S, D, TG, s, n, nb)) {
					const u32 at = atomicAdd(s_nlong, 1u);
					if (at < 256) s_long[at] = s; else *fail = 8;
				}
			__syncthreads();
			const u32 nl = *s_nlong < 256 ? *s_nlong : 256;
			if (tid < 64 * n_waves) for (u32 j = tid >> 6; j < nl; j += n_waves) r2_wave_run<COH, false>(waves[tid >> 6], S, D, TG, s_long[j], n, nb, fail);
		}
		if (COH) __threadfence();
		__syncthreads();
		F = G;
	}
	return F;
}

/* the prefix by the literal rule, then the rounds below R2_SMALL_F; one workgroup per sub-table */
__global__ __launch_bounds__(256)
void k_r2_dsmall(const R2Tab *tabs, const R2Act *acts, u64 *K0, u64 *K1, u32 *TAG, u32 *OCC, u32 *Fcur, u32 *Gcur, u32 *fail, u32 small_f, int defer)
{
	__shared__ R2Wave s_wave[1];
	__shared__ u64 s_S[R2_WS], s_D[R2_WD];                              /* 31 KB in all: the 1024 workgroups of a launch are resident at once */
	__shared__ u32 s_T[R2_WD];
	__shared__ u32 s_long[256];
	__shared__ u32 s_F, s_dyn, s_nlong;
	const u32 p = blockIdx.x, tid = threadIdx.x;
	const R2Act A = acts[p];
	if (A.kind != 2) return;
	const u64 off = tabs[p].off;
	const u32 n = 1u << A.bits, nb = A.bits + 1, Nmask = 2 * n - 1;
	u64 *S = (A.src ? K1 : K0) + off;
	u64 *D = (A.src ? K0 : K1) + off;
	u32 *TG = TAG + (off >> 1);
	u32 *OC = OCC + (off >> 4);
	const u32 ns = n < R2_WS ? n : R2_WS, nd = 2 * n < R2_WD ? 2 * n : R2_WD, nt = n < R2_WD ? n : R2_WD;
	__shared__ u32 s_bail;
	/* The literal rule (khashl.h:171-189) for the scan positions below F0.  A kicked-out key leaves a tombstone: its slot still belongs to
	 * its run.  A chain is followed while it stays inside the prefix (whose slots feed each other) and wherever it meets the run that may
	 * touch the end of the table (that run shares its region with the table's first run: its kicked keys must be in place, in the
	 * reference's order, before the first run's later keys).  Anywhere else the key that lands on an unmoved key of another run just leaves
	 * its tag, exactly as in the rounds (r2_chain): that run takes its kicked keys first, in (c, d) order, when its round comes -- the chain's
	 * remaining steps (one dependent global access each, ~log2 n of them) are not walked by this one lane.  defer == 0: every chain to its end.
	 * With deferred chains on a table of >= 2048 slots everything the prefix touches lies in the first few hundred slots of both tables: the
	 * lane walks LDS copies (the new table and its tags are fresh from k_r2_dinit: no load), the OCC marks are applied afterwards.  An access
	 * that leaves the windows after all (a long first run) drops the copies, nothing global having been written, and the lane starts over
	 * on the arrays */
	bool have_win = false;
	if (defer && n >= 2048) {
		for (u32 i = tid; i < ns; i += 256) s_S[i] = __hip_atomic_load(&S[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		for (u32 i = tid; i < nd; i += 256) s_D[i] = YK_EMPTY;
		for (u32 i = tid; i < nt; i += 256) s_T[i] = R2_NONE;
		if (tid == 0) { s_bail = 0; s_nlong = 0; }
		__syncthreads();
		if (tid == 0) {
			u32 F0 = 8, nl = 0;
			bool bail = false;
			while (F0 < ns && s_S[F0 - 1] != YK_EMPTY) ++F0;
			if (F0 >= ns) bail = true;
			for (u32 j = 0; j < F0 && !bail; ++j) {
				u64 key = s_S[j];
				if (key == YK_EMPTY || key == R2_MOVED) continue;
				u32 d = 0;
				s_S[j] = R2_MOVED;
				for (;;) {
					u32 q = r2_home(key, nb);
					while (q < nd && s_D[q] != YK_EMPTY) ++q;                 /* (no wrap-around this far from the end of the table) */
					if (q >= nd || q >= nt || nl >= 256) { bail = true; break; }
					s_D[q] = key;
					s_long[nl++] = q;
					s_T[q] = j << 6 | (d < 63 ? d : 63);
					if (q >= F0) break;                                         /* the run of slot q reads the tag in its round (q is far below the table's last run) */
					const u64 v = s_S[q];
					if (v == YK_EMPTY || v == R2_MOVED) break;
					key = v; s_S[q] = R2_MOVED; ++d;                           /* an unmoved key of the prefix sits there: kick it out */
				}
			}
			if (bail) s_bail = 1;
			else { for (u32 i = 0; i < nl; ++i) { const u32 q = s_long[i]; OC[q >> 5] |= 1u << (q & 31); } s_F = F0; }
		}
		__syncthreads();
		have_win = s_bail == 0;
	}
	if (!have_win) {
		if (tid == 0) {
			u32 F0 = n < 8 ? n : 8;
			while (F0 < n && S[F0 - 1] != YK_EMPTY) ++F0;               /* slot F0 - 1 unused (or F0 == n) */
			if (F0 > R2_BASE_MAX && F0 < n) *fail = 1;
			const u32 tail0 = n > R2_LMAX + 2 ? n - (R2_LMAX + 2) : 0;     /* a run reaching slot n - 1 starts behind this slot (longer runs are refused) */
			for (u32 j = 0; j < F0; ++j) {
				u64 key = S[j];
				if (key == YK_EMPTY || key == R2_MOVED) continue;
				u32 d = 0;
				S[j] = R2_MOVED;
				for (;;) {
					u32 q = r2_home(key, nb);
					while (D[q] != YK_EMPTY) q = (q + 1) & Nmask;
					D[q] = key;
					OC[q >> 5] |= 1u << (q & 31);
					if (q >= n) break;
					TG[q] = j << 6 | (d < 63 ? d : 63);
					if (defer && q >= F0 && q < tail0) break;                /* the run of slot q reads the tag in its round */
					const u64 v = S[q];
					if (v == YK_EMPTY || v == R2_MOVED) break;
					key = v; S[q] = R2_MOVED; ++d;                          /* an unmoved key sits there: kick it out */
				}
			}
			s_F = F0;
		}
		__threadfence();
		__syncthreads();
	}
	u32 F = s_F;
	const bool do_small = small_f <= R2_WS / 2 && F < small_f && F < n;
	if (have_win || do_small) {
		/* the rounds below small_f touch old slots < 2 small_f and new slots < 4 small_f + 2 only: they run on LDS copies (a dependent
		 * access costs ~100 cycles there instead of a trip to L2), the windows go back to the arrays afterwards */
		if (!have_win) {
			for (u32 i = tid; i < ns; i += 256) s_S[i] = __hip_atomic_load(&S[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (u32 i = tid; i < nd; i += 256) s_D[i] = __hip_atomic_load(&D[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (u32 i = tid; i < nt; i += 256) s_T[i] = __hip_atomic_load(&TG[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		__syncthreads();
		if (do_small) F = r2_small_rounds<false>(s_S, s_D, s_T, F, n, nb, small_f, R2_WS, s_wave, 1, s_long, &s_F, &s_dyn, &s_nlong, fail);
		__syncthreads();
		for (u32 i = tid; i < ns; i += 256) S[i] = s_S[i];
		for (u32 i = tid; i < nd; i += 256) D[i] = s_D[i];
		for (u32 i = tid; i < nt; i += 256) TG[i] = s_T[i];
		__threadfence();
		__syncthreads();
	}
	F = r2_small_rounds<true>(S, D, TG, F, n, nb, small_f, 0, s_wave, 1, s_long, &s_F, &s_dyn, &s_nlong, fail);   /* whatever is left (a run that left the window; small_f beyond the window) */
	if (tid == 0) Fcur[p] = F;                                                /* k_r2_double goes on from here */
	(void)Gcur;
}

/* ------------------------------------------------------------------------------------------
 * The whole doubling of a sub-table in ONE launch, one workgroup per sub-table (k_r2_double).  Nothing leaves the workgroup, so the
 * rounds [F, G) follow each other behind workgroup barriers instead of kernel boundaries (k_r2_dsmall + ~11 x (k_r2_dround + k_r2_long)
 * took 25 launches per doubling, the big ones bound by the dispatcher, the small ones by launch latency).  Inside a round the waves
 * take the chunks of F2_CH old slots in turn, each on LDS buffers of its own (wave-level synchronisation only): the chunk routine of
 * k_r2_dround with a window for runs of up to F2_CHL slots; a longer run is placed by the wave that met it right after the chunk, in
 * the same LDS (up to R2_MMAX slots), and the few beyond that by wave 0 at the end of the round (up to R2_LMAX).  The boundary of the
 * next round is looked up by wave 0 while the others already place.
 * ------------------------------------------------------------------------------------------ */
#define F2_CH  128u
#define F2_CHL 64u
#define F2_CHX (F2_CHL + 8)
#define F2_NA  (F2_CH + F2_CHX + 1)
#define F2_WN  (2 * (F2_CH + F2_CHX) + 2)
#define F2_NOC (F2_WN / 32 + 2)
#define F2_LST 24u
struct R2Chunk {
	union {
		struct { u64 key[F2_NA + 1]; u64 win[F2_WN]; u32 oc[F2_NOC]; } c;
		R2WaveM m;                                                             /* the same memory while a medium run is placed */
	} u;
	u32 lst[F2_LST]; u32 nlst, pad;                                            /* runs of this chunk left to the wave routine */
};

/* largest g in (F, x] such that old slot g - 1 is unused (g == n is always a boundary); 0 if there is none.  One wave, 64 slots per step */
__device__ __forceinline__ u32 r2_boundary_wave(const u64 *S, u32 F, u32 x, u32 n)
{
	const u32 lane = threadIdx.x & 63;
	if (x >= n) return n;
	for (;;) {
		const u32 g = x - lane;                                                /* candidates x, x - 1, ..., x - 63 */
		const bool hit = x >= lane && g > F && S[g - 1] == YK_EMPTY;
		const u64 m = __ballot(hit);
		if (m) return x - ((u32)__ffsll((long long)m) - 1);
		if (x < 64 + F + 1) return 0;
		x -= 64;
	}
}

/* LDS traffic of ONE wave is served in issue order: a compiler barrier is all that lies between a phase's writes and the next phase's
 * reads -- no s_waitcnt, so the global loads requested for the next chunk stay in flight (a fence or __syncthreads() here waits vmcnt(0):
 * gfx950 counts loads and stores on one counter) */
__device__ __forceinline__ void r2_lds_phase() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

/* the chunks c0 = F + F2_CH (wave + n_waves i) of the round [F, G) that belong to this wave.  Window index x <-> old slot c0 - 1 + x; lane l
 * holds the entries x = l + 64 j.  The runs of the window come from ballots (a 64-bit "unused" mask per j), not from LDS scans; a lane keeps
 * its own keys and tags in registers, LDS holds the keys (looked up by window index at write-back) and the window of the new table */
template <bool PROF>
__device__ __forceinline__ void r2_round_wave(R2Chunk &C, const u64 *S, u64 *D, u32 *TG, const u32 *OC, const u32 F, const u32 G, const u32 n, const u32 nb,
                                              const u32 wave, const u32 n_waves, u32 *s_big, u32 *s_nbig, u32 *fail, u64 *pf, u64 &tq)
{
	const u32 lane = threadIdx.x & 63;
#define R2W_LAP(i) if (PROF) { const u64 t_ = wall_clock64(); pf[i] += t_ - tq; tq = t_; }
	constexpr u32 PER = (F2_NA + 63) / 64;
	static_assert(PER == 4, "four 64-bit masks cover the window");
	u64 rk[PER]; u32 rt[PER], roc = 0;
	auto fetch = [&](const u32 c0) {
#pragma unroll
		for (u32 j = 0; j < PER; ++j) {
			const u32 sl = c0 - 1 + lane + 64 * j;                              /* c0 >= F >= 8 */
			const u32 cl = sl < n ? sl : n - 1;
			rk[j] = S[cl]; rt[j] = __hip_atomic_load(&TG[cl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		roc = OC[((2 * c0) >> 5) + (lane < F2_NOC ? lane : 0)];
	};
	u32 c0 = F + wave * F2_CH;
	if (c0 < G) fetch(c0);
	while (c0 < G) {
		const u32 w0 = 2 * c0;
		u64 kc[PER], E[PER]; u32 tc[PER];
#pragma unroll
		for (u32 j = 0; j < PER; ++j) {
			const u32 x = lane + 64 * j, sl = c0 - 1 + x;
			kc[j] = (x < F2_NA && sl < n) ? rk[j] : YK_EMPTY; tc[j] = rt[j];
			if (x >= F2_NA) kc[j] = R2_MOVED;                                   /* behind the window: neither a key nor a gap */
			if (x < F2_NA) C.u.c.key[x] = kc[j];
			E[j] = __ballot(kc[j] == YK_EMPTY);
		}
		const u32 oc_cur = roc;
		const bool any_oc = __ballot(lane < F2_NOC && oc_cur != 0) != 0;      /* slots a chain of the prefix took: only near the two ends of the table */
		if (any_oc && lane < F2_NOC) C.u.c.oc[lane] = oc_cur;
		if (lane == 0) C.nlst = 0;
		const u32 c1 = c0 + n_waves * F2_CH;
		if (c1 < G) fetch(c1);
		r2_lds_phase();
		R2W_LAP(0)
		for (u32 i = lane; i < F2_WN; i += 64) {
			u64 v = ~0ull;
			if (any_oc) { const u32 q = w0 + i, b = (w0 & 31) + i; if (q < 2 * n && (C.u.c.oc[b >> 5] >> (b & 31) & 1)) v = 0ull; }
			C.u.c.win[i] = v;
		}
		r2_lds_phase();
		R2W_LAP(1)
		const u32 lim = (G < c0 + F2_CH ? G : c0 + F2_CH) - c0;              /* runs start in [c0, c0 + lim) */
#pragma unroll
		for (u32 j = 0; j < PER; ++j) {
			const u32 x = lane + 64 * j;
			const u64 key = kc[j];
			if (x == 0 || x >= F2_NA || key == YK_EMPTY) continue;
			/* last unused slot before x, next unused slot behind x (window indices; -1 / F2_NA: none) */
			int le = -1, ne = (int)F2_NA;
			{
				const u64 below = lane == 63 ? ~0ull : (2ull << lane) - 1;      /* bits 0 .. lane */
				bool got = false;
#pragma unroll
				for (int jj = (int)PER - 1; jj >= 0; --jj) {
					if (jj > (int)j || got) continue;
					const u64 m = jj == (int)j ? E[jj] & below : E[jj];
					if (m) { le = 64 * jj + 63 - __clzll((long long)m); got = true; }
				}
				got = false;
#pragma unroll
				for (int jj = 0; jj < (int)PER; ++jj) {
					if (jj < (int)j || got) continue;
					const u64 m = jj == (int)j ? E[jj] & ~below : E[jj];
					if (m) { ne = 64 * jj + (__ffsll((long long)m) - 1); got = true; }
				}
			}
			if (le < 0 || (u32)le >= lim) continue;                             /* its run starts before / behind this chunk */
			const u32 L = (u32)(ne - le - 1), a = c0 + (u32)le;
			if (ne >= (int)F2_NA || L > F2_CHL || a + L >= n) {                 /* too long for the window, or it reaches the end of the table: the wave routine */
				if ((int)x == le + 1) {
					const u32 at = atomicAdd(&C.nlst, 1u);
					if (at < F2_LST) C.lst[at] = a; else *fail = 8;
				}
				continue;
			}
			if (key == R2_MOVED) continue;
			const u32 sl = c0 - 1 + x, t = tc[j];
			const u32 sig = (t != R2_NONE && (t >> 6) < sl) ? ((t & ~63u) | ((t & 63u) < 62 ? (t & 63u) + 1 : 63u)) : sl << 6;
			u64 e = (u64)(sig + 1) << 32 | x;
			u32 q = r2_home(key, nb) - w0;
			for (;;) {
				if (q >= F2_WN) { *fail = 9; break; }
				const u64 old = atomicMin((unsigned long long*)&C.u.c.win[q], (unsigned long long)e);
				if (old == ~0ull) break;
				if (old > e) e = old;                                           /* we took the slot; carry the displaced later key on */
				++q;
			}
		}
		r2_lds_phase();
		R2W_LAP(2)
		for (u32 i = lane; i < F2_WN; i += 64) {
			const u64 e = C.u.c.win[i];
			if (e == ~0ull || (e >> 32) == 0) continue;
			const u32 q = w0 + i;
			D[q] = C.u.c.key[(u32)e];
			if (q < n) TG[q] = (u32)(e >> 32) - 1;
		}
		r2_lds_phase();
		R2W_LAP(3)
		/* the runs this chunk left out: every run of a round is independent of the others, so they are placed now, in the chunk's own LDS */
		const u32 nl = C.nlst < F2_LST ? C.nlst : F2_LST;
		for (u32 j = 0; j < nl; ++j) {
			const u32 a = C.lst[j];
			u32 L = 0;
			for (;;) {
				const u32 idx = a + L + lane;
				const u64 m = __ballot(idx < n && S[idx] != YK_EMPTY);
				if (m == ~0ull) { L += 64; if (L > R2_MMAX) break; continue; }
				L += (u32)__ffsll((long long)~m) - 1;
				break;
			}
			if (L <= R2_MMAX) r2_wave_run<true, false>(C.u.m, (u64*)S, D, TG, a, n, nb, fail);
			else if (lane == 0) { const u32 at = atomicAdd(s_nbig, 1u); if (at < 64) s_big[at] = a; else *fail = 8; }
			r2_wave_sync();
		}
		R2W_LAP(4)
		c0 = c1;
	}
}

template <int NW, bool PROF>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 6 && !PROF ? 6 : 4)))      /* NW = 6: <= 80 VGPRs, four workgroups per CU = 24 waves */
void k_r2_double(const R2Tab *tabs, const R2Act *acts, u64 *K0, u64 *K1, u32 *TAG, const u32 *OCC, const u32 *Fin, u32 *Fout, u32 *fail, u64 *prof)
{
	u64 pf[PROF ? 8 : 1] = { 0 }, tq = PROF ? wall_clock64() : 0;              /* PROF (YAKAMD_VERBOSE > 1): 100 MHz ticks per phase, lane 0 of every wave */
#define R2F_LAP(i) if (PROF) { const u64 t_ = wall_clock64(); pf[i] += t_ - tq; tq = t_; }
	__shared__ union { R2Chunk ch[NW]; R2Wave big; } U;
	__shared__ u32 s_big[64];
	__shared__ u32 s_G, s_F, s_nbig;
	const u32 p = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const R2Act A = acts[p];
	if (A.kind != 2) return;
	const u64 off = tabs[p].off;
	const u32 n = 1u << A.bits, nb = A.bits + 1;
	u64 *S = (A.src ? K1 : K0) + off;
	u64 *D = (A.src ? K0 : K1) + off;
	u32 *TG = TAG + (off >> 1);
	const u32 *OC = OCC + (off >> 4);
	u32 F = Fin[p];                                                           /* k_r2_dsmall did the prefix and the rounds below small_f */
	if (tid == 0) s_nbig = 0;
	if (wave == 0 && F < n) { const u32 g = r2_boundary_wave(S, F, 2 * F < n ? 2 * F : n, n); if (lane == 0) s_G = g; }
	__syncthreads();
	while (F < n) {
		const u32 G = s_G;
		if (G == 0) {
			/* (uniform) no unused slot in (F, 2F]: one run covers [F, 2F] and feeds its own slots.  It is placed alone, by the literal rule in
			 * sigma order on LDS copies (r2_wave_run<.., true>, as in k_r2_dsmall's rounds), up to its own end */
			__syncthreads();
			if (wave == 0) {
				u32 g = 2 * F;
				for (;;) {                                                         /* first unused slot from 2F on */
					const u32 idx = g + lane;
					const u64 m = __ballot(idx >= n || S[idx] == YK_EMPTY);
					if (m) { g += (u32)__ffsll((long long)m) - 1; break; }
					g += 64;
				}
				const u32 G2 = g < n ? g + 1 : n;
				r2_wave_run<true, true>(U.big, S, D, TG, F, n, nb, fail);
				r2_wave_sync();
				const u32 gn = G2 < n ? r2_boundary_wave(S, G2, 2 * G2 < n ? 2 * G2 : n, n) : n;
				if (lane == 0) { s_F = G2; s_G = gn; }
			}
			__syncthreads();
			F = s_F;
			continue;
		}
		__syncthreads();                                                       /* everybody has read s_G */
		if (wave == 0 && G < n) { const u32 g = r2_boundary_wave(S, G, 2 * G < n ? 2 * G : n, n); if (lane == 0) s_G = g; }   /* used slots stay used: the next boundary can be looked up now */
		r2_round_wave<PROF>(U.ch[wave], S, D, TG, OC, F, G, n, nb, wave, NW, s_big, &s_nbig, fail, pf, tq);
		R2F_LAP(5)
		__syncthreads();                                                       /* workgroup scope is all it takes: the sub-table never leaves this workgroup (an agent-scope fence here writes the L2 back, once per round and workgroup) */
		R2F_LAP(6)
		if (s_nbig) {                                                          /* (uniform) runs beyond R2_MMAX slots: wave 0, one after the other */
			if (wave == 0) {
				const u32 nbg = s_nbig < 64 ? s_nbig : 64;
				for (u32 j = 0; j < nbg; ++j) { r2_wave_run<true, false>(U.big, S, D, TG, s_big[j], n, nb, fail); r2_wave_sync(); }
				if (lane == 0) s_nbig = 0;
			}
			__syncthreads();
			R2F_LAP(7)
		}
		F = G;
	}
	if (tid == 0) { Fout[p] = F; if (F < n) *fail = 10; }                     /* did not reach the end of the table: the host replays with k_replay */
	if (PROF && lane == 0) { for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&prof[i], (unsigned long long)pf[i]); atomicAdd((unsigned long long*)&prof[8], 1ull); }
}

/* keys of a stage grouped by the segment of their home slot: pk/pr[rec_off + i0 + ...], seg_start[seg0 + s] relative to the stage's first key */
__global__ __launch_bounds__(1024)
void k_r2_ppart(const R2Tab *tabs, const R2Act *acts, const u64 *__restrict__ kc, u64 *__restrict__ pk, u32 *__restrict__ pr, u32 *seg_start, u32 SEGLOG)
{
	__shared__ u32 s_cnt[1024];
	__shared__ u32 s_w[16];
	const u32 p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const R2Act A = acts[p];
	if (A.kind != 1 || A.bits <= SEGLOG) return;
	const u32 nseg = 1u << (A.bits - SEGLOG);                    /* <= 1024 */
	const u64 base = tabs[p].rec_off + A.i0;
	s_cnt[tid] = 0;
	__syncthreads();
	for (u32 q = tid; q < A.batch; q += 1024) atomicAdd(&s_cnt[r2_home(kc[base + q], A.bits) >> SEGLOG], 1u);
	__syncthreads();
	const u32 c = s_cnt[tid];
	u32 incl = c;
	for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o); if (lane >= (u32)o) incl += t; }
	if (lane == 63) s_w[wave] = incl;
	__syncthreads();
	u32 ex = incl - c;
	for (u32 w = 0; w < wave; ++w) ex += s_w[w];
	if (tid < nseg) seg_start[A.seg0 + tid] = ex;
	if (tid == nseg - 1) seg_start[A.seg0 + nseg] = ex + c;
	__syncthreads();
	s_cnt[tid] = ex;
	__syncthreads();
	for (u32 q = tid; q < A.batch; q += 1024) {
		const u64 key = kc[base + q];
		const u32 d = atomicAdd(&s_cnt[r2_home(key, A.bits) >> SEGLOG], 1u);
		pk[base + d] = key; pr[base + d] = q + 1;
	}
}

/* ordered probing of one segment's keys with the ranks in LDS */
__global__ __launch_bounds__(1024)
void k_r2_place(const R2Tab *tabs, const R2Act *acts, u64 *K0, u64 *K1, const u64 *__restrict__ kc, const u64 *__restrict__ pk, const u32 *__restrict__ pr,
                const u32 *__restrict__ seg_start, u32 *head, u64 *spill, u32 *spill_n, u32 spill_cap, u32 *fail, u32 SEGLOG, u32 HEAD)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_own[];
	const u32 p = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
	const R2Act A = acts[p];
	if (A.kind != 1) return;
	const u32 n = 1u << A.bits, nseg = A.bits > SEGLOG ? 1u << (A.bits - SEGLOG) : 1;
	if (seg >= nseg) return;
	const u32 L = nseg > 1 ? 1u << SEGLOG : n, start = seg * L;
	u64 *keys = (A.src ? K1 : K0) + tabs[p].off;
	const u64 base = tabs[p].rec_off + A.i0;
	for (u32 i = tid; i < L; i += blockDim.x) s_own[i] = keys[start + i] != YK_EMPTY ? 0u : 0xffffffffu;
	__syncthreads();
	const u32 q0 = nseg > 1 ? seg_start[A.seg0 + seg] : 0, q1 = nseg > 1 ? seg_start[A.seg0 + seg + 1] : A.batch;
	for (u32 q = q0 + tid; q < q1; q += blockDim.x) {
		u32 r, li;
		if (nseg > 1) { r = pr[base + q]; li = r2_home(pk[base + q], A.bits) - start; }
		else { r = q + 1; li = r2_home(kc[base + q], A.bits); }
		for (;;) {
			const u32 old = atomicMin(&s_own[li], r);
			if (old == 0xffffffffu) break;
			if (old > r) r = old;                                      /* we took the slot; carry the displaced later key on */
			++li;
			if (nseg == 1) { li &= n - 1; continue; }
			if (li == L) {                                             /* the walk goes on in the next segment's head */
				const u32 at = atomicAdd(spill_n, 1u);
				if (at < spill_cap) spill[at] = (u64)p << 48 | (u64)((seg + 1) & (nseg - 1)) << 32 | r; else *fail = 4;
				break;
			}
		}
	}
	__syncthreads();
	const u64 *src = kc + base;
	const u32 h0 = nseg > 1 ? HEAD : 0;
	u32 *hd = head + (size_t)(A.seg0 + seg) * HEAD;
	for (u32 i = tid; i < L; i += blockDim.x) {
		const u32 o = s_own[i];
		if (i < h0) hd[i] = o;                                          /* finished by k_r2_spill / k_r2_headfill */
		else if (o != 0 && o != 0xffffffffu) keys[start + i] = src[o - 1];
	}
}

__global__ __launch_bounds__(256)
void k_r2_spill(const R2Act *acts, const u64 *spill, const u32 *spill_n, u32 spill_cap, u32 *head, u32 *fail, u32 HEAD)
{
	const u32 ns = *spill_n < spill_cap ? *spill_n : spill_cap;
	for (u32 j = blockIdx.x * 256 + threadIdx.x; j < ns; j += gridDim.x * 256) {
		const u64 e = spill[j];
		const u32 p = (u32)(e >> 48), seg = (u32)(e >> 32) & 0xffffu;
		u32 r = (u32)e;
		u32 *hd = head + (size_t)(acts[p].seg0 + seg) * HEAD;
		u32 li = 0;
		for (;;) {
			const u32 old = atomicMin(&hd[li], r);
			if (old == 0xffffffffu) break;
			if (old > r) r = old;
			if (++li == HEAD) { *fail = 5; break; }
		}
	}
}

__global__ __launch_bounds__(256)
void k_r2_headfill(const R2Tab *tabs, const R2Act *acts, u64 *K0, u64 *K1, const u64 *__restrict__ kc, const u32 *head, u32 SEGLOG, u32 HEAD)
{
	const u32 p = blockIdx.y, seg = blockIdx.x;
	const R2Act A = acts[p];
	if (A.kind != 1 || A.bits <= SEGLOG || seg >= 1u << (A.bits - SEGLOG)) return;
	u64 *keys = (A.src ? K1 : K0) + tabs[p].off + ((u64)seg << SEGLOG);
	const u64 *src = kc + tabs[p].rec_off + A.i0;
	const u32 *hd = head + (size_t)(A.seg0 + seg) * HEAD;
	for (u32 i = threadIdx.x; i < HEAD; i += 256) {
		const u32 o = __hip_atomic_load(&hd[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (o != 0 && o != 0xffffffffu) keys[i] = src[o - 1];
	}
}

/* table of a sub-table into one of the two buffers: a copy of `n` slots of src (unused slots hold YK_EMPTY) or an empty table */
__global__ __launch_bounds__(256)
void k_r2_load(const R2Tab *tabs, const R2Load *ld, const u64 *__restrict__ src1, const u64 *__restrict__ src2, u64 *K0, u64 *K1)
{
	const R2Load Ld = ld[blockIdx.y];
	if (Ld.bits == YK_NOCAP) return;
	u64 *D = (Ld.dst ? K1 : K0) + tabs[blockIdx.y].off;
	const u64 *src = Ld.from_src == 2 ? src2 : src1;
	const u64 n = 1ull << Ld.bits;
	for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) D[i] = Ld.from_src ? src[Ld.src_off + i] : YK_EMPTY;
}

/* does a put-call follow the last new key of the sub-table (khashl.h:202 on an existing key: the trailing doubling)? */
__global__ void k_r2_trail(const u64 *lastput, const u64 *rec_t, const u64 *rec_off, const u32 *m, int P, u32 *out)
{
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= P) return;
	const u64 lp = lastput[p];
	out[p] = lp != 0 && (m[p] == 0 || lp - 1 > rec_t[rec_off[p] + m[p] - 1]);
}

/* final table -> arena + bitmap (one lane per slot; a wave's ballot is two bitmap words) */
__global__ __launch_bounds__(256)
void k_r2_publish(const R2Tab *tabs, const R2Pub *pub, const u64 *K0, const u64 *K1, u64 *__restrict__ nk, u32 *__restrict__ nu)
{
	const R2Pub Pb = pub[blockIdx.y];
	if (Pb.bits == YK_NOCAP) return;
	const u64 *S = (Pb.src ? K1 : K0) + tabs[blockIdx.y].off;
	const u64 n = 1ull << Pb.bits;
	for (u64 i0 = (u64)blockIdx.x * 256; i0 < n; i0 += (u64)gridDim.x * 256) {
		const u64 i = i0 + threadIdx.x;
		const u64 k = i < n ? S[i] : YK_EMPTY;
		if (i < n) nk[Pb.new_off + i] = k;
		const u64 b = __ballot(k != YK_EMPTY);
		if ((threadIdx.x & 63) == 0 && i < n) { nu[(Pb.new_off + i) >> 5] = (u32)b; if (i + 32 < n) nu[((Pb.new_off + i) >> 5) + 1] = (u32)(b >> 32); }
	}
}

/* ------------------------------------------------------------------------------------------
 * shrink (reference htab.c:180-197): keys with min <= count <= max, in ascending OLD slot order
 * ------------------------------------------------------------------------------------------ */
/* which == 0: count range only (yak_ch_shrink); 1: and absent from `other` (yak_ch_subtract,
 * htab.c:287-316); 2: and present in `other` (yak_ch_isec, htab.c:318-347) */
__device__ __forceinline__ bool shrink_keep(const ImgView &img, u64 a, int cmin, int cmax, int which, const ImgView &other, u32 p)
{
	if (!(img.used[a >> 5] >> (a & 31) & 1)) return false;
	const u64 kc = img.keys[a];
	const int c = (int)(kc & 1023);
	if (c < cmin || c > cmax) return false;
	if (which == 0) return true;
	const bool present = img_find(other, (kc >> 10) << img.pre | p) >= 0;
	return which == 1 ? !present : present;
}

__global__ __launch_bounds__(256)
void k_shrink_count(ImgView img, int cmin, int cmax, int which, ImgView other, u32 *seg_cnt)
{
	__shared__ u32 s_tot;
	const u32 bits = img.bits[blockIdx.x];
	if (threadIdx.x == 0) s_tot = 0;
	__syncthreads();
	if (bits != YK_NOCAP) {
		const u64 off = img.off[blockIdx.x];
		u32 c = 0;
		for (u32 i = threadIdx.x; i < 1u << bits; i += 256) c += shrink_keep(img, off + i, cmin, cmax, which, other, blockIdx.x);
		for (int o = 32; o; o >>= 1) c += __shfl_down(c, o);
		if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_tot, c);
	}
	__syncthreads();
	if (threadIdx.x == 0) seg_cnt[blockIdx.x] = s_tot;
}

__global__ __launch_bounds__(256)
void k_shrink_scatter(ImgView img, int cmin, int cmax, int which, ImgView other, const u64 *seg_off, u64 *rec_kc)
{
	__shared__ u32 s_w[4];
	const u32 bits = img.bits[blockIdx.x];
	if (bits == YK_NOCAP) return;
	const u64 off = img.off[blockIdx.x];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	u64 out = seg_off[blockIdx.x];
	for (u32 base = 0; base < 1u << bits; base += 256) {
		const u32 i = base + threadIdx.x;
		const bool keep = i < 1u << bits && shrink_keep(img, off + i, cmin, cmax, which, other, blockIdx.x);
		const u64 b = __ballot(keep);
		if (lane == 0) s_w[wave] = __popcll(b);
		__syncthreads();
		u32 pre = 0, tot = 0;
		for (int w = 0; w < 4; ++w) { if (w < wave) pre += s_w[w]; tot += s_w[w]; }
		if (keep) rec_kc[out + pre + __popcll(b & lanemask_lt())] = img.keys[off + i];
		out += tot;
		__syncthreads();
	}
}

__global__ __launch_bounds__(256)
void k_fill_u64(u64 *p, u64 v, u64 n)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

/* ==========================================================================================
 * FAST PATH: exclusive-ownership counting.
 * Device atomics on gfx950 execute at the memory side whatever their scope, ~10 G/s at best, and a
 * pass needs 1-3 of them per k-mer instance.  So the instances of a pass are radix-partitioned
 * twice -- by sub-table prefix (k_xpart / k_rpart), then into sub-buckets small enough for an
 * LDS hash table (k_part2) -- and ONE workgroup then owns a sub-bucket outright: it counts its
 * k-mers with LDS atomics only, owns the bloom blocks they map to (so the gate of bbf.c:25-42 /
 * htab.c:63-65 is evaluated per 512-bit block, sequentially in stream order, exactly as the
 * reference does), and emits just the keys that enter the table.  No global atomics per instance.
 * ========================================================================================== */
__device__ __forceinline__ u32 sub_of(u64 h, const FastParams &fp)
{
	const u64 x = h >> fp.pre;
	if (fp.s2_bits == 0) return 0;
	if (fp.bloom_mode) {                      /* sub-bucket = a contiguous range of bloom blocks */
		const int bb = fp.nb - 9;
		return (u32)((x & ((1ull << bb) - 1)) >> (bb - fp.s2_bits));
	}
	return (u32)x & ((1u << fp.s2_bits) - 1);                          /* yak_hash64 mixes well: any bit slice is uniform */
}

/* ---- the two 8-byte record formats of the fast path (yk_device.h YK_R8_*, FastParams.rec8_*) ---- */
/* rank of tagged record i in its bucket's stream: records of one round are contiguous and share the toggle bit; inside the round
 * the position field orders them.  `src` points at the chunk's first record; [-before, n + after) is the bucket */
__device__ __forceinline__ u32 r8_rank(const u64 *src, int64_t i, u32 before, u32 n, u32 after)
{
	const u64 me = src[i];
	const u32 tg = (u32)me & YK_R8_TOGGLE, pos = (u32)me & 1023u;
	int64_t a = i, e = i + 1;
	while (a > -(int64_t)before && ((u32)src[a - 1] & YK_R8_TOGGLE) == tg) --a;
	while (e < (int64_t)n + after && ((u32)src[e] & YK_R8_TOGGLE) == tg) ++e;
	u32 r = 0;
	for (int64_t j = a; j < e; ++j) r += ((u32)src[j] & 1023u) < pos;
	return (u32)((int64_t)before + a) + r;
}
/* one level-2 input record -> (hash, time): Rec {hash, position} + tbase, or a tagged record of sub-table c.bucket */
template <bool NEED_T>
__device__ __forceinline__ void p2_load(const Chunk2 &c, u32 i, const FastParams &fp, u64 *h, u32 *t)
{
	if (c.pad == 1) {
		const u64 *src = (const u64*)c.rec;
		*h = (src[i] >> YK_R8_TAG_BITS) << fp.pre | c.bucket;
		if (NEED_T) *t = c.tbase + r8_rank(src, (int64_t)i, c.before, c.n, c.after);
	} else {
		const Rec rc = c.rec[i];
		*h = rc.x;
		if (NEED_T) *t = (u32)rc.y + c.tbase;
	}
}
/* level-2 output / counting input: the sub-bucket's bits taken out of hash >> pre, the rank below it */
__device__ __forceinline__ u64 r8_pack(u64 h, u32 t, const FastParams &fp)
{
	const u64 x = h >> fp.pre;
	const int sh = fp.bloom_mode ? fp.nb - 9 - fp.s2_bits : 0;
	return (((x >> (sh + fp.s2_bits)) << sh) | (x & ((1ull << sh) - 1))) << fp.tb | t;
}
__device__ __forceinline__ Rec r8_unpack(u64 r, u32 sb, const FastParams &fp)
{
	const u64 xs = r >> fp.tb;
	const int sh = fp.bloom_mode ? fp.nb - 9 - fp.s2_bits : 0;
	const u64 x = ((xs >> sh) << (sh + fp.s2_bits)) | ((u64)(sb & ((1u << fp.s2_bits) - 1)) << sh) | (xs & ((1ull << sh) - 1));
	return make_ulonglong2(x << fp.pre | (sb >> fp.s2_bits), r & ((1ull << fp.tb) - 1));
}
__device__ __forceinline__ Rec lc_rec(const FastParams &fp, const Rec *rec, u64 i, u32 sb)
{
	return fp.rec8_out ? r8_unpack(((const u64*)rec)[i], sb, fp) : rec[i];
}
/* the same in two steps, for loads that are requested long before they are used (decoding at once would wait for the data) */
__device__ __forceinline__ Rec lc_raw(const FastParams &fp, const Rec *rec, u64 i)
{
	return fp.rec8_out ? make_ulonglong2(((const u64*)rec)[i], 0) : rec[i];
}
__device__ __forceinline__ Rec lc_dec(const FastParams &fp, const Rec raw, u32 sb)
{
	return fp.rec8_out ? r8_unpack(raw.x, sb, fp) : raw;
}

template <int MODE>   /* 0 = histogram, 1 = scatter */
__global__ __launch_bounds__(1024)
void k_part2(const Chunk2 *chunks, FastParams fp, u32 *rows2, Rec *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_bkt[];
	const Chunk2 c = chunks[blockIdx.x];
	const int S2 = 1 << fp.s2_bits, NT = blockDim.x;
	u32 *row = rows2 + (size_t)blockIdx.x * S2;
	for (int j = threadIdx.x; j < S2; j += NT) s_bkt[j] = MODE ? row[j] : 0;
	__syncthreads();
	for (u32 i = threadIdx.x; i < c.n; i += NT) {
		u64 h; u32 t = 0;
		p2_load<MODE != 0>(c, i, fp, &h, &t);
		const u32 d = atomicAdd(&s_bkt[sub_of(h, fp)], 1u);
		if (MODE) { if (fp.rec8_out) ((u64*)out)[d] = r8_pack(h, t, fp); else out[d] = make_ulonglong2(h, (u64)t); }
	}
	if (!MODE) { __syncthreads(); for (int j = threadIdx.x; j < S2; j += NT) row[j] = s_bkt[j]; }
}

/* the same scatter with write combining (see WcView above): 1024 records per round */
#define WC_CAP 5
#define WC_NT  1024
#define WC_SEG 2048          /* sub-buckets whose stacks fit in LDS at once */
__global__ __launch_bounds__(WC_NT)
void k_part2_wc(const Chunk2 *chunks, FastParams fp, const u32 *__restrict__ rows2, const u64 *__restrict__ sbstart, Rec *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
	const Chunk2 c = chunks[blockIdx.x];
	const int S2 = 1 << fp.s2_bits, SEG = S2 < WC_SEG ? S2 : WC_SEG;
	const u32 tid = threadIdx.x;
	WcView w;
	wc_carve<4, WC_CAP, true>(w, s_dyn, SEG, WC_NT);
	const u32 *row = rows2 + (size_t)blockIdx.x * S2;
	const u32 n_round = (c.n + WC_NT - 1) / WC_NT;
	/* more sub-buckets than stacks fit in LDS: one sweep over the chunk per segment of WC_SEG
	 * sub-buckets (the chunk is small enough to come back from L2) */
	for (int seg0 = 0; seg0 < S2; seg0 += SEG) {
		__syncthreads();
		for (int b = tid; b < SEG; b += WC_NT) {
			w.cnt[b] = 0; w.head[b] = row[seg0 + b];
			w.tail[b] = (c.spare & 1) ? (u32)sbstart[(size_t)c.bucket * S2 + seg0 + b + 1] : row[S2 + seg0 + b];   /* start of the next chunk's run = end of mine */
		}
		if (tid < 2) w.ntask[tid] = 0;
		__syncthreads();
		for (u32 rd = 0; rd < n_round; ++rd) {
			const u32 i = rd * WC_NT + tid, par = rd & 1;
			if (i < c.n) {
				u64 h; u32 t;
				p2_load<true>(c, i, fp, &h, &t);
				const u32 sub = sub_of(h, fp) - (u32)seg0;
				if (sub < (u32)SEG) wc_place<4, WC_CAP, true>(w, sub, h, t, par, out);
			}
			__syncthreads();
			wc_flush<4, WC_CAP, true>(w, par, out);
			__syncthreads();
		}
		wc_drain<4, WC_CAP, true>(w, SEG, out);
	}
}

/* the level-2 scatter for 8-byte output records: groups of 8 (64 bytes), one stack of WC8_CAP entries per sub-bucket (no tail
 * array: a record that finds its stack full is stored at its final place, wcs_place / wcs_flush) */
#define WC8_CAP 8
/* Input: tagged records only (fp.rec8_in).  The rank of a record (r8_rank) needs its round's neighbours; chasing them through
 * global memory is a chain of dependent loads per round, so the tags (11 bits) and toggles of three consecutive rounds of the
 * stream -- the previous, this and the next 1024 records -- live in an LDS ring: a run is at most 1024 records long, so the
 * window always holds it whole.  Records are requested two rounds ahead. */
__global__ __launch_bounds__(WC_NT)
void k_part2_wc8(const Chunk2 *chunks, FastParams fp, const u32 *__restrict__ rows2, u64 *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
	__shared__ unsigned short s_tag[3][WC_NT];
	__shared__ u32 s_tb[3][WC_NT / 32];
	const Chunk2 c = chunks[blockIdx.x];
	const int S2 = 1 << fp.s2_bits, SEG = S2 < WC_SEG ? S2 : WC_SEG;
	const u32 tid = threadIdx.x;
	WcView w;
	w.h = (u64*)s_dyn; w.t = 0;
	w.cnt = s_dyn + 2 * (size_t)SEG * WC8_CAP; w.head = w.cnt + SEG; w.tail = 0; w.task = w.head + SEG; w.ntask = w.task + WC_NT;
	const u32 *row = rows2 + (size_t)blockIdx.x * S2;
	const int n_round = (int)((c.n + WC_NT - 1) / WC_NT);
	const u64 *src = (const u64*)c.rec;
	const int64_t g_lo = -(int64_t)c.before, g_hi = (int64_t)c.n + c.after;      /* the sub-table's stream of this batch, relative to the chunk */
	auto fetch = [&](int q) -> u64 { const int64_t g = (int64_t)q * WC_NT + tid; return g >= g_lo && g < g_hi ? src[g] : 0; };
	auto post = [&](int q, u64 r) {                                                  /* round q's tags and toggles -> ring slot (q + 1) % 3 */
		const u32 sl = (u32)(q + 1) % 3u;
		s_tag[sl][tid] = (unsigned short)((u32)r & 1023u);
		const u64 bal = __ballot(((u32)r & YK_R8_TOGGLE) != 0);
		if ((tid & 63) == 0) { s_tb[sl][tid >> 5] = (u32)bal; s_tb[sl][(tid >> 5) + 1] = (u32)(bal >> 32); }
	};
	for (int seg0 = 0; seg0 < S2; seg0 += SEG) {
		__syncthreads();
		for (int b = tid; b < SEG; b += WC_NT) { w.cnt[b] = 0; w.head[b] = row[seg0 + b]; }
		if (tid < 2) w.ntask[tid] = 0;
		u64 cur, nxt;
		{ const u64 r = fetch(-1); cur = fetch(0); nxt = fetch(1); post(-1, r); post(0, cur); post(1, nxt); }
		__syncthreads();
		for (int rd = 0; rd < n_round; ++rd) {
			const u32 i = (u32)rd * WC_NT + tid, par = (u32)rd & 1;
			const u64 nn = fetch(rd + 2);
			if (i < c.n) {
				/* linear window: word lw covers records (rd - 1) * 1024 + 32 lw ..; block lw / 32 sits in ring slot (rd + lw / 32) % 3 */
				const u32 mine = ((u32)cur & YK_R8_TOGGLE) ? ~0u : 0u, pos = (u32)cur & 1023u;
				auto tw = [&](int lw) -> u32 { return s_tb[(u32)(rd + (lw >> 5)) % 3u][lw & 31] ^ mine; };   /* set bits: the other toggle */
				const int x = WC_NT + (int)tid;
				int a = 0, e = 3 * WC_NT;
				{
					int lw = x >> 5; u32 m = tw(lw) & ((1u << (x & 31)) - 1);
					while (!m && lw > 0) m = tw(--lw);
					if (m) a = (lw << 5) + 32 - __clz(m);
				}
				{
					int lw = x >> 5; u32 m = (x & 31) == 31 ? 0 : tw(lw) & (~0u << ((x & 31) + 1));
					while (!m && lw < 3 * WC_NT / 32 - 1) m = tw(++lw);
					if (m) e = (lw << 5) + __ffs(m) - 1;
				}
				const int64_t base = ((int64_t)rd - 1) * WC_NT;
				if (base + a < g_lo) a = (int)(g_lo - base);
				if (base + e > g_hi) e = (int)(g_hi - base);
				u32 r = 0;
				for (int j = a; j < e; ++j) r += (u32)s_tag[(u32)(rd + (j >> 10)) % 3u][j & (WC_NT - 1)] < pos;
				const u32 t = c.tbase + (u32)(base + a - g_lo) + r;
				const u64 h = (cur >> YK_R8_TAG_BITS) << fp.pre | c.bucket;
				const u32 sub = sub_of(h, fp) - (u32)seg0;
				if (sub < (u32)SEG) wcs_place<WC8_CAP>(w, sub, r8_pack(h, t, fp), par, out);
			}
			__syncthreads();
			wcs_flush<WC8_CAP>(w, par, out);
			post(rd + 2, nn);                                              /* over the slot of round rd - 1, which nobody reads any more */
			cur = nxt; nxt = nn;
			__syncthreads();
		}
		for (u32 b = tid / 8; b < (u32)SEG; b += WC_NT / 8) {             /* what is left in the stacks */
			const u32 q = tid & 7, cn = w.cnt[b];
			if (q < cn) out[w.head[b] + q] = w.h[b * WC8_CAP + q];
		}
	}
}

/* one workgroup per level-1 bucket: rows2 counts -> absolute offsets; sbstart[bucket * S2 + s] */
__global__ __launch_bounds__(256)
void k_part2_scan(const u32 *chunk_first, const u64 *bbase, int s2_bits, u32 *rows2, u64 *sbstart, int P)
{
	__shared__ u64 s_tot[256];
	__shared__ u64 s_carry;
	const int S2 = 1 << s2_bits, b = blockIdx.x;
	const u32 c0 = chunk_first[b], c1 = chunk_first[b + 1];
	if (threadIdx.x == 0) s_carry = bbase[b];
	__syncthreads();
	for (int s0 = 0; s0 < S2; s0 += 256) {
		const int s = s0 + threadIdx.x;
		u64 tot = 0;
		if (s < S2) for (u32 c = c0; c < c1; ++c) tot += rows2[(size_t)c * S2 + s];
		s_tot[threadIdx.x] = tot;
		__syncthreads();
		if (threadIdx.x == 0) {
			u64 acc = s_carry;
			for (int j = 0; j < 256; ++j) { const u64 t = s_tot[j]; s_tot[j] = acc; acc += t; }
			s_carry = acc;
		}
		__syncthreads();
		if (s < S2) {
			u64 run = s_tot[threadIdx.x];
			sbstart[(size_t)b * S2 + s] = run;
			for (u32 c = c0; c < c1; ++c) { const u32 v = rows2[(size_t)c * S2 + s]; rows2[(size_t)c * S2 + s] = (u32)run; run += v; }
		}
		__syncthreads();
	}
	if (b == P - 1 && threadIdx.x == 0) sbstart[(size_t)P * S2] = s_carry;
}

#define LC_EXIST 0x40000000u
#define LC_FP    0x80000000u
#define LC_CMASK 0x000fffffu              /* occurrences, clamped; bits 29:20 = rank inside a bloom block */
#define T32_INF  0xffffffffu
#define LC_BLOOM_WORDS 2048               /* bloom range staged in LDS: up to 128 blocks of 512 bits */
#define LC_GMAX 16                        /* keys per bloom block handled by the all-pairs gate */

/* three tiers of the same algorithm: LC_G = LDS tables, all-pairs bloom gate (35 KB of LDS, 4 WG/CU);
 * LC_S = LDS tables + sort arrays for sub-buckets with crowded bloom blocks or an un-staged range;
 * LC_X = tables in global scratch for sub-buckets whose distinct k-mers overflow the LDS table */
enum { LC_G = 0, LC_S = 1, LC_X = 2 };

struct LcTab { u64 *K; u32 *T1, *T2, *CN, *TM; u64 *SO; u32 *SP; u32 *BL; u32 *GC; unsigned short *G; u32 cap; };

template <int MODE> __device__ __forceinline__ void lc_sync() { if (MODE == LC_X) block_sync_global(); else __syncthreads(); }

__device__ __forceinline__ u32 lc_home(u64 key, int pre, u32 cap) { return (u32)(((key >> pre) * 0x9E3779B97F4A7C15ull) >> 24) & (cap - 1); }

struct BfSeq { u32 h1, h2, nd; };
__device__ __forceinline__ BfSeq lc_seq(u64 key, const FastParams &fp)       /* probe sequence inside the 512-bit block (bbf.c:28-33) */
{
	BfSeq q;
	const u64 x = key >> fp.pre;
	q.h1 = (u32)(x >> (fp.nb - 9)) & 511;
	q.h2 = fp.nb < 64 ? (u32)(x >> fp.nb) & 511 : 0;
	if ((q.h2 & 31) == 0) q.h2 = (q.h2 + 1) & 511;
	const u32 cyc = 512u >> (__ffs((int)q.h2) - 1);           /* the probes repeat after 512 / gcd(h2, 512) steps */
	q.nd = (u32)fp.n_hash < cyc ? (u32)fp.n_hash : cyc;
	return q;
}

/* Returns false when this tier cannot handle the sub-bucket (nothing observable has been modified). */
template <int MODE>
__device__ bool lc_body(const FastParams &fp, const LcTab &T, u32 sb, const u64 *__restrict__ sbstart,
                        const Rec *__restrict__ rec, u32 *bloom32, const ImgView &img, const LcOut &O,
                        u32 *s_misc /* [8] in LDS */)
{
	constexpr bool GLB = MODE == LC_X;
	const int tid = threadIdx.x;
	const u32 p = sb >> fp.s2_bits;
	const u64 lo = sbstart[sb], hi = sbstart[sb + 1];
	const int bb = fp.nb - 9, lb = bb - fp.s2_bits;              /* log2 bloom blocks owned by this sub-bucket */
	const bool stage_bloom = !GLB && fp.bloom_mode && lb <= 7;
	u32 *gw = 0;                                                   /* first word of the owned bloom range */
	if (fp.bloom_mode) gw = bloom32 + ((((u64)p << fp.nb) | ((u64)(sb & ((1u << fp.s2_bits) - 1)) << (lb + 9))) >> 5);
	if (lo == hi) {
		if (stage_bloom && fp.bf_virgin) for (u32 i = tid; i < (16u << lb); i += 256) gw[i] = 0;   /* nothing maps here */
		if (tid == 0) { O.nsel[sb] = 0; O.lp[sb] = 0; O.nd[sb] = 0; }
		return true;
	}
	u32 *s_ndist = s_misc, *s_ovf = s_misc + 1, *s_lp = s_misc + 2, *s_ne = s_misc + 3, *s_run = s_misc + 7;
	const bool pre0 = lo + tid < hi, pre1 = lo + 256 + tid < hi, pre2 = lo + 512 + tid < hi;
	Rec r0 = make_ulonglong2(0, 0), r1 = r0, r2 = r0;
	if (pre0) r0 = lc_raw(fp, rec, lo + tid);
	if (pre1) r1 = lc_raw(fp, rec, lo + 256 + tid);
	if (pre2) r2 = lc_raw(fp, rec, lo + 512 + tid);
	for (u32 i = tid; i < T.cap; i += 256) { T.K[i] = YK_EMPTY; T.T1[i] = T32_INF; T.T2[i] = T32_INF; T.CN[i] = 0; T.TM[i] = 0; }
	if (stage_bloom) { for (u32 i = tid; i < (16u << lb); i += 256) T.BL[i] = fp.bf_virgin ? 0u : gw[i]; T.GC[tid] = 0; }
	if (tid < 8) s_misc[tid] = 0;
	lc_sync<MODE>();

	/* A: count; first / second / last occurrence times (same loser rule as k_acc_insert).  The first
	 * three records of every lane were requested before the table was initialised. */
	u32 tmax = 0;
	auto put = [&](const Rec rc) {
		const u64 key = rc.x;
		const u32 t = (u32)rc.y;
		u32 s = lc_home(key, fp.pre, T.cap), n = 0;
		for (; n < T.cap; ++n, s = (s + 1) & (T.cap - 1)) {
			u64 cur = T.K[s];
			if (cur == key) break;
			if (cur == YK_EMPTY) {
				cur = atomicCAS(&T.K[s], YK_EMPTY, key);
				if (cur == YK_EMPTY) { atomicAdd(s_ndist, 1u); break; }
				if (cur == key) break;
			}
		}
		if (n == T.cap) { *s_ovf = 1; return; }
		if ((T.CN[s] & LC_CMASK) < 0x80000u) atomicAdd(&T.CN[s], 1u);   /* only min(count, 1023) is ever used */
		const u32 old = atomicMin(&T.T1[s], t);
		if (fp.bloom_mode) {
			if (old != T32_INF) atomicMin(&T.T2[s], old > t ? old : t);
			atomicMax(&T.TM[s], t);
		}
		tmax = t + 1 > tmax ? t + 1 : tmax;
	};
	if (pre0) put(lc_dec(fp, r0, sb));
	if (pre1) put(lc_dec(fp, r1, sb));
	if (pre2) put(lc_dec(fp, r2, sb));
	for (u64 i = lo + 768 + tid; i < hi; i += 256) put(lc_rec(fp, rec, i, sb));
	if (!fp.bloom_mode && tmax) atomicMax(s_lp, tmax);      /* without a filter every instance is a put-call */
	lc_sync<MODE>();
	bool give_up = !GLB && (*s_ovf || *s_ndist > T.cap / 4 * 3);
	if (fp.dbg & 16) return true;

	/* B: keys already in the table image only gain counts (htab.c:66-69 on an existing key); done
	 * after the last give-up point, see below */
	bool grouped = false;
	if (!give_up && fp.bloom_mode && !(fp.dbg & 32)) {
		if (stage_bloom) {
			/* group the new keys by 512-bit block with LDS counters (rank kept in CN[29:20]) */
			const u64 lmask = (1ull << lb) - 1;
			u32 *s_gc = T.GC;
			for (u32 s = tid; s < T.cap; s += 256) {
				if (T.K[s] == YK_EMPTY) continue;
				const u32 r = atomicAdd(&s_gc[(T.K[s] >> fp.pre) & lmask], 1u);
				T.CN[s] |= (r & 1023u) << 20;
				atomicMax(s_ne, r + 1);
			}
			lc_sync<MODE>();
			grouped = *s_ne <= LC_GMAX;
		}
		if (!grouped && MODE == LC_G) give_up = true;       /* crowded blocks / un-staged range: needs the sort arrays */
	}
	if (give_up) {
		if (stage_bloom && fp.bf_virgin) for (u32 i = tid; i < (16u << lb); i += 256) gw[i] = 0;   /* the next tier expects real zeros */
		return false;
	}

	if (fp.img_nonempty) {
		for (u32 s = tid; s < T.cap; s += 256) {
			if (T.K[s] == YK_EMPTY) continue;
			const int64_t idx = img_find(img, T.K[s]);
			if (idx >= 0) {
				const u64 kc = img.keys[idx], c = (kc & 1023) + (T.CN[s] & LC_CMASK);
				img.keys[idx] = fp.or_mode == 1 ? kc | (T.T1[s] & 15u) : fp.or_mode == 2 ? kc : (kc & ~1023ull) | (c > 1023 ? 1023 : c);
				T.CN[s] |= LC_EXIST;
			}
		}
		lc_sync<MODE>();
	}

	/* C: the bloom gate (bbf.c:25-42 + htab.c:63-65).  This workgroup is the only one whose k-mers map
	 * to these 512-bit blocks.  A key passes at its first occurrence iff each of its probe bits was
	 * set before -- by the pre-existing filter or by a key of the same block seen earlier. */
	if (fp.bloom_mode && !(fp.dbg & 32)) {
		const u64 lmask = (1ull << lb) - 1;
		if (grouped) {
			/* all pairs inside a block, every key in parallel; then the bits are ORed in (order-free) */
			u32 *s_gc = T.GC, *s_go = T.GC + 128;
			if (tid < 64) {                                           /* exclusive scan of the 128 counters */
				const u32 a0 = s_gc[2 * tid], b0 = s_gc[2 * tid + 1];
				u32 v = a0 + b0;
				for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(v, o); if (tid >= o) v += t; }
				s_go[2 * tid] = v - a0 - b0; s_go[2 * tid + 1] = v - b0;
			}
			lc_sync<MODE>();
			for (u32 s = tid; s < T.cap; s += 256)
				if (T.K[s] != YK_EMPTY) T.G[s_go[(T.K[s] >> fp.pre) & lmask] + (T.CN[s] >> 20 & 1023u)] = (unsigned short)s;
			lc_sync<MODE>();
			/* one lane per 512-bit block applies the block's few new keys in increasing first-occurrence
			 * time (repeated selection of the smallest time above the last one applied) */
			for (u32 b = tid; b < (1u << lb); b += 256) {
				const u32 g = s_gc[b], g0 = s_go[b];
				u32 *w = T.BL + (b << 4);
				u32 last = 0; bool first = true;
				for (u32 it = 0; it < g; ++it) {
					u32 best = T32_INF, bs = 0;
					for (u32 e = 0; e < g; ++e) {
						const u32 sy = T.G[g0 + e], t1 = T.T1[sy];
						if ((first || t1 > last) && t1 < best && !(T.CN[sy] & LC_EXIST)) { best = t1; bs = sy; }
					}
					if (best == T32_INF) break;
					first = false; last = best;
					const BfSeq q = lc_seq(T.K[bs], fp);
					u32 hits = 0;
					for (u32 i = 0, z = q.h1; i < q.nd; ++i, z = (z + q.h2) & 511) {
						const u32 word = w[z >> 5], bit = 1u << (z & 31);
						if (word & bit) ++hits; else w[z >> 5] = word | bit;
					}
					if (hits == q.nd) T.CN[bs] |= LC_FP;                    /* yak_bf_insert() == n_hash */
				}
			}
			lc_sync<MODE>();
		} else if (MODE != LC_G) {
			/* sort the new keys by (block, first occurrence); one lane per block applies them in order */
			if (tid == 0) *s_ne = 0;
			lc_sync<MODE>();
			for (u32 s = tid; s < T.cap; s += 256) {
				if (T.K[s] == YK_EMPTY || (T.CN[s] & LC_EXIST)) continue;
				const u64 blk_local = (T.K[s] >> fp.pre) & lmask;
				const u32 j = atomicAdd(s_ne, 1u);
				if (GLB) { T.SO[j] = blk_local << 32 | T.T1[s]; T.SP[j] = s; }
				else T.SO[j] = blk_local << 43 | (u64)T.T1[s] << 11 | s;
			}
			lc_sync<MODE>();
			const u32 ne = *s_ne;
			u32 m = 1; while (m < ne) m <<= 1;
			for (u32 i = ne + tid; i < m; i += 256) { T.SO[i] = ~0ull; if (GLB) T.SP[i] = 0; }
			lc_sync<MODE>();
			for (u32 k2 = 2; k2 <= m; k2 <<= 1)
				for (u32 j = k2 >> 1; j > 0; j >>= 1) {
					for (u32 i = tid; i < m; i += 256) {
						const u32 l = i ^ j;
						if (l > i) {
							const u64 a0 = T.SO[i], b0 = T.SO[l];
							if ((a0 > b0) == ((i & k2) == 0)) {
								T.SO[i] = b0; T.SO[l] = a0;
								if (GLB) { const u32 t = T.SP[i]; T.SP[i] = T.SP[l]; T.SP[l] = t; }
							}
						}
					}
					lc_sync<MODE>();
				}
			const int bsh = GLB ? 32 : 43;
			u32 *bw = stage_bloom ? T.BL : gw;
			for (u32 j0 = tid; j0 < ne; j0 += 256) {
				const u64 blk_local = T.SO[j0] >> bsh;
				if (j0 && (T.SO[j0 - 1] >> bsh) == blk_local) continue;      /* not the first key of its block */
				u32 *w = bw + (blk_local << 4);
				for (u32 j = j0; j < ne && (T.SO[j] >> bsh) == blk_local; ++j) {
					const u32 s = GLB ? T.SP[j] : (u32)T.SO[j] & 2047u;
					const BfSeq q = lc_seq(T.K[s], fp);
					u32 hits = 0;
					for (u32 i = 0, z = q.h1; i < q.nd; ++i, z = (z + q.h2) & 511) {
						const u32 word = w[z >> 5], bit = 1u << (z & 31);
						if (word & bit) ++hits; else w[z >> 5] = word | bit;
					}
					if (hits == q.nd) T.CN[s] |= LC_FP;
				}
			}
			lc_sync<MODE>();
		}
		if (stage_bloom && !(fp.dbg & 64)) for (u32 i = tid; i < (16u << lb); i += 256) gw[i] = T.BL[i];
		/* D: last put-call = last instance that is not a rejected first occurrence (htab.c:63-65) */
		u32 best = 0;
		for (u32 s = tid; s < T.cap; s += 256) {
			if (T.K[s] == YK_EMPTY) continue;
			const u32 cn = T.CN[s];
			if ((cn & (LC_EXIST | LC_FP)) || (cn & LC_CMASK) >= 2) best = T.TM[s] + 1 > best ? T.TM[s] + 1 : best;
		}
		if (best) atomicMax(s_lp, best);
		lc_sync<MODE>();
	}

	/* E: keys entering the table -> (key<<10|count, insertion time), appended to the sub-table's list
	 * with one reservation per workgroup */
	auto selected = [&](u32 s, u64 *kc, u32 *Tt) {
		if (T.K[s] == YK_EMPTY || (T.CN[s] & LC_EXIST)) return false;
		u32 c = T.CN[s] & LC_CMASK;
		if (!fp.bloom_mode || (T.CN[s] & LC_FP)) *Tt = T.T1[s];
		else if (T.T2[s] != T32_INF) { *Tt = T.T2[s]; c -= 1; }
		else return false;
		if (c > 1023) c = 1023;
		if (fp.or_mode) c = T.T1[s] & (fp.or_mode == 1 ? 15u : 1023u);   /* a key seen once per load: its flag / saved count travels in the time's low bits */
		*kc = (T.K[s] >> fp.pre) << 10 | c;
		return true;
	};
	/* the selected keys go to the front of the sub-bucket's own record range in the output arrays (a
	 * sub-bucket never selects more keys than it has records); k_lc_compact gathers the fragments */
	for (u32 s = tid; s < T.cap; s += 256) {
		u64 kc; u32 Tt;
		if (selected(s, &kc, &Tt)) { const u32 r = atomicAdd(s_run, 1u); O.kc[lo + r] = kc; O.T[lo + r] = fp.t_pass0 + Tt; }
	}
	lc_sync<MODE>();
	if (tid == 0) { O.nsel[sb] = *s_run; O.lp[sb] = *s_lp; O.nd[sb] = *s_ndist; }
	return true;
}

template <int MODE>
__global__ __launch_bounds__(256)
void k_lds_count(FastParams fp, const u64 *sbstart, const Rec *rec, u32 *bloom32, ImgView img, LcOut O, u64 *counters,
                 const u32 *in_list, u32 *ovf_list, int ovf_counter)
{
	__shared__ u64 s_K[YK_LDS_C];
	__shared__ u64 s_SO[MODE == LC_S ? YK_LDS_C : 1];
	__shared__ u32 s_T1[YK_LDS_C], s_T2[YK_LDS_C], s_CN[YK_LDS_C], s_TM[YK_LDS_C];
	__shared__ u32 s_BL[LC_BLOOM_WORDS];
	__shared__ u32 s_GC[256];
	__shared__ unsigned short s_G[YK_LDS_C];
	__shared__ u32 s_misc[8];
	LcTab T; T.K = s_K; T.T1 = s_T1; T.T2 = s_T2; T.CN = s_CN; T.TM = s_TM; T.SO = s_SO; T.SP = 0; T.BL = s_BL; T.GC = s_GC; T.G = s_G; T.cap = YK_LDS_C;
	const u32 sb = in_list ? in_list[blockIdx.x] : ((u32)fp.plo << fp.s2_bits) + blockIdx.x;   /* only the sub-tables of this shard */
	if (!lc_body<MODE>(fp, T, sb, sbstart, rec, bloom32, img, O, s_misc))
		if (threadIdx.x == 0) { ovf_list[atomicAdd(&counters[ovf_counter], 1ull)] = sb; O.nsel[sb] = 0; O.lp[sb] = 0; O.nd[sb] = 0; }
}

__global__ __launch_bounds__(256)
void k_lds_count_ovf(FastParams fp, const u64 *sbstart, const Rec *rec, u32 *bloom32, ImgView img, LcOut O,
                     const u32 *ovf_list, const u64 *scr_off, u64 *scr)
{
	__shared__ u32 s_misc[8];
	const u32 sb = ovf_list[blockIdx.x];
	const u64 n = sbstart[sb + 1] - sbstart[sb];
	u32 cap = 4096; while (cap < 2 * n) cap <<= 1;
	u64 *base = scr + scr_off[blockIdx.x];                      /* 40 B per slot = 5 u64 */
	LcTab T; T.cap = cap; T.BL = 0; T.GC = 0; T.G = 0;
	T.K = base; T.SO = base + cap;
	T.T1 = (u32*)(base + 2 * (u64)cap); T.T2 = T.T1 + cap; T.CN = T.T2 + cap; T.SP = T.CN + cap; T.TM = T.SP + cap;
	lc_body<LC_X>(fp, T, sb, sbstart, rec, bloom32, img, O, s_misc);
}

/* ------------------------------------------------------------------------------------------
 * k_lc2: the first tier of the exclusive-ownership counting, built for latency: a sub-bucket's ~560
 * records keep a workgroup busy for a few thousand cycles only, nearly all of them waiting (global
 * loads, dependent LDS atomics, barriers).  So
 *   * workgroups are PERSISTENT: each walks every gridDim-th sub-bucket and requests the next one's
 *     records while it counts the current one (the global latency disappears behind the LDS work);
 *   * the LDS table is organised by bloom block: the home slot of a key is (block inside the staged
 *     range) x (slots per block) + a few hash bits, so the keys of one 512-bit block sit in one probe
 *     cluster and "the other keys of my block" is a short scan from the block's first slot -- no
 *     grouping passes, no per-block lists, no limit on the keys per block;
 *   * the gate of bbf.c:25-42 / htab.c:63-65 is evaluated per key, all keys in parallel: a key passes at
 *     its first occurrence iff each of its probe bits is set in the filter as it was before this
 *     sub-bucket or belongs to a key of the same block whose first occurrence is earlier (every
 *     instance sets its bits whatever the gate says); then all bits are ORed in (order-free);
 *   * nothing is reserved with global atomics: the selected keys go to the front of the sub-bucket's
 *     own record range in the output arrays (LcOut), k_lc_compact gathers them per sub-table.
 * Three barriers per sub-bucket.  Sub-buckets this tier cannot take (too many distinct k-mers for the
 * LDS table) are listed for k_lds_count_ovf untouched.  Needs n_hash <= 32 and, with a filter, a
 * staged range (<= 128 blocks per sub-bucket); the host falls back to k_lds_count otherwise.
 * ------------------------------------------------------------------------------------------ */
#define LC2_CAP 1024
#define LC2_FULL 768                      /* distinct k-mers a sub-bucket may hold without a filter (the slot list borrows the bloom stage) ... */
#define LC2_FULL_BF 624                   /* ... and with one: 32000 B of LDS per workgroup = 25 allocation granules, 5 workgroups per CU */
#define LC2_FP    0x8000u                 /* 16-bit per-key word: occurrences (12 bits, clamped) | flags */
#define LC2_EXIST 0x4000u
#define LC2_CMASK 0x0fffu

__device__ u64 d_lc2_prof[8];      /* YAKAMD_DBG & 128: shader clocks per phase, summed over the workgroups (lane 0 of each) */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5)))
void k_lc2(FastParams fp, const u64 *__restrict__ sbstart, const Rec *__restrict__ rec, u32 *bloom32, ImgView img, LcOut O,
           u64 *counters, u32 *ovf_list, u32 n_sb)
{
	__shared__ u64 s_K[LC2_CAP];
	__shared__ u32 s_T1[LC2_CAP], s_T2[LC2_CAP], s_TM[LC2_CAP], s_CN[LC2_CAP / 2];     /* 5 workgroups per CU */
	__shared__ u32 s_BL[LC_BLOOM_WORDS];
	__shared__ unsigned short s_listb[LC2_FULL_BF];                /* slots of the distinct k-mers, in claim order */
	__shared__ u32 s_misc[8];
	const u32 tid = threadIdx.x;
	const bool bloom = fp.bloom_mode != 0, virgin = fp.bf_virgin != 0;
	const int lb = bloom ? fp.nb - 9 - fp.s2_bits : 0;             /* log2 bloom blocks owned by a sub-bucket */
	const int rsh = 10 - lb;                                       /* log2 table slots per block */
	const u32 R = 1u << rsh, nbw = bloom ? 16u << lb : 0;
	const u64 lmask = (1ull << lb) - 1;
	const u32 sb0 = (u32)fp.plo << fp.s2_bits;
	u32 *s_ndist = s_misc, *s_ovf = s_misc + 1, *s_lp = s_misc + 2, *s_run = s_misc + 3;
	unsigned short *s_list = bloom ? s_listb : (unsigned short*)s_BL;
	const u32 full = bloom ? LC2_FULL_BF : LC2_FULL;

	auto cn_get = [&](u32 s) -> u32 { return s_CN[s >> 1] >> (16 * (s & 1)) & 0xffffu; };
	for (u32 i = tid; i < LC2_CAP; i += 256) { s_K[i] = YK_EMPTY; s_T1[i] = T32_INF; s_T2[i] = T32_INF; s_TM[i] = 0; }
	for (u32 i = tid; i < LC2_CAP / 2; i += 256) s_CN[i] = 0;
	if (virgin) for (u32 i = tid; i < nbw; i += 256) s_BL[i] = 0;
	if (tid < 8) s_misc[tid] = 0;
	u32 it = blockIdx.x;
	u64 lo = 0, hi = 0;
	Rec r0 = make_ulonglong2(0, 0), r1 = r0, r2 = r0;
	if (it < n_sb) {
		lo = sbstart[sb0 + it]; hi = sbstart[sb0 + it + 1];
		if (lo + tid < hi) r0 = lc_raw(fp, rec, lo + tid);
		if (lo + 256 + tid < hi) r1 = lc_raw(fp, rec, lo + 256 + tid);
		if (lo + 512 + tid < hi) r2 = lc_raw(fp, rec, lo + 512 + tid);
	}
	__syncthreads();

	auto home = [&](u64 key) -> u32 {
		const u64 x = key >> fp.pre;
		return ((u32)(x & lmask) << rsh) | ((u32)((x * 0x9E3779B97F4A7C15ull) >> 24) & (R - 1));   /* middle bits: the top ones name the sub-bucket when there is no filter (sub_of) */
	};
	u32 tmax = 0;
	auto put = [&](const Rec rc) {
		const u64 key = rc.x;
		const u32 t = (u32)rc.y;
		u32 s = home(key), n = 0;
		for (; n < LC2_CAP; ++n, s = (s + 1) & (LC2_CAP - 1)) {
			u64 cur = s_K[s];
			if (cur == key) break;
			if (cur == YK_EMPTY) {
				cur = atomicCAS(&s_K[s], YK_EMPTY, key);
				if (cur == YK_EMPTY) { const u32 at = atomicAdd(s_ndist, 1u); if (at < full) s_list[at] = (unsigned short)s; break; }
				if (cur == key) break;
			}
		}
		if (n == LC2_CAP) { *s_ovf = 1; return; }
		if ((cn_get(s) & LC2_CMASK) < 0x800u) atomicAdd(&s_CN[s >> 1], 1u << (16 * (s & 1)));   /* only min(count, 1024) is ever used; 256 racing lanes cannot carry out of the field */
		const u32 old = atomicMin(&s_T1[s], t);
		if (bloom) {
			if (old != T32_INF) atomicMin(&s_T2[s], old > t ? old : t);
			atomicMax(&s_TM[s], t);
		}
		tmax = t + 1 > tmax ? t + 1 : tmax;
	};

	u64 pf[7] = { 0, 0, 0, 0, 0, 0, 0 };
	while (it < n_sb) {
		const u32 sb = sb0 + it, itn = it + gridDim.x;
		u32 *gw = bloom ? bloom32 + ((((u64)(sb >> fp.s2_bits) << fp.nb) | ((u64)(sb & ((1u << fp.s2_bits) - 1)) << (lb + 9))) >> 5) : 0;
		u64 lon = 0, hin = 0;
		if (itn < n_sb) { lon = sbstart[sb0 + itn]; hin = sbstart[sb0 + itn + 1]; }
		if (bloom && !virgin) for (u32 i = tid; i < nbw; i += 256) s_BL[i] = gw[i];
		const bool prof = (fp.dbg & 128) && tid == 0;
		u64 tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0;
		if (prof) tp0 = __builtin_readcyclecounter();
		/* A: count; first / second / last occurrence (same loser rule as k_acc_insert) */
		tmax = 0;
		u64 tpw = 0;
		if (fp.dbg & 128) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (prof) tpw = __builtin_readcyclecounter(); }
		{
			/* the lane's (up to) three records go through the table side by side: every step is three independent LDS
			 * operations and one wait, instead of three dependent chains one after the other (put, below, is the same
			 * thing for one record; it takes the records beyond 768 and the probes that do not end at the home slot) */
			const Rec rc[3] = { lc_dec(fp, r0, sb), lc_dec(fp, r1, sb), lc_dec(fp, r2, sb) };
			bool v[3] = { lo + tid < hi, lo + 256 + tid < hi, lo + 512 + tid < hi }, res[3], won[3], tried[3];
			u32 sl[3], at[3], cn[3], old[3];
			u64 cur[3];
#pragma unroll
			for (int i = 0; i < 3; ++i) { sl[i] = home(rc[i].x); cur[i] = v[i] ? s_K[sl[i]] : 0; }
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				res[i] = v[i] && cur[i] == rc[i].x; won[i] = false; tried[i] = v[i] && cur[i] == YK_EMPTY;
				if (tried[i]) cur[i] = atomicCAS(&s_K[sl[i]], YK_EMPTY, rc[i].x);
			}
#pragma unroll
			for (int i = 0; i < 3; ++i) if (tried[i]) { won[i] = cur[i] == YK_EMPTY; res[i] = won[i] || cur[i] == rc[i].x; }
#pragma unroll
			for (int i = 0; i < 3; ++i) if (won[i]) at[i] = atomicAdd(s_ndist, 1u);
#pragma unroll
			for (int i = 0; i < 3; ++i) if (won[i] && at[i] < full) s_list[at[i]] = (unsigned short)sl[i];
#pragma unroll
			for (int i = 0; i < 3; ++i) if (v[i] && !res[i]) {                 /* the home slot belongs to another key: walk on */
				const u64 key = rc[i].x;
				u32 q = (sl[i] + 1) & (LC2_CAP - 1), n = 1;
				for (; n < LC2_CAP; ++n, q = (q + 1) & (LC2_CAP - 1)) {
					u64 c2 = s_K[q];
					if (c2 == key) break;
					if (c2 == YK_EMPTY) {
						c2 = atomicCAS(&s_K[q], YK_EMPTY, key);
						if (c2 == YK_EMPTY) { const u32 a2 = atomicAdd(s_ndist, 1u); if (a2 < full) s_list[a2] = (unsigned short)q; break; }
						if (c2 == key) break;
					}
				}
				if (n == LC2_CAP) { *s_ovf = 1; v[i] = false; }
				sl[i] = q;
			}
#pragma unroll
			for (int i = 0; i < 3; ++i) cn[i] = v[i] ? cn_get(sl[i]) : 0;
#pragma unroll
			for (int i = 0; i < 3; ++i) if (v[i]) {
				if ((cn[i] & LC2_CMASK) < 0x800u) atomicAdd(&s_CN[sl[i] >> 1], 1u << (16 * (sl[i] & 1)));   /* only min(count, 1024) is ever used; 256 racing lanes cannot carry out of the field */
				old[i] = atomicMin(&s_T1[sl[i]], (u32)rc[i].y);
			}
#pragma unroll
			for (int i = 0; i < 3; ++i) if (v[i]) {
				const u32 t = (u32)rc[i].y;
				if (bloom) {
					if (old[i] != T32_INF) atomicMin(&s_T2[sl[i]], old[i] > t ? old[i] : t);
					atomicMax(&s_TM[sl[i]], t);
				}
				tmax = t + 1 > tmax ? t + 1 : tmax;
			}
		}
		for (u64 i = lo + 768 + tid; i < hi; i += 256) put(lc_rec(fp, rec, i, sb));
		if (!bloom && tmax) atomicMax(s_lp, tmax);                /* without a filter every instance is a put-call */
		u64 tpp = 0;
		if (prof) tpp = __builtin_readcyclecounter();
		/* the next sub-bucket's records travel while this one is gated and selected */
		if (lon + tid < hin) r0 = lc_raw(fp, rec, lon + tid);
		if (lon + 256 + tid < hin) r1 = lc_raw(fp, rec, lon + 256 + tid);
		if (lon + 512 + tid < hin) r2 = lc_raw(fp, rec, lon + 512 + tid);
		__syncthreads();
		if (prof) tp1 = __builtin_readcyclecounter();
		const u32 ndist = *s_ndist;
		const bool give_up = *s_ovf || ndist > full;
		if (!give_up && !(fp.dbg & 16)) {
			/* B: keys already in the table image only gain counts (htab.c:66-69 on an existing key) */
			if (fp.img_nonempty) {
				for (u32 li = tid; li < ndist; li += 256) {
					const u32 s = s_list[li];
					const int64_t idx = img_find(img, s_K[s]);
					if (idx >= 0) {
						const u64 kc = img.keys[idx], c = (kc & 1023) + (cn_get(s) & LC2_CMASK);
						img.keys[idx] = fp.or_mode == 1 ? kc | (s_T1[s] & 15u) : fp.or_mode == 2 ? kc : (kc & ~1023ull) | (c > 1023 ? 1023 : c);
						atomicOr(&s_CN[s >> 1], LC2_EXIST << (16 * (s & 1)));
					}
				}
				__syncthreads();
			}
			/* C: the gate, one lane per key.  First the earlier keys of the same block are collected (a short
			 * scan of the block's probe cluster), then their probe bits are matched against the key's own:
			 * for n_hash <= 4 the key's probes are packed into four 16-bit fields and one peer probe is
			 * compared with all of them at once */
			if (bloom && !(fp.dbg & 32)) {
				for (u32 li = tid; li < ndist; li += 256) {
					const u32 s = s_list[li];
					const u64 kx = s_K[s];
					if (cn_get(s) & LC2_EXIST) continue;
					const BfSeq q = lc_seq(kx, fp);
					const u32 blk = (u32)((kx >> fp.pre) & lmask), t1x = s_T1[s];
					u32 miss = q.nd >= 32 ? 0xffffffffu : (1u << q.nd) - 1;
					if (!virgin) {
						const u32 *w = s_BL + (blk << 4);
						for (u32 i = 0, z = q.h1; i < q.nd; ++i, z = (z + q.h2) & 511) if (w[z >> 5] >> (z & 31) & 1) miss &= ~(1u << i);
					}
					u64 peers = 0; u32 np = 0;                                    /* up to 6 slot numbers of 10 bits */
					const u32 r0s = blk << rsh;
					auto consider = [&](const u32 j, const u64 ky) {
						if (j == s || (u32)((ky >> fp.pre) & lmask) != blk || s_T1[j] >= t1x || (cn_get(j) & LC2_EXIST)) return;
						if (np < 6) peers |= (u64)j << (10 * np);
						else {                                                      /* a crowded block: the plain way */
							const BfSeq y = lc_seq(ky, fp);
							for (u32 a2 = 0, w = y.h1; a2 < y.nd; ++a2, w = (w + y.h2) & 511)
								for (u32 i = 0, z = q.h1; i < q.nd; ++i, z = (z + q.h2) & 511) if (z == w) miss &= ~(1u << i);
						}
						++np;
					};
					u32 d = 0;
					if (R == 8 && miss) {                                         /* the block's eight home slots with four 16-byte reads, no dependent chain */
						const ulonglong2 *kb = (const ulonglong2*)(s_K + r0s);
						const ulonglong2 q0 = kb[0], q1 = kb[1], q2 = kb[2], q3 = kb[3];
						const u64 kk[8] = { q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y };
#pragma unroll
						for (u32 i = 0; i < 8; ++i) if (kk[i] != YK_EMPTY) consider(r0s + i, kk[i]);
						d = kk[7] == YK_EMPTY ? LC2_CAP : 8;                        /* slot 7 unused: no key of this block can sit beyond */
					}
					for (; miss && d < LC2_CAP; ++d) {
						const u32 j = (r0s + d) & (LC2_CAP - 1);
						const u64 ky = s_K[j];
						if (ky == YK_EMPTY) { if (d + 1 >= R) break; continue; }     /* no key of this block can sit beyond */
						consider(j, ky);
					}
					if (np > 6) np = 6;
					if (miss && np) {
						if (q.nd <= 4) {
							u64 zp = 0;
							for (u32 i = 0, z = q.h1; i < 4; ++i, z = (z + q.h2) & 511) zp |= (u64)(i < q.nd ? z : 0x3ffu) << (16 * i);
							u64 nz = 0x8000800080008000ull;                        /* field i keeps its top bit while probe i is matched by nobody */
							for (u32 e = 0; e < np; ++e) {
								const BfSeq y = lc_seq(s_K[(u32)(peers >> (10 * e)) & 1023u], fp);
								for (u32 a2 = 0, w = y.h1; a2 < y.nd; ++a2, w = (w + y.h2) & 511)
									nz &= (zp ^ (w * 0x0001000100010001ull)) + 0x7fff7fff7fff7fffull;   /* fields are < 2^10: no carry between them */
							}
							const u32 cov = (u32)(~nz >> 15 & 1) | (u32)(~nz >> 30 & 2) | (u32)(~nz >> 45 & 4) | (u32)(~nz >> 60 & 8);
							miss &= ~cov;
						} else {
							for (u32 e = 0; e < np; ++e) {
								const BfSeq y = lc_seq(s_K[(u32)(peers >> (10 * e)) & 1023u], fp);
								for (u32 a2 = 0, w = y.h1; a2 < y.nd; ++a2, w = (w + y.h2) & 511)
									for (u32 i = 0, z = q.h1; i < q.nd; ++i, z = (z + q.h2) & 511) if (z == w) miss &= ~(1u << i);
							}
						}
					}
					if (miss == 0) atomicOr(&s_CN[s >> 1], LC2_FP << (16 * (s & 1)));   /* yak_bf_insert() == n_hash */
				}
				if (!virgin) __syncthreads();                                  /* every gate has read the filter as it was */
			}
			if (prof) tp2 = __builtin_readcyclecounter();
			/* C' + D + E: set the bits, last put-call, keys entering the table */
			u32 best = 0;
			for (u32 li = tid; li < ndist; li += 256) {
				const u32 s = s_list[li];
				const u64 kx = s_K[s];
				const u32 cn = cn_get(s);
				if (bloom && !(fp.dbg & 32)) {
					if (!(cn & LC2_EXIST)) {
						const BfSeq q = lc_seq(kx, fp);
						u32 *w = s_BL + ((u32)((kx >> fp.pre) & lmask) << 4);
						for (u32 i = 0, z = q.h1; i < q.nd; ++i, z = (z + q.h2) & 511) atomicOr(&w[z >> 5], 1u << (z & 31));
					}
					/* last put-call = last instance that is not a rejected first occurrence (htab.c:63-65) */
					if ((cn & (LC2_EXIST | LC2_FP)) || (cn & LC2_CMASK) >= 2) best = s_TM[s] + 1 > best ? s_TM[s] + 1 : best;
				}
				if (cn & LC2_EXIST) continue;
				u32 c = cn & LC2_CMASK, Tt;
				if (!bloom || (cn & LC2_FP)) Tt = s_T1[s];
				else if (s_T2[s] != T32_INF) { Tt = s_T2[s]; c -= 1; }
				else continue;
				if (c > 1023) c = 1023;
				if (fp.or_mode) c = s_T1[s] & (fp.or_mode == 1 ? 15u : 1023u);       /* a key seen once per load: its flag / saved count travels in the time's low bits */
				const u32 r = atomicAdd(s_run, 1u);
				O.kc[lo + r] = (kx >> fp.pre) << 10 | c; O.T[lo + r] = fp.t_pass0 + Tt;
			}
			if (best) atomicMax(s_lp, best);
		}
		__syncthreads();
		if (prof) tp3 = __builtin_readcyclecounter();
		/* write-back, per-sub-bucket results, clean tables for the next sub-bucket */
		if (give_up) {
			if (virgin) for (u32 i = tid; i < nbw; i += 256) gw[i] = 0;       /* the next tier expects real zeros */
			if (tid == 0) { ovf_list[atomicAdd(&counters[YKC_NOVF2], 1ull)] = sb; O.nsel[sb] = 0; O.lp[sb] = 0; O.nd[sb] = 0; }
		} else {
			if (bloom && !(fp.dbg & 64)) for (u32 i = tid; i < nbw; i += 256) gw[i] = s_BL[i];
			if (tid == 0) { O.nsel[sb] = *s_run; O.lp[sb] = *s_lp; O.nd[sb] = ndist; }
		}
		if (tid == 0) { s_misc[0] = 0; s_misc[1] = 0; s_misc[2] = 0; s_misc[3] = 0; }
		if (give_up) {
			for (u32 i = tid; i < LC2_CAP; i += 256) { s_K[i] = YK_EMPTY; s_T1[i] = T32_INF; s_T2[i] = T32_INF; s_TM[i] = 0; }
			for (u32 i = tid; i < LC2_CAP / 2; i += 256) s_CN[i] = 0;
		} else for (u32 li = tid; li < ndist; li += 256) {
			const u32 s = s_list[li];
			s_K[s] = YK_EMPTY; s_T1[s] = T32_INF; s_T2[s] = T32_INF; s_TM[s] = 0; s_CN[s >> 1] = 0;
		}
		if (virgin) for (u32 i = tid; i < nbw; i += 256) s_BL[i] = 0;
		it = itn; lo = lon; hi = hin;
		__syncthreads();
		if (prof) {
			const u64 tp4 = __builtin_readcyclecounter();
			pf[0] += tp1 - tp0; pf[1] += tp2 > tp1 ? tp2 - tp1 : 0; pf[2] += tp3 - (tp2 > tp1 ? tp2 : tp1);
			pf[3] += tp4 - tp3; pf[4] += 1; pf[5] += tpw - tp0; pf[6] += tpp - tpw;
		}
	}
	if ((fp.dbg & 128) && tid == 0) for (int i = 0; i < 7; ++i) atomicAdd(&d_lc2_prof[i], pf[i]);
}

/* ------------------------------------------------------------------------------------------
 * k_cnt2: the count pass (create_new = 0, reference htab.c:71-75) over the SAME input as the pass before, on the records that pass left
 * grouped by sub-bucket (the level-2 partition) and on the list of keys every sub-bucket put into the table.  All instances of a k-mer sit
 * in one sub-bucket, so a workgroup owns the counters of its sub-bucket's keys outright: the keys go into a small LDS set, the records are
 * streamed once (one LDS lookup, one LDS increment per hit), and every key then adds its count to its slot of the table image (one table
 * probe per KEY instead of one per instance; exclusive owner: a plain read-modify-write).  Persistent workgroups with the next sub-bucket's
 * first records in flight, as in k_lc2.  A sub-bucket with more keys than the LDS set takes (other tiers of the insert kernel) looks every
 * instance up in the image with device atomics.  k_img_fold saturates the counts afterwards.
 * ------------------------------------------------------------------------------------------ */
#define C2_CAP 2048
#define C2_FULL 1280
__global__ __launch_bounds__(256)
void k_cnt2(FastParams fp, const u64 *__restrict__ sbstart, const Rec *__restrict__ rec, const u64 *__restrict__ key_off, const u64 *__restrict__ key_kc,
            ImgView img, u32 n_sb, u32 *__restrict__ key_cnt)
{
	__shared__ u64 s_K[C2_CAP];
	__shared__ u32 s_C[C2_CAP];
	__shared__ unsigned short s_slot[C2_FULL];
	const u32 tid = threadIdx.x;
	const u32 sb0 = (u32)fp.plo << fp.s2_bits;
	for (u32 i = tid; i < C2_CAP; i += 256) { s_K[i] = YK_EMPTY; s_C[i] = 0; }
	auto home = [&](u64 x) -> u32 { return (u32)((x * 0x9E3779B97F4A7C15ull) >> 40) & (C2_CAP - 1); };
	u32 it = blockIdx.x;
	u64 lo = 0, hi = 0;
	Rec r0 = make_ulonglong2(0, 0), r1 = r0, r2 = r0;
	if (it < n_sb) {
		lo = sbstart[sb0 + it]; hi = sbstart[sb0 + it + 1];
		if (lo + tid < hi) r0 = lc_raw(fp, rec, lo + tid);
		if (lo + 256 + tid < hi) r1 = lc_raw(fp, rec, lo + 256 + tid);
		if (lo + 512 + tid < hi) r2 = lc_raw(fp, rec, lo + 512 + tid);
	}
	__syncthreads();
	while (it < n_sb) {
		const u32 sb = sb0 + it, itn = it + gridDim.x, p = sb >> fp.s2_bits;
		const u64 ko = key_off[sb];
		const u32 nk = (u32)(key_off[sb + 1] - ko);
		u64 lon = 0, hin = 0;
		if (itn < n_sb) { lon = sbstart[sb0 + itn]; hin = sbstart[sb0 + itn + 1]; }
		const bool in_lds = nk <= C2_FULL;
		if (in_lds) {
			for (u32 i = tid; i < nk; i += 256) {                            /* the keys this sub-bucket put into the table */
				const u64 x = key_kc[ko + i] >> 10;
				u32 s = home(x);
				for (;;) {
					const u64 cur = atomicCAS(&s_K[s], YK_EMPTY, x);
					if (cur == YK_EMPTY || cur == x) break;
					s = (s + 1) & (C2_CAP - 1);
				}
				s_slot[i] = (unsigned short)s;
			}
			__syncthreads();
		}
		auto take = [&](const Rec rc) {
			if (in_lds) {
				const u64 x = rc.x >> fp.pre;
				u32 s = home(x);
				for (;;) {
					const u64 cur = s_K[s];
					if (cur == x) { atomicAdd(&s_C[s], 1u); break; }
					if (cur == YK_EMPTY) break;
					s = (s + 1) & (C2_CAP - 1);
				}
			} else {
				const int64_t idx = img_find(img, rc.x);
				if (idx >= 0) atomicAdd(&img.delta[idx], 1u);
			}
		};
		if (lo + tid < hi) take(lc_dec(fp, r0, sb));
		if (lo + 256 + tid < hi) take(lc_dec(fp, r1, sb));
		if (lo + 512 + tid < hi) take(lc_dec(fp, r2, sb));
		for (u64 i = lo + 768 + tid; i < hi; i += 256) take(lc_rec(fp, rec, i, sb));
		/* the next sub-bucket's records travel while this one's counts go to the table */
		if (lon + tid < hin) r0 = lc_raw(fp, rec, lon + tid);
		if (lon + 256 + tid < hin) r1 = lc_raw(fp, rec, lon + 256 + tid);
		if (lon + 512 + tid < hin) r2 = lc_raw(fp, rec, lon + 512 + tid);
		__syncthreads();
		if (in_lds) {
			/* the counts go out next to the keys (coalesced); k_cnt2_apply adds them to the table image, one lane per key: the table probe is a chain
			 * of dependent global reads that a couple of dozen lanes of this workgroup would wait for at every sub-bucket */
			for (u32 i = tid; i < nk; i += 256) { const u32 s = s_slot[i]; key_cnt[ko + i] = s_C[s]; s_K[s] = YK_EMPTY; s_C[s] = 0; }
		} else for (u32 i = tid; i < nk; i += 256) key_cnt[ko + i] = 0;    /* counted straight into the image */
		(void)p;
		it = itn; lo = lon; hi = hin;
		__syncthreads();
	}
}

/* counts of k_cnt2 -> table image: key i of sub-table p (the list is grouped by sub-table: seg_base) */
__global__ __launch_bounds__(256)
void k_cnt2_apply(const u64 *__restrict__ key_kc, const u32 *__restrict__ key_cnt, const u64 *__restrict__ seg_base, int plo, int pre, ImgView img)
{
	const u32 p = (u32)plo + blockIdx.y;
	const u64 a = seg_base[p], b = seg_base[p + 1];
	for (u64 i = a + (u64)blockIdx.x * 256 + threadIdx.x; i < b; i += (u64)gridDim.x * 256) {
		const u32 c = key_cnt[i];
		if (!c) continue;
		const int64_t idx = img_find(img, (key_kc[i] >> 10) << pre | p);
		if (idx >= 0) img.delta[idx] += c;                                /* a key occurs once in the list: nobody else touches its slot */
	}
}

/* first key of every sub-bucket in the gathered (unsorted) key list: seg_base[p] + exclusive scan of nsel inside sub-table p; key_off[n_sb] = all keys */
__global__ __launch_bounds__(256)
void k_nsel_scan(const u32 *__restrict__ nsel, int s2_bits, int plo, int phi, int P, const u64 *__restrict__ seg_base, u64 *__restrict__ key_off)
{
	__shared__ u32 s_w[5];
	const u32 p = blockIdx.x, S2 = 1u << s2_bits, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const size_t b0 = (size_t)p * S2;
	const bool mine = (int)p >= plo && (int)p < phi;
	u64 run = seg_base[p];
	for (u32 j0 = 0; j0 < S2; j0 += 256) {
		__syncthreads();
		const u32 j = j0 + tid, n = (mine && j < S2) ? nsel[b0 + j] : 0;
		u32 incl = n;
		for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(incl, o); if (lane >= (u32)o) incl += x; }
		if (lane == 63) s_w[wave] = incl;
		__syncthreads();
		u32 e = incl - n, tot = 0;
		for (u32 w = 0; w < 4; ++w) { if (w < wave) e += s_w[w]; tot += s_w[w]; }
		if (j < S2) key_off[b0 + j] = run + e;
		run += tot;
	}
	if ((int)p == P - 1 && tid == 0) key_off[(size_t)P * S2] = run;
}

/* keys selected per sub-table = sum over its sub-buckets */
__global__ __launch_bounds__(256)
void k_lc_sum(const u32 *__restrict__ nsel, int s2_bits, int plo, u32 *seg_cnt)
{
	__shared__ u32 s_tot;
	const u32 p = (u32)plo + blockIdx.x, S2 = 1u << s2_bits;
	if (threadIdx.x == 0) s_tot = 0;
	__syncthreads();
	u32 c = 0;
	for (u32 j = threadIdx.x; j < S2; j += 256) c += nsel[(size_t)p * S2 + j];
	for (int o = 32; o; o >>= 1) c += __shfl_down(c, o);
	if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_tot, c);
	__syncthreads();
	if (threadIdx.x == 0) seg_cnt[p] = s_tot;
}

/* gather the fragments of one sub-table into its contiguous list (any order: the sort by insertion time
 * follows); last put-call and distinct k-mers of the sub-table */
__global__ __launch_bounds__(256)
void k_lc_compact(LcOut O, const u64 *__restrict__ sbstart, int s2_bits, int plo, u64 t_pass0, const u64 *__restrict__ seg_base,
                  u64 *__restrict__ out_kc, u64 *__restrict__ out_T, u64 *lastput, u32 *ndist_p)
{
	__shared__ u32 s_off[256], s_red[2];
	const u32 p = (u32)plo + blockIdx.x, S2 = 1u << s2_bits, tid = threadIdx.x, lane = tid & 63;
	const size_t b0 = (size_t)p * S2;
	if (tid < 2) s_red[tid] = 0;
	u64 run = seg_base[p];
	u32 lpm = 0, nds = 0;
	for (u32 j0 = 0; j0 < S2; j0 += 256) {
		__syncthreads();
		const u32 j = j0 + tid, n = j < S2 ? O.nsel[b0 + j] : 0;
		if (j < S2) { const u32 l = O.lp[b0 + j]; lpm = l > lpm ? l : lpm; nds += O.nd[b0 + j]; }
		s_off[tid] = n;
		__syncthreads();
		if (tid < 64) {                                            /* exclusive scan of 256 counts by one wave */
			u32 v[4], t = 0;
			for (int q = 0; q < 4; ++q) { v[q] = s_off[4 * tid + q]; t += v[q]; }
			u32 incl = t;
			for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(incl, o); if (lane >= (u32)o) incl += x; }
			u32 e = incl - t;
			for (int q = 0; q < 4; ++q) { s_off[4 * tid + q] = e; e += v[q]; }
			if (tid == 63) s_red[1] = incl;
		}
		__syncthreads();
		/* one wave per sub-bucket */
		for (u32 q = tid >> 6; q < 256 && j0 + q < S2; q += 4) {
			const u32 cnt = O.nsel[b0 + j0 + q];
			const u64 src = sbstart[b0 + j0 + q], dst = run + s_off[q];
			for (u32 i = lane; i < cnt; i += 64) { out_kc[dst + i] = O.kc[src + i]; out_T[dst + i] = O.T[src + i]; }
		}
		run += s_red[1];
	}
	for (int o = 32; o; o >>= 1) { const u32 x = __shfl_down(lpm, o); lpm = x > lpm ? x : lpm; nds += __shfl_down(nds, o); }
	__syncthreads();
	if (lane == 0) { atomicMax(&s_red[0], lpm); atomicAdd(&ndist_p[p], nds); }
	__syncthreads();
	if (tid == 0 && s_red[0]) { const u64 v = t_pass0 + (u64)s_red[0]; if (v > lastput[p]) lastput[p] = v; }
}

/* ------------------------------------------------------------------------------------------
 * launch wrappers
 * ------------------------------------------------------------------------------------------ */
static inline int grid_for(u64 n, int per_block = 256, int cap = 256 * 8)
{
	u64 g = (n + per_block - 1) / per_block;
	if (g < 1) g = 1;
	if (g > (u64)cap) g = cap;            /* 256 CUs x 8 workgroups, grid-stride beyond that */
	return (int)g;
}

/* ------------------------------------------------------------------------------------------
 * khashl resize to ANY capacity (khashl.h:152-195; yak_ch_tighten htab.c:102-110 shrinks, the
 * pre-resize of yak_ch_merge htab.c:262-266 grows by more than one doubling).  These are one-off
 * maintenance calls, so each sub-table is simply walked by one lane with the literal rule
 * (replay_double is that rule for any old / new size); the workgroup copies and normalises around it.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256)
void k_resize(const ResizeTask *tasks, const u64 *__restrict__ old_keys, const u32 *__restrict__ old_used,
              u64 *new_keys, u32 *new_used, u32 *scr_used)
{
	__shared__ u32 s_progress;
	const ResizeTask T = tasks[blockIdx.x];
	if (T.old_bits == YK_NOCAP) return;
	const u32 n = 1u << T.old_bits, tid = threadIdx.x;
	u64 *keys = new_keys + T.new_off;
	u32 *nu = new_used + (T.new_off >> 5), *cur = scr_used + (T.new_off >> 5);
	for (u32 i = tid; i < n; i += 256) keys[i] = old_keys[T.old_off + i];
	if (T.new_bits == T.old_bits && !T.rehash) {             /* untouched: plain copy */
		for (u32 w = tid; w < (n + 31) / 32; w += 256) nu[w] = old_used[(T.old_off >> 5) + w];
		return;
	}
	const u32 N = 1u << T.new_bits;
	for (u32 w = tid; w < (n + 31) / 32; w += 256) cur[w] = old_used[(T.old_off >> 5) + w];
	for (u32 w = tid; w < (N + 31) / 32; w += 256) nu[w] = 0;
	__syncthreads();
	if (tid == 0) replay_double(keys, cur, nu, n, N, T.new_bits, &s_progress);
	__syncthreads();
	const u32 span = n > N ? n : N;
	for (u32 i = tid; i < span; i += 256) if (i >= N || !bm_get(nu, i)) keys[i] = YK_EMPTY;
}

/* table keys of one sub-table (slot order) -> full hashes + list positions, for a merge pass */
__global__ __launch_bounds__(256)
void k_keys_to_hashes(const u64 *__restrict__ kc, const u64 *__restrict__ seg_off, int pre, u64 *__restrict__ hash, u32 *__restrict__ t)
{
	const u64 a = seg_off[blockIdx.x], b = seg_off[blockIdx.x + 1];
	for (u64 i = a + threadIdx.x; i < b; i += 256) { hash[i] = (kc[i] >> 10) << pre | blockIdx.x; t[i] = (u32)i; }
}

/* ==========================================================================================
 * Count-existing passes on sub-tables too large for k_img_count_lds (its bitmap + rank table + one
 * counter per key must fit one workgroup's LDS: ~70 K keys).  The hashes of a sub-table are split
 * once more by the HOME-SLOT RANGE of their key -- khashl.h:98 takes the home slot from the top bits
 * of one 32-bit product, so the top bits of that product name a contiguous range of 2^RNG_LOG slots --
 * and ONE workgroup then owns a range: `used` bits and one 16-bit counter per slot in LDS, the key
 * compare against HBM/L2, plain read-modify-writes at the end.  A probe that runs past the end of
 * the range (linear probing across the boundary) is handed over: appended to `list` and counted by
 * k_img_count_h afterwards; if the list overflows, a second sweep (CROSS = 1) redoes exactly those
 * instances with global reads and atomics.
 * ========================================================================================== */
#define RNG_LOG 16
/* range id, out of 2^rb per sub-table; a sub-table with fewer ranges uses every 2^(rb - rbp)-th id */
__device__ __forceinline__ u32 rng_of(u64 h, const ImgView &img, int rb, int rng_log)
{
	const u32 p = (u32)h & ((1u << img.pre) - 1), bits = img.bits[p];
	if (bits == YK_NOCAP || (int)bits <= rng_log || rb == 0) return 0;
	const int rbp = (int)bits - rng_log;
	return ((u32)(h >> img.pre) * 2654435769u) >> (32 - rbp) << (rb - rbp);
}

template <int MODE>   /* 0 = histogram, 1 = write-combining scatter (groups of 8 hashes) */
__global__ __launch_bounds__(WC_NT)
void k_hpart2(const Chunk2 *chunks, ImgView img, int rb, int rng_log, u32 *rows2, const u64 *__restrict__ sbstart, u64 *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
	const Chunk2 c = chunks[blockIdx.x];
	const u64 *src = (const u64*)c.rec;
	const int S2 = 1 << rb;
	const u32 tid = threadIdx.x;
	if (MODE == 0) {
		for (int j = tid; j < S2; j += WC_NT) s_dyn[j] = 0;
		__syncthreads();
		for (u32 i = tid; i < c.n; i += WC_NT) atomicAdd(&s_dyn[rng_of(src[i], img, rb, rng_log)], 1u);
		__syncthreads();
		for (int j = tid; j < S2; j += WC_NT) rows2[(size_t)blockIdx.x * S2 + j] = s_dyn[j];
		return;
	}
	WcView w;
	wc_carve<8, XW_CAP_H, false>(w, s_dyn, S2, WC_NT);
	const u32 *row = rows2 + (size_t)blockIdx.x * S2;
	for (int b = tid; b < S2; b += WC_NT) {
		w.cnt[b] = 0; w.head[b] = row[b];
		w.tail[b] = (c.spare & 1) ? (u32)sbstart[(size_t)c.bucket * S2 + b + 1] : row[S2 + b];
	}
	if (tid < 2) w.ntask[tid] = 0;
	__syncthreads();
	const u32 n_round = (c.n + WC_NT - 1) / WC_NT;
	for (u32 rd = 0; rd < n_round; ++rd) {
		const u32 i = rd * WC_NT + tid, par = rd & 1;
		if (i < c.n) { const u64 h = src[i]; wc_place<8, XW_CAP_H, false>(w, rng_of(h, img, rb, rng_log), h, 0, par, out); }
		__syncthreads();
		wc_flush<8, XW_CAP_H, false>(w, par, out);
		__syncthreads();
	}
	wc_drain<8, XW_CAP_H, false>(w, S2, out);
}

template <int CROSS>
__global__ __launch_bounds__(1024)
void k_img_count_rng(const u64 *__restrict__ rec, const u64 *__restrict__ sbstart, ImgView img, int plo, int rb, int rng_log,
                     u64 *__restrict__ list, u32 *list_n, u32 list_cap)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
	const u32 bkt = ((u32)plo << rb) + blockIdx.x, p = bkt >> rb, r = bkt & ((1u << rb) - 1), tid = threadIdx.x;
	const u32 bits = img.bits[p];
	const u64 lo = sbstart[bkt], hi = sbstart[bkt + 1];
	if (bits == YK_NOCAP || lo == hi) return;
	const int rbp = (int)bits > rng_log ? (int)bits - rng_log : 0;
	const u32 cap = 1u << bits, nmask = cap - 1, len = cap >> rbp, start = (r >> (rb - rbp)) * len, nw = (len + 31) / 32;
	const u64 off = img.off[p];
	u32 *s_bm = s_dyn, *s_ct = s_dyn + nw;                             /* used bits | one 16-bit counter per slot of the range */
	for (u32 w = tid; w < nw; w += 1024) s_bm[w] = img.used[((off + start) >> 5) + w];
	for (u32 i = tid; i < (len + 1) / 2; i += 1024) s_ct[i] = 0;
	__syncthreads();
	for (u64 i = lo + tid; i < hi; i += 1024) {
		const u64 h = rec[i], kid = h >> img.pre;
		const u32 first = yk_h2b((u32)kid, bits) - start;              /* the partition put the home slot in this range */
		u32 li = first;
		for (;;) {
			if (!(s_bm[li >> 5] >> (li & 31) & 1)) break;                 /* khashl get: stop at the first unused slot */
			if (img.keys[off + start + li] >> 10 == kid) {
				if (!CROSS) {
					const u32 sh = 16 * (li & 1);
					if ((s_ct[li >> 1] >> sh & 0xffffu) < 4096u) atomicAdd(&s_ct[li >> 1], 1u << sh);   /* only min(count, 1023) matters */
				}
				break;
			}
			++li;
			if (rbp == 0) { li &= nmask; if (li == first) break; continue; }   /* the range is the whole table: plain wrap-around */
			if (li < len) continue;
			if (!CROSS) {                                                 /* the probe leaves the range */
				const u32 at = atomicAdd(&list_n[0], 1u);
				if (at < list_cap) list[at] = h; else atomicAdd(&list_n[1], 1u);
			} else {
				const u32 home = (start + first) & nmask;
				for (u32 s2 = (start + len) & nmask; s2 != home; s2 = (s2 + 1) & nmask) {
					const u64 g = off + s2;
					if (!(img.used[g >> 5] >> (g & 31) & 1)) break;
					if (img.keys[g] >> 10 == kid) { atomicAdd(&img.delta[g], 1u); break; }
				}
			}
			break;
		}
	}
	if (CROSS) return;
	__syncthreads();
	for (u32 li = tid; li < len; li += 1024) {
		const u32 c = s_ct[li >> 1] >> (16 * (li & 1)) & 0xffffu;
		if (c) img.delta[off + start + li] += c;                          /* exclusive owner: plain read-modify-write */
	}
}

/* ==========================================================================================
 * Count-existing passes with the KEYS in LDS (the default for records grouped by sub-table).
 * k_img_count_lds keeps bitmap + counters of a whole sub-table in one workgroup's LDS but compares
 * every probe against a key in HBM/L2: 64 bytes fetched per 8-byte compare, the measured traffic was
 * 4x the algorithmic bytes.  Here a sub-table is cut into 2^rbp slot ranges and ONE workgroup owns a
 * range outright: its `used` bits, a per-word rank table, the keys of its used slots packed in rank
 * order and a 16-bit counter per key all sit in LDS, so a probe is LDS work only.  The records are
 * not partitioned again: the 2^rbp workgroups of a sub-table all stream the sub-table's records and
 * keep those whose home slot (top bits of khashl's 32-bit product, khashl.h:98) falls in their range.
 * blockIdx is mapped so that these workgroups run on ONE XCD next to each other in time: the stream
 * comes from HBM once and from that XCD's L2 for the others.  A probe that runs past the end of its
 * range goes to `list` (counted by k_img_count_h afterwards; CROSS = 1 is the second sweep if the list
 * overflowed).  A range with more keys than the LDS budget counts its records with device atomics.
 * ========================================================================================== */
#define OWN_U 8
template <int W, int CROSS>
__global__ __launch_bounds__(1024)
void k_img_count_own(const u64 *__restrict__ rec, const u64 *__restrict__ bstart, ImgView img, int plo, int n_p, int rb, int rng_log, u32 kmax,
                     u64 *__restrict__ list, u32 *list_n, u32 list_cap, int ytag)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
	__shared__ u32 s_wsum[16];
	const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const u32 xcd = blockIdx.x & 7, y = blockIdx.x >> 3, pi = ((y >> rb) << 3) + xcd, r = y & ((1u << rb) - 1);
	if (pi >= (u32)n_p) return;
	const u32 p = (u32)plo + pi, bits = img.bits[p];
	const u64 lo = bstart[p], hi = bstart[p + 1];
	if (bits == YK_NOCAP || lo == hi) return;
	const int rbp = (int)bits > rng_log ? (int)bits - rng_log : 0;     /* log2 ranges of this sub-table */
	if (r >> rbp) return;
	const u32 cap = 1u << bits, nmask = cap - 1, L = cap >> rbp, start = r * L, nw = (L + 31) / 32;
	const u64 off = img.off[p];
	u32 *s_br = s_dyn;                                            /* word w of the bitmap at [2w], the rank of its first slot at [2w + 1]: one 8-byte read per probe */
	u64 *s_k = (u64*)(s_dyn + 2 * ((nw + 1) & ~1u));
	u32 *s_ct = (u32*)(s_k + kmax);
	/* bitmap + exclusive popcount scan (every thread owns a contiguous run of words) */
	const u32 per = (nw + 1023) / 1024;
	u32 mine = 0;
	for (u32 j = 0; j < per; ++j) {
		const u32 w = tid * per + j;
		if (w < nw) { const u32 x = img.used[((off + start) >> 5) + w]; s_br[2 * w] = x; mine += __popc(x); }
	}
	u32 incl = mine;
	for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o); if (lane >= (u32)o) incl += t; }
	if (lane == 63) s_wsum[wave] = incl;
	__syncthreads();
	u32 base = incl - mine, total = 0;
	for (u32 w2 = 0; w2 < 16; ++w2) { if (w2 < wave) base += s_wsum[w2]; total += s_wsum[w2]; }
	const bool fits = total <= kmax;
	const u32 rsel = rbp ? 32 - rbp : 0;
	if (!fits) {                                                   /* cannot happen with the host's sizing rule short of a pathological table */
		if (CROSS) return;
		for (u64 i = lo + tid; i < hi; i += 1024) {
			const u64 h = ytag == 2 ? (rec[W * i] >> YK_R8_TAG_BITS) << img.pre | p : ytag ? (rec[W * i] >> img.pre) << img.pre | p : rec[W * i];
			if (rbp && ((u32)(h >> img.pre) * 2654435769u) >> rsel != r) continue;
			const int64_t hit = img_find(img, h);
			if (hit >= 0) atomicAdd(&img.delta[hit], 1u);
		}
		return;
	}
	{
		u32 rr = base;
		for (u32 j = 0; j < per; ++j) {
			const u32 w = tid * per + j;
			if (w >= nw) break;
			s_br[2 * w + 1] = rr;
			u32 x = s_br[2 * w];
			while (x) { const u32 b = __ffs((int)x) - 1; x &= x - 1; s_k[rr++] = img.keys[off + start + w * 32 + b]; }
		}
	}
	for (u32 i = tid; i < (total + 1) / 2; i += 1024) s_ct[i] = 0;
	__syncthreads();
	auto probe = [&](const u64 h) {
		const u64 kid = h >> img.pre;
		const u32 first = (((u32)kid * 2654435769u) >> (32 - bits)) - start;
		u32 s = first;
		u64 wr = *(const u64*)&s_br[2 * (s >> 5)];                   /* a cluster walk stays inside one bitmap word most of the time: one read, then keys only */
		for (;;) {
			const u32 word = (u32)wr;
			if (!(word >> (s & 31) & 1)) break;                          /* khashl get: stop at the first unused slot */
			const u32 rr = (u32)(wr >> 32) + __popc(word & ((1u << (s & 31)) - 1));
			if (s_k[rr] >> 10 == kid) {
				if (!CROSS) {
					const u32 sh = 16 * (rr & 1);
					if ((s_ct[rr >> 1] >> sh & 0xffffu) < 4096u) atomicAdd(&s_ct[rr >> 1], 1u << sh);   /* only min(count, 1023) matters */
				}
				break;
			}
			++s;
			if (rbp == 0) { s &= nmask; if (s == first) break; if ((s & 31) == 0) wr = *(const u64*)&s_br[2 * (s >> 5)]; continue; }   /* the range is the whole table: plain wrap-around */
			if (s < L) { if ((s & 31) == 0) wr = *(const u64*)&s_br[2 * (s >> 5)]; continue; }
			if (!CROSS) {                                                /* the probe leaves the range */
				const u32 at = atomicAdd(&list_n[0], 1u);
				if (at < list_cap) list[at] = ytag ? (h >> img.pre) << img.pre | p : h; else atomicAdd(&list_n[1], 1u);
			} else {
				const u32 home = (start + first) & nmask;
				for (u32 s2 = (start + L) & nmask; s2 != home; s2 = (s2 + 1) & nmask) {
					const u64 g = off + s2;
					if (!(img.used[g >> 5] >> (g & 31) & 1)) break;
					if (img.keys[g] >> 10 == kid) { atomicAdd(&img.delta[g], 1u); break; }
				}
			}
			break;
		}
	};
	/* OWN_U records per lane are requested together (the stream comes from L2 / HBM: one exposed latency per
	 * OWN_U records instead of one per record); the records of this range -- one in 2^rbp -- are packed into
	 * a per-wave LDS queue and probed 64 at a time, so the probe code runs with full waves */
	u64 *s_q = (u64*)(s_ct + ((kmax + 1) / 2 + 1 & ~1u)) + wave * 128;
	u32 qn = 0;
	const u64 STEP = (u64)1024 * OWN_U;
	const u32 ypm = (1u << img.pre) - 1, ysh = (u32)img.pre - (u32)rbp;          /* ytag records: the product's top `pre` bits sit in the low bits */
	auto fetch = [&](u64 (&hv)[OWN_U], const u64 i0) {
#pragma unroll
		for (int u = 0; u < OWN_U; ++u) { const u64 i = i0 + (u64)u * 1024 + tid; hv[u] = rec[W * (i < hi ? i : hi - 1)]; }   /* always a load: the wait counts stay static (consume checks the index) */
	};
	auto consume = [&](const u64 (&hv)[OWN_U], const u64 i0) {
#pragma unroll
		for (int u = 0; u < OWN_U; ++u) {
			const bool valid = i0 + (u64)u * 1024 + tid < hi;
			/* ytag == 2: tagged level-1 records of the pass before (yakamd_count_retained): hash >> pre above the 12 tag bits */
			const u64 hq = ytag == 2 ? (hv[u] >> YK_R8_TAG_BITS) << img.pre : hv[u];
			if (rbp == 0) { if (valid) probe(hq); continue; }
			const bool match = valid && (ytag == 1 ? ((u32)hv[u] & ypm) >> ysh == r : ((u32)(hq >> img.pre) * 2654435769u) >> rsel == r);
			const u64 mk = __ballot(match);
			if (match) s_q[qn + __popcll(mk & lanemask_lt())] = hq;
			qn += (u32)__popcll(mk);
			if (qn >= 64) { qn -= 64; probe(s_q[qn + lane]); }
		}
	};
	/* (a second register set with the next OWN_U records in flight while these are probed was measured: same time, and the eight
	 * range workgroups of a sub-table drift apart in the stream, so HBM fetches rise from 1.2x to 2.2x of the records) */
	u64 ha[OWN_U];
	for (u64 i0 = lo; i0 < hi; i0 += STEP) { fetch(ha, i0); consume(ha, i0); }
	if (lane < qn) probe(s_q[lane]);
	if (CROSS) return;
	__syncthreads();
	for (u32 j = 0; j < per; ++j) {
		const u32 w = tid * per + j;
		if (w >= nw) break;
		u32 x = s_br[2 * w], rr = s_br[2 * w + 1];
		while (x) {
			const u32 b = __ffs((int)x) - 1;
			x &= x - 1;
			const u32 c = s_ct[rr >> 1] >> (16 * (rr & 1)) & 0xffffu;
			if (c) img.delta[off + start + w * 32 + b] += c;             /* exclusive owner: plain read-modify-write */
			++rr;
		}
	}
}

extern "C" {

/* k-mers ENDING at positions [pos0, n) of `bases` (bytes before pos0 are read as left context);
 * emitted position = index in `bases` - t_sub */
void yk_launch_extract(const uint8_t *bases, int64_t pos0, int64_t n, int64_t t_sub, int k, int pre, int plo, int phi,
                       u64 *out_hash, u32 *out_t, u64 *cursor, hipStream_t st)
{
	if (n <= pos0) return;
	const u64 tiles = ((u64)(n - pos0) + XT_TILE - 1) / XT_TILE;
	hipLaunchKernelGGL(k_extract, dim3((unsigned)tiles), dim3(XT_THREADS), 0, st, bases, pos0, n, t_sub, k, pre, plo, phi, out_hash, out_t, cursor);
}

/* ASCII image -> the packed image of yakamd_feed_packed_dev: 32 bases per lane, two code words and one validity word */
__global__ __launch_bounds__(256)
void k_pack(const uint8_t *__restrict__ a, int64_t n, u32 *__restrict__ codes, u32 *__restrict__ valid)
{
	const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x, p0 = w * 32;
	if (p0 >= n) return;
	u32 c0 = 0, c1 = 0, v = 0;
	for (int j = 0; j < 32; ++j) {
		const int64_t p = p0 + j;
		const u32 c = p < n ? d_nt4[a[p]] : 4u;
		if (c < 4) { v |= 1u << j; if (j < 16) c0 |= c << (2 * j); else c1 |= c << (2 * (j - 16)); }
	}
	codes[2 * w] = c0; codes[2 * w + 1] = c1; valid[w] = v;
}
void yk_launch_pack(const uint8_t *a, int64_t n, u32 *codes, u32 *valid, hipStream_t st)
{
	if (n > 0) hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 32 * 256 - 1) / (32 * 256))), dim3(256), 0, st, a, n, codes, valid);
}

/* partitioning extraction: returns through bstart[1 << nb_bits] (device) the record count */
void yk_launch_xpart(const uint8_t *bases, int64_t pos0, int64_t n, int64_t t_sub, int k, int pre, int plo, int phi,
                     int nb_bits, u32 *rows, u64 *partial, u64 *bstart, Rec *out, int hash_only, hipStream_t st, const u32 *valid)
{
	if (n <= pos0) return;
	const int n_blk = (int)(((u64)(n - pos0) + (u64)XP_T * XT_TILE - 1) / ((u64)XP_T * XT_TILE));
	const size_t lds = sizeof(u32) << nb_bits;
	if (hash_only == 2) {                                             /* tagged 8-byte records: needs nb_bits <= 10, k < 32 (the caller checks) */
		static bool attr3 = false;
		if (!attr3) { hipFuncSetAttribute((const void*)k_xpart_wcs<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); attr3 = true; }
		static bool said = false;
		if (!said && getenv("YAKAMD_VERBOSE") && atoi(getenv("YAKAMD_VERBOSE")) > 1) {
			int nb = 0; said = true;
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_xpart_wcs<true>, 1024, wc_lds_bytes<8, XW_CAP_S, false>(1 << nb_bits, 1024));
			fprintf(stderr, "[yak_amd] k_xpart_wcs: %zu B of dynamic LDS, %d workgroups per CU\n", wc_lds_bytes<8, XW_CAP_S, false>(1 << nb_bits, 1024), nb);
		}
		hipLaunchKernelGGL(k_xpart<3>, dim3(n_blk), dim3(XT_THREADS), 3 * lds, st, bases, pos0, n, t_sub, k, pre, plo, phi, nb_bits, rows, out, valid);
		launch_part_scan(rows, n_blk, nb_bits, partial, bstart, st, true);
		hipLaunchKernelGGL(k_xpart_wcs<true>, dim3(n_blk), dim3(1024), (wc_lds_bytes<8, XW_CAP_S, false>(1 << nb_bits, 1024)), st,
		                   bases, pos0, n, k, pre, plo, phi, nb_bits, (const u32*)rows, (u64*)out, 0, valid);
		return;
	}
	hipLaunchKernelGGL(k_xpart<0>, dim3(n_blk), dim3(XT_THREADS), lds, st, bases, pos0, n, t_sub, k, pre, plo, phi, nb_bits, rows, out, valid);
	launch_part_scan(rows, n_blk, nb_bits, partial, bstart, st);
	static const int wc = getenv("YAKAMD_XP_WC") ? atoi(getenv("YAKAMD_XP_WC")) : 3;   /* bit 0: {hash, position} scatter, bit 1: hash-only scatter */
	static bool attr = false;
	if (!attr) {
		hipFuncSetAttribute((const void*)k_xpart_wc<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
		hipFuncSetAttribute((const void*)k_xpart_wc<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
		attr = true;
	}
	if (nb_bits <= 10 && ((wc >> (hash_only ? 1 : 0) & 1) || hash_only == 3)) {
		static const int wcs2 = getenv("YAKAMD_XP_WCS") ? atoi(getenv("YAKAMD_XP_WCS")) : 1;
		if (hash_only && (wcs2 || hash_only == 3) && k < 32) {   /* hash_only == 3: bare hashes with the range tag (the caller checked k and nb_bits) */                         /* the round-stable variant needs 7 slots per stack only: two workgroups per CU */
			static bool attr4 = false;
			if (!attr4) { hipFuncSetAttribute((const void*)k_xpart_wcs<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); attr4 = true; }
			hipLaunchKernelGGL(k_xpart_wcs<false>, dim3(n_blk), dim3(1024), (wc_lds_bytes<8, XW_CAP_S, false>(1 << nb_bits, 1024)), st,
			                   bases, pos0, n, k, pre, plo, phi, nb_bits, (const u32*)rows, (u64*)out, hash_only == 3, valid);
		}
		else if (hash_only) hipLaunchKernelGGL(k_xpart_wc<2>, dim3(n_blk), dim3(XW_NT), (wc_lds_bytes<8, XW_CAP_H, false>(1 << nb_bits, XW_NT)), st,
		                                  bases, pos0, n, t_sub, k, pre, plo, phi, nb_bits, (const u32*)rows, (const u64*)bstart, (void*)out, valid);
		else hipLaunchKernelGGL(k_xpart_wc<1>, dim3(n_blk), dim3(XW_NT), (wc_lds_bytes<4, XW_CAP_T, true>(1 << nb_bits, XW_NT)), st,
		                        bases, pos0, n, t_sub, k, pre, plo, phi, nb_bits, (const u32*)rows, (const u64*)bstart, (void*)out, valid);
	}
	else if (hash_only) hipLaunchKernelGGL(k_xpart<2>, dim3(n_blk), dim3(XT_THREADS), lds, st, bases, pos0, n, t_sub, k, pre, plo, phi, nb_bits, rows, out, valid);
	else hipLaunchKernelGGL(k_xpart<1>, dim3(n_blk), dim3(XT_THREADS), lds, st, bases, pos0, n, t_sub, k, pre, plo, phi, nb_bits, rows, out, valid);
}

int yk_part_groups(void) { return PS_G; }
int yk_xpart_blocks(int64_t n_pos) { return (int)(((u64)n_pos + (u64)XP_T * XT_TILE - 1) / ((u64)XP_T * XT_TILE)); }
int yk_rpart_blocks(int64_t n_rec) { return (int)(((u64)n_rec + RP_CHUNK - 1) / RP_CHUNK); }

void yk_launch_rpart(const u64 *in_hash, const u32 *in_t, int64_t n, int pre, int plo, int phi,
                     int nb_bits, u32 *rows, u64 *partial, u64 *bstart, Rec *out, hipStream_t st)
{
	if (n <= 0) return;
	const int n_blk = yk_rpart_blocks(n);
	const size_t lds = sizeof(u32) << nb_bits;
	hipLaunchKernelGGL(k_rpart<0>, dim3(n_blk), dim3(256), lds, st, in_hash, in_t, n, pre, plo, phi, nb_bits, rows, out);
	launch_part_scan(rows, n_blk, nb_bits, partial, bstart, st);
	hipLaunchKernelGGL(k_rpart<1>, dim3(n_blk), dim3(256), lds, st, in_hash, in_t, n, pre, plo, phi, nb_bits, rows, out);
}

void yk_replay_prof(u64 *out8) { (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(d_rp_prof), 64); u64 z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(d_rp_prof), z, 64); }
void yk_par_counters(u32 *ok, u32 *fail)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(ok, HIP_SYMBOL(d_par_ok), 4);
	(void)hipMemcpyFromSymbol(fail, HIP_SYMBOL(d_par_fail), 4);
}

int yk_bad_hash_seen(hipStream_t st)
{
	u32 v = 0;
	(void)hipStreamSynchronize(st);
	(void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(d_bad_hash), 4);
	return (int)v;
}

void yk_launch_acc_init(AccSlot *s, u64 n, hipStream_t st)
{
	hipLaunchKernelGGL(k_acc_init, dim3(grid_for(n)), dim3(256), 0, st, s, n);
}

void yk_launch_acc_insert(const Rec *rec, int64_t n, u64 t0, AccTab tab, ImgView img,
                          int img_nonempty, int bloom_mode, u64 *newlist, u64 *counters, hipStream_t st)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(k_acc_insert, dim3(grid_for((u64)n)), dim3(256), 0, st, rec, n, t0, tab, img, img_nonempty, bloom_mode, newlist, counters);
}

void yk_launch_acc_rehash(AccTab oldt, AccTab newt, hipStream_t st)
{
	hipLaunchKernelGGL(k_acc_rehash, dim3(grid_for(oldt.mask + 1)), dim3(256), 0, st, oldt, newt);
}

void yk_launch_img_count(const Rec *rec, int64_t n, ImgView img, hipStream_t st)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(k_img_count, dim3(grid_for((u64)n)), dim3(256), 0, st, rec, n, img);
}

/* LDS needed by k_img_count_lds for a sub-table of `cap` slots holding `count` keys */
size_t yk_img_count_lds_bytes(u32 cap, u32 count) { return (size_t)((cap + 31) / 32) * 8 + (size_t)(count + 1) / 2 * 4 + 16; }

int yk_launch_img_count_lds(const void *rec, int hash_only, const u64 *bstart, ImgView img, int plo, int phi, size_t lds, u64 *compact, u32 stride, hipStream_t st)
{
	static bool attr = false;
	if (!attr) {
		if (hipFuncSetAttribute((const void*)k_img_count_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess ||
		    hipFuncSetAttribute((const void*)k_img_count_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) { (void)hipGetLastError(); return -1; }
		attr = true;
	}
	if (hash_only) hipLaunchKernelGGL(k_img_count_lds<1>, dim3(phi - plo), dim3(1024), lds, st, (const u64*)rec, bstart, img, plo, compact, stride);
	else hipLaunchKernelGGL(k_img_count_lds<2>, dim3(phi - plo), dim3(1024), lds, st, (const u64*)rec, bstart, img, plo, compact, stride);
	return 0;
}

void yk_launch_img_count_h(const u64 *hash, int64_t n, ImgView img, hipStream_t st)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(k_img_count_h, dim3(grid_for((u64)n)), dim3(256), 0, st, hash, n, img);
}

void yk_launch_img_inc(ImgView img, u64 hash, u64 *out2, hipStream_t st) { hipLaunchKernelGGL(k_img_inc, dim3(1), dim3(1), 0, st, img, hash, out2); }

void yk_launch_img_fold(ImgView img, u64 n_slots, hipStream_t st)
{
	if (n_slots) hipLaunchKernelGGL(k_img_fold, dim3(grid_for(n_slots)), dim3(256), 0, st, img, n_slots);
}

void yk_launch_img_clear(ImgView img, u64 n_slots, hipStream_t st)
{
	if (n_slots) hipLaunchKernelGGL(k_img_clear, dim3(grid_for(n_slots)), dim3(256), 0, st, img, n_slots);
}

void yk_launch_img_hist(ImgView img, u64 n_slots, u64 *hist, hipStream_t st)
{
	if (n_slots) hipLaunchKernelGGL(k_img_hist, dim3(grid_for(n_slots)), dim3(256), 0, st, img, n_slots, (unsigned long long*)hist);
}

void yk_launch_img_setcnt(ImgView img, u64 n_slots, u32 cnt, hipStream_t st)
{
	if (n_slots) hipLaunchKernelGGL(k_img_setcnt, dim3(grid_for(n_slots)), dim3(256), 0, st, img, n_slots, cnt);
}

void yk_launch_lastput(const Rec *rec, int64_t n, u64 t0, u64 t_from, AccTab tab, ImgView img,
                       int img_nonempty, int bloom_mode, const u32 *only_missing, u64 *lp_batch, hipStream_t st)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(k_lastput, dim3(grid_for((u64)n)), dim3(256), 0, st, rec, n, t0, t_from, tab, img, img_nonempty, bloom_mode, only_missing, lp_batch);
}

void yk_launch_lastput_merge(u64 *lastput, const u64 *lp_batch, u32 *missing, u32 *n_missing, int P, int plo, int phi, hipStream_t st)
{
	hipLaunchKernelGGL(k_lastput_merge, dim3((P + 255) / 256), dim3(256), 0, st, lastput, lp_batch, missing, n_missing, P, plo, phi);
}

void yk_launch_bf_test(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, u64 *miss, hipStream_t st)
{
	if (n_new) hipLaunchKernelGGL(k_bf_test, dim3(grid_for(n_new)), dim3(256), 0, st, tab, newlist, n_new, bf, miss);
}

void yk_launch_bf_set(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
                      u32 *multi, int multi_bits, u64 *counters, hipStream_t st)
{
	if (n_new) hipLaunchKernelGGL(k_bf_set, dim3(grid_for(n_new)), dim3(256), 0, st, tab, newlist, n_new, bf, miss, multi, multi_bits, counters);
}

void yk_launch_bf_check(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
                        const u32 *multi, int multi_bits, u64 *cand, u64 *counters, hipStream_t st)
{
	if (n_new) hipLaunchKernelGGL(k_bf_check, dim3(grid_for(n_new)), dim3(256), 0, st, tab, newlist, n_new, bf, miss, multi, multi_bits, cand, counters);
}

void yk_launch_bf_mapfill(AccTab tab, const u64 *newlist, u64 n_new, BloomView bf, const u64 *miss,
                          const u32 *multi, int multi_bits, u64 *map, int map_bits, hipStream_t st)
{
	if (n_new) hipLaunchKernelGGL(k_bf_mapfill, dim3(grid_for(n_new)), dim3(256), 0, st, tab, newlist, n_new, bf, miss, multi, multi_bits, map, map_bits);
}

void yk_launch_bf_resolve(AccTab tab, const u64 *newlist, const u64 *cand, u64 n_cand, BloomView bf,
                          const u64 *miss, const u64 *map, int map_bits, hipStream_t st)
{
	if (n_cand) hipLaunchKernelGGL(k_bf_resolve, dim3(grid_for(n_cand)), dim3(256), 0, st, tab, newlist, cand, n_cand, bf, miss, map, map_bits);
}

void yk_launch_select_count(AccTab tab, int bloom_mode, int P, u32 *seg_cnt, hipStream_t st)
{
	(void)P;
	hipLaunchKernelGGL(k_select_count, dim3((unsigned)((tab.mask + SEL_CHUNK) / SEL_CHUNK)), dim3(256), 0, st, tab, bloom_mode, seg_cnt);
}

void yk_launch_select_scatter(AccTab tab, int bloom_mode, int P, const u64 *seg_off, u32 *seg_cur,
                              u64 *rec_kc, u64 *rec_t, hipStream_t st)
{
	(void)P;
	hipLaunchKernelGGL(k_select_scatter, dim3((unsigned)((tab.mask + SEL_CHUNK) / SEL_CHUNK)), dim3(256), 0, st, tab, bloom_mode, seg_off, seg_cur, rec_kc, rec_t);
}

void yk_launch_seg_sort_pass(const u64 *seg_off, int P, const u64 *src_kc, const u64 *src_t,
                             u64 *dst_kc, u64 *dst_t, int shift, hipStream_t st)
{
	hipLaunchKernelGGL((k_seg_sort_pass<8, 256>), dim3(P), dim3(256), 0, st, seg_off, (const u32*)0, src_kc, src_t, dst_kc, dst_t, shift);
}

void yk_launch_replay(const ReplayTask *tasks, int n_tasks, int n_threads, const u64 *old_keys, const u32 *old_used,
                      u64 *new_keys, u32 *new_used, u32 *scr_used, u32 *scr_owner, u64 *scr_par,
                      const u64 *rec_kc, const u64 *rec_t, const u64 *lastput,
                      u32 *out_bits, u32 *out_count, u32 lds_words, hipStream_t st)
{
	static bool attr = false;
	if (!attr) { hipFuncSetAttribute((const void*)k_replay, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); attr = true; }
	if (lds_words < 2 * RP_LDS_WORDS) lds_words = 2 * RP_LDS_WORDS;
	hipLaunchKernelGGL(k_replay, dim3(n_tasks), dim3(n_threads), (size_t)lds_words * 4, st, tasks, old_keys, old_used, new_keys, new_used,
	                   scr_used, scr_owner, scr_par, rec_kc, rec_t, lastput, out_bits, out_count, lds_words);
}

void yk_launch_shrink_count(ImgView img, int P, int cmin, int cmax, int which, ImgView other, u32 *seg_cnt, hipStream_t st)
{
	hipLaunchKernelGGL(k_shrink_count, dim3(P), dim3(256), 0, st, img, cmin, cmax, which, other, seg_cnt);
}

void yk_launch_shrink_scatter(ImgView img, int P, int cmin, int cmax, int which, ImgView other, const u64 *seg_off, u64 *rec_kc, hipStream_t st)
{
	hipLaunchKernelGGL(k_shrink_scatter, dim3(P), dim3(256), 0, st, img, cmin, cmax, which, other, seg_off, rec_kc);
}

void yk_launch_resize(const ResizeTask *tasks, int P, const u64 *old_keys, const u32 *old_used, u64 *new_keys, u32 *new_used, u32 *scr_used, hipStream_t st)
{
	hipLaunchKernelGGL(k_resize, dim3(P), dim3(256), 0, st, tasks, old_keys, old_used, new_keys, new_used, scr_used);
}

void yk_launch_keys_to_hashes(const u64 *kc, const u64 *seg_off, int P, int pre, u64 *hash, u32 *t, hipStream_t st)
{
	hipLaunchKernelGGL(k_keys_to_hashes, dim3(P), dim3(256), 0, st, kc, seg_off, pre, hash, t);
}

void yk_launch_part2(const Chunk2 *chunks, int n_chunks, const u32 *chunk_first, const u64 *bbase, FastParams fp, int P,
                     u32 *rows2, u64 *sbstart, Rec *out, hipStream_t st)
{
	const size_t lds = sizeof(u32) << fp.s2_bits;
	static const int nt = getenv("YAKAMD_P2_THREADS") ? atoi(getenv("YAKAMD_P2_THREADS")) : 1024;
	static bool attr = false;
	if (!attr) { hipFuncSetAttribute((const void*)k_part2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); attr = true; }
	if (n_chunks) hipLaunchKernelGGL(k_part2<0>, dim3(n_chunks), dim3(256), lds, st, chunks, fp, rows2, out);
	hipLaunchKernelGGL(k_part2_scan, dim3(P), dim3(256), 0, st, chunk_first, bbase, fp.s2_bits, rows2, sbstart, P);
	static const int wc = getenv("YAKAMD_P2_WC") ? atoi(getenv("YAKAMD_P2_WC")) : 1;
	if (n_chunks && wc && fp.rec8_out && fp.s2_bits <= 13 && fp.s2_bits >= 4) {
		static bool attr8 = false;                                  /* the kernel has 6.5 KB of static LDS besides */
		if (!attr8) { if (hipFuncSetAttribute((const void*)k_part2_wc8, hipFuncAttributeMaxDynamicSharedMemorySize, WC_SEG * WC8_CAP * 8 + WC_SEG * 8 + WC_NT * 4 + 16) != hipSuccess) (void)hipGetLastError(); attr8 = true; }
		const size_t seg = fp.s2_bits < 11 ? (size_t)1 << fp.s2_bits : WC_SEG;
		hipLaunchKernelGGL(k_part2_wc8, dim3(n_chunks), dim3(WC_NT), seg * WC8_CAP * 8 + seg * 8 + WC_NT * 4 + 16, st, chunks, fp, (const u32*)rows2, (u64*)out);
	} else if (n_chunks && wc && fp.s2_bits <= 13 && fp.s2_bits >= 4) {
		static bool attr2 = false;
		if (!attr2) { hipFuncSetAttribute((const void*)k_part2_wc, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); attr2 = true; }
		const size_t l2 = wc_lds_bytes<4, WC_CAP, true>(fp.s2_bits < 11 ? 1 << fp.s2_bits : WC_SEG, WC_NT);
		hipLaunchKernelGGL(k_part2_wc, dim3(n_chunks), dim3(WC_NT), l2, st, chunks, fp, (const u32*)rows2, (const u64*)sbstart, out);
	} else if (n_chunks) hipLaunchKernelGGL(k_part2<1>, dim3(n_chunks), dim3(nt), lds, st, chunks, fp, rows2, out);
}

int yk_rng_log(void)                      /* log2 slots per range; the env knob lets tests split small tables */
{
	static const int v = getenv("YAKAMD_RNG_LOG") ? atoi(getenv("YAKAMD_RNG_LOG")) : RNG_LOG;
	return v < 5 ? 5 : v > RNG_LOG ? RNG_LOG : v;
}
int yk_hpart2_chunk(void) { return 131072; }

void yk_launch_hpart2(const Chunk2 *chunks, int n_chunks, const u32 *chunk_first, const u64 *bbase, ImgView img, int rb, int P,
                      u32 *rows2, u64 *sbstart, u64 *out, hipStream_t st)
{
	const int S2 = 1 << rb;
	static bool attr = false;
	if (!attr) { hipFuncSetAttribute((const void*)k_hpart2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); attr = true; }
	if (n_chunks) hipLaunchKernelGGL(k_hpart2<0>, dim3(n_chunks), dim3(WC_NT), sizeof(u32) * S2, st, chunks, img, rb, yk_rng_log(), rows2, (const u64*)sbstart, out);
	hipLaunchKernelGGL(k_part2_scan, dim3(P), dim3(256), 0, st, chunk_first, bbase, rb, rows2, sbstart, P);
	if (n_chunks) hipLaunchKernelGGL(k_hpart2<1>, dim3(n_chunks), dim3(WC_NT), (wc_lds_bytes<8, XW_CAP_H, false>(S2, WC_NT)), st,
	                                 chunks, img, rb, yk_rng_log(), rows2, (const u64*)sbstart, out);
}

int yk_launch_img_count_rng(const u64 *rec, int cross, const u64 *sbstart, ImgView img, int plo, int phi, int rb, u32 max_len,
                            u64 *list, u32 *list_n, u32 list_cap, hipStream_t st)
{
	static bool attr = false;
	if (!attr) {
		if (hipFuncSetAttribute((const void*)k_img_count_rng<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess ||
		    hipFuncSetAttribute((const void*)k_img_count_rng<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) { (void)hipGetLastError(); return -1; }
		attr = true;
	}
	const size_t lds = (size_t)((max_len + 31) / 32) * 4 + (size_t)(max_len + 1) / 2 * 4 + 16;
	const dim3 grid((unsigned)(phi - plo) << rb), blk(1024);
	if (cross) hipLaunchKernelGGL(k_img_count_rng<1>, grid, blk, lds, st, rec, sbstart, img, plo, rb, yk_rng_log(), list, list_n, list_cap);
	else hipLaunchKernelGGL(k_img_count_rng<0>, grid, blk, lds, st, rec, sbstart, img, plo, rb, yk_rng_log(), list, list_n, list_cap);
	return 0;
}

/* tier = LC_G over every sub-bucket of the shard (in_list == NULL), or LC_S over a list */
void yk_launch_lds_count(int tier, FastParams fp, const u64 *sbstart, const Rec *rec,
                         u32 *bloom32, ImgView img, LcOut O, u64 *counters, const u32 *in_list, u32 n_list, u32 *ovf_list, hipStream_t st)
{
	if (tier == 0) {
		const unsigned n_sb = (unsigned)(fp.phi - fp.plo) << fp.s2_bits;
		hipLaunchKernelGGL(k_lds_count<LC_G>, dim3(n_sb), dim3(256), 0, st, fp, sbstart, rec, bloom32, img,
		                   O, counters, (const u32*)0, ovf_list, (int)YKC_NOVF);
	} else if (n_list) {
		hipLaunchKernelGGL(k_lds_count<LC_S>, dim3(n_list), dim3(256), 0, st, fp, sbstart, rec, bloom32, img,
		                   O, counters, in_list, ovf_list, (int)YKC_NOVF2);
	}
}

void yk_launch_lds_count_ovf(FastParams fp, const u64 *sbstart, const Rec *rec,
                             u32 *bloom32, ImgView img, LcOut O, const u32 *ovf_list, u32 n_ovf, const u64 *scr_off,
                             u64 *scr, hipStream_t st)
{
	if (n_ovf) hipLaunchKernelGGL(k_lds_count_ovf, dim3(n_ovf), dim3(256), 0, st, fp, sbstart, rec, bloom32, img,
	                              O, ovf_list, scr_off, scr);
}

size_t yk_count_own_lds(u32 range_len, u32 kmax) { const u32 nw = (range_len + 31) / 32; return (size_t)2 * ((nw + 1) & ~1u) * 4 + (size_t)kmax * 8 + (size_t)((kmax + 1) / 2 + 1 & ~1u) * 4 + 16 * 128 * 8 + 16; }   /* bitmap, ranks, keys, counters, 16 wave queues */

int yk_launch_img_count_own(const void *rec, int hash_only, int cross, int ytag, const u64 *bstart, ImgView img, int plo, int phi, int rb, int rng_log, u32 kmax,
                            size_t lds, u64 *list, u32 *list_n, u32 list_cap, hipStream_t st)
{
	static bool attr = false;
	if (!attr) {
		const void *fn[4] = { (const void*)k_img_count_own<1, 0>, (const void*)k_img_count_own<1, 1>, (const void*)k_img_count_own<2, 0>, (const void*)k_img_count_own<2, 1> };
		for (int i = 0; i < 4; ++i) if (hipFuncSetAttribute(fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) { (void)hipGetLastError(); return -1; }
		attr = true;
	}
	const int n_p = phi - plo;
	const dim3 grid((unsigned)((n_p + 7) / 8 * 8) << rb), blk(1024);
#define YK_OWN(Wv, Cv) hipLaunchKernelGGL((k_img_count_own<Wv, Cv>), grid, blk, lds, st, (const u64*)rec, bstart, img, plo, n_p, rb, rng_log, kmax, list, list_n, list_cap, ytag)
	if (hash_only) { if (cross) YK_OWN(1, 1); else YK_OWN(1, 0); }
	else { if (cross) YK_OWN(2, 1); else YK_OWN(2, 0); }
#undef YK_OWN
	return 0;
}

/* ---- replay2 launchers (grids: x = blocks per sub-table, y = sub-table) ---- */
void yk_r2_dinit(const R2Tab *tabs, const R2Act *acts, int P, u32 bmax, u64 *K0, u64 *K1, u32 *TAG, u32 *OCC, hipStream_t st)
{
	const u64 N = 2ull << bmax;
	hipLaunchKernelGGL(k_r2_dinit, dim3((unsigned)std::min<u64>((N + 1023) / 1024, 4096), P), dim3(256), 0, st, tabs, acts, K0, K1, TAG, OCC);
}
void yk_r2_dsmall(const R2Tab *tabs, const R2Act *acts, int P, u64 *K0, u64 *K1, u32 *TAG, u32 *OCC, u32 *Fcur, u32 *Gcur, u32 *fail, hipStream_t st)
{
	static const int defer = getenv("YAKAMD_R2_DEFER") ? atoi(getenv("YAKAMD_R2_DEFER")) : 1;   /* 0: the prefix lane follows every chain to its end */
	hipLaunchKernelGGL(k_r2_dsmall, dim3(P), dim3(256), 0, st, tabs, acts, K0, K1, TAG, OCC, Fcur, Gcur, fail, (u32)yk_r2_small_f(), defer);
}
int yk_r2_small_f(void)                                            /* YAKAMD_R2_SMALL_F: test / tuning knob, a power of two in [16, 4096] */
{
	const char *e = getenv("YAKAMD_R2_SMALL_F");
	int v = e ? atoi(e) : 16;   /* the fused rounds (k_r2_double) take over right behind the literal prefix: their chunk routine places a round's runs side by side where k_r2_dsmall walks them lane by lane */
	if (v < 16) v = 16;
	if (v > 4096) v = 4096;
	while (v & (v - 1)) v &= v - 1;
	return v;
}
/* the rounds of a doubling step from k_r2_dsmall's end (Fin) on in one launch (k_r2_double): n_dbl = sub-tables that double in this step */
int yk_r2_double(const R2Tab *tabs, const R2Act *acts, int P, int n_dbl, u64 *K0, u64 *K1, u32 *TAG, u32 *OCC, const u32 *Fin, u32 *Fout, u32 *fail, hipStream_t st)
{
	static const int force_nw = getenv("YAKAMD_R2_NW") ? atoi(getenv("YAKAMD_R2_NW")) : 0;
	/* many sub-tables: 6 waves each, four workgroups per CU (all 1024 sub-tables of a default table resident at once); few, large ones (a shard of
	 * a multi-GPU job): 16 waves each */
	const int nw = force_nw ? force_nw : n_dbl <= 256 ? 16 : 6;
	static u64 *d_prof = 0;
	static const bool prof = getenv("YAKAMD_VERBOSE") && atoi(getenv("YAKAMD_VERBOSE")) > 1;
	if (prof && !d_prof) { hipMalloc((void**)&d_prof, 16 * 8); }
	if (prof) hipMemsetAsync(d_prof, 0, 16 * 8, st);
#define YK_DBL(NWv, PRv) hipLaunchKernelGGL((k_r2_double<NWv, PRv>), dim3(P), dim3(64 * NWv), 0, st, tabs, acts, K0, K1, TAG, (const u32*)OCC, (const u32*)Fin, Fout, fail, d_prof)
	if (nw >= 16) { if (prof) YK_DBL(16, true); else YK_DBL(16, false); }
	else { if (prof) YK_DBL(6, true); else YK_DBL(6, false); }
#undef YK_DBL
	if (prof) {
		u64 h[16];
		hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, st);
		hipStreamSynchronize(st);
		if (h[8]) fprintf(stderr, "[yak_amd] k_r2_double<%d> 100 MHz ticks per wave (%llu waves): stage %llu, window+scans %llu, probing %llu, write-back %llu, medium runs %llu, round total %llu, barrier %llu, big runs %llu\n",
		                  nw, (unsigned long long)h[8], (unsigned long long)(h[0] / h[8]), (unsigned long long)(h[1] / h[8]), (unsigned long long)(h[2] / h[8]), (unsigned long long)(h[3] / h[8]),
		                  (unsigned long long)(h[4] / h[8]), (unsigned long long)(h[5] / h[8]), (unsigned long long)(h[6] / h[8]), (unsigned long long)(h[7] / h[8]));
	}
	return 0;
}
int yk_r2_seg_log(void) { const int v = getenv("YAKAMD_R2_SEG_LOG") ? atoi(getenv("YAKAMD_R2_SEG_LOG")) : R2_SEG_LOG; return v < 10 ? 10 : v > R2_SEG_LOG ? R2_SEG_LOG : v; }   /* the knob lets tests split small tables */
int yk_r2_head(void) { const u32 seg = 1u << yk_r2_seg_log(); return (int)(seg / 2 < R2_HEAD ? seg / 2 : R2_HEAD); }
void yk_r2_place(const R2Tab *tabs, const R2Act *acts, int P, u32 bmax, u64 *K0, u64 *K1, const u64 *kc, u64 *pk, u32 *pr, u32 *seg_start,
                 u32 *head, u64 *spill, u32 *spill_n, u32 spill_cap, u32 *fail, hipStream_t st)
{
	static bool attr = false;
	if (!attr) { hipFuncSetAttribute((const void*)k_r2_place, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + 256); attr = true; }
	const u32 SL = (u32)yk_r2_seg_log(), HD = (u32)yk_r2_head();
	const u32 nseg = bmax > SL ? 1u << (bmax - SL) : 1, L = bmax > SL ? 1u << SL : 1u << bmax;
	hipMemsetAsync(spill_n, 0, 4, st);
	if (nseg > 1) hipLaunchKernelGGL(k_r2_ppart, dim3(P), dim3(1024), 0, st, tabs, acts, kc, pk, pr, seg_start, SL);
	static const int pt = getenv("YAKAMD_R2_PLACE_THREADS") ? std::min(1024, std::max(64, atoi(getenv("YAKAMD_R2_PLACE_THREADS")) & ~63)) : 1024;
	hipLaunchKernelGGL(k_r2_place, dim3(nseg, P), dim3(pt), (size_t)L * 4, st, tabs, acts, K0, K1, kc, (const u64*)pk, (const u32*)pr, (const u32*)seg_start, head, spill, spill_n, spill_cap, fail, SL, HD);
	if (nseg > 1) {
		hipLaunchKernelGGL(k_r2_spill, dim3(64), dim3(256), 0, st, acts, (const u64*)spill, (const u32*)spill_n, spill_cap, head, fail, HD);
		hipLaunchKernelGGL(k_r2_headfill, dim3(nseg, P), dim3(256), 0, st, tabs, acts, K0, K1, kc, (const u32*)head, SL, HD);
	}
}
void yk_r2_load(const R2Tab *tabs, const R2Load *ld, int P, u32 bmax, const u64 *src1, const u64 *src2, u64 *K0, u64 *K1, hipStream_t st)
{
	const u64 n = 1ull << bmax;
	hipLaunchKernelGGL(k_r2_load, dim3((unsigned)std::min<u64>((n + 1023) / 1024, 4096), P), dim3(256), 0, st, tabs, ld, src1, src2, K0, K1);
}
void yk_r2_trail(const u64 *lastput, const u64 *rec_t, const u64 *rec_off, const u32 *m, int P, u32 *out, hipStream_t st)
{
	hipLaunchKernelGGL(k_r2_trail, dim3((P + 255) / 256), dim3(256), 0, st, lastput, rec_t, rec_off, m, P, out);
}
void yk_r2_publish(const R2Tab *tabs, const R2Pub *pub, int P, u32 bmax, const u64 *K0, const u64 *K1, u64 *nk, u32 *nu, hipStream_t st)
{
	const u64 n = 1ull << bmax;
	hipLaunchKernelGGL(k_r2_publish, dim3((unsigned)std::min<u64>((n + 1023) / 1024, 4096), P), dim3(256), 0, st, tabs, pub, K0, K1, nk, nu);
}

int yk_lc2_ok(FastParams fp)
{
	const int on = getenv("YAKAMD_LC2") ? atoi(getenv("YAKAMD_LC2")) : 1;          /* 0: the older three-tier kernels (tests) */
	if (!on || fp.n_hash > 32) return 0;
	if (fp.bloom_mode) { const int lb = fp.nb - 9 - fp.s2_bits; if (lb < 0 || lb > 7) return 0; }
	return 1;
}

void yk_launch_lc2(FastParams fp, const u64 *sbstart, const Rec *rec, u32 *bloom32, ImgView img, LcOut O, u64 *counters, u32 *ovf_list, hipStream_t st)
{
	const unsigned n_sb = (unsigned)(fp.phi - fp.plo) << fp.s2_bits;
	const int wgs = getenv("YAKAMD_LC2_WGS") ? atoi(getenv("YAKAMD_LC2_WGS")) : 256 * 5;   /* 5 workgroups of 32 KB LDS per CU */
	const unsigned grid = n_sb < (unsigned)wgs ? n_sb : (unsigned)wgs;
	if (grid) hipLaunchKernelGGL(k_lc2, dim3(grid), dim3(256), 0, st, fp, sbstart, rec, bloom32, img, O, counters, ovf_list, n_sb);
	if (grid && (fp.dbg & 128)) {
		u64 h[8];
		hipStreamSynchronize(st);
		hipMemcpyFromSymbol(h, HIP_SYMBOL(d_lc2_prof), sizeof(h));
		if (h[4]) fprintf(stderr, "[yak_amd] k_lc2 clocks per sub-bucket (lane 0, mean over %llu): A %llu, gate %llu, set+select %llu, write-back+clean %llu; inside A: wait for the records %llu, lane 0's puts %llu\n", (unsigned long long)h[4],
		                  (unsigned long long)(h[0] / h[4]), (unsigned long long)(h[1] / h[4]), (unsigned long long)(h[2] / h[4]), (unsigned long long)(h[3] / h[4]), (unsigned long long)(h[5] / h[4]), (unsigned long long)(h[6] / h[4]));
		for (int i = 0; i < 8; ++i) h[i] = 0;
		hipMemcpyToSymbol(HIP_SYMBOL(d_lc2_prof), h, sizeof(h));
	}
}

void yk_launch_lc_sum(const u32 *nsel, int s2_bits, int plo, int phi, u32 *seg_cnt, hipStream_t st)
{
	hipLaunchKernelGGL(k_lc_sum, dim3(phi - plo), dim3(256), 0, st, nsel, s2_bits, plo, seg_cnt);
}

void yk_launch_lc_compact(LcOut O, const u64 *sbstart, int s2_bits, int plo, int phi, u64 t_pass0, const u64 *seg_base,
                          u64 *out_kc, u64 *out_T, u64 *lastput, u32 *ndist_p, hipStream_t st)
{
	hipLaunchKernelGGL(k_lc_compact, dim3(phi - plo), dim3(256), 0, st, O, sbstart, s2_bits, plo, t_pass0, seg_base, out_kc, out_T, lastput, ndist_p);
}

void yk_launch_seg_sort_pass2(const u64 *seg_base, const u32 *seg_cnt, int P, const u64 *src_kc, const u64 *src_t,
                              u64 *dst_kc, u64 *dst_t, int shift, hipStream_t st, int big)
{
	/* long segments (an assembly: ~1 M keys per sub-table): 1024 threads per sub-table */
	if (big) hipLaunchKernelGGL((k_seg_sort_pass<8, 1024>), dim3(P), dim3(1024), 0, st, seg_base, seg_cnt, src_kc, src_t, dst_kc, dst_t, shift);
	else hipLaunchKernelGGL((k_seg_sort_pass<8, 256>), dim3(P), dim3(256), 0, st, seg_base, seg_cnt, src_kc, src_t, dst_kc, dst_t, shift);
}
void yk_launch_cnt2(FastParams fp, const u64 *sbstart, const Rec *rec, const u64 *key_off, const u64 *key_kc, const u64 *seg_base, u32 *key_cnt, ImgView img, hipStream_t st)
{
	const u32 n_sb = (u32)(fp.phi - fp.plo) << fp.s2_bits;
	static const int wgs = getenv("YAKAMD_CNT2_WGS") ? atoi(getenv("YAKAMD_CNT2_WGS")) : 256 * 8;   /* 28 KB of LDS, 256 threads: five workgroups per CU and then some waiting */
	hipLaunchKernelGGL(k_cnt2, dim3(std::min<u32>(n_sb, (u32)std::max(1, wgs))), dim3(256), 0, st, fp, sbstart, rec, key_off, key_kc, img, n_sb, key_cnt);
	const int n_p = fp.phi - fp.plo;
	const int per = std::max(1, 8192 / std::max(1, n_p));                  /* ~8 K workgroups in all */
	hipLaunchKernelGGL(k_cnt2_apply, dim3(per, n_p), dim3(256), 0, st, key_kc, (const u32*)key_cnt, seg_base, fp.plo, fp.pre, img);
}
void yk_launch_nsel_scan(const u32 *nsel, int s2_bits, int plo, int phi, int P, const u64 *seg_base, u64 *key_off, hipStream_t st)
{
	hipLaunchKernelGGL(k_nsel_scan, dim3(P), dim3(256), 0, st, nsel, s2_bits, plo, phi, P, seg_base, key_off);
}

void yk_launch_fill_u64(u64 *p, u64 v, u64 n, hipStream_t st)
{
	if (n) hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(n)), dim3(256), 0, st, p, v, n);
}

} /* extern "C" */
