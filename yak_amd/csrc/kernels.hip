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
 *
 * One translation unit, cut by stage into the kern_*.inc files included at the end of this file (device helpers and __device__ globals are
 * shared; separate device compilation would need relocatable device code for them).
 */
#include "yk_device.h"
#include <algorithm>
#include <type_traits>

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

/* A pointer read from a structure in memory has no known address space: the compiler then uses FLAT loads, which count on the LDS counter as
 * well -- every wait for an LDS result also waits for the global load, and a prefetch hides nothing.  Device buffers are global memory: say so */
#define YK_GLOBAL __attribute__((address_space(1)))
#define yk_global(T, p) ((const T YK_GLOBAL*)(p))
#define yk_global_rw(T, p) ((T YK_GLOBAL*)(p))
typedef u64 yk_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Rec yk_global_rec(const Rec *p, u64 i)                  /* one 16-byte global load */
{
	const yk_u64x2 v = ((const yk_u64x2 YK_GLOBAL*)p)[i];
	return make_ulonglong2(v.x, v.y);
}
#define YK_LDS __attribute__((address_space(3)))
#define yk_lds_rw(T, p) ((T YK_LDS*)(p))                                           /* "LDS or global" stores would otherwise be merged into one FLAT store */

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

#include "kern_extract.inc"
#include "kern_general.inc"
#include "kern_layout.inc"
#include "kern_replay2.inc"
#include "kern_count.inc"
#include "kern_pass2.inc"
#include "kern_launch.inc"
