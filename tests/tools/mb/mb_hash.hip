// microbenchmark: yak_hash64 (k = 31 mask) as the compiler lowers it (64-bit multiply-adds for x + (x << a) + (x << b)) against explicit
// shift-adds (v_lshl_add_u64).  build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tests/tools/mb/mb_hash.hip -o /tmp/mb_hash && /tmp/mb_hash
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
#define M62 0x3fffffffffffffffull
__device__ __forceinline__ u64 h_plain(u64 x)
{
	x = (~x + (x << 21)) & M62; x ^= x >> 24; x = (x + (x << 3) + (x << 8)) & M62; x ^= x >> 14;
	x = (x + (x << 2) + (x << 4)) & M62; x ^= x >> 28; x = (x + (x << 31)) & M62; return x;
}
__device__ __forceinline__ u64 lsa(u64 a, int sh, u64 b)   // (a << sh) + b, sh <= 4
{
	u64 r;
	switch (sh) {
	case 2: asm("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(r) : "v"(a), "v"(b)); break;
	case 3: asm("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(r) : "v"(a), "v"(b)); break;
	case 4: asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(r) : "v"(a), "v"(b)); break;
	default: asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b)); break;
	}
	return r;
}
__device__ __forceinline__ u64 shl(u64 a, int sh) { u64 r; asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "v"(sh), "v"(a)); return r; }
__device__ __forceinline__ u64 h_shift(u64 x)
{
	x = lsa(shl(x, 21), 0, ~x) & M62; x ^= x >> 24;
	x = lsa(shl(x, 8), 0, lsa(x, 3, x)) & M62; x ^= x >> 14;
	x = lsa(x, 4, lsa(x, 2, x)) & M62; x ^= x >> 28;
	x = lsa(shl(x, 31), 0, x) & M62; return x;
}
template <int V> __global__ __launch_bounds__(256) void k(u64 *out, int iters)
{
	u64 a = blockIdx.x * 256 + threadIdx.x, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3;
	for (int i = 0; i < iters; ++i) {
		if (V == 0) { a = h_plain(a); b = h_plain(b); c = h_plain(c); d = h_plain(d); }
		else { a = h_shift(a); b = h_shift(b); c = h_shift(c); d = h_shift(d); }
	}
	out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}
int main()
{
	u64 *d; (void)hipMalloc(&d, 8ull * 256 * 8192);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	u64 h[2][4];
	for (int v = 0; v < 2; ++v) {
		const int iters = 2000;
		if (v == 0) k<0><<<8192, 256>>>(d, 10); else k<1><<<8192, 256>>>(d, 10);
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0);
		if (v == 0) k<0><<<8192, 256>>>(d, iters); else k<1><<<8192, 256>>>(d, iters);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		(void)hipMemcpy(h[v], d, 32, hipMemcpyDeviceToHost);
		printf("%s: %.2f ms, %.1f G hashes/s\n", v ? "shift-adds" : "compiler (multiply-adds)", ms, 8192.0 * 256 * 4 * iters / ms / 1e6);
	}
	printf("same results: %s\n", h[0][0] == h[1][0] && h[0][3] == h[1][3] ? "yes" : "NO");
	return 0;
}
