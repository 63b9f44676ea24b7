// micro-benchmark: scattered stores of 8/16-byte records in contiguous groups of `run` records.
// Answers: how much does a partition kernel gain if same-bucket records leave the CU together?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
template <int W>
__global__ __launch_bounds__(256) void k_sc(const u64 *__restrict__ in, u64 *__restrict__ out, u64 n, int lrun, int lbits)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const u64 g = i >> lrun, w = i & ((1ull << lrun) - 1);
		const u64 pg = (g * 0x9E3779B97F4A7C15ull) >> (64 - (lbits - lrun));   // bijection on groups (odd multiplier, top bits)... not exact bijection but spreads
		const u64 d = (pg << lrun) | w;
		if (W == 2) { const ulonglong2 v = ((const ulonglong2*)in)[i]; ((ulonglong2*)out)[d] = v; }
		else out[d] = in[i];
	}
}
int main()
{
	const int lbits = 29; const u64 n = 1ull << lbits;
	u64 *in, *out; hipMalloc(&in, n * 16); hipMalloc(&out, n * 16);
	hipMemset(in, 1, n * 16); hipMemset(out, 0, n * 16);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int W = 1; W <= 2; ++W)
		for (int lrun = 0; lrun <= 6; ++lrun) {
			float best = 1e9;
			for (int rep = 0; rep < 3; ++rep) {
				hipEventRecord(a);
				if (W == 2) hipLaunchKernelGGL(k_sc<2>, dim3(256 * 16), dim3(256), 0, 0, in, out, n, lrun, lbits);
				else hipLaunchKernelGGL(k_sc<1>, dim3(256 * 16), dim3(256), 0, 0, in, out, n, lrun, lbits);
				hipEventRecord(b); hipEventSynchronize(b);
				float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
			}
			printf("rec %2d B  run %2d (%4d B)  %.2f ms  %.1f G rec/s  %.0f GB/s stored\n", W * 8, 1 << lrun, (W * 8) << lrun, best, n / best / 1e6, n * W * 8 / best / 1e6);
		}
	return 0;
}
