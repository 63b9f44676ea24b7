// How fast are random one-line probes as a function of the table's footprint?  (k_lookup, cfg5: 1.5 G lookups into the 1.07 GB khashl image at ~44 G/s --
// would a lookup-only structure of 0.4 GB, or one below the 256 MiB Infinity Cache, answer faster?)  Every lane hashes a counter to a random aligned
// slot of `entry` bytes in a region of R bytes, keeps U loads in flight, and sums what it reads.  Prints G probes/s per region size.
//   hipcc --offload-arch=gfx950 -O2 mb_probe.hip -o mb_probe && ./mb_probe [probes in millions = 1500]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int U, int W>   // W = 8-byte words read per probe (1: one slot; 2: 16 B; 8: a 64-byte bucket as four 16-byte loads)
__global__ __launch_bounds__(256) void probe(const u64 *__restrict__ tab, u64 n_slots, u64 per_lane, u64 *out)
{
	const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
	u64 acc = 0;
	for (u64 i = 0; i < per_lane; i += U) {
		u64 v[U][W];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const u64 s = (mix(gid * per_lane + i + u) % n_slots) * W;
#pragma unroll
			for (int w = 0; w < W; ++w) v[u][w] = tab[s + w];
		}
#pragma unroll
		for (int u = 0; u < U; ++u)
#pragma unroll
			for (int w = 0; w < W; ++w) acc += v[u][w];
	}
	if (acc == 0x1234567) out[0] = acc;
}
__global__ void fill(u64 *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull; }
template <int U, int W> static double run(const u64 *tab, u64 bytes, u64 n_probes, u64 *out)
{
	const u64 lanes = 256ull * 256 * 32, per_lane = (n_probes / lanes + U - 1) / U * U;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL((probe<U, W>), dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, tab, bytes / (8 * W), per_lane, out);
	hipEventRecord(a, 0);
	hipLaunchKernelGGL((probe<U, W>), dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, tab, bytes / (8 * W), per_lane, out);
	hipEventRecord(b, 0); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	return (double)(lanes * per_lane) / (ms * 1e-3) / 1e9;
}
int main(int argc, char **argv)
{
	const u64 n_probes = (argc > 1 ? atol(argv[1]) : 1500) * 1000000ull;
	u64 *tab = 0, *out = 0;
	const u64 maxb = 4ull << 30;
	if (hipMalloc(&tab, maxb) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
	hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, tab, maxb / 8);
	hipDeviceSynchronize();
	printf("%10s %14s %14s %14s %14s   (G probes/s; %llu M probes per launch)\n", "region", "8 B x2", "8 B x4", "16 B x4", "64 B x2", (unsigned long long)(n_probes / 1000000));
	for (u64 mb : { 32ull, 64ull, 128ull, 192ull, 256ull, 320ull, 400ull, 512ull, 768ull, 1024ull, 2048ull, 4096ull }) {
		const u64 bytes = mb << 20;
		printf("%7llu MB %14.1f %14.1f %14.1f %14.1f\n", (unsigned long long)mb, run<2, 1>(tab, bytes, n_probes, out), run<4, 1>(tab, bytes, n_probes, out), run<4, 2>(tab, bytes, n_probes, out), run<2, 8>(tab, bytes, n_probes, out));
		fflush(stdout);
	}
	return 0;
}
