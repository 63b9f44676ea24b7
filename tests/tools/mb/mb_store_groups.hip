// microbenchmark: rate of aligned group stores to scattered addresses (G lanes x 8 bytes per group), the write pattern of the partition kernels.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tests/tools/mb/mb_store_groups.hip -o /tmp/mb && /tmp/mb   (write-only; mb_scatter.hip next to it copies, i.e. reads too)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
template <int G>
__global__ __launch_bounds__(1024) void k_scatter(u64 *out, u64 n_groups, u64 gmask, int reps)
{
	const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x, nthr = (u64)gridDim.x * blockDim.x;
	const u64 lane_in = tid % G;
	for (int r = 0; r < reps; ++r)
		for (u64 g = tid / G; g < n_groups; g += nthr / G) {
			const u64 dst = mix(g + (u64)r * n_groups) & gmask;       // scattered group index
			out[dst * G + lane_in] = g;
		}
}
int main()
{
	const u64 bytes = 8ull << 30;
	u64 *d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMemset(d, 0, bytes);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
#define RUN(G, NT, NB) { const u64 ng = bytes / (G * 8); \
	k_scatter<G><<<NB, NT>>>(d, ng / 8, ng - 1, 1); hipDeviceSynchronize(); \
	hipEventRecord(a); k_scatter<G><<<NB, NT>>>(d, ng, ng - 1, 1); hipEventRecord(b); hipEventSynchronize(b); \
	float ms; hipEventElapsedTime(&ms, a, b); \
	printf("G=%2d (%3d B groups) %4d thr x %5d wg: %.2f ms  %.1f G records/s  %.2f G groups/s  %.0f GB/s\n", G, G * 8, NT, NB, ms, bytes / 8 / ms / 1e6, ng / ms / 1e6, bytes / ms / 1e6); }
	RUN(4, 1024, 512) RUN(8, 1024, 512) RUN(16, 1024, 512) RUN(32, 1024, 512) RUN(64, 1024, 512)
	RUN(8, 1024, 256) RUN(16, 1024, 256) RUN(8, 256, 2048) RUN(16, 256, 2048) RUN(8, 256, 8192) RUN(16, 256, 8192)
	return 0;
}
