// LDS operation throughput on gfx950 (random addresses, 20 waves per CU): clocks per lane-op per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64; typedef unsigned int u32;
template <int OP, int SAME>   // SAME: log2 of lanes sharing an address (0 = all distinct)
__global__ __launch_bounds__(256) void k(u32 *out, int iters)
{
	__shared__ u32 a[2048];
	__shared__ u64 b[1024];
	for (int i = threadIdx.x; i < 2048; i += 256) a[i] = 0x7fffffffu;
	for (int i = threadIdx.x; i < 1024; i += 256) b[i] = ~0ull;
	__syncthreads();
	u32 x = (threadIdx.x >> SAME) * 2654435761u + blockIdx.x * 40503u, acc = 0;
	for (int it = 0; it < iters; ++it) {
		x = x * 1664525u + 1013904223u;
		const u32 s = (x >> 12) & 1023u;
		if (OP == 0) acc += a[s];                                       // ds_read_b32
		if (OP == 1) acc += (u32)b[s];                                  // ds_read_b64
		if (OP == 2) acc += atomicAdd(&a[s], 1u);                       // ds_add_rtn_u32
		if (OP == 3) atomicAdd(&a[s], 1u);                              // ds_add_u32 (no return)
		if (OP == 4) acc += atomicMin(&a[s], x >> 9);                   // ds_min_rtn_u32
		if (OP == 5) atomicMin(&a[s], x >> 9);                          // ds_min_u32
		if (OP == 6) acc += (u32)atomicCAS(&b[s], ~0ull, (u64)x);       // ds_cmpst_rtn_b64
		if (OP == 7) acc += (u32)atomicMin(&b[s], (u64)x << 20);        // ds_min_rtn_u64
		if (OP == 8) a[s] = x;                                          // ds_write_b32
		if (OP == 9) { acc += atomicAdd(&a[s], 1u); acc += atomicMin(&a[1024 + s], x >> 9); }   // two independent returning atomics
	}
	if (acc == 0x12345u) out[0] = acc;
}
template <int OP, int SAME> void run(const char *name, u32 *d)
{
	const int iters = 4096, grid = 256 * 5;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k<OP, SAME>), dim3(grid), dim3(256), 0, 0, d, 64);
	hipEventRecord(e0);
	hipLaunchKernelGGL((k<OP, SAME>), dim3(grid), dim3(256), 0, 0, d, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double ops = (double)grid * 256 * iters * (OP == 9 ? 2 : 1);
	printf("%-28s same-address lanes %2d: %7.3f ms  %6.2f G lane-ops/s  %5.2f clk per lane-op per CU (2.4 GHz, 256 CUs)\n", name, 1 << SAME, ms, ops / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / ops);
}
int main()
{
	u32 *d; hipMalloc(&d, 4);
	run<0, 0>("ds_read_b32", d); run<1, 0>("ds_read_b64", d); run<8, 0>("ds_write_b32", d);
	run<2, 0>("ds_add_rtn_u32", d); run<3, 0>("ds_add_u32", d); run<4, 0>("ds_min_rtn_u32", d); run<5, 0>("ds_min_u32", d);
	run<6, 0>("ds_cmpst_rtn_b64", d); run<7, 0>("ds_min_rtn_u64", d); run<9, 0>("add_rtn + min_rtn", d);
	run<2, 2>("ds_add_rtn_u32", d); run<3, 2>("ds_add_u32", d); run<4, 2>("ds_min_rtn_u32", d); run<2, 4>("ds_add_rtn_u32", d); run<3, 4>("ds_add_u32", d);
	return 0;
}
