// What does it cost to move physical memory between virtual ranges?  (DESIGN.md "first job": a pool whose idle memory is fungible -- physical chunks
// behind virtual ranges -- instead of asking the driver for new memory when no idle range fits.)  Creates N physical chunks of C MiB once, then times, for a
// range of all of them: reserve + map + set-access, a fill, a random-probe sweep (is the translation as good as hipMalloc's?), unmap + address-free.
//   hipcc --offload-arch=gfx950 -O2 mb_vmm.hip -o mb_vmm && ./mb_vmm [GB = 32]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(u64 *p, size_t n, u64 v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + i; }
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ __launch_bounds__(256) void probe(const u64 *__restrict__ tab, u64 n_slots, u64 per_lane, u64 *out)
{
	const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
	u64 acc = 0;
	for (u64 i = 0; i < per_lane; i += 4) { u64 v[4]; for (int u = 0; u < 4; ++u) v[u] = tab[mix(gid * per_lane + i + u) % n_slots]; for (int u = 0; u < 4; ++u) acc += v[u]; }
	if (acc == 0x1234567) out[0] = acc;
}
static void bench(const char *what, u64 *p, size_t bytes, u64 *out)
{
	double t = now(); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, p, bytes / 8, 1ull); hipDeviceSynchronize(); const double f1 = now() - t;
	t = now(); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, p, bytes / 8, 2ull); hipDeviceSynchronize(); const double f2 = now() - t;
	const u64 lanes = 256ull * 256 * 32, per_lane = 256;
	hipLaunchKernelGGL(probe, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, p, bytes / 8, per_lane, out); hipDeviceSynchronize();
	t = now(); hipLaunchKernelGGL(probe, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, p, bytes / 8, per_lane, out); hipDeviceSynchronize(); const double pr = now() - t;
	printf("  %-34s fill %6.2f / %6.2f ms per GB   random 8-byte probes %6.1f G/s\n", what, f1 * 1e3 / (bytes / 1073741824.0), f2 * 1e3 / (bytes / 1073741824.0), lanes * per_lane / pr / 1e9);
}
int main(int argc, char **argv)
{
	const size_t gb = argc > 1 ? atol(argv[1]) : 32, bytes = gb << 30;
	hipFree(0);
	u64 *out = 0; hipMalloc(&out, 8);
	{ u64 *p = 0; double t = now(); if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; } printf("hipMalloc of %zu GB: %.2f ms\n", gb, (now() - t) * 1e3); bench("hipMalloc", p, bytes, out); t = now(); hipFree(p); printf("  hipFree %.2f ms\n", (now() - t) * 1e3); }
	hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	size_t gran = 0; hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
	printf("recommended granularity %zu\n", gran);
	for (size_t cmb : { 2048, 1024, 512, 256, 64 }) {
		const size_t C = cmb << 20, n = bytes / C;
		std::vector<hipMemGenericAllocationHandle_t> h(n);
		double t = now();
		for (size_t i = 0; i < n; ++i) if (hipMemCreate(&h[i], C, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at %zu\n", i); return 1; }
		const double tc = now() - t;
		printf("chunks of %zu MiB: create %zu in %.2f ms (%.1f us each)\n", cmb, n, tc * 1e3, tc * 1e6 / n);
		for (int round = 0; round < 3; ++round) {
			void *va = 0;
			t = now();
			if (hipMemAddressReserve(&va, bytes, 0, 0, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
			const double tr = now() - t; t = now();
			for (size_t i = 0; i < n; ++i) if (hipMemMap((char*)va + i * C, C, 0, h[(i * 7 + round) % n], 0) != hipSuccess) { printf("map failed\n"); return 1; }   // a different order every round
			const double tm = now() - t; t = now();
			hipMemAccessDesc d = {}; d.location.type = hipMemLocationTypeDevice; d.location.id = 0; d.flags = hipMemAccessFlagsProtReadWrite;
			if (hipMemSetAccess(va, bytes, &d, 1) != hipSuccess) { printf("set access failed\n"); return 1; }
			const double ta = now() - t;
			if (round == 0 || round == 2) bench(round ? "mapped range (3rd mapping)" : "mapped range (1st mapping)", (u64*)va, bytes, out);
			t = now();
			if (hipMemUnmap(va, bytes) != hipSuccess) { printf("unmap failed\n"); return 1; }
			const double tu = now() - t; t = now();
			hipMemAddressFree(va, bytes);
			const double tf = now() - t;
			printf("  round %d: reserve %.3f ms, map %.3f ms (%.1f us per chunk), set access %.3f ms, unmap %.3f ms, address free %.3f ms\n", round, tr * 1e3, tm * 1e3, tm * 1e6 / n, ta * 1e3, tu * 1e3, tf * 1e3);
		}
		t = now();
		for (size_t i = 0; i < n; ++i) hipMemRelease(h[i]);
		printf("  release %.2f ms\n", (now() - t) * 1e3);
	}
	return 0;
}
