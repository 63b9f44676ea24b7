// Does hipMemCreate go faster from several threads?  (The first job of a process pays ~14-30 ms per GB of physical memory beyond the first ~112 GB: if the
// driver's work per chunk runs in parallel, a helper that creates the pool's chunks while the job's first buffers are filled would hide most of it.)
//   hipcc --offload-arch=gfx950 -O2 mb_vmm5.hip -o mb_vmm5 -lpthread && ./mb_vmm5 [GB = 200]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(unsigned long long *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 3 + i; }
int main(int argc, char **argv)
{
	const size_t gb = argc > 1 ? atol(argv[1]) : 200, C = (size_t)256 << 20, n = (gb << 30) / C;
	hipFree(0);
	hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	unsigned long long *busy = 0; hipMalloc(&busy, (size_t)1 << 30);
	for (int T : { 1, 2, 4, 8, 1 }) {
		for (int with_kernels = 0; with_kernels < 2; ++with_kernels) {
			std::vector<hipMemGenericAllocationHandle_t> h(n);
			std::atomic<size_t> next{0}, failed{0};
			std::vector<double> t_at(n, 0.0);
			std::atomic<bool> stop{false};
			std::thread kt;
			if (with_kernels) kt = std::thread([&]() { hipSetDevice(0); while (!stop) { hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, 0, busy, ((size_t)1 << 30) / 8); hipDeviceSynchronize(); } });
			const double t0 = now();
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([&]() {
				hipSetDevice(0);
				for (;;) { const size_t i = next++; if (i >= n) break; if (hipMemCreate(&h[i], C, &prop, 0) != hipSuccess) { ++failed; h[i] = 0; } t_at[i] = now() - t0; }
			});
			for (auto &x : th) x.join();
			const double dt = now() - t0;
			stop = true; if (with_kernels) kt.join();
			size_t first_half = 0; for (size_t i = 0; i < n; ++i) if (t_at[i] < dt / 2) ++first_half;
			printf("%zu GB in chunks of 256 MiB, %d thread(s)%s: %8.1f ms (%5.2f ms per GB; %zu of %zu chunks in the first half of the time; %zu failed)\n", gb, T, with_kernels ? ", kernels running" : "", dt * 1e3, dt * 1e3 / gb, first_half, n, (size_t)failed);
			const double tr = now();
			for (size_t i = 0; i < n; ++i) if (h[i]) hipMemRelease(h[i]);
			printf("   release %.1f ms\n", (now() - tr) * 1e3);
		}
	}
	return 0;
}
