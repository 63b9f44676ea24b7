// What does the driver charge for device memory a process has not had before?  (~27 ms per GB through hipMalloc on a box whose
// memory was not handed out yet -- the first job of a process pays seconds for its 100-300 GB.)  Times, per GB: hipMalloc, a first-touch fill kernel, a
// second fill, hipFree; then the same through hipMallocAsync (stream-ordered pool), hipExtMallocWithFlags(uncached / default), and the virtual memory
// API (hipMemCreate + hipMemMap of 2 MiB-granular handles), and hipMalloc again after the frees (does the driver clear again?).
//   hipcc --offload-arch=gfx950 -O2 mb_malloc.hip -o mb_malloc && ./mb_malloc [GB per allocation = 16] [allocations = 4]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(unsigned long long *p, size_t n, unsigned long long v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
static double touch(void *p, size_t bytes) { const double t = now(); hipLaunchKernelGGL(fill, dim3(256 * 8), dim3(256), 0, 0, (unsigned long long*)p, bytes / 8, 1ull); hipDeviceSynchronize(); return now() - t; }
int main(int argc, char **argv)
{
	const size_t gb = argc > 1 ? atol(argv[1]) : 16, n = argc > 2 ? atol(argv[2]) : 4, bytes = gb << 30;
	hipFree(0);
	size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
	printf("device memory: %.1f GB free of %.1f\n", fr / 1e9, tot / 1e9);
	auto report = [&](const char *what, double tm, double t1, double t2, double tf) {
		printf("%-44s alloc %7.2f ms/GB   first fill %6.2f ms/GB   second fill %6.2f ms/GB   free %6.2f ms/GB\n", what, tm * 1e3 / gb, t1 * 1e3 / gb, t2 * 1e3 / gb, tf * 1e3 / gb);
		fflush(stdout);
	};
	for (int round = 0; round < 2; ++round) {
		std::vector<void*> ps;
		for (size_t i = 0; i < n; ++i) {
			void *p = 0; double t = now();
			if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc failed\n"); break; }
			const double tm = now() - t, t1 = touch(p, bytes), t2 = touch(p, bytes);
			ps.push_back(p);
			char w[64]; snprintf(w, sizeof(w), "hipMalloc #%zu (round %d)", i, round);
			report(w, tm, t1, t2, 0);
		}
		double t = now(); for (void *p : ps) hipFree(p); const double tf = (now() - t) / (ps.size() ? ps.size() : 1);
		printf("hipFree: %.2f ms/GB\n", tf * 1e3 / gb);
	}
	{	// stream-ordered allocator, release threshold = everything
		hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
		unsigned long long thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
		for (int round = 0; round < 2; ++round) {
			void *p = 0; double t = now();
			if (hipMallocAsync(&p, bytes, 0) != hipSuccess) { printf("hipMallocAsync failed\n"); (void)hipGetLastError(); break; }
			hipStreamSynchronize(0);
			const double tm = now() - t, t1 = touch(p, bytes), t2 = touch(p, bytes);
			t = now(); hipFreeAsync(p, 0); hipStreamSynchronize(0);
			report(round ? "hipMallocAsync (pool warm)" : "hipMallocAsync (pool cold)", tm, t1, t2, now() - t);
		}
		hipMemPoolTrimTo(pool, 0);
	}
	for (unsigned flags : { (unsigned)hipDeviceMallocDefault, (unsigned)hipDeviceMallocUncached }) {
		void *p = 0; double t = now();
		if (hipExtMallocWithFlags(&p, bytes, flags) != hipSuccess) { printf("hipExtMallocWithFlags(%u) failed\n", flags); (void)hipGetLastError(); continue; }
		const double tm = now() - t, t1 = touch(p, bytes), t2 = touch(p, bytes);
		t = now(); hipFree(p);
		report(flags == hipDeviceMallocDefault ? "hipExtMallocWithFlags(default)" : "hipExtMallocWithFlags(uncached)", tm, t1, t2, now() - t);
	}
	{	// virtual memory management: one reservation, physical handles of `gran` mapped into it
		hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
		size_t gran = 0; hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
		printf("VMM granularity %zu\n", gran);
		for (size_t piece : { bytes, (size_t)1 << 30 }) {
			hipDeviceptr_t va = 0; double t = now();
			if (hipMemAddressReserve(&va, bytes, gran, 0, 0) != hipSuccess) { printf("hipMemAddressReserve failed\n"); (void)hipGetLastError(); break; }
			std::vector<hipMemGenericAllocationHandle_t> hs;
			bool ok = true;
			for (size_t o = 0; o < bytes && ok; o += piece) {
				hipMemGenericAllocationHandle_t h;
				ok = hipMemCreate(&h, piece, &prop, 0) == hipSuccess && hipMemMap((char*)va + o, piece, 0, h, 0) == hipSuccess;
				if (ok) hs.push_back(h);
			}
			hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
			ok = ok && hipMemSetAccess(va, bytes, &ad, 1) == hipSuccess;
			if (!ok) { printf("VMM path failed: %s\n", hipGetErrorString(hipGetLastError())); break; }
			const double tm = now() - t, t1 = touch(va, bytes), t2 = touch(va, bytes);
			t = now(); hipMemUnmap(va, bytes); for (auto h : hs) hipMemRelease(h); hipMemAddressFree(va, bytes);
			report(piece == bytes ? "VMM: one handle" : "VMM: 1 GiB handles", tm, t1, t2, now() - t);
		}
	}
	{	void *p = 0; double t = now(); hipMalloc(&p, bytes); const double tm = now() - t, t1 = touch(p, bytes); t = now(); hipFree(p); report("hipMalloc after all of the above", tm, t1, 0, now() - t); }
	return 0;
}
