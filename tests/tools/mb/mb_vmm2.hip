// does one hipMemUnmap over a range of several mapped chunks unmap them all?  (pool: an idle range gives its chunks up)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__global__ void fill(u64 *p, size_t n, u64 v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void peek(const u64 *p, u64 *out) { out[0] = p[0]; }
int main()
{
	const size_t C = (size_t)256 << 20, n = 4;
	hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	hipMemAccessDesc d = {}; d.location.type = hipMemLocationTypeDevice; d.location.id = 0; d.flags = hipMemAccessFlagsProtReadWrite;
	std::vector<hipMemGenericAllocationHandle_t> h(n);
	for (size_t i = 0; i < n; ++i) if (hipMemCreate(&h[i], C, &prop, 0) != hipSuccess) { printf("create failed\n"); return 1; }
	void *A = 0, *B = 0;
	hipMemAddressReserve(&A, n * C, 0, 0, 0);
	for (size_t i = 0; i < n; ++i) printf("map A+%zu: %d\n", i, (int)hipMemMap((char*)A + i * C, C, 0, h[i], 0));
	printf("access A: %d\n", (int)hipMemSetAccess(A, n * C, &d, 1));
	for (size_t i = 0; i < n; ++i) hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, (u64*)((char*)A + i * C), C / 8, 100 + i);
	hipDeviceSynchronize();
	printf("unmap A whole range in one call: %d (%s)\n", (int)hipMemUnmap(A, n * C), hipGetErrorString(hipGetLastError()));
	// is A + C still mapped?  mapping something else there must fail if it is
	hipMemGenericAllocationHandle_t x; hipMemCreate(&x, C, &prop, 0);
	const int r1 = (int)hipMemMap((char*)A + C, C, 0, x, 0);
	printf("map a new chunk at A+1 after the whole-range unmap: %d (0 = the address was free)\n", r1);
	if (r1 == 0) hipMemUnmap((char*)A + C, C);
	// chunk-wise unmap of whatever is left
	for (size_t i = 0; i < n; ++i) printf("unmap A+%zu alone: %d\n", i, (int)hipMemUnmap((char*)A + i * C, C));
	(void)hipGetLastError();
	hipMemAddressReserve(&B, n * C, 0, 0, 0);
	for (size_t i = 0; i < n; ++i) printf("map B+%zu (handle %zu): %d\n", i, n - 1 - i, (int)hipMemMap((char*)B + i * C, C, 0, h[n - 1 - i], 0));
	printf("access B: %d\n", (int)hipMemSetAccess(B, n * C, &d, 1));
	u64 *out = 0; hipMalloc(&out, 8);
	for (size_t i = 0; i < n; ++i) { u64 v = 0; hipLaunchKernelGGL(peek, dim3(1), dim3(1), 0, 0, (const u64*)((char*)B + i * C), out); hipMemcpy(&v, out, 8, hipMemcpyDeviceToHost); printf("B+%zu holds %llu (expected %zu)\n", i, v, 100 + (n - 1 - i)); }
	return 0;
}
