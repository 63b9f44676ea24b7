// do the runtime's own memset / memcpy work on a virtual range that several physical chunks back (requests that cross chunk boundaries)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__global__ void fill(u64 *p, size_t n, u64 v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void check(const u64 *p, size_t n, u64 v, u64 *bad) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != v) atomicAdd(bad, 1ull); }
static const size_t C = (size_t)256 << 20, N = 4;
static void *range(std::vector<hipMemGenericAllocationHandle_t> &h)
{
	hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	hipMemAccessDesc d = {}; d.location.type = hipMemLocationTypeDevice; d.location.id = 0; d.flags = hipMemAccessFlagsProtReadWrite;
	void *A = 0; hipMemAddressReserve(&A, N * C, 0, 0, 0);
	h.resize(N);
	for (size_t i = 0; i < N; ++i) { hipMemCreate(&h[i], C, &prop, 0); hipMemMap((char*)A + i * C, C, 0, h[i], 0); }
	hipMemSetAccess(A, N * C, &d, 1);
	return A;
}
static u64 *g_bad;
static u64 bad(const void *p, size_t bytes, u64 v) { hipMemset(g_bad, 0, 8); hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, 0, (const u64*)p, bytes / 8, v, g_bad); u64 b = 0; hipMemcpy(&b, g_bad, 8, hipMemcpyDeviceToHost); return b; }
int main()
{
	hipMalloc(&g_bad, 8);
	std::vector<hipMemGenericAllocationHandle_t> ha, hb;
	char *A = (char*)range(ha), *B = (char*)range(hb);
	hipStream_t st; hipStreamCreate(&st);
	hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (u64*)A, N * C / 8, 7ull); hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (u64*)B, N * C / 8, 9ull); hipDeviceSynchronize();
	hipError_t e;
	e = hipMemsetAsync(A, 0xff, N * C, st); hipStreamSynchronize(st);
	printf("memset of the whole range (4 chunks): rc %d, wrong words %llu\n", (int)e, bad(A, N * C, ~0ull));
	e = hipMemsetAsync(A + C / 2, 0, C, st); hipStreamSynchronize(st);
	printf("memset across one chunk boundary: rc %d, wrong words inside %llu, before %llu, behind %llu\n", (int)e, bad(A + C / 2, C, 0), bad(A, C / 2, ~0ull), bad(A + C / 2 + C, N * C - C - C / 2, ~0ull));
	e = hipMemsetD32Async((hipDeviceptr_t)(A + 3 * C - 4096), 0x01010101, 2048, st); hipStreamSynchronize(st);
	printf("memsetD32 of 8 KB across a chunk boundary: rc %d, wrong words %llu\n", (int)e, bad(A + 3 * C - 4096, 8192, 0x0101010101010101ull));
	e = hipMemcpyAsync(B + C / 4, A + C / 2, 2 * C, hipMemcpyDeviceToDevice, st); hipStreamSynchronize(st);
	printf("device-to-device copy of 2 chunks' worth between two ranges, both ends inside chunks: rc %d, wrong words %llu + %llu, untouched before %llu behind %llu\n", (int)e, bad(B + C / 4, C, 0), bad(B + C / 4 + C, C, ~0ull),
	       bad(B, C / 4, 9), bad(B + C / 4 + 2 * C, N * C - C / 4 - 2 * C, 9));
	std::vector<u64> host(C / 8 * 2, 5);
	e = hipMemcpyAsync(A + C - 4096, host.data(), 8192, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
	printf("host-to-device copy of 8 KB across a chunk boundary: rc %d, wrong words %llu\n", (int)e, bad(A + C - 4096, 8192, 5));
	e = hipMemcpyAsync(A + C / 2, host.data(), 2 * C, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
	printf("host-to-device copy of 512 MB across two boundaries: rc %d, wrong words %llu\n", (int)e, bad(A + C / 2, 2 * C, 5));
	hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (u64*)A, N * C / 8, 3ull); hipDeviceSynchronize();
	e = hipMemcpyAsync(host.data(), A + C / 2, 2 * C, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
	size_t w = 0; for (u64 v : host) w += v != 3;
	printf("device-to-host copy of 512 MB across two boundaries: rc %d, wrong words %zu\n", (int)e, w);
	void *pl = 0; hipMalloc(&pl, 2 * C);
	e = hipMemcpyAsync(pl, A + C / 2, 2 * C, hipMemcpyDeviceToDevice, st); hipStreamSynchronize(st);
	printf("copy from the range into hipMalloc memory: rc %d, wrong words %llu\n", (int)e, bad(pl, 2 * C, 3));
	e = hipMemcpyAsync(B + C / 2, pl, 2 * C, hipMemcpyDeviceToDevice, st); hipStreamSynchronize(st);
	printf("copy from hipMalloc memory into the range: rc %d, wrong words %llu\n", (int)e, bad(B + C / 2, 2 * C, 3));
	printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
	return 0;
}
