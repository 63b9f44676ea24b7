// producer kernel / consumer kernel on different XCDs: does a consumer see what the kernel before it wrote, on hipMalloc memory and on a mapped range?
// (workgroup b runs on XCD b % 8; the consumer's workgroup b reads what producer workgroup b + 1, + 3, ... wrote)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__global__ __launch_bounds__(256) void produce(u64 *p, size_t per_wg, u64 gen) { u64 *q = p + (size_t)blockIdx.x * per_wg; for (size_t i = threadIdx.x; i < per_wg; i += 256) q[i] = gen * 0x9E3779B97F4A7C15ull + blockIdx.x * 1000003ull + i; }
__global__ __launch_bounds__(256) void consume(const u64 *p, size_t per_wg, u64 gen, unsigned shift, u64 *bad)
{
	const unsigned src = (blockIdx.x + shift) % gridDim.x;
	const u64 *q = p + (size_t)src * per_wg;
	u64 b = 0;
	for (size_t i = threadIdx.x; i < per_wg; i += 256) b += q[i] != gen * 0x9E3779B97F4A7C15ull + src * 1000003ull + i;
	if (b) atomicAdd(bad, b);
}
static u64 test(u64 *p, size_t bytes, u64 *bad, hipStream_t st)
{
	const unsigned wgs = 4096; const size_t per_wg = bytes / 8 / wgs;
	u64 total = 0;
	for (u64 gen = 1; gen <= 12; ++gen) {
		hipMemsetAsync(bad, 0, 8, st);
		hipLaunchKernelGGL(produce, dim3(wgs), dim3(256), 0, st, p, per_wg, gen);
		hipLaunchKernelGGL(consume, dim3(wgs), dim3(256), 0, st, (const u64*)p, per_wg, gen, (unsigned)(gen % 8), bad);
		u64 b = 0; hipMemcpyAsync(&b, bad, 8, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
		total += b;
	}
	return total;
}
int main()
{
	const size_t C = (size_t)256 << 20, N = 8, bytes = N * C;
	hipStream_t st; hipStreamCreate(&st);
	u64 *bad = 0; hipMalloc(&bad, 8);
	u64 *pl = 0; hipMalloc(&pl, bytes);
	printf("hipMalloc memory: %llu stale words in 12 producer/consumer pairs\n", test(pl, bytes, bad, st));
	hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	hipMemAccessDesc d = {}; d.location.type = hipMemLocationTypeDevice; d.location.id = 0; d.flags = hipMemAccessFlagsProtReadWrite;
	std::vector<hipMemGenericAllocationHandle_t> h(N);
	for (size_t i = 0; i < N; ++i) hipMemCreate(&h[i], C, &prop, 0);
	for (int round = 0; round < 3; ++round) {
		void *A = 0; hipMemAddressReserve(&A, bytes, 0, 0, 0);
		for (size_t i = 0; i < N; ++i) hipMemMap((char*)A + i * C, C, 0, h[(i * 3 + round) % N], 0);
		hipMemSetAccess(A, bytes, &d, 1);
		printf("mapped range, mapping %d: %llu stale words in 12 producer/consumer pairs\n", round, test((u64*)A, bytes, bad, st));
		hipMemUnmap(A, bytes); hipMemAddressFree(A, bytes);
	}
	return 0;
}
