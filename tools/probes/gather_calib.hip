// What does rocprofv3's FETCH_SIZE report per access CLASS on this chip?  The guide calibrates one class only: wide coalesced streams report half their
// bytes (128-byte requests tallied at 64).  The mapping path's largest readers are gathers - 8- and 16-byte records at random places of arrays far larger
// than the 256 MB Infinity Cache - so this probe runs, each as its own kernel (name = class) over a 2 GiB array:
//   cal_stream16 : every lane 16 consecutive bytes, coalesced (the guide's case)              bytes asked = 16 per access
//   cal_stream8  : every lane 8 consecutive bytes, coalesced                                  8
//   cal_gather8  : every lane one 8-byte record at a pseudo-random index                      8    (what it costs in HBM traffic: at least one 32-byte sector)
//   cal_gather16 : one 16-byte record at a pseudo-random index                                16
//   cal_gather8x2: two 8-byte records 8 bytes apart... no: 64 bytes apart (a pointer chase's second look into the line next door)
// Run under  rocprofv3 --pmc FETCH_SIZE  (and, where the counter list has them, TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum: request counts by size) and divide
// per kernel: profiles/collect_pmc.py does that and applies the factor of its class to every kernel of the path.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O2 tools/probes/gather_calib.hip -o /tmp/gather_calib && /tmp/gather_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static constexpr uint64_t BYTES = 2ull << 30;          // 2 GiB: 8 x the Infinity Cache
static constexpr uint32_t ACCESSES = 1u << 26;         // per kernel: 64 M accesses (a gather touches 64 M different 64-byte lines of the 32 M the array has ... twice over)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void cal_stream16(const uint4 *a, uint32_t *sink) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; const uint4 v = a[i]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = 1; }
__global__ void cal_stream8(const uint2 *a, uint32_t *sink) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; const uint2 v = a[i]; if ((v.x ^ v.y) == 0x12345u) *sink = 1; }
__global__ void cal_gather8(const uint2 *a, uint32_t *sink) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; const uint2 v = a[mix(i) % (BYTES / 8)]; if ((v.x ^ v.y) == 0x12345u) *sink = 1; }
__global__ void cal_gather16(const uint4 *a, uint32_t *sink) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; const uint4 v = a[mix(i) % (BYTES / 16)]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = 1; }
__global__ void cal_gather8x2(const uint2 *a, uint32_t *sink)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, j = mix(i) % (BYTES / 8 - 16);
	const uint2 v = a[j], w = a[j + 8];                     // 64 bytes apart
	if ((v.x ^ v.y ^ w.x ^ w.y) == 0x12345u) *sink = 1;
}
int main()
{
	void *buf; uint32_t *sink;
	CHECK(hipMalloc(&buf, BYTES)); CHECK(hipMalloc(&sink, 4));
	CHECK(hipMemset(buf, 0, BYTES)); CHECK(hipMemset(sink, 0, 4));
	const dim3 blk(256), grd(ACCESSES / 256);
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float ms;
	#define RUN(kern, T, asked) do { hipLaunchKernelGGL(kern, grd, blk, 0, 0, (const T*)buf, sink); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, grd, blk, 0, 0, (const T*)buf, sink); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); \
		printf("%-14s accesses %u  bytes asked %llu  %.3f ms  %.1f G accesses/s  %.1f GB/s asked\n", #kern, ACCESSES, (unsigned long long)ACCESSES * (asked), ms, ACCESSES / ms / 1e6, (double)ACCESSES * (asked) / ms / 1e6); } while (0)
	RUN(cal_stream16, uint4, 16);
	RUN(cal_stream8, uint2, 8);
	RUN(cal_gather8, uint2, 8);
	RUN(cal_gather16, uint4, 16);
	RUN(cal_gather8x2, uint2, 16);
	return 0;
}
