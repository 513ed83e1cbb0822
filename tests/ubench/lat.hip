// dev microbenchmark: latency of wave-uniform dependent chains on gfx950 (cycles per step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t v32 __attribute__((vector_size(128)));

__global__ __launch_bounds__(64) void k_lat(uint32_t *out, unsigned long long *cyc, int n, int mode, const uint32_t *init)
{
	__shared__ uint32_t lds[4096];
	const uint32_t lane = threadIdx.x;
	v32 b0, b1, b2;
	for (int q = 0; q < 32; ++q) { b0[q] = init[q * 64 + lane]; b1[q] = init[2048 + q * 64 + lane]; b2[q] = init[4096 + q * 64 + lane]; }
	for (int i = lane; i < 4096; i += 64) lds[i] = init[i];
	__syncthreads();
	uint32_t i = 1, acc = 0;
	const unsigned long long t0 = clock64();
	if (mode == 0) {            // readlane only, single register
		for (int k = 0; k < n; ++k) { i = __builtin_amdgcn_readlane(b0[0], i & 63); acc += i; }
	} else if (mode == 1) {     // movrel + readlane, one bank
		for (int k = 0; k < n; ++k) { i = __builtin_amdgcn_readlane(b0[(i >> 6) & 31], i & 63); acc += i; }
	} else if (mode == 2) {     // 3 banks movrel + readlane + select
		for (int k = 0; k < n; ++k) {
			const uint32_t q = (i >> 6) & 31, l = i & 63, bank = (i >> 11) % 3;
			uint32_t r = __builtin_amdgcn_readlane(b0[q], l);
			const uint32_t r1 = __builtin_amdgcn_readlane(b1[q], l), r2 = __builtin_amdgcn_readlane(b2[q], l);
			r = bank == 1 ? r1 : r; r = bank == 2 ? r2 : r;
			i = r; acc += i;
		}
	} else if (mode == 3) {     // LDS dependent read (uniform address)
		for (int k = 0; k < n; ++k) { i = lds[i & 4095]; i = __builtin_amdgcn_readfirstlane(i); acc += i; }
	} else if (mode == 4) {     // LDS dependent read, per-lane (no readfirstlane)
		uint32_t j = lane;
		for (int k = 0; k < n; ++k) { j = lds[j & 4095]; acc += j; }
		i = j;
	} else if (mode == 5) {     // global dependent read per lane (L2-resident)
		uint32_t j = lane;
		for (int k = 0; k < n; ++k) { j = init[j & 4095]; acc += j; }
		i = j;
	} else if (mode == 6) {     // SALU-only dependent chain
		for (int k = 0; k < n; ++k) { i = (i * 1664525u + 1013904223u) >> 3; i = __builtin_amdgcn_readfirstlane(i); acc += i; }
	} else if (mode == 7) {     // readlane + v_mov from sgpr + readlane (VALU->SGPR->VALU->SGPR)
		for (int k = 0; k < n; ++k) { uint32_t t = __builtin_amdgcn_readlane(b0[0], i & 63); uint32_t v = t + lane; i = __builtin_amdgcn_readlane(v, t & 63); acc += i; }
	}
	const unsigned long long t1 = clock64();
	if (lane == 0) { out[blockIdx.x] = acc + i; cyc[blockIdx.x] = t1 - t0; }
}

int main()
{
	const int n = 20000;
	std::vector<uint32_t> h(6144);
	uint32_t s = 12345;
	for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) % 6144; }
	uint32_t *d_init, *d_out; unsigned long long *d_cyc;
	hipMalloc(&d_init, h.size() * 4); hipMalloc(&d_out, 4096 * 4); hipMalloc(&d_cyc, 4096 * 8);
	hipMemcpy(d_init, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	const char *names[] = {"readlane", "movrel+readlane", "3x(movrel+readlane)+select", "lds uniform+readfirstlane", "lds per-lane", "global per-lane (L2)", "salu chain", "readlane->valu->readlane"};
	for (int blocks : {1, 1024, 4096})
		for (int mode = 0; mode < 8; ++mode) {
			hipLaunchKernelGGL(k_lat, dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, n, mode, d_init);
			hipDeviceSynchronize();
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			hipEventRecord(e0);
			hipLaunchKernelGGL(k_lat, dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, n, mode, d_init);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			std::vector<unsigned long long> c(blocks);
			hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
			double mean = 0; for (auto x : c) mean += (double)x; mean /= blocks;
			printf("blocks %5d  %-30s  %8.1f clk/step  kernel %.3f ms -> %.1f ns/step\n", blocks, names[mode], mean / n, ms, ms * 1e6 / n);
		}
	return 0;
}
