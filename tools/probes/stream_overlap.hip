// Do kernels of different streams run side by side on this box?  N streams, each one launch of G single-wavefront workgroups that spin for ~T ms.
// Build + run (on the GPU box): hipcc --offload-arch=gfx950 -O2 tools/probes/stream_overlap.hip -o /tmp/stream_overlap && /tmp/stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int *sink)
{
	const long long t0 = wall_clock64();
	int x = 0;
	while (wall_clock64() - t0 < cycles) ++x;
	if (x == -1) *sink = x;
}
static double run(int n_streams, int grid, long long cycles, int launches)
{
	std::vector<hipStream_t> st(n_streams);
	for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	int *sink; hipMalloc(&sink, 4);
	for (auto &s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, sink);
	hipDeviceSynchronize();
	const auto t0 = std::chrono::steady_clock::now();
	for (int l = 0; l < launches; ++l) for (auto &s : st) hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s, cycles, sink);
	hipDeviceSynchronize();
	const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	for (auto &s : st) hipStreamDestroy(s);
	hipFree(sink);
	return ms;
}
int main()
{
	const long long cyc = 2000000;   // 20 ms at the 100 MHz wall clock
	for (int grid : {256, 2048, 8000})
		for (int n : {1, 2, 3, 4, 6})
			printf("grid %5d streams %d launches/stream 4: %.1f ms (one stream alone: ~%d ms)\n", grid, n, run(n, grid, cyc, 4), 80);
	return 0;
}
