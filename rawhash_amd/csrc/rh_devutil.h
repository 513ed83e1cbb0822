// Wave64 / block helpers shared by the kernel sources.
#pragma once
#include "rh_gpu.h"
#include <cstdint>

#define NT 256   // threads of the block-cooperative kernels

// ------------------------------------------------------------------------------------------------ wave / block helpers
RH_DEV uint32_t lane_id() { return threadIdx.x & 63u; }
RH_DEV uint32_t wave_id() { return threadIdx.x >> 6; }
RH_DEV uint32_t lanes_below(uint64_t m) { return (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull)); }

// Order-preserving rank of the calling thread among the threads of the block with pred set; total = their number.
// s_w: LDS scratch of (blockDim.x / 64) words.  Contains two block barriers.
// a register pair that a load was issued into earlier: the compiler's wait for that load goes where this stands, and memory operations after it stay after it
#if defined(__HIP_DEVICE_COMPILE__)
#define RH_LANDED(a, b) asm volatile("" : "+v"(a), "+v"(b) : : "memory")
#define RH_LANDED3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c) : : "memory")
#else
#define RH_LANDED(a, b) ((void)0)
#define RH_LANDED3(a, b, c) ((void)0)
#endif

RH_DEV uint32_t block_rank(bool pred, uint32_t *s_w, uint32_t &total)
{
	const uint64_t m = __ballot(pred);
	const uint32_t r = lanes_below(m), w = wave_id(), nw = blockDim.x >> 6;
	if (lane_id() == 0) s_w[w] = (uint32_t)__popcll(m);
	__syncthreads();
	uint32_t base = 0;
	total = 0;
	for (uint32_t i = 0; i < nw; ++i) { const uint32_t c = s_w[i]; if (i < w) base += c; total += c; }
	__syncthreads();
	return base + r;
}


// Exclusive prefix sum over the block's NT (= 256) values, one per thread; total returned to every thread.
// s_w: LDS scratch of (NT / 64) words.  Two block barriers.
RH_DEV uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t &total)
{
	uint32_t inc = v;
	for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if (lane_id() >= (uint32_t)d) inc += t; }
	const uint32_t w = wave_id(), nw = blockDim.x >> 6;
	if (lane_id() == 63) s_w[w] = inc;
	__syncthreads();
	uint32_t base = 0;
	total = 0;
	for (uint32_t i = 0; i < nw; ++i) { const uint32_t c = s_w[i]; if (i < w) base += c; total += c; }
	__syncthreads();
	return base + inc - v;
}

// OR over the whole block of a 64-bit value.  s_r: LDS scratch of (NT / 64) u64.  Two block barriers.
RH_DEV uint64_t block_or64(uint64_t v, uint64_t *s_r)
{
	for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d);
	if (lane_id() == 0) s_r[wave_id()] = v;
	__syncthreads();
	uint64_t r = 0;
	for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) r |= s_r[i];
	__syncthreads();
	return r;
}

// ---- serial reproduction of radix_sort_128x (one thread), used for small auxiliary sorts and oversized reads
// Serial emulation of klib's in-place MSD radix sort (ksort.h:101-151): insertion sort up to 64 records, otherwise an
// "American flag" cycle-leader pass per byte from bit 56 down.  The permutation among equal keys is unstable but
// deterministic and is observed by the chaining DP and the backtracking order, so it is reproduced step by step.
RH_HD inline void rh_ins_sort128(rh_mm128_t *a, uint32_t beg, uint32_t end)
{
	for (uint32_t i = beg + 1; i < end; ++i) {
		if (a[i].x < a[i - 1].x) {
			const rh_mm128_t t = a[i];
			uint32_t j = i;
			while (j > beg && t.x < a[j - 1].x) { a[j] = a[j - 1]; --j; }
			a[j] = t;
		}
	}
}

// one American-flag pass over a[beg, end) on byte (s / 8); cw = 512 words of scratch
RH_HD inline void rh_af_pass(rh_mm128_t *a, uint32_t beg, uint32_t end, int s, uint32_t *cw)
{
	uint32_t *head = cw, *tail = cw + 256;
	for (int c = 0; c < 256; ++c) head[c] = 0;
	for (uint32_t i = beg; i < end; ++i) ++head[(a[i].x >> s) & 255u];
	uint32_t p = beg;
	for (int c = 0; c < 256; ++c) { const uint32_t n = head[c]; head[c] = p; p += n; tail[c] = p; }
	for (int c = 0; c < 256;) {
		if (head[c] == tail[c]) { ++c; continue; }
		uint32_t d = (uint32_t)(a[head[c]].x >> s) & 255u;
		if (d == (uint32_t)c) { ++head[c]; continue; }
		rh_mm128_t carry = a[head[c]];
		do {
			const uint32_t h = head[d]++;
			const rh_mm128_t ev = a[h];
			a[h] = carry;
			carry = ev;
			d = (uint32_t)(carry.x >> s) & 255u;
		} while (d != (uint32_t)c);
		a[head[c]++] = carry;
	}
}

RH_HD inline void rh_radix_sort_128x(rh_mm128_t *a, uint32_t n, uint32_t *cw)
{
	if (n <= 64) { rh_ins_sort128(a, 0, n); return; }
	struct frame { uint32_t beg, end, cur; int s; int passed; } st[9];
	int sp = 0;
	st[0].beg = 0; st[0].end = n; st[0].cur = 0; st[0].s = 56; st[0].passed = 0;
	while (sp >= 0) {
		frame &f = st[sp];
		if (!f.passed) {
			rh_af_pass(a, f.beg, f.end, f.s, cw);
			f.passed = 1; f.cur = f.beg;
			if (f.s == 0) { --sp; continue; }
		}
		if (f.cur >= f.end) { --sp; continue; }
		// next sub-bucket = maximal run sharing the byte just sorted on
		const uint32_t b = f.cur, c = (uint32_t)(a[b].x >> f.s) & 255u;
		uint32_t e = b + 1;
		while (e < f.end && ((uint32_t)(a[e].x >> f.s) & 255u) == c) ++e;
		f.cur = e;
		const uint32_t sz = e - b;
		const int ns = f.s > 8 ? f.s - 8 : 0;
		if (sz > 64) { ++sp; st[sp].beg = b; st[sp].end = e; st[sp].cur = b; st[sp].s = ns; st[sp].passed = 0; }
		else if (sz > 1) rh_ins_sort128(a, b, e);
	}
}

