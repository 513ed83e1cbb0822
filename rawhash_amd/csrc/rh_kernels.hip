// HIP kernels of the RawHash2 mapping path for gfx950 (wave64).  One launch per stage per chunk round over the
// batch's active reads.  Stage -> reference function it replaces:
//   k_prefilter   raw->pA + 30<pA<200 filter bookkeeping            rsig.c:496-503 (+ chunk boundaries of rmap.cpp:415-417)
//   k_events      normalise, prefix sums, t-stats, peaks, events     revent.c:221-316
//   k_sketch      quantise + pack + hash                              rsketch.c:143-204 / :55-141
//   k_probe       index lookup, mid_occ filter, rep_len               rseed.c:60-154, rindex.c:497-514
//   k_expand      hits -> anchors (+ carried anchors)                 rmap.cpp:74-116
//   k_sort        exact radix_sort_128x permutation                   ksort.h:101-151
//   k_chain       chaining DP                                         lchain.c:439-505
//   k_backtrack   backtrack + compaction                              lchain.c:95-281
//   k_regions     regions, parents, MAPQ, mapping decision            hit.c:100-367,502-539; rmap.cpp:423-500
//   k_finalize    record assembly                                     rmap.cpp:507-586
// Block-cooperative stages (prefilter, events, probe, expand) use wave ballots + LDS; the inherently serial,
// order-dependent stages run one read per lane.  FP is fp32/fp64 exactly where the reference uses them; the file must
// be compiled with -ffp-contract=off.
#include "rh_kernels.h"
#include "rh_devutil.h"

// f5 = 0: the SLOW5 / POD5 readers (rsig.c:497, :452): (raw + offset) * scale with a double offset, evaluated in double, narrowed.
// f5 = 1: the FAST5 reader (rsig.c:363-374): offset is a float there, so the whole expression is float arithmetic; the value that
//         passes the 30 < pA < 200 test is then written back into the reader's int16_t vector - truncated - before it becomes the
//         float signal (rh_pa_value below).  The filter looks at the untruncated value in both.
RH_DEV float raw_to_pa(int16_t raw, double cal_off, float cal_scale, uint32_t f5)
{
	if (f5) return ((float)raw + (float)cal_off) * cal_scale;
	return (float)(((double)raw + cal_off) * (double)cal_scale);
}
RH_DEV float rh_pa_value(float pa, uint32_t f5) { return f5 ? (float)(int16_t)pa : pa; }

// ------------------------------------------------------------------------------------------------ k_prefilter
// One block per read: count samples surviving the pA filter and record, for every chunk boundary, the raw index of the
// first surviving sample of that chunk.  Each wavefront owns a contiguous quarter of the read and works on tiles of 256
// samples (four coalesced 128-byte rows) with ballots only: pass 1 counts, one barrier turns the four counts into
// offsets, pass 2 re-reads the quarter (L2) and ranks just the tiles that hold a chunk boundary.
RH_DEV uint64_t pa_tile_ballot(const int16_t *raw, uint32_t i, uint32_t end, double coff, float cscale, uint32_t f5)
{
	bool valid = false;
	if (i < end) { const float pa = raw_to_pa(raw[i], coff, cscale, f5); valid = pa > 30.0f && pa < 200.0f; }
	return __ballot(valid);
}

// survivors of the pA filter among the 256 samples from `base` (clipped at end).  Counting needs no order: each lane takes
// 4 consecutive samples in one 8-byte load (512 B per request instead of 128)
RH_DEV uint32_t pa_tile_count(const int16_t *raw, uint32_t base, uint32_t end, double coff, float cscale, uint32_t f5)
{
	const uint32_t i0 = base + 4u * lane_id();
	int16_t v[4] = {0, 0, 0, 0};
	if (i0 + 4u <= end) __builtin_memcpy(v, raw + i0, 8);
	else { for (uint32_t k = 0; k < 4; ++k) if (i0 + k < end) v[k] = raw[i0 + k]; }
	uint32_t tc = 0;
#pragma unroll
	for (uint32_t k = 0; k < 4; ++k) {
		const float pa = raw_to_pa(v[k], coff, cscale, f5);
		tc += (uint32_t)__popcll(__ballot(i0 + k < end && pa > 30.0f && pa < 200.0f));
	}
	return tc;
}

#ifndef PF_TILES
#define PF_TILES 2048          // tiles of 256 samples whose counts fit LDS (reads up to 512 k samples)
#endif
// act / init / bad: consumed-prefix staging (rh_dev_reads::res_len) runs the kernel again over the reads k_fetch has just extended - the active list,
// per-read state left alone - on the res_len[r] samples that are resident; *bad counts the reads whose caller-given filtered length is not what
// the whole signal holds.
__global__ __launch_bounds__(NT) void k_prefilter(rh_dev_opt o, rh_dev_reads rd, const uint32_t *act, int init, unsigned long long *bad)
{
	__shared__ uint32_t s_tot[NT / 64];
	__shared__ uint16_t s_tc[PF_TILES];
	const uint32_t r = act ? act[blockIdx.x] : blockIdx.x, tid = threadIdx.x, w = wave_id(), l = lane_id();
	const uint64_t o0 = rd.off[r], n64 = rd.off[r + 1] - o0;
	const uint32_t n = rd.res_len ? rd.res_len[r] : (uint32_t)n64;
	const int16_t *raw = rd.raw + o0;
	const double coff = rd.cal_off[r];
	const float cscale = rd.cal_scale[r];
	const uint32_t f5 = rd.fast5;
	uint32_t *cs = rd.chunk_start + (size_t)r * rd.cs_stride;
	for (uint32_t k = tid; k < rd.cs_stride; k += NT) cs[k] = n;
	const uint32_t C = o.chunk_size;
	const uint32_t per = ((n + (NT / 64) * 256u - 1) / ((NT / 64) * 256u)) * 256u;   // samples per wavefront, whole tiles
	const uint32_t beg = w * per < n ? w * per : n, end = beg + per < n ? beg + per : n;
	const bool keep = (n + 255) / 256 <= PF_TILES;                // per-tile counts of pass 1 kept in LDS: pass 2 re-reads boundary tiles only
	uint32_t cnt = 0;
	for (uint32_t base = beg; base < end; base += 256) {
		const uint32_t tc = pa_tile_count(raw, base, end, coff, cscale, f5);
		if (keep && l == 0) s_tc[base >> 8] = (uint16_t)tc;
		cnt += tc;
	}
	if (l == 0) s_tot[w] = cnt;
	__syncthreads();                                              // (also orders the cs[] defaults before the boundary writes)
	uint32_t run = 0, total = 0;
	for (uint32_t q = 0; q < NT / 64; ++q) { const uint32_t c = s_tot[q]; if (q < w) run += c; total += c; }
	for (uint32_t base = beg; base < end; base += 256) {
		uint32_t tc = keep ? (uint32_t)s_tc[base >> 8] : 0u;
		const uint32_t nb = ((run + C - 1) / C) * C;               // first chunk boundary at or after this tile's first survivor
		if (!keep || nb < run + tc) {
			uint64_t B[4];
			tc = 0;
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) { B[k] = pa_tile_ballot(raw, base + k * 64 + l, end, coff, cscale, f5); tc += (uint32_t)__popcll(B[k]); }
			if (nb < run + tc) {
				uint32_t before = run;
#pragma unroll
				for (uint32_t k = 0; k < 4; ++k) {
					if ((B[k] >> l) & 1ull) {
						const uint32_t fi = before + lanes_below(B[k]);
						if (fi % C == 0) { const uint32_t kk = fi / C; if (kk < rd.cs_stride) cs[kk] = base + k * 64 + l; }
					}
					before += (uint32_t)__popcll(B[k]);
				}
			}
		}
		run += tc;
	}
	if (tid == 0) {
		if (rd.res_len) {
			rd.cnt_res[r] = total;
			rd.l_sig[r] = rd.l_sig_given[r];
			if (n == (uint32_t)n64 && total != rd.l_sig_given[r]) atomicAdd(bad, 1ull);
		} else {
			rd.l_sig[r] = total;
			if (rd.l_sig_given && total != rd.l_sig_given[r]) atomicAdd(bad, 1ull);
		}
		if (init) {
			rd.sum[r] = 0.0; rd.sum2[r] = 0.0; rd.n_sum[r] = 0; rd.ev_off[r] = 0; rd.n_prev[r] = 0; rd.prev_off[r] = 0;
			rd.done[r] = 0; rd.stop_chunk[r] = 0; rd.ls_ncregs[r] = 0;
		}
	}
}

// ------------------------------------------------------------------------------------------------ consumed-prefix staging
// A read that maps stops after one or two chunks (rmap.cpp:425/498), so most of a batch's raw signal is never looked at - except by the pA
// filter of the reader, whose survivor count is the sl:i tag (rsig.c:496-503).  A caller whose reader counted while it decoded
// (rh_read_batch_t::n_filtered) and left the samples in page-locked memory gets them fetched over PCIe BY THE DEVICE, stretch by stretch as the
// rounds need them: k_need marks the active reads whose resident stretch does not hold chunk `chunk` yet (the chunk's samples and the first one
// of the next chunk: cnt_res >= (chunk + 1) * chunk_size + 1, or the whole read) and says how far to extend them, k_fetch copies, k_prefilter
// re-ranks the extended reads.
__global__ __launch_bounds__(256) void k_need(rh_dev_opt o, rh_dev_reads rd, const uint32_t *act, uint32_t n, uint32_t chunk, uint32_t grow, uint32_t *new_len, uint32_t *n_need)
{
	const uint32_t a = blockIdx.x * 256u + threadIdx.x;
	if (a >= n) return;
	const uint32_t r = act ? act[a] : a;
	const uint32_t len = (uint32_t)(rd.off[r + 1] - rd.off[r]), res = rd.res_len[r];
	const uint64_t want = ((uint64_t)chunk + 1ull) * (uint64_t)o.chunk_size + 1ull;
	const bool need = res < len && (uint64_t)rd.cnt_res[r] < want;
	uint32_t to = res;
	if (need) { to = (uint64_t)res + grow < (uint64_t)len ? res + grow : len; atomicAdd(n_need, 1u); }
	new_len[r] = to;
}

#define FETCH_VEC 1024u       // 16-byte words per workgroup (16 KB)
__global__ __launch_bounds__(256) void k_fetch(rh_dev_reads rd, const int16_t *host, const uint32_t *act, uint32_t n, const uint32_t *new_len, uint32_t gy)
{
	const uint32_t a = blockIdx.x / gy, by = blockIdx.x % gy, tid = threadIdx.x;   // gy workgroups per read, 16 KB each
	if (a >= n) return;
	const uint32_t r = act ? act[a] : a;
	const uint32_t from = rd.res_len[r], to = new_len[r];
	if (to <= from) return;
	const uint64_t o0 = rd.off[r];
	const int16_t *src = host + o0;                                 // page-locked host memory, read across PCIe
	int16_t *dst = rd.raw_w + o0;                                   // (same misalignment as src: stage_reads places the buffer so)
	const uint32_t mis = (uint32_t)(((uintptr_t)(src + from) & 15u) >> 1);
	uint32_t head = mis ? 8u - mis : 0u;
	if (head > to - from) head = to - from;
	const uint32_t i0 = from + head, nvec = (to - i0) >> 3, tail0 = i0 + (nvec << 3);
	if (by == 0) {
		if (tid < head) dst[from + tid] = src[from + tid];
		if (tid >= 64u && tid - 64u < to - tail0) dst[tail0 + tid - 64u] = src[tail0 + tid - 64u];
	}
	const uint32_t v0 = by * FETCH_VEC, v1 = v0 + FETCH_VEC < nvec ? v0 + FETCH_VEC : nvec;
	const uint4 *sv = reinterpret_cast<const uint4*>(src + i0);
	uint4 *dv = reinterpret_cast<uint4*>(dst + i0);
	for (uint32_t v = v0 + tid; v < v1; v += 256u) dv[v] = sv[v];
}
__global__ __launch_bounds__(256) void k_fetch_commit(rh_dev_reads rd, const uint32_t *act, uint32_t n, const uint32_t *new_len)
{
	const uint32_t a = blockIdx.x * 256u + threadIdx.x;
	if (a >= n) return;
	const uint32_t r = act ? act[a] : a;
	rd.res_len[r] = new_len[r];
}

// number of chunk iterations the read goes through if no decision stops it (loop bounds of rmap.cpp:415)
RH_DEV uint32_t read_n_chunks(const rh_dev_opt &o, uint32_t qlen)
{
	if (qlen == 0) return 0;
	const uint32_t lc = o.chunk_size > qlen ? qlen : o.chunk_size;
	const uint32_t nc = (qlen + lc - 1) / lc;
	return nc < o.max_num_chunk ? nc : o.max_num_chunk;
}

// ------------------------------------------------------------------------------------------------ k_events
RH_DEV float tstat_at(const float *ps, const float *pss, uint32_t n, uint32_t w, uint32_t i)
{
	if (n < 2 * w || w < 2 || i < w || i > n - w) return 0.0f;
	float s1 = ps[i], q1 = pss[i];
	if (i > w) { s1 -= ps[i - w]; q1 -= pss[i - w]; }
	const float s2 = ps[i + w] - ps[i], q2 = pss[i + w] - pss[i];
	const float fw = (float)w;
	const float m1 = s1 / fw, m2 = s2 / fw;
	float var = (q1 / fw - m1 * m1 + q2 / fw - m2 * m2) / fw;
	var = fmaxf(var, FLT_MIN);
	const float dm = m2 - m1;
	return fabsf(dm) / sqrtf(var);
}

// Event detection runs as four launches so that every lane is busy in each of them:
//   k_events_norm   (one block per read)      pA filter, fp64 statistics, z-score + compaction -> z rows in HBM
//   k_events_tstat  (one LANE per read for the order-sensitive fp32 prefix sums, 64 chunks in lock step; the whole block
//                    for the t-statistics of the tile those sums just covered)  -> t1, t2 rows in HBM
//                   (windows wider than TS_WMAX: k_events_norm<true> does both in the one-block-per-read layout)
//   k_events_peaks  (one LANE per read)       the two coupled peak detectors, a 4000-step serial state machine per chunk:
//                                             64 chunks advance in lock step, their t-stat rows staged through LDS tiles
//   k_events_means  (one block per read)      per-segment sort + IQR-fenced mean -> events
#define EV_ROW (RH_CHUNK_MAX + 64)             // row stride (floats) of the z / t1 / t2 staging arrays

// dst[0] = 0, dst[i + 1] = dst[i] + (SQ ? z[i] * z[i] : z[i]) in fp32, strictly left to right, by ONE lane.  The chain of
// dependent adds is the floor (a lone wavefront gets a dependent VALU result every ~8 cycles); everything else is kept
// off it: the samples arrive 16 ahead in 16-byte LDS reads, the results leave in aligned 16-byte LDS writes.
template <bool SQ>
RH_DEV void serial_prefix(const float *z, float *dst, uint32_t n)
{
	float acc = 0.0f;
	dst[0] = 0.0f;
	uint32_t i = 0;
	#define SP_STEP(r, a) do { if (SQ) { r.x = acc = acc + a.x * a.x; r.y = acc = acc + a.y * a.y; r.z = acc = acc + a.z * a.z; r.w = acc = acc + a.w * a.w; } \
	                           else { r.x = acc = acc + a.x; r.y = acc = acc + a.y; r.z = acc = acc + a.z; r.w = acc = acc + a.w; } } while (0)
	if (n >= 16) {
		const float4 *z4 = reinterpret_cast<const float4*>(z);
		float4 a0 = z4[0], a1 = z4[1], a2 = z4[2], a3 = z4[3];
		for (; i + 32 <= n; i += 16) {	// the next 16 samples are requested before the 16 dependent adds of this round
			const float4 b0 = z4[i / 4 + 4], b1 = z4[i / 4 + 5], b2 = z4[i / 4 + 6], b3 = z4[i / 4 + 7];
			float4 r0, r1, r2, r3;
			SP_STEP(r0, a0); SP_STEP(r1, a1); SP_STEP(r2, a2); SP_STEP(r3, a3);
			*reinterpret_cast<float4*>(&dst[i + 1]) = r0; *reinterpret_cast<float4*>(&dst[i + 5]) = r1;
			*reinterpret_cast<float4*>(&dst[i + 9]) = r2; *reinterpret_cast<float4*>(&dst[i + 13]) = r3;
			a0 = b0; a1 = b1; a2 = b2; a3 = b3;
		}
		float4 r0, r1, r2, r3;
		SP_STEP(r0, a0); SP_STEP(r1, a1); SP_STEP(r2, a2); SP_STEP(r3, a3);
		*reinterpret_cast<float4*>(&dst[i + 1]) = r0; *reinterpret_cast<float4*>(&dst[i + 5]) = r1;
		*reinterpret_cast<float4*>(&dst[i + 9]) = r2; *reinterpret_cast<float4*>(&dst[i + 13]) = r3;
		i += 16;
	}
	#undef SP_STEP
	for (; i < n; ++i) { const float v = z[i]; acc = acc + (SQ ? v * v : v); dst[i + 1] = acc; }
}

template <bool FULL>
__global__ __launch_bounds__(NT) void k_events_norm(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ __attribute__((aligned(16))) float s_z[FULL ? RH_CHUNK_MAX : 4];
	__shared__ __attribute__((aligned(16))) float s_a[RH_CHUNK_MAX + 4];      // pA staging -> prefix sums (at +3)
	__shared__ __attribute__((aligned(16))) float s_b[RH_CHUNK_MAX + 4];      // prefix sums of squares (at +3)
	__shared__ uint32_t s_w[NT / 64];
	__shared__ double s_red[2 * (NT / 64)];
	__shared__ double s_stat[2];

	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t r = rr.act[a], c = rr.chunk;
	const uint64_t o0 = rd.off[r];
	const int16_t *raw = rd.raw + o0;
	const double coff = rd.cal_off[r];
	const float cscale = rd.cal_scale[r];
	const uint32_t f5 = rd.fast5;
	const uint32_t *cs = rd.chunk_start + (size_t)r * rd.cs_stride;
	const uint32_t cs0 = cs[c], cs1 = cs[c + 1];
	const uint32_t C = o.chunk_size;

	// 1. load, convert, filter, compact (order preserving) + fp64 partial sums (exact for 30<pA<200, any order).
	//    Each wavefront owns a contiguous quarter of the chunk's raw samples, in tiles of 256 (four coalesced rows); ranks
	//    come from ballots, and the only barrier is the one that turns the four counts into offsets.
	const uint32_t w = wave_id(), l = lane_id();
	uint32_t s_len;
	double dsum = 0.0, dsum2 = 0.0;
	{
		const uint32_t span = cs1 - cs0;
		const uint32_t per = ((span + (NT / 64) * 256u - 1) / ((NT / 64) * 256u)) * 256u;
		const uint32_t beg = cs0 + (w * per < span ? w * per : span), end = beg + per < cs1 ? beg + per : cs1;
		uint32_t cnt = 0;
		for (uint32_t base = beg; base < end; base += 256) cnt += pa_tile_count(raw, base, end, coff, cscale, f5);
		if (l == 0) s_w[w] = cnt;
		__syncthreads();
		uint32_t run = 0, count = 0;
		for (uint32_t q = 0; q < NT / 64; ++q) { const uint32_t c2 = s_w[q]; if (q < w) run += c2; count += c2; }
		for (uint32_t base = beg; base < end && run < C; base += 256) {
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint32_t i = base + k * 64 + l;
				bool valid = false; float pa = 0.0f;
				if (i < end) { pa = raw_to_pa(raw[i], coff, cscale, f5); valid = pa > 30.0f && pa < 200.0f; pa = rh_pa_value(pa, f5); }
				const uint64_t B = __ballot(valid);
				const uint32_t pos = run + lanes_below(B);
				if (valid && pos < C) {
					s_a[pos] = pa;
					dsum += (double)pa;
					const float sq = pa * pa;
					dsum2 += (double)sq;
				}
				run += (uint32_t)__popcll(B);
			}
		}
		s_len = count < C ? count : C;
	}
	for (int d = 32; d > 0; d >>= 1) { dsum += __shfl_down(dsum, d); dsum2 += __shfl_down(dsum2, d); }
	if (lane_id() == 0) { s_red[2 * wave_id()] = dsum; s_red[2 * wave_id() + 1] = dsum2; }
	__syncthreads();
	if (tid == 0) {
		double S = rd.sum[r], S2 = rd.sum2[r];
		for (uint32_t q = 0; q < NT / 64; ++q) { S += s_red[2 * q]; S2 += s_red[2 * q + 1]; }
		const uint32_t N = rd.n_sum[r] + s_len;
		rd.sum[r] = S; rd.sum2[r] = S2; rd.n_sum[r] = N;
		const double mean = S / N;
		s_stat[0] = mean;
		s_stat[1] = sqrt(S2 / N - mean * mean);
		atomicAdd((unsigned long long*)&rr.counters[5], (unsigned long long)s_len);
		atomicAdd((unsigned long long*)&rr.counters[6], 1ull);
	}
	__syncthreads();
	const double mean = s_stat[0], sd = s_stat[1];

	// 2. z-score, drop |z| >= 3, compact: same scheme; the z values of pass 1 wait in s_b for their slots (which are
	//    in the HBM row directly when the prefix sums are another launch's business)
	float *zrow = rr.zbuf + (size_t)a * rr.ev_row;
	uint32_t n;
	{
		const uint32_t per = ((s_len + (NT / 64) * 256u - 1) / ((NT / 64) * 256u)) * 256u;
		const uint32_t beg = w * per < s_len ? w * per : s_len, end = beg + per < s_len ? beg + per : s_len;
		uint32_t cnt = 0;
		for (uint32_t base = beg; base < end; base += 256) {
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint32_t i = base + k * 64 + l;
				bool keep = false;
				if (i < end) { const float v = (float)(((double)s_a[i] - mean) / sd); s_b[i] = v; keep = v < 3.0f && v > -3.0f; }
				cnt += (uint32_t)__popcll(__ballot(keep));
			}
		}
		if (l == 0) s_w[w] = cnt;
		__syncthreads();
		uint32_t run = 0;
		n = 0;
		for (uint32_t q = 0; q < NT / 64; ++q) { const uint32_t c2 = s_w[q]; if (q < w) run += c2; n += c2; }
		for (uint32_t base = beg; base < end; base += 256) {
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint32_t i = base + k * 64 + l;
				bool keep = false; float v = 0.0f;
				if (i < end) { v = s_b[i]; keep = v < 3.0f && v > -3.0f; }
				const uint64_t B = __ballot(keep);
				if (keep) { if (FULL) s_z[run + lanes_below(B)] = v; else zrow[run + lanes_below(B)] = v; }
				run += (uint32_t)__popcll(B);
			}
		}
	}
	if (!FULL) { if (tid == 0) rr.n_norm[a] = n; return; }
	__syncthreads();
	if (tid == 0) rr.n_norm[a] = n;
	if (n == 0) return;

	// 3. fp32 prefix sums, strictly left to right (order-sensitive).  The two sums are independent chains: one lane of
	//    wave 0 accumulates z, one lane of wave 1 accumulates z*z, concurrently.  The prefix arrays are stored shifted by
	//    3 floats so that entries 4k+1 .. 4k+4 form one aligned 16-byte LDS store.
	float *pa = s_a + 3, *pb = s_b + 3;
	if (tid == 0) serial_prefix<false>(s_z, pa, n);
	else if (tid == 64) serial_prefix<true>(s_z, pb, n);
	__syncthreads();

	// 4. t-statistics of both windows and the normalised signal go to HBM rows (coalesced)
	float *t1row = rr.t1buf + (size_t)a * rr.ev_row, *t2row = rr.t2buf + (size_t)a * rr.ev_row;
	for (uint32_t i = tid; i < n; i += NT) {
		zrow[i] = s_z[i];
		t1row[i] = tstat_at(pa, pb, n, o.w1, i);
		t2row[i] = tstat_at(pa, pb, n, o.w2, i);
	}
}

// The fp32 prefix sums of z and z*z must be accumulated strictly left to right (comp_prefix_prefixsq, revent.c:23-36), a
// chain of 4000 dependent adds per chunk.  Here each lane of wave 0 owns the chain of one chunk, 64 chunks per block, and
// the chunks advance together in tiles of TS_TILE samples: the block loads the tile of z transposed into LDS (row stride
// 65: lane-per-chunk and lane-per-sample accesses are both conflict free), wave 0 extends the 64 chains, and then all
// four waves turn the new prefix values into t-statistics, lanes across positions, half a wave per window.  A window of
// width w can be evaluated TS_TILE samples at a time lagging w behind the chains, so the prefix sums only ever live in
// a ring of TS_RING entries per chunk and never touch HBM.
#define TS_TILE 32
#define TS_RING 64
#define TS_WMAX 15                              // TS_TILE + 2 * w + 1 <= TS_RING
#define TS_STRIDE 65
// x / fw for the small integer window widths: product with the rounded reciprocal plus one exact-remainder correction
// (Markstein); compared with the correctly rounded quotient over all 2^32 inputs for every w in 2..TS_WMAX it differs
// only for x = -0, which takes the plain product (tests: rh_debug_div_const_check).  3 VALU ops instead of ~11.
RH_DEV float div_by_const(float x, float fw, float r)
{
	const float q0 = x * r;
	const float e = __builtin_fmaf(-q0, fw, x);
	const float q1 = __builtin_fmaf(e, r, q0);
	return x == 0.0f ? q0 : q1;
}

__global__ void k_div_const_check(float fw, unsigned long long *out)
{
	const float r = 1.0f / fw;
	unsigned long long bad = 0, bad_reach = 0;
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t u = (uint32_t)b;
		if ((u & 0x7f800000u) == 0x7f800000u) continue;
		const float x = __uint_as_float(u);
		const float qt = x / fw, qf = div_by_const(x, fw, r);
		if (__float_as_uint(qt) != __float_as_uint(qf)) { ++bad; const float ax = fabsf(x); if (ax == 0.0f || (ax > 1e-30f && ax < 1e30f)) ++bad_reach; }
	}
	if (bad) atomicAdd(&out[0], bad);
	if (bad_reach) atomicAdd(&out[1], bad_reach);
}

// test hook (not part of the interface): out[0] = inputs where div_by_const differs from x / w, out[1] = those among
// zero and 1e-30 < |x| < 1e30 (the prefix sums of |z| < 3 samples live far inside that range)
extern "C" __attribute__((visibility("default"))) int rh_debug_div_const_check(int w, unsigned long long *out)
{
	unsigned long long *d = nullptr;
	if (hipMalloc(&d, 16) != hipSuccess) return -1;
	int rc = -1;
	if (hipMemset(d, 0, 16) == hipSuccess) {
		RH_LAUNCH(k_div_const_check, 4096, 256, 0, 0, (float)w, d);
		if (hipMemcpy(out, d, 16, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
	}
	(void)hipFree(d);
	return rc;
}

// every index is a valid ring slot, so the arithmetic runs unconditionally (independent rows interleave) and the border
// rule of comp_tstat (revent.c:38-74) is a select at the end
RH_DEV float tstat_ring(const float *ps, const float *pss, uint32_t c, uint32_t n, uint32_t w, float fw, float rw, uint32_t i)
{
	#define TS_AT(arr, idx) arr[((idx) & (TS_RING - 1u)) * TS_STRIDE + c]
	const float p0 = TS_AT(ps, i), r0 = TS_AT(pss, i);
	const float pm = TS_AT(ps, i - w), rm = TS_AT(pss, i - w);
	const float s1 = i > w ? p0 - pm : p0, q1 = i > w ? r0 - rm : r0;
	const float s2 = TS_AT(ps, i + w) - p0, q2 = TS_AT(pss, i + w) - r0;
	#undef TS_AT
	const float m1 = div_by_const(s1, fw, rw), m2 = div_by_const(s2, fw, rw);
	float var = div_by_const(div_by_const(q1, fw, rw) - m1 * m1 + div_by_const(q2, fw, rw) - m2 * m2, fw, rw);
	var = fmaxf(var, FLT_MIN);
	const float dm = m2 - m1;
	const float t = fabsf(dm) / sqrtf(var);
	return (n < 2 * w || w < 2 || i < w || i > n - w) ? 0.0f : t;
}

// CB chunks per block (64, or fewer when a round has too few chunks to fill the chip with blocks of 64)
template <uint32_t CB>
__global__ __launch_bounds__(NT) void k_events_tstat(rh_dev_opt o, rh_dev_round rr)
{
	constexpr uint32_t RW = CB / 4;                                     // rows (chunks) per wave
	__shared__ float s_zt[TS_TILE * TS_STRIDE];
	__shared__ float s_pa[TS_RING * TS_STRIDE], s_pb[TS_RING * TS_STRIDE];
	__shared__ uint32_t s_n[64];
	const uint32_t tid = threadIdx.x, w = tid >> 6, l = tid & 63u, a0 = blockIdx.x * CB;
	if (tid < 64) {
		s_n[tid] = tid < CB && a0 + tid < rr.n_act ? rr.n_norm[a0 + tid] : 0u;
		s_pa[tid] = 0.0f; s_pb[tid] = 0.0f;                         // prefix entry 0
	}
	__syncthreads();
	uint32_t nmax = s_n[l];
	for (int d = 32; d > 0; d >>= 1) { const uint32_t t = __shfl_xor(nmax, d); if (t > nmax) nmax = t; }
	const uint32_t n_tiles = (nmax + TS_TILE - 1) / TS_TILE;

	// loads: thread (w, l) fetches sample (l & 31) of the chunks w * RW + 2 * k + (l >> 5): two 128-byte row pieces per request
	const uint32_t ls = l & 31u, lc0 = w * RW + (l >> 5);
	uint32_t ln[RW / 2];
	const float *lrow[RW / 2];
#pragma unroll
	for (uint32_t k = 0; k < RW / 2; ++k) { ln[k] = s_n[lc0 + 2 * k]; lrow[k] = rr.zbuf + (size_t)(a0 + lc0 + 2 * k) * rr.ev_row + ls; }
	float zr[RW / 2];
#pragma unroll
	for (uint32_t k = 0; k < RW / 2; ++k) zr[k] = ls < ln[k] ? lrow[k][0] : 0.0f;

	float acc = 0.0f, acc2 = 0.0f;                                     // wave 0: the two chains of chunk l
	const uint32_t win = l < 32 ? o.w1 : o.w2;                         // t-statistics: lanes 0..31 the short window, 32..63 the long one
	const float fwin = (float)win, rwin = 1.0f / fwin;
	float *const tbuf = (l < 32 ? rr.t1buf : rr.t2buf) + (size_t)(a0 + w * RW) * rr.ev_row;
	uint32_t rn[RW];
#pragma unroll
	for (uint32_t k = 0; k < RW; ++k) rn[k] = s_n[w * RW + k];
	for (uint32_t t = 0; t <= n_tiles; ++t) {                          // the last round only finishes the lagging t-statistics
		const uint32_t base = t * TS_TILE;
		if (t < n_tiles) {
#pragma unroll
			for (uint32_t k = 0; k < RW / 2; ++k) s_zt[ls * TS_STRIDE + lc0 + 2 * k] = zr[k];
		}
		__syncthreads();                                               // tile staged; everyone is done with the ring of the previous round
		if (t + 1 < n_tiles) {
#pragma unroll
			for (uint32_t k = 0; k < RW / 2; ++k) zr[k] = base + TS_TILE + ls < ln[k] ? lrow[k][base + TS_TILE] : 0.0f;
		}
		if (w == 0 && t < n_tiles) {
			float z[TS_TILE];
#pragma unroll
			for (uint32_t s2 = 0; s2 < TS_TILE; ++s2) z[s2] = s_zt[s2 * TS_STRIDE + l];
#pragma unroll
			for (uint32_t s2 = 0; s2 < TS_TILE; ++s2) {                 // past the end of a chunk the chain runs on zeros into entries nobody reads
				acc = acc + z[s2];
				acc2 = acc2 + z[s2] * z[s2];
				const uint32_t slot = ((base + s2 + 1) & (TS_RING - 1u)) * TS_STRIDE + l;
				s_pa[slot] = acc; s_pb[slot] = acc2;
			}
		}
		__syncthreads();                                               // prefix entries <= base + TS_TILE are in the ring
		const uint32_t i0 = base + ls - win;                           // this lane's position in every row (wraps below zero: fails i0 < n)
#pragma unroll
		for (uint32_t k = 0; k < RW; ++k) {
			const float v = tstat_ring(s_pa, s_pb, w * RW + k, rn[k], win, fwin, rwin, i0);
			if (i0 < rn[k]) tbuf[(size_t)k * rr.ev_row + i0] = v;
		}
	}
}

// Two lanes per chunk, one per peak detector (short / long window), 32 chunks per wavefront.  Within a step the reference
// runs the short detector first, and only it acts on the other (it masks the long one while it rides a strong peak);
// so the long detector's lane simply lags one step behind and receives the short one's masking event of that step
// through a DPP quad permute before it starts.  One generic detector body per iteration instead of two in sequence.
// Tiles of 64 steps x 32 chunks of t1 and t2 are transposed through LDS (row stride 33: conflict-free reads) so that
// HBM is read in full 256-byte rows.
#define PK_TILE 64
#define PK_CHUNKS 32
template <typename PT>   // peak positions: 16 bits for a chunk, 32 for a whole read (RH_M_NO_ADAPTIVE)
__global__ __launch_bounds__(64) void k_events_peaks(rh_dev_opt o, rh_dev_round rr)
{
	__shared__ float s_t[2][PK_TILE * (PK_CHUNKS + 1)];
	const uint32_t lane = threadIdx.x, c = lane >> 1, k = lane & 1u, a0 = blockIdx.x * PK_CHUNKS, a = a0 + c;
	const uint32_t n = a < rr.n_act ? rr.n_norm[a] : 0u;
	uint32_t nmax = n;
	for (int d = 32; d > 0; d >>= 1) { const uint32_t t = __shfl_xor(nmax, d); if (t > nmax) nmax = t; }
	const float thr = k == 0 ? o.thr1 : o.thr2, ph = o.peak_height;
	const uint32_t win = k == 0 ? o.w1 : o.w2;
	uint32_t masked_to = 0, np = 0;
	int32_t pos = -1, valid = 0;
	float val = FLT_MAX, held = 0.0f;                               // held: the long detector's sample of the step it is about to process
	uint32_t ev_in = 0;                                             // masking event of that step: 1u << 31 | masked_to
	PT *pk = (PT*)rr.peaks + (size_t)a * rr.ev_cap;
	const uint32_t rows = rr.n_act - a0 < PK_CHUNKS ? rr.n_act - a0 : PK_CHUNKS;
	for (uint32_t i0 = 0; i0 <= nmax; i0 += PK_TILE) {               // (<=: one more iteration for the lagging lanes)
		__syncthreads();
		{	// all the row loads of the tile are issued before the first LDS store waits for one (lane = step inside the tile)
			float r1[PK_CHUNKS], r2[PK_CHUNKS];
#pragma unroll
			for (uint32_t row = 0; row < PK_CHUNKS; ++row) {
				const size_t g = (size_t)(a0 + (row < rows ? row : 0u)) * rr.ev_row + i0 + lane;
				r1[row] = rr.t1buf[g]; r2[row] = rr.t2buf[g];
			}
#pragma unroll
			for (uint32_t row = 0; row < PK_CHUNKS; ++row) { s_t[0][lane * (PK_CHUNKS + 1) + row] = r1[row]; s_t[1][lane * (PK_CHUNKS + 1) + row] = r2[row]; }
		}
		__syncthreads();
		const uint32_t tend = i0 + PK_TILE < nmax + 1 ? i0 + PK_TILE : nmax + 1;
		for (uint32_t t = i0; t < tend; ++t) {
			const float cur_t = s_t[k][(t - i0) * (PK_CHUNKS + 1) + c];   // sample of step t of this lane's statistic
			const uint32_t i = t - k;                                   // the step this lane processes now
			const bool act = k == 0 ? t < n : (t >= 1 && t <= n);
			const float cur = k == 0 ? cur_t : held;
			held = cur_t;
			// one step of the detector (revent.c:91-150), written with selects: a dozen divergent branch regions per step cost
			// more than evaluating both states
			if (act && (ev_in >> 31)) { masked_to = ev_in & 0x7FFFFFFFu; pos = -1; val = FLT_MAX; valid = 0; }
			const bool go = act && !(masked_to >= i);
			const bool searching = pos == -1;
			// no peak open: follow the minimum; a rise of more than peak_height above it opens a peak
			const bool a_rise = !(cur < val) && (cur - val > ph);
			const float a_val = (cur < val || a_rise) ? cur : val;
			const int32_t a_pos = a_rise ? (int32_t)i : -1;
			// peak open: raise it; strong enough -> valid (and, short detector, mask the long one); past the window -> emit
			const bool b_up = cur > val;
			const float b_val = b_up ? cur : val;
			const int32_t b_pos = b_up ? (int32_t)i : pos;
			const bool b_strong = b_val > thr;
			const int32_t b_valid = (valid != 0 || (b_val - cur > ph && b_strong)) ? 1 : 0;
			const bool b_emit = b_valid != 0 && (i - (uint32_t)b_pos) > win / 2;
			const uint32_t ev_out = (go && !searching && k == 0 && b_strong) ? (1u << 31 | ((uint32_t)b_pos + win)) : 0u;
			const int32_t emit = (go && !searching && b_emit) ? b_pos : -1;
			if (go) {
				val = searching ? a_val : (b_emit ? cur : b_val);
				pos = searching ? a_pos : (b_emit ? -1 : b_pos);
				valid = searching ? valid : (b_emit ? 0 : b_valid);
			}
			// partner exchange: the long lane takes the short lane's event (for the step it processes next); both learn
			// whether the other emitted.  Order inside the iteration as in the reference: (step t-1, long) then (step t, short).
			ev_in = (uint32_t)rh_quad_perm_0022((int32_t)ev_out);
			if (k == 0) ev_in = 0;
			const int32_t e_mine = emit >= 0 ? 1 : 0;
			const int32_t e_short = rh_quad_perm_0022(e_mine), e_long = rh_quad_perm_1133(e_mine);
			if (emit >= 0) { const uint32_t at = np + (k == 0 ? (uint32_t)e_long : 0u); if (at < rr.ev_cap) pk[at] = (PT)emit; }
			np += (uint32_t)(e_short + e_long);
		}
	}
	if (a < rr.n_act && k == 0) {
		rr.n_peaks[a] = np < rr.ev_cap ? np : rr.ev_cap;
		if (np > rr.ev_cap) atomicAdd((unsigned long long*)&rr.counters[7], 1ull);   // more peaks than the per-chunk arrays hold: the call fails (no silent divergence)
	}
}

// IQR-fenced mean of one sorted segment (revent.c:158-180): fp32 sum in ascending order
RH_DEV float fenced_mean_lds(const float *seg, uint32_t len)
{
	const float q1 = seg[len / 4], q3 = seg[3 * len / 4], iqr = q3 - q1, lo = q1 - iqr, hi = q3 + iqr;
	float sum = 0.0f; uint32_t cnt = 0;
	for (uint32_t i = 0; i < len; ++i) if (seg[i] >= lo && seg[i] <= hi) { sum += seg[i]; ++cnt; }
	return cnt > 0 ? sum / (float)cnt : 0.0f;
}

// One lane per segment, the segment in REGISTERS: 88 % of the segments have <= 16 samples, 99 % <= 32, but an in-LDS
// insertion sort makes every lane wait for the longest segment of its wavefront (quadratic, one bank-conflicted LDS round
// trip per shift).  A bitonic network on 32 (or 16) registers is data-independent: a few hundred min / max, no memory.
// The sorted values are unique as a sequence, so any correct sort gives the reference's qsort result.
template <int N>
RH_DEV float sort_mean_regs(const float *seg, uint32_t len)
{
	float v[N];
#pragma unroll
	for (int i = 0; i < N; ++i) v[i] = (uint32_t)i < len ? seg[i] : FLT_MAX;
#pragma unroll
	for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
		for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
			for (int i = 0; i < N; ++i) {
				const int l = i ^ j;
				if (l > i) {
					const float lo = fminf(v[i], v[l]), hi = fmaxf(v[i], v[l]);
					if ((i & k) == 0) { v[i] = lo; v[l] = hi; } else { v[i] = hi; v[l] = lo; }
				}
			}
		}
	}
	float q1 = 0.0f, q3 = 0.0f;
#pragma unroll
	for (int i = 0; i < N; ++i) { if ((uint32_t)i == len / 4) q1 = v[i]; if ((uint32_t)i == 3 * len / 4) q3 = v[i]; }
	const float iqr = q3 - q1, lo = q1 - iqr, hi = q3 + iqr;
	float sum = 0.0f; uint32_t cnt = 0;
#pragma unroll
	for (int i = 0; i < N; ++i) if ((uint32_t)i < len && v[i] >= lo && v[i] <= hi) { sum += v[i]; ++cnt; }
	return cnt > 0 ? sum / (float)cnt : 0.0f;
}

#define EM_LONG 128          // segments the lanes hand over to a whole wavefront
__global__ __launch_bounds__(NT) void k_events_means(rh_dev_opt o, rh_dev_round rr)
{
	__shared__ float s_z[RH_CHUNK_MAX];
	__shared__ uint16_t s_peaks[RH_EV_CAP];
	__shared__ uint16_t s_long[EM_LONG];
	__shared__ uint32_t s_nlong;
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t n = rr.n_norm[a];
	const uint32_t np = n ? rr.n_peaks[a] : 0u;
	const float *zrow = rr.zbuf + (size_t)a * rr.ev_row;
	const uint16_t *pk = rr.peaks + (size_t)a * rr.ev_cap;
	for (uint32_t i = tid; i < n; i += NT) s_z[i] = zrow[i];
	for (uint32_t i = tid; i < np; i += NT) s_peaks[i] = pk[i];
	if (tid == 0) s_nlong = 0;
	__syncthreads();
	float *ev = rr.ev + (size_t)a * rr.ev_cap;
	for (uint32_t k0 = 0; k0 < np; k0 += NT) {
		const uint32_t k = k0 + tid;
		uint32_t start = 0, len = 0;
		if (k < np) { start = k ? s_peaks[k - 1] : 0u; const uint32_t end = s_peaks[k]; len = end > start ? end - start : 0u; }
		const bool small = k < np && len <= 32;
		if (k < np && !small) { const uint32_t q = atomicAdd(&s_nlong, 1u); if (q < EM_LONG) s_long[q] = (uint16_t)k; }
		float res = 0.0f;
		if (__ballot(small && len > 16)) { if (small && len > 0) res = sort_mean_regs<32>(s_z + start, len); }
		else if (small && len > 0) res = sort_mean_regs<16>(s_z + start, len);
		if (small) ev[k] = res;
	}
	__syncthreads();
	// the rare long segments: one wavefront each ranks the samples (broadcast LDS reads), scatters them, and one lane sums
	const uint32_t nl = s_nlong < EM_LONG ? s_nlong : EM_LONG;
	for (uint32_t q = wave_id(); q < nl; q += NT / 64) {
		const uint32_t k = s_long[q], start = k ? s_peaks[k - 1] : 0u, len = s_peaks[k] - start, l = lane_id();
		float *seg = s_z + start;
		if (len <= 256) {
			float v[4]; uint32_t rk[4];
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				const uint32_t e = (uint32_t)t * 64u + l;
				v[t] = e < len ? seg[e] : 0.0f; rk[t] = 0;
			}
			for (uint32_t j = 0; j < len; ++j) {
				const float sj = seg[j];
#pragma unroll
				for (int t = 0; t < 4; ++t) rk[t] += (sj < v[t] || (sj == v[t] && j < (uint32_t)t * 64u + l)) ? 1u : 0u;
			}
			RH_WAVE_SYNC();
#pragma unroll
			for (int t = 0; t < 4; ++t) if ((uint32_t)t * 64u + l < len) seg[rk[t]] = v[t];
			RH_WAVE_SYNC();
		} else if (l == 0) {
			for (uint32_t i = 1; i < len; ++i) { const float x = seg[i]; uint32_t j = i; while (j > 0 && seg[j - 1] > x) { seg[j] = seg[j - 1]; --j; } seg[j] = x; }
		}
		if (l == 0) ev[k] = fenced_mean_lds(seg, len);
	}
	if (s_nlong > EM_LONG && tid == 0) {	// more long segments than the list holds (pathological signal): the plain serial way
		for (uint32_t k = 0; k < np; ++k) {
			const uint32_t start = k ? s_peaks[k - 1] : 0u, end = s_peaks[k], len = end > start ? end - start : 0u;
			if (len <= 32) continue;
			bool listed = false;
			for (uint32_t q = 0; q < EM_LONG; ++q) listed |= s_long[q] == k;
			if (listed) continue;
			float *seg = s_z + start;
			for (uint32_t i = 1; i < len; ++i) { const float x = seg[i]; uint32_t j = i; while (j > 0 && seg[j - 1] > x) { seg[j] = seg[j - 1]; --j; } seg[j] = x; }
			ev[k] = fenced_mean_lds(seg, len);
		}
	}
	if (tid == 0) {
		rr.n_ev[a] = np;
		rr.skip[a] = np < o.min_events ? 1 : 0;
		atomicAdd((unsigned long long*)&rr.counters[0], (unsigned long long)np);
	}
}

// ------------------------------------------------------------------------------------------------ whole reads (RH_M_NO_ADAPTIVE)
// The Rawsamble presets map a read in ONE round over its whole signal (rmap.cpp:404-405: l_chunk = qlen, max_chunk = 1), and
// the signal-target index is built from whole reads too (rindex.c:283-287).  A read no longer fits LDS: the rows live in HBM
// (stride rr.ev_row), the order-sensitive prefix sums / t-statistics and the peak detectors are the streaming kernels above
// (they never assumed a length), and these two replace the LDS-resident ends of the chain.

// One block per read: pA filter + fp64 statistics + z-score + compaction, in tiles of NT samples (order preserving).  The pA
// values wait in the read's t1 row (k_events_tstat overwrites it afterwards).
__global__ __launch_bounds__(NT) void k_events_norm_whole(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ uint32_t s_w[NT / 64];
	__shared__ double s_red[2 * (NT / 64)];
	__shared__ double s_stat[2];
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t r = rr.act[a];
	const uint64_t o0 = rd.off[r];
	// the round's stretch of the raw signal, [chunk_start[c], chunk_start[c + 1]): the whole read (RH_M_NO_ADAPTIVE), or - adaptive rounds whose
	// chunk_size is beyond the LDS-resident kernels - the raw samples of chunk rr.chunk
	const uint32_t *cs = rd.chunk_start + (size_t)r * rd.cs_stride;
	// (whole reads: k_prefilter ran with chunk_size = 2^30, so chunk 0 is [first sample that passes the pA filter, end of the read))
	const uint32_t r_lo = cs[rr.chunk];
	const uint32_t n_raw = cs[rr.chunk + 1] - r_lo;
	const int16_t *raw = rd.raw + o0 + r_lo;
	const double coff = rd.cal_off[r];
	const float cscale = rd.cal_scale[r];
	const uint32_t f5 = rd.fast5;
	float *parow = rr.t1buf + (size_t)a * rr.ev_row, *zrow = rr.zbuf + (size_t)a * rr.ev_row;
	uint32_t s_len = 0;
	double dsum = 0.0, dsum2 = 0.0;
	for (uint32_t base = 0; base < n_raw; base += NT) {
		const uint32_t i = base + tid;
		bool valid = false; float pa = 0.0f;
		if (i < n_raw) { pa = raw_to_pa(raw[i], coff, cscale, f5); valid = pa > 30.0f && pa < 200.0f; pa = rh_pa_value(pa, f5); }
		uint32_t tot;
		const uint32_t pos = s_len + block_rank(valid, s_w, tot);
		if (valid) { parow[pos] = pa; dsum += (double)pa; const float sq = pa * pa; dsum2 += (double)sq; }
		s_len += tot;
	}
	for (int d = 32; d > 0; d >>= 1) { dsum += __shfl_down(dsum, d); dsum2 += __shfl_down(dsum2, d); }
	if (lane_id() == 0) { s_red[2 * wave_id()] = dsum; s_red[2 * wave_id() + 1] = dsum2; }
	__syncthreads();
	if (tid == 0) {
		double S = rd.sum[r], S2 = rd.sum2[r];
		for (uint32_t q = 0; q < NT / 64; ++q) { S += s_red[2 * q]; S2 += s_red[2 * q + 1]; }
		const uint32_t N = rd.n_sum[r] + s_len;
		rd.sum[r] = S; rd.sum2[r] = S2; rd.n_sum[r] = N;
		const double mean = S / N;
		s_stat[0] = mean;
		s_stat[1] = sqrt(S2 / N - mean * mean);
		atomicAdd((unsigned long long*)&rr.counters[5], (unsigned long long)s_len);
		atomicAdd((unsigned long long*)&rr.counters[6], 1ull);
	}
	__syncthreads();                                              // (also: the pA row is complete)
	const double mean = s_stat[0], sd = s_stat[1];
	uint32_t n = 0;
	for (uint32_t base = 0; base < s_len; base += NT) {
		const uint32_t i = base + tid;
		bool keep = false; float v = 0.0f;
		if (i < s_len) { v = (float)(((double)parow[i] - mean) / sd); keep = v < 3.0f && v > -3.0f; }
		uint32_t tot;
		const uint32_t pos = n + block_rank(keep, s_w, tot);
		if (keep) zrow[pos] = v;
		n += tot;
	}
	if (tid == 0) rr.n_norm[a] = n;
}

// One block per read, one lane per segment between consecutive peaks (revent.c:193-219): sorted in registers up to 32
// samples, in place in the (no longer needed) z row beyond that.
__global__ __launch_bounds__(NT) void k_events_means_whole(rh_dev_opt o, rh_dev_round rr)
{
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t n = rr.n_norm[a];
	const uint32_t np = n ? rr.n_peaks[a] : 0u;
	float *zrow = rr.zbuf + (size_t)a * rr.ev_row;
	const uint32_t *pk = (const uint32_t*)rr.peaks + (size_t)a * rr.ev_cap;
	float *ev = rr.ev + (size_t)a * rr.ev_cap;
	for (uint32_t k0 = 0; k0 < np; k0 += NT) {
		const uint32_t k = k0 + tid;
		uint32_t start = 0, len = 0;
		if (k < np) { start = k ? pk[k - 1] : 0u; const uint32_t end = pk[k]; len = end > start ? end - start : 0u; }
		const bool small = k < np && len <= 32;
		float res = 0.0f;
		if (__ballot(small && len > 16)) { if (small && len > 0) res = sort_mean_regs<32>(zrow + start, len); }
		else if (small && len > 0) res = sort_mean_regs<16>(zrow + start, len);
		if (k < np && !small) {
			float *seg = zrow + start;
			for (uint32_t i = 1; i < len; ++i) { const float x = seg[i]; uint32_t j = i; while (j > 0 && seg[j - 1] > x) { seg[j] = seg[j - 1]; --j; } seg[j] = x; }
			res = fenced_mean_lds(seg, len);
		}
		if (k < np) ev[k] = res;
	}
	if (tid == 0) {
		rr.n_ev[a] = np;
		rr.skip[a] = np < o.min_events ? 1 : 0;
		atomicAdd((unsigned long long*)&rr.counters[0], (unsigned long long)np);
	}
}

// ------------------------------------------------------------------------------------------------ k_sketch
struct seed_emit {
	uint64_t *sx, *sy; uint32_t n, cap;
	RH_HD void operator()(uint64_t x, uint64_t y) { if (n < cap) { sx[n] = x; sy[n] = y; } ++n; }
};

// per-lane slices of LDS, slot-major: lane l of the wavefront owns column l (conflict-free whatever slot each lane indexes)
template <bool MINIMISERS>
struct sketch_store_lds {
	uint32_t *r; uint64_t *x, *y;
	__device__ uint32_t &ring(int i) { return r[i * 64]; }
	__device__ uint64_t &bx(int i) { return x[MINIMISERS ? i * 64 : 0]; }
	__device__ uint64_t &by(int i) { return y[MINIMISERS ? i * 64 : 0]; }
};

template <bool MINIMISERS>
__global__ __launch_bounds__(64) void k_sketch(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ uint32_t s_ring[16 * 64];
	__shared__ uint64_t s_bx[MINIMISERS ? RH_DEV_MAXW * 64 : 64], s_by[MINIMISERS ? RH_DEV_MAXW * 64 : 64];
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act) return;
	if (rr.skip[a]) { rr.n_seed[a] = 0; return; }
	seed_emit em = { rr.sx + (size_t)a * rr.ev_cap, rr.sy + (size_t)a * rr.ev_cap, 0u, rr.ev_cap };
	sketch_store_lds<MINIMISERS> st = { s_ring + threadIdx.x, s_bx + threadIdx.x, s_by + threadIdx.x };
	rh_sketch_events<RH_DEV_MAXW>(rr.ev + (size_t)a * rr.ev_cap, rr.n_ev[a], 0u, 0, ix.sp, em, st);
	const uint32_t ns = em.n < rr.ev_cap ? em.n : rr.ev_cap;
	rr.n_seed[a] = ns;
	atomicAdd((unsigned long long*)&rr.counters[1], (unsigned long long)ns);
}

// ------------------------------------------------------------------------------------------------ k_probe
// One block per active read; 8 lanes cooperate on one seed: they fetch the 8 slots (one 128-byte line) of the seed's
// home bucket together and vote.  Then one lane applies the order-dependent bookkeeping (tandem flag, mid_occ filter,
// rep_len interval merge, prefix of occurrences).
__global__ __launch_bounds__(NT) void k_probe(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ uint32_t s_n[RH_EV_CAP];
	__shared__ uint64_t s_val[RH_EV_CAP];
	__shared__ uint32_t s_flt[RH_EV_CAP];       // over-frequent seeds: q_pos | q_span << 26
	__shared__ uint32_t s_w[NT / 64];
	__shared__ int32_t s_rep[3];                // rep_st, rep_en, rep_len carried over the tiles of a whole read
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t ns = rr.skip[a] ? 0u : rr.n_seed[a];
	const uint64_t *sx = rr.sx + (size_t)a * rr.ev_cap, *sy = rr.sy + (size_t)a * rr.ev_cap;
	const uint32_t grp = tid >> 3, gl = tid & 7u, gshift = lane_id() & ~7u;
	const uint64_t bmask = (1ull << ix.lg_buckets) - 1ull;
	uint64_t *m_val = rr.m_val + (size_t)a * rr.ev_cap;
	uint32_t *m_n = rr.m_n + (size_t)a * rr.ev_cap, *m_meta = rr.m_meta + (size_t)a * rr.ev_cap, *m_pref = rr.m_pref + (size_t)a * (rr.ev_cap + 1);
	uint32_t nm = 0, pref = 0;
	if (tid == 0) { s_rep[0] = 0; s_rep[1] = 0; s_rep[2] = 0; }
	// tiles of RH_EV_CAP seeds (a chunk has one; a whole read, RH_M_NO_ADAPTIVE, as many as its events need)
	for (uint32_t t0 = 0; t0 == 0 || t0 < ns; t0 += RH_EV_CAP) {
		const uint32_t tn = ns - t0 < RH_EV_CAP ? ns - t0 : RH_EV_CAP;
		__syncthreads();
		for (uint32_t i = grp; i < tn; i += NT / 8) {
			const uint32_t hash = (uint32_t)(sx[t0 + i] >> 6);
			uint64_t b = (uint64_t)((uint32_t)(hash * 0x9E3779B1u) >> (32 - ix.lg_buckets));
			for (uint64_t left = bmask + 1ull;; --left) {               // (an adopted table without a free slot must not hang the probe: bounded by its size)
				const rh_tslot sl = ix.table[b * RH_TB_SLOTS + gl];
				const bool hit = sl.n != 0 && sl.hash == hash, empty = sl.n == 0 || left == 0;
				const uint32_t mh = (uint32_t)(__ballot(hit) >> gshift) & 0xFFu;
				const uint32_t me = (uint32_t)(__ballot(empty) >> gshift) & 0xFFu;
				if (mh) { if (hit) { s_n[i] = sl.n; s_val[i] = sl.val; } break; }
				if (me) { if (gl == 0) s_n[i] = 0; break; }
				b = (b + 1) & bmask;
			}
		}
		__syncthreads();
		// Bookkeeping of ri_collect_matches (rseed.c:105-154), order preserving and parallel: tandem flag from the neighbouring
		// hashes, mid_occ filter, compaction of the kept matches with the running prefix of their occurrence counts; only the
		// interval merge of the (few) over-frequent seeds is left to one lane.
		uint32_t n_flt = 0;
		for (uint32_t base = 0; base < tn; base += NT) {
			const uint32_t il = base + tid, i = t0 + il;
			bool kept = false, flt = false;
			uint32_t cnt = 0, q_pos = 0, tandem = 0;
			if (il < tn) {
				cnt = s_n[il];
				if (cnt != 0) {
					const uint64_t h = sx[i] >> 6;
					q_pos = (uint32_t)sy[i];
					tandem = ((i > 0 && (sx[i - 1] >> 6) == h) || (i + 1 < ns && (sx[i + 1] >> 6) == h)) ? 1u : 0u;
					flt = cnt > (uint32_t)o.mid_occ;
					kept = !flt;
				}
			}
			uint32_t tot_k, tot_c, tot_f;
			const uint32_t rk = block_rank(kept, s_w, tot_k);
			const uint32_t pc = block_excl_scan(kept ? cnt : 0u, s_w, tot_c);
			const uint32_t rf = block_rank(flt, s_w, tot_f);
			if (kept) { m_val[nm + rk] = s_val[il]; m_n[nm + rk] = cnt; m_meta[nm + rk] = (q_pos >> 1) | (tandem << 31); m_pref[nm + rk] = pref + pc; }
			if (flt) s_flt[n_flt + rf] = (q_pos >> 1) | ((uint32_t)(sx[i] & 63u) << 26);
			nm += tot_k; pref += tot_c; n_flt += tot_f;
		}
		__syncthreads();
		if (tid == 0) {
			int32_t rep_st = s_rep[0], rep_en = s_rep[1], rep_len = s_rep[2];
			for (uint32_t k = 0; k < n_flt; ++k) {
				const int32_t st = (int32_t)(s_flt[k] & 0x3FFFFFFu) + 1, en = st + (int32_t)(s_flt[k] >> 26) + 1;
				if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
				else rep_en = en;
			}
			s_rep[0] = rep_st; s_rep[1] = rep_en; s_rep[2] = rep_len;
		}
	}
	__syncthreads();
	if (tid == 0) {
		m_pref[nm] = pref;
		rr.n_match[a] = nm; rr.n_new[a] = pref; rr.rep_len[a] = s_rep[2] + (s_rep[1] - s_rep[0]);
		atomicAdd((unsigned long long*)&rr.counters[2], (unsigned long long)pref);
	}
}

// ------------------------------------------------------------------------------------------------ k_scan_anchors
// a_off = exclusive scan of (new hits + carried anchors) over the active reads; a_off[n_act] = total, [n_act + 1] = the
// largest count.  One block.
__global__ __launch_bounds__(1024) void k_scan_anchors(rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ uint64_t s_part[1024];
	__shared__ uint32_t s_max[1024];
	const uint32_t tid = threadIdx.x, nt = blockDim.x, n = rr.n_act;
	const uint32_t per = (n + nt - 1) / nt;
	const uint32_t b = tid * per, e = b + per < n ? b + per : n;
	uint64_t s = 0;
	uint32_t mx = 0;
	for (uint32_t i = b; i < e; ++i) { const uint32_t c = rr.n_new[i] + rd.n_prev[rr.act[i]]; s += c; mx = c > mx ? c : mx; }
	s_part[tid] = s; s_max[tid] = mx;
	__syncthreads();
	if (tid == 0) {
		uint64_t run = 0;
		uint32_t m = 0;
		for (uint32_t i = 0; i < nt; ++i) { const uint64_t v = s_part[i]; s_part[i] = run; run += v; m = s_max[i] > m ? s_max[i] : m; }
		rr.a_off[n] = run; rr.a_off[n + 1] = m;                     // total, and the largest read (lets the host skip empty size classes)
		atomicAdd((unsigned long long*)&rr.counters[3], (unsigned long long)run);
	}
	__syncthreads();
	uint64_t run = s_part[tid];
	for (uint32_t i = b; i < e; ++i) { rr.a_off[i] = run; run += (uint64_t)rr.n_new[i] + rd.n_prev[rr.act[i]]; }
}

// ------------------------------------------------------------------------------------------------ k_expand
// One block per active read: output anchor j finds its seed by binary search in the occurrence prefix (LDS), gathers the
// 8-byte position word (the random HBM reads of the path) and writes the 16-byte anchor coalesced.
__global__ __launch_bounds__(NT) void k_expand(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_dev_round rr, const uint8_t *skip2)
{
	__shared__ uint32_t s_pref[RH_EV_CAP + 1];
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	if (skip2 && skip2[a]) return;                                 // (a second run for the reads whose any-order sort found equal keys: rhk_sort; never a dropped chunk)
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const uint32_t np = rd.n_prev[r];
	const rh_mm128_t *pin = rr.prev_in + rd.prev_off[r];
	if (rr.skip[a]) {	// chunk dropped after event detection: carried anchors stay untouched (rmap.cpp:232-235)
		for (uint32_t j = tid; j < np; j += NT) rh_an_cp(rr, rr.prev_out, base + j, rr.prev_in, rd.prev_off[r] + j);
		__syncthreads();
		if (tid == 0) rd.prev_off[r] = base;
		return;
	}
	const uint32_t nm = rr.n_match[a], nn = rr.n_new[a];
	const uint64_t *m_val = rr.m_val + (size_t)a * rr.ev_cap;
	const uint32_t *m_n = rr.m_n + (size_t)a * rr.ev_cap, *m_meta = rr.m_meta + (size_t)a * rr.ev_cap, *m_pref = rr.m_pref + (size_t)a * (rr.ev_cap + 1);
	const bool in_lds = nm <= RH_EV_CAP;                            // (a whole read's matches may not fit LDS: searched where they lie - by two loops, not through one pointer that may be either: that one is generic, its loads flat)
	if (in_lds) for (uint32_t i = tid; i <= nm; i += NT) s_pref[i] = m_pref[i];
	__syncthreads();
	const uint32_t q_off = rd.ev_off[r];
	const uint64_t span = (uint64_t)(ix.sp.k + ix.sp.e - 1);
	rh_mm128_t *anc = rr.raw + base;
	// rr.afmt: the anchor as ONE word - key' (strand, target, position) above the tandem flag and the query position; the span is the
	// index's constant and seg_id is 0, so nothing is lost (k_anchor_unpack) and the sorters move 8 bytes per anchor instead of 16
	const bool pk = rr.afmt.rec8 != 0;
	const uint32_t plo = rr.afmt.lo, pmid = rr.afmt.mid, psh = rr.afmt.shift, pqb = rr.aq_bits;
	uint64_t *anc8 = reinterpret_cast<uint64_t*>(rr.raw) + base;
	// (round 6: four anchors a thread and step - search, match record, position: three dependent trips per anchor, and the late rounds' few thousand reads of 10^5
	// anchors each leave a workgroup little else to hide them behind; the four chains are independent)
	constexpr int XU = 4;
	for (uint32_t j0 = tid; j0 < nn; j0 += XU * NT) {
		uint32_t sv[XU], kv[XU];
#pragma unroll
		for (int u = 0; u < XU; ++u) {
			const uint32_t j = j0 + (uint32_t)u * NT;
			uint32_t lo = 0, hi = nm;   // largest s with s_pref[s] <= j
			uint32_t vlo = 0;                                              // = pref[lo] (an exclusive prefix: pref[0] = 0)
			if (j < nn) {
				if (in_lds) while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1, pv = s_pref[mid]; if (pv <= j) { lo = mid; vlo = pv; } else hi = mid; }
				else while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1, pv = m_pref[mid]; if (pv <= j) { lo = mid; vlo = pv; } else hi = mid; }
			}
			sv[u] = lo; kv[u] = j - vlo;
		}
		uint32_t mn[XU], meta_v[XU]; uint64_t mv[XU];
#pragma unroll
		for (int u = 0; u < XU; ++u) { const bool on = j0 + (uint32_t)u * NT < nn; mn[u] = on ? m_n[sv[u]] : 1u; mv[u] = on ? m_val[sv[u]] : 0ull; meta_v[u] = on ? m_meta[sv[u]] : 0u; }
		uint64_t hitv[XU];
#pragma unroll
		for (int u = 0; u < XU; ++u) hitv[u] = (mn[u] == 1 || j0 + (uint32_t)u * NT >= nn) ? mv[u] : ix.pos[mv[u] + kv[u]];
#pragma unroll
		for (int u = 0; u < XU; ++u) {
			const uint32_t j = j0 + (uint32_t)u * NT;
			if (j >= nn) continue;
			const uint64_t hit = hitv[u];
			const uint32_t meta = meta_v[u];
			rh_mm128_t p;
			p.x = (hit & 0x7FFFFFFF80000000ull) | (uint64_t)((uint32_t)(hit >> 1) & 0x7FFFFFFFu);
			if (hit & 1ull) p.x |= 1ull << 63;
			p.y = span << 32 | (uint64_t)(uint32_t)((meta & 0x7FFFFFFFu) + q_off);   // seg_id (y >> 40) is 0 for reads
			if (meta >> 31) p.y |= 1ull << 38;
			if (pk) anc8[j] = rh_rec8_pack_key(p.x, plo, pmid) << psh | (uint64_t)(meta >> 31) << pqb | (uint64_t)(uint32_t)p.y;
			else anc[j] = p;
		}
	}
	if (pk) { const uint64_t *pin8 = reinterpret_cast<const uint64_t*>(rr.prev_in) + rd.prev_off[r]; for (uint32_t j = tid; j < np; j += NT) anc8[nn + j] = pin8[j]; }   // (carried anchors are words already)
	else
	for (uint32_t j = tid; j < np; j += NT) anc[nn + j] = pin[j];
}

// ------------------------------------------------------------------------------------------------ all-vs-all (RH_M_ALL_CHAINS)
// collect_seed_hits drops, hit by hit, the targets whose name is not greater than the read's (rmap.cpp:86: reads overlap
// themselves and every pair would be reported twice).  Names stay on the host; their order arrives as ranks (rd.name_rank of
// the reads, ix.t_rank of the targets: strcmp(qname, tname) >= 0  <=>  name_rank >= t_rank).  The kept hits must stay in
// their order (the anchor sort is not stable), so the counts are taken first and the prefix rebuilt:
//   k_ava_count   kept hits per match -> m_pref (exclusive prefix), n_new
//   k_expand_ava  one wavefront per match: its kept hits, ranked by ballots, written from the match's prefix
RH_DEV bool ava_keep(const rh_dev_index &ix, uint32_t qrank, uint64_t hit) { return !(qrank >= ix.t_rank[(uint32_t)(hit >> 32)]); }

__global__ __launch_bounds__(NT) void k_ava_count(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint32_t a = blockIdx.x, tid = threadIdx.x, w = wave_id(), l = lane_id();
	if (a >= rr.n_act) return;
	const uint32_t nm = rr.n_match[a], qrank = rd.name_rank[rr.act[a]];
	const uint64_t *m_val = rr.m_val + (size_t)a * rr.ev_cap;
	const uint32_t *m_n = rr.m_n + (size_t)a * rr.ev_cap;
	uint32_t *m_pref = rr.m_pref + (size_t)a * (rr.ev_cap + 1);
	for (uint32_t sd = w; sd < nm; sd += NT / 64) {
		const uint32_t n = m_n[sd];
		const uint64_t v = m_val[sd];
		uint32_t c = 0;
		if (n == 1) c = ava_keep(ix, qrank, v) ? 1u : 0u;
		else for (uint32_t k0 = 0; k0 < n; k0 += 64) { const uint32_t k = k0 + l; c += (uint32_t)__popcll(__ballot(k < n && ava_keep(ix, qrank, ix.pos[v + k]))); }
		if (l == 0) m_pref[sd] = c;
	}
	__syncthreads();
	uint32_t run = 0;
	for (uint32_t base = 0; base < nm; base += NT) {
		const uint32_t i = base + tid;
		const uint32_t c = i < nm ? m_pref[i] : 0u;
		uint32_t tot;
		const uint32_t ex = block_excl_scan(c, s_w, tot);
		if (i < nm) m_pref[i] = run + ex;
		run += tot;
	}
	if (tid == 0) { m_pref[nm] = run; rr.n_new[a] = run; }
}

__global__ __launch_bounds__(NT) void k_expand_ava(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_dev_round rr)
{
	const uint32_t a = blockIdx.x, w = wave_id(), l = lane_id();
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint32_t r = rr.act[a], nm = rr.n_match[a], qrank = rd.name_rank[r];
	const uint64_t *m_val = rr.m_val + (size_t)a * rr.ev_cap;
	const uint32_t *m_n = rr.m_n + (size_t)a * rr.ev_cap, *m_meta = rr.m_meta + (size_t)a * rr.ev_cap, *m_pref = rr.m_pref + (size_t)a * (rr.ev_cap + 1);
	const uint32_t q_off = rd.ev_off[r];
	const uint64_t span = (uint64_t)(ix.sp.k + ix.sp.e - 1);
	rh_mm128_t *anc = rr.raw + rr.a_off[a];
	for (uint32_t sd = w; sd < nm; sd += NT / 64) {
		const uint32_t n = m_n[sd], meta = m_meta[sd];
		const uint64_t v = m_val[sd];
		uint32_t at = m_pref[sd];
		for (uint32_t k0 = 0; k0 < n; k0 += 64) {
			const uint32_t k = k0 + l;
			uint64_t hit = 0; bool keep = false;
			if (k < n) { hit = n == 1 ? v : ix.pos[v + k]; keep = ava_keep(ix, qrank, hit); }
			const uint64_t B = __ballot(keep);
			if (keep) {
				rh_mm128_t p;
				p.x = (hit & 0x7FFFFFFF80000000ull) | (uint64_t)((uint32_t)(hit >> 1) & 0x7FFFFFFFu);
				if (hit & 1ull) p.x |= 1ull << 63;
				p.y = span << 32 | (uint64_t)(uint32_t)((meta & 0x7FFFFFFFu) + q_off);
				if (meta >> 31) p.y |= 1ull << 38;
				anc[at + lanes_below(B)] = p;
			}
			at += (uint32_t)__popcll(B);
		}
	}
}

// ------------------------------------------------------------------------------------------------ k_compact_active
// Reads that continue with chunk `next_chunk`, in their current order.  One block.
__global__ __launch_bounds__(1024) void k_compact_active(rh_dev_opt o, rh_dev_reads rd, const uint32_t *act_in, uint32_t n_in, uint32_t next_chunk,
                                                         uint32_t *act_out, uint32_t *n_out)
{
	__shared__ uint32_t s_part[1024];
	const uint32_t tid = threadIdx.x, nt = blockDim.x;
	const uint32_t per = (n_in + nt - 1) / nt;
	const uint32_t b = tid * per, e = b + per < n_in ? b + per : n_in;
	uint32_t c = 0;
	for (uint32_t i = b; i < e; ++i) { const uint32_t r = act_in ? act_in[i] : i; if (!rd.done[r] && next_chunk < read_n_chunks(o, rd.l_sig[r])) ++c; }
	s_part[tid] = c;
	__syncthreads();
	if (tid == 0) { uint32_t run = 0; for (uint32_t i = 0; i < nt; ++i) { const uint32_t v = s_part[i]; s_part[i] = run; run += v; } *n_out = run; }
	__syncthreads();
	uint32_t run = s_part[tid];
	for (uint32_t i = b; i < e; ++i) { const uint32_t r = act_in ? act_in[i] : i; if (!rd.done[r] && next_chunk < read_n_chunks(o, rd.l_sig[r])) act_out[run++] = r; }
}

// ------------------------------------------------------------------------------------------------ slices of a round, carried anchors
// out[i] = a_off[i] - a_off[0] for the n + 1 offsets of a slice of the active list
__global__ void k_rebase_offsets(const uint64_t *a_off, uint32_t n, uint64_t *out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= n) out[i] = a_off[i] - a_off[0];
}

// The chained anchors a read carries into its next chunk (reg->prev_anchors) sit in the round's staging arena at the
// read's anchor offset; they are packed read after read into the dense carry buffer.  One block: offsets of the slice's
// reads in the carry buffer (from `used` on), total -> *total_out.
__global__ __launch_bounds__(1024) void k_carry_scan(rh_dev_reads rd, const uint32_t *act, uint32_t n, uint64_t used, uint64_t *dst_off, uint64_t *total_out)
{
	__shared__ uint64_t s_part[1024];
	const uint32_t tid = threadIdx.x, nt = blockDim.x;
	const uint32_t per = (n + nt - 1) / nt, b = tid * per, e = b + per < n ? b + per : n;
	uint64_t s = 0;
	for (uint32_t i = b; i < e; ++i) s += rd.n_prev[act[i]];
	s_part[tid] = s;
	__syncthreads();
	if (tid == 0) { uint64_t run = 0; for (uint32_t i = 0; i < nt; ++i) { const uint64_t v = s_part[i]; s_part[i] = run; run += v; } *total_out = run; }
	__syncthreads();
	uint64_t run = used + s_part[tid];
	for (uint32_t i = b; i < e; ++i) { dst_off[i] = run; run += rd.n_prev[act[i]]; }
}
__global__ __launch_bounds__(NT) void k_carry_copy(rh_dev_reads rd, const uint32_t *act, uint32_t n, const rh_mm128_t *staging, const uint64_t *dst_off, rh_mm128_t *carry, int words8)
{
	const uint32_t a = blockIdx.x;
	if (a >= n) return;
	const uint32_t r = act[a], np = rd.n_prev[r];
	if (words8) {	// one-word anchors (rh_dev_round::afmt)
		const uint64_t *src = reinterpret_cast<const uint64_t*>(staging) + rd.prev_off[r];
		uint64_t *dst = reinterpret_cast<uint64_t*>(carry) + dst_off[a];
		for (uint32_t j = threadIdx.x; j < np; j += NT) dst[j] = src[j];
	} else {
	const rh_mm128_t *src = staging + rd.prev_off[r];
	rh_mm128_t *dst = carry + dst_off[a];
	for (uint32_t j = threadIdx.x; j < np; j += NT) dst[j] = src[j];
	}
	__syncthreads();
	if (threadIdx.x == 0) rd.prev_off[r] = dst_off[a];
}

// ------------------------------------------------------------------------------------------------ seeds of whole reads -> index
// Signal-target index (rindex.c:283-305): the sketch of every read, in read order, as (32-bit hash, id << 32 | pos << 1).
__global__ __launch_bounds__(1024) void k_seed_scan(rh_dev_round rr, uint64_t *off)
{
	__shared__ uint64_t s_part[1024];
	const uint32_t tid = threadIdx.x, nt = blockDim.x, n = rr.n_act;
	const uint32_t per = (n + nt - 1) / nt, b = tid * per, e = b + per < n ? b + per : n;
	uint64_t s = 0;
	for (uint32_t i = b; i < e; ++i) s += rr.n_seed[i];
	s_part[tid] = s;
	__syncthreads();
	if (tid == 0) { uint64_t run = 0; for (uint32_t i = 0; i < nt; ++i) { const uint64_t v = s_part[i]; s_part[i] = run; run += v; } off[n] = run; }
	__syncthreads();
	uint64_t run = s_part[tid];
	for (uint32_t i = b; i < e; ++i) { off[i] = run; run += rr.n_seed[i]; }
}
__global__ __launch_bounds__(NT) void k_seed_pack(rh_dev_round rr, const uint64_t *off, uint32_t id0, uint32_t *hash_out, uint64_t *pos_out)
{
	const uint32_t a = blockIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t ns = rr.n_seed[a], id = id0 + rr.act[a];
	const uint64_t *sx = rr.sx + (size_t)a * rr.ev_cap, *sy = rr.sy + (size_t)a * rr.ev_cap;
	for (uint32_t i = threadIdx.x; i < ns; i += NT) { hash_out[off[a] + i] = (uint32_t)(sx[i] >> 6); pos_out[off[a] + i] = (uint64_t)id << 32 | (uint32_t)sy[i]; }
}

// ------------------------------------------------------------------------------------------------ k_finalize
// rmap.cpp:507-586: the record of a read from the summary of its last round
RH_DEV rh_map_record_t finalize_one(const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, uint32_t r, float *scale_out = nullptr)
{
	const uint32_t qlen = rd.l_sig[r];
	const uint32_t l_chunk = o.chunk_size > qlen ? qlen : o.chunk_size;
	const uint32_t iters = read_n_chunks(o, qlen);
	uint32_t c_count;
	int mapped = rd.done[r] != 0;
	if (mapped) c_count = rd.stop_chunk[r];
	else { c_count = iters; if (c_count > 0) --c_count; }
	const uint32_t offset = rd.ev_off[r];
	const float scale = (offset == 0) ? 0.0f : (o.sample_per_base == 0) ? 0.0f : ((float)(c_count + 1) * (float)l_chunk / (float)offset) / o.sample_per_base;
	if (scale_out) *scale_out = scale;
	const int32_t n_cregs = rd.ls_ncregs[r];
	if (!mapped && n_cregs > 0 && rd.ls_mapq[r] > o.min_mapq) mapped = 1;   // last-chance rule, rmap.cpp:515
	rh_map_record_t q;
	q.read_idx = r; q._pad = 0;
	q.tag_ci = (int32_t)c_count + 1; q.tag_sl = (int32_t)qlen;
	if (!mapped) {
		q.read_length = o.sig_target ? offset : (uint32_t)(scale * (float)offset);
		q.ref_id = 0; q.read_start_position = 0; q.read_end_position = 0; q.fragment_start_position = 0; q.fragment_length = 0;
		q.mapq = 0; q.rev = 0; q.mapped = 0;
		if (n_cregs >= 1) { q.tag_cm = rd.ls_cnt[r]; q.tag_nc = n_cregs; q.tag_s1 = rd.ls_score[r]; }
		else { q.tag_cm = 0; q.tag_nc = 0; q.tag_s1 = 0; }
	} else {
		const int32_t qs = rd.ls_qs[r], qe = rd.ls_qe[r], rs = rd.ls_rs[r], re = rd.ls_re[r], rid = rd.ls_rid[r], rev = rd.ls_rev[r];
		q.tag_cm = rd.ls_cnt[r]; q.tag_nc = n_cregs; q.tag_s1 = rd.ls_score[r];
		q.read_length = o.sig_target ? offset : (uint32_t)(scale * (float)qe);
		q.ref_id = (uint32_t)rid;
		q.read_start_position = o.sig_target ? (uint32_t)qs : (uint32_t)(scale * (float)qs);
		q.read_end_position = o.sig_target ? (uint32_t)qe : (uint32_t)(scale * (float)qe);
		const uint32_t tlen = (uint32_t)rid < ix.n_seq ? ix.seq_len[rid] : 0u;
		q.fragment_start_position = rev ? (uint32_t)(tlen + 1u - (uint32_t)re) : (uint32_t)rs;
		q.fragment_length = (uint32_t)(re - rs + 1);
		q.mapq = (uint8_t)rd.ls_mapq[r]; q.rev = rev == 1; q.mapped = 1;
	}
	return q;
}

// One thread per read.
__global__ void k_finalize(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_map_record_t *rec)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= rd.n_reads) return;
	rec[r] = finalize_one(o, ix, rd, r);
}

// All-vs-all: a read has as many records as chains it reports (regions_commit_ava left them, two words each, in the dense
// carry buffer), or one.  rec_off = exclusive scan of the record counts (one block), then one thread per read writes them.
__global__ __launch_bounds__(1024) void k_ava_rec_scan(rh_dev_reads rd, uint64_t *rec_off)
{
	__shared__ uint64_t s_part[1024];
	const uint32_t tid = threadIdx.x, nt = blockDim.x, n = rd.n_reads;
	const uint32_t per = (n + nt - 1) / nt, b = tid * per, e = b + per < n ? b + per : n;
	uint64_t s = 0;
	for (uint32_t i = b; i < e; ++i) { const uint32_t m = rd.n_prev[i] / 2; s += m ? m : 1u; }
	s_part[tid] = s;
	__syncthreads();
	if (tid == 0) { uint64_t run = 0; for (uint32_t i = 0; i < nt; ++i) { const uint64_t v = s_part[i]; s_part[i] = run; run += v; } rec_off[n] = run; }
	__syncthreads();
	uint64_t run = s_part[tid];
	for (uint32_t i = b; i < e; ++i) { rec_off[i] = run; const uint32_t m = rd.n_prev[i] / 2; run += m ? m : 1u; }
}

__global__ void k_finalize_ava(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, const rh_mm128_t *maps, const uint64_t *rec_off, rh_map_record_t *rec)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= rd.n_reads) return;
	const uint32_t nm = rd.n_prev[r] / 2;
	float scale;
	rh_map_record_t q = finalize_one(o, ix, rd, r, &scale);       // ci / sl / nc are the same for all
	if (nm == 0) { rec[rec_off[r]] = q; return; }
	const rh_mm128_t *mp = maps + rd.prev_off[r];
	const uint32_t offset = rd.ev_off[r];
	for (uint32_t m = 0; m < nm; ++m) {
		const rh_mm128_t w0 = mp[2 * m], w1 = mp[2 * m + 1];
		const int32_t rid = (int32_t)(uint32_t)w0.x, rs = (int32_t)(w0.x >> 32), re = (int32_t)(uint32_t)w0.y, qs = (int32_t)(w0.y >> 32);
		const int32_t qe = (int32_t)(uint32_t)w1.x, score = (int32_t)(w1.x >> 32), cnt = (int32_t)(uint32_t)w1.y;
		const uint32_t mapq = (uint32_t)(w1.y >> 32) & 0xFFu, rev = (uint32_t)(w1.y >> 40) & 1u;
		q.tag_cm = cnt; q.tag_s1 = score;
		q.read_length = o.sig_target ? offset : (uint32_t)(scale * (float)qe);   // (signal targets: event coordinates, rmap.cpp:574-578)
		q.ref_id = (uint32_t)rid;
		q.read_start_position = o.sig_target ? (uint32_t)qs : (uint32_t)(scale * (float)qs);
		q.read_end_position = o.sig_target ? (uint32_t)qe : (uint32_t)(scale * (float)qe);
		const uint32_t tlen = (uint32_t)rid < ix.n_seq ? ix.seq_len[rid] : 0u;
		q.fragment_start_position = rev ? (uint32_t)(tlen + 1u - (uint32_t)re) : (uint32_t)rs;
		q.fragment_length = (uint32_t)(re - rs + 1);
		q.mapq = (uint8_t)mapq; q.rev = (uint8_t)rev; q.mapped = 1;
		rec[rec_off[r] + m] = q;
	}
}

// ------------------------------------------------------------------------------------------------ k_synth_reads
// Bench/test support: the synthetic read generator of rh_synth_core.h, one read per lane, writing straight into HBM.
__global__ void k_synth_reads(rh_synth_cfg_t c, const int32_t *level16, uint32_t k, uint64_t first, uint32_t n, int16_t *samples, uint64_t *off, double *cal_off, float *cal_scale)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0) off[n] = (uint64_t)n * c.n_samples;
	if (i >= n) return;
	off[i] = (uint64_t)i * c.n_samples;
	cal_off[i] = c.offset;
	cal_scale[i] = (float)(c.range / c.digitisation);
	rh_sy_generate(c, level16, k, first + i, samples + (size_t)i * c.n_samples);
}

// ------------------------------------------------------------------------------------------------ launchers
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

void rhk_prefilter(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const uint32_t *act, uint32_t n, int init, unsigned long long *bad)
{
	const uint32_t g = act ? n : rd.n_reads;
	if (g) RH_LAUNCH(k_prefilter, g, NT, 0, s, o, rd, act, init, bad);
}
void rhk_need(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const uint32_t *act, uint32_t n, uint32_t chunk, uint32_t grow, uint32_t *new_len, uint32_t *n_need)
{
	if (n) RH_LAUNCH(k_need, (n + 255) / 256, 256, 0, s, o, rd, act, n, chunk, grow, new_len, n_need);
}
void rhk_fetch(hipStream_t s, const rh_dev_reads &rd, const int16_t *host_samples, const uint32_t *act, uint32_t n, const uint32_t *new_len, uint32_t max_span)
{
	if (!n) return;
	uint32_t gy = (max_span / 8u + FETCH_VEC - 1u) / FETCH_VEC;
	if (gy == 0) gy = 1;
	RH_LAUNCH(k_fetch, n * gy, 256, 0, s, rd, host_samples, act, n, new_len, gy);
	RH_LAUNCH(k_fetch_commit, (n + 255) / 256, 256, 0, s, rd, act, n, new_len);
}
void rhk_events_norm(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r)
{
	if (!r.n_act) return;
	if (r.whole) RH_LAUNCH(k_events_norm_whole, r.n_act, NT, 0, s, o, rd, r);
	else if (o.w1 > TS_WMAX || o.w2 > TS_WMAX) { RH_LAUNCH(k_events_norm<true>, r.n_act, NT, 0, s, o, rd, r); return; }
	else RH_LAUNCH(k_events_norm<false>, r.n_act, NT, 0, s, o, rd, r);
	const char *force = getenv("RH_TSTAT_CB");                      // tests: pin the chunks-per-block variant
	const uint32_t cb = force ? (uint32_t)atoi(force) : r.n_act >= 64u * 768u ? 64u : r.n_act >= 16u * 768u ? 16u : 8u;
	if (cb >= 64) RH_LAUNCH(k_events_tstat<64>, cdiv(r.n_act, 64), NT, 0, s, o, r);
	else if (cb >= 16) RH_LAUNCH(k_events_tstat<16>, cdiv(r.n_act, 16), NT, 0, s, o, r);
	else RH_LAUNCH(k_events_tstat<8>, cdiv(r.n_act, 8), NT, 0, s, o, r);
}
void rhk_events_peaks(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r) { if (r.n_act) { if (r.whole) RH_LAUNCH(k_events_peaks<uint32_t>, cdiv(r.n_act, PK_CHUNKS), 64, 0, s, o, r); else RH_LAUNCH(k_events_peaks<uint16_t>, cdiv(r.n_act, PK_CHUNKS), 64, 0, s, o, r); } }
void rhk_events_means(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r) { if (!r.n_act) return; if (r.whole) RH_LAUNCH(k_events_means_whole, r.n_act, NT, 0, s, o, r); else RH_LAUNCH(k_events_means, r.n_act, NT, 0, s, o, r); }
void rhk_sketch(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r) {
	if (!r.n_act) return;
	if (ix.sp.w > 0) RH_LAUNCH(k_sketch<true>, cdiv(r.n_act, 64), 64, 0, s, o, ix, rd, r);
	else RH_LAUNCH(k_sketch<false>, cdiv(r.n_act, 64), 64, 0, s, o, ix, rd, r);
}
void rhk_probe(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r)
{
	if (!r.n_act) return;
	RH_LAUNCH(k_probe, r.n_act, NT, 0, s, o, ix, rd, r);
	if (o.flag & RH_M_ALL_CHAINS) RH_LAUNCH(k_ava_count, r.n_act, NT, 0, s, o, ix, rd, r);
}
void rhk_scan_anchors(hipStream_t s, const rh_dev_reads &rd, const rh_dev_round &r) { RH_LAUNCH(k_scan_anchors, 1, 1024, 0, s, rd, r); }
void rhk_expand(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r, const uint8_t *skip2) { if (!r.n_act) return; if (o.flag & RH_M_ALL_CHAINS) RH_LAUNCH(k_expand_ava, r.n_act, NT, 0, s, o, ix, rd, r); else RH_LAUNCH(k_expand, r.n_act, NT, 0, s, o, ix, rd, r, skip2); }
void rhk_compact_active(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const uint32_t *act_in, uint32_t n_in, uint32_t next_chunk, uint32_t *act_out, uint32_t *n_out)
{ RH_LAUNCH(k_compact_active, 1, 1024, 0, s, o, rd, act_in, n_in, next_chunk, act_out, n_out); }
void rhk_rebase_offsets(hipStream_t s, const uint64_t *a_off, uint32_t n, uint64_t *out) { RH_LAUNCH(k_rebase_offsets, cdiv(n + 1, 256), 256, 0, s, a_off, n, out); }
void rhk_carry_scan(hipStream_t s, const rh_dev_reads &rd, const uint32_t *act, uint32_t n, uint64_t used, uint64_t *dst_off, uint64_t *total_out) { RH_LAUNCH(k_carry_scan, 1, 1024, 0, s, rd, act, n, used, dst_off, total_out); }
void rhk_carry_copy(hipStream_t s, const rh_dev_reads &rd, const uint32_t *act, uint32_t n, const rh_mm128_t *staging, const uint64_t *dst_off, rh_mm128_t *carry, int words8) { if (n) RH_LAUNCH(k_carry_copy, n, NT, 0, s, rd, act, n, staging, dst_off, carry, words8); }
void rhk_finalize(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, rh_map_record_t *rec) { if (rd.n_reads) RH_LAUNCH(k_finalize, cdiv(rd.n_reads, 256), 256, 0, s, o, ix, rd, rec); }
void rhk_seed_scan(hipStream_t s, const rh_dev_round &r, uint64_t *off) { RH_LAUNCH(k_seed_scan, 1, 1024, 0, s, r, off); }
void rhk_seed_pack(hipStream_t s, const rh_dev_round &r, const uint64_t *off, uint32_t id0, uint32_t *hash_out, uint64_t *pos_out) { if (r.n_act) RH_LAUNCH(k_seed_pack, r.n_act, NT, 0, s, r, off, id0, hash_out, pos_out); }
void rhk_ava_rec_scan(hipStream_t s, const rh_dev_reads &rd, uint64_t *rec_off) { RH_LAUNCH(k_ava_rec_scan, 1, 1024, 0, s, rd, rec_off); }
void rhk_finalize_ava(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_mm128_t *maps, const uint64_t *rec_off, rh_map_record_t *rec)
{ if (rd.n_reads) RH_LAUNCH(k_finalize_ava, cdiv(rd.n_reads, 256), 256, 0, s, o, ix, rd, maps, rec_off, rec); }
void rhk_synth_reads(hipStream_t s, const rh_synth_cfg_t &c, const int32_t *level16, uint32_t k, uint64_t first, uint32_t n, int16_t *samples, uint64_t *off, double *cal_off, float *cal_scale)
{ if (n) RH_LAUNCH(k_synth_reads, cdiv(n, 64), 64, 0, s, c, level16, k, first, n, samples, off, cal_off, cal_scale); }
