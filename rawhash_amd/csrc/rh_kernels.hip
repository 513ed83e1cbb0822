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

#define NT 256   // threads of the block-cooperative kernels

// ------------------------------------------------------------------------------------------------ wave / block helpers
RH_DEV uint32_t lane_id() { return threadIdx.x & 63u; }
RH_DEV uint32_t wave_id() { return threadIdx.x >> 6; }
RH_DEV uint32_t lanes_below(uint64_t m) { return (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull)); }

// Order-preserving rank of the calling thread among the threads of the block with pred set; total = their number.
// s_w: LDS scratch of (blockDim.x / 64) words.  Contains two block barriers.
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

RH_DEV float raw_to_pa(int16_t raw, double cal_off, float cal_scale)
{
	return (float)(((double)raw + cal_off) * (double)cal_scale);   // (raw + offset) * scale evaluated in double, rsig.c:497
}

// ------------------------------------------------------------------------------------------------ k_prefilter
// One block per read: count samples surviving the pA filter and record, for every chunk boundary, the raw index of the
// first surviving sample of that chunk.  Signal bytes are read once, coalesced (2 B/sample).
__global__ __launch_bounds__(NT) void k_prefilter(rh_dev_opt o, rh_dev_reads rd)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint32_t r = blockIdx.x, tid = threadIdx.x;
	const uint64_t o0 = rd.off[r], n64 = rd.off[r + 1] - o0;
	const uint32_t n = (uint32_t)n64;
	const int16_t *raw = rd.raw + o0;
	const double coff = rd.cal_off[r];
	const float cscale = rd.cal_scale[r];
	uint32_t *cs = rd.chunk_start + (size_t)r * (RH_MAX_CHUNKS + 1);
	for (uint32_t k = tid; k <= RH_MAX_CHUNKS; k += NT) cs[k] = n;
	__syncthreads();
	const uint32_t C = o.chunk_size;
	uint32_t count = 0;
	for (uint32_t base = 0; base < n; base += NT) {
		const uint32_t i = base + tid;
		bool valid = false;
		if (i < n) { const float pa = raw_to_pa(raw[i], coff, cscale); valid = pa > 30.0f && pa < 200.0f; }
		uint32_t total;
		const uint32_t rank = block_rank(valid, s_w, total);
		if (valid) {
			const uint32_t fi = count + rank;
			if (fi % C == 0) { const uint32_t k = fi / C; if (k <= RH_MAX_CHUNKS) cs[k] = i; }
		}
		count += total;
	}
	if (tid == 0) {
		rd.l_sig[r] = count;
		rd.sum[r] = 0.0; rd.sum2[r] = 0.0; rd.n_sum[r] = 0; rd.ev_off[r] = 0; rd.n_prev[r] = 0; rd.prev_off[r] = 0;
		rd.done[r] = 0; rd.stop_chunk[r] = 0; rd.ls_ncregs[r] = 0;
	}
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

struct peak_det { float thr; uint32_t win, masked_to; int32_t pos; float val; int32_t valid; };

// One block per active read: the whole chunk lives in LDS.
__global__ __launch_bounds__(NT) void k_events(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ float s_z[RH_CHUNK_MAX];          // normalised samples; segments are sorted in place at the end
	__shared__ float s_a[RH_CHUNK_MAX + 1];      // pA staging -> prefix sums -> short-window t-stat
	__shared__ float s_b[RH_CHUNK_MAX + 1];      // prefix sums of squares -> long-window t-stat
	__shared__ uint16_t s_peaks[RH_EV_CAP];
	__shared__ uint32_t s_w[NT / 64];
	__shared__ double s_red[2 * (NT / 64)];
	__shared__ double s_stat[2];
	__shared__ uint32_t s_np;

	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t r = rr.act[a], c = rr.chunk;
	const uint64_t o0 = rd.off[r];
	const int16_t *raw = rd.raw + o0;
	const double coff = rd.cal_off[r];
	const float cscale = rd.cal_scale[r];
	const uint32_t *cs = rd.chunk_start + (size_t)r * (RH_MAX_CHUNKS + 1);
	const uint32_t cs0 = cs[c], cs1 = cs[c + 1];
	const uint32_t C = o.chunk_size;

	// 1. load, convert, filter, compact (order preserving) + fp64 partial sums (exact for 30<pA<200, any order)
	uint32_t count = 0;
	double dsum = 0.0, dsum2 = 0.0;
	for (uint32_t base = cs0; base < cs1; base += NT) {
		const uint32_t i = base + tid;
		bool valid = false; float pa = 0.0f;
		if (i < cs1) { pa = raw_to_pa(raw[i], coff, cscale); valid = pa > 30.0f && pa < 200.0f; }
		uint32_t total;
		const uint32_t rank = block_rank(valid, s_w, total);
		if (valid && count + rank < C) {
			s_a[count + rank] = pa;
			dsum += (double)pa;
			const float sq = pa * pa;
			dsum2 += (double)sq;
		}
		count += total;
	}
	const uint32_t s_len = count < C ? count : C;
	for (int d = 32; d > 0; d >>= 1) { dsum += __shfl_down(dsum, d); dsum2 += __shfl_down(dsum2, d); }
	if (lane_id() == 0) { s_red[2 * wave_id()] = dsum; s_red[2 * wave_id() + 1] = dsum2; }
	__syncthreads();
	if (tid == 0) {
		double S = rd.sum[r], S2 = rd.sum2[r];
		for (uint32_t w = 0; w < NT / 64; ++w) { S += s_red[2 * w]; S2 += s_red[2 * w + 1]; }
		const uint32_t N = rd.n_sum[r] + s_len;
		rd.sum[r] = S; rd.sum2[r] = S2; rd.n_sum[r] = N;
		const double mean = S / N;
		s_stat[0] = mean;
		s_stat[1] = sqrt(S2 / N - mean * mean);
	}
	__syncthreads();
	const double mean = s_stat[0], sd = s_stat[1];

	// 2. z-score, drop |z| >= 3, compact
	uint32_t n = 0;
	for (uint32_t base = 0; base < s_len; base += NT) {
		const uint32_t i = base + tid;
		bool keep = false; float v = 0.0f;
		if (i < s_len) { v = (float)(((double)s_a[i] - mean) / sd); keep = v < 3.0f && v > -3.0f; }
		uint32_t total;
		const uint32_t rank = block_rank(keep, s_w, total);
		if (keep) s_z[n + rank] = v;
		n += total;
	}
	__syncthreads();
	if (n == 0) {
		if (tid == 0) { rr.n_ev[a] = 0; rr.skip[a] = 1; atomicAdd((unsigned long long*)&rr.counters[5], (unsigned long long)s_len); atomicAdd((unsigned long long*)&rr.counters[6], 1ull); }
		return;
	}

	// 3. fp32 prefix sums, strictly left to right (order-sensitive: one lane)
	if (tid == 0) {
		float ps = 0.0f, pss = 0.0f;
		s_a[0] = 0.0f; s_b[0] = 0.0f;
		uint32_t i = 0;
		for (; i + 8 <= n; i += 8) {
			float z[8];
			#pragma unroll
			for (int k = 0; k < 8; ++k) z[k] = s_z[i + k];
			#pragma unroll
			for (int k = 0; k < 8; ++k) { ps = ps + z[k]; pss = pss + z[k] * z[k]; s_a[i + k + 1] = ps; s_b[i + k + 1] = pss; }
		}
		for (; i < n; ++i) { const float z = s_z[i]; ps = ps + z; pss = pss + z * z; s_a[i + 1] = ps; s_b[i + 1] = pss; }
	}
	__syncthreads();

	// 4. t-statistics for both windows (all lanes), written back over the prefix sums
	{
		float t1[(RH_CHUNK_MAX + NT) / NT], t2[(RH_CHUNK_MAX + NT) / NT];
		#pragma unroll
		for (int k = 0; k < (RH_CHUNK_MAX + NT) / NT; ++k) {
			const uint32_t i = tid + k * NT;
			t1[k] = i <= n ? tstat_at(s_a, s_b, n, o.w1, i) : 0.0f;
			t2[k] = i <= n ? tstat_at(s_a, s_b, n, o.w2, i) : 0.0f;
		}
		__syncthreads();
		#pragma unroll
		for (int k = 0; k < (RH_CHUNK_MAX + NT) / NT; ++k) {
			const uint32_t i = tid + k * NT;
			if (i <= n) { s_a[i] = t1[k]; s_b[i] = t2[k]; }
		}
	}
	__syncthreads();

	// 5. two coupled peak detectors (short masks long): serial finite-state machine
	if (tid == 0) {
		peak_det d0 = { o.thr1, o.w1, 0u, -1, FLT_MAX, 0 }, d1 = { o.thr2, o.w2, 0u, -1, FLT_MAX, 0 };
		const float ph = o.peak_height;
		uint32_t np = 0;
		for (uint32_t i = 0; i < n; ++i) {
			#pragma unroll
			for (int k = 0; k < 2; ++k) {
				peak_det &q = k == 0 ? d0 : d1;
				if (q.masked_to >= i) continue;
				const float cur = k == 0 ? s_a[i] : s_b[i];
				if (q.pos == -1) {
					if (cur < q.val) q.val = cur;
					else if (cur - q.val > ph) { q.val = cur; q.pos = (int32_t)i; }
				} else {
					if (cur > q.val) { q.val = cur; q.pos = (int32_t)i; }
					if (k == 0 && q.val > q.thr) { d1.masked_to = (uint32_t)q.pos + d0.win; d1.pos = -1; d1.val = FLT_MAX; d1.valid = 0; }
					if (q.val - cur > ph && q.val > q.thr) q.valid = 1;
					if (q.valid && (i - (uint32_t)q.pos) > q.win / 2) {
						if (np < RH_EV_CAP) s_peaks[np] = (uint16_t)q.pos;
						++np;
						q.pos = -1; q.val = cur; q.valid = 0;
					}
				}
			}
		}
		s_np = np < RH_EV_CAP ? np : RH_EV_CAP;
	}
	__syncthreads();
	const uint32_t np = s_np;

	// 6. one lane per segment: sort, IQR fence, mean
	float *ev = rr.ev + (size_t)a * RH_EV_CAP;
	for (uint32_t k = tid; k < np; k += NT) {
		const uint32_t start = k ? s_peaks[k - 1] : 0u, end = s_peaks[k];
		const uint32_t len = end > start ? end - start : 0u;
		float *seg = s_z + start;
		for (uint32_t i = 1; i < len; ++i) {
			const float v = seg[i];
			uint32_t j = i;
			while (j > 0 && seg[j - 1] > v) { seg[j] = seg[j - 1]; --j; }
			seg[j] = v;
		}
		float res = 0.0f;
		if (len > 0) {
			const float q1 = seg[len / 4], q3 = seg[3 * len / 4], iqr = q3 - q1, lo = q1 - iqr, hi = q3 + iqr;
			float sum = 0.0f; uint32_t cnt = 0;
			for (uint32_t i = 0; i < len; ++i) if (seg[i] >= lo && seg[i] <= hi) { sum += seg[i]; ++cnt; }
			res = cnt > 0 ? sum / (float)cnt : 0.0f;
		}
		ev[k] = res;
	}
	if (tid == 0) {
		rr.n_ev[a] = np;
		rr.skip[a] = np < o.min_events ? 1 : 0;
		atomicAdd((unsigned long long*)&rr.counters[0], (unsigned long long)np);
		atomicAdd((unsigned long long*)&rr.counters[5], (unsigned long long)s_len);
		atomicAdd((unsigned long long*)&rr.counters[6], 1ull);
	}
}

// ------------------------------------------------------------------------------------------------ k_sketch
struct seed_emit {
	uint64_t *sx, *sy; uint32_t n, cap;
	RH_HD void operator()(uint64_t x, uint64_t y) { if (n < cap) { sx[n] = x; sy[n] = y; } ++n; }
};

__global__ void k_sketch(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_dev_round rr)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act) return;
	if (rr.skip[a]) { rr.n_seed[a] = 0; return; }
	seed_emit em = { rr.sx + (size_t)a * RH_EV_CAP, rr.sy + (size_t)a * RH_EV_CAP, 0u, RH_EV_CAP };
	rh_sketch_events<RH_DEV_MAXW>(rr.ev + (size_t)a * RH_EV_CAP, rr.n_ev[a], 0u, 0, ix.sp, em);
	const uint32_t ns = em.n < RH_EV_CAP ? em.n : RH_EV_CAP;
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
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t ns = rr.skip[a] ? 0u : rr.n_seed[a];
	const uint64_t *sx = rr.sx + (size_t)a * RH_EV_CAP, *sy = rr.sy + (size_t)a * RH_EV_CAP;
	const uint32_t grp = tid >> 3, gl = tid & 7u, gshift = lane_id() & ~7u;
	const uint64_t bmask = (1ull << ix.lg_buckets) - 1ull;
	for (uint32_t i = grp; i < ns; i += NT / 8) {
		const uint32_t hash = (uint32_t)(sx[i] >> 6);
		uint64_t b = (uint64_t)((uint32_t)(hash * 0x9E3779B1u) >> (32 - ix.lg_buckets));
		for (;;) {
			const rh_tslot sl = ix.table[b * RH_TB_SLOTS + gl];
			const bool hit = sl.n != 0 && sl.hash == hash, empty = sl.n == 0;
			const uint32_t mh = (uint32_t)(__ballot(hit) >> gshift) & 0xFFu;
			const uint32_t me = (uint32_t)(__ballot(empty) >> gshift) & 0xFFu;
			if (mh) { if (hit) { s_n[i] = sl.n; s_val[i] = sl.val; } break; }
			if (me) { if (gl == 0) s_n[i] = 0; break; }
			b = (b + 1) & bmask;
		}
	}
	__syncthreads();
	if (tid == 0) {
		uint64_t *m_val = rr.m_val + (size_t)a * RH_EV_CAP;
		uint32_t *m_n = rr.m_n + (size_t)a * RH_EV_CAP, *m_meta = rr.m_meta + (size_t)a * RH_EV_CAP, *m_pref = rr.m_pref + (size_t)a * (RH_EV_CAP + 1);
		uint32_t nm = 0, pref = 0;
		int32_t rep_st = 0, rep_en = 0, rep_len = 0;
		uint64_t hprev = 0, hcur = ns ? sx[0] >> 6 : 0, hnext;
		for (uint32_t i = 0; i < ns; ++i) {
			hnext = i + 1 < ns ? sx[i + 1] >> 6 : 0;
			const uint32_t cnt = s_n[i];
			if (cnt != 0) {
				const uint32_t q_pos = (uint32_t)sy[i], q_span = (uint32_t)(sx[i] & 63u);
				const uint32_t tandem = ((i > 0 && hcur == hprev) || (i + 1 < ns && hcur == hnext)) ? 1u : 0u;
				if (cnt > (uint32_t)o.mid_occ) {
					const int32_t st = (int32_t)(q_pos >> 1) + 1, en = st + (int32_t)q_span + 1;
					if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
					else rep_en = en;
				} else {
					m_val[nm] = s_val[i]; m_n[nm] = cnt; m_meta[nm] = (q_pos >> 1) | (tandem << 31); m_pref[nm] = pref;
					pref += cnt; ++nm;
				}
			}
			hprev = hcur; hcur = hnext;
		}
		rep_len += rep_en - rep_st;
		m_pref[nm] = pref;
		rr.n_match[a] = nm; rr.n_new[a] = pref; rr.rep_len[a] = rep_len;
		atomicAdd((unsigned long long*)&rr.counters[2], (unsigned long long)pref);
	}
}

// ------------------------------------------------------------------------------------------------ k_scan_anchors
// a_off = exclusive scan of (new hits + carried anchors) over the active reads; a_off[n_act] = total.  One block.
__global__ __launch_bounds__(1024) void k_scan_anchors(rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ uint64_t s_part[1024];
	const uint32_t tid = threadIdx.x, nt = blockDim.x, n = rr.n_act;
	const uint32_t per = (n + nt - 1) / nt;
	const uint32_t b = tid * per, e = b + per < n ? b + per : n;
	uint64_t s = 0;
	for (uint32_t i = b; i < e; ++i) s += (uint64_t)rr.n_new[i] + rd.n_prev[rr.act[i]];
	s_part[tid] = s;
	__syncthreads();
	if (tid == 0) { uint64_t run = 0; for (uint32_t i = 0; i < nt; ++i) { const uint64_t v = s_part[i]; s_part[i] = run; run += v; } rr.a_off[n] = run; atomicAdd((unsigned long long*)&rr.counters[3], (unsigned long long)run); }
	__syncthreads();
	uint64_t run = s_part[tid];
	for (uint32_t i = b; i < e; ++i) { rr.a_off[i] = run; run += (uint64_t)rr.n_new[i] + rd.n_prev[rr.act[i]]; }
}

// ------------------------------------------------------------------------------------------------ k_expand
// One block per active read: output anchor j finds its seed by binary search in the occurrence prefix (LDS), gathers the
// 8-byte position word (the random HBM reads of the path) and writes the 16-byte anchor coalesced.
__global__ __launch_bounds__(NT) void k_expand(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ uint32_t s_pref[RH_EV_CAP + 1];
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const uint32_t np = rd.n_prev[r];
	const rh_mm128_t *pin = rr.prev_in + rd.prev_off[r];
	if (rr.skip[a]) {	// chunk dropped after event detection: carried anchors stay untouched (rmap.cpp:232-235)
		for (uint32_t j = tid; j < np; j += NT) rr.prev_out[base + j] = pin[j];
		__syncthreads();
		if (tid == 0) rd.prev_off[r] = base;
		return;
	}
	const uint32_t nm = rr.n_match[a], nn = rr.n_new[a];
	const uint64_t *m_val = rr.m_val + (size_t)a * RH_EV_CAP;
	const uint32_t *m_n = rr.m_n + (size_t)a * RH_EV_CAP, *m_meta = rr.m_meta + (size_t)a * RH_EV_CAP, *m_pref = rr.m_pref + (size_t)a * (RH_EV_CAP + 1);
	for (uint32_t i = tid; i <= nm; i += NT) s_pref[i] = m_pref[i];
	__syncthreads();
	const uint32_t q_off = rd.ev_off[r];
	const uint64_t span = (uint64_t)(ix.sp.k + ix.sp.e - 1);
	rh_mm128_t *anc = rr.anc + base;
	for (uint32_t j = tid; j < nn; j += NT) {
		uint32_t lo = 0, hi = nm;   // largest s with s_pref[s] <= j
		while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_pref[mid] <= j) lo = mid; else hi = mid; }
		const uint32_t s = lo, k = j - s_pref[s];
		const uint64_t hit = m_n[s] == 1 ? m_val[s] : ix.pos[m_val[s] + k];
		const uint32_t meta = m_meta[s];
		rh_mm128_t p;
		p.x = (hit & 0x7FFFFFFF80000000ull) | (uint64_t)((uint32_t)(hit >> 1) & 0x7FFFFFFFu);
		if (hit & 1ull) p.x |= 1ull << 63;
		p.y = span << 32 | (uint64_t)(uint32_t)((meta & 0x7FFFFFFFu) + q_off);   // seg_id (y >> 40) is 0 for reads
		if (meta >> 31) p.y |= 1ull << 38;
		anc[j] = p;
	}
	for (uint32_t j = tid; j < np; j += NT) anc[nn + j] = pin[j];
}

// ------------------------------------------------------------------------------------------------ exact radix_sort_128x
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

__global__ void k_sort(rh_dev_round rr)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint64_t base = rr.a_off[a];
	const uint32_t n = (uint32_t)(rr.a_off[a + 1] - base);
	rh_radix_sort_128x(rr.anc + base, n, (uint32_t*)(rr.ws + base * RH_WS_PER_ANCHOR));
}

__global__ void k_sort_segments(uint32_t n_seg, rh_mm128_t *arr, const uint64_t *off, unsigned char *ws)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_seg) return;
	rh_radix_sort_128x(arr + off[s], (uint32_t)(off[s + 1] - off[s]), (uint32_t*)(ws + (size_t)s * 2048));
}

// ------------------------------------------------------------------------------------------------ k_chain (DP)
// One read per lane; f/p/v/t live in the read's scratch slice.  Window start, skip counter, t[] marks and the max_ii
// rescue are order dependent (lchain.c:439-505) and evaluated in the reference's order.
__global__ void k_chain(rh_dev_opt o, rh_dev_round rr)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint64_t base = rr.a_off[a];
	const int32_t n = (int32_t)(rr.a_off[a + 1] - base);
	if (n == 0) return;
	const rh_mm128_t *an = rr.anc + base;
	int32_t *f = (int32_t*)(rr.ws + base * RH_WS_PER_ANCHOR), *p = f + n, *v = p + n, *t = v + n;
	int32_t max_dist_t = o.max_dist_t, max_dist_q = o.max_dist_q;
	const int32_t bw = o.bw;
	if (max_dist_t < bw) max_dist_t = bw;
	if (max_dist_q < bw) max_dist_q = bw;
	for (int32_t i = 0; i < n; ++i) t[i] = 0;
	int32_t st = 0, max_ii = -1;
	for (int32_t i = 0; i < n; ++i) {
		const uint64_t xi = an[i].x, yi = an[i].y;
		int32_t max_j = -1, max_f = (int32_t)((yi >> 32) & 63), n_skip = 0, j;
		while (st < i && (xi >> 32 != an[st].x >> 32 || xi > an[st].x + (uint64_t)max_dist_t)) ++st;
		if (i - st > o.max_iter) st = i - o.max_iter;
		for (j = i - 1; j >= st; --j) {
			int32_t sc = rh_pair_score(xi, yi, an[j].x, an[j].y, max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
			if (sc == RH_SCORE_NONE) continue;
			sc += f[j];
			if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
			else if (t[j] == i) { if (++n_skip > o.max_skip) break; }
			if (p[j] >= 0) t[p[j]] = i;
		}
		const int32_t end_j = j;
		if (max_ii < 0 || xi - an[max_ii].x > (uint64_t)(int64_t)max_dist_t) {
			int32_t mx = INT32_MIN;
			max_ii = -1;
			for (j = i - 1; j >= st; --j) if (mx < f[j]) { mx = f[j]; max_ii = j; }
		}
		if (max_ii >= 0 && max_ii < end_j) {
			const int32_t tmp = rh_pair_score(xi, yi, an[max_ii].x, an[max_ii].y, max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
			if (tmp != RH_SCORE_NONE && max_f < tmp + f[max_ii]) { max_f = tmp + f[max_ii]; max_j = max_ii; }
		}
		f[i] = max_f; p[i] = max_j;
		v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : max_f;
		if (max_ii < 0 || (xi - an[max_ii].x <= (uint64_t)(int64_t)max_dist_t && f[max_ii] < f[i])) max_ii = i;
	}
}

// ------------------------------------------------------------------------------------------------ k_backtrack
// lchain.c:47-75
RH_DEV int32_t bk_end(int32_t max_drop, const rh_mm128_t *z, const int32_t *f, const int32_t *p, int32_t *t, int32_t k)
{
	int32_t i = (int32_t)z[k].y, end_i = -1, max_i = i, max_s = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		t[i] = 2;
		end_i = i = p[i];
		const int32_t s = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (s > max_s) { max_s = s; max_i = i; }
		else if (max_s - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = (int32_t)z[k].y; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
	return max_i;
}

// One read per lane: mg_chain_backtrack (lchain.c:95-194) + compact_a (:214-281).  Outputs: chained anchors (chains
// ordered by target position) over the front of the anchor slice, their pre-sort copy = the next chunk's carried
// anchors, and u[] = score << 32 | count.
__global__ void k_backtrack(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const int32_t n = (int32_t)(rr.a_off[a + 1] - base);
	rh_mm128_t *an = rr.anc + base;
	unsigned char *wsr = rr.ws + base * RH_WS_PER_ANCHOR;
	int32_t *f = (int32_t*)wsr, *p = f + n, *v = p + n, *t = v + n;
	rh_mm128_t *z = (rh_mm128_t*)(wsr + (size_t)16 * n);
	uint64_t *u = rr.u + base;
	rh_mm128_t *pa = rr.prev_out + base;
	int32_t n_u = 0, n_v = 0, n_z = 0;
	const int32_t min_sc = o.min_sc, min_cnt = o.min_cnt, max_drop = o.bw;
	for (int32_t i = 0; i < n; ++i) if (f[i] >= min_sc) { z[n_z].x = (uint64_t)(int64_t)f[i]; z[n_z].y = (uint64_t)i; ++n_z; }
	if (n_z > 0) {
		rh_radix_sort_128x(z, (uint32_t)n_z, (uint32_t*)(wsr + (size_t)56 * n));
		for (int32_t i = 0; i < n; ++i) t[i] = 0;
		for (int32_t k = n_z - 1; k >= 0; --k) {
			if (t[z[k].y] != 0) continue;
			const int32_t n_v0 = n_v;
			const int32_t end_i = bk_end(max_drop, z, f, p, t, k);
			int32_t i;
			for (i = (int32_t)z[k].y; i != end_i; i = p[i]) { v[n_v++] = i; t[i] = 1; }
			const int32_t sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
			if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) u[n_u++] = (uint64_t)(uint32_t)sc << 32 | (uint64_t)(uint32_t)(n_v - n_v0);
			else n_v = n_v0;
		}
	}
	if (n_u == 0) {
		rr.n_u[a] = 0; rr.n_v[a] = 0;
		rd.n_prev[r] = 0; rd.prev_off[r] = base;
		return;
	}
	// gather chain members (reverse of backtrack order) into pa; pa is exactly what the next chunk carries
	int32_t k = 0;
	for (int32_t i = 0; i < n_u; ++i) {
		const int32_t k0 = k, ni = (int32_t)u[i];
		for (int32_t j = 0; j < ni; ++j) pa[k++] = an[v[k0 + (ni - j - 1)]];
	}
	// order chains by the target coordinate of their first anchor
	rh_mm128_t *w = z;                       // z is dead
	uint64_t *u2 = (uint64_t*)(w + n_u);
	k = 0;
	for (int32_t i = 0; i < n_u; ++i) { w[i].x = pa[k].x; w[i].y = (uint64_t)(uint32_t)k << 32 | (uint64_t)(uint32_t)i; k += (int32_t)u[i]; }
	rh_radix_sort_128x(w, (uint32_t)n_u, (uint32_t*)(wsr + (size_t)56 * n));
	k = 0;
	for (int32_t i = 0; i < n_u; ++i) {
		const int32_t j = (int32_t)w[i].y, cnt = (int32_t)u[j], src = (int32_t)(w[i].y >> 32);
		u2[i] = u[j];
		for (int32_t m = 0; m < cnt; ++m) an[k + m] = pa[src + m];
		k += cnt;
	}
	for (int32_t i = 0; i < n_u; ++i) u[i] = u2[i];
	rr.n_u[a] = (uint32_t)n_u; rr.n_v[a] = (uint32_t)n_v;
	rd.n_prev[r] = (uint32_t)n_v; rd.prev_off[r] = base;
	atomicAdd((unsigned long long*)&rr.counters[4], (unsigned long long)n_v);
}

// ------------------------------------------------------------------------------------------------ k_regions
struct rh_reg {
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, n_sub, score0;
	uint32_t mapq, rev, hash;
};

RH_DEV float logf_int(int32_t v, const float *tab) { return (v >= 0 && (uint32_t)v < RH_LOGF_N) ? tab[v] : logf((float)v); }

// hit.c:312-336
RH_DEV void sync_regs(int32_t n, rh_reg *r, int32_t *tmp)
{
	if (n <= 0) return;
	int32_t max_id = -1;
	for (int32_t i = 0; i < n; ++i) max_id = max_id > r[i].id ? max_id : r[i].id;
	const int32_t n_tmp = max_id + 1;
	for (int32_t i = 0; i < n_tmp; ++i) tmp[i] = -1;
	for (int32_t i = 0; i < n; ++i) if (r[i].id >= 0) tmp[r[i].id] = i;
	for (int32_t i = 0; i < n; ++i) {
		rh_reg &q = r[i];
		q.id = i;
		if (q.parent == -2) q.parent = i;
		else if (q.parent >= 0 && tmp[q.parent] >= 0) q.parent = tmp[q.parent];
		else q.parent = -1;
	}
}

// One read per lane: mm_gen_regs (hit.c:100-150), mm_set_parent (:195-263), mm_select_sub (:338-367), mm_set_mapq
// (:502-539), then the mapping decision of map_worker_for (rmap.cpp:423-500) and the bookkeeping at the end of
// ri_map_frag (:386).
__global__ void k_regions(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, const float *logf_tab)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t r = rr.act[a];
	if (rr.skip[a]) { rd.ls_ncregs[r] = 0; return; }   // creg freed at the top of the iteration, chunk dropped: no regions
	const uint64_t base = rr.a_off[a];
	const int32_t n_u = (int32_t)rr.n_u[a];
	const uint32_t n_events = rr.n_ev[a], offset = rd.ev_off[r];
	int32_t n_regs = n_u;
	rh_reg best; best.cnt = 0; best.score = 0; best.mapq = 0; best.qs = best.qe = best.rs = best.re = best.rid = 0; best.rev = 0;
	int stop = 0;
	if (n_u > 0) {
		const rh_mm128_t *an = rr.anc + base;
		const uint64_t *u = rr.u + base;
		unsigned char *wsr = rr.ws + base * RH_WS_PER_ANCHOR;
		rh_reg *rg = (rh_reg*)wsr;
		rh_mm128_t *z = (rh_mm128_t*)(wsr + (size_t)64 * n_u);
		uint64_t *cov = (uint64_t*)(wsr + (size_t)80 * n_u);
		int32_t *w = (int32_t*)(wsr + (size_t)88 * n_u), *tmp = (int32_t*)(wsr + (size_t)92 * n_u);
		uint32_t hash = 0;
		hash ^= rh_wang32(offset + n_events) + rh_wang32(11u);
		hash = rh_wang32(hash);
		// --- regions from chains, ordered by (score, hash of first anchor) descending
		int32_t k = 0;
		for (int32_t i = 0; i < n_u; ++i) {
			const uint32_t h = (uint32_t)rh_mix64_nomask((rh_mix64_nomask(an[k].x) + rh_mix64_nomask(an[k].y)) ^ (uint64_t)hash);
			z[i].x = u[i] ^ (uint64_t)h;
			z[i].y = (uint64_t)(uint32_t)k << 32 | (uint64_t)(uint32_t)(int32_t)u[i];
			k += (int32_t)u[i];
		}
		rh_radix_sort_128x(z, (uint32_t)n_u, (uint32_t*)wsr);   // regs area is still free here
		for (int32_t i = 0; i < n_u >> 1; ++i) { const rh_mm128_t tt = z[i]; z[i] = z[n_u - 1 - i]; z[n_u - 1 - i] = tt; }
		// the sort scratch overlapped rg[]: z must be read before rg[i] is written only for i where areas overlap; z lives
		// behind the regs area, so there is no overlap
		for (int32_t i = 0; i < n_u; ++i) {
			rh_reg q;
			q.id = i; q.parent = -1; q.subsc = 0; q.n_sub = 0;
			q.score = q.score0 = (int32_t)(z[i].x >> 32);
			q.hash = (uint32_t)z[i].x;
			q.cnt = (int32_t)z[i].y;
			q.as = (int32_t)(z[i].y >> 32);
			const int32_t s0 = q.as, s1 = q.as + q.cnt - 1;
			q.rev = (uint32_t)(an[s0].x >> 63);
			q.rid = (int32_t)(an[s0].x << 1 >> 33);
			q.rs = (int32_t)an[s0].x; q.re = (int32_t)an[s1].x + 1;
			q.qs = (int32_t)an[s0].y; q.qe = (int32_t)an[s1].y + 1;
			q.mapq = 0;
			rg[i] = q;
		}
		// --- primary / secondary by query overlap
		{
			int32_t kk = 1;
			w[0] = 0; rg[0].parent = 0;
			const int hard = (o.flag & RH_M_HARD_MLEVEL) != 0;
			for (int32_t i = 1; i < n_u; ++i) {
				rh_reg &ri = rg[i];
				const int32_t si = ri.qs, ei = ri.qe;
				int32_t n_cov = 0, uncov = 0, j;
				bool decided_new = false;
				if (!hard) {
					for (j = 0; j < kk; ++j) {
						const rh_reg &rp = rg[w[j]];
						int32_t sj = rp.qs, ej = rp.qe;
						if (ej <= si || sj >= ei) continue;
						if (sj < si) sj = si;
						if (ej > ei) ej = ei;
						cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint64_t)(uint32_t)ej;
					}
					if (n_cov == 0) decided_new = true;
					else {
						for (int32_t x1 = 1; x1 < n_cov; ++x1) {   // ascending sort of the covered intervals
							const uint64_t cv = cov[x1]; int32_t y1 = x1;
							while (y1 > 0 && cov[y1 - 1] > cv) { cov[y1] = cov[y1 - 1]; --y1; }
							cov[y1] = cv;
						}
						int32_t x = si;
						for (j = 0; j < n_cov; ++j) {
							if ((int32_t)(cov[j] >> 32) > x) uncov += (int32_t)(cov[j] >> 32) - x;
							x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
						}
						if (ei > x) uncov += ei - x;
					}
				}
				j = kk;
				if (!decided_new) {
					for (j = 0; j < kk; ++j) {
						rh_reg &rp = rg[w[j]];
						const int32_t sj = rp.qs, ej = rp.qe;
						if (ej <= si || sj >= ei) continue;
						const int32_t mn = ej - sj < ei - si ? ej - sj : ei - si;
						const int32_t mx = ej - sj > ei - si ? ej - sj : ei - si;
						const int32_t ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
						if ((float)ol / (float)mn - (float)uncov / (float)mx > o.mask_level && uncov <= o.mask_len) {
							const int32_t sci = ri.score;
							ri.parent = rp.parent;
							rp.subsc = rp.subsc > sci ? rp.subsc : sci;
							if (ri.cnt >= rp.cnt) ++rp.n_sub;
							break;
						}
					}
				}
				if (j == kk) { w[kk++] = i; ri.parent = i; ri.n_sub = 0; }
			}
		}
		// --- drop secondaries (mm_select_sub, check_strand = 1)
		if (!(o.flag & RH_M_ALL_CHAINS) && o.pri_ratio > 0.0f) {
			int32_t kk = 0, n_2nd = 0;
			for (int32_t i = 0; i < n_regs; ++i) {
				const int32_t pp = rg[i].parent;
				if (pp == i) rg[kk++] = rg[i];
				else if (((float)rg[i].score >= (float)rg[pp].score * o.pri_ratio) && n_2nd < o.best_n) {
					if (!(rg[i].qs == rg[pp].qs && rg[i].qe == rg[pp].qe && rg[i].rid == rg[pp].rid && rg[i].rs == rg[pp].rs && rg[i].re == rg[pp].re)) { rg[kk++] = rg[i]; ++n_2nd; }
				} else if (n_2nd < o.best_n && rg[i].score > o.min_strand_sc && rg[i].rev != rg[pp].rev) { rg[kk++] = rg[i]; ++n_2nd; }
			}
			if (kk != n_regs) sync_regs(kk, rg, tmp);
			n_regs = kk;
		}
		// --- MAPQ
		{
			int64_t sum_sc = 0;
			for (int32_t i = 0; i < n_regs; ++i) if (rg[i].parent == rg[i].id) sum_sc += rg[i].score;
			const float uniq_ratio = (float)sum_sc / (float)(sum_sc + rr.rep_len[a]);
			for (int32_t i = 0; i < n_regs; ++i) {
				rh_reg &q = rg[i];
				const float pen_s1 = (float)((q.score > 100 ? 1.0 : 0.01 * (double)q.score) * (double)uniq_ratio);
				float pen_cm = q.cnt > 10 ? 1.0f : 0.1f * (float)q.cnt;
				pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
				const int32_t subsc = q.subsc > o.min_sc ? q.subsc : o.min_sc;
				const float x = (float)subsc / (float)q.score0;
				int32_t mapq = (int32_t)(pen_cm * 40.0f * (1.0f - x) * logf_int(q.score, logf_tab));
				mapq -= (int32_t)(4.343f * logf_int(q.n_sub + 1, logf_tab) + .499f);
				mapq = mapq > 0 ? mapq : 0;
				q.mapq = (uint32_t)(mapq < 60 ? mapq : 60);
			}
		}
		// --- mapping decision (non-overlap mode: only chain 0 can be reported)
		if (n_regs == 1 && (int32_t)rg[0].mapq >= o.min_mapq) stop = 1;
		else if (n_regs >= 1) {
			float meanC = 0, meanQ = 0;
			for (int32_t i = 0; i < n_regs; ++i) { meanC += (float)rg[i].score; meanQ += (float)rg[i].mapq; }
			meanC /= (float)n_regs; meanQ /= (float)n_regs;
			const float bestQ = (float)rg[0].mapq, bestC = (float)rg[0].score;
			float r_bestq = (bestQ > 0) ? (bestQ / 30.0f) : 0.0f; if (r_bestq > 1) r_bestq = 1.0f;
			float r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
			float r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
			const float weighted = o.w_bestq * r_bestq + o.w_bestmq * r_bestmq + o.w_bestmc * r_bestmc;
			if (weighted >= o.w_threshold) stop = 1;
		}
		best = rg[0];
	}
	rd.ls_ncregs[r] = n_regs;
	if (n_regs > 0) {
		rd.ls_cnt[r] = best.cnt; rd.ls_score[r] = best.score; rd.ls_mapq[r] = (int32_t)best.mapq;
		rd.ls_qs[r] = best.qs; rd.ls_qe[r] = best.qe; rd.ls_rs[r] = best.rs; rd.ls_re[r] = best.re;
		rd.ls_rid[r] = best.rid; rd.ls_rev[r] = (int32_t)best.rev;
	}
	rd.ev_off[r] = offset + n_events;
	if (stop) { rd.done[r] = 1; rd.stop_chunk[r] = rr.chunk; }
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

// ------------------------------------------------------------------------------------------------ k_finalize
// One thread per read: rmap.cpp:507-586.
__global__ void k_finalize(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_map_record_t *rec)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= rd.n_reads) return;
	const uint32_t qlen = rd.l_sig[r];
	const uint32_t l_chunk = o.chunk_size > qlen ? qlen : o.chunk_size;
	const uint32_t iters = read_n_chunks(o, qlen);
	uint32_t c_count;
	int mapped = rd.done[r] != 0;
	if (mapped) c_count = rd.stop_chunk[r];
	else { c_count = iters; if (c_count > 0) --c_count; }
	const uint32_t offset = rd.ev_off[r];
	const float scale = (offset == 0) ? 0.0f : (o.sample_per_base == 0) ? 0.0f : ((float)(c_count + 1) * (float)l_chunk / (float)offset) / o.sample_per_base;
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
	rec[r] = q;
}

// ------------------------------------------------------------------------------------------------ k_synth_reads
// Bench/test support: the synthetic read generator of rh_synth_core.h, one read per lane, writing straight into HBM.
__global__ void k_synth_reads(rh_synth_cfg_t c, const int32_t *level16, uint64_t first, uint32_t n, int16_t *samples, uint64_t *off, double *cal_off, float *cal_scale)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0) off[n] = (uint64_t)n * c.n_samples;
	if (i >= n) return;
	off[i] = (uint64_t)i * c.n_samples;
	cal_off[i] = c.offset;
	cal_scale[i] = (float)(c.range / c.digitisation);
	rh_sy_generate(c, level16, first + i, samples + (size_t)i * c.n_samples);
}

// ------------------------------------------------------------------------------------------------ launchers
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

void rhk_prefilter(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd) { if (rd.n_reads) RH_LAUNCH(k_prefilter, rd.n_reads, NT, 0, s, o, rd); }
void rhk_events(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_events, r.n_act, NT, 0, s, o, rd, r); }
void rhk_sketch(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_sketch, cdiv(r.n_act, 64), 64, 0, s, o, ix, rd, r); }
void rhk_probe(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_probe, r.n_act, NT, 0, s, o, ix, rd, r); }
void rhk_scan_anchors(hipStream_t s, const rh_dev_reads &rd, const rh_dev_round &r) { RH_LAUNCH(k_scan_anchors, 1, 1024, 0, s, rd, r); }
void rhk_expand(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_expand, r.n_act, NT, 0, s, o, ix, rd, r); }
void rhk_sort(hipStream_t s, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_sort, cdiv(r.n_act, 64), 64, 0, s, r); }
void rhk_chain(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_chain, cdiv(r.n_act, 64), 64, 0, s, o, r); }
void rhk_backtrack(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_backtrack, cdiv(r.n_act, 64), 64, 0, s, o, rd, r); }
void rhk_regions(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r, const float *logf_tab) { if (r.n_act) RH_LAUNCH(k_regions, cdiv(r.n_act, 64), 64, 0, s, o, rd, r, logf_tab); }
void rhk_compact_active(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const uint32_t *act_in, uint32_t n_in, uint32_t next_chunk, uint32_t *act_out, uint32_t *n_out)
{ RH_LAUNCH(k_compact_active, 1, 1024, 0, s, o, rd, act_in, n_in, next_chunk, act_out, n_out); }
void rhk_finalize(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, rh_map_record_t *rec) { if (rd.n_reads) RH_LAUNCH(k_finalize, cdiv(rd.n_reads, 256), 256, 0, s, o, ix, rd, rec); }
void rhk_synth_reads(hipStream_t s, const rh_synth_cfg_t &c, const int32_t *level16, uint64_t first, uint32_t n, int16_t *samples, uint64_t *off, double *cal_off, float *cal_scale)
{ if (n) RH_LAUNCH(k_synth_reads, cdiv(n, 64), 64, 0, s, c, level16, first, n, samples, off, cal_off, cal_scale); }
void rhk_sort_segments(hipStream_t s, uint32_t n_seg, rh_mm128_t *a, const uint64_t *off, unsigned char *ws) { if (n_seg) RH_LAUNCH(k_sort_segments, cdiv(n_seg, 64), 64, 0, s, n_seg, a, off, ws); }
