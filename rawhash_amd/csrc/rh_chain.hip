// Chaining DP stage: mg_lchain_dp's scoring loop (reference lchain.c:439-505), one wavefront per read.
//
// Anchors more than max_dist_t apart on the target (or on different targets / strands) never see each other: the window
// start `st` jumps to i and the max_ii state resets.  The sorted anchor array therefore splits into independent
// clusters.  Most anchors of a chunk are singletons (f = span, no predecessor) and are finished in one coalesced sweep;
// the multi-anchor clusters are walked in order, one anchor per iteration, with the up-to-max_iter predecessor window
// evaluated 64 lanes at a time.  The order-dependent parts of the reference loop are resolved exactly inside the wave:
//   * "sc > max_f" (strict, descending j)      -> exclusive prefix-max across lanes
//   * t[] marks ("was j already a predecessor of something seen for this i")  -> LDS marks written by the whole
//     batch, read after a barrier (marks only ever point below the lane that sets them)
//   * n_skip counter with floor at 0 and the early break -> scalar walk over the ballot masks of the two event kinds
// The last RING anchors (coordinates, f, p, v, t) live in an LDS ring; f/p/v go to HBM once.
#include "rh_kernels.h"
#include "rh_devutil.h"

#ifndef CH_RING
#define CH_RING 256
#endif
#define CH_MAX_ITER (CH_RING - 1)

#ifndef CH_SMALL
#define CH_SMALL 16     // clusters of up to this many anchors take the one-anchor-per-lane path (measured: 12 -> 37 ms, 16 -> 33 ms, 20 -> 42 ms per step)
#endif

struct chain_lds {
	uint32_t xlo[CH_RING], ylo[CH_RING];
	int32_t f[CH_RING], p[CH_RING], v[CH_RING], t[CH_RING];
	uint8_t span[CH_RING];
	// small clusters: coordinates of the last two tiles and the max_ii state after each of their anchors
	uint32_t s_xlo[128], s_ylo[128];
	int32_t s_mi[128];
	uint8_t s_span[128];
};

#define CH_FAR (INT32_MIN + 1)   // pair further apart than max_dist_t on the target: ends the predecessor window

#ifndef CH_TILE
#define CH_TILE 4096             // anchors per wavefront when a read's anchors are split (multiple of 64)
#endif

// A read with many anchors (large indexes: tens of thousands per chunk) is split over several wavefronts: the clusters are
// independent, so wavefront `tix` takes the clusters that START in [tix * tile_len, (tix + 1) * tile_len) - it skips the
// tail of a cluster that began before its slice and runs past the end of the slice until the next cluster starts.
__global__ __launch_bounds__(64) void k_chain_wave(rh_dev_opt o, rh_dev_round rr, uint32_t tiles_per_read, uint32_t tile_len)
{
	__shared__ chain_lds L;
	const uint32_t a = blockIdx.x / tiles_per_read, tix = blockIdx.x % tiles_per_read, lane = threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint64_t base = rr.a_off[a];
	const int32_t n = (int32_t)(rr.a_off[a + 1] - base);
	if (n == 0) return;
	const int32_t T0 = (int32_t)(tix * tile_len);
	if (T0 >= n) return;
	const int32_t T1 = tile_len && T0 + (int32_t)tile_len < n ? T0 + (int32_t)tile_len : n;
	bool begun = T0 == 0;
	const rh_mm128_t *an = rr.anc + base;
	// DP output per anchor: {f, p} interleaved (one 8-byte record: the backtrack walk needs both per step), then v[]
	int32_t *gfp = (int32_t*)(rr.ws + base * RH_WS_PER_ANCHOR), *gv = gfp + 2 * (size_t)n;
	int32_t max_dist_t = o.max_dist_t, max_dist_q = o.max_dist_q;
	const int32_t bw = o.bw, max_iter = o.max_iter, max_skip = o.max_skip;
	if (max_dist_t < bw) max_dist_t = bw;
	if (max_dist_q < bw) max_dist_q = bw;
	const uint64_t D64 = (uint64_t)max_dist_t;
	const uint32_t D32 = (uint32_t)max_dist_t;
	for (uint32_t k = lane; k < CH_RING; k += 64) L.t[k] = -1;
	__syncthreads();

	// Anchors are read once, 64 at a time (coalesced); the sequential walk below only touches registers (shuffles) and LDS.
	int32_t st = 0, max_ii = -1, f_ii = 0, open_start = 0;
	uint32_t xlo_ii = 0;
	uint64_t x_before = T0 > 0 ? an[T0 - 1].x : 0ull;              // x of the anchor preceding the tile
	// software pipeline over the tiles: A = current, B = next (needed to size a cluster that runs over the tile edge), C in flight
	uint64_t xB = 0, yB = 0, xC = 0, yC = 0;
	if (T0 + (int32_t)lane < n) { xB = an[T0 + lane].x; yB = an[T0 + lane].y; }
	if (T0 + 64 + (int32_t)lane < n) { xC = an[T0 + 64 + lane].x; yC = an[T0 + 64 + lane].y; }
	for (int32_t i0 = T0; i0 < n; i0 += 64) {
		const int32_t ii = i0 + (int32_t)lane;
		const bool inb_r = ii < n;
		const uint64_t x = xB, y = yB;
		xB = xC; yB = yC;
		if (ii + 128 < n) { xC = an[ii + 128].x; yC = an[ii + 128].y; } else { xC = 0; yC = 0; }
		const uint64_t xprev = (uint64_t)rh_wave_shr1((uint32_t)(x >> 32), (uint32_t)(x_before >> 32)) << 32 | rh_wave_shr1((uint32_t)x, (uint32_t)x_before);
		const bool start_r = inb_r && (ii == 0 || (x >> 32) != (xprev >> 32) || x > xprev + D64);
		const uint64_t x_last = (uint64_t)rh_readlane((uint32_t)(x >> 32), 63u) << 32 | rh_readlane((uint32_t)x, 63u);
		// this wavefront's share: from the first cluster start at or after T0 up to (not including) the first one at or after T1
		const uint64_t sreal = __ballot(start_r);
		bool incl = true, last_tile = false;
		if (!begun) {
			if (sreal) { begun = true; incl = lane >= (uint32_t)__builtin_ctzll(sreal); }
			else incl = false;
		}
		if (T1 < n && i0 + 64 > T1) {
			const uint64_t m = sreal & (T1 > i0 ? ~((1ull << (T1 - i0)) - 1ull) : ~0ull);
			if (m) { incl = incl && lane < (uint32_t)__builtin_ctzll(m); last_tile = true; }
		}
		if (__ballot(incl && inb_r) == 0) { x_before = x_last; if (last_tile) break; continue; }
		const bool inb = inb_r && incl, start = start_r && incl;
		const uint64_t smask = __ballot(start);
		const uint64_t bmask = __ballot(start || !inb);               // cluster boundaries, the end of the array included
		const uint64_t xprevB = (uint64_t)rh_wave_shr1((uint32_t)(xB >> 32), (uint32_t)(x_last >> 32)) << 32 | rh_wave_shr1((uint32_t)xB, (uint32_t)x_last);
		const uint64_t bmaskB = __ballot(ii + 64 >= n || (xB >> 32) != (xprevB >> 32) || xB > xprevB + D64);   // same for the next tile
		const bool nstart = lane < 63 ? ((bmask >> (lane + 1)) & 1ull) != 0 : (bmaskB & 1ull) != 0;
		const bool single = start && nstart;
		if (inb && single) { const int32_t sp = (int32_t)((y >> 32) & 63); gfp[2 * ii] = sp; gfp[2 * ii + 1] = -1; gv[ii] = sp; }
		// the cluster [cs_g, ce_g] (global indices) of every anchor of the tile, from the boundary masks
		int32_t cs_g, ce_g;
		{
			const uint64_t sm = smask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
			cs_g = sm ? i0 + 63 - (int32_t)__clzll(sm) : open_start;
			const uint64_t after = lane < 63 ? bmask >> (lane + 1) : 0ull;
			if (after) ce_g = ii + (int32_t)__builtin_ctzll(after);
			else if (bmaskB) ce_g = i0 + 63 + (int32_t)__builtin_ctzll(bmaskB);
			else ce_g = 0x3FFFFFFF;
		}
		const bool small = inb && !single && ce_g - cs_g < CH_SMALL;
		const int32_t pos = ii - cs_g;
		if (!(bmaskB & 1ull) && smask) open_start = i0 + 63 - (int32_t)__clzll(smask);   // cluster still open at the tile's end
		uint64_t mmask = __ballot(inb && !single && !small);       // members of the larger clusters, walked in order
		while (mmask) {
			const int b = __builtin_ctzll(mmask);
			mmask &= mmask - 1;
			const int32_t i = i0 + b;
			const uint64_t xi = (uint64_t)rh_readlane((uint32_t)(x >> 32), (uint32_t)b) << 32 | rh_readlane((uint32_t)x, (uint32_t)b);
			const uint64_t yi = (uint64_t)rh_readlane((uint32_t)(y >> 32), (uint32_t)b) << 32 | rh_readlane((uint32_t)y, (uint32_t)b);
			if ((smask >> b) & 1ull) { st = i; max_ii = -1; }       // cluster start: window and max_ii state reset
			const uint32_t xi_lo = (uint32_t)xi, yi_lo = (uint32_t)yi;
			const int32_t span_i = (int32_t)((yi >> 32) & 63);
			if (i - st > max_iter) st = i - max_iter;   // clamp first: older ring slots are gone (same result, the test is monotone in st)
			while (st < i && (uint32_t)(xi_lo - L.xlo[st & (CH_RING - 1)]) > D32) ++st;
			int32_t max_f = span_i, max_j = -1, n_skip = 0, end_j = st - 1;
			bool broke = false;
			for (int32_t jtop = i - 1; jtop >= st && !broke; jtop -= 64) {
				const int32_t j = jtop - (int32_t)lane;
				const bool inw = j >= st;
				const uint32_t slot = (uint32_t)j & (CH_RING - 1);
				int32_t sc = RH_SCORE_NONE, fj = 0, pj = -1;
				if (inw) {
					sc = rh_pair_score_d((int32_t)yi_lo - (int32_t)L.ylo[slot], (int32_t)(xi_lo - L.xlo[slot]), (int32_t)L.span[slot], max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
					fj = L.f[slot]; pj = L.p[slot];
				}
				const bool valid = inw && sc != RH_SCORE_NONE;
				if (valid && pj >= st) L.t[(uint32_t)pj & (CH_RING - 1)] = i;
				__syncthreads();
				const bool marked = valid && L.t[slot] == i;
				const int32_t scv = valid ? sc + fj : INT32_MIN;
				// "sc > max_f" in descending-j order = the record highs of scv along the lanes, starting from max_f.  There are
				// only a few per batch: find them one by one with a ballot + readlane instead of a 6-step shuffle scan.
				uint64_t Dm = 0, rem = ~0ull;
				int32_t cur = max_f;
				for (;;) {
					const uint64_t m = __ballot(valid && scv > cur) & rem;
					if (!m) break;
					const int l = __builtin_ctzll(m);
					Dm |= 1ull << l;
					cur = (int32_t)rh_readlane((uint32_t)scv, (uint32_t)l);
					if (l == 63) break;
					rem = ~((2ull << l) - 1ull);
				}
				const bool newmax = ((Dm >> lane) & 1ull) != 0;
				const uint64_t Um = __ballot(valid && !newmax && marked);
				uint64_t ev = Dm | Um;
				int B = 64;
				while (ev) {
					const int e = __builtin_ctzll(ev);
					ev &= ev - 1;
					if ((Dm >> e) & 1ull) { if (n_skip > 0) --n_skip; }
					else if (++n_skip > max_skip) { B = e; break; }
				}
				const uint64_t Deff = B < 64 ? (Dm & ((1ull << B) - 1ull)) : Dm;
				if (Deff) { const int Lm = 63 - __clzll(Deff); max_f = (int32_t)rh_readlane((uint32_t)scv, (uint32_t)Lm); max_j = jtop - Lm; }
				if (B < 64) { end_j = jtop - B; broke = true; }
			}
			// best-scoring anchor still within max_dist_t ("max_ii"), re-derived from the window when it fell out of range
			if (max_ii < 0 || (uint32_t)(xi_lo - xlo_ii) > D32) {
				uint64_t best = 0;
				for (int32_t jtop = i - 1; jtop >= st; jtop -= 64) {
					const int32_t j = jtop - (int32_t)lane;
					uint64_t key = j >= st ? ((uint64_t)(uint32_t)L.f[(uint32_t)j & (CH_RING - 1)] << 32 | (uint64_t)(uint32_t)j) : 0ull;
					for (int d = 32; d > 0; d >>= 1) { const uint64_t ok = __shfl_xor(key, d); if (ok > key) key = ok; }
					if (key > best) best = key;
				}
				if (best) { max_ii = (int32_t)(uint32_t)best; f_ii = (int32_t)(best >> 32); xlo_ii = L.xlo[(uint32_t)max_ii & (CH_RING - 1)]; }
				else max_ii = -1;
			}
			if (max_ii >= 0 && max_ii < end_j) {
				uint32_t xj, yj; int32_t spj;
				if (i - max_ii < CH_RING) { const uint32_t sl = (uint32_t)max_ii & (CH_RING - 1); xj = L.xlo[sl]; yj = L.ylo[sl]; spj = (int32_t)L.span[sl]; }
				else { xj = (uint32_t)an[max_ii].x; yj = (uint32_t)an[max_ii].y; spj = (int32_t)((an[max_ii].y >> 32) & 63); }
				const int32_t tmp = rh_pair_score_d((int32_t)yi_lo - (int32_t)yj, (int32_t)(xi_lo - xj), spj, max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
				if (tmp != RH_SCORE_NONE && max_f < tmp + f_ii) { max_f = tmp + f_ii; max_j = max_ii; }
			}
			int32_t vv = max_f;
			if (max_j >= 0) {
				const int32_t vmj = (i - max_j < CH_RING) ? L.v[(uint32_t)max_j & (CH_RING - 1)] : __hip_atomic_load(&gv[max_j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (vmj > max_f) vv = vmj;
			}
			if (lane == 0) {
				const uint32_t sl = (uint32_t)i & (CH_RING - 1);
				gfp[2 * i] = max_f; gfp[2 * i + 1] = max_j; gv[i] = vv;
				L.xlo[sl] = xi_lo; L.ylo[sl] = yi_lo; L.span[sl] = (uint8_t)span_i; L.f[sl] = max_f; L.p[sl] = max_j; L.v[sl] = vv;
			}
			if (max_ii < 0 || ((uint32_t)(xi_lo - xlo_ii) <= D32 && f_ii < max_f)) { max_ii = i; f_ii = max_f; xlo_ii = xi_lo; }
			__syncthreads();
		}
		// Small clusters (up to CH_SMALL anchors: the bulk of a chunk's anchors), one anchor per lane.  The pair scores - the
		// expensive, DP-independent part of lchain.c:439-505 - are computed for all anchors of the tile at once (predecessor
		// r back, r = 1, 2, ...); the DP itself then runs in steps over the position inside the cluster: step s settles the
		// s-th anchor of every cluster of the tile together, from f / p of its predecessors in the LDS ring.  Every index
		// is a compile-time constant after unrolling, so the per-lane scores stay in VGPRs.
		if (__ballot(small)) {
			const uint32_t xi = (uint32_t)x, yi = (uint32_t)y;
			const int32_t span_i = (int32_t)((y >> 32) & 63);
			if (inb) { const uint32_t sl = (uint32_t)ii & 127u; L.s_xlo[sl] = xi; L.s_ylo[sl] = yi; L.s_span[sl] = (uint8_t)span_i; }
			__syncthreads();
			int32_t sc[CH_SMALL];
#pragma unroll
			for (int r = 1; r < CH_SMALL; ++r) {
				sc[r] = RH_SCORE_NONE;
				if (__ballot(small && pos >= r) == 0) break;
				if (small && pos >= r) {
					const uint32_t sl = (uint32_t)(ii - r) & 127u;
					const uint32_t xj = L.s_xlo[sl], yj = L.s_ylo[sl];
					sc[r] = (uint32_t)(xi - xj) > D32 ? CH_FAR
					        : rh_pair_score_d((int32_t)yi - (int32_t)yj, (int32_t)(xi - xj), (int32_t)L.s_span[sl], max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
				}
			}
#pragma unroll
			for (int sidx = 0; sidx < CH_SMALL; ++sidx) {
				if (__ballot(small && pos >= sidx) == 0) break;
				if (small && pos == sidx) {
					int32_t max_f = span_i, max_j = -1, n_skip = 0, end_j = 0, nwin = 0;
					uint32_t tm = 0;                                   // bit k: the cluster's k-th anchor carries t[] == i
					bool broke = false, far = false;
					int32_t fr[CH_SMALL];
#pragma unroll
					for (int r = 1; r <= sidx; ++r) {
						fr[r] = 0;
						if (far || sc[r] == CH_FAR || r > max_iter) far = true;
						else {
							nwin = r;
							const uint32_t sl = (uint32_t)(ii - r) & (CH_RING - 1);
							const int32_t fj = L.f[sl], pj = L.p[sl];
							fr[r] = fj;
							if (!broke && sc[r] != RH_SCORE_NONE) {
								const int32_t cand = sc[r] + fj;
								if (cand > max_f) { max_f = cand; max_j = ii - r; if (n_skip > 0) --n_skip; }
								else if ((tm >> (sidx - r)) & 1u) { if (++n_skip > max_skip) { broke = true; end_j = ii - r; } }
								if (!broke && pj >= 0) tm |= 1u << (pj - cs_g);
							}
						}
					}
					if (!broke) end_j = ii - nwin - 1;
					// max_ii: state left by the previous anchor of the cluster; re-derived from the window when out of reach
					int32_t mi = sidx == 0 ? -1 : L.s_mi[(uint32_t)(ii - 1) & 127u], fmi = 0;
					uint32_t xmi = 0;
					if (mi >= 0) { xmi = L.s_xlo[(uint32_t)mi & 127u]; fmi = L.f[(uint32_t)mi & (CH_RING - 1)]; }
					if (mi < 0 || (uint32_t)(xi - xmi) > D32) {
						int32_t mx = INT32_MIN;
						mi = -1;
#pragma unroll
						for (int r = 1; r <= sidx; ++r) if (r <= nwin && mx < fr[r]) { mx = fr[r]; mi = ii - r; }
						if (mi >= 0) { fmi = mx; xmi = L.s_xlo[(uint32_t)mi & 127u]; }
					}
					if (mi >= 0 && mi < end_j) {
						const uint32_t sl = (uint32_t)mi & 127u;
						const int32_t tmp = rh_pair_score_d((int32_t)yi - (int32_t)L.s_ylo[sl], (int32_t)(xi - xmi), (int32_t)L.s_span[sl], max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
						if (tmp != RH_SCORE_NONE && max_f < tmp + fmi) { max_f = tmp + fmi; max_j = mi; }
					}
					int32_t vv = max_f;
					if (max_j >= 0) { const int32_t vmj = L.v[(uint32_t)max_j & (CH_RING - 1)]; if (vmj > max_f) vv = vmj; }
					const uint32_t sl = (uint32_t)ii & (CH_RING - 1);
					L.f[sl] = max_f; L.p[sl] = max_j; L.v[sl] = vv;
					gfp[2 * ii] = max_f; gfp[2 * ii + 1] = max_j; gv[ii] = vv;
					if (mi < 0 || ((uint32_t)(xi - xmi) <= D32 && fmi < max_f)) mi = ii;
					L.s_mi[(uint32_t)ii & 127u] = mi;
				}
				__syncthreads();                                     // the step's f / p / v / max_ii are in the ring
			}
		}
		x_before = x_last;
		if (last_tile) break;
	}
}

// Fallback for max_chain_iter > CH_MAX_ITER: the plain serial loop, one read per lane.
__global__ void k_chain_serial(rh_dev_opt o, rh_dev_round rr)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint64_t base = rr.a_off[a];
	const int32_t n = (int32_t)(rr.a_off[a + 1] - base);
	if (n == 0) return;
	const rh_mm128_t *an = rr.anc + base;
	int32_t *fp = (int32_t*)(rr.ws + base * RH_WS_PER_ANCHOR), *v = fp + 2 * (size_t)n, *t = v + n;   // {f,p} interleaved
	#define F_(i) fp[2 * (i)]
	#define P_(i) fp[2 * (i) + 1]
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
			sc += F_(j);
			if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
			else if (t[j] == i) { if (++n_skip > o.max_skip) break; }
			if (P_(j) >= 0) t[P_(j)] = i;
		}
		const int32_t end_j = j;
		if (max_ii < 0 || xi - an[max_ii].x > (uint64_t)(int64_t)max_dist_t) {
			int32_t mx = INT32_MIN;
			max_ii = -1;
			for (j = i - 1; j >= st; --j) if (mx < F_(j)) { mx = F_(j); max_ii = j; }
		}
		if (max_ii >= 0 && max_ii < end_j) {
			const int32_t tmp = rh_pair_score(xi, yi, an[max_ii].x, an[max_ii].y, max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
			if (tmp != RH_SCORE_NONE && max_f < tmp + F_(max_ii)) { max_f = tmp + F_(max_ii); max_j = max_ii; }
		}
		F_(i) = max_f; P_(i) = max_j;
		v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : max_f;
		if (max_ii < 0 || (xi - an[max_ii].x <= (uint64_t)(int64_t)max_dist_t && F_(max_ii) < F_(i))) max_ii = i;
	}
	#undef F_
	#undef P_
}

void rhk_chain(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r)
{
	if (!r.n_act) return;
	if (o.max_iter <= CH_MAX_ITER) {
		const uint32_t tiles = r.max_anchors > (uint32_t)CH_TILE ? (r.max_anchors + CH_TILE - 1) / CH_TILE : 1u;
		RH_LAUNCH(k_chain_wave, r.n_act * tiles, 64, 0, s, o, r, tiles, tiles > 1 ? (uint32_t)CH_TILE : 0u);
	}
	else RH_LAUNCH(k_chain_serial, (r.n_act + 63) / 64, 64, 0, s, o, r);
}
