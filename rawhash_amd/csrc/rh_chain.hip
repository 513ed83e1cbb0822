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

#ifdef RH_KPROF
// development aid: what k_chain_wave's tiles are made of - [0] tiles, [1] anchors, [2] singletons, [3] anchors of small clusters, [4] of large ones, [5] tiles that enter the small path,
// [6] tiles that enter the large path, [7] pair-score rounds (r) of the small path, [8] DP steps (sidx) of the small path, [9] anchors in clusters of exactly two, [10] shader clocks in the large path, [11] in the small path, [12] before either (tile set-up), [13] small path: staging + pair scores, [14] its skip-free steps, [15] max_ii catch-up + generic steps
__device__ unsigned long long rh_kprof_chain[16];
extern "C" __attribute__((visibility("default"))) int rh_debug_kprof_chain(unsigned long long *out, int reset)
{
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rh_kprof_chain), sizeof(rh_kprof_chain)) != hipSuccess) return -1;
	if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rh_kprof_chain), z, sizeof(z)) != hipSuccess) return -1; }
	return 0;
}
#define CPROF(slot, v) do { if (threadIdx.x == 0) atomicAdd(&rh_kprof_chain[slot], (unsigned long long)(v)); } while (0)
#else
#define CPROF(slot, v)
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
// fsn ("free of skips up to n"): min(max_skip, max_iter, CH_SMALL - 1), or 0 to keep the generic step everywhere (RH_CHAIN_GENERIC=1).  The step that settles position s <= fsn of a
// small cluster can neither break on skips (that takes max_skip + 1 non-improving predecessors) nor have its window clamped by max_iter, so its window ends only where the cluster
// starts or the target distance exceeds max_dist_t - and then lchain.c's "max_ii" (the best anchor in reach, tried when it lies BEFORE the walked window) cannot apply: whatever
// lies before the window is out of reach.  Such a step is just the max over the window's scores; max_ii's state, which the later generic steps of longer clusters (s > fsn) start
// from, is caught up for those clusters only.
__global__ __launch_bounds__(64) void k_chain_wave(rh_dev_opt o, rh_dev_round rr, uint32_t tiles_per_read, uint32_t tile_len, int32_t fsn)
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
	#define AN_(i) rh_an_ld(rr, rr.anc, base + (uint64_t)(i))
	// DP output per anchor: {f, p} interleaved (one 8-byte record: the backtrack walk needs both per step), then v[]
	int32_t *gfp = (int32_t*)(rr.ws + base * rr.ws_stride), *gv = gfp + 2 * (size_t)n;
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
	uint64_t x_before = T0 > 0 ? AN_(T0 - 1).x : 0ull;             // x of the anchor preceding the tile
	// software pipeline over the tiles: A = current, B = next (needed to size a cluster that runs over the tile edge), C in flight
	// C stays AS LOADED (one-word anchors are taken apart only when they become B, a whole tile after the load was issued): consumed any earlier, the wait for it would sit right
	// behind the load and every tile would pay the memory latency in full
	uint64_t xB = 0, yB = 0;
	rh_mm128_t rawC; rawC.x = 0; rawC.y = 0;
	if (T0 + (int32_t)lane < n) { const rh_mm128_t q = AN_(T0 + lane); xB = q.x; yB = q.y; }
	if (T0 + 64 + (int32_t)lane < n) rawC = rh_an_raw_ld(rr, rr.anc, base + (uint64_t)(T0 + 64 + (int32_t)lane));
	// An anchor's DP output {f, p}, v stays in its lane's registers while its tile is worked on and is stored - two coalesced stores a tile - at the top of the NEXT tile, after the
	// wait for the prefetched anchors: stored where it is computed (a few lanes a step, several steps a tile) the wait at the next tile's top, which waits for every outstanding
	// memory operation, would also sit right behind the tile's last stores.  (Nothing reads them back before: the ring serves the last 256 anchors.)
	int32_t of = 0, op = -1, ov = 0, oii = 0;
	bool ohave = false;
	int2 *gfp2 = reinterpret_cast<int2*>(gfp);
	#define CH_FLUSH() do { if (ohave) { int2 w_; w_.x = of; w_.y = op; gfp2[oii] = w_; gv[oii] = ov; ohave = false; } } while (0)
	for (int32_t i0 = T0; i0 < n; i0 += 64) {
#ifdef RH_KPROF
		const unsigned long long cp_top = clock64();
#endif
		const int32_t ii = i0 + (int32_t)lane;
		const bool inb_r = ii < n;
		const uint64_t x = xB, y = yB;
		RH_LANDED(rawC.x, rawC.y);                                  // the load issued one tile ago is waited for HERE, before the next one is issued
		CH_FLUSH();
		if (ii + 64 < n) { const rh_mm128_t q = rr.afmt.rec8 ? rh_anchor_unpack(rawC.x, rr.afmt, rr.aq_bits, rr.a_span) : rawC; xB = q.x; yB = q.y; } else { xB = 0; yB = 0; }
		if (ii + 128 < n) rawC = rh_an_raw_ld(rr, rr.anc, base + (uint64_t)(ii + 128));
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
		if (inb && single) { const int32_t sp = (int32_t)((y >> 32) & 63); of = sp; op = -1; ov = sp; }
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
#ifdef RH_KPROF
		CPROF(12, clock64() - cp_top);
		CPROF(0, 1); CPROF(1, __popcll(__ballot(inb))); CPROF(2, __popcll(__ballot(inb && single))); CPROF(3, __popcll(__ballot(small))); CPROF(4, __popcll(mmask));
		CPROF(9, __popcll(__ballot(inb && !single && ce_g - cs_g == 1)));
		if (__ballot(small)) CPROF(5, 1);
		if (mmask) CPROF(6, 1);
		const unsigned long long cp_t0 = clock64();
#endif
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
				else { const rh_mm128_t q = AN_(max_ii); xj = (uint32_t)q.x; yj = (uint32_t)q.y; spj = (int32_t)((q.y >> 32) & 63); }
				const int32_t tmp = rh_pair_score_d((int32_t)yi_lo - (int32_t)yj, (int32_t)(xi_lo - xj), spj, max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
				if (tmp != RH_SCORE_NONE && max_f < tmp + f_ii) { max_f = tmp + f_ii; max_j = max_ii; }
			}
			int32_t vv = max_f;
			if (max_j >= 0) {
				const int32_t vmj = (i - max_j < CH_RING) ? L.v[(uint32_t)max_j & (CH_RING - 1)] : __hip_atomic_load(&gv[max_j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (vmj > max_f) vv = vmj;
			}
			if (lane == (uint32_t)b) { of = max_f; op = max_j; ov = vv; }
			if (lane == 0) {
				const uint32_t sl = (uint32_t)i & (CH_RING - 1);
				L.xlo[sl] = xi_lo; L.ylo[sl] = yi_lo; L.span[sl] = (uint8_t)span_i; L.f[sl] = max_f; L.p[sl] = max_j; L.v[sl] = vv;
			}
			if (max_ii < 0 || ((uint32_t)(xi_lo - xlo_ii) <= D32 && f_ii < max_f)) { max_ii = i; f_ii = max_f; xlo_ii = xi_lo; }
			__syncthreads();
		}
#ifdef RH_KPROF
		const unsigned long long cp_t1 = clock64(); CPROF(10, cp_t1 - cp_t0);
#endif
		// Small clusters (up to CH_SMALL anchors: the bulk of a chunk's anchors), one anchor per lane.  The pair scores - the
		// expensive, DP-independent part of lchain.c:439-505 - are computed for all anchors of the tile at once (predecessor
		// r back, r = 1, 2, ...); the DP itself then runs in steps over the position inside the cluster: step s settles the
		// s-th anchor of every cluster of the tile together, from f / p of its predecessors in the LDS ring.  Every index
		// is a compile-time constant after unrolling, so the per-lane scores stay in VGPRs.
		if (__ballot(small)) {
			const uint32_t xi = (uint32_t)x, yi = (uint32_t)y;
			const int32_t span_i = (int32_t)((y >> 32) & 63);
			if (inb) { const uint32_t sl = (uint32_t)ii & 127u; L.s_xlo[sl] = xi; L.s_ylo[sl] = yi; L.s_span[sl] = (uint8_t)span_i; }
			// (round 6) the FIRST anchor of every small cluster is settled here, with the staging - it has no predecessor: f = v = span, p = -1, max_ii = itself - instead of in
			// a DP step of its own (a ballot, the generic step's code under an empty mask for 63 lanes, a barrier: one of ~3.3 steps a tile)
			if (small && pos == 0) {
				const uint32_t sl = (uint32_t)ii & (CH_RING - 1);
				L.f[sl] = span_i; L.p[sl] = -1; L.v[sl] = span_i;
				of = span_i; op = -1; ov = span_i;
				L.s_mi[(uint32_t)ii & 127u] = ii;
			}
			__syncthreads();
			int32_t sc[CH_SMALL];
#pragma unroll
			for (int r = 1; r < CH_SMALL; ++r) {
				sc[r] = RH_SCORE_NONE;
				if (__ballot(small && pos >= r) == 0) break;
				CPROF(7, 1);
				if (small && pos >= r) {
					const uint32_t sl = (uint32_t)(ii - r) & 127u;
					const uint32_t xj = L.s_xlo[sl], yj = L.s_ylo[sl];
					sc[r] = (uint32_t)(xi - xj) > D32 ? CH_FAR
					        : rh_pair_score_d((int32_t)yi - (int32_t)yj, (int32_t)(xi - xj), (int32_t)L.s_span[sl], max_dist_t, max_dist_q, bw, o.pen_gap, o.pen_skip);
				}
			}
#ifdef RH_KPROF
			const unsigned long long cp_t2 = clock64(); CPROF(13, cp_t2 - cp_t1);
#endif
#pragma unroll
			for (int sidx = 1; sidx < CH_SMALL; ++sidx) {           // the skip-free steps
				if (__ballot(small && pos >= sidx && sidx <= fsn) == 0) break;      // (fsn inside the ballot: as an exit test of its own it keeps the compiler from unrolling the loop)
				CPROF(8, 1);
				if (small && pos == sidx) {
					int32_t max_f = span_i, max_j = -1;
#pragma unroll
					for (int r = 1; r <= sidx; ++r) {
						if (sc[r] > CH_FAR) {                          // a pair that scores (neither RH_SCORE_NONE nor out of reach: x ascends, so everything behind an out-of-reach anchor is too)
							const int32_t cand = sc[r] + L.f[(uint32_t)(ii - r) & (CH_RING - 1)];
							if (cand > max_f) { max_f = cand; max_j = ii - r; }
						}
					}
					int32_t vv = max_f;
					if (max_j >= 0) { const int32_t vmj = L.v[(uint32_t)max_j & (CH_RING - 1)]; if (vmj > max_f) vv = vmj; }
					const uint32_t sl = (uint32_t)ii & (CH_RING - 1);
					L.f[sl] = max_f; L.p[sl] = max_j; L.v[sl] = vv;
					of = max_f; op = max_j; ov = vv;
				}
				__syncthreads();
			}
#ifdef RH_KPROF
			const unsigned long long cp_t3 = clock64(); CPROF(14, cp_t3 - cp_t2);
#endif
			const bool need_mi = small && ce_g - cs_g > fsn;         // the cluster has positions beyond fsn: generic steps follow (here or in the next tile)
			if (__ballot(need_mi)) {
#pragma unroll
			for (int k = 1; k < CH_SMALL; ++k) {                    // max_ii after each skip-free position of those clusters (the rule of lchain.c:479-497, minus the pair it cannot reach)
				if (__ballot(need_mi && pos >= k && k <= fsn) == 0) break;
				if (need_mi && pos == k) {
					int32_t mi = L.s_mi[(uint32_t)(ii - 1) & 127u], fmi = 0;
					uint32_t xmi = 0;
					if (mi >= 0) { xmi = L.s_xlo[(uint32_t)mi & 127u]; fmi = L.f[(uint32_t)mi & (CH_RING - 1)]; }
					if (mi < 0 || (uint32_t)(xi - xmi) > D32) {
						int32_t mx = INT32_MIN;
						mi = -1;
						bool reach = true;
#pragma unroll
						for (int r = 1; r <= k; ++r) {
							reach = reach && sc[r] != CH_FAR;
							if (reach) { const int32_t fj = L.f[(uint32_t)(ii - r) & (CH_RING - 1)]; if (mx < fj) { mx = fj; mi = ii - r; } }
						}
						if (mi >= 0) { fmi = mx; xmi = L.s_xlo[(uint32_t)mi & 127u]; }
					}
					if (mi < 0 || ((uint32_t)(xi - xmi) <= D32 && fmi < L.f[(uint32_t)ii & (CH_RING - 1)])) mi = ii;
					L.s_mi[(uint32_t)ii & 127u] = mi;
				}
				__syncthreads();
			}
#pragma unroll
			for (int sidx = 1; sidx < CH_SMALL; ++sidx) {
				if (sidx <= fsn) continue;
				if (__ballot(small && pos >= sidx) == 0) break;
				CPROF(8, 1);
				if (small && pos == sidx) {
					int32_t max_f = span_i, max_j = -1, n_skip = 0, end_j = 0, nwin = 0;
					uint32_t tm = 0;                                   // bit k: the cluster's k-th anchor carries t[] == i
					bool broke = false;
					// the window: the predecessors in reach (x ascends along the cluster: behind the first one out of reach all are) and within max_iter
#pragma unroll
					for (int r = 1; r <= sidx; ++r) if (nwin == r - 1 && sc[r] != CH_FAR && r <= max_iter) nwin = r;
#pragma unroll
					for (int r = 1; r <= sidx; ++r) {
						if (__ballot(!broke && r <= nwin) == 0) break;  // every lane of the step is through with its walk (a chain that runs straight breaks on max_skip at its 7th predecessor: the steps of the long clusters stop there instead of running all sidx rounds under an empty mask)
						if (!broke && r <= nwin && sc[r] != RH_SCORE_NONE) {
							const uint32_t sl = (uint32_t)(ii - r) & (CH_RING - 1);
							const int32_t fj = L.f[sl], pj = L.p[sl];
							const int32_t cand = sc[r] + fj;
							if (cand > max_f) { max_f = cand; max_j = ii - r; if (n_skip > 0) --n_skip; }
							else if ((tm >> (sidx - r)) & 1u) { if (++n_skip > max_skip) { broke = true; end_j = ii - r; } }
							if (!broke && pj >= 0) tm |= 1u << (pj - cs_g);
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
						for (int r = 1; r <= sidx; ++r) if (r <= nwin) { const int32_t fj = L.f[(uint32_t)(ii - r) & (CH_RING - 1)]; if (mx < fj) { mx = fj; mi = ii - r; } }
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
					of = max_f; op = max_j; ov = vv;
					if (mi < 0 || ((uint32_t)(xi - xmi) <= D32 && fmi < max_f)) mi = ii;
					L.s_mi[(uint32_t)ii & 127u] = mi;
				}
				__syncthreads();                                     // the step's f / p / v / max_ii are in the ring
			}
			}
#ifdef RH_KPROF
			CPROF(15, clock64() - cp_t3);
#endif
		}
#ifdef RH_KPROF
		CPROF(11, clock64() - cp_t1);
#endif
		ohave = inb; oii = ii;
		x_before = x_last;
		if (last_tile) break;
	}
	CH_FLUSH();
	#undef CH_FLUSH
	#undef AN_
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
	int32_t *fp = (int32_t*)(rr.ws + base * rr.ws_stride), *v = fp + 2 * (size_t)n, *t = v + n;   // {f,p} interleaved
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

// ------------------------------------------------------------------------------------------------ RMQ chaining (f4)
// mg_lchain_rmq (lchain.c:606-756): the predecessor of an anchor is the element of least priority pri = -(f + 0.5 gap_pen (x + y))
// among the anchors within max_dist on the query, looked up in a balanced tree keyed by (y, index); a second "inner" tree over the
// last max_dist_inner target bases is walked anchor by anchor when the first choice is not exact.  The reference's tree (klib krmq.h)
// is an AVL tree whose nodes carry their subtree's size and a pointer to its element of least pri, and equal pri values are told
// apart by POSITION IN THE TREE (an ancestor beats its descendants, the left subtree the right one) - so which of two equally good
// predecessors is chosen depends on the exact insertion / deletion / rotation history.  That history is reproduced here: one lane
// per read, the AVL tree as index arrays in the read's scratch (node id = anchor index), every operation the klib way (insertion
// with a single re-balancing point, deletion by in-order successor, the two rotations with their subtree-minimum updates).
// An opt-in, accuracy-for-speed mode of the reference (--rmq, --bw-long); serial per read by construction.
// Node arrays are indexed by node id = anchor index.  Round 5: the live nodes of a read are the anchors within max_dist of the current one - a window of
// consecutive indices - so where that window never exceeds RQ_RING nodes (k_chain_rmq finds out first) the arrays are RINGS IN LDS (index & mask) instead
// of arrays in the read's scratch in HBM, and a node's priority and query position are cached beside them: a tree operation is ~40 dependent loads per anchor,
// 1 - 2 us each from HBM under load, ~100 ns from LDS.
template <class V> struct rq_arr { V *p; uint32_t mask; __device__ __forceinline__ V &operator[](int32_t i) const { return p[(uint32_t)i & mask]; } };
struct rq_tree {
	rq_arr<int32_t> l, r, s; rq_arr<uint32_t> sz; rq_arr<int8_t> bal;   // children, subtree minimum (node id), subtree size, balance factor
	int32_t root;
};
#ifndef RQ_RING_S
#define RQ_RING_S 64                                               // the narrowest class: 4.9 KB a wavefront, 32 wavefronts per CU (the walk is one lane following pointers: resident wavefronts are what hides its latency)
#endif
#ifndef RQ_RING
#define RQ_RING 128                                                // nodes of a tree held in LDS (a power of two); 66 bytes per node - both trees, priority, anchor, {f, p} -: 8.4 KB a wavefront, 18 wavefronts per CU
#endif
#ifndef RQ_RING_BIG
#define RQ_RING_BIG 512                                            // ... for the reads whose window is wider (a second launch, 34 KB a wavefront); wider still: the arrays in HBM
#endif
#define RQ_NIL (-1)
#define RQ_FAKE (-2)                                                // the stand-in parent of the root during a deletion (krmq.h:247)
#define RQ_DEPTH 64

struct rq_env { const rh_mm128_t *an; const int32_t *fp; float pen_gap; bool ring; double *pri; uint64_t *ay; uint32_t mask; int32_t *pp0, *pp1; int8_t *pb0, *pb1; };   // pp / pb: the search paths of an operation (RQ_DEPTH entries each, in LDS: as local arrays they cost the kernel 357 VGPRs - one wavefront per SIMD - and scratch memory on every step)   // pri / ay: LDS rings of the live nodes' priorities and y words (ring; else computed from an / fp - the pointers are to LDS either way, never null: a pointer that may be null is generic and read with flat instructions)
RH_DEV double rq_pri_val(float pen_gap, int32_t f, uint64_t x, uint64_t y)
{
	const double g = 0.5 * (double)pen_gap;                       // (0.5 * chn_pen_gap) * (x + y): the reference's association (lchain.c:672)
	return -((double)f + g * (double)((int32_t)x + (int32_t)y));
}
RH_DEV double rq_pri_of(const rq_env &E, int32_t j) { return rq_pri_val(E.pen_gap, E.fp[2 * j], E.an[j].x, E.an[j].y); }
RH_DEV double rq_pri(const rq_env &E, int32_t j) { return E.ring ? E.pri[(uint32_t)j & E.mask] : rq_pri_of(E, j); }
RH_DEV int rq_cmp_key(int32_t ya, int64_t ia, const rq_env &E, int32_t b)	// lc_elem_cmp (lchain.c:539) of key (ya, ia) against node b
{
	const int32_t yb = E.ring ? (int32_t)E.ay[(uint32_t)b & E.mask] : (int32_t)E.an[b].y;
	return ya < yb ? -1 : ya > yb ? 1 : (ia > (int64_t)b) - (ia < (int64_t)b);
}
// (no references that may be to the fake root's local OR to a ring: such a pointer is generic, and every access through it a flat instruction)
RH_DEV int32_t &rq_kid(const rq_tree &T, int32_t p, int which) { return which ? T.r[p] : T.l[p]; }
RH_DEV int32_t rq_child_get(const rq_tree &T, int32_t fake_l, int32_t p, int which) { return p == RQ_FAKE ? fake_l : rq_kid(T, p, which); }
RH_DEV void rq_child_set(const rq_tree &T, int32_t &fake_l, int32_t p, int which, int32_t v) { if (p == RQ_FAKE) fake_l = v; else rq_kid(T, p, which) = v; }
RH_DEV uint32_t rq_csize(const rq_tree &T, int32_t p, int which) { const int32_t c = which ? T.r[p] : T.l[p]; return c == RQ_NIL ? 0u : T.sz[c]; }
RH_DEV void rq_update_min(rq_tree &T, const rq_env &E, int32_t p, int32_t q, int32_t r)	// krmq.h:154-157
{
	// (the walk is one lane waiting for its own loads: everything a step may need is loaded at once - an absent child reads p's own entry instead -, then chosen)
	const int32_t sq = T.s[q == RQ_NIL ? p : q], sr = T.s[r == RQ_NIL ? p : r];
	const double pp = rq_pri(E, p), pq = rq_pri(E, q == RQ_NIL ? p : sq), pr = rq_pri(E, r == RQ_NIL ? p : sr);
	const bool keep = q == RQ_NIL || pp < pq;
	int32_t s = keep ? p : sq;
	const double ps = keep ? pp : pq;
	s = (r == RQ_NIL || ps < pr) ? s : sr;
	T.s[p] = s;
}
RH_DEV int32_t rq_rotate1(rq_tree &T, const rq_env &E, int32_t p, int dir)	// (a,(b,c)q)p => ((a,b)p,c)q   krmq.h:159-170
{
	const int opp = 1 - dir;
	const int32_t q = rq_kid(T, p, opp), s = T.s[p];
	const uint32_t size_p = T.sz[p];
	T.sz[p] -= T.sz[q] - rq_csize(T, q, dir);
	T.sz[q] = size_p;
	rq_update_min(T, E, p, rq_kid(T, p, dir), rq_kid(T, q, dir));
	T.s[q] = s;
	rq_kid(T, p, opp) = rq_kid(T, q, dir);
	rq_kid(T, q, dir) = p;
	return q;
}
RH_DEV int32_t rq_rotate2(rq_tree &T, const rq_env &E, int32_t p, int dir)	// (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r   krmq.h:172-192
{
	const int opp = 1 - dir;
	const int32_t q = rq_kid(T, p, opp), r = rq_kid(T, q, dir), s = T.s[p];
	const uint32_t size_x_dir = rq_csize(T, r, dir);
	T.sz[r] = T.sz[p];
	T.sz[p] -= T.sz[q] - size_x_dir;
	T.sz[q] -= size_x_dir + 1u;
	rq_update_min(T, E, p, rq_kid(T, p, dir), rq_kid(T, r, dir));
	rq_update_min(T, E, q, rq_kid(T, q, opp), rq_kid(T, r, opp));
	T.s[r] = s;
	rq_kid(T, p, opp) = rq_kid(T, r, dir); rq_kid(T, r, dir) = p;
	rq_kid(T, q, dir) = rq_kid(T, r, opp); rq_kid(T, r, opp) = q;
	const int b1 = dir == 0 ? +1 : -1;
	if (T.bal[r] == b1) { T.bal[q] = 0; T.bal[p] = (int8_t)-b1; }
	else if (T.bal[r] == 0) { T.bal[q] = 0; T.bal[p] = 0; }
	else { T.bal[q] = (int8_t)b1; T.bal[p] = 0; }
	T.bal[r] = 0;
	return r;
}
RH_DEV void rq_insert(rq_tree &T, const rq_env &E, int32_t x)	// krmq.h:194-242 (x is never present already: node ids are anchor indices)
{
	uint8_t *stack = reinterpret_cast<uint8_t*>(E.pb0);
	int32_t *path = E.pp0;
	const int32_t yx = E.ring ? (int32_t)E.ay[(uint32_t)x & E.mask] : (int32_t)E.an[x].y;
	int32_t bp = T.root, bq = RQ_NIL, p, q;
	int which = 0, top = 0, path_len = 0;
	for (p = bp, q = bq; p != RQ_NIL;) {
		const int32_t pl = T.l[p], pr = T.r[p];                      // (both children with the key: one wait per level, not two)
		const int8_t pbal = T.bal[p];
		const int c = rq_cmp_key(yx, x, E, p);
		if (pbal != 0) { bq = q; bp = p; top = 0; }
		stack[top++] = (uint8_t)(which = (c > 0));
		path[path_len++] = p;
		q = p; p = which ? pr : pl;
	}
	T.bal[x] = 0; T.sz[x] = 1; T.l[x] = RQ_NIL; T.r[x] = RQ_NIL; T.s[x] = x;
	if (q == RQ_NIL) T.root = x; else rq_kid(T, q, which) = x;
	if (bp == RQ_NIL) return;
	for (int i = 0; i < path_len; ++i) ++T.sz[path[i]];
	for (int i = path_len - 1; i >= 0; --i) { rq_update_min(T, E, path[i], T.l[path[i]], T.r[path[i]]); if (T.s[path[i]] != x) break; }
	for (p = bp, top = 0; p != x; p = rq_kid(T, p, stack[top]), ++top) { if (stack[top] == 0) --T.bal[p]; else ++T.bal[p]; }
	if (T.bal[bp] > -2 && T.bal[bp] < 2) return;
	which = T.bal[bp] < 0;
	const int b1 = which == 0 ? +1 : -1;
	q = rq_kid(T, bp, 1 - which);
	int32_t r;
	if (T.bal[q] == b1) { r = rq_rotate1(T, E, bp, which); T.bal[q] = 0; T.bal[bp] = 0; }
	else r = rq_rotate2(T, E, bp, which);
	if (bq == RQ_NIL) T.root = r; else rq_kid(T, bq, bp != T.l[bq]) = r;
}
RH_DEV int32_t rq_find(const rq_tree &T, const rq_env &E, int32_t y, int64_t i)	// krmq.h:81-96
{
	int32_t p = T.root;
	while (p != RQ_NIL) { const int c = rq_cmp_key(y, i, E, p); if (c < 0) p = T.l[p]; else if (c > 0) p = T.r[p]; else break; }
	return p;
}
RH_DEV void rq_erase(rq_tree &T, const rq_env &E, int32_t x)	// krmq.h:244-327, x in the tree
{
	int32_t *path = E.pp0;
	uint8_t *dir = reinterpret_cast<uint8_t*>(E.pb0);
	int32_t fake_l = T.root;                                      // fake.p[0] = root, fake.p[1] = 0
	const int32_t yx = E.ring ? (int32_t)E.ay[(uint32_t)x & E.mask] : (int32_t)E.an[x].y;
	int d = 0, c;
	int32_t p;
	int32_t nl = fake_l, nr = RQ_NIL;                               // the children of the node the search stands on (the fake root's: root, nil)
	for (c = -1, p = RQ_FAKE; c;) {
		const int which = c > 0;
		dir[d] = (uint8_t)which; path[d++] = p;
		p = which ? nr : nl;
		if (p == RQ_NIL) return;
		nl = T.l[p]; nr = T.r[p];
		c = rq_cmp_key(yx, x, E, p);
	}
	for (int i = 1; i < d; ++i) --T.sz[path[i]];
	if (T.r[p] == RQ_NIL) rq_child_set(T, fake_l, path[d - 1], dir[d - 1], T.l[p]);
	else {
		int32_t q = T.r[p];
		if (T.l[q] == RQ_NIL) {
			T.l[q] = T.l[p]; T.bal[q] = T.bal[p];
			rq_child_set(T, fake_l, path[d - 1], dir[d - 1], q);
			path[d] = q; dir[d++] = 1;
			T.sz[q] = T.sz[p] - 1u;
		} else {
			int32_t r;
			const int e = d++;
			for (;;) { dir[d] = 0; path[d++] = q; r = T.l[q]; if (T.l[r] == RQ_NIL) break; q = r; }
			T.l[r] = T.l[p]; T.l[q] = T.r[r]; T.r[r] = T.r[p];
			T.bal[r] = T.bal[p];
			rq_child_set(T, fake_l, path[e - 1], dir[e - 1], r);
			path[e] = r; dir[e] = 1;
			for (int i = e + 1; i < d; ++i) --T.sz[path[i]];
			T.sz[r] = T.sz[p] - 1u;
		}
	}
	for (int i = d - 1; i >= 1; --i) rq_update_min(T, E, path[i], T.l[path[i]], T.r[path[i]]);   // (path[0] is the fake root: its minimum is never read)
	while (--d > 0) {
		const int32_t q = path[d];
		int b1 = 1, b2 = 2;
		const int which = dir[d], other = 1 - which;
		if (which) { b1 = -b1; b2 = -b2; }
		T.bal[q] = (int8_t)(T.bal[q] + b1);
		if (T.bal[q] == b1) break;
		else if (T.bal[q] == b2) {
			const int32_t r = rq_kid(T, q, other);
			if (T.bal[r] == -b1) rq_child_set(T, fake_l, path[d - 1], dir[d - 1], rq_rotate2(T, E, q, which));
			else {
				rq_child_set(T, fake_l, path[d - 1], dir[d - 1], rq_rotate1(T, E, q, which));
				if (T.bal[r] == 0) { T.bal[r] = (int8_t)-b1; T.bal[q] = (int8_t)b1; break; }
				else { T.bal[r] = 0; T.bal[q] = 0; }
			}
		}
	}
	T.root = fake_l;
}
RH_DEV int32_t rq_rmq(const rq_tree &T, const rq_env &E, int32_t ylo, int64_t ilo, int32_t yup, int64_t iup)	// closed interval, krmq.h:110-150
{
	int32_t *path[2] = { E.pp0, E.pp1 };
	int8_t *pcmp[2] = { E.pb0, E.pb1 };
	int plen[2] = {0, 0}, i, c;
	if (T.root == RQ_NIL) return RQ_NIL;
	int32_t p = T.root;
	while (p != RQ_NIL) { const int32_t pl = T.l[p], pr = T.r[p]; c = rq_cmp_key(ylo, ilo, E, p); path[0][plen[0]] = p; pcmp[0][plen[0]++] = (int8_t)c; if (c < 0) p = pl; else if (c > 0) p = pr; else break; }
	p = T.root;
	while (p != RQ_NIL) { const int32_t pl = T.l[p], pr = T.r[p]; c = rq_cmp_key(yup, iup, E, p); path[1][plen[1]] = p; pcmp[1][plen[1]++] = (int8_t)c; if (c < 0) p = pl; else if (c > 0) p = pr; else break; }
	for (i = 0; i < plen[0] && i < plen[1]; ++i) if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
	if (i == plen[0] || i == plen[1]) return RQ_NIL;
	const int lca = i;
	int32_t mn = path[0][lca];
	double mp = rq_pri(E, mn);
	for (i = lca + 1; i < plen[0]; ++i) if (pcmp[0][i] <= 0) {
		const int32_t u = path[0][i];
		if (rq_pri(E, u) < mp) { mn = u; mp = rq_pri(E, u); }
		if (T.r[u] != RQ_NIL && rq_pri(E, T.s[T.r[u]]) < mp) { mn = T.s[T.r[u]]; mp = rq_pri(E, mn); }
	}
	for (i = lca + 1; i < plen[1]; ++i) if (pcmp[1][i] >= 0) {
		const int32_t u = path[1][i];
		if (rq_pri(E, u) < mp) { mn = u; mp = rq_pri(E, u); }
		if (T.l[u] != RQ_NIL && rq_pri(E, T.s[T.l[u]]) < mp) { mn = T.s[T.l[u]]; mp = rq_pri(E, mn); }
	}
	return mn;
}
// comput_sc_simple (lchain.c:557-581)
RH_DEV int32_t rq_sc_simple(const rh_mm128_t &ai, const rh_mm128_t &aj, float pen_gap, float pen_skip, int32_t *exact, int32_t *width)
{
	const int32_t dq = (int32_t)ai.y - (int32_t)aj.y, dr = (int32_t)(ai.x - aj.x);
	const int32_t dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq, q_span = (int32_t)((aj.y >> 32) & 63);
	*width = dd;
	int32_t sc = q_span < dg ? q_span : dg;
	if (exact) *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		const float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		const float lg = dd >= 1 ? rh_log2_approx((float)(dd + 1)) : 0.0f;
		sc -= (int32_t)(lin + .5f * lg);
	}
	return sc;
}

// One read per WAVEFRONT, walked by its lane 0: the tree operations of two reads share no control flow, so 64 reads per wavefront run one after the
// other (SIMT divergence) while the chip holds a few hundred wavefronts; a wavefront per read keeps every SIMD busy with independent walks instead
// (E. coli-scale --rmq: 5.7 k reads/s with a lane per read, CPU reference 21 k).  counts[a] (optional) = anchors of read a when they are not
// a_off[a + 1] - a_off[a] (the re-chaining of chains)
// The widest window of live nodes each read will have - anchors [st, i) with st as the walk below moves it (the size cap only evicts more) - sorted into the
// storage class that holds it: 0 / 1 / 2 = LDS rings of RQ_RING_S / RQ_RING / RQ_RING_BIG nodes, 3 = the arrays in HBM.  The anchors are sorted by x, so
// "first anchor still in range of anchor i - 1" is a binary search: a lane per anchor, 64 at a time.
__global__ __launch_bounds__(64) void k_rmq_class(rh_dev_opt o, rh_dev_round rr, const uint32_t *counts, int32_t max_dist_in, uint8_t *cls, uint32_t min_cls)
{
	const uint32_t a = blockIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint64_t base = rr.a_off[a];
	const int32_t n_lay = (int32_t)(rr.a_off[a + 1] - base), n = counts ? (int32_t)counts[a] : n_lay;
	const rh_mm128_t *an = rr.anc + base;
	const uint64_t md = (uint64_t)(max_dist_in < o.bw ? o.bw : max_dist_in);
	__shared__ uint32_t s_widest;
	if (threadIdx.x == 0) s_widest = 1;
	__syncthreads();
	uint32_t widest = 1;
	for (int32_t i = 1 + (int32_t)threadIdx.x; i < n; i += 64) {
		const uint64_t xs = an[i - 1].x;
		int32_t lo = 0, hi = i - 1;                                     // st after anchor i - 1: the first j with an[j] on xs's target and xs <= an[j].x + md (j = i - 1 is)
		while (lo < hi) {
			const int32_t m = (lo + hi) >> 1;
			const uint64_t xm = an[m].x;
			if (xs >> 32 == xm >> 32 && xs <= xm + md) hi = m; else lo = m + 1;
		}
		const uint32_t w = (uint32_t)(i - lo + 1);                      // (the anchors before i are inserted before the far ones are evicted)
		if (w > widest) widest = w;
	}
	atomicMax(&s_widest, widest);
	__syncthreads();
	if (threadIdx.x == 0) {	// (a wider class than the window needs is as good: min_cls, RH_RQ_MIN_CLASS, puts every read into class min_cls or above - the hardware tests of each class)
		const uint32_t w = s_widest;
		uint32_t k = w < RQ_RING_S ? 0u : w < RQ_RING ? 1u : w < RQ_RING_BIG ? 2u : 3u;
		if (k < min_cls) k = min_cls;
		cls[a] = (uint8_t)k;
		if (rr.counters) atomicAdd((unsigned long long*)&rr.counters[9 + k], 1ull);
	}
}

// (the storage is a property of the launch - RING nodes in LDS, or RING = 0: the arrays in HBM - so that the compiler sees LDS addresses, not pointers that may be either)
template <int RING, int CLS>
__global__ __launch_bounds__(64) void k_chain_rmq(rh_dev_opt o, rh_dev_round rr, const uint32_t *counts, int32_t max_dist_in, int32_t max_dist_inner_in, int32_t cap_rmq_size, const uint8_t *cls)
{
	const uint32_t a = blockIdx.x;
	if (threadIdx.x != 0 || a >= rr.n_act || rr.skip[a] || cls[a] != CLS) return;
	const uint64_t base = rr.a_off[a];
	const int32_t n_lay = (int32_t)(rr.a_off[a + 1] - base);         // the segment: what the later stages lay their arrays out by
	const int32_t n = counts ? (int32_t)counts[a] : n_lay;           // the anchors to chain (the chained anchors of the first pass when re-chaining)
	if (n_lay == 0) return;
	const rh_mm128_t *an = rr.anc + base;
	int32_t *fp = (int32_t*)(rr.ws + base * rr.ws_stride), *v = fp + 2 * (size_t)n_lay, *t = v + n_lay;   // {f,p} interleaved, as the DP kernels leave them
	// the two trees behind f / p / v / t in the read's scratch: 2 x 17 bytes per anchor (RH_WS_PER_ANCHOR = 64 covers 16 + 34)
	constexpr bool ring = RING > 0;
	constexpr int NR = ring ? RING : 1;
	__shared__ int32_t s_tree[2][4][NR];
	__shared__ int8_t s_bal[2][NR];
	__shared__ double s_pri[NR];
	__shared__ uint64_t s_ay[NR], s_ax[NR];                         // ... and the nodes' anchors and {f, p}: what the walk down the inner tree reads of every node it passes
	__shared__ int32_t s_f[NR], s_p[NR], s_v[NR], s_t[NR];          // (s_t: the mark "seen in iteration i" of the inner walk - compared with the current i only, so a slot's older values never match)
	__shared__ int32_t s_pp[2][RQ_DEPTH];
	__shared__ int8_t s_pb[2][RQ_DEPTH];
	rq_tree T[2];
	if (ring) {
		const uint32_t m = (uint32_t)RING - 1u;
		for (int k = 0; k < 2; ++k) {
			T[k].l = rq_arr<int32_t>{s_tree[k][0], m}; T[k].r = rq_arr<int32_t>{s_tree[k][1], m}; T[k].s = rq_arr<int32_t>{s_tree[k][2], m};
			T[k].sz = rq_arr<uint32_t>{reinterpret_cast<uint32_t*>(s_tree[k][3]), m}; T[k].bal = rq_arr<int8_t>{s_bal[k], m}; T[k].root = RQ_NIL;
		}
	} else {
		int32_t *w = t + n_lay;
		const uint32_t m = 0xFFFFFFFFu;
		for (int k = 0; k < 2; ++k) {
			T[k].l = rq_arr<int32_t>{w, m}; T[k].r = rq_arr<int32_t>{w + n_lay, m}; T[k].s = rq_arr<int32_t>{w + 2 * (size_t)n_lay, m};
			T[k].sz = rq_arr<uint32_t>{(uint32_t*)(w + 3 * (size_t)n_lay), m}; w += 4 * (size_t)n_lay; T[k].root = RQ_NIL;
		}
		int8_t *b = (int8_t*)w;
		T[0].bal = rq_arr<int8_t>{b, m}; T[1].bal = rq_arr<int8_t>{b + n_lay, m};
	}
	// positions beyond the anchors of this pass: never a backtrack candidate, never a predecessor
	for (int32_t i = n; i < n_lay; ++i) { fp[2 * i] = INT32_MIN / 2; fp[2 * i + 1] = -1; v[i] = INT32_MIN / 2; }
	#define F_(i) fp[2 * (i)]
	#define P_(i) fp[2 * (i) + 1]
	const rq_env E = { an, fp, o.pen_gap, ring, s_pri, s_ay, (uint32_t)RING - 1u, s_pp[0], s_pp[1], s_pb[0], s_pb[1] };
	const int32_t bw = o.bw;
	int32_t max_dist = max_dist_in, max_dist_inner = max_dist_inner_in;
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	// Ring classes: everything the walk reads again lives in the rings, filled as the walk arrives at an anchor (x, y) and leaves it (f, p, v) - the window
	// [st, i] has fewer than RING anchors (k_rmq_class), so slot i & mask is free when i arrives -, and the next anchor is loaded one iteration ahead:
	// no load from HBM is waited for inside an iteration (there were five or six in a row: an[i], an[i0], an[st], an[st_inner], t[j], f[j] / p[j] / v[max_j]).
	if (!ring) for (int32_t i = 0; i < n; ++i) t[i] = 0;
	else for (int32_t k = 0; k < NR; ++k) s_t[k] = -1;
	#define AX_(j_) (ring ? s_ax[(uint32_t)(j_) & E.mask] : an[j_].x)
	int32_t i0 = 0, st = 0, st_inner = 0;
	rh_mm128_t nxt{};
	if (ring && n > 0) nxt = an[0];                                 // (an empty last read of a slice starts at the end of the anchor array)
	for (int32_t i = 0; i < n; ++i) {
		const rh_mm128_t ai = ring ? nxt : an[i];
		if (ring) { if (i + 1 < n) nxt = an[i + 1]; s_ax[(uint32_t)i & E.mask] = ai.x; s_ay[(uint32_t)i & E.mask] = ai.y; }
		int32_t max_j = -1, max_f = (int32_t)((ai.y >> 32) & 63);
		if (i0 < i && AX_(i0) != ai.x) {	// add in-range anchors
			for (int32_t j = i0; j < i; ++j) {
				if (ring) { const uint32_t sl = (uint32_t)j & E.mask; s_pri[sl] = rq_pri_val(o.pen_gap, s_f[sl], s_ax[sl], s_ay[sl]); }   // (f[j] is final: j < i)
				rq_insert(T[0], E, j); if (max_dist_inner > 0) rq_insert(T[1], E, j);
			}
			i0 = i;
		}
		while (st < i && (ai.x >> 32 != AX_(st) >> 32 || ai.x > AX_(st) + (uint64_t)max_dist || (T[0].root != RQ_NIL && T[0].sz[T[0].root] > (uint32_t)cap_rmq_size))) {
			if (st < i0) rq_erase(T[0], E, st);                          // krmq_find, then krmq_erase (lchain.c:688-690): anchor st is in the tree iff it was inserted - st < i0 -, since nothing else erases
			++st;
		}
		if (max_dist_inner > 0) {
			while (st_inner < i && (ai.x >> 32 != AX_(st_inner) >> 32 || ai.x > AX_(st_inner) + (uint64_t)max_dist_inner || (T[1].root != RQ_NIL && T[1].sz[T[1].root] > (uint32_t)cap_rmq_size))) {
				if (st_inner < i0) rq_erase(T[1], E, st_inner);
				++st_inner;
			}
		}
		#define NF_(j_) (ring ? s_f[(uint32_t)(j_) & E.mask] : F_(j_))
		#define NP_(j_) (ring ? s_p[(uint32_t)(j_) & E.mask] : P_(j_))
		#define NY_(j_) (ring ? (int32_t)s_ay[(uint32_t)(j_) & E.mask] : (int32_t)an[j_].y)
		#define NA_(v_, j_) rh_mm128_t v_; if (ring) { v_.x = s_ax[(uint32_t)(j_) & E.mask]; v_.y = s_ay[(uint32_t)(j_) & E.mask]; } else v_ = an[j_]
		const int32_t q = rq_rmq(T[0], E, (int32_t)ai.y - max_dist, (int64_t)INT32_MAX, (int32_t)ai.y, 0);
		if (q != RQ_NIL) {
			int32_t exact, width, n_skip = 0;
			int32_t j = q;
			NA_(aq, j);
			int32_t sc = NF_(j) + rq_sc_simple(ai, aq, o.pen_gap, o.pen_skip, &exact, &width);
			if (width <= bw && sc > max_f) { max_f = sc; max_j = j; }
			if (!exact && T[1].root != RQ_NIL && (int32_t)ai.y > 0) {
				// krmq_interval for (y - 1, n): the greatest element below it; then walk down the keys from there (krmq_itr_find + itr_prev)
				const int32_t ys = (int32_t)ai.y - 1;
				int32_t lo = RQ_NIL;
				for (int32_t p = T[1].root; p != RQ_NIL;) { const int c = rq_cmp_key(ys, (int64_t)n, E, p); if (c < 0) p = T[1].l[p]; else if (c > 0) { lo = p; p = T[1].r[p]; } else { lo = p; break; } }
				if (lo != RQ_NIL) {
					int32_t *stack = E.pp0;
					int top = -1;
					for (int32_t p = T[1].root; p != RQ_NIL;) { stack[++top] = p; const int c = rq_cmp_key(NY_(lo), (int64_t)lo, E, p); if (c < 0) p = T[1].l[p]; else if (c > 0) p = T[1].r[p]; else break; }
					while (top >= 0) {
						const int32_t qq = stack[top];
						if (NY_(qq) < (int32_t)ai.y - max_dist_inner) break;
						j = qq;
						int32_t width2;
						NA_(aw, j);
						sc = NF_(j) + rq_sc_simple(ai, aw, o.pen_gap, o.pen_skip, nullptr, &width2);
						if (width2 <= bw) {
							if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
							else if ((ring ? s_t[(uint32_t)j & E.mask] : t[j]) == i) { if (++n_skip > o.max_skip) break; }
							{ const int32_t pj = NP_(j); if (pj >= 0) { if (!ring) t[pj] = i; else if (pj > i - NR) s_t[(uint32_t)pj & E.mask] = i; } }   // (a mark on an anchor RING or more behind is never read: the walk meets live nodes only)
						}
						// krmq_itr_prev
						int32_t p = T[1].l[stack[top]];
						if (p != RQ_NIL) { for (; p != RQ_NIL; p = T[1].r[p]) stack[++top] = p; }
						else {
							do { p = stack[top--]; } while (top >= 0 && p == T[1].l[stack[top]]);
							if (top < 0) break;
						}
					}
				}
			}
		}
		F_(i) = max_f; P_(i) = max_j;
		int32_t vi = max_f;
		if (max_j >= 0) { const int32_t vj = ring ? s_v[(uint32_t)max_j & E.mask] : v[max_j]; if (vj > max_f) vi = vj; }   // (max_j is a live node)
		v[i] = vi;
		if (ring) { const uint32_t sl = (uint32_t)i & E.mask; s_f[sl] = max_f; s_p[sl] = max_j; s_v[sl] = vi; }
		#undef NF_
		#undef NP_
		#undef NY_
		#undef NA_
	}
	#undef AX_
	#undef F_
	#undef P_
}

void rhk_chain_rmq(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r, const uint32_t *counts, int32_t max_dist, int32_t max_dist_inner, int32_t cap_rmq_size)
{
	if (!r.n_act) return;
	// (need_exact2 is idle between the anchor sort and the chain-order sort: here it holds each read's storage class)
	const char *mc = getenv("RH_RQ_MIN_CLASS");                    // 0 .. 3: no read takes a narrower storage class than this (test aid: every class on the hardware)
	const uint32_t min_cls = mc ? (uint32_t)(atoi(mc) < 0 ? 0 : atoi(mc) > 3 ? 3 : atoi(mc)) : 0u;
	RH_LAUNCH(k_rmq_class, r.n_act, 64, 0, s, o, r, counts, max_dist, r.need_exact2, min_cls);
	RH_LAUNCH((k_chain_rmq<RQ_RING_S, 0>), r.n_act, 64, 0, s, o, r, counts, max_dist, max_dist_inner, cap_rmq_size, r.need_exact2);
	RH_LAUNCH((k_chain_rmq<RQ_RING, 1>), r.n_act, 64, 0, s, o, r, counts, max_dist, max_dist_inner, cap_rmq_size, r.need_exact2);
	RH_LAUNCH((k_chain_rmq<RQ_RING_BIG, 2>), r.n_act, 64, 0, s, o, r, counts, max_dist, max_dist_inner, cap_rmq_size, r.need_exact2);
	RH_LAUNCH((k_chain_rmq<0, 3>), r.n_act, 64, 0, s, o, r, counts, max_dist, max_dist_inner, cap_rmq_size, r.need_exact2);
}

void rhk_chain(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r)
{
	if (!r.n_act) return;
	if (o.max_iter <= CH_MAX_ITER) {
		const uint32_t tiles = r.max_anchors > (uint32_t)CH_TILE ? (r.max_anchors + CH_TILE - 1) / CH_TILE : 1u;
		int32_t fsn = o.max_skip < o.max_iter ? o.max_skip : o.max_iter;
		fsn = fsn < 0 || getenv("RH_CHAIN_GENERIC") ? 0 : fsn > CH_SMALL - 1 ? CH_SMALL - 1 : fsn;
		RH_LAUNCH(k_chain_wave, r.n_act * tiles, 64, rh_wave_lds(), s, o, r, tiles, tiles > 1 ? (uint32_t)CH_TILE : 0u, fsn);
	}
	else RH_LAUNCH(k_chain_serial, (r.n_act + 63) / 64, 64, 0, s, o, r);
}
