// Post-DP stages:
//   k_zbuild          candidates f[i] >= min_sc for backtracking (lchain.c:126-140); they are then put into the reference's
//                     radix_sort_128x order by the block sorter of rh_sort.hip (scores are full of ties -> exact mode)
//   k_backtrack_spec  mg_chain_backtrack (lchain.c:95-194, mg_chain_bk_end :47-75): a workgroup per read, 256 / 512 candidates
//                     walked in parallel per round, conflicts re-walked; used-marks and claim stamps in LDS (HBM for the largest reads)
//   k_chain_reorder (+ block sorter)   compact_a (lchain.c:214-281), its gather being the backtrack's
//   k_regions_*       mm_gen_regs (hit.c:100-150), mm_set_parent (:195-263), mm_select_sub (:338-367), mm_set_mapq (:502-539),
//                     the mapping decision of map_worker_for (rmap.cpp:423-500) and the bookkeeping of ri_map_frag (:386):
//                     k_regions_reg keeps the primaries in registers; k_regions_wave (LDS), k_regions / k_regions_big
//                     (serial core) take the reads and option sets it cannot
#include <cstdio>
#include <cstdlib>
#include "rh_kernels.h"
#include "rh_devutil.h"

#ifdef RH_KPROF
__device__ unsigned long long rh_kprof_post[32];
#define KPROF_DECL unsigned long long kp_t0 = clock64()
#define KPROF(slot) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&rh_kprof_post[slot], t_ - kp_t0); kp_t0 = t_; } } while (0)
#define KPROF_ADD(slot, v) do { if (threadIdx.x == 0) atomicAdd(&rh_kprof_post[slot], (unsigned long long)(v)); } while (0)
extern "C" __attribute__((visibility("default"))) int rh_debug_kprof_post(unsigned long long *out, int reset)
{
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rh_kprof_post), sizeof(rh_kprof_post)) != hipSuccess) return -1;
	if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rh_kprof_post), z, sizeof(z)) != hipSuccess) return -1; }
	return 0;
}
#else
#define KPROF_DECL
#define KPROF(slot)
#define KPROF_ADD(slot, v)
#endif

// ------------------------------------------------------------------------------------------------ k_zbuild
// candidates without a predecessor are passed over by the backtrack (see k_zbuild): plain chaining, one-word anchors (one span), chains of >= 2 anchors
RH_HD inline bool bt_lone_on(const rh_dev_opt &o, const rh_dev_round &rr) { return o.min_cnt >= 2 && rr.afmt.rec8 && !(o.flag & RH_M_RMQ) && !(o.bw_long > o.bw); }
// The backtrack's "used" marks and claim stamps of a read with up to BT_LDS_ANCHORS anchors live in LDS (k_backtrack_spec<BT, LMW > 0>): a bit per anchor, the stamps in a table of
// BT_LDS_CLAIMS (x 2 for 512 threads) words indexed by the anchor's low bits.  lds_cap = that limit, or 0 when the LDS form is off (RH_BT_LDS=0, RH_BT_WAVE=1): k_zbuild zeroes
// the marks and stamps in HBM only for the reads that keep them there.
#ifndef BT_LDS_ANCHORS
#define BT_LDS_ANCHORS 524288   // the largest of three classes (a quarter, a half, all of it: 16 / 32 / 64 KB of marks)
#endif
#ifndef BT_LDS_THREADS
#define BT_LDS_THREADS 512   // a workgroup of the two larger LDS classes: eight wavefronts share a read's marks - the 32 KB class runs 3 workgroups = 24 wavefronts a CU instead of 4 = 16 (measured: 67 -> 59 ms a step; the 16 KB class is better off with 256 threads, 54 against 64 ms)
#endif
#ifndef BT_LDS_CLAIMS
#define BT_LDS_CLAIMS 2048
#endif
inline int bt_lds_min_class() { const char *e = getenv("RH_BT_LDS_MIN_CLASS"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : v > 3 ? 3 : v; }   // test aid: no read takes a narrower class than this (3 = HBM)
inline uint32_t bt_lds_cap() { return (getenv("RH_BT_LDS") && atoi(getenv("RH_BT_LDS")) == 0) || getenv("RH_BT_WAVE") || bt_lds_min_class() == 3 ? 0u : (uint32_t)BT_LDS_ANCHORS; }
__global__ __launch_bounds__(NT) void k_zbuild(rh_dev_opt o, rh_dev_round rr, uint32_t lds_cap)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act) return;
	if (rr.skip[a]) { if (tid == 0) rr.n_z[a] = 0; return; }
	const uint64_t base = rr.a_off[a];
	const int32_t n = (int32_t)(rr.a_off[a + 1] - base);
	const int32_t *fp = (const int32_t*)(rr.ws + base * rr.ws_stride);   // {f, p} interleaved
	rh_mm128_t *z = rr.raw + base;               // the unsorted anchor copy is dead after the anchor sort
	uint64_t *z8 = reinterpret_cast<uint64_t*>(rr.raw) + base;   // rr.z8: 8-byte candidates  score << 32 | anchor
	uint32_t *t4 = (uint32_t*)(rr.ws + base * rr.ws_stride + (size_t)16 * n);   // backtrack's "touched" marks (1 B per anchor)
	uint32_t *claim = (uint32_t*)(rr.ws + base * rr.ws_stride + (size_t)20 * n);   // k_backtrack_spec's per-anchor claim stamps
	if ((uint32_t)n > lds_cap) {
		for (int32_t i = (int32_t)tid; i < (n + 3) / 4; i += NT) t4[i] = 0u;
		for (int32_t i = (int32_t)tid; i < n; i += NT) claim[i] = 0u;
	}
	// A candidate WITHOUT A PREDECESSOR - most anchors of an unmappable read - is a chain of one anchor: with min_num_anchors >= 2
	// mg_chain_backtrack marks it used and drops it (lchain.c:148-170), and nobody ever looks at that mark: an anchor that chains onto another scores
	// more than its own span (mg_lchain_dp starts max_f at the span and takes a predecessor only if that beats it, lchain.c:443, 463, 489), every anchor of the round has the
	// same span (one-word anchors: rr.afmt), so everything whose path could lead here scores more than this candidate and has been processed
	// before it.  It has to be sorted with the others - it is part of the permutation - but the backtrack passes over it: bit 31 of the
	// candidate's anchor word says so.  (Measured and rejected: the weaker, span-independent form "... that is nobody's predecessor either" with a
	// marking pass over the predecessors here - k_zbuild 47 -> 110 ms for 46 ms less in k_backtrack_spec.)
	const bool lone_on = bt_lone_on(o, rr);
	uint32_t nz = 0, n_lone = 0;
	const int2 *fp2 = reinterpret_cast<const int2*>(fp);            // .x = f, .y = p
	constexpr int ZB = 4;                                           // consecutive anchors per thread and step: a quarter of the barriers, 32-byte loads
	for (int32_t i0 = 0; i0 < n; i0 += NT * ZB) {                   // (a workgroup per read: an unmappable read on a large index has 10^5 anchors)
		const int32_t ib = i0 + (int32_t)tid * ZB;
		int2 rec[ZB];
		bool ok[ZB];
		uint32_t cnt = 0;
#pragma unroll
		for (int k = 0; k < ZB; ++k) { rec[k].x = 0; rec[k].y = 0; if (ib + k < n) rec[k] = fp2[ib + k]; ok[k] = ib + k < n && rec[k].x >= o.min_sc; cnt += ok[k] ? 1u : 0u; }
		uint32_t tot;
		uint32_t pos = nz + block_excl_scan(cnt, s_w, tot);
#pragma unroll
		for (int k = 0; k < ZB; ++k) {
			if (!ok[k]) continue;
			const uint32_t lone = (lone_on && rec[k].y < 0) ? 0x80000000u : 0u;
			if (lone) ++n_lone;                                        // (this thread's; summed below)
			if (rr.z8) z8[pos] = (uint64_t)(uint32_t)rec[k].x << 32 | (uint64_t)((uint32_t)(ib + k) | lone);
			else { rh_mm128_t e; e.x = (uint64_t)(int64_t)rec[k].x; e.y = (uint64_t)((uint32_t)(ib + k) | lone); z[pos] = e; }
			++pos;
		}
		nz += tot;
	}
	if (lone_on) { uint32_t lt; (void)block_excl_scan(n_lone, s_w, lt); n_lone = lt; }
	// (they all score their span, less than any candidate with a predecessor: the first n_lone of the sorted candidates - the backtrack stops there;
	// n_v is the backtrack's to write, its input until then)
	if (tid == 0) { rr.n_z[a] = nz; if (lone_on) rr.n_v[a] = n_lone; }
}

// ------------------------------------------------------------------------------------------------ backtrack
// One read per wavefront, 64 candidates at a time - the walk of mg_chain_backtrack (lchain.c:148-170) made parallel without
// changing its result.  A candidate's outcome depends on the candidates before it only through the "used" marks of the
// anchors its own mg_chain_bk_end walk reads (its path; a used anchor that ends the walk stays used and needs no watching), and a candidate only ever marks anchors of
// its own path.  So, per
// batch of 64 candidates (lane 0 = best score), rounds of:
//   1. every pending lane walks its path under the current marks and stamps each anchor of it with (round, priority)
//      through atomicMax;
//   2. a lane whose path carries no stamp of a better lane of this round cannot be influenced by any pending lane (a later
//      re-walk of those only shortens their paths): it commits - marks its chain, fixes score and count;
//   3. the others walk again next round, now seeing the new marks.  The best pending lane always commits.
// When the batch is settled the accepted chains get their slots in u[] / v[] by a prefix sum in candidate order.  The
// dependent loads of 64 walks are in flight together, where the serial walk paid one memory round trip per step.
// BT = 64: one wavefront per read.  BT = 256 (round 5): a WORKGROUP per read, 256 candidates per round - the reads of the late rounds are fewer than the chip has
// wave slots (6 500 unmappable reads of 10^5 candidates each) and a read's batches are a serial chain of ~10 us rounds, so the kernel took as long as
// its slowest read; four wavefronts walk four times the candidates per round (same CU, same L1: the marks stay plain loads and stores).
// LM (round 6): the marks and stamps in LDS for the reads of up to lds_cap anchors - a bit per anchor (atomic OR), the stamps in a table indexed by the anchor's low bits.  Two
// anchors that share a table word can only make a lane see a better lane's stamp where there is none: it walks again next round (the best pending lane holds the highest stamp
// of the round wherever it stamped, so it always commits), the result is the same.  Per chain of two this takes the mark line, the stamp line (an L2 read-modify-write) and their
// write-backs out of the kernel's random HBM lines, which leaves the {f, p} line and the anchors' line.  The reads above lds_cap take the <BT, false> launch.
// LMW = the marks' 32-bit words (0: marks and stamps in HBM); the launch takes the reads of lo_cap < n <= 32 LMW anchors (LMW = 0: n > lo_cap).
template <int BT, int LMW>
__global__ __launch_bounds__(BT) void k_backtrack_spec(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, uint32_t lo_cap)
{
	constexpr bool LM = LMW > 0;
	constexpr int SH = BT > 256 ? 9 : BT > 64 ? 8 : 6;               // stamp = round << SH | priority
	__shared__ uint32_t s_w[BT / 64 + 1];
	__shared__ uint32_t s_bits[LM ? LMW : 1];
	constexpr uint32_t CL = (uint32_t)BT_LDS_CLAIMS * (BT > 256 ? 2u : 1u);
	__shared__ uint32_t s_claim[LM ? CL : 1];
	const uint32_t a = blockIdx.x, lane = threadIdx.x;               // ("lane" = the candidate's place in the batch, 0 = best score)
	if (a >= rr.n_act) return;
	const uint64_t base = rr.a_off[a];
	const int32_t n = (int32_t)(rr.a_off[a + 1] - base);
	if ((lo_cap && (uint32_t)n <= lo_cap) || (LM && (uint32_t)n > 32u * (uint32_t)LMW)) return;   // another launch's read (lo_cap = 0: the narrowest class launched, which also takes the reads without anchors)
	if (rr.skip[a]) { if (lane == 0) { rr.n_u[a] = 0; rr.n_v[a] = 0; } return; }
	const uint32_t r = rr.act[a];
	if (LM) {
		for (uint32_t k = lane; k < ((uint32_t)n + 31u) / 32u; k += BT) s_bits[k] = 0u;
		for (uint32_t k = lane; k < CL; k += BT) s_claim[k] = 0u;
	}
	#define BK_USED(i) (LM ? (uint8_t)((s_bits[(uint32_t)(i) >> 5] >> ((uint32_t)(i) & 31u)) & 1u) : t[(i)])
	#define BK_MARK(i) do { if (LM) atomicOr(&s_bits[(uint32_t)(i) >> 5], 1u << ((uint32_t)(i) & 31u)); else t[(i)] = 1; } while (0)
	#define BK_STAMP(i) do { if (LM) atomicMax(&s_claim[(uint32_t)(i) & (CL - 1u)], stamp); else atomicMax(&claim[(i)], stamp); } while (0)
	const int32_t n_z = (int32_t)rr.n_z[a];
	unsigned char *wsr = rr.ws + base * rr.ws_stride;
	const int2 *fp = (const int2*)wsr;                              // .x = f, .y = p
	// "used" marks, zeroed by k_zbuild: one BYTE per anchor, plain loads and stores through the L1.  Every candidate starts with a look at its own
	// mark (a random access, most of them used already).  Measured alternatives (human-scale step, this kernel 228 ms): marks inside a 16-byte
	// {f, p, claim, used} record 404 ms; one BIT per anchor set with L2 atomics and read at L2 576 ms - the L1 serves most of these looks.
	uint8_t *t = (uint8_t*)(wsr + (size_t)16 * n);
	uint32_t *claim = (uint32_t*)(wsr + (size_t)20 * n);            // stamps, zeroed by k_zbuild
	const rh_mm128_t *zs = rr.zs + base;
	const uint64_t *zs8 = reinterpret_cast<const uint64_t*>(rr.zs) + base;
	uint64_t *u = rr.u + base;
	uint32_t *ck0 = (uint32_t*)(wsr + (size_t)32 * n);               // where each accepted chain starts among the read's chained anchors (compact_a, below)
	const int32_t min_sc = o.min_sc, min_cnt = o.min_cnt, max_drop = o.bw;
	int32_t n_u = 0, n_v = 0;
	uint32_t epoch = 0;
	const int32_t n_lone = bt_lone_on(o, rr) ? (int32_t)rr.n_v[a] : 0;   // candidates without a predecessor: the lowest scores, passed over (k_zbuild)
	__syncthreads();                                                // (n_v[a] is this kernel's to write at the end: every wavefront has read it first)
	for (int32_t kt = n_z; kt > n_lone; kt -= BT) {                // candidates from the best score down (lchain.c:148)
		const int32_t k = kt - 1 - (int32_t)lane;
		const uint32_t w0 = k >= n_lone ? (rr.z8 ? (uint32_t)zs8[k] : (uint32_t)zs[k].y) : 0x80000000u;   // (the first n_lone sorted candidates are not there to be read: the sorter leaves their stretch unwritten, rh_sort_job::dead_cnt)
		const int32_t i0 = (int32_t)(w0 & 0x7FFFFFFFu);
		bool pending = k >= n_lone && !(w0 >> 31), accepted = false;     // (bit 31: a chain of one anchor that nothing else touches - k_zbuild)
		int32_t r_cnt = 0, r_sc = 0;
		int32_t pn1 = 0, pn2 = 0, pn3 = 0;                             // the first anchors of the path after i0 (most chains are this short)
		rh_mm128_t a0{}, a1{}, a2{}, a3{};                             // ... and, once the chain is accepted, the anchors themselves: requested when the lane commits, stored when the batch is settled
		for (;;) {
			uint32_t n_pend;
			if (BT == 64) n_pend = (uint32_t)__popcll(__ballot(pending));
			else (void)block_rank(pending, s_w, n_pend);
			if (!n_pend) break;
			const bool solo = n_pend == 1;                            // a single pending lane has nobody to collide with
			++epoch;
			const uint32_t stamp = epoch << SH | ((uint32_t)BT - 1u - lane);
			bool walked = false;
			int32_t zx = 0, path = 0, max_s = 0, emit = 0;             // path = unused anchors reached after i0; emit = anchors i0 .. before max_i
			if (pending && BK_USED(i0) == 0) {
				walked = true;
				int2 rec = fp[i0];
				zx = rec.x;
				if (!solo) BK_STAMP(i0);
				for (;;) {	// mg_chain_bk_end (lchain.c:47-75): back until a used anchor, the start, or a score drop > max_drop
					const int32_t i = rec.y;
					int32_t sdrop = zx;
					uint8_t ti = 0;
					if (i >= 0) {
						rec = fp[i]; ti = BK_USED(i); sdrop = zx - rec.x;
						if (ti == 0) {	// (a used anchor ends every walk that reaches it: nobody's to take, nothing to stamp)
							if (!solo) BK_STAMP(i);
							++path;
							if (path == 1) pn1 = i; else if (path == 2) pn2 = i; else if (path == 3) pn3 = i;
						}
					}
					if (sdrop > max_s) { max_s = sdrop; emit = (i >= 0 && ti == 0) ? path : path + 1; }   // max_i = i: the chain ends before it
					else if (max_s - sdrop > max_drop) break;
					if (i < 0 || ti != 0) break;
				}
			}
			RH_WG_FENCE();
			__syncthreads();                                          // every stamp of the round is in
			bool conflict = false;
			if (walked && !solo) {	// read at L2, where the stamps were combined; the cached anchors' reads are independent of each other
				#define BK_CLAIM(i) (LM ? s_claim[(uint32_t)(i) & (CL - 1u)] : __hip_atomic_load(&claim[(i)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))   /* HBM form: an L2 read, no read-modify-write */
				const uint32_t c0 = BK_CLAIM(i0);
				const uint32_t c1 = path >= 1 ? BK_CLAIM(pn1) : stamp;
				const uint32_t c2 = path >= 2 ? BK_CLAIM(pn2) : stamp;
				const uint32_t c3 = path >= 3 ? BK_CLAIM(pn3) : stamp;
				conflict = c0 != stamp || c1 != stamp || c2 != stamp || c3 != stamp;
				int32_t x = pn3;
				for (int32_t j = 3; j < path && !conflict; ++j) { x = fp[x].y; if (BK_CLAIM(x) != stamp) conflict = true; }
				#undef BK_CLAIM
			}
			if (pending && !conflict) {
				if (walked) {	// anchors i0 .. (exclusive) max_i form the chain; the marks stay even if it is rejected, as in the reference
					if (emit >= 1) BK_MARK(i0);
					if (emit >= 2) BK_MARK(pn1);
					if (emit >= 3) BK_MARK(pn2);
					if (emit >= 4) BK_MARK(pn3);
					int32_t x = pn3;
					for (int32_t j = 4; j < emit; ++j) { x = fp[x].y; BK_MARK(x); }
					accepted = max_s >= min_sc && emit > 0 && emit >= min_cnt;   // (score of the chain = the best drop seen = max_s)
					r_cnt = emit; r_sc = max_s;
					if (accepted) {	// (the loads travel while the other lanes' rounds go on)
						a0 = rh_an_raw_ld(rr, rr.anc, base + (uint32_t)i0);
						if (emit >= 2) a1 = rh_an_raw_ld(rr, rr.anc, base + (uint32_t)pn1);
						if (emit >= 3) a2 = rh_an_raw_ld(rr, rr.anc, base + (uint32_t)pn2);
						if (emit >= 4) a3 = rh_an_raw_ld(rr, rr.anc, base + (uint32_t)pn3);
					}
				}
				pending = false;
			}
			RH_WG_FENCE();
			__syncthreads();                                          // the new marks are in before anybody walks again (one CU, one L1)
		}
		// slots in candidate order
		uint32_t n_acc, total;
		const uint32_t my_rank = block_rank(accepted, s_w, n_acc);
		const uint32_t my_off = block_excl_scan(accepted ? (uint32_t)r_cnt : 0u, s_w, total);
		if (accepted) {
			// The first half of compact_a (lchain.c:214-243) right here, from the registers that hold the chain (round 6; until then the walk left the members'
			// indices in v[] and a second kernel, k_chain_gather, looked every chain up again): the chain's anchors, reversed into ascending order, go to
			// the carry staging at the chain's offset among the read's chained anchors - what the next chunk carries, in backtrack order -, the offset to
			// ck0[], and the key compact_a orders the chains by (first anchor's x, lchain.c:262-266) to the chain sorter's input.
			const uint32_t ci = (uint32_t)n_u + my_rank, off = (uint32_t)n_v + my_off, last = off + (uint32_t)r_cnt - 1u;
			u[ci] = (uint64_t)(uint32_t)r_sc << 32 | (uint64_t)(uint32_t)r_cnt;
			ck0[ci] = off;
			rh_mm128_t af = a0;                                          // the chain's first anchor = the last of the walk
			rh_an_raw_st(rr, rr.prev_out, base + last, a0);
			if (r_cnt >= 2) { rh_an_raw_st(rr, rr.prev_out, base + last - 1u, a1); af = a1; }
			if (r_cnt >= 3) { rh_an_raw_st(rr, rr.prev_out, base + last - 2u, a2); af = a2; }
			if (r_cnt >= 4) { rh_an_raw_st(rr, rr.prev_out, base + last - 3u, a3); af = a3; }
			int32_t x = pn3;
			for (int32_t j = 4; j < r_cnt; ++j) { x = fp[x].y; af = rh_an_raw_ld(rr, rr.anc, base + (uint32_t)x); rh_an_raw_st(rr, rr.prev_out, base + last - (uint32_t)j, af); }
			const uint64_t x0 = rh_an_raw_x(rr, af);
			if (rr.cfmt.rec8) reinterpret_cast<uint64_t*>(rr.raw)[base + ci] = rh_rec8_pack_key(x0, rr.cfmt.lo, rr.cfmt.mid) << rr.cfmt.shift | (uint64_t)ci;
			else { rh_mm128_t e; e.x = x0; e.y = (uint64_t)off << 32 | (uint64_t)ci; rr.raw[base + ci] = e; }
		}
		n_u += (int32_t)n_acc;
		n_v += (int32_t)total;
	}
	if (lane == 0) {
		rr.n_u[a] = (uint32_t)n_u; rr.n_v[a] = (uint32_t)n_v;
		if (n_u == 0) { rd.n_prev[r] = 0; rd.prev_off[r] = base; }
	}
	#undef BK_USED
	#undef BK_MARK
	#undef BK_STAMP
}

// compact_a (lchain.c:214-281) around the block sorter:
//   k_backtrack_spec (above): a committed chain's start offset, its members reversed into ascending order -> the carry staging (= what
//                    the next chunk carries), its sort key (first-anchor x, start << 32 | chain) -> rr.raw
//   [rhk_sort_job  : chains into the reference's order of their first anchor]
//   k_chain_reorder: destination offsets (scan in sorted order), chains copied back over the anchor slice, u[] permuted
#ifndef CG_CAP
#define CG_CAP 2048
#endif
// sort keys of a read's chains (first-anchor x; lchain.c:262-266) from the gathered chains in the carry staging -> rr.raw
RH_DEV void chain_keys(const rh_dev_round &rr, uint64_t base, uint32_t n_u, const uint32_t *ck0, uint32_t tid)
{
	if (rr.cfmt.rec8) {	// 8-byte keys: first-anchor x, packed, above the chain's number (its start offset stays in ck0)
		uint64_t *w8 = reinterpret_cast<uint64_t*>(rr.raw) + base;
		for (uint32_t i = tid; i < n_u; i += NT) w8[i] = rh_rec8_pack_key(rh_an_ld(rr, rr.prev_out, base + ck0[i]).x, rr.cfmt.lo, rr.cfmt.mid) << rr.cfmt.shift | (uint64_t)i;
	} else {
		rh_mm128_t *w = rr.raw + base;
		for (uint32_t i = tid; i < n_u; i += NT) { rh_mm128_t e; e.x = rh_an_ld(rr, rr.prev_out, base + ck0[i]).x; e.y = (uint64_t)ck0[i] << 32 | (uint64_t)i; w[i] = e; }
	}
}
#define CG_SHORT 8            // anchors a lane copies on its own when a read has more than CG_CAP chains
// the chain-order keys of the reads with skip2[a] == 0 once more, from the gathered chains (the sorter overwrote its input): rhk_backtrack's exact re-run
__global__ __launch_bounds__(NT) void k_chain_keys(rh_dev_round rr, const uint8_t *skip2)
{
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act || rr.skip[a] || skip2[a]) return;
	const uint32_t n_u = rr.n_u[a];
	if (n_u == 0) return;
	const uint64_t base = rr.a_off[a];
	const uint32_t n = (uint32_t)(rr.a_off[a + 1] - base);
	chain_keys(rr, base, n_u, (const uint32_t*)(rr.ws + base * rr.ws_stride + (size_t)32 * n), tid);
}

// rr.lazy_reorder: where sorted chain i starts among the gathered chains of the read (carry staging, rr.prev_out): 4 bytes per chain in the read's scratch, behind
// the offsets / counts of compact_a (ck0 at 32 n, dk at 36 n, u2 at 40 n: chains have >= 2 anchors, so each ends before the next begins); k_regions_prep writes its
// heads below 16 n, the serial region kernels read all of it before their core's arrays (from 32 n on at most) are written
RH_DEV uint32_t *chain_from(const rh_dev_round &rr, uint64_t base, uint32_t n) { return reinterpret_cast<uint32_t*>(rr.ws + base * rr.ws_stride + (size_t)48 * n); }
// first and last anchor of sorted chain i (k = its offset among the read's chained anchors in sorted order, cnt = its length)
RH_DEV void chain_ends(const rh_dev_round &rr, uint64_t base, uint32_t n, uint32_t i, uint32_t k, uint32_t cnt, rh_mm128_t &f0, rh_mm128_t &f1)
{
	if (rr.lazy_reorder) { const uint32_t fr = chain_from(rr, base, n)[i]; f0 = rh_an_ld(rr, rr.prev_out, base + fr); f1 = rh_an_ld(rr, rr.prev_out, base + fr + cnt - 1); }
	else { f0 = rh_an_ld(rr, rr.anc, base + k); f1 = rh_an_ld(rr, rr.anc, base + k + cnt - 1); }
}

__global__ __launch_bounds__(NT) void k_chain_reorder(rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ uint32_t s_w[NT / 64];
	__shared__ uint32_t s_off[CG_CAP];
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const uint32_t n_u = rr.n_u[a], n_v = rr.n_v[a];
	if (n_u == 0) { if (tid == 0) { rd.n_prev[r] = 0; rd.prev_off[r] = base; } return; }
	const uint32_t n = (uint32_t)(rr.a_off[a + 1] - base);
	unsigned char *wsr = rr.ws + base * rr.ws_stride;
	uint32_t *dk = (uint32_t*)(wsr + (size_t)36 * n);                // destination offsets in sorted order
	uint64_t *u2 = (uint64_t*)(wsr + (size_t)40 * n);
	uint64_t *u = rr.u + base;
	const rh_mm128_t *w = rr.zs + base;                            // w: sorted keys
	const uint64_t *w8 = reinterpret_cast<const uint64_t*>(rr.zs) + base;
	const bool c8 = rr.cfmt.rec8 != 0;
	const uint64_t cmask = (1ull << rr.cfmt.shift) - 1ull;
	const uint32_t *ck0 = (const uint32_t*)(wsr + (size_t)32 * n);   // start offsets in backtrack order (k_backtrack_spec)
	#define CR_CHAIN(i_) (c8 ? (uint32_t)(w8[(i_)] & cmask) : (uint32_t)w[(i_)].y)
	#define CR_FROM(i_) (c8 ? ck0[(uint32_t)(w8[(i_)] & cmask)] : (uint32_t)(w[(i_)].y >> 32))
	uint32_t run = 0;
	constexpr uint32_t RU = 4;                                      // (round 6) four sorted chains a thread and step, NT apart: their random looks at u[] in flight together
	for (uint32_t i0 = 0; i0 < n_u; i0 += NT * RU) {
		uint32_t cv[RU];
		uint64_t uq[RU];
#pragma unroll
		for (uint32_t q = 0; q < RU; ++q) { const uint32_t i = i0 + q * NT + tid; cv[q] = i < n_u ? CR_CHAIN(i) : 0u; }
#pragma unroll
		for (uint32_t q = 0; q < RU; ++q) { const uint32_t i = i0 + q * NT + tid; uq[q] = i < n_u ? u[cv[q]] : 0ull; }
#pragma unroll
		for (uint32_t q = 0; q < RU; ++q) {
			const uint32_t i = i0 + q * NT + tid;
			uint32_t tot;
			const uint32_t ex = block_excl_scan((uint32_t)uq[q], s_w, tot);
			if (i < n_u) { u2[i] = uq[q]; dk[i] = run + ex; }
			run += tot;
		}
	}
	__syncthreads();
	const uint32_t *tab = dk;
	if (rr.lazy_reorder) {	// no copy: the region stage finds the chains where they were gathered
		uint32_t *fk = chain_from(rr, base, n);
		for (uint32_t i = tid; i < n_u; i += NT) fk[i] = CR_FROM(i);
	} else {
	if (n_u <= CG_CAP) { for (uint32_t i = tid; i < n_u; i += NT) s_off[i] = dk[i]; tab = s_off; __syncthreads(); }
	if (n_u <= CG_CAP) {
		for (uint32_t q = tid; q < n_v; q += NT) {
			uint32_t lo = 0, hi = n_u;
			while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tab[mid] <= q) lo = mid; else hi = mid; }
			rh_an_cp(rr, rr.anc, base + q, rr.prev_out, base + CR_FROM(lo) + (q - tab[lo]));
		}
	} else {	// thousands of chains (an unmappable read on a large index: ~2 anchors per chain): one lane per chain copies its few anchors - neighbouring lanes write neighbouring slots and nobody searches; a long chain is the whole wavefront's
		const uint32_t l = lane_id();
		for (uint32_t i0 = wave_id() * 64u; i0 < n_u; i0 += NT) {
			const uint32_t i = i0 + l;
			uint32_t from = 0, to = 0, ni = 0;
			if (i < n_u) { from = CR_FROM(i); to = dk[i]; ni = (uint32_t)u2[i]; }
			if (ni <= CG_SHORT) for (uint32_t j = 0; j < ni; ++j) rh_an_cp(rr, rr.anc, base + to + j, rr.prev_out, base + from + j);
			uint64_t longm = __ballot(ni > CG_SHORT);
			while (longm) {
				const int src = __ffsll((unsigned long long)longm) - 1;
				longm &= longm - 1;
				const uint32_t ff = __shfl(from, src), tt = __shfl(to, src), nn = __shfl(ni, src);
				for (uint32_t j = l; j < nn; j += 64) rh_an_cp(rr, rr.anc, base + tt + j, rr.prev_out, base + ff + j);
			}
		}
	}
	}
	#undef CR_CHAIN
	#undef CR_FROM
	for (uint32_t i = tid; i < n_u; i += NT) u[i] = u2[i];
	if (tid == 0) {
		rd.n_prev[r] = n_v; rd.prev_off[r] = base;
		atomicAdd((unsigned long long*)&rr.counters[4], (unsigned long long)n_v);
	}
}

// ------------------------------------------------------------------------------------------------ regions
struct rh_reg {
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, n_sub, score0;
	uint32_t mapq, rev, hash;
};
struct rh_chain_head { uint64_t x0, y0; int32_t x1, y1, cnt, k; };   // first anchor, low words of the last anchor
// A region sort record's payload (k_regions_prep): what mm_set_parent looks at of a chain - query interval and anchor count - travels with the key,
//   y = chain (26 bits) | cnt (6) << 26 | qs (16) << 32 | qe (16) << 48,
// so the region kernels stream the sorted keys instead of gathering a 32-byte head per chain (k_regions_batch: 359 GB of 64-byte sectors a step for
// 16 bytes each).  cnt = 63: something does not fit (>= 63 anchors, a coordinate >= 65536): look at the head after all.
#define RG_IDX(y_) ((uint32_t)(y_) & 0x3FFFFFFu)
RH_DEV uint64_t rg_pack(uint32_t chain, uint32_t cnt, int64_t qs, int64_t qe)
{
	if (cnt >= 63u || qs < 0 || qe < 0 || qs >= 65536 || qe >= 65536) return (uint64_t)chain | 63ull << 26;
	return (uint64_t)chain | (uint64_t)cnt << 26 | (uint64_t)qs << 32 | (uint64_t)qe << 48;
}
RH_DEV void rg_unpack(uint64_t y, const rh_chain_head *heads, int32_t &qs, int32_t &qe, int32_t &cnt)
{
	const uint32_t c6 = (uint32_t)(y >> 26) & 63u;
	if (c6 != 63u) { cnt = (int32_t)c6; qs = (int32_t)((y >> 32) & 0xFFFFu); qe = (int32_t)(y >> 48); }
	else { const rh_chain_head *h = heads + RG_IDX(y); cnt = h->cnt; qs = (int32_t)h->y0; qe = h->y1 + 1; }
}

// Stage-level export of kept region k of a read (rh_regions_batch; rr.reg_out is null on the mapping path, which needs creg[0], the count and two sums only):
// the reference's region record in the field order of its dump, {id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0, mapq, rev, hash};
// mlen / blen (mm_cal_fuzzy_len hit.c:10-29) are read by nothing on the path and are therefore computed here, from the chain's anchors
RH_DEV void reg_export(const rh_dev_round &rr, uint64_t base, int32_t k, const rh_reg &q)
{
	int32_t mlen = 0, blen = 0;
	if (q.cnt > 0) {
		rh_mm128_t p = rh_an_ld(rr, rr.anc, base + (uint32_t)q.as);
		mlen = blen = (int32_t)((p.y >> 32) & 63u);
		for (int32_t i = 1; i < q.cnt; ++i) {
			const rh_mm128_t c = rh_an_ld(rr, rr.anc, base + (uint32_t)q.as + (uint32_t)i);
			const int32_t span = (int32_t)((c.y >> 32) & 63u), tl = (int32_t)c.x - (int32_t)p.x, ql = (int32_t)c.y - (int32_t)p.y;
			blen += tl > ql ? tl : ql;
			mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
			mlen += tl < ql ? tl : ql;
			p = c;
		}
	}
	int32_t *w = rr.reg_out + (base + (uint64_t)(uint32_t)k) * 18u;
	w[0] = q.id; w[1] = q.cnt; w[2] = q.rid; w[3] = q.score; w[4] = q.qs; w[5] = q.qe; w[6] = q.rs; w[7] = q.re; w[8] = q.parent; w[9] = q.subsc;
	w[10] = q.as; w[11] = mlen; w[12] = blen; w[13] = q.n_sub; w[14] = q.score0; w[15] = (int32_t)q.mapq; w[16] = (int32_t)q.rev; w[17] = (int32_t)q.hash;
}
// ... of a primary of the kernels that keep primaries only (best_n = 0): sorted region `i` (descending score) that became kept region k
RH_DEV void reg_export_primary(const rh_dev_round &rr, uint64_t base, int32_t n_u, int32_t k, int32_t i, int32_t subsc, int32_t n_sub, int32_t mapq)
{
	const rh_mm128_t zi = (rr.zs + base)[n_u - 1 - i];
	const rh_chain_head h = ((const rh_chain_head*)(rr.ws + base * rr.ws_stride))[RG_IDX(zi.y)];
	rh_reg q;
	q.id = k; q.parent = k; q.cnt = h.cnt; q.as = h.k; q.score = q.score0 = (int32_t)(zi.x >> 32); q.hash = (uint32_t)zi.x;
	q.rev = (uint32_t)(h.x0 >> 63); q.rid = (int32_t)(h.x0 << 1 >> 33); q.rs = (int32_t)h.x0; q.re = h.x1 + 1; q.qs = (int32_t)h.y0; q.qe = h.y1 + 1;
	q.subsc = subsc; q.n_sub = n_sub; q.mapq = (uint32_t)mapq;
	reg_export(rr, base, k, q);
}

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

// Serial core on whatever memory the arrays live in.  Returns the number of regions kept; *stop = mapping decision.
RH_DEV int32_t regions_core(const rh_dev_opt &o, int32_t n_u, const uint64_t *u, const rh_chain_head *ch, rh_reg *rg, rh_mm128_t *z, uint64_t *cov,
                            int32_t *w, int32_t *tmp, uint32_t *cw, int32_t rep_len, uint32_t n_events, uint32_t offset, const float *logf_tab, int *stop, bool before_mapq = false)
{
	int32_t n_regs = n_u;
	*stop = 0;
	uint32_t hash = 0;
	hash ^= rh_wang32(offset + n_events) + rh_wang32(11u);         // rmap.cpp:346-348
	hash = rh_wang32(hash);
	// regions from chains, ordered by (score, hash of first anchor) descending
	for (int32_t i = 0; i < n_u; ++i) {
		const uint32_t h = (uint32_t)rh_mix64_nomask((rh_mix64_nomask(ch[i].x0) + rh_mix64_nomask(ch[i].y0)) ^ (uint64_t)hash);
		z[i].x = u[i] ^ (uint64_t)h;
		z[i].y = (uint64_t)(uint32_t)i;
	}
	rh_radix_sort_128x(z, (uint32_t)n_u, cw);
	for (int32_t i = 0; i < n_u >> 1; ++i) { const rh_mm128_t tt = z[i]; z[i] = z[n_u - 1 - i]; z[n_u - 1 - i] = tt; }
	for (int32_t i = 0; i < n_u; ++i) {
		const rh_chain_head &c = ch[(uint32_t)z[i].y];
		rh_reg q;
		q.id = i; q.parent = -1; q.subsc = 0; q.n_sub = 0;
		q.score = q.score0 = (int32_t)(z[i].x >> 32);
		q.hash = (uint32_t)z[i].x;
		q.cnt = c.cnt; q.as = c.k;
		q.rev = (uint32_t)(c.x0 >> 63);
		q.rid = (int32_t)(c.x0 << 1 >> 33);
		q.rs = (int32_t)c.x0; q.re = c.x1 + 1;
		q.qs = (int32_t)c.y0; q.qe = c.y1 + 1;
		q.mapq = 0;
		rg[i] = q;
	}
	// primary / secondary by query overlap
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
					for (int32_t x1 = 1; x1 < n_cov; ++x1) {
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
	// drop secondaries (mm_select_sub, check_strand = 1)
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
	if (before_mapq) return n_regs;                                 // (DTW re-scoring: alignment scores first, MAPQ and decision on the host)
	// MAPQ
	{
		int64_t sum_sc = 0;
		for (int32_t i = 0; i < n_regs; ++i) if (rg[i].parent == rg[i].id) sum_sc += rg[i].score;
		const float uniq_ratio = (float)sum_sc / (float)(sum_sc + rep_len);
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
	// mapping decision (non-overlap mode: only chain 0 can be reported)
	if (n_regs == 1 && (int32_t)rg[0].mapq >= o.min_mapq) *stop = 1;
	else if (n_regs >= 1) {
		float meanC = 0, meanQ = 0;
		for (int32_t i = 0; i < n_regs; ++i) { meanC += (float)rg[i].score; meanQ += (float)rg[i].mapq; }
		meanC /= (float)n_regs; meanQ /= (float)n_regs;
		const float bestQ = (float)rg[0].mapq, bestC = (float)rg[0].score;
		float r_bestq = (bestQ > 0) ? (bestQ / 30.0f) : 0.0f; if (r_bestq > 1) r_bestq = 1.0f;
		float r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
		float r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
		const float weighted = o.w_bestq * r_bestq + o.w_bestmq * r_bestmq + o.w_bestmc * r_bestmc;
		if (weighted >= o.w_threshold) *stop = 1;
	}
	return n_regs;
}

RH_DEV void regions_commit(const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &rr, uint32_t a, uint32_t r, int32_t n_regs, const rh_reg *best, int stop)
{
	rd.ls_ncregs[r] = n_regs;
	if (n_regs > 0) {
		rd.ls_cnt[r] = best->cnt; rd.ls_score[r] = best->score; rd.ls_mapq[r] = (int32_t)best->mapq;
		rd.ls_qs[r] = best->qs; rd.ls_qe[r] = best->qe; rd.ls_rs[r] = best->rs; rd.ls_re[r] = best->re;
		rd.ls_rid[r] = best->rid; rd.ls_rev[r] = (int32_t)best->rev;
	}
	rd.ev_off[r] = rd.ev_off[r] + rr.n_ev[a];                    // reg->offset += n_events (rmap.cpp:386)
	if (stop) { rd.done[r] = 1; rd.stop_chunk[r] = rr.chunk; }
}

// All-vs-all (RI_M_ALL_CHAINS, rmap.cpp:421-500): a read reports one chain with enough MAPQ, or else every chain whose score
// reaches min_chaining_score2.  The reported chains leave as pairs of 16-byte words in the read's slot of the carry staging
// (a whole-read round carries no anchors anywhere), so that the carry compaction gathers them densely for k_finalize_ava:
//   x0 = rs << 32 | rid, y0 = qs << 32 | re, x1 = score << 32 | qe, y1 = rev << 40 | mapq << 32 | cnt
RH_DEV void regions_commit_ava(const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &rr, uint32_t a, uint32_t r, int32_t n_regs, const rh_reg *rg)
{
	rh_mm128_t *out = rr.prev_out + rr.a_off[a];
	uint32_t nm = 0;
	const bool single = n_regs == 1 && (int32_t)rg[0].mapq >= o.min_mapq;
	for (int32_t i = 0; i < n_regs; ++i) {
		const rh_reg &q = rg[i];
		if (!(single || q.score >= o.min_sc2)) continue;
		rh_mm128_t w0, w1;
		w0.x = (uint64_t)(uint32_t)q.rs << 32 | (uint32_t)q.rid; w0.y = (uint64_t)(uint32_t)q.qs << 32 | (uint32_t)q.re;
		w1.x = (uint64_t)(uint32_t)q.score << 32 | (uint32_t)q.qe; w1.y = (uint64_t)(q.rev & 1u) << 40 | (uint64_t)(q.mapq & 0xFFu) << 32 | (uint32_t)q.cnt;
		out[2 * nm] = w0; out[2 * nm + 1] = w1;
		++nm;
	}
	rd.n_prev[r] = 2 * nm; rd.prev_off[r] = rr.a_off[a];
	if (nm) { rd.done[r] = 1; rd.stop_chunk[r] = rr.chunk; }
}

#ifndef RG_CAP
#define RG_CAP 512
#endif
#ifndef RG_SMALL
#define RG_SMALL 8      // up to this many chains: one read per lane
#endif
#ifndef RGW_CAP
#define RGW_CAP 1536   // chains the wave-cooperative kernel holds in LDS (large class)
#endif
#ifndef RGW_CAP0
#define RGW_CAP0 256   // ... small class
#endif

struct rg_lds {
	rh_reg rg[RG_CAP];
	rh_chain_head ch[RG_CAP];
	rh_mm128_t z[RG_CAP];
	uint64_t cov[RG_CAP], u[RG_CAP];
	int32_t w[RG_CAP], tmp[RG_CAP];
	uint32_t cw[512];
	uint32_t k0[RG_CAP];
};

__global__ __launch_bounds__(64) void k_regions(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, const float *logf_tab, uint32_t n_lo, int only_flagged)
{
	__shared__ rg_lds L;
	const uint32_t a = blockIdx.x, lane = threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t r = rr.act[a];
	if (rr.skip[a]) { if (lane == 0) { rd.ls_ncregs[r] = 0; if (o.flag & RH_M_ALL_CHAINS) rd.n_prev[r] = 0; } return; }   // chunk dropped: creg stays NULL (rmap.cpp:232-235, :419)
	const uint64_t base = rr.a_off[a];
	const int32_t n_u = (int32_t)rr.n_u[a];
	if (n_u == 0) { if (lane == 0) { regions_commit(o, rd, rr, a, r, 0, nullptr, 0); if (o.flag & RH_M_ALL_CHAINS) rd.n_prev[r] = 0; } return; }
	if (n_u > RG_CAP || n_u <= (int32_t)n_lo) return;             // others: k_regions_big
	if (only_flagged && !rr.need_exact[a]) return;                // done by k_regions_wave
	const rh_mm128_t *an = rr.anc + base;
	const uint64_t *u = rr.u + base;
	for (int32_t i = (int32_t)lane; i < n_u; i += 64) L.u[i] = u[i];
	__syncthreads();
	if (lane == 0) { uint32_t k = 0; for (int32_t i = 0; i < n_u; ++i) { L.k0[i] = k; k += (uint32_t)L.u[i]; } }
	__syncthreads();
	const uint32_t n_an = (uint32_t)(rr.a_off[a + 1] - base);
	// (rr.lazy_reorder: the gathered chains of a read with a long chain list may have served the region sort as scratch by now - k_regions_prep took its heads before)
	const bool from_prep = rr.lazy_reorder && n_u > RG_SMALL;
	for (int32_t i = (int32_t)lane; i < n_u; i += 64) {
		const uint32_t k = L.k0[i], cnt = (uint32_t)L.u[i];
		if (from_prep) { L.ch[i] = ((const rh_chain_head*)(rr.ws + base * rr.ws_stride))[i]; continue; }
		rh_mm128_t f0, f1;
		chain_ends(rr, base, n_an, (uint32_t)i, k, cnt, f0, f1);
		rh_chain_head h; h.x0 = f0.x; h.y0 = f0.y; h.x1 = (int32_t)f1.x; h.y1 = (int32_t)f1.y; h.cnt = (int32_t)cnt; h.k = (int32_t)k;
		L.ch[i] = h;
	}
	__syncthreads();
	if (lane == 0) {
		int stop;
		const int32_t n_regs = regions_core(o, n_u, L.u, L.ch, L.rg, L.z, L.cov, L.w, L.tmp, L.cw, rr.rep_len[a], rr.n_ev[a], rd.ev_off[r], logf_tab, &stop);
		regions_commit(o, rd, rr, a, r, n_regs, &L.rg[0], (o.flag & RH_M_ALL_CHAINS) ? 0 : stop);
		if (rr.reg_out) for (int32_t i = 0; i < n_regs; ++i) reg_export(rr, base, i, L.rg[i]);
		if (o.flag & RH_M_ALL_CHAINS) regions_commit_ava(o, rd, rr, a, r, n_regs, L.rg);
	}
}

// reads with more than RG_CAP chains: the same core on HBM scratch, one read per lane
__global__ void k_regions_big(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, const float *logf_tab, uint32_t n_lo, uint32_t n_hi, int only_flagged)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const int32_t n_u = (int32_t)rr.n_u[a];
	if (n_u <= (int32_t)n_lo || n_u > (int32_t)n_hi) return;
	if (only_flagged && !rr.need_exact[a]) return;
	const rh_mm128_t *an = rr.anc + base;
	const uint64_t *u = rr.u + base;
	unsigned char *wsr = rr.ws + base * rr.ws_stride;           // 64 B per anchor >= 128 B per chain (min_cnt >= 2)
	rh_reg *rg = (rh_reg*)wsr;
	rh_chain_head *ch = (rh_chain_head*)(wsr + (size_t)64 * n_u);
	rh_mm128_t *z = (rh_mm128_t*)(wsr + (size_t)96 * n_u);
	uint64_t *cov = (uint64_t*)(wsr + (size_t)112 * n_u);
	int32_t *w = (int32_t*)(wsr + (size_t)120 * n_u), *tmp = (int32_t*)(wsr + (size_t)124 * n_u);
	uint32_t *cw = (uint32_t*)rg;                                    // the sort runs before rg[] is populated
	uint32_t k = 0;
	const uint32_t n_an = (uint32_t)(rr.a_off[a + 1] - base);
	const bool from_prep = rr.lazy_reorder && n_u > RG_SMALL;          // (see k_regions; the heads lie where rg[] goes: copied first)
	for (int32_t i = 0; i < n_u; ++i) {
		const uint32_t cnt = (uint32_t)u[i];
		if (from_prep) { rh_chain_head h = ((const rh_chain_head*)wsr)[i]; ch[i] = h; }
		else {
			rh_mm128_t f0, f1;
			chain_ends(rr, base, n_an, (uint32_t)i, k, cnt, f0, f1);
			rh_chain_head h; h.x0 = f0.x; h.y0 = f0.y; h.x1 = (int32_t)f1.x; h.y1 = (int32_t)f1.y; h.cnt = (int32_t)cnt; h.k = (int32_t)k;
			ch[i] = h;
		}
		k += cnt;
	}
	int stop;
	const int32_t n_regs = regions_core(o, n_u, u, ch, rg, z, cov, w, tmp, cw, rr.rep_len[a], rr.n_ev[a], rd.ev_off[r], logf_tab, &stop);
	regions_commit(o, rd, rr, a, r, n_regs, &rg[0], (o.flag & RH_M_ALL_CHAINS) ? 0 : stop);
	if (rr.reg_out) for (int32_t i = 0; i < n_regs; ++i) reg_export(rr, base, i, rg[i]);
	if (o.flag & RH_M_ALL_CHAINS) regions_commit_ava(o, rd, rr, a, r, n_regs, rg);
}

// ------------------------------------------------------------------------------------------------ regions, wave-cooperative
// Reads with many chains (unmapped reads accumulate hundreds of short chains over the chunks): mm_set_parent is
// quadratic in the number of chains, but each of its inner loops over the current primaries is an independent
// interval test -> 64 primaries per step.  Default selection only (pri_ratio > 0, best_n == 0: secondaries dropped).
#define RGW_COV 512

// chain heads + sort keys (hit.c:111-120) for reads with more than RG_SMALL chains; heads -> scratch, keys -> rr.raw
__global__ __launch_bounds__(NT) void k_regions_prep(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, const uint8_t *skip2)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint32_t a = blockIdx.x, lane = threadIdx.x;               // (a workgroup per read: unmappable reads have tens of thousands of chains)
	if (a >= rr.n_act || rr.skip[a] || (skip2 && skip2[a])) return;
	const int32_t n_u = (int32_t)rr.n_u[a];
	if (n_u <= RG_SMALL) return;
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const rh_mm128_t *an = rr.anc + base;
	const uint64_t *u = rr.u + base;
	rh_chain_head *heads = (rh_chain_head*)(rr.ws + base * rr.ws_stride);
	rh_mm128_t *z = rr.raw + base;
	uint32_t hash = 0;
	hash ^= rh_wang32(rd.ev_off[r] + rr.n_ev[a]) + rh_wang32(11u);
	hash = rh_wang32(hash);
	uint32_t carry = 0;
	// (round 6: four chains a thread and step, NT apart - every load of the four still coalesced; the two random looks at a chain's ends are in flight for four chains at once.
	// Four CONSECUTIVE chains a thread - a quarter of the scans - made the kernel twice as slow: loads and the 32-byte head stores four lines wide)
	constexpr int PU = 4;
	const uint32_t n_seg = (uint32_t)(rr.a_off[a + 1] - base);
	for (int32_t i0 = 0; i0 < n_u; i0 += NT * PU) {
		uint64_t uv[PU]; uint32_t kk[PU];
#pragma unroll
		for (int q = 0; q < PU; ++q) { const int32_t i = i0 + q * NT + (int32_t)lane; uv[q] = i < n_u ? u[i] : 0ull; }
#pragma unroll
		for (int q = 0; q < PU; ++q) { uint32_t tot; kk[q] = carry + block_excl_scan((uint32_t)uv[q], s_w, tot); carry += tot; }
		rh_mm128_t f0[PU], f1[PU];
#pragma unroll
		for (int q = 0; q < PU; ++q) {
			const int32_t i = i0 + q * NT + (int32_t)lane;
			f0[q].x = 0; f0[q].y = 0; f1[q].x = 0; f1[q].y = 0;
			if (i >= n_u) continue;
			if (skip2 && rr.lazy_reorder) {	// the keys once more (exact re-run): the heads are there - and the gathered chains may have been the first sort's scratch
				const rh_chain_head h0 = heads[i];
				f0[q].x = h0.x0; f0[q].y = h0.y0; f1[q].x = (uint64_t)(uint32_t)h0.x1; f1[q].y = (uint64_t)(uint32_t)h0.y1;
			} else chain_ends(rr, base, n_seg, (uint32_t)i, kk[q], (uint32_t)uv[q], f0[q], f1[q]);
		}
#pragma unroll
		for (int q = 0; q < PU; ++q) {
			const int32_t i = i0 + q * NT + (int32_t)lane;
			if (i >= n_u) continue;
			const uint32_t cnt = (uint32_t)uv[q], k = kk[q];
			rh_chain_head h; h.x0 = f0[q].x; h.y0 = f0[q].y; h.x1 = (int32_t)f1[q].x; h.y1 = (int32_t)f1[q].y; h.cnt = (int32_t)cnt; h.k = (int32_t)k;
			heads[i] = h;
			const uint32_t hh = (uint32_t)rh_mix64_nomask((rh_mix64_nomask(f0[q].x) + rh_mix64_nomask(f0[q].y)) ^ (uint64_t)hash);
			rh_mm128_t e; e.x = uv[q] ^ (uint64_t)hh; e.y = rg_pack((uint32_t)i, cnt, (int64_t)(int32_t)f0[q].y, (int64_t)(int32_t)f1[q].y + 1);
			z[i] = e;
		}
	}
}

template <int CAP>
struct rgw_lds {
	int32_t qs[CAP], qe[CAP];                    // per region (sorted order): query interval
	uint32_t sc[CAP];                            // per region: score | cnt << 20
	int32_t pqs[CAP], pqe[CAP], psub[CAP], pns[CAP];   // per primary (dense, list order): interval, subsc, n_sub
	uint16_t w[CAP];                             // per primary: region index
	uint64_t cov[RGW_COV < CAP ? RGW_COV : CAP];
	uint16_t covj[RGW_COV < CAP ? RGW_COV : CAP];
	int32_t bc[4];
};

// size classes: (n_lo, CAP] chains per read; the small class keeps many reads per CU in flight
template <int CAP>
__global__ __launch_bounds__(64) void k_regions_wave(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, const float *logf_tab, uint32_t n_lo)
{
	__shared__ rgw_lds<CAP> L;
	constexpr int RGW_COVC = RGW_COV < CAP ? RGW_COV : CAP;
	const uint32_t a = blockIdx.x, lane = threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const int32_t n_u = (int32_t)rr.n_u[a];
	if (n_u <= (int32_t)n_lo || n_u > CAP) return;
	if (!rr.need_exact[a]) return;                                   // done by k_regions_reg
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const rh_chain_head *heads = (const rh_chain_head*)(rr.ws + base * rr.ws_stride);
	const rh_mm128_t *zs = rr.zs + base;                             // keys in radix_sort_128x order (ascending)
	bool unfit = false;
	KPROF_DECL;
	for (int32_t i = (int32_t)lane; i < n_u; i += 64) {
		const rh_mm128_t zi = zs[n_u - 1 - i];                      // descending: larger score first (hit.c:124-126)
		int32_t hqs, hqe, hcnt;
		rg_unpack(zi.y, heads, hqs, hqe, hcnt);
		const uint32_t score = (uint32_t)(zi.x >> 32);
		if (score >= (1u << 20) || (uint32_t)hcnt >= (1u << 12)) unfit = true;
		L.sc[i] = score | (uint32_t)hcnt << 20; L.qs[i] = hqs; L.qe[i] = hqe;
	}
	if (__ballot(unfit)) { if (lane == 0) rr.need_exact[a] = 1; return; }   // does not fit the packed layout: serial kernel
	__syncthreads();
	if (lane == 0) { L.w[0] = 0; L.pqs[0] = L.qs[0]; L.pqe[0] = L.qe[0]; L.psub[0] = 0; L.pns[0] = 0; }
	__syncthreads();
	const bool hard = (o.flag & RH_M_HARD_MLEVEL) != 0;
	KPROF(1);
	int32_t kk = 1;
	bool overflow = false;
	for (int32_t i = 1; i < n_u && !overflow; ++i) {
		const int32_t si = L.qs[i], ei = L.qe[i];
		// overlapping primaries, in list order
		int32_t n_cov = 0;
		for (int32_t j0 = 0; j0 < kk; j0 += 64) {
			const int32_t j = j0 + (int32_t)lane;
			bool ov = false; int32_t sj = 0, ej = 0;
			if (j < kk) { sj = L.pqs[j]; ej = L.pqe[j]; ov = !(ej <= si || sj >= ei); }
			const uint64_t m = __ballot(ov);
			if (ov) {
				const int32_t c = n_cov + (int32_t)lanes_below(m);
				if (c < RGW_COVC) { L.cov[c] = (uint64_t)(uint32_t)(sj < si ? si : sj) << 32 | (uint64_t)(uint32_t)(ej > ei ? ei : ej); L.covj[c] = (uint16_t)j; }
			}
			n_cov += (int32_t)__popcll(m);
		}
		if (n_cov > RGW_COVC) { overflow = true; break; }
		KPROF(2); KPROF_ADD(10, n_cov); KPROF_ADD(11, 1); KPROF_ADD(12, kk);
		int32_t sel = -1, uncov = 0;
		if (n_cov > 0) {
			__syncthreads();
			if (!hard) {	// length of [si, ei) not covered by the overlapping primaries
				if (lane == 0) {
					for (int32_t x1 = 1; x1 < n_cov; ++x1) { const uint64_t cv = L.cov[x1]; int32_t y1 = x1; while (y1 > 0 && L.cov[y1 - 1] > cv) { L.cov[y1] = L.cov[y1 - 1]; --y1; } L.cov[y1] = cv; }
					int32_t x = si, un = 0;
					for (int32_t j = 0; j < n_cov; ++j) { const uint64_t cv = L.cov[j]; if ((int32_t)(cv >> 32) > x) un += (int32_t)(cv >> 32) - x; x = (int32_t)cv > x ? (int32_t)cv : x; }
					if (ei > x) un += ei - x;
					L.bc[0] = un;
				}
				__syncthreads();
				uncov = L.bc[0];
			}
			KPROF(3);
			// first overlapping primary (list order) that masks region i
			for (int32_t c0 = 0; c0 < n_cov && sel < 0; c0 += 64) {
				const int32_t c = c0 + (int32_t)lane;
				bool hit = false;
				if (c < n_cov) {
					const uint32_t j = L.covj[c];
					const int32_t sj = L.pqs[j], ej = L.pqe[j];
					const int32_t mn = ej - sj < ei - si ? ej - sj : ei - si;
					const int32_t mx = ej - sj > ei - si ? ej - sj : ei - si;
					const int32_t ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
					hit = (float)ol / (float)mn - (float)uncov / (float)mx > o.mask_level && uncov <= o.mask_len;
				}
				const uint64_t m = __ballot(hit);
				if (m) sel = (int32_t)L.covj[c0 + __builtin_ctzll(m)];
			}
			KPROF(4);
		}
		if (lane == 0) {
			if (sel >= 0) {
				const int32_t sci = (int32_t)(L.sc[i] & 0xFFFFFu);
				if (L.psub[sel] < sci) L.psub[sel] = sci;
				if ((L.sc[i] >> 20) >= (L.sc[L.w[sel]] >> 20)) ++L.pns[sel];
			} else { L.w[kk] = (uint16_t)i; L.pqs[kk] = si; L.pqe[kk] = ei; L.psub[kk] = 0; L.pns[kk] = 0; }
		}
		if (sel < 0) ++kk;
		__syncthreads();
		KPROF(5);
	}
	if (overflow) { if (lane == 0) rr.need_exact[a] = 1; return; }   // re-done by the serial kernel
	// secondaries dropped (mm_select_sub with best_n = 0): the kept regions are exactly the primaries, in order
	const int32_t n_regs = kk;
	int64_t sum_sc = 0;
	for (int32_t k0 = 0; k0 < n_regs; k0 += 64) { const int32_t k = k0 + (int32_t)lane; int32_t v = k < n_regs ? (int32_t)(L.sc[L.w[k]] & 0xFFFFFu) : 0; for (int d = 32; d > 0; d >>= 1) v += (int32_t)__shfl_xor((uint32_t)v, d); sum_sc += v; }
	const float uniq_ratio = (float)sum_sc / (float)(sum_sc + rr.rep_len[a]);
	int64_t sumQ = 0;
	int32_t mapq0 = 0;
	for (int32_t k0 = 0; k0 < n_regs; k0 += 64) {
		const int32_t k = k0 + (int32_t)lane;
		int32_t mq = 0;
		if (k < n_regs) {
			const uint32_t pk = L.sc[L.w[k]];
			const int32_t sc = (int32_t)(pk & 0xFFFFFu), cn = (int32_t)(pk >> 20);
			const float pen_s1 = (float)((sc > 100 ? 1.0 : 0.01 * (double)sc) * (double)uniq_ratio);
			float pen_cm = cn > 10 ? 1.0f : 0.1f * (float)cn;
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			const int32_t subsc = L.psub[k] > o.min_sc ? L.psub[k] : o.min_sc;
			const float x = (float)subsc / (float)sc;
			mq = (int32_t)(pen_cm * 40.0f * (1.0f - x) * logf_int(sc, logf_tab));
			mq -= (int32_t)(4.343f * logf_int(L.pns[k] + 1, logf_tab) + .499f);
			mq = mq > 0 ? mq : 0;
			mq = mq < 60 ? mq : 60;
			if (rr.reg_out) reg_export_primary(rr, base, n_u, k, (int32_t)L.w[k], L.psub[k], L.pns[k], mq);
		}
		if (k0 == 0) mapq0 = __shfl(mq, 0);
		int32_t v = mq;
		for (int d = 32; d > 0; d >>= 1) v += (int32_t)__shfl_xor((uint32_t)v, d);
		sumQ += v;
	}
	if (lane == 0) {
		int stop = 0;
		const int32_t score0 = (int32_t)(L.sc[0] & 0xFFFFFu);
		if (n_regs == 1 && mapq0 >= o.min_mapq) stop = 1;
		else {
			float meanC = (float)sum_sc, meanQ = (float)sumQ;        // sums of small integers: exact in fp32 in any order
			meanC /= (float)n_regs; meanQ /= (float)n_regs;
			const float bestQ = (float)mapq0, bestC = (float)score0;
			float r_bestq = (bestQ > 0) ? (bestQ / 30.0f) : 0.0f; if (r_bestq > 1) r_bestq = 1.0f;
			float r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
			float r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
			const float weighted = o.w_bestq * r_bestq + o.w_bestmq * r_bestmq + o.w_bestmc * r_bestmc;
			if (weighted >= o.w_threshold) stop = 1;
		}
		const rh_chain_head h = heads[RG_IDX(zs[n_u - 1].y)];
		rh_reg best;
		best.cnt = h.cnt; best.score = score0; best.mapq = (uint32_t)mapq0;
		best.qs = (int32_t)h.y0; best.qe = h.y1 + 1; best.rs = (int32_t)h.x0; best.re = h.x1 + 1;
		best.rid = (int32_t)(h.x0 << 1 >> 33); best.rev = (uint32_t)(h.x0 >> 63);
		regions_commit(o, rd, rr, a, r, n_regs, &best, stop);
		rr.need_exact[a] = 0;
	}
	KPROF(6);
}

// ------------------------------------------------------------------------------------------------ regions, primaries in registers
// Measured on unmapped reads (hundreds of chains, carried over every chunk): only a handful of chains are primaries and a
// chain overlaps one or two of them, so mm_set_parent (hit.c:136-195) is bound by the latency of its per-chain step, not
// by interval tests.  Here primary j lives in the VGPRs of lane (j & 63), slot (j >> 6); the chains are streamed through
// registers 64 at a time and broadcast with v_readlane; overlap / mask tests are one ballot; the parent's sub-score and
// n_sub update is a masked register write.  No LDS, no barrier.  Reads with more primaries than fit fall back to
// k_regions_wave.  Instantiated for one slot (half the vector work per chain) and for all of them.  Default selection only (pri_ratio > 0, best_n == 0: secondaries dropped).
#ifndef RGR_SLOTS
#define RGR_SLOTS 2
#endif
#ifndef RGR_PRIM_CAP
#define RGR_PRIM_CAP (64 * RGR_SLOTS)
#endif
#ifndef RGR_DIRECT
#define RGR_DIRECT 4096     // chains per read from which the one-slot instance is skipped
#endif

template <int P>
__global__ __launch_bounds__(64) void k_regions_reg(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, const float *logf_tab, uint32_t n_lo, int only_flagged)
{
	constexpr uint32_t PRIM_CAP = 64u * P < (uint32_t)RGR_PRIM_CAP ? 64u * P : (uint32_t)RGR_PRIM_CAP;
	const uint32_t a = blockIdx.x, lane = threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const int32_t n_u = (int32_t)rr.n_u[a];
	if (n_u <= (int32_t)n_lo) return;
	if (only_flagged && !rr.need_exact[a]) return;                   // settled by the narrower instance
	// tens of thousands of chains (an unmappable read on a large index, chunk after chunk) hold more than 64 primaries as a
	// rule: such reads go to the wider instance directly instead of streaming through this one first
	if (!only_flagged && P < RGR_SLOTS && n_u > RGR_DIRECT) { if (lane == 0) rr.need_exact[a] = 1; return; }
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const rh_chain_head *heads = (const rh_chain_head*)(rr.ws + base * rr.ws_stride);
	const rh_mm128_t *zs = rr.zs + base;                             // keys in radix_sort_128x order (ascending)
	const bool hard = (o.flag & RH_M_HARD_MLEVEL) != 0;
	int32_t pqs[P], pqe[P], psc[P], pcn[P], psub[P], pns[P];
#pragma unroll
	for (int p = 0; p < P; ++p) { pqs[p] = 0; pqe[p] = 0; psc[p] = 0; pcn[p] = 0; psub[p] = 0; pns[p] = 0; }
	uint32_t kk = 0;
	bool overflow = false;
	// region i (descending score: larger first, hit.c:124-126) of the current tile sits in lane i - i0
	int32_t nqs = 0, nqe = 0, nsc = 0, ncn = 0;
	if ((int32_t)lane < n_u) {
		const rh_mm128_t zi = zs[n_u - 1 - (int32_t)lane];
		nsc = (int32_t)(zi.x >> 32); rg_unpack(zi.y, heads, nqs, nqe, ncn);
	}
	for (int32_t i0 = 0; i0 < n_u && !overflow; i0 += 64) {
		const int32_t tqs = nqs, tqe = nqe, tsc = nsc, tcn = ncn;
		const int32_t inext = i0 + 64 + (int32_t)lane;
		if (inext < n_u) {	// next tile's loads fly while this tile is processed
			const rh_mm128_t zi = zs[n_u - 1 - inext];
			nsc = (int32_t)(zi.x >> 32); rg_unpack(zi.y, heads, nqs, nqe, ncn);
		}
		const uint32_t nt = (uint32_t)(n_u - i0 < 64 ? n_u - i0 : 64);
		for (uint32_t t = 0; t < nt; ++t) {
			const int32_t si = (int32_t)rh_readlane((uint32_t)tqs, t), ei = (int32_t)rh_readlane((uint32_t)tqe, t);
			const int32_t sci = (int32_t)rh_readlane((uint32_t)tsc, t), cni = (int32_t)rh_readlane((uint32_t)tcn, t);
			// primaries overlapping [si, ei)
			uint64_t m[P];
			bool ov[P];
			uint32_t n_cov = 0;
#pragma unroll
			for (int p = 0; p < P; ++p) {
				ov[p] = (uint32_t)p * 64u + lane < kk && !(pqe[p] <= si || pqs[p] >= ei);
				m[p] = __ballot(ov[p]);
				n_cov += (uint32_t)__popcll(m[p]);
			}
			int32_t sel = -1;
			if (n_cov > 0) {
				int32_t uncov = 0;
				if (!hard) {
					// length of [si, ei) not covered by the overlapping primaries = (ei - si) - |union of the clipped intervals|;
					// in sweep order (start, end, list index) an interval adds what lies beyond everything before it
					int32_t cs[P], ce[P], reach[P];
#pragma unroll
					for (int p = 0; p < P; ++p) { cs[p] = pqs[p] < si ? si : pqs[p]; ce[p] = pqe[p] > ei ? ei : pqe[p]; reach[p] = si; }
					if (n_cov > 1) {
#pragma unroll
						for (int q = 0; q < P; ++q) {
							uint64_t mm = m[q];
							while (mm) {
								const uint32_t l2 = (uint32_t)__builtin_ctzll(mm);
								mm &= mm - 1;
								const int32_t s2 = (int32_t)rh_readlane((uint32_t)cs[q], l2), e2 = (int32_t)rh_readlane((uint32_t)ce[q], l2);
								const uint32_t j2 = (uint32_t)q * 64u + l2;
#pragma unroll
								for (int p = 0; p < P; ++p) {
									const uint32_t j = (uint32_t)p * 64u + lane;
									const bool before = s2 < cs[p] || (s2 == cs[p] && (e2 < ce[p] || (e2 == ce[p] && j2 < j)));
									if (before && e2 > reach[p]) reach[p] = e2;
								}
							}
						}
					}
					int32_t uni = 0;
#pragma unroll
					for (int p = 0; p < P; ++p) {
						const int32_t from = cs[p] > reach[p] ? cs[p] : reach[p];
						const int32_t add = ov[p] && ce[p] > from ? ce[p] - from : 0;
						uint64_t mm = m[p];
						while (mm) { const uint32_t l2 = (uint32_t)__builtin_ctzll(mm); mm &= mm - 1; uni += (int32_t)rh_readlane((uint32_t)add, l2); }
					}
					uncov = (ei - si) - uni;
				}
				// first overlapping primary (list order) that masks region i
#pragma unroll
				for (int p = 0; p < P; ++p) {
					if (m[p] == 0) continue;                                 // (wave-uniform: no primary of this slot overlaps)
					bool hit = false;
					if (ov[p]) {
						const int32_t sj = pqs[p], ej = pqe[p];
						const int32_t mn = ej - sj < ei - si ? ej - sj : ei - si;
						const int32_t mx = ej - sj > ei - si ? ej - sj : ei - si;
						const int32_t ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
						hit = (float)ol / (float)mn - (float)uncov / (float)mx > o.mask_level && uncov <= o.mask_len;
					}
					const uint64_t hm = __ballot(hit);
					if (sel < 0 && hm) sel = p * 64 + (int32_t)__builtin_ctzll(hm);
				}
			}
			if (i0 == 0 && t == 0) sel = -1;                       // the best chain opens the list
			if (sel >= 0) {
#pragma unroll
				for (int p = 0; p < P; ++p)
					if ((uint32_t)sel == (uint32_t)p * 64u + lane) { if (psub[p] < sci) psub[p] = sci; if (cni >= pcn[p]) ++pns[p]; }
			} else {
				if (kk >= PRIM_CAP) { overflow = true; break; }
				if (rr.reg_out && lane == 0) rr.reg_out[(base + kk) * 18u] = i0 + (int32_t)t;   // (stage-level export: which sorted region the primary is)
#pragma unroll
				for (int p = 0; p < P; ++p)
					if (kk == (uint32_t)p * 64u + lane) { pqs[p] = si; pqe[p] = ei; psc[p] = sci; pcn[p] = cni; psub[p] = 0; pns[p] = 0; }
				++kk;
			}
		}
	}
	if (overflow) { if (lane == 0) rr.need_exact[a] = 1; return; }   // re-done by k_regions_wave / the serial kernel
	// secondaries dropped (mm_select_sub with best_n = 0): the kept regions are exactly the primaries, in order
	const int32_t n_regs = (int32_t)kk;
	int64_t sum_sc = 0;
#pragma unroll
	for (int p = 0; p < P; ++p) { int32_t v = (uint32_t)p * 64u + lane < kk ? psc[p] : 0; for (int d = 32; d > 0; d >>= 1) v += (int32_t)__shfl_xor((uint32_t)v, d); sum_sc += v; }
	const float uniq_ratio = (float)sum_sc / (float)(sum_sc + rr.rep_len[a]);
	int64_t sumQ = 0;
	int32_t mapq0 = 0;
#pragma unroll
	for (int p = 0; p < P; ++p) {
		int32_t mq = 0;
		if ((uint32_t)p * 64u + lane < kk) {
			const int32_t sc = psc[p], cn = pcn[p];
			const float pen_s1 = (float)((sc > 100 ? 1.0 : 0.01 * (double)sc) * (double)uniq_ratio);
			float pen_cm = cn > 10 ? 1.0f : 0.1f * (float)cn;
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			const int32_t subsc = psub[p] > o.min_sc ? psub[p] : o.min_sc;
			const float x = (float)subsc / (float)sc;
			mq = (int32_t)(pen_cm * 40.0f * (1.0f - x) * logf_int(sc, logf_tab));
			mq -= (int32_t)(4.343f * logf_int(pns[p] + 1, logf_tab) + .499f);
			mq = mq > 0 ? mq : 0;
			mq = mq < 60 ? mq : 60;
			if (rr.reg_out) { __threadfence_block(); const int32_t k = p * 64 + (int32_t)lane; reg_export_primary(rr, base, n_u, k, rr.reg_out[(base + (uint32_t)k) * 18u], psub[p], pns[p], mq); }
		}
		if (p == 0) mapq0 = (int32_t)rh_readlane((uint32_t)mq, 0);
		int32_t v = mq;
		for (int d = 32; d > 0; d >>= 1) v += (int32_t)__shfl_xor((uint32_t)v, d);
		sumQ += v;
	}
	if (lane == 0) {
		int stop = 0;
		const int32_t score0 = psc[0];
		if (n_regs == 1 && mapq0 >= o.min_mapq) stop = 1;
		else {
			float meanC = (float)sum_sc, meanQ = (float)sumQ;        // sums of small integers: exact in fp32 in any order
			meanC /= (float)n_regs; meanQ /= (float)n_regs;
			const float bestQ = (float)mapq0, bestC = (float)score0;
			float r_bestq = (bestQ > 0) ? (bestQ / 30.0f) : 0.0f; if (r_bestq > 1) r_bestq = 1.0f;
			float r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
			float r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
			const float weighted = o.w_bestq * r_bestq + o.w_bestmq * r_bestmq + o.w_bestmc * r_bestmc;
			if (weighted >= o.w_threshold) stop = 1;
		}
		const rh_chain_head h = heads[RG_IDX(zs[n_u - 1].y)];
		rh_reg best;
		best.cnt = h.cnt; best.score = score0; best.mapq = (uint32_t)mapq0;
		best.qs = (int32_t)h.y0; best.qe = h.y1 + 1; best.rs = (int32_t)h.x0; best.re = h.x1 + 1;
		best.rid = (int32_t)(h.x0 << 1 >> 33); best.rev = (uint32_t)(h.x0 >> 63);
		regions_commit(o, rd, rr, a, r, n_regs, &best, stop);
		rr.need_exact[a] = 0;
	}
}

// ------------------------------------------------------------------------------------------------ regions, 64 chains at a time
// Reads with thousands of chains (an unmappable read on a large index carries tens of thousands by its last chunks): streaming
// them one by one through mm_set_parent costs a latency-bound step per chain.  Here the CHAINS of a tile sit in the lanes and
// the primaries in LDS: every lane evaluates its chain against the whole primary list (uncovered length by a sweep over the
// primaries sorted by start, then the first primary in list order that masks it - two uniform loops, LDS broadcasts).  A
// chain's outcome depends on earlier chains only through the primaries they add, and a new primary is appended at the END
// of the list: so every lane up to the first one that found no masking primary is final (secondaries: max / count updates of
// their parents are order-free); that lane becomes a primary, and only the later lanes of the tile that overlap it are
// evaluated again.  A tile takes 1 + (primaries it adds) rounds instead of 64 serial steps.
#ifndef RGB_PCAP
#define RGB_PCAP 1024      // primaries held in LDS
#endif
// PCAP: the kernel comes in three sizes of the primary list (RGB_PCAP / 4, / 2, / 1: 8.7, 17.4, 34.8 KB of LDS - a 64-thread workgroup of the
// largest gets one wavefront per SIMD); a read goes to the smallest one that holds 1.25 x + 32 the primaries of its previous chunk (they
// accumulate with the chains a read carries) and, should it overflow after all, again to the next
template <int PCAP>
struct rgb_lds {
	int32_t pqs[PCAP], pqe[PCAP];        // primaries in list order: query interval
	uint32_t psc[PCAP], pcn[PCAP], psub[PCAP], pns[PCAP];   // score, anchors, best secondary score, secondaries with >= as many anchors
	int32_t sqs[PCAP], sqe[PCAP];        // the same intervals sorted by (start, end)
	uint16_t sid[PCAP];                       // ... and their place in the list
	uint32_t kk;
	int32_t lmax;                                 // the longest primary: a primary that overlaps [si, ei) starts after si - lmax
};

template <int PCAP>
__global__ __launch_bounds__(64) void k_regions_batch(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, const float *logf_tab, uint32_t n_lo)
{
	__shared__ rgb_lds<PCAP> L;
	const uint32_t a = blockIdx.x, lane = threadIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const int32_t n_u = (int32_t)rr.n_u[a];
	if (n_u <= (int32_t)n_lo || !rr.need_exact[a]) return;
	const uint32_t r = rr.act[a];
	if (PCAP < RGB_PCAP) { const uint32_t prev = (uint32_t)rd.ls_ncregs[r]; if (prev + prev / 4u + 32u > (uint32_t)PCAP) return; }   // (need_exact stays set: a larger instance follows)
	const uint64_t base = rr.a_off[a];
	const rh_chain_head *heads = (const rh_chain_head*)(rr.ws + base * rr.ws_stride);
	const rh_mm128_t *zs = rr.zs + base;                             // keys in radix_sort_128x order (ascending)
	const bool hard = (o.flag & RH_M_HARD_MLEVEL) != 0;
	if (lane == 0) { L.kk = 0; L.lmax = 0; }
	__syncthreads();
	// the next tile's keys are requested a tile ahead and stay as loaded until the tile starts (taken apart at once, the wait would sit right behind the load: a round trip a tile)
	rh_mm128_t zn; zn.x = 0; zn.y = 0;
	if ((int32_t)lane < n_u) zn = zs[n_u - 1 - (int32_t)lane];         // descending: larger score first (hit.c:124-126)
	for (int32_t i0 = 0; i0 < n_u; i0 += 64) {
		const int32_t i = i0 + (int32_t)lane;
		int32_t si = 0, ei = 0, sci = 0, cni = 0;
		bool pending = i < n_u;
		RH_LANDED(zn.x, zn.y);
		if (pending) { sci = (int32_t)(zn.x >> 32); rg_unpack(zn.y, heads, si, ei, cni); }
		RH_LANDED3(si, ei, cni);                                       // (a head looked up by rg_unpack - rare - is waited for before the next request goes out, not behind it)
		if (i + 64 < n_u) zn = zs[n_u - 1 - (i + 64)];
		bool need_eval = pending;
		int32_t sel = -1;
		while (__ballot(pending)) {
			const uint32_t kk = L.kk;
			if (__ballot(need_eval)) {
				// Only the primaries that overlap the chain matter, and in the list sorted by start they sit in one stretch: those that
				// start after si - (longest primary) and before ei.  A junk read's thousands of short chains overlap a handful of its
				// hundreds of primaries each: a binary search and two short per-lane loops instead of two sweeps over the whole list.
				if (need_eval) {
					const int32_t from_s = si - L.lmax;
					uint32_t lo = 0, hi = kk;                           // first j with sqs[j] > from_s
					while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (L.sqs[mid] > from_s) hi = mid; else lo = mid + 1; }
					int32_t reach = si, cov = 0, uncov = 0;
					uint64_t ovm = 0;                                   // which of the 64 primaries from lo on overlap the chain (the sweep finds them; the masking test below looks at those only)
					bool ov_far = hard;                                 // ... one further on does (or no sweep was made): the masking test scans the stretch itself
					if (!hard) {
						for (uint32_t j = lo; j < kk; ++j) {
							const int32_t s = L.sqs[j], e = L.sqe[j];
							if (s >= ei) break;
							if (e > si) {
								const int32_t cs = s < si ? si : s, ce = e > ei ? ei : e, from = cs > reach ? cs : reach;
								if (ce > from) cov += ce - from;
								if (ce > reach) reach = ce;
								if (j - lo < 64u) ovm |= 1ull << (j - lo); else ov_far = true;
							}
						}
						uncov = (ei - si) - cov;
					}
					int32_t first = 0x7FFFFFFF;                        // the first primary IN LIST ORDER that masks the chain (hit.c:231-246)
					#define RGB_MASK_TEST(j_) do { \
						const int32_t sj = L.sqs[(j_)], ej = L.sqe[(j_)]; \
						const int32_t mn = ej - sj < ei - si ? ej - sj : ei - si; \
						const int32_t mx = ej - sj > ei - si ? ej - sj : ei - si; \
						const int32_t ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si); \
						if ((float)ol / (float)mn - (float)uncov / (float)mx > o.mask_level && uncov <= o.mask_len) { const int32_t id = (int32_t)L.sid[(j_)]; if (id < first) first = id; } } while (0)
					if (!ov_far) {
						while (ovm) { const uint32_t j = lo + (uint32_t)__builtin_ctzll(ovm); ovm &= ovm - 1; RGB_MASK_TEST(j); }
					} else {
						for (uint32_t j = lo; j < kk; ++j) {
							if (L.sqs[j] >= ei) break;
							if (L.sqe[j] > si) RGB_MASK_TEST(j);
						}
					}
					#undef RGB_MASK_TEST
					sel = first == 0x7FFFFFFF ? -1 : first;
				}
				need_eval = false;
			}
			const uint64_t m_new = __ballot(pending && sel < 0);
			const uint32_t f = m_new ? (uint32_t)__builtin_ctzll(m_new) : 64u;
			if (pending && lane < f) {	// secondaries of existing primaries: final
				atomicMax(&L.psub[sel], (uint32_t)sci);
				if ((uint32_t)cni >= L.pcn[sel]) atomicAdd(&L.pns[sel], 1u);
				pending = false;
			}
			if (f < 64u) {	// chain i0 + f opens a new primary
				if (kk >= (uint32_t)PCAP) return;                    // (need_exact stays set: the serial kernels take the read)
				const int32_t fs = __shfl(si, (int)f), fe = __shfl(ei, (int)f), fsc = __shfl(sci, (int)f), fcn = __shfl(cni, (int)f);
				// its place in the intervals sorted by (start, end)
				uint32_t below = 0;
				for (uint32_t j = lane; j < kk; j += 64) below += (L.sqs[j] < fs || (L.sqs[j] == fs && L.sqe[j] <= fe)) ? 1u : 0u;
				for (int d = 32; d > 0; d >>= 1) below += __shfl_xor(below, d);
				int32_t ms[(PCAP + 63) / 64], me[(PCAP + 63) / 64];
				uint16_t mi[(PCAP + 63) / 64];
#pragma unroll
				for (int q = 0; q < (PCAP + 63) / 64; ++q) { const uint32_t j = (uint32_t)q * 64u + lane; ms[q] = 0; me[q] = 0; mi[q] = 0; if (j >= below && j < kk) { ms[q] = L.sqs[j]; me[q] = L.sqe[j]; mi[q] = L.sid[j]; } }
				__syncthreads();
#pragma unroll
				for (int q = 0; q < (PCAP + 63) / 64; ++q) { const uint32_t j = (uint32_t)q * 64u + lane; if (j >= below && j < kk) { L.sqs[j + 1] = ms[q]; L.sqe[j + 1] = me[q]; L.sid[j + 1] = mi[q]; } }
				if (lane == 0) {
					L.sqs[below] = fs; L.sqe[below] = fe; L.sid[below] = (uint16_t)kk;
					if (fe - fs > L.lmax) L.lmax = fe - fs;
					L.pqs[kk] = fs; L.pqe[kk] = fe; L.psc[kk] = (uint32_t)fsc; L.pcn[kk] = (uint32_t)fcn; L.psub[kk] = 0; L.pns[kk] = 0;
					if (rr.reg_out) rr.reg_out[(base + kk) * 18u] = i0 + (int32_t)f;   // (stage-level export: which sorted region the primary is)
					L.kk = kk + 1;
				}
				if (lane == f) pending = false;
				if (pending && !(fe <= si || fs >= ei)) need_eval = true;   // (lanes after f that overlap it see one more primary)
			}
			__syncthreads();
		}
	}
	// secondaries dropped (mm_select_sub with best_n = 0): the kept regions are exactly the primaries, in order
	const int32_t n_regs = (int32_t)L.kk;
	int64_t sum_sc = 0;
	for (int32_t k0 = 0; k0 < n_regs; k0 += 64) { const int32_t k = k0 + (int32_t)lane; int32_t v = k < n_regs ? (int32_t)L.psc[k] : 0; for (int d = 32; d > 0; d >>= 1) v += (int32_t)__shfl_xor((uint32_t)v, d); sum_sc += v; }
	const float uniq_ratio = (float)sum_sc / (float)(sum_sc + rr.rep_len[a]);
	int64_t sumQ = 0;
	int32_t mapq0 = 0;
	for (int32_t k0 = 0; k0 < n_regs; k0 += 64) {
		const int32_t k = k0 + (int32_t)lane;
		int32_t mq = 0;
		if (k < n_regs) {
			const int32_t sc = (int32_t)L.psc[k], cn = (int32_t)L.pcn[k];
			const float pen_s1 = (float)((sc > 100 ? 1.0 : 0.01 * (double)sc) * (double)uniq_ratio);
			float pen_cm = cn > 10 ? 1.0f : 0.1f * (float)cn;
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			const int32_t subsc = (int32_t)L.psub[k] > o.min_sc ? (int32_t)L.psub[k] : o.min_sc;
			const float x = (float)subsc / (float)sc;
			mq = (int32_t)(pen_cm * 40.0f * (1.0f - x) * logf_int(sc, logf_tab));
			mq -= (int32_t)(4.343f * logf_int((int32_t)L.pns[k] + 1, logf_tab) + .499f);
			mq = mq > 0 ? mq : 0;
			mq = mq < 60 ? mq : 60;
			if (rr.reg_out) { __threadfence_block(); reg_export_primary(rr, base, n_u, k, rr.reg_out[(base + (uint32_t)k) * 18u], (int32_t)L.psub[k], (int32_t)L.pns[k], mq); }
		}
		if (k0 == 0) mapq0 = __shfl(mq, 0);
		int32_t v = mq;
		for (int d = 32; d > 0; d >>= 1) v += (int32_t)__shfl_xor((uint32_t)v, d);
		sumQ += v;
	}
	if (lane == 0) {
		int stop = 0;
		const int32_t score0 = (int32_t)L.psc[0];
		if (n_regs == 1 && mapq0 >= o.min_mapq) stop = 1;
		else {
			float meanC = (float)sum_sc, meanQ = (float)sumQ;        // sums of small integers: exact in fp32 in any order
			meanC /= (float)n_regs; meanQ /= (float)n_regs;
			const float bestQ = (float)mapq0, bestC = (float)score0;
			float r_bestq = (bestQ > 0) ? (bestQ / 30.0f) : 0.0f; if (r_bestq > 1) r_bestq = 1.0f;
			float r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
			float r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
			const float weighted = o.w_bestq * r_bestq + o.w_bestmq * r_bestmq + o.w_bestmc * r_bestmc;
			if (weighted >= o.w_threshold) stop = 1;
		}
		const rh_chain_head h = heads[RG_IDX(zs[n_u - 1].y)];
		rh_reg best;
		best.cnt = h.cnt; best.score = score0; best.mapq = (uint32_t)mapq0;
		best.qs = (int32_t)h.y0; best.qe = h.y1 + 1; best.rs = (int32_t)h.x0; best.re = h.x1 + 1;
		best.rid = (int32_t)(h.x0 << 1 >> 33); best.rev = (uint32_t)(h.x0 >> 63);
		regions_commit(o, rd, rr, a, r, n_regs, &best, stop);
		rr.need_exact[a] = 0;
	}
}

// ------------------------------------------------------------------------------------------------ DTW re-scoring of chains (f4)
// --dtw-evaluate-chains (rmap.cpp:128-208, 355-374; dtw.cpp): every region kept by mm_select_sub is aligned - read events against
// the target's expected signal (RH_I_STORE_SIG) - between consecutive anchors ("sparse", the default) or over the whole chain
// ("global"), with a slanted band (default) or the full matrix; regions are taken in order and one whose best attainable score falls
// below the best alignment found so far is abandoned.  An opt-in accuracy mode of the reference: one lane per read, the reference's
// loops as they are - its band DP keeps three anti-diagonal buffers and reads cells an earlier anti-diagonal left behind, which a
// re-formulation would not reproduce.  The MAPQ of this mode takes logf of the (fractional) alignment score: that and the mapping
// decision are left to the host's libm (rh_api.cpp), the device computes the alignment scores and commits the host's verdict.
RH_DEV float dtw_dist(float a, float b) { return fabsf(a - b); }
RH_DEV float dtw_min3(float top, float left, float topleft) { const float m = left < top ? left : top; return topleft < m ? topleft : m; }   // std::min(std::min(top, left), topleft)

RH_DEV float dtw_full(const float *a, uint32_t a_len, const float *b, uint32_t b_len, bool excl, float *dp)	// DTW_global dtw.cpp:37-66
{
	dp[0] = dtw_dist(a[0], b[0]);
	for (uint32_t j = 1; j < a_len; ++j) dp[j] = dp[j - 1] + dtw_dist(a[j], b[0]);
	for (uint32_t i = 1; i < b_len; ++i) {
		float old_left = dp[0];
		dp[0] = dp[0] + dtw_dist(a[0], b[i]);
		for (uint32_t j = 1; j < a_len; ++j) {
			const float top = dp[j - 1], left = dp[j], topleft = old_left;
			dp[j] = dtw_min3(top, left, topleft) + dtw_dist(a[j], b[i]);
			old_left = left;
		}
	}
	return excl ? dp[a_len - 1] - dtw_dist(a[a_len - 1], b[b_len - 1]) : dp[a_len - 1];
}
// DTW_global_slantedbanded_antidiagonalwise dtw.cpp:273-523; store: 3 * dpsize floats.  Returns NaN if the buffers do not fit `cap`.
// WAVE: all 64 lanes of a (one-wavefront) workgroup call it with the same arguments and share the work the way the reference's AVX code does - the
// cells of an anti-diagonal only read the two anti-diagonals before it, so they go to the lanes, a barrier between anti-diagonals; every cell is the
// same expression on the same operands as in the serial form (stale cells outside the clipped range included: nobody writes them), so the result is
// the serial one bit for bit.  A stretch between two anchors of a noise chain spans hundreds of events (E. coli-scale run: 8 % of the stretches are
// longer than 256 events and hold 76 % of all anti-diagonals), and one lane walked each of them.
template <bool WAVE>
RH_DEV float dtw_banded_t(const float *a, uint32_t a_length, const float *b, uint32_t b_length, int band_radius, bool excl, float *store, uint32_t cap, uint32_t lane)
{
	const int L0 = WAVE ? (int)lane : 0, LS = WAVE ? 64 : 1;
	if (a_length < b_length) { const float *tv = a; const uint32_t tl = a_length; a = b; a_length = b_length; b = tv; b_length = tl; }
	const int extra = (int)(((a_length - b_length) * (uint32_t)band_radius + a_length - 1u) / a_length);
	band_radius += extra;
	const int plen = band_radius + (band_radius % 2 == 0 ? 1 : 0), slen = band_radius + (band_radius % 2 == 1 ? 1 : 0);
	const bool primary_larger = plen > slen;
	const int dpsize = plen > slen ? plen : slen;
	if ((uint64_t)dpsize * 3u > cap) return __uint_as_float(0x7FC00000u);
	float *dp0 = store, *dp1 = store + dpsize, *dp2 = store + 2 * dpsize, *tmp;
	if (WAVE) { RH_WG_FENCE(); __syncthreads(); }                    // (the buffers may hold what an earlier call of the wavefront left)
	for (int i = L0; i < dpsize * 3; i += LS) store[i] = 1e10f;
	if (WAVE) { RH_WG_FENCE(); __syncthreads(); }
	int center_row = 0;
	{	// iteration 0: the top left corner
		const int off = plen / 2;
		if (L0 == 0 && 0 < (int)b_length && 0 < (int)a_length) { if (primary_larger) dp2[off] = dtw_dist(a[0], b[0]); else dp2[off + 1] = dtw_dist(a[0], b[0]); }
		tmp = dp0; dp0 = dp1; dp1 = dp2; dp2 = tmp;
		if (WAVE) { RH_WG_FENCE(); __syncthreads(); }
	}
	bool prev_inc = false;
	for (int it = 1; (uint32_t)it < a_length; ++it) {
		const int center_column = it;
		bool inc = false;
		if ((int64_t)(center_row + 1) * (int64_t)a_length <= (int64_t)b_length * (int64_t)center_column) { ++center_row; inc = true; }
		if (inc) {	// the secondary anti-diagonal of a step down
			const int si = center_column + slen / 2 - 1, sj = center_row - slen / 2;
			int o0 = 0; if (si - (int)a_length + 1 > o0) o0 = si - (int)a_length + 1; if (-sj > o0) o0 = -sj;
			int o1 = slen; if (si + 1 < o1) o1 = si + 1; if ((int)b_length - sj < o1) o1 = (int)b_length - sj;
			for (int off = o0 + L0; off < o1; off += LS) {
				const int i = si - off, j = sj + off;
				float top, topleft, left;
				if (primary_larger) { top = dp1[off]; topleft = dp0[off]; left = dp1[off + 1]; }
				else {
					const bool is_first = off == 0, is_last = off == slen - 1;
					top = is_first ? 1e10f : dp1[off];
					topleft = is_first && !prev_inc ? 1e10f : dp0[off];
					left = is_last ? 1e10f : dp1[off + 1];
				}
				dp2[off] = dtw_min3(top, left, topleft) + dtw_dist(a[i], b[j]);
			}
			tmp = dp0; dp0 = dp1; dp1 = dp2; dp2 = tmp;
			if (WAVE) { RH_WG_FENCE(); __syncthreads(); }
		}
		const int si = center_column + plen / 2, sj = center_row - plen / 2;
		int o0 = 0; if (si - (int)a_length + 1 > o0) o0 = si - (int)a_length + 1; if (-sj > o0) o0 = -sj;
		int o1 = plen; if (si + 1 < o1) o1 = si + 1; if ((int)b_length - sj < o1) o1 = (int)b_length - sj;
		for (int off = o0 + L0; off < o1; off += LS) {
			const int i = si - off, j = sj + off;
			const bool is_first = off == 0, is_last = off == plen - 1;
			float top, topleft, left;
			if (primary_larger) {
				if (inc) { top = is_first ? 1e10f : dp1[off - 1]; topleft = dp0[off]; left = is_last ? 1e10f : dp1[off]; }
				else { top = is_first ? 1e10f : dp1[off - 1]; topleft = is_first ? 1e10f : dp0[off - 1]; left = dp1[off]; }
				dp2[off] = dtw_min3(top, left, topleft) + dtw_dist(a[i], b[j]);
			} else {	// (accesses to a primary anti-diagonal start at [1])
				if (inc) { top = dp1[off]; topleft = dp0[off + 1]; left = dp1[off + 1]; }
				else { top = is_first ? 1e10f : dp1[off]; topleft = is_first && !prev_inc ? 1e10f : dp0[off]; left = dp1[off + 1]; }
				dp2[off + 1] = dtw_min3(top, left, topleft) + dtw_dist(a[i], b[j]);
			}
		}
		tmp = dp0; dp0 = dp1; dp1 = dp2; dp2 = tmp;
		if (WAVE) { RH_WG_FENCE(); __syncthreads(); }
		prev_inc = inc;
	}
	float res = primary_larger ? dp1[plen / 2] : dp1[plen / 2 + 1];
	if (excl) res -= dtw_dist(a[a_length - 1], b[b_length - 1]);
	return res;
}
RH_DEV float dtw_banded(const float *a, uint32_t a_length, const float *b, uint32_t b_length, int band_radius, bool excl, float *store, uint32_t cap)
{
	return dtw_banded_t<false>(a, a_length, b, b_length, band_radius, excl, store, cap, 0u);
}
// one alignment of the "sparse" border constraint: the stretch between anchors `part` and `part + 1` of a chain (rmap.cpp:171-196).  *fits = false
// (and nothing computed) when the DP does not fit `cap` floats of `dp`.
RH_DEV float dtw_part(const rh_dev_opt &o, const rh_mm128_t *anchors, uint32_t part, uint32_t parts, const float *ref, const float *ev, float *dp, uint32_t cap, uint32_t *qlen_out, bool *fits)
{
	const rh_mm128_t sa = anchors[part], ea = anchors[part + 1];
	const float *rv = ref + (uint32_t)sa.x; const uint32_t rlen = (uint32_t)ea.x - (uint32_t)sa.x + 1u;
	const float *qv = ev + (uint32_t)sa.y; const uint32_t qlen = (uint32_t)ea.y - (uint32_t)sa.y + 1u;
	const bool excl = part != parts - 1u;
	*qlen_out = qlen; *fits = true;
	float sub;
	if (o.dtw_fill == 0u) { if (qlen > cap) { *fits = false; return 0.0f; } sub = dtw_full(qv, qlen, rv, rlen, excl, dp); }
	else { int band = (int)((float)qlen * o.dtw_band_frac); if (band < 1) band = 1; sub = dtw_banded(qv, qlen, rv, rlen, band, excl, dp, cap); if (sub != sub) { *fits = false; return 0.0f; } }
	return sub;
}
// ... by the whole wavefront (a long stretch, left over by the lanes' pass): the banded DP in `lds`, or - a band that does not fit it - in the read's
// global buffer, cooperative either way; the full matrix stays lane 0's.  Lane 0's return value is the one that counts.
#ifndef DTW_COOP_MIN
#define DTW_COOP_MIN 48u      // stretches of more events than this (the longer side) are the wavefront's, shorter ones a lane's
#endif
RH_DEV float dtw_part_wave(const rh_dev_opt &o, const rh_mm128_t *anchors, uint32_t part, uint32_t parts, const float *ref, const float *ev, float *lds, uint32_t lds_cap, float *gdp, uint32_t g_cap, uint32_t lane, bool *bad)
{
	const rh_mm128_t sa = anchors[part], ea = anchors[part + 1];
	const float *rv = ref + (uint32_t)sa.x; const uint32_t rlen = (uint32_t)ea.x - (uint32_t)sa.x + 1u;
	const float *qv = ev + (uint32_t)sa.y; const uint32_t qlen = (uint32_t)ea.y - (uint32_t)sa.y + 1u;
	const bool excl = part != parts - 1u;
	if (o.dtw_fill == 0u) {
		float sub = 0.0f;
		if (lane == 0) { if (qlen > g_cap) *bad = true; else sub = dtw_full(qv, qlen, rv, rlen, excl, gdp); }
		return sub;
	}
	int band = (int)((float)qlen * o.dtw_band_frac); if (band < 1) band = 1;
	float sub = dtw_banded_t<true>(qv, qlen, rv, rlen, band, excl, lds, lds_cap, lane);
	if (sub != sub) sub = dtw_banded_t<true>(qv, qlen, rv, rlen, band, excl, gdp, g_cap, lane);
	if (sub != sub) { *bad = true; sub = 0.0f; }
	return sub;
}
// align_chain rmap.cpp:128-208.  Returns the alignment score (-1e10: abandoned); *bad set if the DP buffers were too small.
RH_DEV float dtw_align_chain(const rh_dev_opt &o, const rh_reg &c, const rh_mm128_t *anchors, const float *ref, const float *ev, float min_score, float *dp, uint32_t dp_cap, bool *bad)
{
	float cost = 0.0f;
	uint32_t n_aligned = 0;
	if (o.dtw_border == 0u) {	// global
		const float *rv = ref + c.rs; const uint32_t rlen = (uint32_t)(c.re - c.rs + 1);
		const float *qv = ev + c.qs; const uint32_t qlen = (uint32_t)(c.qe - c.qs + 1);
		if ((float)qlen * o.dtw_match_bonus < min_score) return -1e10f;
		if (o.dtw_fill == 0u) { if (qlen > dp_cap) { *bad = true; return 0.0f; } cost = dtw_full(qv, qlen, rv, rlen, false, dp); }
		else { int band = (int)((float)qlen * o.dtw_band_frac); if (band < 1) band = 1; cost = dtw_banded(qv, qlen, rv, rlen, band, false, dp, dp_cap); }
		n_aligned = qlen;
	} else {	// between consecutive anchors
		const uint32_t parts = (uint32_t)c.cnt - 1u;
		float cur_max = (float)(uint32_t)(c.qe - c.qs + 1) * o.dtw_match_bonus;
		for (uint32_t part = 0; part < parts; ++part) {
			const rh_mm128_t sa = anchors[part], ea = anchors[part + 1];
			const float *rv = ref + (uint32_t)sa.x; const uint32_t rlen = (uint32_t)ea.x - (uint32_t)sa.x + 1u;
			const float *qv = ev + (uint32_t)sa.y; const uint32_t qlen = (uint32_t)ea.y - (uint32_t)sa.y + 1u;
			if (cur_max < min_score) return -1e10f;
			const bool excl = part != parts - 1u;
			float sub;
			if (o.dtw_fill == 0u) { if (qlen > dp_cap) { *bad = true; return 0.0f; } sub = dtw_full(qv, qlen, rv, rlen, excl, dp); }
			else { int band = (int)((float)qlen * o.dtw_band_frac); if (band < 1) band = 1; sub = dtw_banded(qv, qlen, rv, rlen, band, excl, dp, dp_cap); }
			cost += sub;
			cur_max -= sub;
			n_aligned += qlen;
		}
	}
	if (cost != cost) { *bad = true; return 0.0f; }
	return (float)n_aligned * o.dtw_match_bonus - cost;
}

// reg->events: the events of this round's chunk behind those of the chunks before (rmap.cpp:237-241; a dropped chunk adds nothing)
__global__ __launch_bounds__(NT) void k_events_append(rh_dev_reads rd, rh_dev_round rr)
{
	const uint32_t a = blockIdx.x;
	if (a >= rr.n_act || rr.skip[a]) return;
	const uint32_t r = rr.act[a], n = rr.n_ev[a], off = rd.ev_off[r];
	const float *src = rr.ev + (size_t)a * rr.ev_cap;
	float *dst = rd.events + (size_t)r * rd.ev_stride + off;
	for (uint32_t i = threadIdx.x; i < n && off + i < rd.ev_stride; i += NT) dst[i] = src[i];
}

// one read per lane: regions up to mm_select_sub (the serial core on HBM scratch, as k_regions_big), then the alignment score of every kept
// region in order (rmap.cpp:355-374).  Leaves rg[] in the read's scratch, rr.dtw_n[a] = regions kept (0x80000000 | .. on a buffer overflow)
// One WAVEFRONT per read.  The region core (up to mm_select_sub) is lane 0's; the alignments of a region under the default "sparse" border
// constraint - one small DP between every two consecutive anchors of the chain, independent of each other - are dealt to the 64 lanes, each
// with DTW_LANE_CAP floats of LDS for its DP (a stretch that needs more is lane 0's afterwards, in the read's global buffer); lane 0 then adds
// the costs up IN PART ORDER (fp32 sums are order dependent) and applies the reference's running early exit (rmap.cpp:176), so the score is the
// serial one bit for bit.  (One lane per read, 64 reads of divergent control flow per wavefront: 886 reads/s on the E. coli-scale run against
// 21 k for the CPU reference.)
#ifndef DTW_LANE_CAP
#define DTW_LANE_CAP 48
#endif
__global__ __launch_bounds__(64) void k_regions_dtw(rh_dev_opt o, rh_dev_index ix, rh_dev_reads rd, rh_dev_round rr)
{
	__shared__ float s_dp[64 * DTW_LANE_CAP];
	__shared__ float s_sub[64];
	__shared__ uint32_t s_ql[64];
	__shared__ int32_t s_n, s_flag;
	__shared__ float s_best;
	const uint32_t a = blockIdx.x, lane = threadIdx.x;
	if (a >= rr.n_act) return;
	if (lane == 0) rr.dtw_n[a] = 0;
	if (rr.skip[a]) return;
	const uint32_t r = rr.act[a];
	const uint64_t base = rr.a_off[a];
	const int32_t n_u = (int32_t)rr.n_u[a];
	if (n_u == 0) return;
	const rh_mm128_t *an = rr.anc + base;
	const uint64_t *u = rr.u + base;
	unsigned char *wsr = rr.ws + base * rr.ws_stride;           // 64 B per anchor >= 128 B per chain (min_cnt >= 2)
	rh_reg *rg = (rh_reg*)wsr;
	rh_chain_head *ch = (rh_chain_head*)(wsr + (size_t)64 * n_u);
	rh_mm128_t *z = (rh_mm128_t*)(wsr + (size_t)96 * n_u);
	uint64_t *cov = (uint64_t*)(wsr + (size_t)112 * n_u);
	int32_t *w = (int32_t*)(wsr + (size_t)120 * n_u), *tmp = (int32_t*)(wsr + (size_t)124 * n_u);
	uint32_t *cw = (uint32_t*)rg;
	if (lane == 0) {
		uint32_t k = 0;
		for (int32_t i = 0; i < n_u; ++i) {
			const uint32_t cnt = (uint32_t)u[i];
			const rh_mm128_t f0 = rh_an_ld(rr, rr.anc, base + k), f1 = rh_an_ld(rr, rr.anc, base + k + cnt - 1);
			rh_chain_head h; h.x0 = f0.x; h.y0 = f0.y; h.x1 = (int32_t)f1.x; h.y1 = (int32_t)f1.y; h.cnt = (int32_t)cnt; h.k = (int32_t)k;
			ch[i] = h;
			k += cnt;
		}
		int stop;
		s_n = regions_core(o, n_u, u, ch, rg, z, cov, w, tmp, cw, rr.rep_len[a], rr.n_ev[a], rd.ev_off[r], nullptr, &stop, true);
	}
	RH_WG_FENCE();
	__syncthreads();
	const int32_t n_regs = s_n;
	// alignment scores go where the chain heads were (4 B per region; the heads are not needed any more)
	float *ascore = (float*)ch;
	const float *ev = rd.events + (size_t)r * rd.ev_stride;
	float *dp = rr.dtw_ws + (size_t)a * rr.dtw_stride;
	bool bad = false;
	float best = 0.0f;                                               // (lane 0's; the other lanes never read it)
	for (int32_t i = 0; i < n_regs; ++i) {
		const rh_reg c = rg[i];
		const float *ref = ix.sig + ix.sig_off[2 * (size_t)c.rid + (c.rev ? 1u : 0u)];
		float as = 0.0f;
		if (o.dtw_border == 0u && o.dtw_fill != 0u) {	// one alignment over the whole chain (rmap.cpp:141-170), its band across the lanes
			if (lane == 0) s_best = best;
			__syncthreads();
			const uint32_t rlen = (uint32_t)(c.re - c.rs + 1), qlen = (uint32_t)(c.qe - c.qs + 1);
			if ((float)qlen * o.dtw_match_bonus < s_best) as = -1e10f;
			else {
				int band = (int)((float)qlen * o.dtw_band_frac); if (band < 1) band = 1;
				float cost = dtw_banded_t<true>(ev + c.qs, qlen, ref + c.rs, rlen, band, false, s_dp, 64u * (uint32_t)DTW_LANE_CAP, lane);
				if (cost != cost) cost = dtw_banded_t<true>(ev + c.qs, qlen, ref + c.rs, rlen, band, false, dp, rr.dtw_stride, lane);
				if (cost != cost) { bad = true; as = 0.0f; } else as = (float)qlen * o.dtw_match_bonus - cost;
			}
			__syncthreads();
		}
		else if (o.dtw_border == 0u) { if (lane == 0) as = dtw_align_chain(o, c, an + c.as, ref, ev, best, dp, rr.dtw_stride, &bad); }   // ... full matrix: serial
		else {
			const uint32_t parts = (uint32_t)c.cnt - 1u;
			float cost = 0.0f, cur_max = (float)(uint32_t)(c.qe - c.qs + 1) * o.dtw_match_bonus;
			uint32_t n_aligned = 0;
			bool gone = false;
			for (uint32_t p0 = 0; p0 < parts; p0 += 64) {             // (wave-uniform: `gone` is broadcast below)
				const uint32_t part = p0 + lane;
				if (part < parts) {	// the lanes' pass: short stretches, one each, in the lane's 48 floats of LDS; long ones are left (NaN) for the wavefront
					const rh_mm128_t sa = (an + c.as)[part], ea = (an + c.as)[part + 1];
					const uint32_t rl = (uint32_t)ea.x - (uint32_t)sa.x + 1u, ql0 = (uint32_t)ea.y - (uint32_t)sa.y + 1u;
					if (o.dtw_fill != 0u && (rl > DTW_COOP_MIN || ql0 > DTW_COOP_MIN)) { s_sub[lane] = __uint_as_float(0x7FC00000u); s_ql[lane] = ql0; }
					else {
						bool fits; uint32_t ql;
						const float sub = dtw_part(o, an + c.as, part, parts, ref, ev, s_dp + (size_t)lane * DTW_LANE_CAP, (uint32_t)DTW_LANE_CAP, &ql, &fits);
						s_sub[lane] = fits ? sub : __uint_as_float(0x7FC00000u); s_ql[lane] = ql;
					}
				}
				__syncthreads();
				// in part order (fp32 sums are order dependent), with the reference's running early exit; a stretch still open is computed now, by all lanes
				const uint32_t m = parts - p0 < 64u ? parts - p0 : 64u;
				for (uint32_t q = 0; q < m; ++q) {
					if (lane == 0) s_flag = cur_max < best ? 1 : 0;         // rmap.cpp:176: the chain cannot beat the best alignment so far any more
					__syncthreads();
					if (s_flag) { gone = true; break; }
					float sub = s_sub[q];
					if (sub != sub) sub = dtw_part_wave(o, an + c.as, p0 + q, parts, ref, ev, s_dp, 64u * (uint32_t)DTW_LANE_CAP, dp, rr.dtw_stride, lane, &bad);
					if (lane == 0) { cost += sub; cur_max -= sub; n_aligned += s_ql[q]; }
					__syncthreads();
				}
				if (lane == 0) s_n = gone ? -1 : n_regs;
				__syncthreads();
				if (s_n < 0) break;
			}
			if (lane == 0) { as = gone ? -1e10f : (float)n_aligned * o.dtw_match_bonus - cost; s_n = n_regs; }
			__syncthreads();
		}
		if (lane == 0) {
			if (as >= o.dtw_min_score) { if (as > best) best = as; }
			else if (as < o.dtw_min_score && as < 0.0f) as = o.dtw_min_score > 0.0f ? 0.0f : o.dtw_min_score;
			ascore[i] = as;
		}
	}
	if (lane == 0) rr.dtw_n[a] = (uint32_t)n_regs | (bad ? 0x80000000u : 0u);
}

// mm_set_mapq's DTW branch (hit.c:502-539) and the mapping decision (rmap.cpp:423-500) ON THE DEVICE (round 6).  The MAPQ is
//     (int)(pen * 40 * (1 - x) * 2 * logf(alignment score))      with a fractional argument
// and the reference's logf is the host libm's - not correctly rounded, so no device routine reproduces it bit for bit.  But the result is TRUNCATED to an
// integer: the double-precision logarithm, rounded to float, is within half an ulp of the true value, the host's logf within an ulp or so of it (glibc: 0.82),
// so the host's result is one of the five floats around ours - and if all five give the same integer, that integer is the reference's whatever its libm
// returned.  A read all of whose regions are settled that way (all but ~1 in 10^4) is decided and committed here; the others are flagged (bit 30 of dtw_n) and
// take the old way - 32 bytes per region to the host, its libm, the verdict back - which used to be two blocking round trips per slice for EVERY read.
#define DTW_DECIDED 0x40000000u
__global__ void k_dtw_decide(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr, const float *logf_tab, uint32_t *n_host)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t r = rr.act[a];
	if (rr.skip[a]) { rd.ls_ncregs[r] = 0; rr.dtw_n[a] = DTW_DECIDED; return; }   // chunk dropped: creg stays NULL (rmap.cpp:232-235, :419)
	const uint32_t nw = rr.dtw_n[a];
	if (nw & 0x80000000u) { atomicAdd(n_host, 1u); return; }          // a DP buffer overflowed: the host reports it
	const int32_t nr = (int32_t)(nw & 0x3FFFFFFFu);
	if (!nr) { regions_commit(o, rd, rr, a, r, 0, nullptr, 0); rr.dtw_n[a] = DTW_DECIDED; return; }
	const uint64_t base = rr.a_off[a];
	const int32_t n_u = (int32_t)rr.n_u[a];
	unsigned char *wsr = rr.ws + base * rr.ws_stride;
	rh_reg *rg = (rh_reg*)wsr;
	const float *ascore = (const float*)(wsr + (size_t)64 * n_u);
	int64_t sum_sc = 0;
	for (int32_t i = 0; i < nr; ++i) if (rg[i].parent == rg[i].id) sum_sc += rg[i].score;
	const float uniq_ratio = (float)sum_sc / (float)(sum_sc + (int64_t)rr.rep_len[a]);
	bool sure = true;
	for (int32_t i = 0; i < nr && sure; ++i) {
		const rh_reg &q = rg[i];
		float pen_s1 = (float)((q.score > 100 ? 1.0 : 0.01 * (double)q.score) * (double)uniq_ratio);
		float pen_cm = q.cnt > 10 ? 1.0f : 0.1f * (float)q.cnt;
		pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
		const int32_t subsc = q.subsc > o.min_sc ? q.subsc : o.min_sc;
		const float x = (float)subsc / (float)q.score0;
		const float as = ascore[i];
		int mapq = 0;
		if (as > 0.0f) {
			const float t = pen_cm * 40.0f * (1.0f - x) * 2.0f;
			const float l0 = (float)log((double)as);
			float c[5]; c[2] = l0; c[1] = nextafterf(l0, -INFINITY); c[0] = nextafterf(c[1], -INFINITY); c[3] = nextafterf(l0, INFINITY); c[4] = nextafterf(c[3], INFINITY);
			mapq = (int)(t * c[0]);
			for (int k = 1; k < 5; ++k) if ((int)(t * c[k]) != mapq) sure = false;
		}
		const int32_t ns1 = q.n_sub + 1;
		if (ns1 < 0 || ns1 >= (1 << 20)) sure = false;                // (beyond the table of the host's logf(integer))
		else mapq -= (int)(4.343f * logf_tab[ns1] + .499f);
		mapq = mapq > 0 ? mapq : 0;
		rg[i].mapq = (uint32_t)(mapq < 60 ? mapq : 60);
	}
	if (!sure) { atomicAdd(n_host, 1u); return; }
	// the decision (rmap.cpp:423-500, one chain reported: no all-chains mode here) - the statements of the host version in dtw_regions_stage, in their order
	int sel = 0, stop = 0;
	if (nr == 1 && ((int32_t)rg[0].mapq >= o.min_mapq || ascore[0] >= o.dtw_min_score)) stop = 1;
	else {
		float meanC = 0, meanQ = 0;
		for (int32_t i = 0; i < nr; ++i) { meanC += (float)rg[i].score; meanQ += (float)(int32_t)rg[i].mapq; }
		meanC /= (float)nr; meanQ /= (float)nr;
		float bestA = ascore[0];
		int best = 0;
		for (int32_t i = 1; i < nr; ++i) if (ascore[i] > bestA) { bestA = ascore[i]; best = i; }
		const float bestQ = (float)(int32_t)rg[best].mapq, bestC = (float)rg[best].score;
		float weighted = 0.0f;
		if (bestA >= o.dtw_min_score) {
			float r_bestma = (bestA > 0) ? (bestA / 50.0f) : 0.0f; if (r_bestma < 0) r_bestma = 0.0f;
			float r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
			float r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
			weighted = o.w_bestma * r_bestma + o.w_bestmq * r_bestmq + o.w_bestmc * r_bestmc;
		}
		if (weighted >= o.w_threshold) { stop = 1; sel = best; }
	}
	const rh_reg selr = rg[sel];
	regions_commit(o, rd, rr, a, r, nr, &selr, stop);
	rr.dtw_n[a] = nw | DTW_DECIDED;
}

__global__ __launch_bounds__(NT) void k_dtw_pack(rh_dev_round rr)
{
	const uint32_t a = blockIdx.x;
	if (a >= rr.n_act) return;
	if (rr.dtw_n[a] & DTW_DECIDED) return;
	const uint32_t n = rr.dtw_n[a] & 0x3FFFFFFFu;
	if (!n) return;
	const uint64_t base = rr.a_off[a];
	const int32_t n_u = (int32_t)rr.n_u[a];
	const unsigned char *wsr = rr.ws + base * rr.ws_stride;
	const rh_reg *rg = (const rh_reg*)wsr;
	const float *ascore = (const float*)(wsr + (size_t)64 * n_u);
	int32_t *out = (int32_t*)rr.dtw_rec + rr.dtw_off[a] * 8;
	for (uint32_t i = threadIdx.x; i < n; i += NT) {
		const rh_reg &q = rg[i];
		int32_t *o8 = out + (size_t)i * 8;
		o8[0] = q.score; o8[1] = q.cnt; o8[2] = q.subsc; o8[3] = q.score0; o8[4] = q.n_sub; o8[5] = q.parent == q.id ? 1 : 0; o8[6] = (int32_t)__float_as_uint(ascore[i]); o8[7] = 0;
	}
}

// the host's verdict per read {region to commit, its MAPQ, stop}: the state the record is built from (rmap.cpp:423-500, 507-586)
__global__ void k_dtw_commit(rh_dev_opt o, rh_dev_reads rd, rh_dev_round rr)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= rr.n_act) return;
	const uint32_t r = rr.act[a];
	if (rr.dtw_n[a] & DTW_DECIDED) return;                            // decided and committed on the device (k_dtw_decide)
	if (rr.skip[a]) { rd.ls_ncregs[r] = 0; return; }                // chunk dropped: creg stays NULL (rmap.cpp:232-235, :419)
	const uint32_t n = rr.dtw_n[a] & 0x3FFFFFFFu;
	if (!n) { regions_commit(o, rd, rr, a, r, 0, nullptr, 0); return; }
	const rh_reg *rg = (const rh_reg*)(rr.ws + rr.a_off[a] * rr.ws_stride);
	rh_reg sel = rg[rr.dtw_dec[3 * (size_t)a]];
	sel.mapq = (uint32_t)rr.dtw_dec[3 * (size_t)a + 1];
	regions_commit(o, rd, rr, a, r, (int32_t)n, &sel, rr.dtw_dec[3 * (size_t)a + 2]);
}

// ------------------------------------------------------------------------------------------------ launchers
// the multi-workgroup sorter's second record array is whichever 16-byte-per-anchor arena is idle during that sort
static void sort_scratch(rh_sort_job &jb, const rh_dev_round &r, rh_mm128_t *idle) { jb.big_alt = idle; jb.big_ws = r.sort_ws; jb.big_ws_bytes = r.sort_ws_bytes; jb.big_pin = r.sort_pin; jb.big_total = r.sort_total; }

int rhk_zsort(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r)
{
	if (!r.n_act) return 0;
	RH_LAUNCH(k_zbuild, r.n_act, NT, 0, s, o, r, bt_lds_cap());
	// candidates (score, anchor index) -> reference order; scores are full of ties: exact permutation for every read
	rh_sort_job jb = { r.n_act, r.skip, r.a_off, r.n_z, r.raw, r.zs, r.need_exact, r.ws, RH_WS_PER_ANCHOR, 64, (uint8_t)(o.min_sc >= 0), 32, 0, 0, r.max_anchors };   // keys = scores >= min_sc: non-negative int32
	sort_scratch(jb, r, r.prev_out);                               // (the carry staging is written by the backtrack, later)
	jb.kind = 2;
	if (r.z8) jb.rf = rh_rec_fmt{1, 32, 32, 0};                     // 8-byte candidates: key = the high word
	if (bt_lone_on(o, r)) jb.dead_cnt = r.n_v;                      // the candidates without a predecessor (k_zbuild): the backtrack stops before them
	return rhk_sort_job(s, jb, true, 0u);
}

int rhk_backtrack(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r)
{
	if (!r.n_act) return 0;
	const bool bt_wave = getenv("RH_BT_WAVE") != nullptr;            // RH_BT_WAVE=1: one wavefront per read, 64 candidates a round (A/B and test aid; read per call)
	const uint32_t lds_cap = bt_lds_cap();                           // (the same value k_zbuild got: both read the environment per call)
	const uint32_t top = r.max_anchors ? r.max_anchors : 0xFFFFFFFFu;
	if (bt_wave) RH_LAUNCH((k_backtrack_spec<64, 0>), r.n_act, 64, rh_wave_lds(), s, o, rd, r, 0u);
	else if (!lds_cap) RH_LAUNCH((k_backtrack_spec<256, 0>), r.n_act, 256, rh_wave_lds(), s, o, rd, r, 0u);
	else {   // three LDS classes (16 / 32 / 64 KB of marks: 6 / 4 / 2 workgroups a CU), each launched if the round has such reads, and HBM beyond
		constexpr int W0 = BT_LDS_ANCHORS / 128, W1 = BT_LDS_ANCHORS / 64, W2 = BT_LDS_ANCHORS / 32;
		const int mc = bt_lds_min_class();
		if (mc == 0) RH_LAUNCH((k_backtrack_spec<256, W0>), r.n_act, 256, 0, s, o, rd, r, 0u);
		if (mc <= 1 && (mc == 1 || top > 32u * W0)) RH_LAUNCH((k_backtrack_spec<BT_LDS_THREADS, W1>), r.n_act, BT_LDS_THREADS, 0, s, o, rd, r, mc == 1 ? 0u : 32u * W0);
		if (mc == 2 || top > 32u * W1) RH_LAUNCH((k_backtrack_spec<BT_LDS_THREADS, W2>), r.n_act, BT_LDS_THREADS, 0, s, o, rd, r, mc == 2 ? 0u : 32u * W1);
		if (top > 32u * W2) RH_LAUNCH((k_backtrack_spec<256, 0>), r.n_act, 256, rh_wave_lds(), s, o, rd, r, 32u * W2);
	}
	// compact_a: the chains (gathered by the backtrack itself) put into the reference's order of their first anchor, written back
	rh_sort_job jb = { r.n_act, r.skip, r.a_off, r.n_u, r.raw, r.zs, r.need_exact2, r.ws, RH_WS_PER_ANCHOR, 64, r.akey_on, r.akey_lo, r.akey_mid, 1, r.max_anchors };   // keys = first anchors
	sort_scratch(jb, r, r.anc);                                    // (the backtrack has copied every chain out of the sorted anchors; k_chain_reorder rewrites them)
	jb.kind = 3;
	jb.rf = r.cfmt;
	// Two chains agree on the key only where two anchors of the read do (see rhk_sort): long lists - an unmappable read on a large index has tens
	// of thousands of chains - are placed level by level in any order, the reads whose chains do hold equal keys get their keys again and the exact passes
	static const bool exact_all = RH_DEVENV("RH_CSORT_EXACT") != nullptr;   // development aid: the exact passes for every read
	if (exact_all || (jb.n_max && jb.n_max <= rhk_sort_lds_max(jb))) { if (rhk_sort_job(s, jb, false, 0u)) return -1; }   // (nothing beyond the LDS classes: their fast pass / tie redo is exact already)
	else {
		uint32_t n_redo = 0;
		jb.any_order = 1; jb.redo_skip = r.need_exact; jb.n_redo = &n_redo;   // (need_exact: idle between the candidate sort and the region stage)
		if (jb.rf.rec8) jb.any_up = (uint8_t)(64u - ((uint32_t)jb.rf.mid + 1u));   // (rh_rec_fmt::up)
		RH_HIP(hipMemsetAsync(r.need_exact, 1, r.n_act, s));
		if (rhk_sort_job(s, jb, false, 0u)) return -1;
		if (n_redo) {
			RH_LAUNCH(k_chain_keys, r.n_act, NT, 0, s, r, (const uint8_t*)r.need_exact);
			jb.any_order = 0; jb.redo_skip = nullptr; jb.n_redo = nullptr; jb.skip = r.need_exact; jb.tie_path = 1;
			if (rhk_sort_job(s, jb, false, 0u)) return -1;
		}
	}
	RH_LAUNCH(k_chain_reorder, r.n_act, NT, 0, s, rd, r);
	return 0;
}

static bool regions_wave_ok(const rh_dev_opt &o) { return o.best_n == 0 && o.pri_ratio > 0.0f && !(o.flag & RH_M_ALL_CHAINS); }
bool rhk_regions_fast_ok(const rh_dev_opt &o) { return regions_wave_ok(o); }

// chain heads + hashed keys of reads with many chains, put into the reference's order (hit.c:111-126)
int rhk_regions_sort(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r)
{
	if (!r.n_act || !regions_wave_ok(o)) return 0;
	if (r.max_anchors > 0x3FFFFFFu) { rh_set_error("a read with %u anchors in one round: beyond the 2^26 chains a region sort record numbers", r.max_anchors); return -1; }
	RH_LAUNCH(k_regions_prep, r.n_act, NT, 0, s, o, rd, r, (const uint8_t*)nullptr);
	rh_sort_job jb = { r.n_act, r.skip, r.a_off, r.n_u, r.raw, r.zs, r.need_exact2, r.ws, RH_WS_PER_ANCHOR, 64, 0, 0, 0, 0, r.max_anchors };   // keys = hashed: full 64 bits
	sort_scratch(jb, r, r.prev_out);                               // (the carried anchors have left the staging: the round loop packs them before this sort)
	jb.kind = 4;
	// The keys are score << 32 | (count ^ 32-bit hash): two of a read's chains agree on one with probability ~2^-32 per pair, so the
	// long segments (unmappable reads: tens of thousands of chains) are placed without the token walks, the few reads that do hold
	// equal keys are found afterwards and only they are sorted again with the exact passes (need_exact is idle here: rhk_regions
	// resets it).  RH_RSORT_EXACT=1 (development aid) takes the exact passes for every read.
	static const bool exact_all = RH_DEVENV("RH_RSORT_EXACT") != nullptr;
	if (exact_all) return rhk_sort_job(s, jb, false, (uint32_t)RG_SMALL);
	uint32_t n_redo = 0;
	jb.any_order = 1; jb.redo_skip = r.need_exact; jb.n_redo = &n_redo;
	RH_HIP(hipMemsetAsync(r.need_exact, 1, r.n_act, s));           // (1 = no redo; the sorter clears the reads whose chains hold equal keys)
	if (rhk_sort_job(s, jb, false, (uint32_t)RG_SMALL)) return -1;
	static const bool trace = RH_DEVENV("RH_BS_TRACE") != nullptr;
	if (trace) fprintf(stderr, "RSORT any-order: %u of %u reads hold equal region keys and are redone\n", n_redo, r.n_act);
	if (!n_redo) return 0;
	RH_LAUNCH(k_regions_prep, r.n_act, NT, 0, s, o, rd, r, (const uint8_t*)r.need_exact);   // their keys again (the sorter overwrote its input)
	jb.any_order = 0; jb.redo_skip = nullptr; jb.n_redo = nullptr; jb.skip = r.need_exact; jb.tie_path = 1;   // (covers r.skip: the check marks skipped reads "no redo")
	return rhk_sort_job(s, jb, false, (uint32_t)RG_SMALL);
}

void rhk_regions(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r, const float *logf_tab)
{
	if (!r.n_act) return;
	const bool wave_ok = regions_wave_ok(o);
	// need_exact[] doubles as "this read still needs the serial region kernel"
	RH_HIP_VOID(hipMemsetAsync(r.need_exact, wave_ok ? 0 : 1, r.n_act, s));
	if (wave_ok) {
		// one register slot (<= 64 primaries: nearly every read) first; the reads that overflow it again with all slots
		RH_LAUNCH(k_regions_reg<1>, r.n_act, 64, 0, s, o, rd, r, logf_tab, (uint32_t)RG_SMALL, 0);
		RH_LAUNCH(k_regions_batch<RGB_PCAP / 4>, r.n_act, 64, rh_wave_lds(), s, o, rd, r, logf_tab, (uint32_t)RG_SMALL);   // reads with many chains / more than 64 primaries
		RH_LAUNCH(k_regions_batch<RGB_PCAP / 2>, r.n_act, 64, rh_wave_lds(), s, o, rd, r, logf_tab, (uint32_t)RG_SMALL);
		RH_LAUNCH(k_regions_batch<RGB_PCAP>, r.n_act, 64, rh_wave_lds(), s, o, rd, r, logf_tab, (uint32_t)RG_SMALL);
		if (RGR_SLOTS > 1) RH_LAUNCH(k_regions_reg<RGR_SLOTS>, r.n_act, 64, 0, s, o, rd, r, logf_tab, (uint32_t)RG_SMALL, 1);
		RH_LAUNCH(k_regions_wave<RGW_CAP0>, r.n_act, 64, 0, s, o, rd, r, logf_tab, (uint32_t)RG_SMALL);
		RH_LAUNCH(k_regions_wave<RGW_CAP>, r.n_act, 64, 0, s, o, rd, r, logf_tab, (uint32_t)RGW_CAP0);
	}
	// skip / no-chain bookkeeping for every read + LDS serial core for reads the wave kernel could not take
	RH_LAUNCH(k_regions, r.n_act, 64, 0, s, o, rd, r, logf_tab, (uint32_t)RG_SMALL, wave_ok ? 1 : 0);
	RH_LAUNCH(k_regions_big, (r.n_act + 63) / 64, 64, 0, s, o, rd, r, logf_tab, 0u, (uint32_t)RG_SMALL, 0);
	RH_LAUNCH(k_regions_big, (r.n_act + 63) / 64, 64, 0, s, o, rd, r, logf_tab, (uint32_t)RG_CAP, 0x7FFFFFFFu, wave_ok ? 1 : 0);
}

void rhk_events_append(hipStream_t s, const rh_dev_reads &rd, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_events_append, r.n_act, NT, 0, s, rd, r); }
void rhk_regions_dtw(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_regions_dtw, r.n_act, 64, 0, s, o, ix, rd, r); }
void rhk_dtw_pack(hipStream_t s, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_dtw_pack, r.n_act, NT, 0, s, r); }
void rhk_dtw_decide(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r, const float *logf_tab, uint32_t *n_host) { if (r.n_act) RH_LAUNCH(k_dtw_decide, (r.n_act + 63) / 64, 64, 0, s, o, rd, r, logf_tab, n_host); }
void rhk_dtw_commit(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r) { if (r.n_act) RH_LAUNCH(k_dtw_commit, (r.n_act + 63) / 64, 64, 0, s, o, rd, r); }
