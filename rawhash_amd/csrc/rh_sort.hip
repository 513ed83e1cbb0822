// Anchor sort stage: block-cooperative, LDS-resident reproduction of klib's radix_sort_128x (reference ksort.h:101-151,
// called at rmap.cpp:121) for one read per workgroup.
//
// The reference sort is an in-place MSD "American flag" radix sort (8-bit digits from bit 56, insertion sort for
// ranges <= 64).  It is unstable, and the order it leaves equal keys in is observed by the chaining DP, so the result
// must be the reference's exact permutation.  Structure used here:
//   * ranges whose keys agree on a byte are untouched by that pass -> jump straight to the highest differing byte;
//   * the final insertion sorts (ranges <= 64) are stable and independent -> one wavefront per range, in parallel;
//   * a pass over >= 2 buckets is a permutation that only matters for EQUAL keys:
//       fast pass : digit scatter with LDS atomics (order inside a bucket arbitrary), then the sorted result is checked
//                   for adjacent equal keys; a read without ties has a unique sorted order, so it is already exact;
//       exact pass: (same launch, reads with ties only) the sort is redone from the input order, and every range that
//                   holds a tied key gets the reference's cycle-leader permutation - closed form for two buckets
//                   (prefix ranks), otherwise one wavefront replays the cycle walk with the range's digits and the
//                   bucket heads held in VGPRs (v_readlane + s_set_gpr_idx, no memory on the dependency chain);
//                   tie-free ranges keep the atomic scatter.
// LDS per record: 8 B key (by original index) + 2 B permutation + 2 B scratch map; permutations are applied through
// registers.  Records (16 B) are gathered from / written to HBM once.
#include <type_traits>
#include "rh_kernels.h"
#include "rh_devutil.h"

// -DRH_KPROF: development aid, shader-clock cycles spent per phase of k_sort_block summed over workgroups (thread 0)
#ifdef RH_KPROF
__device__ unsigned long long rh_kprof_acc[32];
#define KPROF_DECL unsigned long long kp_t0 = clock64()
#define KPROF(slot) do { if (threadIdx.x == 0 && L.prof != 0) { const unsigned long long t_ = clock64(); atomicAdd(&rh_kprof_acc[slot], t_ - kp_t0); kp_t0 = t_; } } while (0)
extern "C" __attribute__((visibility("default"))) int rh_debug_kprof(unsigned long long *out, int reset)
{
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rh_kprof_acc), sizeof(rh_kprof_acc)) != hipSuccess) return -1;
	if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rh_kprof_acc), z, sizeof(z)) != hipSuccess) return -1; }
	return 0;
}
#else
#define KPROF_DECL
#define KPROF(slot)
#endif

// how a 64-bit key  hi << 63 | mid << 32 | lo  is kept in a 32-bit word (all zero: kept whole)
struct sort_kc { uint32_t lo_bits, mid_bits, hi_bits; };

template <int CAP, class KT>
struct alignas(16) sort_lds {
	KT key[CAP];                               // by original index; 32-bit words when the job says its keys fit (sort_kc)
	uint16_t ia[CAP];                          // current arrangement: position -> original index
	uint16_t xm[CAP];                          // scratch of a pass: gather map (position -> source position) or rank lists
	uint32_t sbit[CAP / 32 + 3], ebit[CAP / 32 + 3];   // ranges <= 64 awaiting the stable insertion sort: first / last position marks
	uint32_t tbit[CAP / 32 + 3];               // original indices whose key occurs more than once
	uint32_t rng[2][CAP / 64 + 4];             // ranges > 64 still to be split: beg | end << 16
	uint8_t rsh[2][CAP / 64 + 4];              // ... and the byte shift they are to be split on next
	uint32_t cnt[256], head[256];
	uint8_t dmap[256], inv[256];                 // digit -> rank among the non-empty buckets, and back
	uint32_t w[NT / 64];
	uint64_t r64[NT / 64];
	uint32_t n_rng[2], tie, misc[4], prof;
	// records with equal keys (at most SORT_TG per segment for the short cut below): original index, first sorted position
	// of their group, final position once settled; records of the range being walked, in pop order
	uint16_t tg_idx[32], tg_pos[32], tg_fin[32], tlist[32];
	uint8_t tg_rng[32];
	uint32_t n_tg, n_tl;
	sort_kc kc;
};
#define SORT_TG 32

enum { SORT_FAST = 0, SORT_EXACT_TIED = 1, SORT_EXACT_ALL = 2 };

// digit (byte s / 8 of the ORIGINAL 64-bit key) of a stored key
template <int CAP, class KT>
RH_DEV uint32_t sort_key_digit(const sort_lds<CAP, KT> &L, KT k, int s)
{
	if (sizeof(KT) == 8) return (uint32_t)((uint64_t)k >> s) & 255u;
	const uint32_t ck = (uint32_t)k, lo = L.kc.lo_bits, mid = L.kc.mid_bits;
	if (s < 32) return (uint32_t)(((uint64_t)ck & ((1ull << lo) - 1ull)) >> s) & 255u;
	if (s < 56) return (((ck >> lo) & ((1u << mid) - 1u)) >> (s - 32)) & 255u;          // (mid < 2^24: byte 7 holds the top bit only)
	return L.kc.hi_bits ? ((ck >> (lo + mid)) & 1u) << 7 : 0u;
}
// a XOR of stored keys, back at the original bit positions
template <int CAP, class KT>
RH_DEV uint64_t sort_key_spread(const sort_lds<CAP, KT> &L, uint64_t d)
{
	if (sizeof(KT) == 8) return d;
	const uint32_t lo = L.kc.lo_bits, mid = L.kc.mid_bits;
	return (d & ((1ull << lo) - 1ull)) | ((d >> lo) & ((1ull << mid) - 1ull)) << 32 | (L.kc.hi_bits ? (d >> (lo + mid)) & 1ull : 0ull) << 63;
}
template <int CAP, class KT>
RH_DEV uint32_t sort_digit(const sort_lds<CAP, KT> &L, uint32_t i, int s) { return sort_key_digit(L, L.key[L.ia[i]], s); }

// ia[j] = ia[xm[j]] for j in [beg, end), through registers
template <int CAP, class KT>
RH_DEV void sort_apply_gather(sort_lds<CAP, KT> &L, uint32_t beg, uint32_t end)
{
	constexpr int K = (CAP + NT - 1) / NT;
	uint16_t v[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (beg + (uint32_t)k * NT >= end) break;
		const uint32_t j = beg + (uint32_t)k * NT + threadIdx.x;
		v[k] = j < end ? L.ia[L.xm[j]] : (uint16_t)0;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (beg + (uint32_t)k * NT >= end) break;
		const uint32_t j = beg + (uint32_t)k * NT + threadIdx.x;
		if (j < end) L.ia[j] = v[k];
	}
	__syncthreads();
}

// Two buckets A < B.  Cycle-leader result in closed form: the k-th misplaced element of region A trades places with the
// k-th misplaced element of region B, except that in B every run of in-place elements between two misplaced ones is
// shifted right by one slot and the arrival lands in front of the run.
template <int CAP, class KT>
RH_DEV void sort_two_buckets(sort_lds<CAP, KT> &L, uint32_t beg, uint32_t end, int s, uint32_t cA, uint32_t cB, uint32_t startB)
{
	constexpr int K = (CAP + NT - 1) / NT;
	const uint32_t tid = threadIdx.x;
	uint32_t code[K];                                       // kind | rank << 2; kind: 0 A in place, 1 A misplaced, 2 B misplaced, 3 B in place
	uint32_t nA = 0, nB = 0;                                // misplaced elements seen so far in A / in B
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (beg + (uint32_t)k * NT >= end) break;
		const uint32_t i = beg + (uint32_t)k * NT + tid;
		const bool in = i < end;
		const uint32_t d = in ? sort_digit(L, i, s) : 0u;
		const bool fa = in && i < startB && d == cB, fb = in && i >= startB && d == cA;
		const uint64_t mA = __ballot(fa), mB = __ballot(fb);
		if (lane_id() == 0) L.w[wave_id()] = (uint32_t)__popcll(mA) | (uint32_t)__popcll(mB) << 16;
		__syncthreads();
		uint32_t base = 0, tot = 0;
		for (uint32_t q = 0; q < NT / 64; ++q) { const uint32_t c = L.w[q]; if (q < wave_id()) base += c; tot += c; }
		__syncthreads();
		const uint32_t ra = nA + (base & 0xFFFFu) + lanes_below(mA), rb = nB + (base >> 16) + lanes_below(mB);
		if (fa) L.xm[beg + ra] = (uint16_t)i;              // positions of A's misplaced elements, ascending
		if (fb) L.xm[end - 1 - rb] = (uint16_t)i;          // positions of B's misplaced elements, stored from the back
		code[k] = in ? ((fa ? 1u : fb ? 2u : i < startB ? 0u : 3u) | (fa ? ra : rb) << 2) : ~0u;
		nA += tot & 0xFFFFu; nB += tot >> 16;
	}
	__syncthreads();
	const uint32_t m = nA;
	uint16_t val[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (beg + (uint32_t)k * NT >= end) break;
		const uint32_t i = beg + (uint32_t)k * NT + tid, c = code[k];
		if (c == ~0u) continue;
		const uint32_t kind = c & 3u, r = c >> 2;
		uint32_t dest;
		if (kind == 0) dest = i;
		else if (kind == 1) dest = r == 0 ? startB : (uint32_t)L.xm[end - r] + 1u;
		else if (kind == 2) dest = L.xm[beg + r];
		else dest = r < m ? i + 1 : i;
		code[k] = dest;
		val[k] = L.ia[i];
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (beg + (uint32_t)k * NT >= end) break;
		if (code[k] != ~0u) L.ia[code[k]] = val[k];
	}
	__syncthreads();
}

// The reference's cycle walk over >= 3 buckets, replayed by ONE wavefront (all lanes in lock step on wave-uniform
// values).  A single wavefront issues roughly one instruction every 5 cycles, so the walk is as fast as its step is
// short.  State lives in registers: the non-empty buckets are renumbered 0 .. nbk-1 in digit order; the dense digit of
// relative position p is byte (p & 3) of VGPR [p >> 8] in lane ((p >> 2) & 63); lane (c & 63) holds the head / tail of
// dense bucket c in VGPR [c >> 6] (HB = 1, 2 or 4 of them, picked by nbk).  Each step is a couple of v_readlane; the only
// memory operation is the fire-and-forget LDS store of the gather map xm[dest] = source.
template <int CAP, class KT, int HB>
RH_DEV void sort_cycle_walk_hb(sort_lds<CAP, KT> &L, uint32_t beg, uint32_t end, int s, uint32_t nbk)
{
	constexpr int NR = (CAP + 255) / 256;
	const uint32_t lane = lane_id(), n = end - beg;
	uint32_t dg[NR];
#pragma unroll
	for (int q = 0; q < NR; ++q) {
		uint32_t w = 0;
		const uint32_t p0 = ((uint32_t)q * 64u + lane) * 4u;
		if (p0 < n) {
			for (uint32_t b = 0; b < 4; ++b) if (p0 + b < n) w |= (uint32_t)L.dmap[sort_digit(L, beg + p0 + b, s)] << (8 * b);
		}
		dg[q] = w;
	}
	uint32_t hd[HB], tl[HB];
#pragma unroll
	for (int q = 0; q < HB; ++q) {
		const uint32_t id = (uint32_t)q * 64u + lane;
		hd[q] = 0; tl[q] = 0;
		if (id < nbk) { const uint32_t dgt = L.inv[id]; hd[q] = L.head[dgt] - beg; tl[q] = hd[q] + L.cnt[dgt]; }
	}
	const uint32_t ubeg = rh_uniform(beg);
	uint16_t *xmr = L.xm + ubeg;
	for (uint32_t c = 0; c < nbk; ++c) {
		uint32_t tlc = rh_readlane(tl[0], c & 63u), h = rh_readlane(hd[0], c & 63u);
#pragma unroll
		for (int q = 1; q < HB; ++q) { const uint32_t t2 = rh_readlane(tl[q], c & 63u), h2 = rh_readlane(hd[q], c & 63u); if ((c >> 6) == (uint32_t)q) { tlc = t2; h = h2; } }
		while (h != tlc) {
			uint32_t src = h;
			uint32_t d = (rh_readlane(dg[h >> 8], (h >> 2) & 63u) >> ((h & 3u) * 8u)) & 255u;
			while (d != c) {
				uint32_t r[HB];
#pragma unroll
				for (int k = 0; k < HB; ++k) r[k] = rh_readlane(hd[k], d & 63u);
				uint32_t q = r[0];
#pragma unroll
				for (int k = 1; k < HB; ++k) q = (d >> 6) == (uint32_t)k ? r[k] : q;
#pragma unroll
				for (int k = 0; k < HB; ++k) hd[k] = rh_writelane(hd[k], (HB == 1 || (d >> 6) == (uint32_t)k) ? q + 1 : r[k], d & 63u);
				xmr[q] = (uint16_t)(ubeg + src);
				src = q;
				d = (rh_readlane(dg[q >> 8], (q >> 2) & 63u) >> ((q & 3u) * 8u)) & 255u;
			}
			xmr[h] = (uint16_t)(ubeg + src);
			++h;
		}
	}
}

// Short cut for the usual case - a few pairs of equal keys whose bucket in this pass is final (<= 64 records, sorted
// stably afterwards): inside a bucket the records end up in the order in which the walk pops them, so the order of two
// equal keys is settled the moment the first of them is popped.  The walk therefore stops after all but one of the
// range's tied records have been popped (a third of the way for one pair) and writes no gather map at all: everything
// else in the range already has its final place from the fast pass.  Bit 7 of a cached digit = "tied record".
template <int CAP, class KT, int HB>
RH_DEV void sort_cycle_walk_early(sort_lds<CAP, KT> &L, uint32_t beg, uint32_t end, int s, uint32_t nbk, uint32_t stop)
{
	constexpr int NR = (CAP + 255) / 256;
	const uint32_t lane = lane_id(), n = end - beg;
	uint32_t dg[NR];
#pragma unroll
	for (int q = 0; q < NR; ++q) {
		uint32_t w = 0;
		const uint32_t p0 = ((uint32_t)q * 64u + lane) * 4u;
		if (p0 < n) {
			for (uint32_t b = 0; b < 4; ++b) if (p0 + b < n) {
				const uint32_t idx = L.ia[beg + p0 + b];
				w |= ((uint32_t)L.dmap[sort_key_digit(L, L.key[idx], s)] | ((L.tbit[idx >> 5] >> (idx & 31u)) & 1u) << 7) << (8 * b);
			}
		}
		dg[q] = w;
	}
	uint32_t hd[HB], tl[HB];
#pragma unroll
	for (int q = 0; q < HB; ++q) {
		const uint32_t id = (uint32_t)q * 64u + lane;
		hd[q] = 0; tl[q] = 0;
		if (id < nbk) { const uint32_t dgt = L.inv[id]; hd[q] = L.head[dgt] - beg; tl[q] = hd[q] + L.cnt[dgt]; }
	}
	const uint32_t ubeg = rh_uniform(beg);
	uint32_t ntl = 0;
	for (uint32_t c = 0; c < nbk && ntl < stop; ++c) {
		uint32_t tlc = rh_readlane(tl[0], c & 63u), h = rh_readlane(hd[0], c & 63u);
#pragma unroll
		for (int q = 1; q < HB; ++q) { const uint32_t t2 = rh_readlane(tl[q], c & 63u), h2 = rh_readlane(hd[q], c & 63u); if ((c >> 6) == (uint32_t)q) { tlc = t2; h = h2; } }
		while (h != tlc && ntl < stop) {
			uint32_t db = (rh_readlane(dg[h >> 8], (h >> 2) & 63u) >> ((h & 3u) * 8u)) & 255u;
			if (db >> 7) L.tlist[ntl++] = (uint16_t)(ubeg + h);
			uint32_t d = db & 127u;
			while (d != c && ntl < stop) {
				uint32_t r[HB];
#pragma unroll
				for (int k = 0; k < HB; ++k) r[k] = rh_readlane(hd[k], d & 63u);
				uint32_t q = r[0];
#pragma unroll
				for (int k = 1; k < HB; ++k) q = (d >> 6) == (uint32_t)k ? r[k] : q;
#pragma unroll
				for (int k = 0; k < HB; ++k) hd[k] = rh_writelane(hd[k], (HB == 1 || (d >> 6) == (uint32_t)k) ? q + 1 : r[k], d & 63u);
				db = (rh_readlane(dg[q >> 8], (q >> 2) & 63u) >> ((q & 3u) * 8u)) & 255u;
				if (db >> 7) L.tlist[ntl++] = (uint16_t)(ubeg + q);
				d = db & 127u;
			}
			++h;
		}
	}
	if (lane == 0) L.n_tl = ntl;
}

template <int CAP, class KT>
RH_DEV void sort_cycle_walk(sort_lds<CAP, KT> &L, uint32_t beg, uint32_t end, int s, uint32_t nbk)
{
	if (nbk <= 64) sort_cycle_walk_hb<CAP, KT, 1>(L, beg, end, s, nbk);
	else if (nbk <= 128) sort_cycle_walk_hb<CAP, KT, 2>(L, beg, end, s, nbk);
	else sort_cycle_walk_hb<CAP, KT, 4>(L, beg, end, s, nbk);
}

template <int CAP, class KT>
RH_DEV void sort_split_range(sort_lds<CAP, KT> &L, uint32_t beg, uint32_t end, int shift, int nxt, int pass)
{
	const uint32_t tid = threadIdx.x;
	KPROF_DECL;
	// highest byte on which the range's keys differ (passes above it are identities in the reference); does it hold ties?
	const uint64_t k0 = (uint64_t)L.key[L.ia[beg]];
	uint64_t diff = 0;
	bool tied = false;
	for (uint32_t i = beg + tid; i < end; i += NT) {
		const uint32_t idx = L.ia[i];
		diff |= (uint64_t)L.key[idx] ^ k0;
		if (pass == SORT_EXACT_TIED) tied |= (L.tbit[idx >> 5] >> (idx & 31u) & 1u) != 0;
	}
	const uint64_t tm = __ballot(tied);
	if (lane_id() == 0) L.w[wave_id()] = tm != 0;
	const bool exact = pass != SORT_FAST;
	diff = block_or64(diff, L.r64);                           // (barriers inside publish L.w as well)
	if (exact) diff = sort_key_spread(L, diff);               // (the fast pass works on the stored keys as they are: same order)
	KPROF(1);
	if (diff == 0) return;                                   // all keys equal: every remaining pass is an identity
	// redo pass: a range without tied keys already has its (unique) final order from the fast pass -> nothing to do
	if (pass == SORT_EXACT_TIED && (L.w[0] | L.w[1] | L.w[2] | L.w[3]) == 0) {
		__syncthreads();                                     // every wave has read the flags before the next range's call rewrites them
		return;
	}
	// The byte the reference splits on - or, in the fast pass, the top eight DIFFERING bits wherever they start: a segment without
	// equal keys has one sorted order, however it is reached, and a chromosome's worth of positions (27 bits) then falls into 256
	// buckets of a dozen at once instead of 8 buckets that each need a pass of their own.
	int s;
	if (exact) { s = (63 - __clzll(diff)) & ~7; if (s > shift) s = shift; }
	else { s = 63 - __clzll(diff) - 7; if (s < 0) s = 0; }
	#define SORT_DIGIT(i_) (exact ? sort_digit(L, (i_), s) : (uint32_t)((uint64_t)L.key[L.ia[(i_)]] >> s) & 255u)
	// digit histogram
	L.cnt[tid] = 0;
	__syncthreads();
	for (uint32_t i = beg + tid; i < end; i += NT) atomicAdd(&L.cnt[SORT_DIGIT(i)], 1u);
	__syncthreads();
	const uint32_t my_cnt = L.cnt[tid];
	uint32_t total;
	const uint32_t my_start = beg + block_excl_scan(my_cnt, L.w, total);
	L.head[tid] = my_start;
	uint32_t nbk;
	const uint32_t dense = block_rank(my_cnt != 0, L.w, nbk);
	if (exact && my_cnt != 0) { L.dmap[tid] = (uint8_t)dense; L.inv[dense] = (uint8_t)tid; }
	KPROF(2);
	// permutation of the pass
	if (!exact) {
		for (uint32_t i = beg + tid; i < end; i += NT) { const uint32_t pos = atomicAdd(&L.head[SORT_DIGIT(i)], 1u); L.xm[pos] = (uint16_t)i; }
		__syncthreads();
		KPROF(3);
		sort_apply_gather<CAP, KT>(L, beg, end);
		KPROF(4);
	} else if (nbk == 2) {
		if (my_cnt != 0) { const uint32_t which = my_start == beg ? 0u : 1u; L.misc[which] = tid; L.misc[2 + which] = my_start; }
		__syncthreads();
		sort_two_buckets<CAP, KT>(L, beg, end, s, L.misc[0], L.misc[1], L.misc[3]);
		KPROF(5);
	} else {
		__syncthreads();
		if (pass == SORT_EXACT_TIED && nbk <= 128 && L.n_tg <= SORT_TG) {
			// can the order of the tied records be settled by pop order alone?  (their buckets must be final after this pass)
			if (tid < SORT_TG) L.tg_rng[tid] = 0;
			if (tid == 0) L.misc[0] = 0;
			__syncthreads();
			bool bad = false;
			const uint32_t ntg = L.n_tg;
			for (uint32_t i = beg + tid; i < end; i += NT) {
				const uint32_t idx = L.ia[i];
				if ((L.tbit[idx >> 5] >> (idx & 31u)) & 1u) {
					if (s > 0 && L.cnt[sort_key_digit(L, L.key[idx], s)] > 64u) bad = true;
					for (uint32_t e = 0; e < ntg; ++e) if (L.tg_idx[e] == idx) L.tg_rng[e] = 1;
					atomicAdd(&L.misc[0], 1u);
				}
			}
			const uint64_t bm = __ballot(bad);
			if (lane_id() == 0) L.w[wave_id()] = bm != 0;
			__syncthreads();
			if ((L.w[0] | L.w[1] | L.w[2] | L.w[3]) == 0) {
				const uint32_t stop = L.misc[0] - 1;
				if (wave_id() == 0) { if (nbk <= 64) sort_cycle_walk_early<CAP, KT, 1>(L, beg, end, s, nbk, stop); else sort_cycle_walk_early<CAP, KT, 2>(L, beg, end, s, nbk, stop); }
				__syncthreads();
				KPROF(6);
				if (tid < ntg && L.tg_rng[tid]) {	// my place in my group: pop order; the record never popped comes last
					const uint32_t idx = L.tg_idx[tid], gs = L.tg_pos[tid], ntl = L.n_tl;
					uint32_t in_group = 0, mine = 0xFFFFu;
					for (uint32_t r = 0; r < ntl; ++r) {
						const uint32_t qi = L.ia[L.tlist[r]];
						uint32_t g2 = 0xFFFFu;
						for (uint32_t e = 0; e < ntg; ++e) if (L.tg_idx[e] == qi) g2 = L.tg_pos[e];
						if (g2 == gs) { if (qi == idx) mine = in_group; ++in_group; }
					}
					L.tg_fin[tid] = (uint16_t)(gs + (mine != 0xFFFFu ? mine : in_group));
					atomicAnd(&L.tbit[idx >> 5], ~(1u << (idx & 31u)));      // settled: not for the final rewrite of tied records
				}
				__syncthreads();
				return;
			}
		}
		if (wave_id() == 0) sort_cycle_walk<CAP, KT>(L, beg, end, s, nbk);
		__syncthreads();
		KPROF(6);
		sort_apply_gather<CAP, KT>(L, beg, end);
		KPROF(4);
	}
	// children: one bucket per thread
	if (s > 0 && my_cnt > 1) {
		if (my_cnt > 64) { const uint32_t k = atomicAdd(&L.n_rng[nxt], 1u); L.rng[nxt][k] = my_start | (my_start + my_cnt) << 16; L.rsh[nxt][k] = (uint8_t)(s >= 8 ? s - 8 : 0); }
		else { const uint32_t e = my_start + my_cnt - 1; atomicOr(&L.sbit[my_start >> 5], 1u << (my_start & 31u)); atomicOr(&L.ebit[e >> 5], 1u << (e & 31u)); }
	}
	__syncthreads();
	KPROF(7);
	#undef SORT_DIGIT
}

// Ranges of up to W records whose keys differ in their low 27 bits only: ONE LANE per range sorts the words
// (key bits << 5 | position) in registers with a bitonic network - data-independent, no memory, and the position in the low bits
// makes it stable.  The list: ns pairs (first position, length) in xm, from the front or (back) from the end; a range settled here
// gets bit 15 of its length set, the others are left to the wavefront path.
template <int CAP, class KT, int W>
RH_DEV void sort_lane_ranges(sort_lds<CAP, KT> &L, uint32_t ns, bool back, int pass)
{
	const uint32_t tid = threadIdx.x;
	for (uint32_t q0 = 0; q0 < ns; q0 += NT) {
		const uint32_t q = q0 + tid, p = back ? (uint32_t)(CAP / 2) - 1u - q : q;
		uint32_t b = 0, m = 0;
		if (q < ns) { b = L.xm[2 * p]; m = L.xm[2 * p + 1]; }
		bool mine = q < ns && m <= (uint32_t)W;
		if (mine && (((uint64_t)L.key[L.ia[b]] ^ (uint64_t)L.key[L.ia[b + m - 1]]) >> 27) != 0) mine = false;   // cheap look before loading the range
		if (__ballot(mine) == 0) continue;
		uint32_t c[W];
		uint64_t dif = 0, k0 = 0;
		uint32_t tiedr = 0;
#pragma unroll
		for (int j = 0; j < W; ++j) {
			c[j] = 0xFFFFFFFFu;
			if (mine && (uint32_t)j < m) {
				const uint32_t idx = L.ia[b + j];
				const uint64_t k = (uint64_t)L.key[idx];
				if (j == 0) k0 = k;
				dif |= k ^ k0;
				tiedr |= (L.tbit[idx >> 5] >> (idx & 31u)) & 1u;
				c[j] = (uint32_t)k << 5 | (uint32_t)j;
			}
		}
		if (mine && (dif >> 27) != 0) mine = false;               // keys differ above bit 26: the wavefront path
		const bool skip = mine && pass == SORT_EXACT_TIED && !tiedr;   // final since the fast pass
		if (mine) L.xm[2 * p + 1] = (uint16_t)(m | 0x8000u);          // settled here
		if (__ballot(mine && !skip) == 0) continue;
#pragma unroll
		for (int kk = 2; kk <= W; kk <<= 1)
#pragma unroll
			for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
				for (int i = 0; i < W; ++i) {
					const int l2 = i ^ jj;
					if (l2 > i) { const uint32_t lo = c[i] < c[l2] ? c[i] : c[l2], hi = c[i] < c[l2] ? c[l2] : c[i]; if ((i & kk) == 0) { c[i] = lo; c[l2] = hi; } else { c[i] = hi; c[l2] = lo; } }
				}
		if (mine && !skip) {	// the record that was at position (word & 31) moves to the word's rank; all reads before the first write
			uint16_t nx[W];
#pragma unroll
			for (int j = 0; j < W; ++j) nx[j] = (uint32_t)j < m ? L.ia[b + (c[j] & 31u)] : (uint16_t)0;
#pragma unroll
			for (int j = 0; j < W; ++j) if ((uint32_t)j < m) L.ia[b + j] = nx[j];
		}
	}
}

// the whole radix sort of the n keys in L.key, from the input order, into L.ia
template <int CAP, class KT>
RH_DEV void sort_run(sort_lds<CAP, KT> &L, uint32_t n, int pass)
{
	const uint32_t tid = threadIdx.x;
	KPROF_DECL;
	for (uint32_t i = tid; i < n; i += NT) L.ia[i] = (uint16_t)i;
	for (uint32_t i = tid; i < CAP / 32 + 3; i += NT) { L.sbit[i] = 0; L.ebit[i] = 0; }
	__syncthreads();
	if (tid == 0) {
		L.n_rng[0] = 0; L.n_rng[1] = 0;
		if (n > 64) { L.rng[0][0] = 0u | n << 16; L.rsh[0][0] = 56; L.n_rng[0] = 1; }
		else if (n > 1) { L.sbit[0] = 1u; L.ebit[(n - 1) >> 5] = 1u << ((n - 1) & 31u); }
	}
	__syncthreads();
	for (int cur = 0;; cur ^= 1) {
		const uint32_t nr = L.n_rng[cur];
		if (nr == 0) break;
		for (uint32_t ri = 0; ri < nr; ++ri) {
			const uint32_t be = L.rng[cur][ri];
			sort_split_range<CAP, KT>(L, be & 0xFFFFu, be >> 16, (int)L.rsh[cur][ri], cur ^ 1, pass);
		}
		__syncthreads();
		if (tid == 0) L.n_rng[cur] = 0;
		__syncthreads();
	}
	// Ranges of <= 64 records get klib's insertion sort, i.e. any STABLE sort.  Most hold a dozen or two records whose keys
	// differ in their low bits only: ONE LANE per range then sorts the words (key bits << 5 | position) in registers with a
	// bitonic network - data-independent, no memory, and the position in the low bits makes it stable.  The others (33..64
	// records, keys differing high up) are ranked by a whole wavefront from its lanes' registers.
	KPROF(8);
	if (tid == 0) { L.misc[0] = 0; L.misc[1] = 0; }
	__syncthreads();
	// two lists of the ranges (first position, length) in xm, which is free between passes: up to 16 records from the front, longer
	// ones from the back - a wavefront whose lanes all hold short ranges runs the 16-word network, a third of the 32-word one
	constexpr uint32_t QC = CAP / 2;
	const uint32_t nw32 = (n + 31) / 32;
	for (uint32_t wi = tid; wi < nw32; wi += NT) {
		uint32_t sb = L.sbit[wi];
		while (sb) {
			const uint32_t bit = (uint32_t)__builtin_ctz(sb);
			sb &= sb - 1;
			const uint32_t b = wi * 32 + bit;
			uint32_t ew = L.ebit[wi] >> bit, e = b;
			if (ew) e = b + (uint32_t)__builtin_ctz(ew);
			else { uint32_t x = wi + 1; while ((ew = L.ebit[x]) == 0) ++x; e = x * 32 + (uint32_t)__builtin_ctz(ew); }
			const uint32_t m = e - b + 1;
			const uint32_t p = m <= 16u ? atomicAdd(&L.misc[0], 1u) : QC - 1u - atomicAdd(&L.misc[1], 1u);
			L.xm[2 * p] = (uint16_t)b; L.xm[2 * p + 1] = (uint16_t)m;
		}
	}
	__syncthreads();
	const uint32_t nsA = L.misc[0], nsB = L.misc[1];
	sort_lane_ranges<CAP, KT, 16>(L, nsA, false, pass);
	sort_lane_ranges<CAP, KT, 32>(L, nsB, true, pass);
	__syncthreads();
	const uint32_t wv = rh_uniform(wave_id());
	for (uint32_t qq = wv; qq < nsA + nsB; qq += NT / 64) {
		const uint32_t q = qq < nsA ? qq : QC - 1u - (qq - nsA);
		const uint32_t mm = rh_uniform((uint32_t)L.xm[2 * q + 1]);
		if (mm & 0x8000u) continue;
		const uint32_t b = rh_uniform((uint32_t)L.xm[2 * q]), m = mm, l = lane_id();
		const uint16_t idx = L.ia[b + (l < m ? l : 0u)];
		if (pass == SORT_EXACT_TIED && __ballot((L.tbit[idx >> 5] >> (idx & 31u)) & 1u) == 0) continue;   // final since the fast pass
		const uint64_t k = (uint64_t)L.key[idx];
		const uint32_t klo = (uint32_t)k, khi = (uint32_t)(k >> 32);
		uint32_t rank = 0;
		// the other records' keys come from their lanes' registers (v_readlane with the wave-uniform j), not from LDS
		const uint32_t khi0 = rh_readlane(khi, 0), klo0 = rh_readlane(klo, 0);
		if (__ballot(khi != khi0 || ((klo ^ klo0) >> 26) != 0) == 0) {
			const uint32_t cw = klo << 6 | l;                       // (key bits, position) in one word makes the stable order a plain '<'
			for (uint32_t j = 0; j < m; ++j) rank += rh_readlane(cw, j) < cw ? 1u : 0u;
		} else {
			for (uint32_t j = 0; j < m; ++j) {
				const uint64_t kj = (uint64_t)rh_readlane(khi, j) << 32 | rh_readlane(klo, j);
				rank += (kj < k || (kj == k && j < l)) ? 1u : 0u;
			}
		}
		RH_WAVE_SYNC();                                        // every lane has read the range before it is rewritten
		if (l < m) L.ia[b + rank] = idx;
	}
	__syncthreads();
	KPROF(9);
}


// ------------------------------------------------------------------------------------------------ tie-free segments of 8-byte records
// Round 6.  What the block sorter mostly sees - the strand x target buckets of a large index's chunks, a small index's whole chunks, chain lists - are
// segments WITHOUT equal keys, and such a segment has one sorted order however it is reached (ksort.h:101-151 leaves equal keys in an order of its own;
// distinct keys it simply sorts).  sort_fast gets there with the records themselves in LDS and a handful of LDS operations per record, where the general
// path keeps keys, a 16-bit arrangement and a scratch map and goes through them ~25 times (its own counters: 40 % of those cycles were bank conflicts,
// and more than a quarter of the kernel's time was waiting for the first load and for the second read of the records at the end):
//   1. records -> registers; minimum and maximum of the packed keys (record >> shift: the fields keep their significance, rh_rec_fmt);
//   2. bucket = ((key - min) * m) >> s2, m in 16 .. 31 and s2 such that the key range fills NB = CAP / 8 buckets to 94 % or more: 4 - 8 records a bucket
//      whatever the range is (a plain shift fills between half and all of them); ONE LDS atomic per record gives both the bucket's count and the
//      record's rank inside it;
//   3. counts -> start offsets (a thread owns NB / NT consecutive buckets, here and in 5);
//   4. records scattered into LDS at start + rank;
//   5. every lane sorts its buckets: the words (key - the bucket's lower bound) << 5 | place of up to 16 records in a bitonic network in registers
//      (first version: CAP / 2 buckets of 1 - 4 records, eight rounds of a 16-word network per thread - the networks were 62 % of the kernel, a
//      round costs what it costs however few records it holds), equal neighbours = equal keys (they share a bucket), the records permuted inside the
//      bucket; a bucket of more than 16 records - the tail, or the anchors of a mapped read at its locus, a few hundred within a few hundred positions - is sorted
//      by a whole wavefront (network across the lanes);
//   6. the sorted records stream out of LDS to the destination - no second read of the source.
// Returns 0: done; 1: the segment holds equal keys (nothing written unless write_tied: the caller's exact passes take it from the input order); 2: not for this path
// (more buckets of more than 16 records than a short list holds, or one of more than 256 / 512; a key range of more than 2^35).
template <int CAP> struct sort_fast_cfg {
	static constexpr int nb() { int v = NT; while (v < CAP / 8) v *= 2; return v; }
	static constexpr int NB = nb(), BPT = NB / NT, K = (CAP + NT - 1) / NT;
	static constexpr int lb() { int l = 0; while ((1 << l) < NB) ++l; return l; }
	static constexpr int LB = lb();
	static constexpr bool HOLD = K <= 16;                     // records stay in registers between the passes (else they are read again: L2)
};
#define SORT_FAST_BIG 48
// a wavefront's bucket holds up to 64 x EMAX records; EMAX words a lane: 8 spill in the classes that run six wavefronts a SIMD (80 registers)
template <int CAP> struct sort_fast_big { static constexpr int EMAX = CAP >= 4096 ? 8 : 4, MAX = 64 * EMAX; };
template <int CAP>
struct alignas(16) sort_fast_lds {
	uint64_t rec[CAP + 16 + NT];                              // (+ 16: a lane reads the 16 slots from its bucket's start whatever the bucket holds; + NT: where each thread's masked stores go)
	uint32_t cnt[sort_fast_cfg<CAP>::NB];                     // per bucket: count, then start offset
	uint64_t r64[2 * (NT / 64)];
	uint32_t w[NT / 64];
	uint32_t flag, n_big;
	uint32_t big[SORT_FAST_BIG];                              // buckets of more than 16 records (start | count << 16): a wavefront each
	uint32_t big_lo[2 * SORT_FAST_BIG];                        // ... and the lower bound of their keys (relative to the minimum; two words)
};
// how a key finds its bucket, and the lower bound of a bucket's keys
struct sort_fast_map {
	uint64_t kmin; uint32_t shift, m, s2; double inv;
	__device__ __forceinline__ uint64_t rel(uint64_t rec) const { return (rec >> shift) - kmin; }
	__device__ __forceinline__ uint32_t bucket(uint64_t r) const { return (uint32_t)((r * m) >> s2); }
	// rel >= d * 2^s2 / m for every key of bucket d; one below the rounded quotient: the double product is exact to far less than 1
	__device__ __forceinline__ uint64_t lower(uint32_t d) const { if (m == 1u) return (uint64_t)d; const uint64_t q = (uint64_t)((double)d * inv); return q ? q - 1ull : 0ull; }
};

template <int W>
RH_DEV void sort_lane_net(uint32_t (&c)[W])
{
#pragma unroll
	for (int kk = 2; kk <= W; kk <<= 1)
#pragma unroll
		for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
			for (int i = 0; i < W; ++i) {
				const int l2 = i ^ jj;
				if (l2 > i) { const uint32_t lo = c[i] < c[l2] ? c[i] : c[l2], hi = c[i] < c[l2] ? c[l2] : c[i]; if ((i & kk) == 0) { c[i] = lo; c[l2] = hi; } else { c[i] = hi; c[l2] = lo; } }
			}
}
// one bucket [st, st + m) per lane, m <= W <= 16; lo = lower bound of its keys (relative to the minimum): key - lo < 2^26 + 2.
// Straight-line code, no lane-dependent branches: the W slots from st are read whatever m is (rec[] is padded), the slots beyond m take words that sort
// behind every key and differ from one another in their upper 27 bits (no false "equal neighbours"), the stores of the slots beyond m go to a slot of the
// thread's own behind the array.
template <int CAP, int W>
RH_DEV bool sort_fast_bucket(sort_fast_lds<CAP> &F, uint32_t st, uint32_t m, const sort_fast_map &M, uint64_t lo)
{
	uint32_t c[W];
	const uint32_t lo32 = (uint32_t)(lo + M.kmin);                // (the difference is below 2^27: 32-bit arithmetic on the low words)
	const uint64_t *b = F.rec + st;
#pragma unroll
	for (int j = 0; j < W; ++j) {
		const uint32_t w = ((uint32_t)(b[j] >> M.shift) - lo32) << 5 | (uint32_t)j;
		c[j] = (uint32_t)j < m ? w : ((0x7FFFFFFu - (uint32_t)j) << 5 | (uint32_t)j);
	}
	sort_lane_net<W>(c);
	uint32_t mind = 0xFFFFFFFFu;                                   // equal keys = neighbours that agree above bit 4
#pragma unroll
	for (int j = 1; j < W; ++j) { const uint32_t x = c[j] ^ c[j - 1]; mind = x < mind ? x : mind; }
	uint64_t nx[W];
#pragma unroll
	for (int j = 0; j < W; ++j) nx[j] = b[c[j] & 31u];
	uint64_t *dummy = F.rec + CAP + 16 + threadIdx.x;
#pragma unroll
	for (int j = 0; j < W; ++j) { uint64_t *q = (uint32_t)j < m ? F.rec + st + j : dummy; *q = nx[j]; }
	return mind < 32u;
}
// one bucket [st, st + m) of 17 .. 64 E records per WAVEFRONT: words (key - lower bound) << 16 | place, element i in lane i & 63, register i >> 6; bitonic
// network, the steps between lanes by shuffles
template <int CAP, int E>
RH_DEV bool sort_fast_wave(sort_fast_lds<CAP> &F, uint32_t st, uint32_t m, const sort_fast_map &M, uint64_t lo)
{
	const uint32_t lane = lane_id();
	const uint32_t lo32 = (uint32_t)(lo + M.kmin);
	uint64_t w[E];
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const uint32_t i = lane + 64u * (uint32_t)e;
		w[e] = ~0ull;
		if (i < m) w[e] = (uint64_t)((uint32_t)(F.rec[st + i] >> M.shift) - lo32) << 16 | (uint64_t)i;
	}
#pragma unroll
	for (int k = 2; k <= 64 * E; k <<= 1) {
#pragma unroll
		for (int j = k >> 1; j > 0; j >>= 1) {
			if (j >= 64) {	// partners in the same lane
#pragma unroll
				for (int e = 0; e < E; ++e) {
					const int pe = e ^ (j >> 6);
					if (pe > e) {
						const bool up = ((64 * e) & k) == 0;
						const uint64_t lo_ = w[e] < w[pe] ? w[e] : w[pe], hi_ = w[e] < w[pe] ? w[pe] : w[e];
						w[e] = up ? lo_ : hi_; w[pe] = up ? hi_ : lo_;
					}
				}
			} else {
#pragma unroll
				for (int e = 0; e < E; ++e) {
					const uint32_t i = lane + 64u * (uint32_t)e;
					const bool up = (i & (uint32_t)k) == 0, keep_min = ((lane & (uint32_t)j) == 0) == up;
					const uint64_t p = __shfl_xor(w[e], j);
					const uint64_t lo_ = w[e] < p ? w[e] : p, hi_ = w[e] < p ? p : w[e];
					w[e] = keep_min ? lo_ : hi_;
				}
			}
		}
	}
	bool tie = false;
	uint64_t nx[E];
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const uint32_t i = lane + 64u * (uint32_t)e;
		uint64_t prev = __shfl_up(w[e], 1);
		if (e > 0) { const uint64_t last = __shfl(w[e > 0 ? e - 1 : 0], 63); if (lane == 0) prev = last; }
		if (i > 0 && i < m && (prev >> 16) == (w[e] >> 16)) tie = true;
		nx[e] = i < m ? F.rec[st + (uint32_t)(w[e] & 0xFFFFu)] : 0ull;
	}
	RH_WAVE_SYNC();                                               // every lane has read the bucket before it is rewritten
#pragma unroll
	for (int e = 0; e < E; ++e) { const uint32_t i = lane + 64u * (uint32_t)e; if (i < m) F.rec[st + i] = nx[e]; }
	return __ballot(tie) != 0;
}

#ifdef RH_KPROF
#define FPROF(slot) do { if (prof && threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&rh_kprof_acc[slot], t_ - fp_t0); fp_t0 = t_; } } while (0)
#else
#define FPROF(slot)
#endif
// R16: 16-byte records {x = key, y = payload} (the region sort).  What travels through LDS is then a WORD per record, (key - least key) << 16 | place in
// the segment - the same 8 bytes, sorted the same way with "shift" 16 and "least key" 0 - and the records themselves are fetched once, at the end, by
// the places the sorted words name (the segment was read microseconds ago: L2).  Needs a key range below 2^48.
template <int CAP, bool R16>
RH_DEV int sort_fast(sort_fast_lds<CAP> &F, const void *src_, void *dst_, uint32_t n, uint32_t shift, bool write_tied, bool prof)
{
#ifdef RH_KPROF
	unsigned long long fp_t0 = clock64();
#endif
	typedef sort_fast_cfg<CAP> CF;
	constexpr int K = CF::K, NB = CF::NB, BPT = CF::BPT, LB = CF::LB;
	constexpr int KH = CF::HOLD ? K : 1;
	const uint32_t tid = threadIdx.x;
	const uint64_t *src8 = reinterpret_cast<const uint64_t*>(src_);
	const rh_mm128_t *src16 = reinterpret_cast<const rh_mm128_t*>(src_);
	if (n < 2) { if (n == 1 && tid == 0) { if (R16) reinterpret_cast<rh_mm128_t*>(dst_)[0] = src16[0]; else reinterpret_cast<uint64_t*>(dst_)[0] = src8[0]; } return 0; }
	// raw(i): what a thread keeps of record i - the record itself, or the key of a 16-byte one; keyof(raw): its sort key; wordof(raw, i): what goes to LDS
	#define SF_RAW(i_) (R16 ? src16[(i_)].x : src8[(i_)])
	#define SF_KEY(raw_) (R16 ? (raw_) : (raw_) >> shift)
	uint64_t r[KH];
	uint64_t kmin = ~0ull, kmax = 0ull;
	if constexpr (CF::HOLD) {
#pragma unroll
		for (int k = 0; k < KH; ++k) { const uint32_t i = tid + (uint32_t)k * NT; r[k] = i < n ? SF_RAW(i) : 0ull; }
#pragma unroll
		for (int k = 0; k < KH; ++k) { const uint32_t i = tid + (uint32_t)k * NT; if (i < n) { const uint64_t kk = SF_KEY(r[k]); kmin = kk < kmin ? kk : kmin; kmax = kk > kmax ? kk : kmax; } }
	} else {
		for (uint32_t i0 = tid; i0 < n; i0 += 8u * NT) {
			uint64_t x[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) { const uint32_t i = i0 + (uint32_t)u * NT; x[u] = i < n ? SF_RAW(i) : 0ull; }
#pragma unroll
			for (int u = 0; u < 8; ++u) { const uint32_t i = i0 + (uint32_t)u * NT; if (i < n) { const uint64_t kk = SF_KEY(x[u]); kmin = kk < kmin ? kk : kmin; kmax = kk > kmax ? kk : kmax; } }
		}
	}
	for (uint32_t b = tid; b < (uint32_t)NB; b += NT) F.cnt[b] = 0;
	if (tid == 0) { F.flag = 0; F.n_big = 0; }
	for (int d = 32; d > 0; d >>= 1) { const uint64_t a = __shfl_xor(kmin, d), b = __shfl_xor(kmax, d); kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; }
	if (lane_id() == 0) { F.r64[wave_id()] = kmin; F.r64[NT / 64 + wave_id()] = kmax; }
	__syncthreads();
#pragma unroll
	for (int q = 0; q < NT / 64; ++q) { const uint64_t a = F.r64[q], b = F.r64[NT / 64 + q]; kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; }
	FPROF(13);
	const uint64_t range = kmax - kmin;
	if (range == 0) return 1;                                    // n >= 2 equal keys
	const int bits = 64 - __clzll(range);
	if (R16 && bits > 48) return 2;                               // (key - least key) << 16 | place would not fit the word
	sort_fast_map M;
	M.kmin = R16 ? 0ull : kmin; M.shift = R16 ? 16u : shift; M.m = 1u; M.s2 = 0u; M.inv = 1.0;
	#define SF_WORD(raw_, i_) (R16 ? ((raw_) - kmin) << 16 | (uint64_t)(i_) : (raw_))
	if (bits > LB) {	// (range + 1) * m / 2^s2 <= NB with m the largest of 16 .. 31 that keeps it so: the last bucket used is NB * m / (m + 1) or later
		M.s2 = (uint32_t)(bits + 4 - LB);
		if (M.s2 > 30u) return 2;                                 // a bucket's keys would not fit 27 bits
		const uint64_t q = ((uint64_t)NB << M.s2) / (range + 1ull);
		M.m = q > 31ull ? 31u : (uint32_t)q;
		M.inv = (double)(1ull << M.s2) / (double)M.m;
	}
	// bucket and rank of every record: one atomic each
	uint32_t dr[K];
	if constexpr (CF::HOLD) {
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const uint32_t i = tid + (uint32_t)k * NT;
			dr[k] = 0;
			if (i < n) { const uint32_t d = M.bucket(M.rel(SF_WORD(r[k], i))); dr[k] = d | atomicAdd(&F.cnt[d], 1u) << 16; }
		}
	} else {
#pragma unroll
		for (int k0 = 0; k0 < K; k0 += 8) {
			uint64_t x[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) { const uint32_t i = tid + (uint32_t)(k0 + u) * NT; x[u] = (k0 + u < K && i < n) ? SF_RAW(i) : 0ull; }
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				if (k0 + u >= K) continue;
				const uint32_t i = tid + (uint32_t)(k0 + u) * NT;
				dr[k0 + u] = 0;
				if (i < n) { const uint32_t d = M.bucket(M.rel(SF_WORD(x[u], i))); dr[k0 + u] = d | atomicAdd(&F.cnt[d], 1u) << 16; }
			}
		}
	}
	__syncthreads();
	FPROF(14);
	// counts -> start offsets; this thread's buckets stay in its registers for step 5
	uint32_t bc[BPT], bs[BPT], sum = 0;
#pragma unroll
	for (int q = 0; q < BPT; ++q) { bc[q] = F.cnt[tid * (uint32_t)BPT + (uint32_t)q]; sum += bc[q]; }
	uint32_t tot;
	uint32_t ex = block_excl_scan(sum, F.w, tot);                 // (its first barrier: every thread has read its counts)
#pragma unroll
	for (int q = 0; q < BPT; ++q) { bs[q] = ex; ex += bc[q]; F.cnt[tid * (uint32_t)BPT + (uint32_t)q] = bs[q]; }
	// (a bucket of more than 16 records - the tail of the distribution, or a mapped read's anchors at its locus - is a wavefront's, below; more or larger ones than the list takes: not for this path)
#pragma unroll
	for (int q = 0; q < BPT; ++q) if (bc[q] > 16u) {
		const uint32_t k = bc[q] <= (uint32_t)sort_fast_big<CAP>::MAX ? atomicAdd(&F.n_big, 1u) : (uint32_t)SORT_FAST_BIG;
		if (k < (uint32_t)SORT_FAST_BIG) { F.big[k] = bs[q] | bc[q] << 16; const uint64_t lo = M.lower(tid * (uint32_t)BPT + (uint32_t)q); F.big_lo[2 * k] = (uint32_t)lo; F.big_lo[2 * k + 1] = (uint32_t)(lo >> 32); }
		else F.flag = 1;
	}
	__syncthreads();
	FPROF(15);
	if (F.flag) return 2;
	// scatter
	if constexpr (CF::HOLD) {
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const uint32_t i = tid + (uint32_t)k * NT;
			if (i < n) F.rec[F.cnt[dr[k] & 0xFFFFu] + (dr[k] >> 16)] = SF_WORD(r[k], i);
		}
	} else {
#pragma unroll
		for (int k0 = 0; k0 < K; k0 += 8) {
			uint64_t x[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) { const uint32_t i = tid + (uint32_t)(k0 + u) * NT; x[u] = (k0 + u < K && i < n) ? SF_RAW(i) : 0ull; }
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				if (k0 + u >= K) continue;
				const uint32_t i = tid + (uint32_t)(k0 + u) * NT;
				if (i < n) F.rec[F.cnt[dr[k0 + u] & 0xFFFFu] + (dr[k0 + u] >> 16)] = SF_WORD(x[u], i);
			}
		}
	}
	__syncthreads();
	FPROF(16);
	// every lane sorts its buckets
	bool tie = false;
#pragma unroll
	for (int q = 0; q < BPT; ++q) {
		const uint32_t m = bc[q] > 16u ? 0u : bc[q];
		const uint64_t lo = M.lower(tid * (uint32_t)BPT + (uint32_t)q);
		if (__ballot(m > 8u)) tie |= sort_fast_bucket<CAP, 16>(F, bs[q], m, M, lo);
		else if (__ballot(m > 1u)) tie |= sort_fast_bucket<CAP, 8>(F, bs[q], m, M, lo);
	}
	{
		const uint32_t nbig = F.n_big;                                // (written before the barriers above)
		for (uint32_t q = wave_id(); q < nbig; q += NT / 64) {
			const uint32_t e = rh_uniform(F.big[q]), st = e & 0xFFFFu, m = e >> 16;
			const uint64_t lo = (uint64_t)rh_uniform(F.big_lo[2 * q]) | (uint64_t)rh_uniform(F.big_lo[2 * q + 1]) << 32;
			bool t2;
			if (m <= 64u) t2 = sort_fast_wave<CAP, 1>(F, st, m, M, lo);
			else if (m <= 128u) t2 = sort_fast_wave<CAP, 2>(F, st, m, M, lo);
			else if (m <= 256u || sort_fast_big<CAP>::EMAX < 8) t2 = sort_fast_wave<CAP, 4>(F, st, m, M, lo);
			else t2 = sort_fast_wave<CAP, sort_fast_big<CAP>::EMAX>(F, st, m, M, lo);
			tie |= t2;
		}
	}
	if (tie) F.flag = 1;
	__syncthreads();
	FPROF(17);
	const bool tied = F.flag != 0;
	if (tied && !write_tied) return 1;                             // (write_tied: the caller only wants to know - the segment is sorted, its equal keys in no particular order)
	if (R16) { rh_mm128_t *dst16 = reinterpret_cast<rh_mm128_t*>(dst_); for (uint32_t i = tid; i < n; i += NT) dst16[i] = src16[(uint32_t)F.rec[i] & 0xFFFFu]; }
	else { uint64_t *dst8 = reinterpret_cast<uint64_t*>(dst_); for (uint32_t i = tid; i < n; i += NT) dst8[i] = F.rec[i]; }
	FPROF(18);
	#undef SF_RAW
	#undef SF_KEY
	#undef SF_WORD
	return tied ? 1 : 0;
}

// mode 0: fast pass; reads whose sorted keys show ties are redone with the exact permutation on the tied ranges
// mode 2: exact pass on every range (keys known to be full of ties, e.g. chain scores)
// workgroups of a class that fit the 160 KB of LDS of a CU (1 KB allocation granules) = wavefronts per SIMD the compiler
// has to leave registers for (4 wavefronts per workgroup, 4 SIMDs per CU).  Round 6: five, not six, for the classes up to 2048 records - with 80 registers the
// tie-free path spilled 100 bytes a lane; with 96 it does not (k_sort_block<2048> 60 -> 56 ms)
template <int CAP, class KT>
constexpr int sort_wg_per_cu() { return (int)((160u * 1024u) / ((sizeof(sort_lds<CAP, KT>) + 1023u) / 1024u * 1024u)) < (CAP <= 2048 ? 5 : 4) ? (int)((160u * 1024u) / ((sizeof(sort_lds<CAP, KT>) + 1023u) / 1024u * 1024u)) : (CAP <= 2048 ? 5 : 4); }

template <int CAP, class KT, class REC>
__global__ __launch_bounds__(NT, (sort_wg_per_cu<CAP, KT>())) void k_sort_block(rh_sort_job jb, uint32_t n_lo, uint32_t n_hi, int mode)
{
	constexpr bool FAST = sizeof(sort_fast_lds<CAP>) <= sizeof(sort_lds<CAP, KT>) + 2048;   // (the tie-free path shares the LDS of the general one)
	__shared__ union U_ { sort_lds<CAP, KT> L; typename std::conditional<FAST, sort_fast_lds<CAP>, uint32_t>::type F; } U;
	sort_lds<CAP, KT> &L = U.L;
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	if (a >= jb.n_seg || (jb.skip && jb.skip[a])) return;
	const uint64_t base = jb.off[a];
	const uint32_t n = jb.cnt ? jb.cnt[a] : (uint32_t)(jb.off[a + 1] - base);
	if (n <= n_lo || n > n_hi) return;
	const REC *src = reinterpret_cast<const REC*>(jb.src) + base;
	REC *dst = reinterpret_cast<REC*>(jb.dst) + base;
	const rh_rec_fmt rf = jb.rf;
	KPROF_DECL;
	if constexpr (FAST) {
		if (mode == 0 && jb.fast_on) {	// segments without equal keys (nearly all): records in LDS, one pass into CAP / 2 buckets, a register network per bucket
			const int fr = sort_fast<CAP, sizeof(REC) == 16>(U.F, src, dst, n, rf.shift, jb.no_redo != 0, jb.scratch_skip == 0);
#ifdef RH_KPROF
			kp_t0 = clock64();
			if (tid == 0 && jb.scratch_skip == 0) { atomicAdd(&rh_kprof_acc[20 + fr], 1ull); atomicAdd(&rh_kprof_acc[23], (unsigned long long)n); }   // outcomes of the tie-free path: done / equal keys / not for it; records seen
#endif
#ifdef RH_FAST_TRACE
			if (tid == 0) fprintf(stderr, "FAST cap %d n %u -> %d\n", CAP, n, fr);
#endif
			if (fr == 0 || (fr == 1 && jb.no_redo)) { if (tid == 0 && jb.need_exact) jb.need_exact[a] = (uint8_t)fr; return; }   // (no_redo: whoever asked redoes the segments that hold equal keys from their input anyway)
			__syncthreads();
		}
	}
#ifdef RH_KPROF
	if (tid == 0) L.prof = jb.scratch_skip == 0 ? 1u : 0u;      // profile the anchor sort only
	__syncthreads();
#endif
	const sort_kc kc = { jb.kc_lo, jb.kc_mid, jb.kc_hi };
	if (tid == 0) L.kc = kc;
	// (every load of a thread in flight at once, up to 16: a workgroup's wall time is round trips to HBM, and what its LDS footprint
	// lets the CU overlap with three others)
	{
		constexpr int KL = (CAP + NT - 1) / NT < 16 ? (CAP + NT - 1) / NT : 16;
		for (uint32_t i0 = tid; i0 < n; i0 += (uint32_t)KL * NT) {
			uint64_t xs[KL];
#pragma unroll
			for (int u = 0; u < KL; ++u) { const uint32_t i = i0 + (uint32_t)u * NT; xs[u] = i < n ? rh_rec_ops<REC>::key(src[i], rf) : 0ull; }
#pragma unroll
			for (int u = 0; u < KL; ++u) {
				const uint32_t i = i0 + (uint32_t)u * NT;
				const uint64_t x = xs[u];
				if (i >= n) continue;
				if (sizeof(KT) == 8) L.key[i] = (KT)x;
				else L.key[i] = (KT)((x & ((1ull << kc.lo_bits) - 1ull)) | ((x >> 32) & ((1ull << kc.mid_bits) - 1ull)) << kc.lo_bits | (kc.hi_bits ? x >> 63 : 0ull) << (kc.lo_bits + kc.mid_bits));
			}
		}
	}
	for (uint32_t i = tid; i < CAP / 32 + 3; i += NT) L.tbit[i] = 0;
	if (tid == 0) { L.tie = 0; L.n_tg = 0; }
	__syncthreads();
	KPROF(10);
	sort_run<CAP, KT>(L, n, mode == 0 ? SORT_FAST : SORT_EXACT_ALL);
#ifdef RH_KPROF
	kp_t0 = clock64();
#endif
	// equal keys among the sorted neighbours?  (before the write-out below takes the key array as its staging area)
	uint32_t tie = 0;
	if (mode == 0) {
		for (uint32_t i = tid; i < n; i += NT) {
			const uint32_t idx = L.ia[i];
			const KT k = L.key[idx];
			if ((i > 0 && L.key[L.ia[i - 1]] == k) || (i + 1 < n && L.key[L.ia[i + 1]] == k)) {
				atomicOr(&L.tbit[idx >> 5], 1u << (idx & 31u));
				L.tie = 1;
				uint32_t gs = i;
				while (gs > 0 && L.key[L.ia[gs - 1]] == k) --gs;
				const uint32_t slot = atomicAdd(&L.n_tg, 1u);
				if (slot < SORT_TG) { L.tg_idx[slot] = (uint16_t)idx; L.tg_pos[slot] = (uint16_t)gs; L.tg_fin[slot] = 0xFFFFu; }
			}
		}
		__syncthreads();
		tie = L.tie;
		if (tid == 0 && jb.need_exact) jb.need_exact[a] = (uint8_t)tie;
		KPROF(11);
	}
	// Write-out.  dst[i] = src[ia[i]] straight from HBM is a 16-byte gather per record: with a thousand workgroups in flight the
	// segments (64 KB each) have left the L2 by now, and random 64-byte sectors come in at a fraction of the sequential rate.  So
	// the records are streamed once more, in order, through LDS - the key / arrangement / scratch arrays are done with and hold
	// STG records at a time - and every thread picks the ones its output positions want (kept in registers) from there.
	{
		constexpr int K = (CAP + NT - 1) / NT;
		constexpr uint32_t STG = (uint32_t)((sizeof(KT) + 4u) * (size_t)CAP / sizeof(REC));
		uint16_t iav[K];
#pragma unroll
		for (int k = 0; k < K; ++k) { const uint32_t i = tid + (uint32_t)k * NT; iav[k] = i < n ? L.ia[i] : (uint16_t)0xFFFFu; }
		__syncthreads();
		REC *stage = reinterpret_cast<REC*>(&L);
		for (uint32_t sb = 0; sb < n; sb += STG) {
			const uint32_t m = n - sb < STG ? n - sb : STG;
			constexpr int KS = (int)((STG + NT - 1) / NT) < 8 ? (int)((STG + NT - 1) / NT) : 8;   // loads in flight per thread
			for (uint32_t i0 = tid; i0 < m; i0 += (uint32_t)KS * NT) {
				REC rv[KS];
#pragma unroll
				for (int u = 0; u < KS; ++u) { const uint32_t i = i0 + (uint32_t)u * NT; rv[u] = src[sb + (i < m ? i : 0u)]; }
#pragma unroll
				for (int u = 0; u < KS; ++u) { const uint32_t i = i0 + (uint32_t)u * NT; if (i < m) stage[i] = rv[u]; }
			}
			__syncthreads();
#pragma unroll
			for (int k = 0; k < K; ++k) {
				const uint32_t rel = (uint32_t)iav[k] - sb;                 // (0xFFFF: no output position - never below sb + m, n <= CAP < 0xFFFF)
				if (iav[k] != 0xFFFFu && rel < m) dst[tid + (uint32_t)k * NT] = stage[rel];
			}
			__syncthreads();
		}
	}
	KPROF(12);
	if (mode != 0 || !tie || jb.no_redo) return;
	// Equal keys: their order is the reference's cycle-leader permutation.  Redo the sort from the input order on the ranges
	// that hold tied keys only; every other record already sits at its final place, and so does each group of equal keys
	// as a whole - only the records inside the groups are rewritten.  (The keys again: the staging above overwrote them.)
	for (uint32_t i = tid; i < n; i += NT) {
		const uint64_t x = rh_rec_ops<REC>::key(src[i], rf);
		if (sizeof(KT) == 8) L.key[i] = (KT)x;
		else L.key[i] = (KT)((x & ((1ull << kc.lo_bits) - 1ull)) | ((x >> 32) & ((1ull << kc.mid_bits) - 1ull)) << kc.lo_bits | (kc.hi_bits ? x >> 63 : 0ull) << (kc.lo_bits + kc.mid_bits));
	}
	__syncthreads();
	sort_run<CAP, KT>(L, n, SORT_EXACT_TIED);
	for (uint32_t i = tid; i < n; i += NT) { const uint32_t idx = L.ia[i]; if ((L.tbit[idx >> 5] >> (idx & 31u)) & 1u) dst[i] = src[idx]; }
	if (tid < SORT_TG && tid < L.n_tg && L.tg_fin[tid] != 0xFFFFu) dst[L.tg_fin[tid]] = src[L.tg_idx[tid]];   // settled by pop order
}

// Segments of up to RH_SORT_TINY records - klib sorts a range of <= 64 with its insertion sort (ksort.h:139-151), i.e. any STABLE
// sort reproduces it, equal keys included.  A workgroup per such segment is a launch of millions of workgroups that each move a
// dozen records (the region sort of an unmappable read ends in ~6000 buckets of ~10 chains): here ONE LANE takes a segment.  Keys
// that agree above bit 26 (a bucket of the level-by-level sorter: always) are sorted as words (key bits << 5 | position) by a
// bitonic network in the lane's registers, the position in the low bits making the order stable; other keys are ranked by
// counting.  Records are read through the L1 (a wavefront's segments are neighbours in memory) and written once.
#define RH_SORT_TINY 32
template <class REC>
__global__ __launch_bounds__(NT) void k_sort_tiny(rh_sort_job jb, uint32_t n_lo)
{
	const rh_rec_fmt rf = jb.rf;
	const uint32_t a = blockIdx.x * NT + threadIdx.x;
	uint32_t n = 0;
	uint64_t base = 0;
	if (a < jb.n_seg && !(jb.skip && jb.skip[a])) { base = jb.off[a]; n = jb.cnt ? jb.cnt[a] : (uint32_t)(jb.off[a + 1] - base); }
	const bool mine = n > n_lo && n <= (uint32_t)RH_SORT_TINY;
	if (__ballot(mine) == 0) return;
	const REC *src = reinterpret_cast<const REC*>(jb.src) + base;
	REC *dst = reinterpret_cast<REC*>(jb.dst) + base;
	uint32_t c[RH_SORT_TINY];
	uint64_t k0 = 0, dif = 0;
#pragma unroll
	for (int j = 0; j < RH_SORT_TINY; ++j) {
		c[j] = 0xFFFFFFFFu;
		if (mine && (uint32_t)j < n) {
			const uint64_t k = rh_rec_ops<REC>::key(src[j], rf);
			if (j == 0) k0 = k;
			dif |= k ^ k0;
			c[j] = (uint32_t)k << 5 | (uint32_t)j;
		}
	}
	const bool narrow = mine && (dif >> 27) == 0, wide = mine && !narrow;
	bool tie = false;
	if (__ballot(narrow)) {
		if (__ballot(narrow && n > 16)) {
#pragma unroll
			for (int kk = 2; kk <= 32; kk <<= 1)
#pragma unroll
				for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
					for (int i = 0; i < 32; ++i) {
						const int l2 = i ^ jj;
						if (l2 > i) { const uint32_t lo = c[i] < c[l2] ? c[i] : c[l2], hi = c[i] < c[l2] ? c[l2] : c[i]; if ((i & kk) == 0) { c[i] = lo; c[l2] = hi; } else { c[i] = hi; c[l2] = lo; } }
					}
		} else {
#pragma unroll
			for (int kk = 2; kk <= 16; kk <<= 1)
#pragma unroll
				for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
					for (int i = 0; i < 16; ++i) {
						const int l2 = i ^ jj;
						if (l2 > i) { const uint32_t lo = c[i] < c[l2] ? c[i] : c[l2], hi = c[i] < c[l2] ? c[l2] : c[i]; if ((i & kk) == 0) { c[i] = lo; c[l2] = hi; } else { c[i] = hi; c[l2] = lo; } }
					}
		}
		if (narrow) {
#pragma unroll
			for (int j = 0; j < RH_SORT_TINY; ++j) if ((uint32_t)j < n) {
				dst[j] = src[c[j] & 31u];
				if (j > 0 && (c[j] >> 5) == (c[j - 1] >> 5)) tie = true;
			}
		}
	}
	if (wide) {	// (rare: keys of a free-standing short segment) rank = records that sort before this one, earlier position first among equals
		for (uint32_t j = 0; j < n; ++j) {
			const REC rj = src[j];
			const uint64_t kj = rh_rec_ops<REC>::key(rj, rf);
			uint32_t rank = 0;
			for (uint32_t i = 0; i < n; ++i) { const uint64_t ki = rh_rec_ops<REC>::key(src[i], rf); rank += (ki < kj || (ki == kj && i < j)) ? 1u : 0u; if (ki == kj && i != j) tie = true; }
			dst[rank] = rj;
		}
	}
	if (mine && jb.need_exact) jb.need_exact[a] = tie ? 1 : 0;
}

// Buckets of the level-by-level sorter with 33 .. 256 records (the region sort of an unmappable read ends in hundreds of buckets
// of ~100 chains: millions of them a round): ONE WAVEFRONT per bucket instead of a workgroup with its barriers.  A bucket's keys
// agree above the byte it was split on; when they agree above bit 23 and no two are equal, the sorted order is unique and any
// sort gives the reference's result: the lanes hold four words (key bits << 8 | position) each and run a bitonic network
// (cross-lane steps by ds_bpermute).  Equal keys (adjacent after the sort) or wider keys: the bucket is left, untouched, to the LDS
// block sorter and its exact passes; a finished bucket's count is zeroed so that those launches pass over it.
#define RH_SORT_WAVE 256
template <class REC>
__global__ __launch_bounds__(NT) void k_sort_wave(rh_sort_job jb, uint32_t n_lo)
{
	const rh_rec_fmt rf = jb.rf;
	constexpr int E = RH_SORT_WAVE / 64;
	const uint32_t a = blockIdx.x * (NT / 64) + wave_id(), lane = lane_id();
	if (a >= jb.n_seg) return;
	const uint32_t n = rh_uniform(jb.cnt_rw[a]);
	if (n <= n_lo || n > (uint32_t)RH_SORT_WAVE) return;
	const uint64_t base = jb.off[a];
	const REC *src = reinterpret_cast<const REC*>(jb.src) + base;
	REC *dst = reinterpret_cast<REC*>(jb.dst) + base;
	uint32_t w[E];
	uint64_t dif = 0;
	const uint64_t first = rh_rec_ops<REC>::key(src[0], rf);
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const uint32_t i = lane + 64u * (uint32_t)e;
		w[e] = 0xFFFFFFFFu;
		if (i < n) { const uint64_t k = rh_rec_ops<REC>::key(src[i], rf); dif |= k ^ first; w[e] = (uint32_t)k << 8 | i; }
	}
	if (__ballot((dif >> 24) != 0)) return;                          // wider keys: the block sorter
#pragma unroll
	for (int k = 2; k <= RH_SORT_WAVE; k <<= 1) {
#pragma unroll
		for (int j = k >> 1; j > 0; j >>= 1) {
			if (j >= 64) {	// partners in the same lane
#pragma unroll
				for (int e = 0; e < E; ++e) {
					const int pe = e ^ (j >> 6);
					if (pe > e) {
						const bool up = ((64 * e) & k) == 0;
						const uint32_t lo = w[e] < w[pe] ? w[e] : w[pe], hi = w[e] < w[pe] ? w[pe] : w[e];
						w[e] = up ? lo : hi; w[pe] = up ? hi : lo;
					}
				}
			} else {
#pragma unroll
				for (int e = 0; e < E; ++e) {
					const uint32_t i = lane + 64u * (uint32_t)e;
					const bool up = (i & (uint32_t)k) == 0, keep_min = ((lane & (uint32_t)j) == 0) == up;
					const uint32_t p = __shfl_xor(w[e], j);
					const uint32_t lo = w[e] < p ? w[e] : p, hi = w[e] < p ? p : w[e];
					w[e] = keep_min ? lo : hi;
				}
			}
		}
	}
	bool tie = false;
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const uint32_t i = lane + 64u * (uint32_t)e;
		uint32_t prev = __shfl_up(w[e], 1);
		if (e > 0) { const uint32_t last = rh_readlane(w[e > 0 ? e - 1 : 0], 63u); if (lane == 0) prev = last; }
		if (i > 0 && i < n && (prev >> 8) == (w[e] >> 8)) tie = true;
	}
	if (__ballot(tie)) return;                                       // equal keys: the exact passes of the block sorter
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const uint32_t i = lane + 64u * (uint32_t)e;
		if (i < n) dst[i] = src[w[e] & 255u];
	}
	if (lane == 0) jb.cnt_rw[a] = 0;
}

template <int CAP, class KT>
static void launch_class(hipStream_t s, const rh_sort_job &jb, bool all_exact, uint32_t lo, uint32_t hi)
{
	if (jb.n_max && lo >= jb.n_max) return;                       // no segment reaches this class
	if (jb.rf.rec8) RH_LAUNCH((k_sort_block<CAP, KT, uint64_t>), jb.n_seg, NT, 0, s, jb, lo, hi, all_exact ? 2 : 0);
	else RH_LAUNCH((k_sort_block<CAP, KT, rh_mm128_t>), jb.n_seg, NT, 0, s, jb, lo, hi, all_exact ? 2 : 0);
}

static bool sort_keys32(const rh_sort_job &jb) { return jb.kc_on && (uint32_t)jb.kc_lo + jb.kc_mid + jb.kc_hi <= 32u && jb.kc_mid <= 24u; }

uint32_t rhk_sort_lds_max(const rh_sort_job &jb) { return sort_keys32(jb) ? (uint32_t)RH_SORT32_CAP3 : (uint32_t)RH_SORT_CAP4; }

int rhk_sort_job(hipStream_t s, const rh_sort_job &job, bool all_exact, uint32_t min_n)
{
	if (!job.n_seg) return 0;
	rh_sort_job jb = job;
	static const bool fast_on = !(getenv("RH_SORT_FAST") && atoi(getenv("RH_SORT_FAST")) == 0);   // RH_SORT_FAST=0: the general LDS path for every segment (A/B aid)
	jb.fast_on = fast_on && !all_exact && !job.tie_path ? 1 : 0;   // (an exact re-run's buckets are the ones that hold equal keys - the others are dropped, rh_bigsort.hip - so the tie-free path would only find out again)
	uint32_t top;
	static const bool tiny_on = !(RH_DEVENV("RH_SORT_TINY") && atoi(RH_DEVENV("RH_SORT_TINY")) == 0);
	if (tiny_on && min_n < (uint32_t)RH_SORT_TINY) {	// one lane per segment of up to 32 records
		if (jb.rf.rec8) RH_LAUNCH(k_sort_tiny<uint64_t>, (jb.n_seg + NT - 1) / NT, NT, 0, s, jb, min_n);
		else RH_LAUNCH(k_sort_tiny<rh_mm128_t>, (jb.n_seg + NT - 1) / NT, NT, 0, s, jb, min_n);
		min_n = (uint32_t)RH_SORT_TINY;
		if (jb.n_max && jb.n_max <= min_n) return 0;
	}
	static const bool wave_on = !(RH_DEVENV("RH_SORT_WAVE") && atoi(RH_DEVENV("RH_SORT_WAVE")) == 0);
	if (wave_on && jb.cnt_rw && !jb.skip && min_n < (uint32_t)RH_SORT_WAVE) {
		if (jb.rf.rec8) RH_LAUNCH(k_sort_wave<uint64_t>, (jb.n_seg + NT / 64 - 1) / (NT / 64), NT, 0, s, jb, min_n);
		else RH_LAUNCH(k_sort_wave<rh_mm128_t>, (jb.n_seg + NT / 64 - 1) / (NT / 64), NT, 0, s, jb, min_n);
	}   // a wavefront per bucket of up to 256 records (the rest, and its leftovers, below)
	if (sort_keys32(jb)) {
		launch_class<RH_SORT_CAP0, uint32_t>(s, jb, all_exact, min_n, (uint32_t)RH_SORT_CAP0);
		launch_class<RH_SORT32_CAPH, uint32_t>(s, jb, all_exact, (uint32_t)RH_SORT_CAP0, (uint32_t)RH_SORT32_CAPH);
		launch_class<RH_SORT32_CAP1, uint32_t>(s, jb, all_exact, (uint32_t)RH_SORT32_CAPH, (uint32_t)RH_SORT32_CAP1);
		launch_class<RH_SORT32_CAP2, uint32_t>(s, jb, all_exact, (uint32_t)RH_SORT32_CAP1, (uint32_t)RH_SORT32_CAP2);
		launch_class<RH_SORT32_CAP3, uint32_t>(s, jb, all_exact, (uint32_t)RH_SORT32_CAP2, (uint32_t)RH_SORT32_CAP3);
		top = (uint32_t)RH_SORT32_CAP3;
	} else {
		launch_class<RH_SORT_CAP0, uint64_t>(s, jb, all_exact, min_n, (uint32_t)RH_SORT_CAP0);
		launch_class<RH_SORT_CAP1, uint64_t>(s, jb, all_exact, (uint32_t)RH_SORT_CAP0, (uint32_t)RH_SORT_CAP1);
		launch_class<RH_SORT_CAP2, uint64_t>(s, jb, all_exact, (uint32_t)RH_SORT_CAP1, (uint32_t)RH_SORT_CAP2);
		launch_class<RH_SORT_CAP3, uint64_t>(s, jb, all_exact, (uint32_t)RH_SORT_CAP2, (uint32_t)RH_SORT_CAP3);
		launch_class<RH_SORT_CAP4, uint64_t>(s, jb, all_exact, (uint32_t)RH_SORT_CAP3, (uint32_t)RH_SORT_CAP4);
		top = (uint32_t)RH_SORT_CAP4;
	}
	// longer segments: many workgroups per segment, level by level (rh_bigsort.hip)
	if (!jb.n_max || top < jb.n_max) return rhk_bigsort(s, jb, all_exact, top);
	return 0;
}

static void sort_scratch(rh_sort_job &jb, const rh_dev_round &r, rh_mm128_t *idle) { jb.big_alt = idle; jb.big_ws = r.sort_ws; jb.big_ws_bytes = r.sort_ws_bytes; jb.big_pin = r.sort_pin; jb.big_total = r.sort_total; }

// anchor sort of a chunk round: unsorted expand output -> reference order.
// Anchor keys (strand, target, position) are equal only where two seeds of the read's chunks share a hash - measured on the human-scale batch:
// 0.2 % of the reads in round 0, 15 % of the unmappable reads by round 9, carried anchors accumulating - and the sorted order of a segment
// without equal keys is unique.  So, when the caller can restore its input (`reexpand`), the segments beyond the LDS classes are first placed
// level by level in ANY order (no hole lists, no token walk: rh_sort_job::any_order), whoever finishes a bucket reports the segments that do hold equal
// keys, and only those are expanded again and sorted with the exact passes.  reexpand(mask): r.raw of every segment a with mask[a] == 0 as it
// was before the call.  Without it: the exact passes for all.
int rhk_sort(hipStream_t s, const rh_dev_index &ix, const rh_dev_round &r, const std::function<int(const uint8_t*)> &reexpand)
{
	rh_sort_job jb = { r.n_act, r.skip, r.a_off, nullptr, r.raw, r.anc, r.need_exact, r.ws, RH_WS_PER_ANCHOR, 0, r.akey_on, r.akey_lo, r.akey_mid, 1, r.max_anchors };
	sort_scratch(jb, r, r.zs);                                     // (the candidate array is idle until the chain DP has run)
	jb.kind = 1;
	jb.rf = r.afmt;                                                // (one-word anchors stay one word: r.raw -> r.anc as uint64_t arrays)
	(void)ix;
	static const bool exact_all = RH_DEVENV("RH_ASORT_EXACT") != nullptr;   // development aid: the exact passes for every read
	if (!reexpand || exact_all || (jb.n_max && jb.n_max <= rhk_sort_lds_max(jb))) return rhk_sort_job(s, jb, false, 0u);   // (nothing beyond the LDS classes: their fast pass / tie redo is exact already)
	uint32_t n_redo = 0;
	jb.any_order = 1; jb.redo_skip = r.need_exact2; jb.n_redo = &n_redo;
	if (jb.rf.rec8) jb.any_up = (uint8_t)(64u - ((uint32_t)jb.rf.mid + 1u));   // any order: the levels need not be the reference's bytes (rh_rec_fmt::up)
	RH_HIP(hipMemsetAsync(r.need_exact2, 1, r.n_act, s));
	if (rhk_sort_job(s, jb, false, 0u)) return -1;
	static const bool trace = RH_DEVENV("RH_BS_TRACE") != nullptr;
	if (trace) fprintf(stderr, "ASORT any-order: chunk %u: %u of %u reads hold equal anchor keys and are redone\n", r.chunk, n_redo, r.n_act);
	if (!n_redo) return 0;
	if (reexpand(r.need_exact2)) return -1;
	jb.any_order = 0; jb.redo_skip = nullptr; jb.n_redo = nullptr; jb.skip = r.need_exact2; jb.tie_path = 1;   // (covers r.skip: the check marks skipped segments "no redo")
	return rhk_sort_job(s, jb, false, 0u);
}
