// Anchor sort stage: block-cooperative, LDS-resident reproduction of klib's radix_sort_128x (reference ksort.h:101-151,
// called at rmap.cpp:121) for one read per workgroup.
//
// The reference sort is an in-place MSD "American flag" radix sort (8-bit digits from bit 56, insertion sort for
// ranges <= 64).  It is unstable, and the order it leaves equal keys in is observed by the chaining DP, so the result
// must be the reference's exact permutation.  Structure used here:
//   * ranges whose keys agree on a byte are untouched by that pass -> jump straight to the highest differing byte;
//   * the final insertion sorts (ranges <= 64) are stable and independent -> one lane per range, in parallel;
//   * a pass over >= 2 buckets is a permutation that only matters for EQUAL keys:
//       fast mode : digit scatter with LDS atomics (order inside a bucket arbitrary), then check the sorted result for
//                   adjacent equal keys; a read without ties has a unique sorted order, so it is already exact;
//       exact mode: (reads flagged by the fast pass, ~5 % on an E. coli-scale index) the reference's cycle-leader
//                   permutation: closed form for two buckets (prefix ranks), one lane walking the cycles otherwise.
// Keys live in LDS by original index (8 B); the permutation is carried as 16-bit indices.  Records (16 B) are gathered
// from / written to HBM once.
#include "rh_kernels.h"
#include "rh_devutil.h"

template <int CAP>
struct sort_lds {
	uint64_t key[CAP];
	uint16_t ia[CAP], ib[CAP], tmp[CAP];
	uint8_t db[CAP];
	uint32_t small[CAP / 2 + 2];               // ranges <= 64 awaiting the stable insertion sort: beg | end << 16
	uint32_t rng[2][CAP / 64 + 4];             // ranges > 64 still to be split: beg | end << 16
	uint8_t rsh[2][CAP / 64 + 4];              // ... and the byte shift they are to be split on next
	uint32_t cnt[256], head[256], tail[256];
	uint32_t w[NT / 64];
	uint64_t r64[NT / 64];
	uint32_t n_rng[2], n_small, tie, misc[4];
};

template <int CAP>
RH_DEV void sort_split_range(sort_lds<CAP> &L, uint32_t beg, uint32_t end, int shift, int nxt, int exact)
{
	const uint32_t tid = threadIdx.x;
	// highest byte on which the range's keys differ (passes above it are identities in the reference)
	const uint64_t k0 = L.key[L.ia[beg]];
	uint64_t diff = 0;
	for (uint32_t i = beg + tid; i < end; i += NT) diff |= L.key[L.ia[i]] ^ k0;
	diff = block_or64(diff, L.r64);
	if (diff == 0) return;                                   // all keys equal: every remaining pass is an identity
	int s = (63 - __clzll(diff)) & ~7;
	if (s > shift) s = shift;
	// digit histogram
	L.cnt[tid] = 0;
	__syncthreads();
	for (uint32_t i = beg + tid; i < end; i += NT) { const uint32_t d = (uint32_t)(L.key[L.ia[i]] >> s) & 255u; L.db[i] = (uint8_t)d; atomicAdd(&L.cnt[d], 1u); }
	__syncthreads();
	const uint32_t my_cnt = L.cnt[tid];
	uint32_t total;
	const uint32_t my_start = beg + block_excl_scan(my_cnt, L.w, total);
	L.head[tid] = my_start; L.tail[tid] = my_start + my_cnt;
	uint32_t nbk;
	(void)block_rank(my_cnt != 0, L.w, nbk);
	// permutation of the pass: ia[beg,end) -> ib[beg,end)
	if (!exact) {
		for (uint32_t i = beg + tid; i < end; i += NT) { const uint32_t pos = atomicAdd(&L.head[L.db[i]], 1u); L.ib[pos] = L.ia[i]; }
	} else if (nbk == 2) {
		// Two buckets A < B.  Cycle-leader result in closed form: the k-th misplaced element of region A trades places
		// with the k-th misplaced element of region B, except that in B every run of in-place elements between two
		// misplaced ones is shifted right by one slot and the arrival lands in front of the run.
		if (my_cnt != 0) { const uint32_t which = my_start == beg ? 0u : 1u; L.misc[which] = tid; L.misc[2 + which] = my_start; }
		__syncthreads();
		const uint32_t cA = L.misc[0], cB = L.misc[1], startB = L.misc[3];
		uint32_t m = 0;                                        // misplaced elements seen so far in A
		for (uint32_t base = beg; base < startB; base += NT) {
			const uint32_t i = base + tid;
			const bool foreign = i < startB && L.db[i] == cB;
			uint32_t tot;
			const uint32_t rk = block_rank(foreign, L.w, tot);
			if (i < startB) { if (foreign) L.tmp[m + rk] = (uint16_t)i; else L.ib[i] = L.ia[i]; }
			m += tot;
		}
		uint32_t fb = 0;                                       // misplaced elements seen so far in B
		for (uint32_t base = startB; base < end; base += NT) {
			const uint32_t i = base + tid;
			const bool foreign = i < end && L.db[i] == cA;
			uint32_t tot;
			const uint32_t rk = block_rank(foreign, L.w, tot);
			if (i < end) {
				const uint32_t r = fb + rk;                      // misplaced elements of B before slot i
				if (foreign) { L.ib[L.tmp[r]] = L.ia[i]; L.tmp[m + r] = (uint16_t)i; }
				else L.ib[r < m ? i + 1 : i] = L.ia[i];
			}
			fb += tot;
		}
		__syncthreads();
		for (uint32_t k = tid; k < m; k += NT) L.ib[k == 0 ? startB : (uint32_t)L.tmp[m + k - 1] + 1u] = L.ia[L.tmp[k]];
	} else {
		__syncthreads();
		if (tid == 0) {	// the reference's cycle walk, on (digit, index) pairs
			for (uint32_t c = 0; c < 256; ++c) {
				const uint32_t tl = L.tail[c];
				uint32_t h = L.head[c];
				while (h != tl) {
					uint32_t carry = L.ia[h], d = L.db[h];
					if (d != c) {
						do {
							const uint32_t hh = L.head[d];
							L.head[d] = hh + 1;
							const uint32_t ev = L.ia[hh], dn = L.db[hh];
							L.ib[hh] = (uint16_t)carry;
							carry = ev; d = dn;
						} while (d != c);
					}
					L.ib[h++] = (uint16_t)carry;
				}
				L.head[c] = h;
			}
		}
	}
	__syncthreads();
	for (uint32_t i = beg + tid; i < end; i += NT) L.ia[i] = L.ib[i];
	// children: one bucket per thread
	if (s > 0 && my_cnt > 1) {
		if (my_cnt > 64) { const uint32_t k = atomicAdd(&L.n_rng[nxt], 1u); L.rng[nxt][k] = my_start | (my_start + my_cnt) << 16; L.rsh[nxt][k] = (uint8_t)(s - 8); }
		else { const uint32_t k = atomicAdd(&L.n_small, 1u); L.small[k] = my_start | (my_start + my_cnt) << 16; }
	}
	__syncthreads();
}

// mode 0: fast pass over every read of the size class, sets flag[a] (1 = has equal keys, output not written)
// mode 1: exact pass over the flagged reads
// mode 2: exact pass over every segment (keys known to be full of ties, e.g. chain scores)
template <int CAP>
__global__ __launch_bounds__(NT) void k_sort_block(rh_sort_job jb, uint32_t n_lo, uint32_t n_hi, int mode)
{
	__shared__ sort_lds<CAP> L;
	const uint32_t a = blockIdx.x, tid = threadIdx.x;
	const int exact = mode != 0;
	if (a >= jb.n_seg || (jb.skip && jb.skip[a])) return;
	const uint64_t base = jb.off[a];
	const uint32_t n = jb.cnt ? jb.cnt[a] : (uint32_t)(jb.off[a + 1] - base);
	if (n <= n_lo || n > n_hi) return;
	if (mode == 1 && !jb.need_exact[a]) return;
	const rh_mm128_t *src = jb.src + base;
	rh_mm128_t *dst = jb.dst + base;
	for (uint32_t i = tid; i < n; i += NT) { L.key[i] = src[i].x; L.ia[i] = (uint16_t)i; }
	if (tid == 0) {
		L.n_rng[0] = 0; L.n_rng[1] = 0; L.n_small = 0; L.tie = 0;
		if (n > 64) { L.rng[0][0] = 0u | n << 16; L.rsh[0][0] = 56; L.n_rng[0] = 1; }
		else if (n > 1) { L.small[0] = 0u | n << 16; L.n_small = 1; }
	}
	__syncthreads();
	for (int cur = 0;; cur ^= 1) {
		const uint32_t nr = L.n_rng[cur];
		if (nr == 0) break;
		for (uint32_t ri = 0; ri < nr; ++ri) {
			const uint32_t be = L.rng[cur][ri];
			sort_split_range<CAP>(L, be & 0xFFFFu, be >> 16, (int)L.rsh[cur][ri], cur ^ 1, exact);
		}
		__syncthreads();
		if (tid == 0) L.n_rng[cur] = 0;
		__syncthreads();
	}
	// Ranges of <= 64 records get klib's insertion sort, i.e. any STABLE sort: one wavefront per range computes each
	// record's rank (smaller keys + equal keys that come earlier) with broadcast LDS reads and scatters in one step.
	const uint32_t ns = L.n_small;
	for (uint32_t q = wave_id(); q < ns; q += NT / 64) {
		const uint32_t b = L.small[q] & 0xFFFFu, m = (L.small[q] >> 16) - b, l = lane_id();
		const uint16_t idx = l < m ? L.ia[b + l] : (uint16_t)0;
		const uint64_t k = L.key[idx];
		uint32_t rank = 0;
		for (uint32_t j = 0; j < m; ++j) {
			const uint64_t kj = L.key[L.ia[b + j]];
			rank += (kj < k || (kj == k && j < l)) ? 1u : 0u;
		}
		if (l < m) L.ib[b + rank] = idx;
	}
	__syncthreads();
	for (uint32_t q = wave_id(); q < ns; q += NT / 64) {
		const uint32_t b = L.small[q] & 0xFFFFu, m = (L.small[q] >> 16) - b, l = lane_id();
		if (l < m) L.ia[b + l] = L.ib[b + l];
	}
	__syncthreads();
	if (!exact) {
		uint32_t tie = 0;
		for (uint32_t i = tid + 1; i < n; i += NT) if (L.key[L.ia[i]] == L.key[L.ia[i - 1]]) tie = 1;
		if (tie) L.tie = 1;
		__syncthreads();
		if (tid == 0) jb.need_exact[a] = (uint8_t)L.tie;
		if (L.tie) return;
	}
	for (uint32_t i = tid; i < n; i += NT) dst[i] = src[L.ia[i]];
}

// reads too large for LDS: copy, then the serial in-place emulation (one read per lane)
__global__ void k_sort_big(rh_sort_job jb, uint32_t n_lo)
{
	const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= jb.n_seg || (jb.skip && jb.skip[a])) return;
	const uint64_t base = jb.off[a];
	const uint32_t n = jb.cnt ? jb.cnt[a] : (uint32_t)(jb.off[a + 1] - base);
	if (n <= n_lo) return;
	for (uint32_t i = 0; i < n; ++i) jb.dst[base + i] = jb.src[base + i];
	rh_radix_sort_128x(jb.dst + base, n, (uint32_t*)(jb.scratch + base * jb.scratch_stride + jb.scratch_skip * (jb.off[a + 1] - base)));
}

#ifndef RH_SORT_CAP1
#define RH_SORT_CAP1 4096     // ~72 KB of LDS: two workgroups per CU
#endif
#ifndef RH_SORT_CAP2
#define RH_SORT_CAP2 8192     // ~141 KB of LDS: one workgroup per CU (reads that carry many chained anchors)
#endif

#ifndef RH_SORT_CAP0
#define RH_SORT_CAP0 512      // ~12 KB of LDS: candidate / chain-key sorts and short anchor lists
#endif

template <int CAP>
static void launch_class(hipStream_t s, const rh_sort_job &jb, bool all_exact, uint32_t lo, uint32_t hi)
{
	if (all_exact) RH_LAUNCH(k_sort_block<CAP>, jb.n_seg, NT, 0, s, jb, lo, hi, 2);
	else {
		RH_LAUNCH(k_sort_block<CAP>, jb.n_seg, NT, 0, s, jb, lo, hi, 0);
		RH_LAUNCH(k_sort_block<CAP>, jb.n_seg, NT, 0, s, jb, lo, hi, 1);
	}
}

void rhk_sort_job(hipStream_t s, const rh_sort_job &jb, bool all_exact, uint32_t min_n)
{
	if (!jb.n_seg) return;
	launch_class<RH_SORT_CAP0>(s, jb, all_exact, min_n, (uint32_t)RH_SORT_CAP0);
	launch_class<RH_SORT_CAP1>(s, jb, all_exact, (uint32_t)RH_SORT_CAP0, (uint32_t)RH_SORT_CAP1);
	launch_class<RH_SORT_CAP2>(s, jb, all_exact, (uint32_t)RH_SORT_CAP1, (uint32_t)RH_SORT_CAP2);
	RH_LAUNCH(k_sort_big, (jb.n_seg + 63) / 64, 64, 0, s, jb, (uint32_t)RH_SORT_CAP2);
}

// anchor sort of a chunk round: unsorted expand output -> reference order
void rhk_sort(hipStream_t s, const rh_dev_round &r)
{
	rh_sort_job jb = { r.n_act, r.skip, r.a_off, nullptr, r.raw, r.anc, r.need_exact, r.ws, RH_WS_PER_ANCHOR, 0 };
	rhk_sort_job(s, jb, false, 0u);
}
