// Segments beyond the LDS classes of rh_sort.hip (tens of thousands to millions of records: large indexes put that many
// anchors into one chunk): the reference's radix_sort_128x (ksort.h:101-151) level by level with MANY workgroups per
// segment, still producing its exact - unstable - permutation.
//
// One level of rs_sort on a range = an in-place American-flag pass on one byte of the key.  What it does to a range
// follows from two facts ("Exact unstable sort"):
//   * Split the range into the buckets' regions.  A record sitting in its own region is "in place", every other position
//     is a "hole".  The cycle walk only ever reads the holes of a region in position order: it is a token that moves
//     between regions, each visit consuming the region's next hole and continuing with the bucket of the record found
//     there.  So the walk needs nothing but one byte per hole (the digit of its record), region by region, and it is
//     the only serial part: one lane walks the byte streams (staged through LDS windows) and notes, for the record of
//     every hole, which hole of its own region it lands in (`dest`).
//   * Where the j-th arrival of a region ends up, and what happens to the in-place records, is closed form: while an
//     earlier bucket is being filled ("phase 1", the first J arrivals) an arrival is written at the region's head and
//     pushes the run of in-place records before the next hole one slot to the right; once the walk works on the region
//     itself ("phase 2") arrivals drop straight into the holes and nothing shifts.
// Everything except the token walk is data parallel over fixed tiles of a range: OR / AND of the keys (highest differing
// byte: levels on which all keys agree are identities), digit histogram, hole compaction (digits + positions in
// position order), final placement.  Ranges of one level are independent; a range's buckets become the next level's
// ranges (still too large), segments for the LDS block sorter (<= its largest class; they finish there, insertion
// sorts of <= 64 records included), or are final.  Records move once per level between two scratch copies (the
// job's own source array and `alt`); finished buckets go straight to the destination array.
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdlib>
#include <atomic>
#include "rh_kernels.h"
#include "rh_devutil.h"

#ifndef BS_TILE_IT
#define BS_TILE_IT 8                      // records per thread of a tile
#endif
#define BS_TILE (NT * BS_TILE_IT)
#define BS_CW_WORDS ((BS_TILE_IT < 2 ? 2 : BS_TILE_IT) * (NT / 64))   // bs_classify's per-wavefront counts (at least two rows: the regions of a tile's two ends)
#ifndef BS_WIN_BYTES
#define BS_WIN_BYTES 4096                 // LDS of the token walk's stream windows (>= 512: two entries for each of 256 regions)
#endif
// the block-parallel walk (K7' below)
#ifndef PW_MIN_HOLES
#define PW_MIN_HOLES 4096                 // ranges with fewer holes stay with the serial walkers
#endif
#ifndef PW_MIN_C0
#define PW_MIN_C0 64                      // ... and so do ranges whose first region starts fewer cycles than this
#endif
#ifndef PW_WIN_CAP
#define PW_WIN_CAP 12288                  // holes of one work item of k_bs_pw_walk (their digits are staged in LDS; 16-bit window addresses)
#endif
#ifndef PW_BLK_HOLES
#define PW_BLK_HOLES (PW_WIN_CAP / 3)     // holes a block of cycles should pop at most (expected: cycles per round x average cycle length)
#endif
#define PW_MIN_CPR 4                      // fewest cycles followed at a time
#ifndef PW_MAX_CYCLE
#define PW_MAX_CYCLE 8                    // ranges whose cycles are longer than this on average stay with the serial walkers (measured at human scale: evenly filled regions - the exact re-sorts of reads with equal anchor keys, 24 targets or 256 values of a position byte - gain nothing: a block of their cycles is thousands of holes for one lane)
#endif
#ifdef RH_DEV
#define PW_STAT(i, n) atomicAdd(&C.hdr[20 + (i)], (uint32_t)(n))   // development builds: [20] blocks walked by one lane, [21] ranges out of slots, [22] window misses, [23] cycles walked by one lane, [24] window reloads
#else
#define PW_STAT(i, n) ((void)0)
#endif
#ifndef PW_RING_BYTES
#define PW_RING_BYTES 4096                // LDS of k_bs_pw_count's digit windows, shared by the regions in proportion to their holes
#endif

struct bs_range {
	uint64_t beg;                         // absolute record offset of the range in the job's arrays
	uint32_t n;
	uint32_t tile0;                       // first tile of the range in this level's tile numbering
	uint8_t buf;                          // which copy holds it: 0 = job source, 1 = alt
	uint8_t shift;                        // highest byte shift this level may split on
	uint8_t has_dg;                       // its digits were written by the placement of the level above (at the byte predicted from dmask)
	uint8_t exact;                        // this level's passes on the range reproduce the reference's permutation (hole lists + walk); 0: placed in any order (no equal keys in it)
	uint32_t dead;                        // the range starts with this many records that nobody reads once sorted (rh_sort_job::dead_cnt): all one key, the lowest
	uint64_t dmask;                       // has_dg: bits on which the keys of the PARENT range differ below the parent's byte - a superset of this range's
};

// per-range tables of a level
struct bs_meta {
	uint64_t k_or, k_and;
	uint32_t cnt[256];                    // digit histogram
	uint32_t start[257];                  // exclusive scan
	uint32_t inpl[256];                   // records already in their region
	uint32_t hst[257];                    // holes before each region (= index of the region's first hole)
	uint32_t J[256];                      // arrivals of a region before the walk turns to it
	uint8_t fate[256];                    // 0 empty, 1 final, 2 block sorter, 3 next level
	uint8_t act[256], dmap[256];          // regions with holes, renumbered 0 .. nh-1 in digit order: dense -> digit, digit -> dense
	int32_t s;                            // byte shift of this level (-1: all keys equal)
	uint32_t nh;
	uint32_t pw, pw_ncpr, pw_off, pw_slots;   // block-parallel walk (k_bs_pw_count / k_bs_pw_walk): 0 or cycles per block; cycles followed at a time; the range's snapshot slots in C.pw_snap (word offset, number)
};
enum { BS_EMPTY = 0, BS_FINAL = 1, BS_SMALL = 2, BS_BIG = 3, BS_DROP = 4 };   // DROP: a bucket of an exact re-run (tie_path) that holds no equal keys - the any-order job before it left that stretch of the destination sorted: not placed, not sorted again

struct bs_ctx {
	void *buf[2];                         // 0 = job source (overwritten), 1 = alt; records of the job's type (rh_mm128_t, or uint64_t: rf.rec8)
	void *dst;
	rh_rec_fmt rf;
	bs_range *rng[2];                     // this level's ranges / next level's
	bs_meta *meta;
	uint32_t *tile_h;                     // per tile: holes -> (after the scan) holes of the range before the tile
	uint32_t *tile_rng;                   // per tile: its range
	uint8_t *dg, *dg_next, *hd;           // per record: digit (this level's / written for the next level's ranges); per hole: digit of its record
	uint32_t *hp, *dest;                  // per hole: position in the range; hole (of the record's own region) it moves to
	uint64_t *small_off[4]; uint32_t *small_cnt[4];   // segments for the block sorter, by (copy that holds them) * 2 + (keys differ below bit 32 only)
	uint32_t *hdr;                        // [0] ranges of this level [1] tiles [2..5] block-sorter segments per list [6] next level's ranges [7] error [8..11] largest segment per list
	uint32_t small_cap, rng_cap, n_lo;
	// any-order jobs (rh_sort_job::redo_skip): equal keys end up in one bucket whatever the order, so whoever finishes a bucket knows - a final
	// bucket of more than one record (k_bs_plan), or the block sorter's tie flag of a bucket (small_tie, read by k_bs_tie_map) - and clears
	// redo_skip of the segment the bucket lies in (found by position in seg_off): no pass over the sorted records to look for equal neighbours
	const uint64_t *seg_off; uint32_t seg_n; uint8_t *redo_skip; uint8_t *small_tie[4];
	// ... and sets a bit for every 64 positions of the sorted order the bucket covers (tie_bits; tie_path: the exact re-run of those segments, which
	// reads them - a child range is exact only if its interval holds a marked position)
	unsigned long long *tie_bits; uint8_t tie_path, any_order;
	// block-parallel walk: pointer snapshots (pw_words 32-bit words, handed out per range by k_bs_scan through hdr[18]) and the blocks grouped
	// into work items for k_bs_pw_walk, two lists (ranges with up to 64 / up to 256 regions that have holes; counted in hdr[16], hdr[17])
	uint32_t *pw_snap; uint32_t pw_words; struct bs_pw_item *pw_items[2]; uint32_t pw_item_cap;
};
struct bs_pw_item { uint32_t r, slot0, nb, pad; };
RH_DEV void bs_mark_tie(const bs_ctx &C, uint64_t pos, uint32_t len)
{
	uint32_t lo = 0, hi = C.seg_n;                                  // the last segment that starts at or before pos
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (C.seg_off[mid] <= pos) lo = mid; else hi = mid; }
	C.redo_skip[lo] = 0;
	if (C.tie_bits) for (uint64_t g = pos >> 6; g <= (pos + (len ? len - 1u : 0u)) >> 6; ++g) atomicOr(&C.tie_bits[g >> 6], 1ull << (g & 63u));
}
RH_DEV bool bs_has_tie(const bs_ctx &C, uint64_t pos, uint32_t len)   // a marked position in [pos, pos + len)
{
	const uint64_t g0 = pos >> 6, g1 = (pos + (len ? len - 1u : 0u)) >> 6;
	for (uint64_t w = g0 >> 6; w <= g1 >> 6; ++w) {
		unsigned long long m = ~0ull;
		if (w == g0 >> 6) m &= ~0ull << (g0 & 63u);
		if (w == g1 >> 6) m &= ~0ull >> (63u - (g1 & 63u));
		if (C.tie_bits[w] & m) return true;
	}
	return false;
}

// the range a tile belongs to: looked up once per level (k_bs_tile_map), not by every kernel of the level (a binary search over
// the ranges is ~14 dependent loads by one lane while its workgroup waits)
RH_DEV uint32_t bs_find_range(const bs_ctx &C, uint32_t tile, uint32_t, uint32_t *) { return C.tile_rng[tile]; }

__global__ __launch_bounds__(NT) void k_bs_tile_map(bs_ctx C)
{
	const uint32_t tile = blockIdx.x * NT + threadIdx.x, n_rng = C.hdr[0];
	if (tile >= C.hdr[1]) return;
	uint32_t lo = 0, hi = n_rng;          // largest r with tile0[r] <= tile
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (C.rng[0][mid].tile0 <= tile) lo = mid; else hi = mid; }
	C.tile_rng[tile] = lo;
}

// ------------------------------------------------------------------------------------------------ level set-up
// level 0: one range per segment longer than n_lo
__global__ __launch_bounds__(NT) void k_bs_init(rh_sort_job jb, bs_ctx C)
{
	__shared__ uint32_t s_w[NT / 64];
	__shared__ uint32_t s_run[2];
	const uint32_t tid = threadIdx.x;
	if (tid == 0) { s_run[0] = 0; s_run[1] = 0; }
	__syncthreads();
	for (uint32_t a0 = 0; a0 < jb.n_seg; a0 += NT) {
		const uint32_t a = a0 + tid;
		uint32_t n = 0;
		if (a < jb.n_seg && !(jb.skip && jb.skip[a])) n = jb.cnt ? jb.cnt[a] : (uint32_t)(jb.off[a + 1] - jb.off[a]);
		const bool big = n > C.n_lo;
		uint32_t tot_r, tot_t;
		const uint32_t rk = block_rank(big, s_w, tot_r);
		const uint32_t tk = block_excl_scan(big ? (n + BS_TILE - 1) / BS_TILE : 0u, s_w, tot_t);
		if (big) {
			bs_range q;
			q.beg = jb.off[a]; q.n = n; q.tile0 = s_run[1] + tk; q.buf = 0; q.shift = 56; q.has_dg = 0; q.dmask = 0; q.exact = C.any_order ? 0 : 1; q.dead = jb.dead_cnt && !C.any_order ? jb.dead_cnt[a] : 0u;
			if (s_run[0] + rk < C.rng_cap) C.rng[0][s_run[0] + rk] = q;
		}
		__syncthreads();
		if (tid == 0) { s_run[0] += tot_r; s_run[1] += tot_t; }
		__syncthreads();
	}
	if (tid == 0) { C.hdr[0] = s_run[0]; C.hdr[1] = s_run[1]; C.hdr[2] = 0; C.hdr[3] = 0; C.hdr[4] = 0; C.hdr[5] = 0; C.hdr[6] = 0; C.hdr[7] = s_run[0] > C.rng_cap ? 1u : 0u; C.hdr[8] = 0; C.hdr[9] = 0; C.hdr[10] = 0; C.hdr[11] = 0; C.hdr[12] = 0; C.hdr[13] = 0; C.hdr[14] = 0; C.hdr[15] = 0; C.hdr[16] = 0; C.hdr[17] = 0; C.hdr[18] = 0; C.hdr[19] = 0; }
}

__global__ __launch_bounds__(NT) void k_bs_clear(bs_ctx C)
{
	const uint32_t r = blockIdx.x, tid = threadIdx.x;
	if (r >= C.hdr[0]) return;
	bs_meta &M = C.meta[r];
	M.cnt[tid] = 0; M.inpl[tid] = 0;
	if (r == 0 && tid < 12) C.hdr[16 + tid] = 0;                      // (the level's work items and snapshot cursor of the block-parallel walk)
	// a range whose digits came with it: its keys are not read again; the parent's differing bits stand in for its own (their
	// highest byte is the byte the digits were taken from; k_bs_fix looks at the keys if the range turns out to agree on it)
	if (tid == 0) { const bs_range R = C.rng[0][r]; if (R.has_dg) { M.k_or = R.dmask; M.k_and = 0; } else { M.k_or = 0; M.k_and = ~0ull; } }
}

// ------------------------------------------------------------------------------------------------ K1: OR / AND of the keys
template <class REC>
__global__ __launch_bounds__(NT) void k_bs_diff(bs_ctx C)
{
	__shared__ uint32_t s_r;
	__shared__ uint64_t s_red[2 * (NT / 64)];
	const uint32_t n_rng = C.hdr[0], tid = threadIdx.x;
	if (blockIdx.x >= C.hdr[1]) return;
	const uint32_t r = bs_find_range(C, blockIdx.x, n_rng, &s_r);
	const bs_range R = C.rng[0][r];
	if (R.has_dg) return;
	const uint32_t t0 = (blockIdx.x - R.tile0) * BS_TILE;
	const REC *src = reinterpret_cast<const REC*>(C.buf[R.buf]) + R.beg;
	uint64_t vo = 0, va = ~0ull;
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		if (p < R.n) { const uint64_t k = rh_rec_ops<REC>::key(src[p], C.rf); vo |= k; va &= k; }
	}
	for (int d = 32; d > 0; d >>= 1) { vo |= __shfl_xor(vo, d); va &= __shfl_xor(va, d); }
	if (lane_id() == 0) { s_red[2 * wave_id()] = vo; s_red[2 * wave_id() + 1] = va; }
	__syncthreads();
	if (tid == 0) {
		for (uint32_t q = 1; q < NT / 64; ++q) { vo |= s_red[2 * q]; va &= s_red[2 * q + 1]; }
		atomicOr((unsigned long long*)&C.meta[r].k_or, (unsigned long long)vo);
		atomicAnd((unsigned long long*)&C.meta[r].k_and, (unsigned long long)va);
	}
}

RH_DEV int bs_level_shift(const bs_meta &M, uint32_t shift_max)
{
	const uint64_t diff = M.k_or & ~M.k_and;
	if (diff == 0) return -1;
	int s = (63 - __clzll(diff)) & ~7;
	if (s > (int)shift_max) s = (int)shift_max;                   // (cannot happen: the bytes above agree; kept as the reference's bound)
	return s;
}

// ------------------------------------------------------------------------------------------------ K2: digits + histogram
template <class REC>
__global__ __launch_bounds__(NT) void k_bs_hist(bs_ctx C)
{
	__shared__ uint32_t s_r;
	__shared__ uint32_t s_cnt[256];
	const uint32_t n_rng = C.hdr[0], tid = threadIdx.x;
	if (blockIdx.x >= C.hdr[1]) return;
	const uint32_t r = bs_find_range(C, blockIdx.x, n_rng, &s_r);
	const bs_range R = C.rng[0][r];
	const int s = bs_level_shift(C.meta[r], R.shift);
	const uint32_t t0 = (blockIdx.x - R.tile0) * BS_TILE;
	const REC *src = reinterpret_cast<const REC*>(C.buf[R.buf]) + R.beg;
	uint8_t *dg = C.dg + R.beg;
	s_cnt[tid] = 0;
	__syncthreads();
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		if (p < R.n) {
			uint32_t d;
			if (R.has_dg) d = dg[p];
			else { d = s < 0 ? 0u : (uint32_t)(rh_rec_ops<REC>::key(src[p], C.rf) >> s) & 255u; dg[p] = (uint8_t)d; }
			atomicAdd(&s_cnt[d], 1u);
		}
	}
	__syncthreads();
	if (s_cnt[tid]) atomicAdd(&C.meta[r].cnt[tid], s_cnt[tid]);
}

// K1+K2 of level 0 in one read of the keys: the byte a job's first level splits on hardly ever changes between calls (the strand bit
// of anchor keys, the second byte of chain scores), so the histogram and the digits are taken at the byte the previous call of this
// kind of job found (`gs`) while the OR / AND are gathered; k_bs_fix0 redoes the ranges whose keys differ on another byte.
template <class REC>
__global__ __launch_bounds__(NT) void k_bs_hist0(bs_ctx C, int gs)
{
	__shared__ uint32_t s_r;
	__shared__ uint32_t s_cnt[256];
	__shared__ uint64_t s_red[2 * (NT / 64)];
	const uint32_t n_rng = C.hdr[0], tid = threadIdx.x;
	if (blockIdx.x >= C.hdr[1]) return;
	const uint32_t r = bs_find_range(C, blockIdx.x, n_rng, &s_r);
	const bs_range R = C.rng[0][r];
	const uint32_t t0 = (blockIdx.x - R.tile0) * BS_TILE;
	const REC *src = reinterpret_cast<const REC*>(C.buf[R.buf]) + R.beg;
	uint8_t *dg = C.dg + R.beg;
	s_cnt[tid] = 0;
	__syncthreads();
	uint64_t vo = 0, va = ~0ull;
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		if (p < R.n) {
			const uint64_t k = rh_rec_ops<REC>::key(src[p], C.rf);
			vo |= k; va &= k;
			const uint32_t d = (uint32_t)(k >> gs) & 255u;
			dg[p] = (uint8_t)d;
			atomicAdd(&s_cnt[d], 1u);
		}
	}
	for (int d = 32; d > 0; d >>= 1) { vo |= __shfl_xor(vo, d); va &= __shfl_xor(va, d); }
	if (lane_id() == 0) { s_red[2 * wave_id()] = vo; s_red[2 * wave_id() + 1] = va; }
	__syncthreads();
	if (s_cnt[tid]) atomicAdd(&C.meta[r].cnt[tid], s_cnt[tid]);
	if (tid == 0) {
		for (uint32_t q = 1; q < NT / 64; ++q) { vo |= s_red[2 * q]; va &= s_red[2 * q + 1]; }
		atomicOr((unsigned long long*)&C.meta[r].k_or, (unsigned long long)vo);
		atomicAnd((unsigned long long*)&C.meta[r].k_and, (unsigned long long)va);
	}
}

// level 0, one workgroup per range: hdr[14] = 1 + the byte shift some range of the level really splits on (what the host remembers
// for the next job of this kind), and - after k_bs_hist0 - histogram and digits of a range that splits on another byte than `gs`
// (hdr[13] counts them) redone from its keys
template <class REC>
__global__ __launch_bounds__(NT) void k_bs_fix0(bs_ctx C, int gs)
{
	__shared__ uint32_t s_cnt[256];
	const uint32_t r = blockIdx.x, tid = threadIdx.x;
	if (r >= C.hdr[0]) return;
	const bs_range R = C.rng[0][r];
	bs_meta &M = C.meta[r];
	const int s = bs_level_shift(M, R.shift);
	if (tid == 0 && s >= 0 && (s != gs || r == 0)) C.hdr[14] = (uint32_t)s + 1u;
	if (gs < 0 || s == gs) return;
	if (tid == 0) atomicAdd(&C.hdr[13], 1u);
	const REC *src = reinterpret_cast<const REC*>(C.buf[R.buf]) + R.beg;
	uint8_t *dg = C.dg + R.beg;
	s_cnt[tid] = 0;
	__syncthreads();
	for (uint32_t p = tid; p < R.n; p += NT) {
		const uint32_t d = s < 0 ? 0u : (uint32_t)(rh_rec_ops<REC>::key(src[p], C.rf) >> s) & 255u;
		dg[p] = (uint8_t)d;
		atomicAdd(&s_cnt[d], 1u);
	}
	__syncthreads();
	M.cnt[tid] = s_cnt[tid];
}

// K2b: a range that came with its digits (has_dg) and turns out to agree on the byte they were taken from - its histogram has one
// bucket - is the one case where the parent's differing bits said too much: its own keys are read after all (one workgroup per
// range; rare - a target whose hits fall into one 16 Mbp stretch, a score byte with one value), OR / AND, histogram and digits redone
template <class REC>
__global__ __launch_bounds__(NT) void k_bs_fix(bs_ctx C)
{
	__shared__ uint64_t s_red[2 * (NT / 64)];
	__shared__ uint32_t s_w[NT / 64];
	const uint32_t r = blockIdx.x, tid = threadIdx.x;
	if (r >= C.hdr[0]) return;
	const bs_range R = C.rng[0][r];
	if (!R.has_dg) return;
	bs_meta &M = C.meta[r];
	uint32_t nb;
	(void)block_rank(M.cnt[tid] != 0, s_w, nb);
	if (nb != 1 || R.n < 2) return;
	const REC *src = reinterpret_cast<const REC*>(C.buf[R.buf]) + R.beg;
	uint64_t vo = 0, va = ~0ull;
	for (uint32_t p = tid; p < R.n; p += NT) { const uint64_t k = rh_rec_ops<REC>::key(src[p], C.rf); vo |= k; va &= k; }
	for (int d = 32; d > 0; d >>= 1) { vo |= __shfl_xor(vo, d); va &= __shfl_xor(va, d); }
	if (lane_id() == 0) { s_red[2 * wave_id()] = vo; s_red[2 * wave_id() + 1] = va; }
	__syncthreads();
	for (uint32_t q = 0; q < NT / 64; ++q) { vo |= s_red[2 * q]; va &= s_red[2 * q + 1]; }
	const uint64_t diff = vo & ~va;
	int s = -1;
	if (diff) { s = (63 - __clzll(diff)) & ~7; if (s > (int)R.shift) s = (int)R.shift; }
	M.cnt[tid] = 0;
	__syncthreads();
	if (tid == 0) { M.k_or = vo; M.k_and = va; }
	uint8_t *dg = C.dg + R.beg;
	for (uint32_t p = tid; p < R.n; p += NT) {
		const uint32_t d = s < 0 ? 0u : (uint32_t)(rh_rec_ops<REC>::key(src[p], C.rf) >> s) & 255u;
		dg[p] = (uint8_t)d;
		atomicAdd(&M.cnt[d], 1u);
	}
}

// ------------------------------------------------------------------------------------------------ K3: regions, fates, children
__global__ __launch_bounds__(NT) void k_bs_plan(bs_ctx C)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint32_t r = blockIdx.x, tid = threadIdx.x;
	if (r >= C.hdr[0]) return;
	const bs_range R = C.rng[0][r];
	bs_meta &M = C.meta[r];
	const int s = bs_level_shift(M, R.shift);
	const uint32_t c = M.cnt[tid];
	uint32_t total;
	const uint32_t st = block_excl_scan(c, s_w, total);
	M.start[tid] = st;
	if (tid == 0) { M.start[256] = total; M.s = s; }
	// bits on which the range's keys differ below this level's byte: what its buckets can still differ on (none: every bucket is
	// final; else their highest byte is where the placement takes the next level's digits from)
	const uint64_t low = s > 0 ? (M.k_or & ~M.k_and) & ((1ull << s) - 1ull) : 0ull;
	uint8_t fate = BS_EMPTY;
	if (c) {
		if (s <= 0 || c == 1 || low == 0) fate = BS_FINAL;        // all keys equal, last byte done, nothing below differs, or a single record
		else if (c <= C.n_lo) fate = BS_SMALL;
		else fate = BS_BIG;
	}
	if (C.redo_skip && fate == BS_FINAL && c > 1) bs_mark_tie(C, R.beg + st, c);   // (any order: a final bucket of several records = equal keys)
	// Round 6.  The exact re-run of the segments an any-order job found equal keys in: a bucket is an interval of the sorted order, and one without equal keys has ONE sorted
	// order - the one that job has already written to the destination.  Only the buckets on the way to the equal keys go on (the walk of THIS level still needed all of the
	// range's records: who arrives where depends on every one of them); the others are dropped here - until round 6 they were placed and sorted a second time.
	if (C.tie_path && R.exact && fate != BS_EMPTY && !bs_has_tie(C, R.beg + st, c)) fate = BS_DROP;
	const uint8_t alt = R.buf ^ 1;
	// Buckets for the block sorter go to one of two lists of the copy that holds them: 32-bit LDS keys when the bucket's keys
	// agree on every bit from bit 32 up and that class takes a bucket of this size, 64-bit keys otherwise.  One atomic per
	// wavefront and list (ranked by ballot): a level has up to a million such buckets.
	// (keys ordered with their upper fields moved up, rh_rec_fmt::up: the original key's bits from 32 up are this key's bits from `up` up)
	const bool small = fate == BS_SMALL, narrow = small && s <= (C.rf.up ? (int)C.rf.up : 32) && c <= (uint32_t)RH_SORT32_CAP3;
#pragma unroll
	for (int w = 0; w < 2; ++w) {
		const bool mine = small && (narrow == (w == 1));
		const uint64_t m = __ballot(mine);
		if (m == 0) continue;
		const uint32_t li = (uint32_t)alt * 2u + (uint32_t)w, leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
		uint32_t base = 0;
		uint32_t cmax = mine ? c : 0u;                               // the list's largest bucket: the host launches no block-sorter class above it
		for (int d = 32; d > 0; d >>= 1) { const uint32_t t = __shfl_xor(cmax, d); cmax = t > cmax ? t : cmax; }
		if (lane_id() == leader) { base = atomicAdd(&C.hdr[2 + li], (uint32_t)__popcll(m)); atomicMax(&C.hdr[8 + li], cmax); }
		base = __shfl(base, (int)leader);
		if (mine) {
			const uint32_t k = base + lanes_below(m);
			if (k < C.small_cap) { C.small_off[li][k] = R.beg + st; C.small_cnt[li][k] = c; }
			else C.hdr[7] = 1;
		}
	}
	if (fate == BS_BIG) {
		const uint32_t k = atomicAdd(&C.hdr[6], 1u);
		if (k < C.rng_cap) {
			bs_range q;
			q.beg = R.beg + st; q.n = c; q.tile0 = 0; q.buf = alt; q.shift = (uint8_t)(s - 8); q.has_dg = 1; q.dmask = low;
			q.exact = R.exact && (!C.tie_path || bs_has_tie(C, R.beg + st, c)) ? 1 : 0;   // (the way to the equal keys only)
			q.dead = st == 0 ? R.dead : 0u;                             // (the lowest keys are in the first bucket that is not empty)
			C.rng[1][k] = q;
		} else C.hdr[7] = 1;
	}
	M.fate[tid] = fate;
}

// ------------------------------------------------------------------------------------------------ tiles: classification
// region of position p: largest b with start[b] <= p (empty buckets share their start with the next one)
RH_DEV uint32_t bs_region(const uint32_t *start, uint32_t p, uint32_t lo = 0, uint32_t hi = 256)   // the region of position p, known to be in [lo, hi)
{
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (start[mid] <= p) lo = mid; else hi = mid; }
	return lo;
}

// For the thread's BS_TILE_IT records of the tile (record `it` at position t0 + it * NT + tid): digit, region, and the
// number of holes of the tile before it (position order).  s_cw: BS_CW_WORDS words.  Returns the tile's holes.
struct bs_cls { uint32_t d[BS_TILE_IT], b[BS_TILE_IT], hb[BS_TILE_IT]; };
RH_DEV uint32_t bs_classify(const uint8_t *dg, const uint32_t *s_start, uint32_t t0, uint32_t n, uint32_t *s_cw, bs_cls &q)
{
	const uint32_t tid = threadIdx.x, w = wave_id();
	uint64_t bal[BS_TILE_IT];
	// a tile lies in one region or a few: the regions of its ends (the same LDS words for every lane) bracket every record's
	const uint32_t p_last = t0 + BS_TILE - 1u < n ? t0 + BS_TILE - 1u : n - 1u;
	// (region of p = number of buckets whose start is <= p, minus one: the starts do not decrease - one ballot per wavefront for each end
	// instead of two binary searches of eight dependent LDS reads by every thread)
	{
		const uint32_t st = s_start[tid];
		const uint64_t m_lo = __ballot(st <= t0), m_hi = __ballot(st <= p_last);
		if (lane_id() == 0) { s_cw[w] = (uint32_t)__popcll(m_lo); s_cw[NT / 64 + w] = (uint32_t)__popcll(m_hi); }
	}
	__syncthreads();
	uint32_t b_lo = 0, b_hi = 0;
	for (uint32_t ww = 0; ww < NT / 64; ++ww) { b_lo += s_cw[ww]; b_hi += s_cw[NT / 64 + ww]; }
	b_lo -= 1u; b_hi -= 1u;
	__syncthreads();
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		bool hole = false;
		q.d[it] = 0; q.b[it] = 0;
		if (p < n) { q.d[it] = dg[p]; q.b[it] = b_lo == b_hi ? b_lo : bs_region(s_start, p, b_lo, b_hi + 1u); hole = q.d[it] != q.b[it]; }
		bal[it] = __ballot(hole);
		if (lane_id() == 0) s_cw[it * (NT / 64) + w] = (uint32_t)__popcll(bal[it]);
	}
	__syncthreads();
	uint32_t run = 0, total = 0;
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		for (uint32_t ww = 0; ww < NT / 64; ++ww) {
			const uint32_t c = s_cw[it * (NT / 64) + ww];
			if (ww == w) q.hb[it] = total + lanes_below(bal[it]);
			total += c;
		}
	}
	(void)run;
	__syncthreads();
	return total;
}

// K4: holes per tile, in-place records per bucket
__global__ __launch_bounds__(NT) void k_bs_count(bs_ctx C)
{
	__shared__ uint32_t s_r;
	__shared__ uint32_t s_start[257], s_inpl[256], s_cw[BS_CW_WORDS];
	const uint32_t n_rng = C.hdr[0], tid = threadIdx.x;
	if (blockIdx.x >= C.hdr[1]) return;
	const uint32_t r = bs_find_range(C, blockIdx.x, n_rng, &s_r);
	const bs_range R = C.rng[0][r];
	if (!R.exact) return;
	const bs_meta &M = C.meta[r];
	s_start[tid] = M.start[tid]; s_inpl[tid] = 0;
	if (tid == 0) s_start[256] = M.start[256];
	__syncthreads();
	const uint32_t t0 = (blockIdx.x - R.tile0) * BS_TILE;
	bs_cls q;
	const uint32_t holes = bs_classify(C.dg + R.beg, s_start, t0, R.n, s_cw, q);
	// (a wavefront's 64 consecutive positions lie in one region or two: one LDS atomic per region and wavefront, not one per
	// in-place record on the same word)
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		const bool inp = p < R.n && q.d[it] == q.b[it];
		uint64_t m = __ballot(inp);
		while (m) {
			const uint32_t bl = (uint32_t)__shfl((int)q.b[it], __ffsll((unsigned long long)m) - 1);
			const uint64_t same = __ballot(inp && q.b[it] == bl);
			if (lane_id() == 0) atomicAdd(&s_inpl[bl], (uint32_t)__popcll(same));
			m &= ~same;
		}
	}
	__syncthreads();
	if (s_inpl[tid]) atomicAdd(&C.meta[r].inpl[tid], s_inpl[tid]);
	if (tid == 0) C.tile_h[blockIdx.x] = holes;
}

// K5: per range - holes before each tile, holes before each region
__global__ __launch_bounds__(NT) void k_bs_scan(bs_ctx C)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint32_t r = blockIdx.x, tid = threadIdx.x;
	if (r >= C.hdr[0]) return;
	const bs_range R = C.rng[0][r];
	bs_meta &M = C.meta[r];
	if (!R.exact) { if (tid == 0) { M.nh = 0; M.pw = 0; M.hst[256] = 0; } return; }   // (placed in any order: nothing to walk)
	const uint32_t nt = (R.n + BS_TILE - 1) / BS_TILE;
	uint32_t run = 0;
	for (uint32_t i0 = 0; i0 < nt; i0 += NT) {
		const uint32_t i = i0 + tid;
		const uint32_t v = i < nt ? C.tile_h[R.tile0 + i] : 0u;
		uint32_t tot;
		const uint32_t ex = block_excl_scan(v, s_w, tot);
		if (i < nt) C.tile_h[R.tile0 + i] = run + ex;
		run += tot;
	}
	uint32_t total, nh;
	const uint32_t m = M.cnt[tid] - M.inpl[tid];
	const uint32_t hs = block_excl_scan(m, s_w, total);
	M.hst[tid] = hs;
	const uint32_t q = block_rank(m != 0, s_w, nh);
	M.dmap[tid] = (uint8_t)q;
	if (m != 0) M.act[q] = (uint8_t)tid;
	M.J[tid] = 0;
	// Block-parallel walk (k_bs_pw_count / k_bs_pw_walk below) for the ranges whose first region starts a sizeable share of the cycles - the
	// candidates of the backtrack: the chains of one anchor all have the lowest score.  The range gets its snapshot slots here.
	__shared__ uint32_t s_c0;
	if (m != 0 && q == 0) s_c0 = m;
	__syncthreads();
	if (tid == 0) {
		M.hst[256] = total; M.nh = nh;
		uint32_t pw = 0, ncpr = 0, off = 0, slots = 0;
		if (C.pw_snap && nh >= 3 && nh <= 256 && total >= (uint32_t)PW_MIN_HOLES && s_c0 >= (uint32_t)PW_MIN_C0 && (uint64_t)s_c0 * PW_MAX_CYCLE >= total && R.beg + total < (1ull << 32)) {
			const uint32_t len = total / s_c0;                         // average cycle length if the first region's cycles were all (>= 2: a cycle pops its own hole and one that ends it)
			ncpr = 64u;                                               // cycles followed at a time: fewer when they are long (one lane walks a block again)
			while (ncpr > (uint32_t)PW_MIN_CPR && ncpr * len > (uint32_t)PW_BLK_HOLES) ncpr >>= 1;
			uint32_t rounds = (2u * (nh + 1u) + ncpr * len - 1u) / (ncpr * len);   // rounds per block: a snapshot (nh + 1 words) for at least twice as many holes
			if (rounds > 16u) rounds = 16u;
			const uint32_t bc = ncpr * rounds;
			// snapshot slots: an estimate (the later base regions start cycles too) - a range that runs out finishes with one lane
			uint32_t cyc = 2u * s_c0 + nh;
			if (cyc > total / 2u) cyc = total / 2u;
			slots = cyc / bc + 4u;
#ifdef PW_TEST_FEW_SLOTS
			slots = slots / 8u + 3u;                                  // (test builds: ranges run out of slots and finish with one lane)
#endif
			const uint32_t need = slots * (nh + 1u);
			const unsigned long long o64 = atomicAdd(reinterpret_cast<unsigned long long*>(C.hdr + 18), (unsigned long long)need);   // (hdr[18..19]: one 64-bit cursor)
			if (ncpr * len <= 2u * (uint32_t)PW_BLK_HOLES && o64 + need <= (unsigned long long)C.pw_words) { pw = bc; off = (uint32_t)o64; }
		}
		M.pw = pw; M.pw_ncpr = ncpr; M.pw_off = off; M.pw_slots = slots;
	}
}

// K6: the holes in position order: digit of the record, position
__global__ __launch_bounds__(NT) void k_bs_holes(bs_ctx C)
{
	__shared__ uint32_t s_r;
	__shared__ uint32_t s_start[257], s_cw[BS_CW_WORDS];
	const uint32_t n_rng = C.hdr[0], tid = threadIdx.x;
	if (blockIdx.x >= C.hdr[1]) return;
	const uint32_t r = bs_find_range(C, blockIdx.x, n_rng, &s_r);
	const bs_range R = C.rng[0][r];
	if (!R.exact) return;
	const bs_meta &M = C.meta[r];
	__shared__ uint8_t s_dmap[256];
	s_start[tid] = M.start[tid]; s_dmap[tid] = M.dmap[tid];
	if (tid == 0) s_start[256] = M.start[256];
	__syncthreads();
	const uint32_t t0 = (blockIdx.x - R.tile0) * BS_TILE, hbase = C.tile_h[blockIdx.x];
	bs_cls q;
	bs_classify(C.dg + R.beg, s_start, t0, R.n, s_cw, q);
	uint8_t *hd = C.hd + R.beg;
	uint32_t *hp = C.hp + R.beg;
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		if (p < R.n && q.d[it] != q.b[it]) { const uint32_t g = hbase + q.hb[it]; hd[g] = s_dmap[q.d[it]]; hp[g] = p; }   // (a misplaced record's own region has a hole for it)
	}
}

// ------------------------------------------------------------------------------------------------ K7: the token walk
// The regions with holes are renumbered 0 .. nh-1 (digit order); the hole streams hold these dense numbers.
//
// k_bs_walk_lanes<NHM, L>: ONE LANE per range (L walkers per wavefront, ranges with up to NHM regions).  A wavefront issues
// one instruction for all its walkers, so many walks advance at the price of one; a walker's state - per region the
// pointer and a 16-byte window of the hole digits waiting there - sits in its LDS column.  The walkers of a wavefront
// advance in blocks of 8 steps on LDS alone, then the windows that are more than half used are reloaded from HBM/L2, all
// loads in flight together: one memory round trip per block, one load per >= 9 pops of a region.
// With tens of thousands of ranges per level (two per read on a large index) the serial walks run at the chip's
// throughput instead of one walk's latency.
// k_bs_walk_wave: one WAVEFRONT per range - nh == 2 is closed form (the i-th hole of the lower region trades with the
// i-th of the upper one; all lanes); otherwise (levels with few ranges) lane 0 walks with the streams staged through
// LDS windows (a ring per region, filled in aligned half-window blocks by all lanes) and one LDS read on the
// dependency chain per step (s_head[region] = pointer + the digit waiting there).
#ifndef BS_LANES_MIN_RANGES
#define BS_LANES_MIN_RANGES 262144          // ranges of a level from which the one-lane-per-range walkers are used (measured on MI355X: below ~10^5 ranges a
                                            // wavefront per range wins - its window refills are coalesced and rare, the lanes' are one cache line per region)
#endif
#define BS_LSTEPS 8
#ifndef BS_MULTI_MIN_RANGES
#define BS_MULTI_MIN_RANGES 1024            // ranges of a level from which several walks share a wavefront (measured with three sub-batch streams sharing the chip: what counts there is the issue time a level takes from the others, and G walks per wavefront cost one walk's)
#endif
#ifndef BS_MW_G16_FROM
#define BS_MW_G16_FROM 8192                 // more ranges than this: 16 walks per wavefront, ...
#define BS_MW_G32_FROM 16384                // ... 32
#endif
#define BS_MW_NHM 24                        // ... for ranges with up to this many regions that have holes (targets x strands of a large index)

template <int NHM, int L>
__global__ __launch_bounds__(64) void k_bs_walk_lanes(bs_ctx C, uint32_t nh_lo)
{
	__shared__ uint32_t s_ptr[NHM * L], s_wb[NHM * L];
	__shared__ uint64_t s_w0[NHM * L], s_w1[NHM * L];              // 16 digits from s_wb on
	const uint32_t lane = threadIdx.x, r = blockIdx.x * L + lane, n_rng = C.hdr[0];
	bool mine = lane < (uint32_t)L && r < n_rng;
	uint32_t nh = 0;
	if (mine) { nh = C.meta[r].nh; mine = nh > nh_lo && nh <= (uint32_t)NHM && nh > 2 && !C.meta[r].pw; }
	if (__ballot(mine) == 0) return;
	const bs_range R = C.rng[0][mine ? r : 0];
	bs_meta &M = C.meta[mine ? r : 0];
	if (!mine) nh = 0;
	const uint8_t *hd = C.hd + R.beg;
	uint32_t *dest = C.dest + R.beg;
	#define LW(q) ((q) * (uint32_t)L + lane)
	// the 16 digits from pointer p_ on: three aligned 8-byte loads (one or two cache lines), joined after they arrive
	#define BS_WLOAD(p_, a0_, a1_, a2_, sh_) do { const uintptr_t ua_ = (uintptr_t)(hd + (p_)); const uint64_t *al_ = (const uint64_t*)(ua_ & ~(uintptr_t)7); \
		sh_ = (uint32_t)(ua_ & 7u) * 8u; a0_ = al_[0]; a1_ = al_[1]; a2_ = al_[2]; } while (0)
	#define BS_WJOIN(lo_, hi_, sh_) ((sh_) ? ((lo_) >> (sh_)) | ((hi_) << (64u - (sh_))) : (lo_))
	for (uint32_t q = 0; q < nh; ++q) {
		const uint32_t p = M.hst[M.act[q]];
		uint64_t a0, a1, a2; uint32_t sh;
		BS_WLOAD(p, a0, a1, a2, sh);
		s_ptr[LW(q)] = p; s_wb[LW(q)] = p; s_w0[LW(q)] = BS_WJOIN(a0, a1, sh); s_w1[LW(q)] = BS_WJOIN(a1, a2, sh);
	}
	uint32_t k = 0, ek = 0, i0 = 0, i = 0, d = 0;
	bool inchase = false, live = mine;
	if (live) ek = M.hst[M.act[0] + 1u];
	while (__ballot(live)) {
		uint64_t hist = 0;
		uint32_t npop = 0;
#pragma unroll
		for (int st = 0; st < BS_LSTEPS; ++st) {
			if (live) {
				uint32_t q = d;
				if (!inchase) {
					uint32_t pk = s_ptr[LW(k)];
					while (pk >= ek) {	// region k is complete: the walk turns to the next one (arrivals so far = its J)
						if (++k >= nh) { live = false; break; }
						const uint32_t dk = M.act[k];
						pk = s_ptr[LW(k)]; ek = M.hst[dk + 1u];
						M.J[dk] = pk - M.hst[dk];
					}
					q = k;
				}
				if (live) {
					const uint32_t j = s_ptr[LW(q)];
					s_ptr[LW(q)] = j + 1u;
					if (inchase) dest[i] = j; else i0 = j;
					const uint32_t off = j - s_wb[LW(q)];
					const uint64_t w = off & 8u ? s_w1[LW(q)] : s_w0[LW(q)];
					d = (uint32_t)(w >> ((off & 7u) * 8u)) & 255u;
					hist |= (uint64_t)q << (8u * npop);
					++npop;
					i = j;
					inchase = d != k;
					if (!inchase) dest[i] = i0;
				}
			}
		}
		// A window holds 16 digits and a block pops at most 8 of a region: the windows of the regions popped in this block
		// that are more than half used are reloaded (every load first, then the LDS stores) - one reload per >= 9 pops
		uint64_t a0[BS_LSTEPS], a1[BS_LSTEPS], a2[BS_LSTEPS];
		uint32_t sh[BS_LSTEPS], pp[BS_LSTEPS];
		bool rl[BS_LSTEPS];
#pragma unroll
		for (int t = 0; t < BS_LSTEPS; ++t) {
			a0[t] = 0; a1[t] = 0; a2[t] = 0; sh[t] = 0; pp[t] = 0; rl[t] = false;
			if ((uint32_t)t < npop) {
				const uint32_t q = (uint32_t)(hist >> (8 * t)) & 255u;
				pp[t] = s_ptr[LW(q)];
				rl[t] = pp[t] - s_wb[LW(q)] > 8u;
				if (rl[t]) BS_WLOAD(pp[t], a0[t], a1[t], a2[t], sh[t]);
			}
		}
#pragma unroll
		for (int t = 0; t < BS_LSTEPS; ++t)
			if (rl[t]) { const uint32_t q = (uint32_t)(hist >> (8 * t)) & 255u; s_w0[LW(q)] = BS_WJOIN(a0[t], a1[t], sh[t]); s_w1[LW(q)] = BS_WJOIN(a1[t], a2[t], sh[t]); s_wb[LW(q)] = pp[t]; }
	}
	#undef LW
	#undef BS_WLOAD
	#undef BS_WJOIN
}

#define BS_INVALID (1u << 31)
struct bs_walk_state { uint32_t k, i0, i, d, inchase, done; };

__global__ __launch_bounds__(64) void k_bs_walk_wave(bs_ctx C, uint32_t skip_lo, uint32_t skip_hi)   // ranges with skip_lo <= regions with holes <= skip_hi are another kernel's
{
	__shared__ uint8_t s_win[BS_WIN_BYTES];
	__shared__ uint64_t s_head[256];                               // ptr | (digit | invalid << 31) << 32, by dense region
	__shared__ uint32_t s_lim[256], s_end[256], s_h0[256];
	__shared__ bs_walk_state S;
	const uint32_t r = blockIdx.x, lane = threadIdx.x;
	if (r >= C.hdr[0]) return;
	const bs_range R = C.rng[0][r];
	bs_meta &M = C.meta[r];
	const uint32_t nh = M.nh;
	if (nh == 0 || (nh >= skip_lo && nh <= skip_hi) || M.pw) return;                // (M.pw: the block-parallel walk's)
	for (uint32_t q = lane; q < nh; q += 64) { const uint32_t dk = M.act[q]; s_h0[q] = M.hst[dk]; s_end[q] = M.hst[dk + 1u]; }
	__syncthreads();
	const uint8_t *hd = C.hd + R.beg;
	uint32_t *dest = C.dest + R.beg;
	if (nh == 2) {
		const uint32_t hA = s_h0[0], hB = s_h0[1], m = s_end[0] - hA;
		for (uint32_t i = lane; i < m; i += 64) { dest[hA + i] = hB + i; dest[hB + i] = hA + i; }
		if (lane == 0) M.J[M.act[1]] = m;
		return;
	}
	uint32_t logW = 1;                                             // window entries per region: 2^logW, nh << logW <= BS_WIN_BYTES (>= 512)
	while ((nh << (logW + 1)) <= (uint32_t)BS_WIN_BYTES && logW < 16) ++logW;
	const uint32_t Wm = (1u << logW) - 1, logH = logW - 1, Hm = (1u << logH) - 1;
	// a region's loaded stretch ends at s_lim (a multiple of the half window; it may lie beyond the region's end: nothing
	// past the end is ever popped); blocks (ptr >> logH) and the one after it fit the ring
	for (uint32_t q = lane; q < nh; q += 64) { s_head[q] = (uint64_t)s_h0[q] | (uint64_t)BS_INVALID << 32; s_lim[q] = (s_h0[q] >> logH) << logH; }
	if (lane == 0) { S.k = 0; S.i0 = 0; S.i = 0; S.d = 0; S.inchase = 0; S.done = 0; }
	__syncthreads();
	for (;;) {
		for (uint32_t q = 0; q < nh; ++q) {
			const uint32_t pt = (uint32_t)s_head[q], lim = s_lim[q], en = s_end[q];
			const uint32_t tgt = ((pt >> logH) + 2u) << logH;
			if (pt >= en || lim >= tgt) continue;
			const uint32_t base = q << logW, top = tgt < en ? tgt : en;
			for (uint32_t g = lim + lane; g < top; g += 64) s_win[base + (g & Wm)] = hd[g];
		}
		__syncthreads();
		if (lane == 0) {
			for (uint32_t q = 0; q < nh; ++q) {
				const uint32_t pt = (uint32_t)s_head[q], en = s_end[q];
				if (pt >= en) continue;
				const uint32_t tgt = ((pt >> logH) + 2u) << logH;
				if (s_lim[q] < tgt) s_lim[q] = tgt;
				s_head[q] = (uint64_t)pt | (uint64_t)s_win[(q << logW) + (pt & Wm)] << 32;
			}
			// the walk (ksort.h:124-138), until it is finished or needs a record beyond a window
			uint32_t k = S.k, i0 = S.i0, i = S.i, d = S.d;
			bool inchase = S.inchase != 0, need = false;
			#define BS_ADVANCE(b, j) do { \
				const uint32_t jn_ = (j) + 1u; \
				const uint32_t nx_ = s_win[((b) << logW) + (jn_ & Wm)]; \
				const uint32_t inv_ = ((jn_ & Hm) == 0u && jn_ >= s_lim[b]) ? BS_INVALID : 0u; \
				s_head[b] = (uint64_t)jn_ | (uint64_t)(nx_ | inv_) << 32; } while (0)
			while (k < nh) {
				if (!inchase) {
					const uint64_t h = s_head[k];
					const uint32_t pk = (uint32_t)h, hi = (uint32_t)(h >> 32);
					if (pk >= s_end[k]) { ++k; if (k < nh) M.J[M.act[k]] = (uint32_t)s_head[k] - s_h0[k]; continue; }
					if (hi & BS_INVALID) { need = true; break; }
					BS_ADVANCE(k, pk);
					d = hi & 255u;
					i0 = i = pk;
					inchase = true;
				}
				while (d != k) {
					const uint64_t h = s_head[d];
					const uint32_t j = (uint32_t)h, hi = (uint32_t)(h >> 32);
					if (hi & BS_INVALID) { need = true; break; }
					BS_ADVANCE(d, j);
					dest[i] = j;
					i = j; d = hi & 255u;
				}
				if (need) break;
				dest[i] = i0;
				inchase = false;
			}
			#undef BS_ADVANCE
			S.k = k; S.i0 = i0; S.i = i; S.d = d; S.inchase = inchase ? 1u : 0u; S.done = need ? 0u : 1u;
		}
		__syncthreads();
		if (S.done) break;
	}
}

// k_bs_walk_multi<NHM, G>: G ranges per wavefront (ranges with 3 .. NHM regions that have holes) - lanes 0 .. G-1 walk, one
// range each, one pop per iteration; the walk is bound by instruction issue (a wavefront pays for 64 lanes whatever is active),
// and here G walks share every instruction.  The hole-digit streams come through a 64-byte LDS ring per (walker, region),
// indexed by the hole's absolute address so that a refill is one aligned 16-byte load + one 16-byte LDS store; the refills are
// all lanes' work (pair p = walker * NHM + region belongs to lane p & 63) and are software pipelined: every BS_MW_PERIOD
// iterations the loads for the stretch after next are issued, and committed one period later, so that HBM latency is never
// waited for.  A walker that does run into the end of its ring (flagged in the head) idles until the next commit.
#ifndef BS_MW_PERIOD
#define BS_MW_PERIOD 32
#endif
#define BS_MW_RING 64
template <int NHM, int G>
__global__ __launch_bounds__(64) void k_bs_walk_multi(bs_ctx C)
{
	static_assert((NHM * G) % 64 == 0, "whole rows of (walker, region) pairs");
	constexpr int NP = NHM * G / 64;                              // pairs per lane: p = lane + 64 t
	__shared__ __attribute__((aligned(16))) uint8_t s_win[G * NHM * BS_MW_RING];
	__shared__ uint64_t s_head[G * NHM];                           // hole index | (digit | invalid << 31) << 32
	__shared__ uint32_t s_lim[G * NHM], s_end[G * NHM];            // committed up to (absolute, multiple of 16) / end of the region's holes (hole index)
	__shared__ uint32_t s_nh[G], s_beg[G];
	const uint32_t lane = threadIdx.x, n_rng = C.hdr[0], r0 = blockIdx.x * G;
	if (lane < (uint32_t)G) {
		uint32_t nh = 0, beg = 0;
		if (r0 + lane < n_rng) { nh = C.meta[r0 + lane].nh; beg = C.rng[0][r0 + lane].beg; if (nh < 3 || nh > (uint32_t)NHM || C.meta[r0 + lane].pw) nh = 0; }
		s_nh[lane] = nh; s_beg[lane] = beg;
	}
	__syncthreads();
	{ bool any = false; for (int g = 0; g < G; ++g) any |= s_nh[g] != 0; if (!any) return; }
	// this lane's pairs (registers: every loop over t below is fully unrolled)
	bool p_on[NP];
	uint32_t p_end[NP], p_abs0[NP], ld_n[NP];
	uint4 la[NP], lb[NP];
#pragma unroll
	for (int t = 0; t < NP; ++t) {
		const uint32_t p = lane + 64u * (uint32_t)t, g = p / NHM, q = p % NHM;
		p_on[t] = q < s_nh[g];
		p_end[t] = 0; p_abs0[t] = 0; ld_n[t] = 0; la[t] = uint4{0, 0, 0, 0}; lb[t] = uint4{0, 0, 0, 0};
		if (p_on[t]) {
			const bs_meta &Mg = C.meta[r0 + g];
			const uint32_t dk = Mg.act[q], h0 = Mg.hst[dk], h1 = Mg.hst[dk + 1u], beg = s_beg[g];
			s_head[p] = (uint64_t)h0 | (uint64_t)BS_INVALID << 32; s_end[p] = h1; s_lim[p] = (beg + h0) & ~15u;
			p_end[t] = beg + h1; p_abs0[t] = beg;
		}
	}
	__syncthreads();
	// chunks (16 bytes at s_lim, s_lim + 16) a pair can take now: the ring holds the 64 bytes from the pointer's chunk on
	#define BS_MW_ISSUE_ALL() _Pragma("unroll") for (int t = 0; t < NP; ++t) { \
		ld_n[t] = 0; \
		if (p_on[t]) { const uint32_t p = lane + 64u * (uint32_t)t, pa = ((uint32_t)s_head[p] + p_abs0[t]) & ~15u, lm = s_lim[p]; \
		               if (lm < p_end[t] && lm + 16u - pa <= (uint32_t)BS_MW_RING) { ld_n[t] = 1; if (lm + 16u < p_end[t] && lm + 32u - pa <= (uint32_t)BS_MW_RING) ld_n[t] = 2; } \
		               if (ld_n[t] >= 1) la[t] = *reinterpret_cast<const uint4*>(C.hd + lm); \
		               if (ld_n[t] >= 2) lb[t] = *reinterpret_cast<const uint4*>(C.hd + lm + 16u); } }
	#define BS_MW_COMMIT_ALL() _Pragma("unroll") for (int t = 0; t < NP; ++t) { \
		if (p_on[t] && ld_n[t]) { const uint32_t p = lane + 64u * (uint32_t)t, lm = s_lim[p]; uint8_t *ring = s_win + (size_t)p * BS_MW_RING; \
		               *reinterpret_cast<uint4*>(ring + (lm & (BS_MW_RING - 1u))) = la[t]; \
		               if (ld_n[t] == 2) *reinterpret_cast<uint4*>(ring + ((lm + 16u) & (BS_MW_RING - 1u))) = lb[t]; \
		               const uint32_t nl = lm + 16u * ld_n[t]; s_lim[p] = nl; \
		               const uint64_t hh = s_head[p]; const uint32_t aa = (uint32_t)hh + p_abs0[t]; \
		               if (((uint32_t)(hh >> 32) & BS_INVALID) && aa < nl) s_head[p] = (uint64_t)(uint32_t)hh | (uint64_t)ring[aa & (BS_MW_RING - 1u)] << 32; } }
	BS_MW_ISSUE_ALL() BS_MW_COMMIT_ALL()
	__syncthreads();
	BS_MW_ISSUE_ALL() BS_MW_COMMIT_ALL()                            // rings full: 64 bytes from each region's first hole
	__syncthreads();
	const bool walker = lane < (uint32_t)G && s_nh[lane < (uint32_t)G ? lane : 0] != 0;
	const uint32_t g = walker ? lane : 0, nh = walker ? s_nh[g] : 0, beg = s_beg[g];
	bs_meta &M = C.meta[r0 + g];
	uint32_t *dest = C.dest + beg;
	uint64_t *hd_ = s_head + g * NHM;
	const uint32_t *lim_ = s_lim + g * NHM, *end_ = s_end + g * NHM;
	const uint8_t *win_ = s_win + (size_t)g * NHM * BS_MW_RING;
	// The walk, one pop per iteration: (q, h) = the region to pop next and its head, already in registers.  A lone wavefront
	// gets a dependent instruction through every ~8 cycles, so the iteration is kept short: the common case (every live walker
	// pops) is branch-free selects; a region running out of holes and a walker at the end of its ring are the wavefront's slow
	// path, taken only when a lane needs it.  The head of the next region is requested before this pop's head update is written
	// back (LDS answers in order: one round trip per step); the store that closes a cycle waits for the next iteration, so that
	// an iteration stores once.
	uint32_t k = 0, i0 = 0, i = 0, q = 0, endk = walker ? end_[0] : 0, pend_a = 0, pend_v = 0;
	bool inchase = false, live = walker, pend = false;
	uint64_t h = walker ? hd_[0] : 0;
	for (;;) {
		BS_MW_ISSUE_ALL()
#pragma unroll 4
		for (int it = 0; it < BS_MW_PERIOD; ++it) {
			const uint32_t j = (uint32_t)h, hi = (uint32_t)(h >> 32);
			const bool out = !inchase && j >= endk, stall = (hi & BS_INVALID) != 0;
			if (__ballot(live && (out || stall))) {               // slow path (whole wavefront, rarely)
				if (live && out) {                                  // region k has no hole left: the next one becomes the base
					if (pend) { dest[pend_a] = pend_v; pend = false; }
					++k;
					if (k >= nh) live = false;
					else { M.J[M.act[k]] = (uint32_t)hd_[k] - M.hst[M.act[k]]; endk = end_[k]; q = k; h = hd_[k]; }
				}
				continue;                                           // (a stalled walker waits for the next commit; the others lose one turn)
			}
			if (live) {
				const uint32_t jn = j + 1u, an = jn + beg, d = hi & 255u;   // the record found in the hole belongs to region d: it goes there next
				const bool closes = d == k;                       // ... unless that is the cycle's own region (never for the pop that starts a cycle)
				const uint32_t nx = win_[q * BS_MW_RING + (an & (BS_MW_RING - 1u))];
				const uint32_t lm = lim_[q];
				const uint64_t hn = hd_[d];                         // (stale if d == q: replaced below)
				const uint64_t hq = (uint64_t)jn | (uint64_t)(nx | (((an & 15u) == 0u && an >= lm) ? BS_INVALID : 0u)) << 32;
				hd_[q] = hq;
				const uint32_t sa = inchase ? i : pend_a, sv = inchase ? j : pend_v;
				if (inchase || pend) dest[sa] = sv;
				i0 = inchase ? i0 : j;
				pend = closes; pend_a = j; pend_v = i0;
				i = j;
				inchase = !closes;
				h = d == q ? hq : hn;
				q = d;
			}
		}
		if (walker && !live && pend) { dest[pend_a] = pend_v; pend = false; }
		__syncthreads();
		if (!__ballot(live)) break;
		BS_MW_COMMIT_ALL()
		__syncthreads();
		if (live) h = hd_[q];                                      // (a commit may have made it valid)
	}
	#undef BS_MW_ISSUE_ALL
	#undef BS_MW_COMMIT_ALL
}

// k_bs_walk_tok<RPL>: one WAVEFRONT per range, one LANE per region (RPL regions per lane: up to 64 * RPL regions with holes), and
// the token itself in scalar registers.  A region's state - the index of its next hole and the digits waiting in it and in the one
// after it - are VGPRs of its lane; a step is two v_readlane (digit and hole of the region the token stands on) and a handful of
// scalar instructions, the only dependency chain being v_readlane -> SGPR -> lane select of the next v_readlane (~60 cycles on
// gfx950, against ~800 for a step of the LDS-resident walkers above: a level costs its LONGEST range x the step time, and the late
// rounds' unmappable reads have ranges of 10^5 holes).  The lane of the popped region advances on its own: the digit after next
// comes from its 64-byte LDS ring two pops of the region ahead of its use, so no LDS latency is on the chain (consecutive pops are
// never in the same region).  The rings are refilled as in k_bs_walk_multi - aligned 16-byte loads issued at the start of a period
// of BS_TK_PERIOD = 32 pops, committed at its end - and never run dry: a region is popped at most every other step, 16 times a
// period, and a commit leaves >= 33 digits (or the rest of the region) ahead of its pointer as it was when the loads were issued.
// dest[] leaves through two log registers (lane t = the t-th pop's hole and successor), stored 64 entries at a time.
// Everything the token touches must stay wave-uniform: one lane-dependent branch out of the walk loop and the compiler keeps every
// token register in a VGPR (a v_readfirstlane per use).
#define BS_TK_PERIOD 32
template <int RPL, int ADV = 0>
__global__ __launch_bounds__(64) void k_bs_walk_tok(bs_ctx C, uint32_t nh_lo, uint32_t nh_hi)
{
	static_assert(BS_TK_PERIOD == 32 && BS_MW_RING == 64, "the no-underflow argument above is for these");
	__shared__ __attribute__((aligned(16))) uint8_t s_win[RPL * 64 * BS_MW_RING];
	const uint32_t lane = threadIdx.x, r = blockIdx.x;
	if (r >= C.hdr[0]) return;
	bs_meta &M = C.meta[r];
	const uint32_t nh = M.nh;
	if (nh < nh_lo || nh > nh_hi || M.pw) return;
	const uint32_t beg = (uint32_t)C.rng[0][r].beg;                 // (the host takes this kernel only when every absolute hole address fits 32 bits)
	// this lane's regions: q = lane + 64 t
	bool on[RPL];
	uint32_t jr[RPL], en[RPL], lim[RPL], head[RPL], ld_n[RPL];   // next hole and end of the region's holes (ABSOLUTE hole addresses: the ring slot of a hole is its low six bits), ring committed up to, digit at jr
	uint4 la[RPL], lb[RPL];
#pragma unroll
	for (int t = 0; t < RPL; ++t) {
		const uint32_t q = lane + 64u * (uint32_t)t;
		on[t] = q < nh;
		jr[t] = 0; en[t] = 0; lim[t] = 0; head[t] = 0; ld_n[t] = 0; la[t] = uint4{0, 0, 0, 0}; lb[t] = uint4{0, 0, 0, 0};
		if (on[t]) { const uint32_t dk = M.act[q]; jr[t] = beg + M.hst[dk]; en[t] = beg + M.hst[dk + 1u]; lim[t] = jr[t] & ~15u; }
	}
	#define BS_TK_RING(t) ((uint32_t)(lane + 64u * (uint32_t)(t)) * (uint32_t)BS_MW_RING)
	uint32_t ringb[RPL];
#pragma unroll
	for (int t = 0; t < RPL; ++t) ringb[t] = BS_TK_RING(t);
	const uint32_t ringa0 = rh_lds_addr(s_win) + ringb[0];         // LDS byte address of this lane's first ring
	#define BS_TK_ISSUE() _Pragma("unroll") for (int t = 0; t < RPL; ++t) { \
		ld_n[t] = 0; \
		if (on[t]) { const uint32_t pa_ = jr[t] & ~15u, lm_ = lim[t], ea_ = en[t]; \
		             if (lm_ < ea_ && lm_ + 16u - pa_ <= (uint32_t)BS_MW_RING) { ld_n[t] = 1; if (lm_ + 16u < ea_ && lm_ + 32u - pa_ <= (uint32_t)BS_MW_RING) ld_n[t] = 2; } \
		             if (ld_n[t] >= 1) la[t] = *reinterpret_cast<const uint4*>(C.hd + lm_); \
		             if (ld_n[t] >= 2) lb[t] = *reinterpret_cast<const uint4*>(C.hd + lm_ + 16u); } }
	#define BS_TK_COMMIT() do { _Pragma("unroll") for (int t = 0; t < RPL; ++t) { \
		if (on[t] && ld_n[t]) { const uint32_t lm_ = lim[t]; \
		             *reinterpret_cast<uint4*>(s_win + BS_TK_RING(t) + (lm_ & (BS_MW_RING - 1u))) = la[t]; \
		             if (ld_n[t] == 2) *reinterpret_cast<uint4*>(s_win + BS_TK_RING(t) + ((lm_ + 16u) & (BS_MW_RING - 1u))) = lb[t]; \
		             lim[t] = lm_ + 16u * ld_n[t]; } } \
		RH_WAVE_SYNC(); } while (0)
	#define BS_TK_READ(arr, sl, ln) (RPL == 1 ? rh_readlane(arr[0], ln) : (sl) == 0u ? rh_readlane(arr[0], ln) : (sl) == 1u ? rh_readlane(arr[RPL > 1 ? 1 : 0], ln) : (sl) == 2u ? rh_readlane(arr[RPL > 2 ? 2 : 0], ln) : rh_readlane(arr[RPL > 3 ? 3 : 0], ln))
	BS_TK_ISSUE() BS_TK_COMMIT();
	BS_TK_ISSUE() BS_TK_COMMIT();                                    // rings full: 64 bytes from each region's first hole on
#pragma unroll
	for (int t = 0; t < RPL; ++t)
		if (on[t]) head[t] = s_win[BS_TK_RING(t) | (jr[t] & 63u)];
	BS_TK_ISSUE()
	// The token.  The loop nest is the reference's (ksort.h:124-138): for each base region k, for each of its holes, chase the
	// record found there until one that belongs to k turns up.
	uint32_t k = 0, nlog = 0, per = BS_TK_PERIOD;
	uint32_t logA = 0, logV = 0;
	// ADV (fewer instructions a pop - the walk is bound by the number of instructions a SIMD issues, whatever their kind):
	//  * ONE log register.  dest[] of a cycle i0 -> j1 -> .. -> jm -> i0 is a chain (the record of a hole lands in the next hole popped), so the
	//    holes are logged in pop order, a cycle as [i0 (marked in smask: nothing lands from the entry before it), j1, .., jm, i0], and a block is
	//    stored as dest[entry t-1] = entry t (the lane below by a DPP shift, lane 0 from the previous block's last entry);
	//  * one counter for the log block and the ring period: a block is 32 entries, and a cycle has one entry more than pops, so a period is at most
	//    32 pops and the no-underflow argument above holds; the period check leaves the pop.
	uint32_t carry = 0, smask = 0;
	#define BS_TK_FLUSH1(n_) do { const uint32_t pv_ = rh_wave_shr1(logV, carry); if (lane < (n_) && !((smask >> lane) & 1u)) C.dest[pv_] = logV - beg; } while (0)
	#define BS_TK_LOG(a_, v_) do { \
		if (ADV) { logV = rh_writelane(logV, (v_), nlog); \
		           if (__builtin_expect(++nlog == 32u, 0)) { BS_TK_COMMIT(); BS_TK_ISSUE() BS_TK_FLUSH1(32u); carry = rh_readlane(logV, 31u); smask = 0; nlog = 0; } } \
		else { rh_writelane2(logA, logV, (a_), (v_), nlog); if (++nlog == 64u) { C.dest[logA] = logV - beg; nlog = 0; } } } while (0)   /* (dest[] holds hole indices within the range) */
	// pop region c_: d_ = the digit in its next hole, j_ = that hole; the region's lane moves on
	#define BS_TK_POP(c_, d_, j_) do { \
		const uint32_t l_ = RPL == 1 ? (c_) : (c_) & 63u, sl_ = RPL == 1 ? 0u : (c_) >> 6;   /* (one region per lane: c_ < nh <= 64) */ \
		if (!ADV) { if (per == 0u) { BS_TK_COMMIT(); BS_TK_ISSUE() per = BS_TK_PERIOD; }   /* a period is over: commit the loads in flight, issue the next */ \
		            --per; } \
		if (RPL == 1 && ADV) rh_lds_wait(head[0]); \
		d_ = BS_TK_READ(head, sl_, l_); j_ = BS_TK_READ(jr, sl_, l_); \
		if (RPL == 1 && ADV) rh_tok_advance(jr[0], head[0], s_win + ringb[0], ringa0, l_, lane);   /* (rh_gpu.h) */ \
		else { _Pragma("unroll") for (int t = 0; t < RPL; ++t) \
			if (lane == l_ && sl_ == (uint32_t)t) { \
				jr[t] += 1u; \
				head[t] = s_win[rh_and_or(jr[t], 63u, ringb[t])];   /* (wanted at this region's next pop, at least a step away) */ \
			} } \
		} while (0)
	while (k < nh) {
		const uint32_t lk = k & 63u, sk = RPL == 1 ? 0u : k >> 6;
		const uint32_t endk = BS_TK_READ(en, sk, lk);
		if (k) { const uint32_t jk = BS_TK_READ(jr, sk, lk); if (lane == 0) { const uint32_t dk = M.act[k]; M.J[dk] = jk - beg - M.hst[dk]; } }   // arrivals so far = J
		while (BS_TK_READ(jr, sk, lk) < endk) {                      // until region k has no hole left (then the next one becomes the base)
			uint32_t d, i0, j, d2, j2;
			BS_TK_POP(k, d, i0);                                      // the hole that starts a cycle (its record belongs elsewhere: d != k)
			if (ADV) { smask |= 1u << nlog; BS_TK_LOG(0u, i0); }
			j2 = i0;
			for (;;) {	// (two steps an iteration: the carried values change names instead of registers)
				BS_TK_POP(d, d2, j);
				BS_TK_LOG(j2, j);                                     // the record carried from the previous hole lands in this one
				if (d2 == k) break;
				BS_TK_POP(d2, d, j2);
				BS_TK_LOG(j, j2);
				if (d == k) { j = j2; break; }
			}
			BS_TK_LOG(j, i0);                                         // ... and the one that belongs to k in the hole the cycle started from
		}
		++k;
	}
	if (ADV) BS_TK_FLUSH1(nlog); else if (lane < nlog) C.dest[logA] = logV - beg;
	#undef BS_TK_LOG
	#undef BS_TK_FLUSH1
	#undef BS_TK_POP
	#undef BS_TK_RING
	#undef BS_TK_ISSUE
	#undef BS_TK_COMMIT
	#undef BS_TK_READ
}


// ------------------------------------------------------------------------------------------------ K7': the block-parallel walk
// The token walk is a ROTOR WALK: every region hands its holes out in position order, whoever arrives.  Seen from one base region k it is
// a sequence of cycles - pop k's next hole, follow the record found there to its region, pop that region's next hole, ... until a record
// of k turns up - and networks of that kind are ABELIAN: if several cycles of one base region are followed at the same time, in any
// interleaving, every region has popped exactly as many holes at the end as if they had been followed one after the other (each region
// serves its arrivals in its own fixed order, so the NUMBER of pops per region does not depend on the schedule; only who gets which hole does).  So
//   * k_bs_pw_count (one wavefront per range) follows up to 64 cycles at a time, one per lane, popping with LDS atomics: the pointers of all
//     regions after every block of cycles - "snapshots" - are exact although the pops in between were handed out in the wrong order;
//   * k_bs_pw_walk walks every block of cycles between two snapshots again, serially and therefore in the reference's order, but ALL BLOCKS
//     AT ONCE, one lane per block (up to 64 blocks = one work item per wavefront, the digits of the item's stretch of every region staged in
//     LDS, the lane's private pointers as 16-bit window addresses): this pass writes dest[].
// A block is a fixed number of cycles (fewer per round and per block the longer the range's cycles are: k_bs_scan) and runs across base regions;
// a block that pops more holes than an item's window takes, and whatever is left when a range runs out of snapshot slots, are walked by one
// lane of k_bs_pw_count itself.  Measured at human scale (one level of the backtrack candidates' sort: 6 500 - 19 000 ranges of 10 000 -
// 40 000 holes, the lowest score - the chains of one anchor - holding most records, so that every second hole elsewhere ends a cycle; the
// exact re-sorts of the reads with equal anchor keys: 24 or 256 evenly filled regions, up to 10^5 holes): the serial walkers took one
// wavefront 60 - 250 ns per hole and a level as long as its longest range.
#define PW_CHUNKS (PW_RING_BYTES / 16)
#define PW_NLD (PW_CHUNKS / 64)
template <int NHM>
__global__ __launch_bounds__(64) void k_bs_pw_count(bs_ctx C)
{
	static_assert(PW_CHUNKS >= 256 && PW_CHUNKS % 64 == 0 && (PW_RING_BYTES & (PW_RING_BYTES - 1)) == 0 && PW_WIN_CAP < 65000, "a chunk for each of 256 regions; window addresses are 16 bits");
	constexpr int RPL = NHM / 64, CLS = NHM > 64 ? 1 : 0;
	// every region's window: the digits of its holes [wst, lim) at s_win[hole + off]; capacities in proportion to the regions' holes
	__shared__ __attribute__((aligned(16))) uint8_t s_win[PW_RING_BYTES];
	__shared__ uint32_t s_ptr[NHM], s_end[NHM], s_lim[NHM], s_off[NHM], s_wst[NHM], s_base[NHM];   // next hole, end of the region's holes, window end (absolute hole addresses); window offset; window start; its place in s_win
	__shared__ uint8_t s_map[PW_CHUNKS];                              // 16-byte chunk of s_win -> region
	__shared__ uint32_t s_pops, s_err;
	const uint32_t lane = threadIdx.x, r = blockIdx.x;
	if (r >= C.hdr[0]) return;
	bs_meta &M = C.meta[r];
	const uint32_t bcyc = M.pw, nh = M.nh, ncpr = M.pw_ncpr;
	if (!bcyc || nh > (uint32_t)NHM || (CLS == 1 && nh <= 64u)) return;
	const uint32_t beg = (uint32_t)C.rng[0][r].beg, stride = nh + 1u, n_slots = M.pw_slots, n_holes = M.hst[256];
	uint32_t *snap = C.pw_snap + M.pw_off;
	const uint8_t *hd = C.hd;
	uint32_t *dest = C.dest;
	bool on[RPL];
	uint32_t cap[RPL], wbase[RPL], thr[RPL], snp[RPL];
	uint32_t used = 0;
	{
		const uint32_t mc = (uint32_t)PW_CHUNKS >= 2u * nh ? 2u : 1u, extra = (uint32_t)PW_CHUNKS - mc * nh;
		uint32_t run = 0;
#pragma unroll
		for (int t = 0; t < RPL; ++t) {
			const uint32_t q = lane + 64u * (uint32_t)t;
			on[t] = q < nh; cap[t] = 0; wbase[t] = 0; thr[t] = 0; snp[t] = 0;
			uint32_t nc = 0;
			if (on[t]) {
				const uint32_t dk = M.act[q], h0 = M.hst[dk], h1 = M.hst[dk + 1u];
				s_ptr[q] = beg + h0; s_end[q] = beg + h1;
				nc = mc + (uint32_t)((uint64_t)extra * (h1 - h0) / n_holes);
			}
			uint32_t inc = nc;
			for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t u = __shfl_up(inc, dd); if (lane >= (uint32_t)dd) inc += u; }
			const uint32_t tot = __shfl(inc, 63);
			if (on[t]) {
				cap[t] = 16u * nc; wbase[t] = 16u * (run + inc - nc); s_base[q] = wbase[t];
				thr[t] = cap[t] / 8u < 4u ? 4u : cap[t] / 8u > 64u ? 64u : cap[t] / 8u;
				for (uint32_t c = 0; c < nc; ++c) s_map[run + inc - nc + c] = (uint8_t)q;
			}
			run += tot;
		}
		used = run;
	}
	if (lane == 0) { s_pops = 0; s_err = 0; }
	__syncthreads();
	// every region's window reloaded from its pointer on (all lanes, PW_NLD 16-byte loads each, in flight together: one trip to memory)
	#define PW_EVENT() do { \
		_Pragma("unroll") for (int t = 0; t < RPL; ++t) if (on[t]) { const uint32_t q_ = lane + 64u * (uint32_t)t; s_wst[q_] = s_ptr[q_] & ~15u; } \
		__syncthreads(); \
		uint4 v_[PW_NLD]; \
		_Pragma("unroll") for (int i = 0; i < PW_NLD; ++i) { v_[i] = uint4{0, 0, 0, 0}; const uint32_t c_ = lane + 64u * (uint32_t)i; \
			if (c_ < used) { const uint32_t q_ = s_map[c_], a_ = s_wst[q_] + (16u * c_ - s_base[q_]); if (a_ < s_end[q_]) v_[i] = *reinterpret_cast<const uint4*>(hd + a_); } } \
		_Pragma("unroll") for (int i = 0; i < PW_NLD; ++i) { const uint32_t c_ = lane + 64u * (uint32_t)i; if (c_ < used) *reinterpret_cast<uint4*>(s_win + 16u * c_) = v_[i]; } \
		_Pragma("unroll") for (int t = 0; t < RPL; ++t) if (on[t]) { const uint32_t q_ = lane + 64u * (uint32_t)t, w_ = s_wst[q_]; s_lim[q_] = w_ + cap[t]; s_off[q_] = wbase[t] - w_; } \
		__syncthreads(); PW_STAT(4, lane == 0 ? 1 : 0); } while (0)
	// one lane, the cycles of base region k from the pointers as they stand until k's pointer reaches `until`, in the reference's order (ksort.h:124-138)
	auto serial = [&](uint32_t k, uint32_t until) RH_INLINE_LAMBDA {
		if (lane == 0) {
			uint32_t steps = 0;
			while (s_ptr[k] < until && !s_err) {
				const uint32_t i0 = s_ptr[k];
				s_ptr[k] = i0 + 1u;
				uint32_t i = i0, q = k;
				for (;;) {
					const uint32_t d = i < s_lim[q] && i >= s_wst[q] ? (uint32_t)s_win[(i + s_off[q]) & (uint32_t)(PW_RING_BYTES - 1)] : (uint32_t)hd[i];   // the digit in hole i of region q
					if (d == k) break;
					if (d >= nh || s_ptr[d] >= s_end[d] || ++steps > n_holes) { s_err = 1; break; }
					const uint32_t j = s_ptr[d];
					s_ptr[d] = j + 1u;
					dest[i] = j - beg;
					i = j; q = d;
				}
				dest[i] = i0 - beg;
			}
		}
		__syncthreads();
	};
	auto serial_until = [&](uint32_t k, uint32_t until) RH_INLINE_LAMBDA {   // ... in stretches of 64 cycles, the windows reloaded in between
		for (;;) {
			const uint32_t p = rh_uniform(s_ptr[k]);
			const bool stop = p >= until || s_err;
			__syncthreads();                                           // (every lane has looked before lane 0 walks on)
			if (stop) break;
			serial(k, until - p < 64u ? until : p + 64u);
			PW_STAT(3, lane == 0 ? (until - p < 64u ? until - p : 64u) : 0);
			PW_EVENT();
		}
	};
	PW_EVENT();
	uint32_t slot = 0, it_slot0 = 0, it_nb = 0, it_pops = 0;       // (wave-uniform)
	auto emit = [&]() RH_INLINE_LAMBDA {
		if (lane == 0 && it_nb) {
			const uint32_t ix = atomicAdd(&C.hdr[16 + CLS], 1u);
			if (ix < C.pw_item_cap) { bs_pw_item I; I.r = r; I.slot0 = it_slot0; I.nb = it_nb; I.pad = 0; C.pw_items[CLS][ix] = I; }
			else C.hdr[7] = 1;
		}
		it_nb = 0; it_pops = 0;
	};
	auto snapshot = [&](uint32_t sl, uint32_t kword) RH_INLINE_LAMBDA {
#pragma unroll
		for (int t = 0; t < RPL; ++t) if (on[t]) { const uint32_t q = lane + 64u * (uint32_t)t; snp[t] = s_ptr[q]; snap[(size_t)sl * stride + q] = snp[t]; }
		if (lane == 0) snap[(size_t)sl * stride + nh] = kword;
		__syncthreads();                                               // (every lane has read its pointers before lane 0 moves the base region's)
	};
	auto close_item = [&](uint32_t kword) RH_INLINE_LAMBDA { if (it_nb) { snapshot(slot, kword); ++slot; emit(); } };   // the pointers as they stand end the open item
	bool over = false, blk_open = false;
	uint32_t blk_cyc = 0, blk_slot = 0, blk_k = 0;
	uint32_t cur_k = 0, cur_kp = 0;                                  // (where the walk stands: for the block that has to be walked again)
	// the open block is complete: the holes it popped = the sum of the pointers' advances.  false: inconsistent hole lists
	auto close_block = [&]() RH_INLINE_LAMBDA -> bool {
		{
			uint32_t adv = 0;
#pragma unroll
			for (int t = 0; t < RPL; ++t) if (on[t]) { const uint32_t q = lane + 64u * (uint32_t)t; adv += s_ptr[q] - snp[t]; }
			if (adv) atomicAdd(&s_pops, adv);
		}
		__syncthreads();
		const uint32_t bp = rh_uniform(s_pops);
		bool bad = s_err != 0;
#pragma unroll
		for (int t = 0; t < RPL; ++t) if (on[t]) { const uint32_t q = lane + 64u * (uint32_t)t; bad |= s_ptr[q] > s_end[q]; }
		__syncthreads();
		if (lane == 0) s_pops = 0;
		if (__ballot(bad)) return false;
		blk_open = false;
		if (bp > (uint32_t)PW_WIN_CAP) {
			// more holes than an item's window takes: the open item ends where this block began (its snapshot), and the block is walked
			// again by one lane from there
			emit();
			if (lane == 0) PW_STAT(0, 1);
#pragma unroll
			for (int t = 0; t < RPL; ++t) if (on[t]) { const uint32_t q = lane + 64u * (uint32_t)t; s_ptr[q] = snp[t]; }
			__syncthreads();
			PW_EVENT();
			for (uint32_t kk = blk_k; kk <= cur_k; ++kk) serial_until(kk, kk < cur_k ? rh_uniform(s_end[kk]) : cur_kp);
			if (s_err) return false;
		} else {
			if (it_nb == 64u || it_pops + bp > (uint32_t)PW_WIN_CAP) emit();   // (the open item ends at this block's snapshot, which follows its last one)
			if (it_nb == 0) it_slot0 = blk_slot;
			++it_nb; it_pops += bp;
		}
		return true;
	};
	for (uint32_t k = 0; k < nh; ++k) {
		uint32_t kp = rh_uniform(s_ptr[k]);
		const uint32_t endk = rh_uniform(s_end[k]);
		if (k && lane == 0) { const uint32_t dk = M.act[k]; M.J[dk] = kp - beg - M.hst[dk]; }   // arrivals so far = J
		while (kp < endk) {
			if (!blk_open) {
				if (over || slot + 2u > n_slots) {                        // out of snapshot slots: the rest of the range is one lane's
					if (!over) { over = true; if (lane == 0) PW_STAT(1, 1); close_item(k); }
					serial_until(k, endk);
					if (s_err) { if (lane == 0) C.hdr[7] = 2; return; }
					break;
				}
				blk_slot = slot++; blk_k = k; blk_cyc = 0; blk_open = true;
				snapshot(blk_slot, k);
			}
			// a round: up to ncpr cycles, one per lane
			const uint32_t nch = endk - kp < ncpr ? endk - kp : ncpr;
			{
				bool need = false;
#pragma unroll
				for (int t = 0; t < RPL; ++t) if (on[t]) { const uint32_t q = lane + 64u * (uint32_t)t, lm = s_lim[q]; need |= (int32_t)(lm - s_ptr[q]) < (int32_t)(q == k ? 64u : thr[t]) && lm < s_end[q]; }
				if (__ballot(need)) PW_EVENT();
			}
			bool live = lane < nch, parked = false;
			uint32_t d = k, pj = 0, bad_d = 0;
			auto take = [&](uint32_t dd, uint32_t j) RH_INLINE_LAMBDA {   // the digit in hole j of region dd: the cycle ends, goes on to that region, or waits for memory
				const bool miss = j >= s_lim[dd];
				const uint32_t dn = s_win[(j + s_off[dd]) & (uint32_t)(PW_RING_BYTES - 1)];
				if (miss) { parked = true; pj = j; live = false; PW_STAT(2, 1); }   // beyond the window (a region popped more than expected): the lane waits below
				else { bad_d |= dn >= nh ? 1u : 0u; live = dn != k && dn < nh; d = dn; }
			};
			if (live) take(k, kp + lane);
			if (lane == 0) s_ptr[k] = kp + nch;
			for (uint32_t itn = 0;;) {
				for (; __ballot(live); ++itn) {
					if (itn > n_holes) { bad_d = 1; live = false; parked = false; continue; }   // (cannot happen with consistent hole lists: every pop uses up a hole)
					if (live) take(d, atomicAdd(&s_ptr[d], 1u));
				}
				if (!__ballot(parked)) break;
				if (parked) {                                               // ... for the digit from memory, and goes on (any interleaving of the cycles is as good as any other)
					const uint32_t dn = hd[pj];
					parked = false; bad_d |= dn >= nh ? 1u : 0u; live = dn != k && dn < nh; d = dn;
				}
			}
			if (bad_d) s_err = 1;
			__syncthreads();
			kp += nch; blk_cyc += nch;
			if (blk_cyc >= bcyc) { cur_k = k; cur_kp = kp; if (!close_block()) { if (lane == 0) C.hdr[7] = 2; return; } }
		}
	}
	if (blk_open) { cur_k = nh - 1u; cur_kp = rh_uniform(s_ptr[nh - 1u]); if (!close_block()) { if (lane == 0) C.hdr[7] = 2; return; } }
	close_item(nh);
	#undef PW_EVENT
}

template <int NHM>
__global__ __launch_bounds__(64) void k_bs_pw_walk(bs_ctx C)
{
	constexpr int RPL = NHM / 64, CS = 66, CLS = NHM > 64 ? 1 : 0;       // CS: row of one region's 64 pointers, padded (bank conflicts of the transposed set-up)
	__shared__ uint16_t s_cnt[NHM * CS];                               // [region][block]: the block's next hole of the region, as a window address
	__shared__ uint8_t s_win[PW_WIN_CAP + 16];
	__shared__ uint32_t s_wb[NHM], s_woff[NHM + 1], s_jadj[NHM];      // first hole of the item's stretch, where the stretch starts in the window, window address -> hole index within the range
	__shared__ uint16_t s_wend[NHM];                                   // window address of the end of the region's holes (0xFFFF: beyond the item's stretch)
	const uint32_t lane = threadIdx.x;
	uint32_t n_items = C.hdr[16 + CLS];
	if (n_items > C.pw_item_cap) n_items = C.pw_item_cap;
	for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
		const bs_pw_item I = C.pw_items[CLS][it];
		const bs_meta &M = C.meta[I.r];
		const uint32_t nh = M.nh, stride = nh + 1u, nb = I.nb, beg = (uint32_t)C.rng[0][I.r].beg;
		const uint32_t *S = C.pw_snap + M.pw_off + (size_t)I.slot0 * stride;
		uint32_t run = 0;
#pragma unroll
		for (int t = 0; t < RPL; ++t) {
			const uint32_t q = lane + 64u * (uint32_t)t;
			uint32_t wb = 0, len = 0, en = 0;
			if (q < nh) { wb = S[q]; len = S[(size_t)nb * stride + q] - wb; en = beg + M.hst[M.act[q] + 1u]; }
			uint32_t inc = len;
			for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d); if (lane >= (uint32_t)d) inc += u; }
			const uint32_t tot = __shfl(inc, 63);
			if (q < nh) { const uint32_t wo = run + inc - len; s_wb[q] = wb; s_woff[q] = wo; s_jadj[q] = wb - beg - wo; s_wend[q] = en - wb <= len ? (uint16_t)(wo + (en - wb)) : (uint16_t)0xFFFFu; }
			run += tot;
		}
		if (lane == 0) s_woff[nh] = run;
		__syncthreads();
		if (run > (uint32_t)PW_WIN_CAP || nb == 0 || nb > 64u) { if (lane == 0) C.hdr[7] = 2; __syncthreads(); continue; }
		// the digits of the item's holes
#pragma unroll 4
		for (uint32_t q = 0; q < nh; ++q) {
			const uint32_t wb = s_wb[q], wo = s_woff[q], len = s_woff[q + 1u] - wo;
			for (uint32_t o = lane; o < len; o += 64u) s_win[wo + o] = C.hd[wb + o];
		}
		// every block's pointers at its start
#pragma unroll 4
		for (uint32_t b = 0; b < nb; ++b) {
#pragma unroll
			for (int t = 0; t < RPL; ++t) {
				const uint32_t q = lane + 64u * (uint32_t)t;
				if (q < nh) s_cnt[q * (uint32_t)CS + b] = (uint16_t)(s_woff[q] + (S[(size_t)b * stride + q] - s_wb[q]));
			}
		}
		__syncthreads();
		// a block: from its first base region, every base region's remaining cycles until the region the next block starts in, and that one's
		// up to the next block's pointer
		bool live = lane < nb;
		uint32_t k = 0, kend = 0, kfin = 0;
		if (live) {
			k = S[(size_t)lane * stride + nh]; kend = S[(size_t)(lane + 1u) * stride + nh];
			if (k >= nh || kend > nh || kend < k) live = false;
			else if (kend < nh) kfin = s_woff[kend] + (S[(size_t)(lane + 1u) * stride + kend] - s_wb[kend]);
		}
		uint32_t d = 0, i = 0, i0 = 0, steps = 0;
		bool inchase = false;
		while (__ballot(live)) {
			if (live) {
				if (!inchase && k >= kend && (k >= nh || s_cnt[k * (uint32_t)CS + lane] >= kfin)) live = false;
				else if (!inchase && k < kend && s_cnt[k * (uint32_t)CS + lane] >= s_wend[k]) ++k;   // this base region is complete: the next one
				else {
					const uint32_t q = inchase ? d : k;
					const uint32_t a = s_cnt[q * (uint32_t)CS + lane];
					if (a >= run || ++steps > (uint32_t)PW_WIN_CAP + 64u) live = false;   // (past the window: inconsistent snapshots - k_bs_pw_count reports those)
					else {
						s_cnt[q * (uint32_t)CS + lane] = (uint16_t)(a + 1u);
						const uint32_t dn = s_win[a], j = a + s_jadj[q];
						if (inchase) C.dest[beg + i] = j; else i0 = j;
						i = j;
						if (dn == k) { C.dest[beg + j] = i0; inchase = false; }
						else { d = dn < nh ? dn : k; inchase = true; }
					}
				}
			}
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ K8: placement
template <class REC>
__global__ __launch_bounds__(NT) void k_bs_scatter(bs_ctx C)
{
	__shared__ uint32_t s_r;
	__shared__ uint32_t s_start[257], s_hst[257], s_J[256], s_cw[BS_CW_WORDS];
	__shared__ uint8_t s_fate[256];
	const uint32_t n_rng = C.hdr[0], tid = threadIdx.x;
	// workgroup b runs on XCD b % 8: the tiles are dealt so that each XCD takes a contiguous eighth of them and a range's
	// records, hole lists and destinations meet in one L2
	const uint32_t per_xcd = gridDim.x >> 3, tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	if (tile >= C.hdr[1]) return;
	if (C.hdr[7]) return;                                         // a walk of this level gave up (the host reports it with the next header): dest[] is not to be trusted
	const uint32_t r = bs_find_range(C, tile, n_rng, &s_r);
	const bs_range R = C.rng[0][r];
	if (!R.exact) return;
	const bs_meta &M = C.meta[r];
	s_start[tid] = M.start[tid]; s_hst[tid] = M.hst[tid]; s_J[tid] = M.J[tid]; s_fate[tid] = M.fate[tid];
	if (tid == 0) { s_start[256] = M.start[256]; s_hst[256] = M.hst[256]; }
	__syncthreads();
	const uint32_t t0 = (tile - R.tile0) * BS_TILE, hbase = C.tile_h[tile];
	bs_cls q;
	bs_classify(C.dg + R.beg, s_start, t0, R.n, s_cw, q);
	const REC *src = reinterpret_cast<const REC*>(C.buf[R.buf]) + R.beg;
	REC *out_alt = reinterpret_cast<REC*>(C.buf[R.buf ^ 1]) + R.beg, *out_fin = reinterpret_cast<REC*>(C.dst) + R.beg;
	const uint32_t *hp = C.hp + R.beg, *dest = C.dest + R.beg;
	// records nobody reads once sorted (bs_range::dead): when their bucket - the first that is not empty - is final and holds just them, they stay where they are
	uint32_t dead_b = 256u;
	if (R.dead) { const uint32_t b0 = bs_region(s_start, 0u); if (s_fate[b0] == BS_FINAL && s_start[b0 + 1u] - s_start[b0] == R.dead) dead_b = b0; }
	// the buckets that are next-level ranges get their digits now, while the record is in a register: the byte they will be
	// split on is the highest one on which this range's keys differ below its own byte
	uint8_t *dgn = C.dg_next + R.beg;
	int ps = 0;
	{ const uint64_t low = M.s > 0 ? (M.k_or & ~M.k_and) & ((1ull << M.s) - 1ull) : 0ull; if (low) ps = (63 - __clzll(low)) & ~7; }
	// (round 6: the loads of all the thread's records - dest[], then hp[], and the records themselves - are issued in three batches before the first store; see k_bs_scatter_any)
	constexpr uint32_t NP_SKIP = 0xFFFFFFFFu;
	uint32_t jv[BS_TILE_IT], npv[BS_TILE_IT], look = 0;             // look: bit it = record it has a hole to look up
	REC recs[BS_TILE_IT];
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		jv[it] = 0; npv[it] = NP_SKIP;
		if (p >= R.n) continue;
		const uint32_t d = q.d[it], b = q.b[it], hb = hbase + q.hb[it];
		if (d == dead_b || s_fate[d] == BS_DROP) continue;
		recs[it] = src[p];
		if (d == b) npv[it] = p + ((hb - s_hst[b]) < s_J[b] ? 1u : 0u);
		else { jv[it] = dest[hb]; look |= 1u << it; }
	}
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		if (!((look >> it) & 1u)) continue;
		const uint32_t d = q.d[it], j = jv[it], jj = j - s_hst[d];
		if (j >= s_hst[256]) { C.hdr[7] = 2; continue; }               // not a hole of this range: a walk left dest[] unwritten -> "token walk made no progress", no access out of bounds
		if (jj < s_J[d]) npv[it] = jj == 0 ? s_start[d] : hp[j - 1] + 1u;
		else npv[it] = hp[j];
	}
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		if (npv[it] == NP_SKIP) continue;
		const uint32_t d = q.d[it], np = npv[it];
		const REC rec = recs[it];
		const uint32_t ft = s_fate[d];
		if (ft == BS_FINAL) out_fin[np] = rec; else out_alt[np] = rec;
		if (ft == BS_BIG) dgn[np] = (uint8_t)(rh_rec_ops<REC>::key(rec, C.rf) >> ps);
	}
}

// K8': placement in any order (jobs whose keys are almost never equal: the sorted order of a segment without ties is unique, so
// neither the holes nor the walk are needed): a tile counts its digits in LDS, reserves a stretch of every bucket it feeds with one
// atomic on the range's cursor and drops its records there.  Which tile gets which stretch is decided by the scheduler - harmless
// for distinct keys; segments that do hold equal keys are found where their buckets finish (bs_mark_tie) and redone with the exact passes.
template <class REC>
__global__ __launch_bounds__(NT) void k_bs_scatter_any(bs_ctx C)
{
	__shared__ uint32_t s_r;
	__shared__ uint32_t s_start[256], s_cnt[256], s_base[256];
	__shared__ uint8_t s_fate[256];
	const uint32_t n_rng = C.hdr[0], tid = threadIdx.x, tile = blockIdx.x;
	if (tile >= C.hdr[1]) return;
	const uint32_t r = bs_find_range(C, tile, n_rng, &s_r);
	const bs_range R = C.rng[0][r];
	if (R.exact) return;
	bs_meta &M = C.meta[r];
	s_start[tid] = M.start[tid]; s_fate[tid] = M.fate[tid]; s_cnt[tid] = 0;
	__syncthreads();
	const uint32_t t0 = (tile - R.tile0) * BS_TILE;
	const uint8_t *dg = C.dg + R.beg;
	uint32_t d[BS_TILE_IT], lr[BS_TILE_IT];
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		d[it] = 0; lr[it] = 0;
		if (p < R.n) { d[it] = dg[p]; lr[it] = atomicAdd(&s_cnt[d[it]], 1u); }
	}
	__syncthreads();
	if (s_cnt[tid]) s_base[tid] = atomicAdd(&M.inpl[tid], s_cnt[tid]);      // (inpl: zeroed by k_bs_clear, otherwise unused on this path)
	__syncthreads();
	const REC *src = reinterpret_cast<const REC*>(C.buf[R.buf]) + R.beg;
	REC *out_alt = reinterpret_cast<REC*>(C.buf[R.buf ^ 1]) + R.beg, *out_fin = reinterpret_cast<REC*>(C.dst) + R.beg;
	uint8_t *dgn = C.dg_next + R.beg;
	int ps = 0;
	{ const uint64_t low = M.s > 0 ? (M.k_or & ~M.k_and) & ((1ull << M.s) - 1ull) : 0ull; if (low) ps = (63 - __clzll(low)) & ~7; }
	// (round 6: every record of the thread requested before the first store - the stores may alias the loads for all the compiler knows, so a load-then-store loop
	// waits for memory once per record and a wavefront has one 512-byte request in flight; with 32 wavefronts a CU that is ~2 TB/s whatever HBM could deliver)
	REC recs[BS_TILE_IT];
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) { const uint32_t p = t0 + (uint32_t)it * NT + tid; if (p < R.n) recs[it] = src[p]; }
#pragma unroll
	for (int it = 0; it < BS_TILE_IT; ++it) {
		const uint32_t p = t0 + (uint32_t)it * NT + tid;
		if (p >= R.n) continue;
		const uint32_t np = s_start[d[it]] + s_base[d[it]] + lr[it], ft = s_fate[d[it]];
		const REC rec = recs[it];
		if (ft == BS_FINAL) out_fin[np] = rec; else out_alt[np] = rec;
		if (ft == BS_BIG) dgn[np] = (uint8_t)(rh_rec_ops<REC>::key(rec, C.rf) >> ps);
	}
}

// after the block sorter has finished an any-order job's small buckets: the segments of the buckets it found equal keys in
__global__ __launch_bounds__(NT) void k_bs_tie_map(bs_ctx C, int q, uint32_t ns)
{
	const uint32_t b = blockIdx.x * NT + threadIdx.x;
	if (b < ns && C.small_tie[q][b]) bs_mark_tie(C, C.small_off[q][b], C.small_cnt[q][b] ? C.small_cnt[q][b] : C.n_lo);   // (a count the block sorter has cleared: no bucket is longer than n_lo)
}
// ... and how many segments that makes: hdr[12]
__global__ __launch_bounds__(NT) void k_bs_tie_count(bs_ctx C)
{
	__shared__ uint32_t s_w[NT / 64];
	uint32_t run = 0;
	for (uint32_t a0 = 0; a0 < C.seg_n; a0 += NT) {
		const uint32_t a = a0 + threadIdx.x;
		uint32_t tot;
		(void)block_rank(a < C.seg_n && C.redo_skip[a] == 0, s_w, tot);
		run += tot;
	}
	if (threadIdx.x == 0) C.hdr[12] = run;
}

// K9: the next level's ranges get their tile numbers; they become "this level"
__global__ __launch_bounds__(NT) void k_bs_next(bs_ctx C)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint32_t tid = threadIdx.x;
	const uint32_t n = C.hdr[6] < C.rng_cap ? C.hdr[6] : C.rng_cap;
	uint32_t run = 0;
	for (uint32_t i0 = 0; i0 < n; i0 += NT) {
		const uint32_t i = i0 + tid;
		const uint32_t v = i < n ? (C.rng[1][i].n + BS_TILE - 1) / BS_TILE : 0u;
		uint32_t tot;
		const uint32_t ex = block_excl_scan(v, s_w, tot);
		if (i < n) C.rng[1][i].tile0 = run + ex;
		run += tot;
	}
	__syncthreads();
	if (tid == 0) { C.hdr[0] = n; C.hdr[1] = run; C.hdr[6] = 0; }
}

// ------------------------------------------------------------------------------------------------ host
#define BS_KINDS 8
static std::atomic<int> g_bs_guess[BS_KINDS] = { {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1} };   // per kind of job (rh_sort_job.kind): byte shift its first level split on last time
size_t rhk_bigsort_ws_bytes(uint64_t total, uint32_t n_lo)
{
	const uint64_t t = total ? total : 1, lo = n_lo ? n_lo : 1;
	const uint64_t rng_cap = t / (lo + 1) + 2, small_cap = t / 8 + 256, tiles = t / BS_TILE + rng_cap + 2;
	size_t b = 256;                                                // hdr
	b += ((t / 512 + 64) + 255) & ~(size_t)255;                     // tie_bits: one bit per 64 positions
	b += 2 * ((rng_cap * sizeof(bs_range) + 255) & ~(size_t)255);
	b += (rng_cap * sizeof(bs_meta) + 255) & ~(size_t)255;
	b += 2 * ((tiles * 4 + 255) & ~(size_t)255);
	b += 3 * ((t + 128 + 255) & ~(size_t)255);                      // dg (two: this level's and the next one's), hd
	b += 2 * ((t * 4 + 255) & ~(size_t)255);                       // hp, dest
	b += 4 * ((small_cap * 8 + 255) & ~(size_t)255) + 4 * ((small_cap * 4 + 255) & ~(size_t)255) + 4 * ((small_cap + 255) & ~(size_t)255);
	const uint64_t pw_items = t / 64 + 2 * rng_cap + 16;            // block-parallel walk: snapshots (4 bytes per record), work items
	b += ((t + 64) * 4 + 255) & ~(size_t)255;
	b += 2 * ((pw_items * sizeof(bs_pw_item) + 255) & ~(size_t)255);
	return b;
}

int rhk_bigsort(hipStream_t s, const rh_sort_job &jb, bool all_exact, uint32_t n_lo)
{
	if (!jb.n_seg || !jb.big_alt || !jb.big_ws || !jb.big_pin) { rh_set_error("segment sorter: no scratch for segments beyond the LDS classes"); return -1; }
	const uint64_t t = jb.big_total ? jb.big_total : 1, lo = n_lo ? n_lo : 1;
	bs_ctx C{};
	C.buf[0] = const_cast<rh_mm128_t*>(jb.src); C.buf[1] = jb.big_alt; C.dst = jb.dst; C.rf = jb.rf;
	C.rf.up = (jb.any_order && jb.rf.rec8) ? jb.any_up : 0;      // (only the levels of this file order by the packed key; the block sorter and the check take the keys as they are)
	const bool r8 = jb.rf.rec8 != 0;
	#define BS_LAUNCH_REC(kern, grid, ...) do { if (r8) RH_LAUNCH(kern<uint64_t>, grid, NT, 0, s, __VA_ARGS__); else RH_LAUNCH(kern<rh_mm128_t>, grid, NT, 0, s, __VA_ARGS__); } while (0)
	C.n_lo = n_lo;
	C.rng_cap = (uint32_t)(t / (lo + 1) + 2); C.small_cap = (uint32_t)(t / 8 + 256);
	const uint64_t tiles_cap = t / BS_TILE + C.rng_cap + 2;
	unsigned char *p = jb.big_ws;
	auto take = [&](size_t bytes) { unsigned char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
	C.hdr = (uint32_t*)take(256);
	unsigned long long *tie_bits = (unsigned long long*)take((size_t)(t / 512 + 64));   // (first, so that the exact re-run of an any-order job finds it where that job left it)
	C.any_order = jb.any_order ? 1 : 0; C.tie_path = !jb.any_order && jb.tie_path ? 1 : 0;
	C.tie_bits = (jb.any_order && jb.redo_skip) || C.tie_path ? tie_bits : nullptr;
	if (jb.any_order && C.tie_bits) RH_HIP(hipMemsetAsync(tie_bits, 0, (size_t)(t / 512 + 64), s));
	C.rng[0] = (bs_range*)take((size_t)C.rng_cap * sizeof(bs_range)); C.rng[1] = (bs_range*)take((size_t)C.rng_cap * sizeof(bs_range));
	C.meta = (bs_meta*)take((size_t)C.rng_cap * sizeof(bs_meta));
	C.tile_h = (uint32_t*)take(tiles_cap * 4); C.tile_rng = (uint32_t*)take(tiles_cap * 4);
	uint8_t *dgb[2] = { (uint8_t*)take(t + 128), (uint8_t*)take(t + 128) };
	C.hd = (uint8_t*)take(t + 128);   // (+64: the lane walkers read whole aligned words around a pointer)
	C.hp = (uint32_t*)take(t * 4); C.dest = (uint32_t*)take(t * 4);
	for (int q = 0; q < 4; ++q) { C.small_off[q] = (uint64_t*)take((size_t)C.small_cap * 8); C.small_cnt[q] = (uint32_t*)take((size_t)C.small_cap * 4); C.small_tie[q] = (uint8_t*)take((size_t)C.small_cap); }
	static const bool pw_on = !(getenv("RH_BS_PW") && atoi(getenv("RH_BS_PW")) == 0);   // RH_BS_PW=0: the serial token walkers for every range (A/B aid)
	{
		const uint64_t pww = t + 64;
		uint32_t *sn = (uint32_t*)take((size_t)pww * 4);
		C.pw_item_cap = (uint32_t)(t / 64 + 2 * (uint64_t)C.rng_cap + 16);
		for (int q = 0; q < 2; ++q) C.pw_items[q] = (bs_pw_item*)take((size_t)C.pw_item_cap * sizeof(bs_pw_item));
		C.pw_snap = pw_on && !jb.any_order && t < (1ull << 32) ? sn : nullptr; C.pw_words = (uint32_t)(pww < 0xFFFFFFFFull ? pww : 0xFFFFFFFFull);
	}
	C.seg_off = jb.off; C.seg_n = jb.n_seg; C.redo_skip = jb.any_order ? jb.redo_skip : nullptr;
	if ((size_t)(p - jb.big_ws) > jb.big_ws_bytes) { rh_set_error("segment sorter: scratch of %zu bytes is too small (%zu needed)", jb.big_ws_bytes, (size_t)(p - jb.big_ws)); return -1; }
	uint32_t *pin = (uint32_t*)jb.big_pin;
	// The exact re-run of an any-order job (tie_path) trusts the bits that job left in this scratch.  They are only good if THAT job - same segments, same scratch layout - was the last
	// to use the scratch: the any-order job signs the scratch when it is done (hdr[48..53]), every other job wipes the signature when it starts, and a re-run that does not find the
	// signature it expects takes the exact passes for every range instead (correct, slower) - never somebody else's bits.
	const uint32_t sig[6] = { 0x54494531u, jb.n_seg, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)jb.kind, (uint32_t)(uintptr_t)jb.off };
	if (C.tie_path) {
		RH_HIP(hipMemcpyAsync(pin + 24, C.hdr + 48, sizeof(sig), hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		if (memcmp(pin + 24, sig, sizeof(sig)) != 0) { C.tie_path = 0; C.tie_bits = nullptr; }
	}
	RH_HIP(hipMemsetAsync(C.hdr + 48, 0, sizeof(sig), s));
	RH_LAUNCH(k_bs_init, 1, NT, 0, s, jb, C);
	static const bool trace = RH_DEVENV("RH_BS_TRACE") != nullptr;   // development aid: per-level launch shapes and times on stderr
	static const bool walk_old = RH_DEVENV("RH_BS_WALK_OLD") != nullptr;   // development aid: the LDS-resident walkers instead of the scalar-token one
	static const uint32_t tok_max = RH_DEVENV("RH_BS_TOK_MAX") ? (uint32_t)strtoul(RH_DEVENV("RH_BS_TOK_MAX"), nullptr, 10) : 0xFFFFFFFFu;
	static const bool tok2 = !(RH_DEVENV("RH_BS_TOK2") && atoi(RH_DEVENV("RH_BS_TOK2")) == 0);   // two regions per lane for ranges with 65 .. 128 regions that have holes (the candidate sort's first level; measured +2 % on one stream)
	static const bool tok_adv = !(getenv("RH_BS_TOK_ADV") && atoi(getenv("RH_BS_TOK_ADV")) == 0);   // RH_BS_TOK_ADV=0: development aid, the compiler-scheduled pop
	static const bool tok4 = RH_DEVENV("RH_BS_TOK4") != nullptr;       // development aid: four regions per lane for ranges with more than 64 regions that have holes
	// the byte the first level of this kind of job split on last time (-1: not known yet; RH_BS_NO_GUESS: never used)
	static const bool no_guess = RH_DEVENV("RH_BS_NO_GUESS") != nullptr;   // development aid
	const uint32_t kind = jb.kind < BS_KINDS ? jb.kind : 0u;
	const int gs = no_guess ? -1 : g_bs_guess[kind].load(std::memory_order_relaxed);
	uint32_t n_rng0 = 0;
	// Walk wavefronts live for milliseconds and there are more of them than wave slots: left alone they end up holding every slot of the
	// chip and the other streams' bandwidth-bound kernels wait behind an issue-bound one.  Unused dynamic LDS caps them per CU.
	static const uint32_t walk_lds = RH_DEVENV("RH_BS_WALK_LDS") ? (uint32_t)strtoul(RH_DEVENV("RH_BS_WALK_LDS"), nullptr, 10) : 0u;
	static const int walk_reps = RH_DEVENV("RH_BS_WALK_REPS") ? atoi(RH_DEVENV("RH_BS_WALK_REPS")) : 1, scat_reps = RH_DEVENV("RH_BS_SCAT_REPS") ? atoi(RH_DEVENV("RH_BS_SCAT_REPS")) : 1;   // development aid: the (idempotent) walks / placement launched several times - what a pass costs the step with the other streams' kernels around it
	hipEvent_t ev[5] = {};
	if (trace) for (auto &e : ev) (void)hipEventCreate(&e);
	for (int level = 0; level < 9; ++level) {
		RH_HIP(hipMemcpyAsync(pin, C.hdr, 64, hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		const uint32_t n_rng = pin[0], n_tiles = pin[1];
		if (level == 1 && pin[14] && (gs < 0 || 2u * pin[13] > n_rng0)) g_bs_guess[kind].store((int)pin[14] - 1, std::memory_order_relaxed);   // most ranges split elsewhere (or nothing was known): the next job of this kind starts from what this one found
		if (pin[7]) { rh_set_error(pin[7] == 2 ? "segment sorter: a token walk made no progress" : "segment sorter: range / segment list overflow"); return -1; }
		if (n_rng == 0) break;
		if (trace) (void)hipEventRecord(ev[0], s);
		C.dg = dgb[level & 1]; C.dg_next = dgb[(level & 1) ^ 1];
		RH_LAUNCH(k_bs_tile_map, (n_tiles + NT - 1) / NT, NT, 0, s, C);
		RH_LAUNCH(k_bs_clear, n_rng, NT, 0, s, C);
		if (level == 0 && gs >= 0) BS_LAUNCH_REC(k_bs_hist0, n_tiles, C, gs);   // OR / AND and the histogram at the remembered byte in one read of the keys
		else {
			BS_LAUNCH_REC(k_bs_diff, n_tiles, C);
			BS_LAUNCH_REC(k_bs_hist, n_tiles, C);
		}
		if (level) BS_LAUNCH_REC(k_bs_fix, n_rng, C);
		else { BS_LAUNCH_REC(k_bs_fix0, n_rng, C, gs); n_rng0 = n_rng; }
		RH_LAUNCH(k_bs_plan, n_rng, NT, 0, s, C);
		if (jb.any_order) BS_LAUNCH_REC(k_bs_scatter_any, n_tiles, C);   // no holes, no walk: tiles reserve stretches of their buckets (an exact re-run has no such ranges any more: its buckets off the way to the equal keys are dropped, k_bs_plan)
		if (jb.any_order) {
			RH_LAUNCH(k_bs_next, 1, NT, 0, s, C);
			bs_range *tmp2 = C.rng[0]; C.rng[0] = C.rng[1]; C.rng[1] = tmp2;
			continue;
		}
		RH_LAUNCH(k_bs_count, n_tiles, NT, 0, s, C);
		RH_LAUNCH(k_bs_scan, n_rng, NT, 0, s, C);
		RH_LAUNCH(k_bs_holes, n_tiles, NT, 0, s, C);
		// few ranges: a wavefront each (nothing to gain from 64 walks per wavefront); many: one lane each where the regions fit
		const int lanes = n_rng >= (uint32_t)BS_LANES_MIN_RANGES, multi = !lanes && n_rng >= (uint32_t)BS_MULTI_MIN_RANGES && t < (1ull << 32);   // (k_bs_walk_multi keeps absolute hole addresses in 32 bits)
		const int tok = !lanes && !walk_old && t < (1ull << 32) && n_rng <= tok_max;   // scalar token, one lane per region (absolute hole addresses in 32 bits): levels too narrow for the walks to fill the chip
		if (trace) (void)hipEventRecord(ev[1], s);
		for (int rep = 0; rep < walk_reps; ++rep)
		if (tok) {
			RH_LAUNCH(k_bs_walk_wave, n_rng, 64, walk_lds, s, C, 3u, tok4 ? 256u : tok2 ? 128u : 64u);   // two regions with holes: closed form (and, measured faster there, more than 128: one LDS-resident walker per wavefront)
			if (tok_adv) RH_LAUNCH((k_bs_walk_tok<1, 1>), n_rng, 64, walk_lds, s, C, 3u, 64u);   // (fewer scalar instructions a pop: rh_tok_advance, one period counter)
			else RH_LAUNCH((k_bs_walk_tok<1, 0>), n_rng, 64, walk_lds, s, C, 3u, 64u);
			if (tok2) RH_LAUNCH((k_bs_walk_tok<2>), n_rng, 64, walk_lds, s, C, 65u, 128u);
			if (tok4) RH_LAUNCH((k_bs_walk_tok<4>), n_rng, 64, walk_lds, s, C, tok2 ? 129u : 65u, 256u);
		} else {
		RH_LAUNCH(k_bs_walk_wave, n_rng, 64, 0, s, C, lanes ? 3u : multi ? 3u : 1u, lanes ? 256u : multi ? (uint32_t)BS_MW_NHM : 0u);
		if (multi) {	// walks per wavefront: so that the level takes about one wavefront per SIMD (a walk's step time does not depend on how many lanes walk)
			if (n_rng > (uint32_t)BS_MW_G32_FROM) RH_LAUNCH((k_bs_walk_multi<BS_MW_NHM, 32>), (n_rng + 31) / 32, 64, 0, s, C);
			else if (n_rng > (uint32_t)BS_MW_G16_FROM) RH_LAUNCH((k_bs_walk_multi<BS_MW_NHM, 16>), (n_rng + 15) / 16, 64, 0, s, C);
			else RH_LAUNCH((k_bs_walk_multi<BS_MW_NHM, 8>), (n_rng + 7) / 8, 64, 0, s, C);
		}
		if (lanes) {	// by number of regions with holes: 64 / 32 / 8 walkers per wavefront
			RH_LAUNCH((k_bs_walk_lanes<24, 64>), (n_rng + 63) / 64, 64, 0, s, C, 2u);
			RH_LAUNCH((k_bs_walk_lanes<64, 32>), (n_rng + 31) / 32, 64, 0, s, C, 24u);
			RH_LAUNCH((k_bs_walk_lanes<256, 8>), (n_rng + 7) / 8, 64, 0, s, C, 64u);   // LDS: 24 B per region and walker (36 / 48 / 48 KB)
		}
		}
		if (trace) (void)hipEventRecord(ev[4], s);
		if (C.pw_snap) {	// the ranges k_bs_scan gave snapshot slots: pointers after every block of cycles, then all blocks walked at once
			RH_LAUNCH((k_bs_pw_count<64>), n_rng, 64, 0, s, C);
			RH_LAUNCH((k_bs_pw_count<256>), n_rng, 64, 0, s, C);
			const uint32_t gb = C.pw_item_cap < 16384u ? C.pw_item_cap : 16384u;
			RH_LAUNCH((k_bs_pw_walk<64>), gb, 64, 0, s, C);
			RH_LAUNCH((k_bs_pw_walk<256>), gb, 64, 0, s, C);
			if (RH_DEVENV("RH_BS_PW_CHECK")) {	// development aid: every hole of every range has its dest[]
				(void)hipStreamSynchronize(s);
				std::vector<bs_meta> mv(n_rng); std::vector<bs_range> rv(n_rng);
				(void)hipMemcpy(mv.data(), C.meta, (size_t)n_rng * sizeof(bs_meta), hipMemcpyDeviceToHost); (void)hipMemcpy(rv.data(), C.rng[0], (size_t)n_rng * sizeof(bs_range), hipMemcpyDeviceToHost);
				uint32_t pwh[8]; (void)hipMemcpy(pwh, C.hdr + 16, 32, hipMemcpyDeviceToHost);
				fprintf(stderr, "PW level %d: items %u + %u, words %u; blocks by one lane %u, ranges out of slots %u, ring misses %u, cycles by one lane %u\n", level, pwh[0], pwh[1], pwh[2], pwh[4], pwh[5], pwh[6], pwh[7]);
				for (uint32_t r = 0; r < n_rng; ++r) {
					if (!mv[r].pw) continue;
					const uint32_t nhl = mv[r].hst[256];
					std::vector<uint32_t> dv(nhl); std::vector<uint8_t> seen(nhl, 0);
					(void)hipMemcpy(dv.data(), C.dest + rv[r].beg, (size_t)nhl * 4, hipMemcpyDeviceToHost);
					uint64_t bad = 0, dup = 0;
					for (uint32_t i = 0; i < nhl; ++i) { if (dv[i] >= nhl) ++bad; else if (seen[dv[i]]++) ++dup; }
					fprintf(stderr, "PW range %u: n %u holes %u nh %u pw %u slots %u: dest out of range %llu, duplicate targets %llu\n", r, rv[r].n, nhl, mv[r].nh, mv[r].pw, mv[r].pw_slots, (unsigned long long)bad, (unsigned long long)dup);
					// the walk itself, on the host (ksort.h:124-138 on the hole lists)
					std::vector<uint8_t> hdv(nhl); (void)hipMemcpy(hdv.data(), C.hd + rv[r].beg, nhl, hipMemcpyDeviceToHost);
					const uint32_t nhr = mv[r].nh; std::vector<uint32_t> pt(nhr), en(nhr), want(nhl, 0xFFFFFFFFu);
					for (uint32_t q = 0; q < nhr; ++q) { pt[q] = mv[r].hst[mv[r].act[q]]; en[q] = mv[r].hst[mv[r].act[q] + 1u]; }
					for (uint32_t k = 0; k < nhr; ++k) while (pt[k] < en[k]) { const uint32_t i0 = pt[k]++; uint32_t i = i0, d = hdv[i0]; while (d != k) { const uint32_t j = pt[d]++; want[i] = j; i = j; d = hdv[j]; } want[i] = i0; }
					uint32_t nbad = 0;
					for (uint32_t i = 0; i < nhl; ++i) if (dv[i] != want[i]) { if (nbad++ < 6) { uint32_t q = 0; while (q + 1 < nhr && mv[r].hst[mv[r].act[q + 1]] <= i) ++q; fprintf(stderr, "   hole %u (region %u, its hole %u): dest %u, the walk says %u\n", i, q, i - mv[r].hst[mv[r].act[q]], dv[i], want[i]); } }
					if (nbad) { fprintf(stderr, "   %u holes differ; snapshots:\n", nbad);
						std::vector<uint32_t> sv((size_t)mv[r].pw_slots * (nhr + 1)); (void)hipMemcpy(sv.data(), C.pw_snap + mv[r].pw_off, sv.size() * 4, hipMemcpyDeviceToHost);
						for (uint32_t sl = 0; sl < mv[r].pw_slots && sl < 12; ++sl) { fprintf(stderr, "   slot %u k %u:", sl, sv[(size_t)sl * (nhr + 1) + nhr]); for (uint32_t q = 0; q < nhr && q < 6; ++q) fprintf(stderr, " %u", sv[(size_t)sl * (nhr + 1) + q] - (uint32_t)rv[r].beg); fprintf(stderr, "\n"); }
						for (int cl = 0; cl < 2; ++cl) { std::vector<bs_pw_item> iv(pwh[cl]); (void)hipMemcpy(iv.data(), C.pw_items[cl], iv.size() * sizeof(bs_pw_item), hipMemcpyDeviceToHost); for (auto &I : iv) if (I.r == r) fprintf(stderr, "   item: slot0 %u nb %u\n", I.slot0, I.nb); } }
				}
			}
		}
		if (trace) (void)hipEventRecord(ev[2], s);
		for (int rep = 0; rep < scat_reps; ++rep)
		BS_LAUNCH_REC(k_bs_scatter, ((n_tiles + 7) / 8) * 8, C);
		RH_LAUNCH(k_bs_next, 1, NT, 0, s, C);
		if (trace) {
			(void)hipEventRecord(ev[3], s); (void)hipEventSynchronize(ev[3]);
			float a = 0, b = 0, c = 0;
			(void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]); (void)hipEventElapsedTime(&c, ev[2], ev[3]);
			float bpw = 0; (void)hipEventElapsedTime(&bpw, ev[4], ev[2]);
			// records by the number of regions with holes of their range: <= 2 (closed form), 3..24, 25..64, 65..128, more
			std::vector<uint32_t> nhv(n_rng); std::vector<bs_range> rv(n_rng);
			uint64_t pw_rng = 0, pw_holes = 0, all_holes = 0; uint32_t pwh[12] = {};
			uint64_t why[6][2] = {}; uint32_t max_h = 0, max_h_pw = 0;   // ranges / holes left to the serial walkers, by reason: <= 2 regions, few holes, few cycles of the first region, long cycles, more than 256 regions / no slots
			{ std::vector<bs_meta> mv(n_rng); (void)hipMemcpy(mv.data(), C.meta, (size_t)n_rng * sizeof(bs_meta), hipMemcpyDeviceToHost); for (uint32_t r = 0; r < n_rng; ++r) { nhv[r] = mv[r].nh; const uint32_t P = mv[r].hst[256]; all_holes += P; if (mv[r].pw) { ++pw_rng; pw_holes += P; if (P > max_h_pw) max_h_pw = P; } else {
				const uint32_t c0 = mv[r].nh ? mv[r].hst[mv[r].act[0] + 1u] - mv[r].hst[mv[r].act[0]] : 0u;
				const int w = mv[r].nh <= 2 ? 0 : P < (uint32_t)PW_MIN_HOLES ? 1 : c0 < (uint32_t)PW_MIN_C0 ? 2 : (uint64_t)c0 * PW_MAX_CYCLE < P ? 3 : 4;
				++why[w][0]; why[w][1] += P; if (w && P > max_h) max_h = P; } } }
			(void)hipMemcpy(pwh, C.hdr + 16, 48, hipMemcpyDeviceToHost);
			fprintf(stderr, "BS level %d block-parallel walk %.3f ms [blocks by one lane %u, out of slots %u, window misses %u, cycles by one lane %u, window reloads %u]: %llu of %u ranges, %llu of %llu holes (largest %u), items %u + %u, snapshot words %llu of %u; serial: <=2 regions %llu/%llu, few holes %llu/%llu, few cycles %llu/%llu, long cycles %llu/%llu, no slots %llu/%llu, largest %u\n", level, bpw, pwh[4], pwh[5], pwh[6], pwh[7], pwh[8], (unsigned long long)pw_rng, n_rng,
			        (unsigned long long)pw_holes, (unsigned long long)all_holes, max_h_pw, pwh[0], pwh[1], (unsigned long long)pwh[2] | (unsigned long long)pwh[3] << 32, C.pw_words,
			        (unsigned long long)why[0][0], (unsigned long long)why[0][1], (unsigned long long)why[1][0], (unsigned long long)why[1][1], (unsigned long long)why[2][0], (unsigned long long)why[2][1], (unsigned long long)why[3][0], (unsigned long long)why[3][1], (unsigned long long)why[4][0], (unsigned long long)why[4][1], max_h);
			(void)hipMemcpy(rv.data(), C.rng[0], (size_t)n_rng * sizeof(bs_range), hipMemcpyDeviceToHost);
			uint64_t bins[5] = {0, 0, 0, 0, 0};
			for (uint32_t r = 0; r < n_rng; ++r) bins[nhv[r] <= 2 ? 0 : nhv[r] <= 24 ? 1 : nhv[r] <= 64 ? 2 : nhv[r] <= 128 ? 3 : 4] += rv[r].n;
			fprintf(stderr, "BS level %d segs %u total %llu rng %u tiles %u pre %.3f walk %.3f post %.3f  records by regions: <=2 %llu, <=24 %llu, <=64 %llu, <=128 %llu, more %llu\n", level, jb.n_seg,
			        (unsigned long long)t, n_rng, n_tiles, a, b, c, (unsigned long long)bins[0], (unsigned long long)bins[1], (unsigned long long)bins[2], (unsigned long long)bins[3], (unsigned long long)bins[4]);
		}
		bs_range *tmp = C.rng[0]; C.rng[0] = C.rng[1]; C.rng[1] = tmp;
	}
	if (trace) for (auto &e : ev) (void)hipEventDestroy(e);
	// the buckets that fit the LDS classes finish in the block sorter, from the copy that holds them
	const bool job32 = jb.kc_on && (uint32_t)jb.kc_lo + jb.kc_mid + jb.kc_hi <= 32u && jb.kc_mid <= 24u;
	for (int q = 0; q < 4; ++q) {
		const uint32_t ns = pin[2 + q];
		if (!ns) continue;
		rh_sort_job sj = jb;
		sj.n_seg = ns; sj.skip = nullptr; sj.off = C.small_off[q]; sj.cnt = C.small_cnt[q];
		sj.src = (const rh_mm128_t*)C.buf[q >> 1]; sj.dst = jb.dst; sj.need_exact = C.redo_skip ? C.small_tie[q] : nullptr; sj.n_max = pin[8 + q] < n_lo ? pin[8 + q] : n_lo;   // (an LDS class above the list's largest bucket would be a launch of blocks that all leave at once, each waiting for its LDS)
		if ((q & 1) && !job32) { sj.kc_on = 1; sj.kc_lo = 32; sj.kc_mid = 0; sj.kc_hi = 0; }   // keys that differ below bit 32 only: 32-bit words in LDS
		sj.big_alt = nullptr; sj.big_ws = nullptr; sj.dead_cnt = nullptr; sj.rf.up = 0;       // (the block sorter takes its keys at their original positions)
		sj.any_order = 0; sj.redo_skip = nullptr; sj.n_redo = nullptr; sj.cnt_rw = C.small_cnt[q];
		sj.no_redo = C.redo_skip ? 1 : 0;                              // (an any-order job: the segments whose buckets hold equal keys are redone from their input, whatever order these buckets are left in)
		if (C.redo_skip) RH_HIP(hipMemsetAsync(C.small_tie[q], 0, ns, s));
		if (rhk_sort_job(s, sj, all_exact, 1u)) return -1;
		if (C.redo_skip) RH_LAUNCH(k_bs_tie_map, (ns + NT - 1) / NT, NT, 0, s, C, q, ns);
	}
	if (jb.any_order) {
		RH_LAUNCH(k_bs_tie_count, 1, NT, 0, s, C);
		RH_HIP(hipMemcpyAsync(pin, C.hdr + 12, 4, hipMemcpyDeviceToHost, s));
		if (C.tie_bits) { memcpy(pin + 24, sig, sizeof(sig)); RH_HIP(hipMemcpyAsync(C.hdr + 48, pin + 24, sizeof(sig), hipMemcpyHostToDevice, s)); }   // signed: see above
		RH_HIP(hipStreamSynchronize(s));
		if (jb.n_redo) *jb.n_redo = pin[0];
	}
	#undef BS_LAUNCH_REC
	return 0;
}
