// Index construction on the GPU (ri_idx_gen rindex.c:900, worker_pipeline :100, ri_seq_to_sig rsig.c:13,
// ri_sketch_reg rsketch.c:143, worker_post rindex.c:311), producing the HBM-resident table of rh_index_upload directly.
// A human-sized reference (3.1 Gbp, 6.2 G seeds) is indexed in seconds instead of minutes on the host.
//
// Per target sequence (both strands), all data parallel:
//   k_ix_levels   expected signal: one normalised model level per k-mer, ambiguous bases repeat the previous k-mer
//                 (= the k most recent valid bases: host-found runs of invalid bases make that a closed form)
//   k_ix_keep     the greedy |ev[f] - ev[last kept]| >= diff filter of the sketch (rsketch.c:160-163).  It is a serial
//                 chain over the whole chromosome, but its state is one float (the last kept level) and chains started
//                 anywhere merge as soon as both keep the same event: every block of IX_L events is run by one lane
//                 from a speculative start IX_WU events earlier; k_ix_verify compares each block's assumed entering
//                 state with its predecessor's true exit state and the (rare) wrong blocks are re-run until none is.
//   k_ix_compact  quantised codes + positions of the kept events (prefix popcounts of the keep mask)
//   k_ix_seeds    hash of every e consecutive kept codes; the seed's place among the sequence's seeds ordered by
//                 (position, strand) - the order worker_post leaves a key's positions in - is its own rank plus the other
//                 strand's prefix popcount: no merge, no comparison sort
// Then one stable LSD radix sort of (hash, position word) by the 32-bit hash (4 passes of 8 bits, 64-bit indices),
// key boundaries -> (hash, first, count), and the bucketed open-addressing table is filled with atomicCAS.
#include "rh_kernels.h"
#include "rh_devutil.h"
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#ifndef IX_L
#define IX_L 2048                         // events per lane of the greedy filter (multiple of 32)
#endif
#ifndef IX_WU
#define IX_WU 256                         // speculative warm-up before a block
#endif
#define IX_RT 4096                        // records per tile of the radix sort (NT threads x 16)
#define IX_RT_IT (IX_RT / NT)

namespace {

struct DevMem {
	void *p = nullptr;
	~DevMem() { if (p) (void)hipFree(p); }
	int alloc(size_t bytes) { if (p) { (void)hipFree(p); p = nullptr; } hipError_t e = hipMalloc(&p, bytes ? bytes : 1); if (e != hipSuccess) { p = nullptr; rh_set_error("index build: hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e)); return -1; } return 0; }
	void release() { if (p) (void)hipFree(p); p = nullptr; }
	template <class T> T *as() const { return (T*)p; }
};

struct ix_runs { const uint32_t *rs, *re, *cum; uint32_t n, n_valid; };   // runs of invalid bases (forward coordinates), cum[j] = invalid bases before run j, cum[n] = all

} // namespace

RH_DEV int ix_nt4(unsigned char c)
{
	switch (c) {
		case 'A': case 'a': return 0;
		case 'C': case 'c': return 1;
		case 'G': case 'g': return 2;
		case 'T': case 't': case 'U': case 'u': return 3;
		default: return 4;
	}
}

RH_DEV uint32_t ix_invalid_lt(const ix_runs &R, uint32_t p)
{
	uint32_t lo = 0, hi = R.n;                                     // r = runs starting before p
	while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (R.rs[mid] < p) lo = mid + 1; else hi = mid; }
	if (lo == 0) return 0;
	const uint32_t e = R.re[lo - 1] < p ? R.re[lo - 1] : p;
	return R.cum[lo - 1] + (e - R.rs[lo - 1]);
}
RH_DEV uint32_t ix_pos_of_valid(const ix_runs &R, uint32_t j)
{
	uint32_t lo = 0, hi = R.n;                                     // r = runs with (valid bases before the run) <= j
	while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (R.rs[mid] - R.cum[mid] <= j) lo = mid + 1; else hi = mid; }
	return j + R.cum[lo];
}

// ------------------------------------------------------------------------------------------------ levels
__global__ __launch_bounds__(NT) void k_ix_levels(const char *seq, uint32_t len, int strand, int k, const float *model, ix_runs R, float *lv, uint32_t n_ev)
{
	const uint32_t e = blockIdx.x * NT + threadIdx.x;
	if (e >= n_ev) return;
	const uint32_t i = e + (uint32_t)k - 1u;                       // strand-order index of the k-mer's last base
	uint32_t kmer = 0;
	if (R.n == 0) {
		for (int t = 0; t < k; ++t) {
			const uint32_t si = i - (uint32_t)k + 1u + (uint32_t)t, p = strand ? len - 1u - si : si;
			const int c = ix_nt4((unsigned char)seq[p]);
			kmer = kmer << 2 | (uint32_t)(strand ? 3 - c : c);
		}
	} else if (!strand) {
		const uint32_t c = (i + 1u) - ix_invalid_lt(R, i + 1u);     // valid bases at positions <= i
		for (int t = 0; t < k; ++t) {
			const int64_t vj = (int64_t)c - k + t;
			uint32_t b = 0;
			if (vj >= 0) b = (uint32_t)ix_nt4((unsigned char)seq[ix_pos_of_valid(R, (uint32_t)vj)]);
			kmer = kmer << 2 | b;
		}
	} else {
		const uint32_t p = len - 1u - i;
		const uint32_t c = R.n_valid - (p - ix_invalid_lt(R, p));   // valid bases at strand positions <= i (= forward positions >= p)
		for (int t = 0; t < k; ++t) {
			const int64_t sj = (int64_t)c - k + t;
			uint32_t b = 0;
			if (sj >= 0) b = 3u - (uint32_t)ix_nt4((unsigned char)seq[ix_pos_of_valid(R, R.n_valid - 1u - (uint32_t)sj)]);
			kmer = kmer << 2 | b;
		}
	}
	lv[e] = model[kmer];
}

// ------------------------------------------------------------------------------------------------ greedy filter
// lane b owns events [b * IX_L, (b + 1) * IX_L); flag == null: speculative run of every block; else exact re-run of the flagged ones
__global__ __launch_bounds__(64) void k_ix_keep(const float *lv, uint32_t n_ev, float diff, float *in_state, float *out_state, uint32_t *mask, const uint8_t *flag, uint32_t n_blocks)
{
	const uint32_t b = blockIdx.x * 64 + threadIdx.x;
	if (b >= n_blocks || (flag && !flag[b])) return;
	const uint32_t start = b * (uint32_t)IX_L, end = start + IX_L < n_ev ? start + IX_L : n_ev;
	float last;
	uint32_t f = start;
	uint32_t word = 0;
	if (flag) last = in_state[b];
	else if (start <= (uint32_t)IX_WU) {	// exact from the first event of the sequence (always kept)
		last = lv[0];
		for (uint32_t g = 1; g < start; ++g) { const float d = lv[g] - last; if (!((d < 0 ? -d : d) < diff)) last = lv[g]; }
		if (start == 0) { word = 1u; f = 1; }
		in_state[b] = last;
	} else {
		const uint32_t s = start - IX_WU;
		last = lv[s];
		for (uint32_t g = s + 1; g < start; ++g) { const float d = lv[g] - last; if (!((d < 0 ? -d : d) < diff)) last = lv[g]; }
		in_state[b] = last;
	}
	for (; f < end; ++f) {
		const float v = lv[f], d = v - last;
		const bool kept = !((d < 0 ? -d : d) < diff);
		if (kept) { last = v; word |= 1u << (f & 31u); }
		if ((f & 31u) == 31u) { mask[f >> 5] = word; word = 0; }
	}
	if (end & 31u) mask[end >> 5] = word;                           // (only the sequence's last block ends inside a word)
	out_state[b] = last;
}

__global__ __launch_bounds__(NT) void k_ix_verify(float *in_state, const float *out_state, uint8_t *flag, uint32_t n_blocks, uint32_t *n_bad)
{
	const uint32_t b = blockIdx.x * NT + threadIdx.x;
	if (b >= n_blocks) return;
	bool bad = false;
	if (b > 0 && b * (uint32_t)IX_L > (uint32_t)IX_WU) {            // (blocks within the first warm-up ran exactly)
		const float want = out_state[b - 1];
		bad = __float_as_uint(want) != __float_as_uint(in_state[b]);
		if (bad) in_state[b] = want;
	}
	flag[b] = bad ? 1 : 0;
	if (bad) atomicAdd(n_bad, 1u);
}

// ------------------------------------------------------------------------------------------------ scans
// out[i] = sum of in[0 .. i) as OUT_T; tile sums through `sums` (n / 2048 + 1 entries of uint64), total to *total
template <class OUT_T>
__global__ __launch_bounds__(NT) void k_ix_scan_tiles(const uint32_t *in, uint64_t n, OUT_T *out, uint64_t *sums)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint64_t base = (uint64_t)blockIdx.x * 2048u;
	uint32_t run = 0;
	for (uint32_t it = 0; it < 8; ++it) {
		const uint64_t i = base + (uint64_t)it * NT + threadIdx.x;
		const uint32_t v = i < n ? in[i] : 0u;
		uint32_t tot;
		const uint32_t ex = block_excl_scan(v, s_w, tot);
		if (i < n) out[i] = (OUT_T)(run + ex);
		run += tot;
	}
	if (threadIdx.x == 0) sums[blockIdx.x] = run;
}
__global__ __launch_bounds__(NT) void k_ix_scan_sums(uint64_t *sums, uint64_t n_tiles, uint64_t *total)
{
	__shared__ uint64_t s_part[NT];
	const uint32_t tid = threadIdx.x;
	const uint64_t per = (n_tiles + NT - 1) / NT, b = tid * per, e = b + per < n_tiles ? b + per : n_tiles;
	uint64_t s = 0;
	for (uint64_t i = b; i < e; ++i) s += sums[i];
	s_part[tid] = s;
	__syncthreads();
	if (tid == 0) { uint64_t run = 0; for (uint32_t i = 0; i < NT; ++i) { const uint64_t v = s_part[i]; s_part[i] = run; run += v; } *total = run; }
	__syncthreads();
	uint64_t run = s_part[tid];
	for (uint64_t i = b; i < e; ++i) { const uint64_t v = sums[i]; sums[i] = run; run += v; }
}
template <class OUT_T>
__global__ __launch_bounds__(NT) void k_ix_scan_add(OUT_T *out, uint64_t n, const uint64_t *sums)
{
	const uint64_t base = (uint64_t)blockIdx.x * 2048u;
	const OUT_T add = (OUT_T)sums[blockIdx.x];
	for (uint32_t it = 0; it < 8; ++it) { const uint64_t i = base + (uint64_t)it * NT + threadIdx.x; if (i < n) out[i] += add; }
}

__global__ __launch_bounds__(NT) void k_ix_popc(const uint32_t *mask, uint32_t n_words, uint32_t *pc)
{
	const uint32_t w = blockIdx.x * NT + threadIdx.x;
	if (w < n_words) pc[w] = (uint32_t)__popcll((unsigned long long)mask[w]);
}

// ------------------------------------------------------------------------------------------------ kept events, seeds
__global__ __launch_bounds__(NT) void k_ix_compact(const float *lv, const uint32_t *mask, const uint32_t *prefix, uint32_t n_words, rh_sketch_par sp, uint8_t *kcode, uint32_t *kpos)
{
	const uint32_t w = blockIdx.x * NT + threadIdx.x;
	if (w >= n_words) return;
	uint32_t bits = mask[w], t = prefix[w];
	const uint32_t n_buckets = 1u << sp.q;
	while (bits) {
		const uint32_t bit = (uint32_t)__builtin_ctz(bits);
		bits &= bits - 1;
		const uint32_t f = w * 32u + bit;
		kcode[t] = (uint8_t)(rh_quantise(lv[f], sp.fine_min, sp.fine_max, sp.fine_range, n_buckets) & (n_buckets - 1u));
		kpos[t] = f;
		++t;
	}
}

RH_DEV uint32_t ix_kept_below(const uint32_t *mask, const uint32_t *prefix, uint32_t n_ev, uint32_t n_kept, uint32_t p)
{
	if (p >= n_ev) return n_kept;
	const uint32_t w = p >> 5;
	return prefix[w] + (uint32_t)__popcll((unsigned long long)(mask[w] & ((1u << (p & 31u)) - 1u)));
}

// seeds of one strand; o_* = the other strand's keep mask (0 seeds there: o_seed = 0)
__global__ __launch_bounds__(NT) void k_ix_seeds(const uint8_t *kcode, const uint32_t *kpos, uint32_t n_seed, int strand, uint32_t id, rh_sketch_par sp,
                                                 const uint32_t *o_mask, const uint32_t *o_prefix, uint32_t o_kept, uint32_t o_seed, uint32_t n_ev,
                                                 uint32_t *H, uint64_t *Y, uint64_t base)
{
	const uint32_t t = blockIdx.x * NT + threadIdx.x;
	if (t >= n_seed) return;
	const uint32_t qb = (uint32_t)sp.q;
	const uint64_t mask_events = (qb * sp.e >= 64) ? ~0ULL : ((1ULL << (qb * sp.e)) - 1);
	uint64_t qv = 0;
	for (int j = 0; j < sp.e; ++j) qv = ((qv << qb) | (uint64_t)kcode[t + (uint32_t)j]) & mask_events;
	const uint32_t pos = kpos[t];
	uint32_t ro = 0;
	if (o_seed) { ro = ix_kept_below(o_mask, o_prefix, n_ev, o_kept, strand ? pos + 1u : pos); if (ro > o_seed) ro = o_seed; }
	const uint64_t idx = base + t + ro;
	H[idx] = (uint32_t)rh_seed_hash32(qv);
	Y[idx] = (uint64_t)id << 32 | (uint64_t)(pos << 1) | (uint64_t)(uint32_t)strand;
}

// ------------------------------------------------------------------------------------------------ minimisers (w > 0)
// ri_sketch_min (rsketch.c:55-141) keeps, of every w consecutive seeds, the one with the least hash - minimap2's loop, quirks
// included: the NEWEST of equal minima is "the" minimum, the older equal ones are pushed when a minimum is (re)established, a
// minimum is pushed when it is replaced or leaves the window and once more at the end (so a seed can be emitted twice).  The loop is
// serial, but its state after seed t-1 is a function of the last w seeds only: min = the most recent minimum among seeds
// [t-w, t-1], buf_pos = t mod w.  So every seed t replays step t on its own from the w seeds before it: a counting launch, an
// exclusive scan, an emitting launch.  (l of the reference = t + e: the first seed exists once e events are kept.)
struct ix_min_state { uint32_t x; int32_t at; };                 // hash and ordinal of a minimum (at < 0: none yet)
RH_DEV ix_min_state ix_recent_min(const uint32_t *Hs, int64_t lo, int64_t hi)   // most recent minimum of seeds [lo, hi]
{
	ix_min_state m = {0xFFFFFFFFu, -1};
	for (int64_t j = lo; j <= hi; ++j) if (m.at < 0 || Hs[j] <= m.x) { m.x = Hs[j]; m.at = (int32_t)j; }
	return m;
}
// step t of the loop: calls emit(ordinal) for every seed pushed at this step, in the reference's order
template <class F>
RH_DEV void ix_min_step(const uint32_t *Hs, uint32_t n_seed, uint32_t t, int w, F emit)
{
	const int64_t T = (int64_t)t;
	const ix_min_state m = T > 0 ? ix_recent_min(Hs, T - w > 0 ? T - w : 0, T - 1) : ix_min_state{0xFFFFFFFFu, -1};
	const uint32_t hx = Hs[t];
	if (T == (int64_t)w - 1 && m.at >= 0)                            // the first full window: identical seeds have not been stored yet
		for (int64_t j = 0; j < T; ++j) if (Hs[j] == m.x && j != m.at) emit((uint32_t)j);
	ix_min_state cur = m;
	if (m.at < 0 || hx <= m.x) {                                     // a new minimum: write the old one
		if (T >= (int64_t)w && m.at >= 0) emit((uint32_t)m.at);
		cur.x = hx; cur.at = (int32_t)T;
	} else if (m.at == T - (int64_t)w) {                             // the old minimum has left the window
		emit((uint32_t)m.at);
		cur = ix_recent_min(Hs, T - w + 1, T);
		for (int64_t j = T - w + 1; j <= T; ++j) if (Hs[j] == cur.x && j != cur.at) emit((uint32_t)j);
	}
	if (t + 1u == n_seed && cur.at >= 0) emit((uint32_t)cur.at);     // after the loop: the last minimum
}
__global__ __launch_bounds__(NT) void k_ix_hash(const uint8_t *kcode, const uint32_t *kpos, uint32_t n_seed, int strand, uint32_t id, rh_sketch_par sp, uint32_t *Hs, uint64_t *Ys)
{
	const uint32_t t = blockIdx.x * NT + threadIdx.x;
	if (t >= n_seed) return;
	const uint32_t qb = (uint32_t)sp.q;
	const uint64_t mask_events = (qb * sp.e >= 64) ? ~0ULL : ((1ULL << (qb * sp.e)) - 1);
	uint64_t qv = 0;
	for (int j = 0; j < sp.e; ++j) qv = ((qv << qb) | (uint64_t)kcode[t + (uint32_t)j]) & mask_events;
	Hs[t] = (uint32_t)rh_seed_hash32(qv);
	Ys[t] = (uint64_t)id << 32 | (uint64_t)(kpos[t] << 1) | (uint64_t)(uint32_t)strand;
}
__global__ __launch_bounds__(NT) void k_ix_min_count(const uint32_t *Hs, uint32_t n_seed, int w, uint32_t *cnt)
{
	const uint32_t t = blockIdx.x * NT + threadIdx.x;
	if (t >= n_seed) return;
	uint32_t c = 0;
	ix_min_step(Hs, n_seed, t, w, [&](uint32_t) { ++c; });
	cnt[t] = c;
}
__global__ __launch_bounds__(NT) void k_ix_min_emit(const uint32_t *Hs, const uint64_t *Ys, uint32_t n_seed, int w, const uint64_t *off, uint32_t *H, uint64_t *Y, uint64_t base, uint64_t cap)
{
	const uint32_t t = blockIdx.x * NT + threadIdx.x;
	if (t >= n_seed) return;
	uint64_t o = base + off[t];
	ix_min_step(Hs, n_seed, t, w, [&](uint32_t j) { if (o < cap) { H[o] = Hs[j]; Y[o] = Ys[j]; } ++o; });
}

// ------------------------------------------------------------------------------------------------ radix sort by hash
// counts[d * n_tiles + tile]
// (ykey: the digit comes from the position word Y instead of the hash - the passes that put a minimiser index's seeds into position order)
__global__ __launch_bounds__(NT) void k_ix_rs_count(const uint32_t *H, const uint64_t *Y, int ykey, uint64_t n, int shift, uint32_t *counts, uint64_t n_tiles)
{
	__shared__ uint32_t s_cnt[256];
	const uint32_t tid = threadIdx.x;
	s_cnt[tid] = 0;
	__syncthreads();
	const uint64_t base = (uint64_t)blockIdx.x * IX_RT;
	for (uint32_t it = 0; it < IX_RT_IT; ++it) {
		const uint64_t i = base + (uint64_t)it * NT + tid;
		if (i < n) atomicAdd(&s_cnt[ykey ? (uint32_t)(Y[i] >> shift) & 255u : (H[i] >> shift) & 255u], 1u);
	}
	__syncthreads();
	counts[(uint64_t)tid * n_tiles + blockIdx.x] = s_cnt[tid];
}
// block d: offsets of digit d's records per tile (relative to the digit's first record), digit total
__global__ __launch_bounds__(NT) void k_ix_rs_scan(const uint32_t *counts, uint64_t *offs, uint64_t n_tiles, uint64_t *dig_total)
{
	__shared__ uint64_t s_part[NT];
	const uint32_t tid = threadIdx.x, d = blockIdx.x;
	const uint32_t *c = counts + (uint64_t)d * n_tiles;
	uint64_t *o = offs + (uint64_t)d * n_tiles;
	const uint64_t per = (n_tiles + NT - 1) / NT, b = tid * per, e = b + per < n_tiles ? b + per : n_tiles;
	uint64_t s = 0;
	for (uint64_t i = b; i < e; ++i) s += c[i];
	s_part[tid] = s;
	__syncthreads();
	if (tid == 0) { uint64_t run = 0; for (uint32_t i = 0; i < NT; ++i) { const uint64_t v = s_part[i]; s_part[i] = run; run += v; } dig_total[d] = run; }
	__syncthreads();
	uint64_t run = s_part[tid];
	for (uint64_t i = b; i < e; ++i) { o[i] = run; run += c[i]; }
}
__global__ void k_ix_rs_base(uint64_t *dig_total)
{
	if (threadIdx.x == 0) { uint64_t run = 0; for (int d = 0; d < 256; ++d) { const uint64_t v = dig_total[d]; dig_total[d] = run; run += v; } }
}
// stable scatter of one tile: records in tile order = (round, wavefront, lane); ranks from wave ballots
__global__ __launch_bounds__(NT) void k_ix_rs_scatter(const uint32_t *H, const uint64_t *Y, int ykey, uint64_t n, int shift, const uint64_t *offs, const uint64_t *dig_base, uint64_t n_tiles,
                                                      uint32_t *H2, uint64_t *Y2)
{
	__shared__ uint16_t s_hist[IX_RT_IT * (NT / 64) * 256];         // [round][wave][digit]: records of the digit, then (exclusive) those before
	__shared__ uint64_t s_off[256];
	const uint32_t tid = threadIdx.x, w = wave_id(), l = lane_id();
	for (uint32_t i = tid; i < IX_RT_IT * (NT / 64) * 256; i += NT) s_hist[i] = 0;
	s_off[tid] = dig_base[tid] + offs[(uint64_t)tid * n_tiles + blockIdx.x];
	__syncthreads();
	const uint64_t base = (uint64_t)blockIdx.x * IX_RT;
	uint32_t h[IX_RT_IT], rk[IX_RT_IT], dg[IX_RT_IT];
#pragma unroll
	for (uint32_t it = 0; it < IX_RT_IT; ++it) {
		const uint64_t i = base + (uint64_t)it * NT + tid;
		const bool in = i < n;
		h[it] = in ? H[i] : 0u;
		const uint32_t d = in ? (ykey ? (uint32_t)(Y[i] >> shift) & 255u : (h[it] >> shift) & 255u) : 0u;
		dg[it] = d;
		uint64_t peers = __ballot(in);
#pragma unroll
		for (int bit = 0; bit < 8; ++bit) { const uint64_t m = __ballot((d >> bit) & 1u); peers &= ((d >> bit) & 1u) ? m : ~m; }
		rk[it] = (uint32_t)__popcll(peers & ((1ull << l) - 1ull));
		if (in && rk[it] == 0) s_hist[(it * (NT / 64) + w) * 256 + d] = (uint16_t)__popcll(peers);
	}
	__syncthreads();
	{	// per digit: exclusive prefix over (round, wave)
		uint32_t run = 0;
		for (uint32_t q = 0; q < IX_RT_IT * (NT / 64); ++q) { const uint32_t v = s_hist[q * 256 + tid]; s_hist[q * 256 + tid] = (uint16_t)run; run += v; }
	}
	__syncthreads();
#pragma unroll
	for (uint32_t it = 0; it < IX_RT_IT; ++it) {
		const uint64_t i = base + (uint64_t)it * NT + tid;
		if (i < n) {
			const uint32_t d = dg[it];
			const uint64_t dst = s_off[d] + s_hist[(it * (NT / 64) + w) * 256 + d] + rk[it];
			H2[dst] = h[it]; Y2[dst] = Y[i];
		}
	}
}

// ------------------------------------------------------------------------------------------------ keys, table
__global__ __launch_bounds__(NT) void k_ix_key_count(const uint32_t *H, uint64_t n, uint32_t *tile_cnt)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint64_t base = (uint64_t)blockIdx.x * 2048u;
	uint32_t c = 0;
	for (uint32_t it = 0; it < 8; ++it) {
		const uint64_t i = base + (uint64_t)it * NT + threadIdx.x;
		const bool first = i < n && (i == 0 || H[i] != H[i - 1]);
		c += (uint32_t)__popcll(__ballot(first));
	}
	if (lane_id() == 0) s_w[wave_id()] = c;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t q = 0; q < NT / 64; ++q) t += s_w[q]; tile_cnt[blockIdx.x] = t; }
}
__global__ __launch_bounds__(NT) void k_ix_key_write(const uint32_t *H, uint64_t n, const uint64_t *tile_base, uint32_t *khash, uint64_t *kstart)
{
	__shared__ uint32_t s_w[NT / 64];
	const uint64_t base = (uint64_t)blockIdx.x * 2048u;
	uint64_t run = tile_base[blockIdx.x];
	for (uint32_t it = 0; it < 8; ++it) {
		const uint64_t i = base + (uint64_t)it * NT + threadIdx.x;
		const bool first = i < n && (i == 0 || H[i] != H[i - 1]);
		uint32_t tot;
		const uint32_t r = block_rank(first, s_w, tot);
		if (first) { khash[run + r] = H[i]; kstart[run + r] = i; }
		run += tot;
	}
}
// one thread per key: occurrence count, slot in the bucketed table (rh_index_make_table's probing), occupancy histogram
__global__ __launch_bounds__(NT) void k_ix_table(const uint32_t *khash, const uint64_t *kstart, uint64_t n_keys, uint64_t n_pos, const uint64_t *Y, rh_tslot *table, int lg, uint32_t *occ_hist, uint32_t occ_bins)
{
	const uint64_t j = (uint64_t)blockIdx.x * NT + threadIdx.x;
	if (j >= n_keys) return;
	const uint32_t h = khash[j];
	const uint64_t st = kstart[j], cnt = (j + 1 < n_keys ? kstart[j + 1] : n_pos) - st;
	const uint32_t n32 = cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt;
	atomicAdd(&occ_hist[n32 < occ_bins - 1 ? n32 : occ_bins - 1], 1u);
	const uint64_t nb = 1ULL << lg, word = (uint64_t)h | (uint64_t)n32 << 32;
	uint64_t b = (uint64_t)((uint32_t)(h * 0x9E3779B1u) >> (32 - lg));
	for (;;) {
		rh_tslot *s = table + b * RH_TB_SLOTS;
		for (int q = 0; q < RH_TB_SLOTS; ++q)
			if (atomicCAS((unsigned long long*)&s[q], 0ull, (unsigned long long)word) == 0ull) { s[q].val = n32 == 1 ? Y[st] : st; return; }
		b = (b + 1) & (nb - 1);
	}
}

// ------------------------------------------------------------------------------------------------ host
namespace {

template <class OUT_T>
int ix_scan(hipStream_t s, const uint32_t *in, uint64_t n, OUT_T *out, uint64_t *sums, uint64_t *total_dev)
{
	const uint64_t tiles = (n + 2047) / 2048;
	if (!tiles) { RH_HIP(hipMemsetAsync(total_dev, 0, 8, s)); return 0; }
	RH_LAUNCH(k_ix_scan_tiles<OUT_T>, (uint32_t)tiles, NT, 0, s, in, n, out, sums);
	RH_LAUNCH(k_ix_scan_sums, 1, NT, 0, s, sums, tiles, total_dev);
	RH_LAUNCH(k_ix_scan_add<OUT_T>, (uint32_t)tiles, NT, 0, s, out, n, (const uint64_t*)sums);
	return 0;
}

// runs of bases that are not A/C/G/T/U (either case) of one sequence
void find_invalid_runs(const char *seq, uint32_t len, std::vector<uint32_t> &rs, std::vector<uint32_t> &re)
{
	static const auto valid = [](unsigned char c) { switch (c) { case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': case 'U': case 'u': return true; default: return false; } };
	uint32_t i = 0;
	while (i < len) {
		if (valid((unsigned char)seq[i])) { ++i; continue; }
		uint32_t j = i + 1;
		while (j < len && !valid((unsigned char)seq[j])) ++j;
		rs.push_back(i); re.push_back(j);
		i = j;
	}
}

} // namespace

// Builds the resident index blob [table | positions | target lengths] on the current device.  Outputs: the device blob
// (caller owns it), its header fields, the per-key occupancy histogram (mid_occ calibration), counts.
int rhk_index_build_device(hipStream_t s, uint32_t n_seq, const char *const *seqs, const uint32_t *lens, const std::vector<float> &model /* normalised, 4^k */,
                           const rh_idxopt_t *io, rh_blob_header *hdr, void **blob_out, std::vector<uint32_t> &occ_hist, uint64_t *n_keys_out, int n_threads)
{
	const int k = io->k, e = io->e;
	const rh_sketch_par sp = {io->e, io->w, io->q, io->k, io->diff, io->fine_min, io->fine_max, io->fine_range};
	const int n_strands = (io->flag & RH_I_NO_REV_TARGET) ? 1 : 2;
	uint32_t max_len = 0;
	uint64_t total_ev = 0;
	for (uint32_t i = 0; i < n_seq; ++i) { if (lens[i] > max_len) max_len = lens[i]; if (lens[i] >= (uint32_t)k) total_ev += (uint64_t)(lens[i] - k + 1) * n_strands; }
	if (max_len >= (1u << 31)) { rh_set_error("index build: target sequences of 2^31 bases or more are not supported"); return -1; }
	const uint32_t max_ev = max_len >= (uint32_t)k ? max_len - k + 1 : 0, max_words = max_ev / 32 + 2, max_blocks = max_ev / IX_L + 2;
	// seeds of all sequences: at most one per event
	DevMem dH[2], dY[2], dModel, dSeq, dLv, dMask[2], dPc, dPrefix[2], dCode[2], dPos[2], dIn, dOut, dFlag, dSums, dScal, dRuns, dHs, dYs, dCnt, dOffs;
	const uint64_t seed_cap = total_ev + 1;
	if (io->w > 0) {	// minimisers: a strand's seeds (hash, position word) before the window filter, emission counts and offsets
		if (io->w > 255) { rh_set_error("index build: minimiser window %d > 255", io->w); return -1; }
		if (dHs.alloc((size_t)max_ev * 4 + 16) || dYs.alloc((size_t)max_ev * 8 + 16) || dCnt.alloc((size_t)max_ev * 4 + 16) || dOffs.alloc((size_t)max_ev * 8 + 16)) return -1;
	}
	if (dH[0].alloc((total_ev + 1) * 4) || dY[0].alloc((total_ev + 1) * 8) || dModel.alloc(model.size() * 4) || dSeq.alloc((size_t)max_len + 16) || dLv.alloc((size_t)max_ev * 4 + 16)) return -1;
	for (int q = 0; q < 2; ++q) if (dMask[q].alloc((size_t)max_words * 4) || dPrefix[q].alloc((size_t)max_words * 4) || dCode[q].alloc((size_t)max_ev + 64) || dPos[q].alloc((size_t)max_ev * 4 + 16)) return -1;
	if (dPc.alloc((size_t)max_words * 4) || dIn.alloc((size_t)max_blocks * 4) || dOut.alloc((size_t)max_blocks * 4) || dFlag.alloc(max_blocks) || dSums.alloc(((size_t)(io->w > 0 ? max_ev : max_words) / 2048 + 4) * 8) || dScal.alloc(64)) return -1;
	RH_HIP(hipMemcpyAsync(dModel.p, model.data(), model.size() * 4, hipMemcpyHostToDevice, s));
	// RH_I_STORE_SIG (--store-sig, rindex.c:133-160, 590-598): the levels k_ix_levels computes ARE the targets' expected signals - kept, strand
	// after strand, and laid out behind the target lengths as [u64 so[2 n + 1] | floats] (what rh_index_upload lays out for a host-built index)
	const bool store_sig = (io->flag & RH_I_STORE_SIG) != 0;
	std::vector<uint64_t> so;
	DevMem dSig;
	if (store_sig) {
		so.assign((size_t)2 * n_seq + 1, 0);
		for (uint32_t i = 0; i < n_seq; ++i) {
			const uint64_t ne = lens[i] >= (uint32_t)k ? lens[i] - k + 1 : 0;
			so[2 * (size_t)i + 1] = so[2 * (size_t)i] + ne;
			so[2 * (size_t)i + 2] = so[2 * (size_t)i + 1] + (n_strands == 2 ? ne : 0);
		}
		if (dSig.alloc((so.back() + 2) * 4)) return -1;
		RH_HIP(hipMemsetAsync(dSig.p, 0, (so.back() + 2) * 4, s));    // (two floats of zero padding: the DTW's global border looks one element past a signal, as the reference does)
	}
	uint64_t n_seeds = 0;
	uint64_t *scal = nullptr;
	RH_HIP(hipHostMalloc((void**)&scal, 64, 0));
	struct PinFree { uint64_t *p; ~PinFree() { if (p) (void)hipHostFree(p); } } pin_free{scal};
	// runs of invalid bases of every sequence (host, all threads)
	std::vector<std::vector<uint32_t>> all_rs(n_seq), all_re(n_seq);
	{
		if (n_threads < 1) n_threads = 1;
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t)
			th.emplace_back([&, t]() { for (uint32_t i = (uint32_t)t; i < n_seq; i += (uint32_t)n_threads) find_invalid_runs(seqs[i], lens[i], all_rs[i], all_re[i]); });
		for (auto &t : th) t.join();
	}
	std::vector<uint32_t> cum;
	for (uint32_t id = 0; id < n_seq; ++id) {
		const uint32_t len = lens[id];
		if (len < (uint32_t)k) continue;
		const uint32_t n_ev = len - k + 1, n_words = (n_ev + 31) / 32, n_blocks = (n_ev + IX_L - 1) / IX_L;
		RH_HIP(hipMemcpyAsync(dSeq.p, seqs[id], len, hipMemcpyHostToDevice, s));
		const std::vector<uint32_t> &rs = all_rs[id], &re = all_re[id];
		ix_runs R = {nullptr, nullptr, nullptr, (uint32_t)rs.size(), len};
		if (!rs.empty()) {
			cum.assign(rs.size() + 1, 0);
			for (size_t j = 0; j < rs.size(); ++j) cum[j + 1] = cum[j] + (re[j] - rs[j]);
			R.n_valid = len - cum.back();
			const size_t nr = rs.size();
			if (dRuns.alloc((3 * nr + 1) * 4)) return -1;
			RH_HIP(hipMemcpy(dRuns.p, rs.data(), nr * 4, hipMemcpyHostToDevice));
			RH_HIP(hipMemcpy(dRuns.as<uint32_t>() + nr, re.data(), nr * 4, hipMemcpyHostToDevice));
			RH_HIP(hipMemcpy(dRuns.as<uint32_t>() + 2 * nr, cum.data(), (nr + 1) * 4, hipMemcpyHostToDevice));
			R.rs = dRuns.as<uint32_t>(); R.re = R.rs + nr; R.cum = R.rs + 2 * nr;
		}
		uint32_t kept[2] = {0, 0}, seeds[2] = {0, 0};
		for (int st = 0; st < n_strands; ++st) {
			RH_LAUNCH(k_ix_levels, (n_ev + NT - 1) / NT, NT, 0, s, dSeq.as<char>(), len, st, k, dModel.as<float>(), R, dLv.as<float>(), n_ev);
			if (store_sig) RH_HIP(hipMemcpyAsync(dSig.as<float>() + so[2 * (size_t)id + st], dLv.p, (size_t)n_ev * 4, hipMemcpyDeviceToDevice, s));
			RH_LAUNCH(k_ix_keep, (n_blocks + 63) / 64, 64, 0, s, dLv.as<float>(), n_ev, io->diff, dIn.as<float>(), dOut.as<float>(), dMask[st].as<uint32_t>(), (const uint8_t*)nullptr, n_blocks);
			for (int iter = 0;; ++iter) {
				RH_HIP(hipMemsetAsync(dScal.p, 0, 8, s));
				RH_LAUNCH(k_ix_verify, (n_blocks + NT - 1) / NT, NT, 0, s, dIn.as<float>(), dOut.as<float>(), dFlag.as<uint8_t>(), n_blocks, dScal.as<uint32_t>());
				RH_HIP(hipMemcpyAsync(scal, dScal.p, 8, hipMemcpyDeviceToHost, s));
				RH_HIP(hipStreamSynchronize(s));
				if ((uint32_t)scal[0] == 0) break;
				if (iter > (int)n_blocks + 2) { rh_set_error("index build: the event filter did not settle"); return -1; }
				RH_LAUNCH(k_ix_keep, (n_blocks + 63) / 64, 64, 0, s, dLv.as<float>(), n_ev, io->diff, dIn.as<float>(), dOut.as<float>(), dMask[st].as<uint32_t>(), dFlag.as<uint8_t>(), n_blocks);
			}
			RH_LAUNCH(k_ix_popc, (n_words + NT - 1) / NT, NT, 0, s, dMask[st].as<uint32_t>(), n_words, dPc.as<uint32_t>());
			if (ix_scan<uint32_t>(s, dPc.as<uint32_t>(), n_words, dPrefix[st].as<uint32_t>(), dSums.as<uint64_t>(), dScal.as<uint64_t>())) return -1;
			RH_HIP(hipMemcpyAsync(scal, dScal.p, 8, hipMemcpyDeviceToHost, s));
			RH_HIP(hipStreamSynchronize(s));
			kept[st] = (uint32_t)scal[0];
			seeds[st] = kept[st] >= (uint32_t)e ? kept[st] - e + 1 : 0;
			RH_LAUNCH(k_ix_compact, (n_words + NT - 1) / NT, NT, 0, s, dLv.as<float>(), dMask[st].as<uint32_t>(), dPrefix[st].as<uint32_t>(), n_words, sp, dCode[st].as<uint8_t>(), dPos[st].as<uint32_t>());
		}
		if (io->w > 0) {	// ri_sketch_min: window filter of every strand's seed stream; the position order is restored by the assembly's sort
			for (int st = 0; st < n_strands; ++st) {
				if (!seeds[st]) continue;
				const uint32_t ns = seeds[st];
				RH_LAUNCH(k_ix_hash, (ns + NT - 1) / NT, NT, 0, s, dCode[st].as<uint8_t>(), dPos[st].as<uint32_t>(), ns, st, id, sp, dHs.as<uint32_t>(), dYs.as<uint64_t>());
				RH_LAUNCH(k_ix_min_count, (ns + NT - 1) / NT, NT, 0, s, dHs.as<uint32_t>(), ns, io->w, dCnt.as<uint32_t>());
				if (ix_scan<uint64_t>(s, dCnt.as<uint32_t>(), ns, dOffs.as<uint64_t>(), dSums.as<uint64_t>(), dScal.as<uint64_t>())) return -1;
				RH_HIP(hipMemcpyAsync(scal, dScal.p, 8, hipMemcpyDeviceToHost, s));
				RH_HIP(hipStreamSynchronize(s));
				const uint64_t emitted = scal[0];
				if (n_seeds + emitted > seed_cap) { rh_set_error("index build: more minimisers than events (%llu > %llu)", (unsigned long long)(n_seeds + emitted), (unsigned long long)seed_cap); return -1; }
				RH_LAUNCH(k_ix_min_emit, (ns + NT - 1) / NT, NT, 0, s, dHs.as<uint32_t>(), dYs.as<uint64_t>(), ns, io->w, dOffs.as<uint64_t>(), dH[0].as<uint32_t>(), dY[0].as<uint64_t>(), n_seeds, seed_cap);
				n_seeds += emitted;
			}
			RH_HIP(hipStreamSynchronize(s));
			continue;
		}
		for (int st = 0; st < n_strands; ++st) {
			if (!seeds[st]) continue;
			const int o = st ^ 1;
			const uint32_t o_seed = n_strands == 2 ? seeds[o] : 0u;
			RH_LAUNCH(k_ix_seeds, (seeds[st] + NT - 1) / NT, NT, 0, s, dCode[st].as<uint8_t>(), dPos[st].as<uint32_t>(), seeds[st], st, id, sp,
			          dMask[o].as<uint32_t>(), dPrefix[o].as<uint32_t>(), n_strands == 2 ? kept[o] : 0u, o_seed, n_ev, dH[0].as<uint32_t>(), dY[0].as<uint64_t>(), n_seeds);
		}
		n_seeds += (uint64_t)seeds[0] + seeds[1];
		RH_HIP(hipStreamSynchronize(s));                            // the staging buffers are reused by the next sequence
	}
	// per-sequence scratch is done with
	dSeq.release(); dLv.release(); dPc.release(); dIn.release(); dOut.release(); dFlag.release(); dRuns.release(); dHs.release(); dYs.release(); dCnt.release(); dOffs.release();
	for (int q = 0; q < 2; ++q) { dMask[q].release(); dPrefix[q].release(); dCode[q].release(); dPos[q].release(); }
	void *hp = dH[0].p, *yp = dY[0].p;
	dH[0].p = nullptr; dY[0].p = nullptr;                         // (handed over)
	const uint64_t extra = store_sig ? 8 + so.size() * 8 + (so.back() + 2) * 4 : 0;
	if (rhk_index_assemble(s, hp, yp, n_seeds, n_seq, lens, max_len, io, hdr, blob_out, occ_hist, n_keys_out, io->w > 0, extra)) return -1;
	if (store_sig) {
		unsigned char *bp = (unsigned char*)*blob_out;
		hdr->bytes = (hdr->bytes + 7) & ~7ull;
		hdr->sig_off = hdr->bytes;
		RH_HIP(hipMemcpyAsync(bp + hdr->sig_off, so.data(), so.size() * 8, hipMemcpyHostToDevice, s));
		RH_HIP(hipMemcpyAsync(bp + hdr->sig_off + so.size() * 8, dSig.p, (so.back() + 2) * 4, hipMemcpyDeviceToDevice, s));
		RH_HIP(hipStreamSynchronize(s));
		hdr->bytes += so.size() * 8 + (so.back() + 2) * 4;
	}
	return 0;
}

// Seeds (32-bit hash, position word) in target order -> resident blob [table | positions | target lengths].  Takes ownership
// of the two device arrays (hipMalloc'ed, n_seeds + 1 entries).
int rhk_index_assemble(hipStream_t s, void *seed_hash, void *seed_pos, uint64_t n_seeds, uint32_t n_seq, const uint32_t *lens, uint32_t max_len,
                       const rh_idxopt_t *io, rh_blob_header *hdr, void **blob_out, std::vector<uint32_t> &occ_hist, uint64_t *n_keys_out, bool sort_pos, uint64_t extra_bytes)
{
	const rh_sketch_par sp = {io->e, io->w, io->q, io->k, io->diff, io->fine_min, io->fine_max, io->fine_range};
	DevMem dH[2], dY[2], dSums, dScal;
	dH[0].p = seed_hash; dY[0].p = seed_pos;
	if (dScal.alloc(64)) return -1;
	uint64_t *scal = nullptr;
	RH_HIP(hipHostMalloc((void**)&scal, 64, 0));
	struct PinFree { uint64_t *p; ~PinFree() { if (p) (void)hipHostFree(p); } } pin_free{scal};
	// ---- stable LSD radix sort by hash
	const uint64_t N = n_seeds, n_tiles = (N + IX_RT - 1) / IX_RT;
	int cur = 0;
	if (N) {
		DevMem dCnt, dOff, dDig;
		if (dH[1].alloc((N + 1) * 4) || dY[1].alloc((N + 1) * 8) || dCnt.alloc(n_tiles * 256 * 4) || dOff.alloc(n_tiles * 256 * 8) || dDig.alloc(256 * 8)) return -1;
		// seeds that do not arrive in position order (minimiser indexes: emitted strand by strand, some twice) are first put into the
		// order of their position words id << 32 | pos << 1 | strand - what radix_sort_64 leaves a key's list in (rindex.c:350)
		int y_bytes = 0;
		if (sort_pos) { y_bytes = 4; for (uint64_t v = n_seq ? n_seq - 1 : 0; v; v >>= 8) ++y_bytes; }
		for (int pass = 0; pass < y_bytes + 4; ++pass) {
			const int ykey = pass < y_bytes, shift = ykey ? pass * 8 : (pass - y_bytes) * 8;
			RH_LAUNCH(k_ix_rs_count, (uint32_t)n_tiles, NT, 0, s, dH[cur].as<uint32_t>(), dY[cur].as<uint64_t>(), ykey, N, shift, dCnt.as<uint32_t>(), n_tiles);
			RH_LAUNCH(k_ix_rs_scan, 256, NT, 0, s, dCnt.as<uint32_t>(), dOff.as<uint64_t>(), n_tiles, dDig.as<uint64_t>());
			RH_LAUNCH(k_ix_rs_base, 1, 64, 0, s, dDig.as<uint64_t>());
			RH_LAUNCH(k_ix_rs_scatter, (uint32_t)n_tiles, NT, 0, s, dH[cur].as<uint32_t>(), dY[cur].as<uint64_t>(), ykey, N, shift, dOff.as<uint64_t>(), dDig.as<uint64_t>(), n_tiles,
			          dH[cur ^ 1].as<uint32_t>(), dY[cur ^ 1].as<uint64_t>());
			cur ^= 1;
		}
		RH_HIP(hipStreamSynchronize(s));
		dH[cur ^ 1].release(); dY[cur ^ 1].release();
	}
	// ---- keys
	uint64_t n_keys = 0;
	DevMem dKh, dKs, dTc, dTb;
	const uint64_t ktiles = (N + 2047) / 2048;
	if (N) {
		if (dTc.alloc(ktiles * 4) || dTb.alloc(ktiles * 8) || dSums.alloc((ktiles / 2048 + 4) * 8)) return -1;
		RH_LAUNCH(k_ix_key_count, (uint32_t)ktiles, NT, 0, s, dH[cur].as<uint32_t>(), N, dTc.as<uint32_t>());
		if (ix_scan<uint64_t>(s, dTc.as<uint32_t>(), ktiles, dTb.as<uint64_t>(), dSums.as<uint64_t>(), dScal.as<uint64_t>())) return -1;
		RH_HIP(hipMemcpyAsync(scal, dScal.p, 8, hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		n_keys = scal[0];
		if (dKh.alloc(n_keys * 4) || dKs.alloc(n_keys * 8)) return -1;
		RH_LAUNCH(k_ix_key_write, (uint32_t)ktiles, NT, 0, s, dH[cur].as<uint32_t>(), N, dTb.as<uint64_t>(), dKh.as<uint32_t>(), dKs.as<uint64_t>());
	}
	dH[cur].release();
	// ---- blob: [table | positions | target lengths]
	int lg = 4;
	while (((uint64_t)RH_TB_SLOTS << lg) < n_keys * 2) ++lg;
	memset(hdr, 0, sizeof(*hdr));
	hdr->magic = RH_BLOB_MAGIC;
	hdr->table_off = 0;
	hdr->pos_off = ((uint64_t)RH_TB_SLOTS << lg) * sizeof(rh_tslot);
	hdr->n_pos = N;
	hdr->len_off = hdr->pos_off + (N ? N : 1) * 8;
	hdr->bytes = hdr->len_off + (n_seq ? n_seq : 1) * 4;
	hdr->lg_buckets = lg; hdr->n_seq = n_seq; hdr->flag = io->flag; hdr->sp = sp; hdr->max_len = max_len;
	DevMem blob, dHist;
	const uint32_t occ_bins = 1u << 20;
	if (blob.alloc(hdr->bytes + extra_bytes) || dHist.alloc((size_t)occ_bins * 4)) return -1;   // (extra_bytes: room behind the lengths - the caller's --store-sig signals)
	unsigned char *bp = blob.as<unsigned char>();
	RH_HIP(hipMemsetAsync(bp + hdr->table_off, 0, hdr->pos_off, s));
	RH_HIP(hipMemsetAsync(dHist.p, 0, (size_t)occ_bins * 4, s));
	if (N) RH_HIP(hipMemcpyAsync(bp + hdr->pos_off, dY[cur].p, N * 8, hipMemcpyDeviceToDevice, s));
	if (n_seq) RH_HIP(hipMemcpyAsync(bp + hdr->len_off, lens, (size_t)n_seq * 4, hipMemcpyHostToDevice, s));
	if (n_keys) RH_LAUNCH(k_ix_table, (uint32_t)((n_keys + NT - 1) / NT), NT, 0, s, dKh.as<uint32_t>(), dKs.as<uint64_t>(), n_keys, N, (const uint64_t*)(bp + hdr->pos_off), (rh_tslot*)(bp + hdr->table_off), lg, dHist.as<uint32_t>(), occ_bins);
	occ_hist.assign(occ_bins, 0);
	RH_HIP(hipMemcpyAsync(occ_hist.data(), dHist.p, (size_t)occ_bins * 4, hipMemcpyDeviceToHost, s));
	RH_HIP(hipStreamSynchronize(s));
	RH_HIP(hipGetLastError());
	*blob_out = blob.p; blob.p = nullptr;
	*n_keys_out = n_keys;
	return 0;
}
