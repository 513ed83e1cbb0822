// Arithmetic shared by host code (index construction) and device kernels (mapping): seed hashing, adaptive
// quantisation, the sketch state machine, the chaining pair score.  Pure functions, no memory management.
// Compile with -ffp-contract=off on every side: the reference's results depend on unfused fp32.
#pragma once
#include "rh_gpu.h"   // RH_HD = __host__ __device__
#include <cstdint>
#include <cfloat>

struct rh_sketch_par {
	int32_t e, w, q, k;
	float diff, fine_min, fine_max, fine_range;
};

// 32-bit-masked invertible integer mix of the packed quantised events (reference rsketch.c:7-16)
RH_HD inline uint64_t rh_seed_hash32(uint64_t key)
{
	const uint64_t m = 0xFFFFFFFFULL;
	key = (~key + (key << 21)) & m;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & m;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & m;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & m;
	return key;
}

// Unmasked variant used to randomise chain order (reference hit.c:73-83)
RH_HD inline uint64_t rh_mix64_nomask(uint64_t key)
{
	key = (~key + (key << 21));
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4));
	key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}

// 32-bit Wang hash (reference khash.h:400-409), seeds the chain-order hash (rmap.cpp:346-348)
RH_HD inline uint32_t rh_wang32(uint32_t key)
{
	key += ~(key << 15);
	key ^= (key >> 10);
	key += (key << 3);
	key ^= (key >> 6);
	key += ~(key << 11);
	key ^= (key >> 16);
	return key;
}

// Adaptive quantisation of a normalised event value into n_buckets codes: fine resolution inside
// [fine_min, fine_max], coarse outside (reference rsketch.c:18-53).  All fp32, evaluation order as written there.
RH_HD inline uint32_t rh_quantise(float s, float fine_min, float fine_max, float fine_range, uint32_t n_buckets)
{
	const float lo = -3.0f, hi = 3.0f;
	const float range = hi - lo;
	const float c1 = (1 - fine_range) / 2;
	const float c2 = fine_range + c1;
	const float nrm = (s - lo) / range;
	const float a = (fine_min - lo) / range;
	const float b = (fine_max - lo) / range;
	float qz;
	if (s >= fine_min && s <= fine_max) qz = fine_range * ((nrm - a) / (b - a));
	else if (nrm < 0.5f) qz = fine_range + c1 * nrm;
	else qz = c2 + c1 * nrm;
	return (uint32_t)(qz * (n_buckets - 1));
}

// Working storage of the sketch loop, by policy: plain arrays on the host; on the device an LDS slice per lane (a
// lane-dependent index into a register array would be spilled to scratch memory).
template <int MAXW>
struct rh_sketch_store_local {
	uint32_t r[16];
	uint64_t x[MAXW > 0 ? MAXW : 1], y[MAXW > 0 ? MAXW : 1];
	RH_HD uint32_t &ring(int i) { return r[i]; }
	RH_HD uint64_t &bx(int i) { return x[i]; }
	RH_HD uint64_t &by(int i) { return y[i]; }
};

// Sketch of one event array (reference ri_sketch rsketch.c:271 -> ri_sketch_reg :143-204 for w == 0,
// ri_sketch_min :55-141 for w > 0).  `emit(x, y)` receives every seed in output order:
//   x = hash << 6 | span,  y = id << 32 | first_event_pos << 1 | strand.
// MAXW bounds the minimiser window the caller is prepared to hold (256 on the host, small on the device).
template <int MAXW, class Emit, class Store>
RH_HD inline void rh_sketch_events(const float *ev, uint32_t len, uint32_t id, int strand, const rh_sketch_par &sp, Emit &emit, Store &st)
{
	const int e = sp.e, w = sp.w;
	const uint32_t qb = (uint32_t)sp.q, n_buckets = 1u << qb;
	const uint64_t span = (uint64_t)(sp.k + e - 1);
	const uint64_t id_shift = (uint64_t)id << 32;
	const uint64_t mask_events = (qb * e >= 64) ? ~0ULL : ((1ULL << (qb * e)) - 1);
	const uint64_t mask_q = (1ULL << qb) - 1;
	// ring of the last e kept events: slot r holds the position of the event that STARTS the e-mer whose hash (x) is
	// written e-1 kept events later
	if (len == 0 || e > 16) return;
	for (int i = 0; i < 16; ++i) st.ring(i) = 0;
	int rp = 0, full = 0;
	uint32_t last = 0, n_kept = 0;
	uint64_t qv = 0;
	// minimiser state (w > 0)
	uint64_t min_x = ~0ULL, min_y = ~0ULL;
	int buf_pos = 0, min_pos = 0;
	if (w > 0) { if (w > MAXW) return; for (int i = 0; i < w; ++i) st.bx(i) = st.by(i) = ~0ULL; }

	for (uint32_t f = 0; f < len; ++f) {
		if (f > 0) { float d = ev[f] - ev[last]; if ((d < 0 ? -d : d) < sp.diff) continue; }
		last = f;
		++n_kept;
		const uint64_t code = rh_quantise(ev[f], sp.fine_min, sp.fine_max, sp.fine_range, n_buckets) & mask_q;
		qv = ((qv << qb) | code) & mask_events;
		st.ring(rp) = f;
		if (++rp == e) { full = 1; rp = 0; }
		if (!full) continue;
		const uint64_t x = (rh_seed_hash32(qv) << 6) | span, y = id_shift | (uint64_t)(uint32_t)(st.ring(rp) << 1) | (uint64_t)(uint32_t)strand;
		if (w == 0) { emit(x, y); continue; }
		// ---- minimiser selection over windows of w consecutive e-mers (duplicates of the minimum are kept)
		const uint32_t l = n_kept;
		st.bx(buf_pos) = x; st.by(buf_pos) = y;
		if (l == (uint32_t)(w + e - 1) && min_x != ~0ULL) {
			for (int j = buf_pos + 1; j < w; ++j) if (min_x == st.bx(j) && st.by(j) != min_y) emit(st.bx(j), st.by(j));
			for (int j = 0; j < buf_pos; ++j) if (min_x == st.bx(j) && st.by(j) != min_y) emit(st.bx(j), st.by(j));
		}
		if (x <= min_x) {
			if (l >= (uint32_t)(w + e) && min_x != ~0ULL) emit(min_x, min_y);
			min_x = x; min_y = y; min_pos = buf_pos;
		} else if (buf_pos == min_pos) {
			if (l >= (uint32_t)(w + e - 1) && min_x != ~0ULL) emit(min_x, min_y);
			min_x = ~0ULL;
			for (int j = buf_pos + 1; j < w; ++j) if (min_x >= st.bx(j)) { min_x = st.bx(j); min_y = st.by(j); min_pos = j; }
			for (int j = 0; j <= buf_pos; ++j) if (min_x >= st.bx(j)) { min_x = st.bx(j); min_y = st.by(j); min_pos = j; }
			if (l >= (uint32_t)(w + e - 1) && min_x != ~0ULL) {
				for (int j = buf_pos + 1; j < w; ++j) if (min_x == st.bx(j) && min_y != st.by(j)) emit(st.bx(j), st.by(j));
				for (int j = 0; j <= buf_pos; ++j) if (min_x == st.bx(j) && min_y != st.by(j)) emit(st.bx(j), st.by(j));
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (w > 0 && min_x != ~0ULL) emit(min_x, min_y);
}

// log2 approximation used by the gap penalty (reference lchain.c:23-31); valid for x >= 2
RH_HD inline float rh_log2_approx(float x)
{
	union { float f; uint32_t i; } z;
	z.f = x;
	float l = (float)(int)(((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	l += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return l;
}

#define RH_SCORE_NONE INT32_MIN

// Score of chaining anchor i after anchor j (reference compute_score lchain.c:297-356), on the coordinate differences:
//   dq = (int32)y_i - (int32)y_j (query), dr = (int32)(x_i - x_j) (target), q_span = span of anchor j.
RH_HD inline int32_t rh_pair_score_d(int32_t dq, int32_t dr, int32_t q_span, int32_t max_dist_t, int32_t max_dist_q,
                                     int32_t bw, float pen_gap, float pen_skip)
{
	if (dq <= 0 || dq > max_dist_q) return RH_SCORE_NONE;
	if (dr == 0 || dr > max_dist_t) return RH_SCORE_NONE;
	const int32_t dd = dr > dq ? dr - dq : dq - dr;
	if (dd > bw || dr > max_dist_q) return RH_SCORE_NONE;
	const int32_t dg = dr < dq ? dr : dq;
	int32_t sc = q_span < dg ? q_span : dg;
	if (dd || dg > q_span) {
		const float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		const float lg = dd >= 1 ? rh_log2_approx((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

// Same on full anchors: x = rev<<63 | rid<<32 | ref_pos, y = flags<<40 | q_span<<32 | q_pos.
RH_HD inline int32_t rh_pair_score(uint64_t xi, uint64_t yi, uint64_t xj, uint64_t yj, int32_t max_dist_t, int32_t max_dist_q,
                                   int32_t bw, float pen_gap, float pen_skip)
{
	return rh_pair_score_d((int32_t)yi - (int32_t)yj, (int32_t)(xi - xj), (int32_t)((yj >> 32) & 63), max_dist_t, max_dist_q, bw, pen_gap, pen_skip);
}
