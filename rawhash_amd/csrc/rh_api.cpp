// C-ABI entry points that touch the GPU: context, index residency, the batched mapping call
// (= kt_for(map_worker_for), reference rmap.cpp:700) and the stage-level calls used by the parity tests.
// Host orchestration only: all arithmetic of the path is in rh_kernels.hip.  There is no CPU fallback.
#include "rh_kernels.h"
#include <chrono>
#include <cmath>
#include <memory>
#include <string>
#include <thread>
#include <cstdlib>
#include <atomic>
#include <dlfcn.h>

namespace {

thread_local bool g_oom = false;   // the last failure on this thread was a device allocation that did not fit

struct DevBuf {
	void *p = nullptr; size_t cap = 0; bool owned = true;
	int ensure(size_t bytes, bool margin = true)
	{
		if (bytes <= cap) return 0;
		if (p) { RH_HIP(hipFree(p)); p = nullptr; cap = 0; }
		size_t want = margin ? bytes + bytes / 4 + 256 : bytes + 256;   // (anchor-sized arenas are budgeted: no growth margin)
		// keep head-room on the device: the runtime itself allocates at dispatch time (kernel scratch, queues) and aborts the
		// process when that fails, so an arena that would eat the last gigabytes is refused here like a failed hipMalloc
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
			const size_t reserve = total_b / 32 > ((size_t)2 << 30) ? total_b / 32 : ((size_t)2 << 30);
			if (want + reserve > free_b) want = bytes + 256;
			if (want + reserve > free_b) {
				g_oom = true;
				rh_set_error("%s:%d: device arena of %zu bytes does not fit (%zu free, %zu kept in reserve)", __FILE__, __LINE__, want, free_b, reserve);
				return -1;
			}
		}
		hipError_t e = hipMalloc(&p, want);
		if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); want = bytes + 256; e = hipMalloc(&p, want); }   // without the growth margin
		if (e != hipSuccess) {
			p = nullptr;
			if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); g_oom = true; }
			rh_set_error("%s:%d: hipMalloc of %zu bytes failed: %s", __FILE__, __LINE__, want, hipGetErrorString(e));
			return -1;
		}
		cap = want;
		return 0;
	}
	// grow to at least `bytes`, keeping the first `used` bytes
	int ensure_keep(size_t bytes, size_t used, hipStream_t s)
	{
		if (bytes <= cap) return 0;
		DevBuf nb;
		if (nb.ensure(bytes + bytes / 2)) { if (nb.ensure(bytes)) return -1; g_oom = false; }   // (the fallback fitted: no out-of-memory condition is pending)
		if (p && used) { RH_HIP(hipMemcpyAsync(nb.p, p, used, hipMemcpyDeviceToDevice, s)); RH_HIP(hipStreamSynchronize(s)); }
		if (p) (void)hipFree(p);
		p = nb.p; cap = nb.cap; nb.p = nullptr;
		return 0;
	}
	// grow to `want` bytes if that fits, else to `bytes` (the least that will do), keeping the first `used` bytes; no growth margin on top: the caller has sized `want`
	int ensure_keep_sized(size_t bytes, size_t want, size_t used, hipStream_t s)
	{
		if (bytes <= cap) return 0;
		DevBuf nb;
		if (want <= bytes || nb.ensure(want, false)) { if (nb.ensure(bytes, false)) return -1; g_oom = false; }
		if (p && used) { RH_HIP(hipMemcpyAsync(nb.p, p, used, hipMemcpyDeviceToDevice, s)); RH_HIP(hipStreamSynchronize(s)); }
		if (p) (void)hipFree(p);
		p = nb.p; cap = nb.cap; nb.p = nullptr;
		return 0;
	}
	void release() { if (p && owned) (void)hipFree(p); p = nullptr; cap = 0; owned = true; }
	template <class T> T *as() const { return (T*)p; }
};

enum Stage { ST_H2D = 0, ST_PREFILTER, ST_EV_NORM, ST_EV_PEAKS, ST_EV_MEANS, ST_SKETCH, ST_PROBE, ST_SCAN, ST_EXPAND, ST_SORT, ST_CHAIN, ST_ZSORT, ST_BACKTRACK, ST_RSORT, ST_REGIONS, ST_COMPACT, ST_FINALIZE, ST_D2H, ST_N };
const char *kStageName[24] = {"h2d", "prefilter", "events_norm", "events_peaks", "events_means", "sketch", "probe", "scan", "expand", "sort", "chain", "zsort", "backtrack", "rsort", "regions", "compact", "finalize", "d2h", "", "", "", "", "", ""};

} // namespace

struct rh_ctx_s {
	int device = -1;
	hipStream_t stream = nullptr;
	// resident index
	DevBuf blob; bool blob_owned = true;
	rh_dev_index dix{};
	bool have_index = false;
	bool akey_on = false; uint8_t akey_lo = 0, akey_mid = 0;      // dimensions of the anchor keys of the resident index
	uint64_t dtw_dev_reads = 0, dtw_host_reads = 0;              // DTW re-scoring: (read, chunk) pairs whose MAPQ / decision the device settled / the host's libm had to
	size_t carry_per_read = 0;                                    // most bytes of chained anchors a round carried per read of its call, over this context's calls (map_batch_single sizes its calls by it)
	unsigned char header[256] = {0};
	// logf table
	DevBuf logf_tab;
	// batch state + per-round arenas (grow only)
	DevBuf raw, off, cal_off, cal_scale;
	DevBuf res_len, cnt_res, new_len, lsig_given;                 // consumed-prefix staging (rh_read_batch_t::n_filtered + page-locked samples): see k_need / k_fetch
	const int16_t *lazy_host = nullptr; uint32_t lazy_maxlen = 0; uint64_t lazy_fetched = 0;   // the batch's samples[] as the device sees it; its longest read
	DevBuf st[24];
	DevBuf act[2], n_act_dev;
	DevBuf zbuf, t1buf, t2buf, n_norm, peaks, n_peaks;
	DevBuf ev, n_ev, skip, sx, sy, n_seed, m_val, m_n, m_meta, m_pref, n_match, n_new, rep_len, a_off;
	DevBuf anc, raw_anc, zs, n_z, need_exact, need_exact2, prev_stage, u, n_u, n_v, ws, counters, rec;
	DevBuf events, dtw_ws, dtw_n, dtw_off, dtw_rec, dtw_dec;       // RH_M_DTW_EVALUATE_CHAINS: reads' events, DP buffers, per-region values for the host's MAPQ, its decisions
	DevBuf name_rank, t_rank, rec_off;                            // all-vs-all: name ranks of the reads / of the targets, record offsets
	uint64_t arena_room = 0;                                       // anchors the per-anchor arenas were sized for when a round was last cut into slices (the budget sticks to it)
	uint32_t ws_stride = RH_WS_PER_ANCHOR;                         // bytes of per-anchor scratch of the current batch (doubled when min_num_anchors < 2)
	uint32_t cs_stride = 2;                                        // chunk boundaries kept per read of the current batch: max_num_chunk + 1
	uint32_t ev_row = RH_CHUNK_MAX + 64, ev_cap = RH_EV_CAP, whole = 0;   // strides of the per-read rows of the current batch (whole-read rounds: sized by its longest read)
	DevBuf carry[2], carry_off, a_off_slice;                      // chained anchors carried into the next chunk, dense, ping-pong over the rounds
	int share = 1;                                                 // sub-batches running concurrently on this device (memory budget per context)
	int flight_mult = 1;                                           // batches in flight that share the device with this context's
	DevBuf sort_ws;                                               // multi-workgroup segment sorter: tables (only when a read exceeds the LDS classes)
	DevBuf sy_samples, sy_off, sy_cal_off, sy_cal_scale, sy_levels;
	// timing
	hipEvent_t e0 = nullptr, e1 = nullptr;
	std::vector<hipEvent_t> ev_pool; std::vector<int> ev_stage; size_t ev_used = 0;   // stage timers of the batch in flight
	uint64_t *pin = nullptr;                                       // pinned host words for the per-round read-backs
	rh_map_stats_t stats{};
	// batches in flight (rh_map_submit / rh_map_wait): a slot = a borrowed context + the thread mapping on it
	struct Flight { rh_ctx *ctx = nullptr; std::thread th; bool busy = false; uint32_t serial = 0; int rc = 0; uint64_t n_out = 0; std::string err; };
	Flight flight[RH_MAX_IN_FLIGHT];
	// concurrent sub-batches: extra contexts (own stream + arenas) that borrow this context's index and tables
	std::vector<rh_ctx*> subs;
	int n_sub = 1;
	bool is_sub = false;
	uint32_t slice_hint = 0;                                       // reads per slice that fitted the device last time (0 = whole batches fit)
	size_t mem_allow = 0;                                          // bytes of device memory this (sub-batch) context may hold for a batch: an equal part of what rh_map_batch found free for all of them (0: on its own)
};

namespace {

// Stage timing without stalling the stream: start/stop events come from a per-context pool, are only recorded here, and
// are read back by stage_timers_collect() after the batch's final synchronisation.
struct StageTimer {
	rh_ctx *c; int stage; size_t slot;
	StageTimer(rh_ctx *ctx, int st) : c(ctx), stage(st)
	{
		slot = c->ev_used;
		while (c->ev_pool.size() < slot + 2) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) break; c->ev_pool.push_back(e); }
		if (c->ev_pool.size() >= slot + 2) { c->ev_used += 2; c->ev_stage.push_back(stage); (void)hipEventRecord(c->ev_pool[slot], c->stream); }
		else slot = (size_t)-1;
	}
	~StageTimer() { if (slot != (size_t)-1) (void)hipEventRecord(c->ev_pool[slot + 1], c->stream); }
};

void stage_timers_collect(rh_ctx *c)
{
	for (size_t k = 0; k < c->ev_stage.size(); ++k) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, c->ev_pool[2 * k], c->ev_pool[2 * k + 1]) == hipSuccess) c->stats.ms_kernel[c->ev_stage[k]] += ms;
		c->stats.n_launch[c->ev_stage[k]] += 1;
	}
	c->ev_used = 0; c->ev_stage.clear();
}

// the chaining of a round as ri_map_frag runs it (rmap.cpp:317-342): DP or RMQ, candidates sorted, backtracked and compacted; then,
// if bw_long > bw, the chained anchors are chained AGAIN by the RMQ variant with the long bandwidth (same backtrack / compaction)
static int chain_stages(rh_ctx *c, hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &rr, bool timed);

int fill_dev_opt(rh_ctx *c, const rh_mapopt_t *mo, rh_dev_opt *o)
{
	c->ws_stride = mo->min_num_anchors < 2 ? 2 * RH_WS_PER_ANCHOR : RH_WS_PER_ANCHOR;   // (chains of one anchor: the region stage keeps 128 B per chain)
	if (mo->chunk_size == 0 || mo->chunk_size >= (1u << 26)) { rh_set_error("chunk_size %u not supported on the device (1..2^26-1)", mo->chunk_size); return -1; }
	if (!(mo->flag & RH_M_NO_ADAPTIVE) && mo->chunk_size > RH_CHUNK_MAX && (mo->window_length1 > 15 || mo->window_length2 > 15)) { rh_set_error("chunks of more than %d samples need segmentation windows <= 15", RH_CHUNK_MAX); return -1; }
	if (mo->max_num_chunk > (1u << 16)) { rh_set_error("max_num_chunk %u > 65536 not supported", mo->max_num_chunk); return -1; }
	if ((mo->flag & RH_M_ALL_CHAINS) && !(mo->flag & RH_M_NO_ADAPTIVE)) { rh_set_error("all-chains output is built for whole-read rounds only (RH_M_ALL_CHAINS needs RH_M_NO_ADAPTIVE, as in the ava presets)"); return -1; }
	if ((mo->flag & RH_M_NO_ADAPTIVE) && (mo->window_length1 > 15 || mo->window_length2 > 15)) { rh_set_error("whole-read rounds need segmentation windows <= 15"); return -1; }
	if ((mo->flag & RH_M_ALL_CHAINS) && c->have_index && !c->dix.t_rank) { rh_set_error("all-vs-all mapping needs the name ranks of the targets (rh_index_set_target_ranks)"); return -1; }
	if (mo->flag & RH_M_DTW_EVALUATE_CHAINS) {
		if (mo->flag & RH_M_ALL_CHAINS) { rh_set_error("DTW re-scoring with all-chains output is not supported on the device"); return -1; }
		if (c->have_index && !c->dix.sig) { rh_set_error("DTW re-scoring needs the targets' signals: build the index with RH_I_STORE_SIG (--store-sig) and upload it (rh_index_upload)"); return -1; }
		if (mo->dtw_border_constraint > 1u || mo->dtw_fill_method > 1u) { rh_set_error("DTW border constraint %u / fill method %u not supported (global | sparse, full | banded)", mo->dtw_border_constraint, mo->dtw_fill_method); return -1; }
	}
	if (mo->min_num_anchors < 1) { rh_set_error("min_num_anchors %d < 1", mo->min_num_anchors); return -1; }
	if ((mo->flag & RH_M_ALL_CHAINS) && mo->min_num_anchors < 2) { rh_set_error("all-chains output with chains of one anchor (min_num_anchors %d) is not supported: a read's reported chains are staged in one 16-byte slot per anchor, two words per chain", mo->min_num_anchors); return -1; }
	if (mo->window_length1 > 64 || mo->window_length2 > 64) { rh_set_error("segmentation windows > 64 not supported"); return -1; }
	memset(o, 0, sizeof(*o));
	o->chunk_size = mo->chunk_size; o->max_num_chunk = mo->max_num_chunk; o->min_events = mo->min_events;
	if (mo->flag & RH_M_NO_ADAPTIVE) { o->chunk_size = 1u << 30; o->max_num_chunk = 1; }   // rmap.cpp:404-405: one round over the whole read
	o->w1 = mo->window_length1; o->w2 = mo->window_length2; o->thr1 = mo->threshold1; o->thr2 = mo->threshold2; o->peak_height = mo->peak_height;
	o->mid_occ = mo->mid_occ;
	o->max_dist_t = mo->max_target_gap_length; o->max_dist_q = mo->max_query_gap_length; o->bw = mo->bw;
	o->max_skip = mo->max_num_skips; o->max_iter = mo->max_chain_iter; o->min_cnt = mo->min_num_anchors;
	o->bw_long = mo->bw_long; o->rmq_inner_dist = mo->rmq_inner_dist; o->rmq_size_cap = mo->rmq_size_cap;
	o->dtw_border = mo->dtw_border_constraint; o->dtw_fill = mo->dtw_fill_method; o->dtw_band_frac = mo->dtw_band_radius_frac; o->dtw_match_bonus = mo->dtw_match_bonus; o->dtw_min_score = mo->dtw_min_score;
	o->min_sc = mo->min_chaining_score; o->min_sc2 = mo->min_chaining_score2;
	if (c->have_index) {	// rmap.cpp:318: computed in double, narrowed to float
		const int span = c->dix.sp.e + c->dix.sp.k - 1;
		o->pen_gap = (float)((double)mo->chain_gap_scale * 0.01 * span);
		o->pen_skip = (float)((double)mo->chain_skip_scale * 0.01 * span);
		o->sig_target = (c->dix.flag & RH_I_SIG_TARGET) ? 1 : 0;
	}
	o->mask_level = mo->mask_level; o->mask_len = mo->mask_len; o->pri_ratio = mo->pri_ratio; o->best_n = mo->best_n;
	o->min_strand_sc = (int32_t)(mo->max_target_gap_length * 0.8);   // rmap.cpp:354
	o->w_bestq = mo->w_bestq; o->w_bestmq = mo->w_bestmq; o->w_bestmc = mo->w_bestmc; o->w_threshold = mo->w_threshold; o->w_bestma = mo->w_bestma;
	o->min_mapq = mo->min_mapq; o->sample_per_base = mo->sample_per_base; o->flag = mo->flag;
	return 0;
}

// RH_M_DTW_EVALUATE_CHAINS, the region stage of a slice of n active reads: regions + alignment scores on the device (k_regions_dtw), then
// mm_set_mapq's DTW branch (hit.c:502-539: logf of the fractional alignment score - the host's libm, like the reference) and the mapping
// decision of rmap.cpp:423-500 on the host, and the verdict committed on the device.
static int dtw_regions_stage(rh_ctx *c, hipStream_t s, const rh_dev_opt &o, const rh_mapopt_t *mo, const rh_dev_reads &rd, rh_dev_round rs, uint32_t n, uint32_t chunk)
{
	const uint64_t ev_now = (uint64_t)(chunk + 1) * rs.ev_cap < rd.ev_stride ? (uint64_t)(chunk + 1) * rs.ev_cap : rd.ev_stride;
	rs.dtw_stride = (uint32_t)(4 * ev_now + 64);                    // full matrix: one row of the read's events; bands: 3 x (2 x band + 3) with band <= frac x events
	if (c->dtw_n.ensure((size_t)n * 4) || c->dtw_ws.ensure((size_t)n * rs.dtw_stride * 4) || c->dtw_off.ensure((size_t)(n + 1) * 8) || c->dtw_dec.ensure((size_t)n * 12)) return -1;
	rs.dtw_n = c->dtw_n.as<uint32_t>(); rs.dtw_ws = c->dtw_ws.as<float>();
	rhk_regions_dtw(s, o, c->dix, rd, rs);
	// MAPQ and the decision on the device wherever the host's logf cannot change the truncated MAPQ (k_dtw_decide); what is left - a read in ten thousand - goes
	// the old way below.  RH_DTW_HOST_MAPQ=1: the host for every read (A/B and test aid).
	const bool host_all = getenv("RH_DTW_HOST_MAPQ") != nullptr;    // (read per call)
	uint32_t *n_host_d = c->n_act_dev.as<uint32_t>() + 12;
	if (!host_all) {
		RH_HIP(hipMemsetAsync(n_host_d, 0, 4, s));
		rhk_dtw_decide(s, o, rd, rs, c->logf_tab.as<float>(), n_host_d);
		RH_HIP(hipMemcpyAsync(c->pin + 7, n_host_d, 4, hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		const uint32_t n_host = (uint32_t)c->pin[7];
		c->dtw_host_reads += n_host; c->dtw_dev_reads += n - n_host;
		if (!n_host) return 0;
	} else c->dtw_host_reads += n;
	std::vector<uint32_t> nreg(n);
	std::vector<int32_t> rep(n);
	RH_HIP(hipMemcpyAsync(nreg.data(), rs.dtw_n, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	RH_HIP(hipMemcpyAsync(rep.data(), rs.rep_len, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	RH_HIP(hipStreamSynchronize(s));
	std::vector<uint64_t> off((size_t)n + 1, 0);
	for (uint32_t a = 0; a < n; ++a) {
		if (nreg[a] & 0x80000000u) { rh_set_error("DTW re-scoring: a band / matrix row does not fit the per-read DP buffer (dtw_band_radius_frac %.2f too large for this device path)", (double)mo->dtw_band_radius_frac); return -1; }
		if (nreg[a] & 0x40000000u) nreg[a] = 0;                         // decided and committed on the device
		off[a + 1] = off[a] + nreg[a];
	}
	const uint64_t T = off[n];
	std::vector<int32_t> recs((size_t)(T ? T : 1) * 8);
	if (T) {
		if (c->dtw_rec.ensure((size_t)T * 32)) return -1;
		RH_HIP(hipMemcpyAsync(c->dtw_off.p, off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, s));
		rs.dtw_off = c->dtw_off.as<uint64_t>(); rs.dtw_rec = c->dtw_rec.as<float>();
		rhk_dtw_pack(s, rs);
		RH_HIP(hipMemcpyAsync(recs.data(), c->dtw_rec.p, (size_t)T * 32, hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
	}
	std::vector<int32_t> dec((size_t)n * 3, 0);
	std::vector<int32_t> mq;
	for (uint32_t a = 0; a < n; ++a) {
		const int32_t nr = (int32_t)nreg[a];
		if (!nr) continue;
		const int32_t *R8 = recs.data() + off[a] * 8;                  // per region: score, cnt, subsc, score0, n_sub, is-primary, alignment score (float bits)
		auto ascore = [&](int32_t i) { float f; memcpy(&f, &R8[(size_t)i * 8 + 6], 4); return f; };
		// mm_set_mapq, is_dtw = 1 (hit.c:502-539)
		int64_t sum_sc = 0;
		for (int32_t i = 0; i < nr; ++i) if (R8[(size_t)i * 8 + 5]) sum_sc += R8[(size_t)i * 8];
		const float uniq_ratio = (float)sum_sc / (sum_sc + rep[a]);
		mq.assign((size_t)nr, 0);
		for (int32_t i = 0; i < nr; ++i) {
			const int32_t score = R8[(size_t)i * 8], cnt = R8[(size_t)i * 8 + 1], subsc0 = R8[(size_t)i * 8 + 2], score0 = R8[(size_t)i * 8 + 3], n_sub = R8[(size_t)i * 8 + 4];
			int mapq = 0;
			float pen_s1 = (score > 100 ? 1.0f : 0.01 * score) * uniq_ratio;
			float pen_cm = cnt > 10 ? 1.0f : 0.1f * cnt;
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			const int subsc = subsc0 > mo->min_chaining_score ? subsc0 : mo->min_chaining_score;
			float x = (float)subsc / score0;
			if (ascore(i) > 0) mapq = (int)(pen_cm * 40.0f * (1.0f - x) * 2 * logf(ascore(i)));
			mapq -= (int)(4.343f * logf(n_sub + 1) + .499f);
			mapq = mapq > 0 ? mapq : 0;
			mq[i] = mapq < 60 ? mapq : 60;
		}
		// the decision (rmap.cpp:423-500, one chain reported: no all-chains mode here)
		int sel = 0, stop = 0;
		if (nr == 1 && (mq[0] >= mo->min_mapq || ascore(0) >= mo->dtw_min_score)) stop = 1;
		else {
			float meanC = 0, meanQ = 0;
			for (int32_t i = 0; i < nr; ++i) { meanC += R8[(size_t)i * 8]; meanQ += mq[i]; }
			meanC /= nr; meanQ /= nr;
			float bestA = ascore(0);
			int best = 0;
			for (int32_t i = 1; i < nr; ++i) if (ascore(i) > bestA) { bestA = ascore(i); best = i; }
			const float bestQ = mq[best], bestC = R8[(size_t)best * 8];
			float weighted = 0.0f;
			if (bestA >= mo->dtw_min_score) {
				float r_bestma = (bestA > 0) ? (bestA / 50.0f) : 0.0f; if (r_bestma < 0) r_bestma = 0.0f;
				float r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
				float r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
				weighted = mo->w_bestma * r_bestma + mo->w_bestmq * r_bestmq + mo->w_bestmc * r_bestmc;
			}
			if (weighted >= mo->w_threshold) { stop = 1; sel = best; }
		}
		dec[(size_t)a * 3] = sel; dec[(size_t)a * 3 + 1] = mq[sel]; dec[(size_t)a * 3 + 2] = stop;
	}
	RH_HIP(hipMemcpyAsync(c->dtw_dec.p, dec.data(), (size_t)n * 12, hipMemcpyHostToDevice, s));
	rs.dtw_dec = c->dtw_dec.as<int32_t>();
	rhk_dtw_commit(s, o, rd, rs);
	RH_HIP(hipStreamSynchronize(s));                                  // (dec is a stack vector)
	return 0;
}

static int chain_stages(rh_ctx *c, hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &rr, bool timed)
{
	const int32_t max_gap = o.max_dist_t > o.max_dist_q ? o.max_dist_t : o.max_dist_q;
	struct Opt { rh_ctx *c; bool on; StageTimer *t = nullptr; Opt(rh_ctx *c_, bool on_, int st) : c(c_), on(on_) { if (on) t = new StageTimer(c, st); } ~Opt() { delete t; } };
	for (int pass = 0; pass < 2; ++pass) {
		rh_dev_opt op = o;
		if (pass == 1) { if (o.bw_long <= o.bw) break; op.bw = o.bw_long; }   // (max_drop of the backtrack = the bandwidth of the pass, lchain.c:625)
		{
			Opt t(c, timed, ST_CHAIN);
			if (pass == 1) rhk_chain_rmq(s, op, rr, rr.n_v, max_gap, o.rmq_inner_dist, o.rmq_size_cap);   // the chained anchors of the first pass, in place
			else if (o.flag & RH_M_RMQ) rhk_chain_rmq(s, op, rr, nullptr, max_gap, o.rmq_inner_dist, o.rmq_size_cap);
			else rhk_chain(s, op, rr);
		}
		{ Opt t(c, timed, ST_ZSORT); if (rhk_zsort(s, op, rr)) return -1; }
		{ Opt t(c, timed, ST_BACKTRACK); if (rhk_backtrack(s, op, rd, rr)) return -1; }
	}
	return 0;
}

// upload (or adopt) the read batch; fills rd with device pointers and allocates the per-read state
int stage_reads(rh_ctx *c, const rh_read_batch_t *in, rh_dev_reads *rd, bool allow_lazy = false)
{
	const uint32_t R = in->n_reads;
	memset(rd, 0, sizeof(*rd));
	rd->n_reads = R;
	rd->fast5 = in->fast5_ingest ? 1u : 0u;
	c->lazy_host = nullptr;
	if (in->samples_on_device) {
		if (!in->cal_offset || !in->cal_scale) { rh_set_error("device batches must carry cal_offset and cal_scale"); return -1; }
		rd->raw = in->samples; rd->off = in->offsets; rd->cal_off = in->cal_offset; rd->cal_scale = in->cal_scale;
	} else {
		const uint64_t first = R ? in->offsets[0] : 0, total = R ? in->offsets[R] - first : 0;   // a slice keeps absolute offsets
		if (c->raw.ensure(total * 2 + 32) || c->off.ensure((size_t)(R + 1) * 8) || c->cal_off.ensure((size_t)(R + 1) * 8) || c->cal_scale.ensure((size_t)(R + 1) * 4)) return -1;
		// Consumed-prefix staging: the caller knows every read's filtered length (its reader counted while decoding) and the samples lie in
		// page-locked memory the device can read -> nothing is copied here; the rounds fetch the stretches they consume (ensure_resident).
		size_t place = 0;
		if (allow_lazy && in->n_filtered && R) {
			if (c->lsig_given.ensure((size_t)R * 4)) return -1;
			RH_HIP(hipMemcpyAsync(c->lsig_given.p, in->n_filtered, (size_t)R * 4, hipMemcpyHostToDevice, c->stream));
			rd->l_sig_given = c->lsig_given.as<uint32_t>();            // (checked against the filter's own count wherever the device sees a whole read)
			hipPointerAttribute_t at;
			if (total && hipPointerGetAttributes(&at, in->samples) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
				c->lazy_host = (const int16_t*)at.devicePointer;
				place = (size_t)((uintptr_t)(c->lazy_host + first) & 15u);   // same misalignment on both sides: k_fetch copies aligned 16-byte words
				uint64_t mx = 0;
				for (uint32_t r = 0; r < R; ++r) { const uint64_t l = in->offsets[r + 1] - in->offsets[r]; if (l > mx) mx = l; }
				c->lazy_maxlen = (uint32_t)mx; c->lazy_fetched = 0;
				if (c->res_len.ensure((size_t)R * 4) || c->cnt_res.ensure((size_t)R * 4) || c->new_len.ensure((size_t)R * 4)) return -1;
				RH_HIP(hipMemsetAsync(c->res_len.p, 0, (size_t)R * 4, c->stream)); RH_HIP(hipMemsetAsync(c->cnt_res.p, 0, (size_t)R * 4, c->stream));
				rd->res_len = c->res_len.as<uint32_t>(); rd->cnt_res = c->cnt_res.as<uint32_t>();
			} else (void)hipGetLastError();                            // (pageable memory: the whole batch is copied, as without n_filtered)
		}
		int16_t *const d0 = (int16_t*)((char*)c->raw.p + place);
		if (total && !c->lazy_host) RH_HIP(hipMemcpyAsync(d0, in->samples + first, total * 2, hipMemcpyHostToDevice, c->stream));
		RH_HIP(hipMemcpyAsync(c->off.p, in->offsets, (size_t)(R + 1) * 8, hipMemcpyHostToDevice, c->stream));
		std::vector<double> co(R, 0.0); std::vector<float> cs(R, 1.0f);
		if (in->cal_offset) memcpy(co.data(), in->cal_offset, (size_t)R * 8);
		if (in->cal_scale) memcpy(cs.data(), in->cal_scale, (size_t)R * 4);
		if (R) {
			RH_HIP(hipMemcpyAsync(c->cal_off.p, co.data(), (size_t)R * 8, hipMemcpyHostToDevice, c->stream));
			RH_HIP(hipMemcpyAsync(c->cal_scale.p, cs.data(), (size_t)R * 4, hipMemcpyHostToDevice, c->stream));
		}
		RH_HIP(hipStreamSynchronize(c->stream));   // co/cs are stack-owned
		rd->raw = d0 - first; rd->raw_w = d0 - first; rd->off = c->off.as<uint64_t>(); rd->cal_off = c->cal_off.as<double>(); rd->cal_scale = c->cal_scale.as<float>();
	}
	if (in->name_rank) {
		if (c->name_rank.ensure((size_t)(R + 1) * 4)) return -1;
		if (R) RH_HIP(hipMemcpyAsync(c->name_rank.p, in->name_rank, (size_t)R * 4, hipMemcpyHostToDevice, c->stream));
		rd->name_rank = c->name_rank.as<uint32_t>();
	}
	const size_t n = R ? R : 1;
	size_t k = 0;
	auto U32 = [&](uint32_t *&p, size_t cnt) { if (c->st[k].ensure(cnt * 4)) return -1; p = c->st[k].as<uint32_t>(); ++k; return 0; };
	auto I32 = [&](int32_t *&p, size_t cnt) { if (c->st[k].ensure(cnt * 4)) return -1; p = c->st[k].as<int32_t>(); ++k; return 0; };
	if (U32(rd->l_sig, n) || U32(rd->chunk_start, n * (size_t)c->cs_stride) || U32(rd->n_sum, n) || U32(rd->ev_off, n) || U32(rd->n_prev, n) || U32(rd->stop_chunk, n)) return -1;
	if (c->st[k].ensure(n * 8)) return -1; rd->sum = c->st[k++].as<double>();
	if (c->st[k].ensure(n * 8)) return -1; rd->sum2 = c->st[k++].as<double>();
	if (c->st[k].ensure(n * 8)) return -1; rd->prev_off = c->st[k++].as<uint64_t>();
	if (c->st[k].ensure(n)) return -1; rd->done = c->st[k++].as<uint8_t>();
	rd->cs_stride = c->cs_stride;
	if (I32(rd->ls_ncregs, n) || I32(rd->ls_cnt, n) || I32(rd->ls_score, n) || I32(rd->ls_mapq, n) || I32(rd->ls_qs, n) || I32(rd->ls_qe, n) ||
	    I32(rd->ls_rs, n) || I32(rd->ls_re, n) || I32(rd->ls_rid, n) || I32(rd->ls_rev, n)) return -1;
	return 0;
}

// per-round arrays for n_act active reads
int stage_round(rh_ctx *c, uint32_t n_act, rh_dev_round *rr)
{
	const size_t n = n_act ? n_act : 1, cap = (size_t)n * c->ev_cap;
	const size_t rowb = (size_t)(n + 64) * c->ev_row * 4;           // + 64 rows: the peak kernel reads whole 64-read tiles
	rr->ev_row = c->ev_row; rr->ev_cap = c->ev_cap; rr->whole = c->whole;
	if (c->zbuf.ensure(rowb) || c->t1buf.ensure(rowb) || c->t2buf.ensure(rowb) || c->n_norm.ensure(n * 4) || c->peaks.ensure(cap * (c->whole ? 4 : 2)) || c->n_peaks.ensure(n * 4)) return -1;
	rr->zbuf = c->zbuf.as<float>(); rr->t1buf = c->t1buf.as<float>(); rr->t2buf = c->t2buf.as<float>(); rr->n_norm = c->n_norm.as<uint32_t>();
	rr->peaks = c->peaks.as<uint16_t>(); rr->n_peaks = c->n_peaks.as<uint32_t>();
	if (c->ev.ensure(cap * 4) || c->n_ev.ensure(n * 4) || c->skip.ensure(n) || c->sx.ensure(cap * 8) || c->sy.ensure(cap * 8) || c->n_seed.ensure(n * 4) ||
	    c->m_val.ensure(cap * 8) || c->m_n.ensure(cap * 4) || c->m_meta.ensure(cap * 4) || c->m_pref.ensure((size_t)n * (c->ev_cap + 1) * 4) ||
	    c->n_match.ensure(n * 4) || c->n_new.ensure(n * 4) || c->rep_len.ensure(n * 4) || c->a_off.ensure((n + 2) * 8) || c->n_u.ensure(n * 4) || c->n_v.ensure(n * 4) ||
	    c->counters.ensure(16 * 8) || c->need_exact.ensure(n) || c->need_exact2.ensure(n) || c->n_z.ensure(n * 4)) return -1;
	rr->n_z = c->n_z.as<uint32_t>();
	rr->n_act = n_act;
	rr->need_exact = c->need_exact.as<uint8_t>(); rr->need_exact2 = c->need_exact2.as<uint8_t>();
	rr->ev = c->ev.as<float>(); rr->n_ev = c->n_ev.as<uint32_t>(); rr->skip = c->skip.as<uint8_t>();
	rr->sx = c->sx.as<uint64_t>(); rr->sy = c->sy.as<uint64_t>(); rr->n_seed = c->n_seed.as<uint32_t>();
	rr->m_val = c->m_val.as<uint64_t>(); rr->m_n = c->m_n.as<uint32_t>(); rr->m_meta = c->m_meta.as<uint32_t>(); rr->m_pref = c->m_pref.as<uint32_t>();
	rr->n_match = c->n_match.as<uint32_t>(); rr->n_new = c->n_new.as<uint32_t>(); rr->rep_len = c->rep_len.as<int32_t>();
	rr->a_off = c->a_off.as<uint64_t>(); rr->n_u = c->n_u.as<uint32_t>(); rr->n_v = c->n_v.as<uint32_t>();
	rr->counters = c->counters.as<uint64_t>();
	return 0;
}

// anchor-sized arenas for a slice of `total` anchors; a round that is cut into slices sizes them for `room` anchors (the slice
// budget) once, so that the slices do not re-allocate tens of gigabytes each
int stage_anchors(rh_ctx *c, uint64_t total, rh_dev_round *rr, uint64_t room = 0)
{
	const size_t t = total > room ? (total ? total : 1) : room;
	const bool mg = room == 0;                                     // (budget-sized arenas: no growth margin)
	if (c->anc.ensure(t * 16, mg) || c->raw_anc.ensure(t * 16, mg) || c->zs.ensure(t * 16, mg) || c->prev_stage.ensure(t * 16, mg) || c->u.ensure(t * 8, mg) ||
	    c->ws.ensure(t * c->ws_stride + 4096, mg)) return -1;
	rr->anc = c->anc.as<rh_mm128_t>(); rr->raw = c->raw_anc.as<rh_mm128_t>(); rr->zs = c->zs.as<rh_mm128_t>(); rr->prev_out = c->prev_stage.as<rh_mm128_t>();
	rr->u = c->u.as<uint64_t>(); rr->ws = c->ws.as<unsigned char>(); rr->ws_stride = c->ws_stride;
	rr->arena_n = t;
	// segments longer than the LDS sort classes (large indexes): scratch of the multi-workgroup sorter
	if (rr->max_anchors == 0 || rr->max_anchors > (uint32_t)RH_SORT_LDS_MIN_TOP) {
		const size_t wsb = rhk_bigsort_ws_bytes(t, (uint32_t)RH_SORT_LDS_MIN_TOP);
		if (c->sort_ws.ensure(wsb, mg)) return -1;
		if (!c->pin) RH_HIP(hipHostMalloc((void**)&c->pin, 256, 0));
		rr->sort_ws = c->sort_ws.as<unsigned char>(); rr->sort_ws_bytes = c->sort_ws.cap; rr->sort_pin = c->pin + 16; rr->sort_total = t;
	}
	return 0;
}
// bytes of device memory a slice needs per anchor (the arrays above + the sorter's tables)
size_t bytes_per_anchor(const rh_ctx *c) { return 16 * 4 + 8 + (size_t)c->ws_stride + 23; }

// anchors one slice of a round may hold: what is free on the device (plus what this context's arenas hold already), shared
// by the sub-batches running concurrently; RH_ARENA_MAX_BYTES caps the per-anchor scratch (shared devices, tests)
size_t ctx_bytes_held(rh_ctx *c);
uint64_t slice_budget(rh_ctx *c)
{
	size_t free_b = 0, total_b = 0;
	uint64_t budget = ~0ull;
	if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
		DevBuf *mine[] = {&c->anc, &c->raw_anc, &c->zs, &c->prev_stage, &c->u, &c->ws, &c->sort_ws};
		size_t held = 0;
		for (DevBuf *d : mine) held += d->cap;
		const size_t reserve = total_b / 24 > ((size_t)3 << 30) ? total_b / 24 : ((size_t)3 << 30);
		const size_t avail = free_b > reserve ? free_b - reserve : 0;
		// three quarters of what this context may use (its share of the free memory + what its arenas hold already; the rest is
		// for the dense carry buffers and the per-read arrays), but never less than the arenas hold: they stay as they are
		// A sub-batch context has an allowance - its equal part of what rh_map_batch found for all of them (whoever asked first used to get
		// three times the arenas of whoever asked last, and the call waited for the slowest) -: three quarters of what its other buffers (signal,
		// per-read rows: up to a third, map_batch_single) leave of it go to the anchor arenas, the rest is for the dense carry buffers, which grow with the rounds.
		const size_t other = c->mem_allow ? ctx_bytes_held(c) - held : 0;   // what this context holds besides the anchor arenas: the batch's signal, the per-read rows, the carry buffers
		const double may_use = c->mem_allow ? (c->mem_allow > other ? (double)(c->mem_allow - other) : 0.0) : (double)(avail / (size_t)(c->share > 0 ? c->share : 1)) + (double)held;
		const double use = 0.75 * may_use > (double)held ? 0.75 * may_use : (double)held;
		budget = (uint64_t)(use / (double)bytes_per_anchor(c));
		if (budget < (1u << 16)) return 0;                          // the device is full (other contexts / processes hold it)
		// The arenas were sized for a budget once: a slightly larger one (the free memory moves by rounding and by what the other
		// sub-batches hold at the moment) must not re-allocate tens of gigabytes on a nearly full device - that takes seconds.
		if (c->arena_room && held && budget > c->arena_room && budget < c->arena_room + c->arena_room / 2) budget = c->arena_room;
	}
	if (const char *cap_env = getenv("RH_ARENA_MAX_BYTES")) { const uint64_t m = strtoull(cap_env, nullptr, 10) / RH_WS_PER_ANCHOR; if (m < budget) budget = m; }
	return budget > 1024 ? budget : 1024;
}

// the view of a round's per-slot arrays for the active reads [lo, lo + n)
rh_dev_round slice_view(const rh_dev_round &rr, uint32_t lo, uint32_t n, uint64_t *a_off)
{
	rh_dev_round v = rr;
	const size_t cap = rr.ev_cap;
	v.n_act = n; v.act = rr.act + lo; v.a_off = a_off;
	v.n_norm = rr.n_norm + lo; v.peaks = rr.peaks + lo * cap * (rr.whole ? 2 : 1); v.n_peaks = rr.n_peaks + lo;
	v.ev = rr.ev + lo * cap; v.n_ev = rr.n_ev + lo; v.skip = rr.skip + lo;
	v.sx = rr.sx + lo * cap; v.sy = rr.sy + lo * cap; v.n_seed = rr.n_seed + lo;
	v.m_val = rr.m_val + lo * cap; v.m_n = rr.m_n + lo * cap; v.m_meta = rr.m_meta + lo * cap; v.m_pref = rr.m_pref + (size_t)lo * (cap + 1);
	v.n_match = rr.n_match + lo; v.n_new = rr.n_new + lo; v.rep_len = rr.rep_len + lo;
	v.need_exact = rr.need_exact + lo; v.need_exact2 = rr.need_exact2 + lo;
	v.n_u = rr.n_u + lo; v.n_v = rr.n_v + lo; v.n_z = rr.n_z + lo;
	return v;
}

// device memory of the per-batch arenas of a context and of its sub-batch contexts (the resident index stays)
void release_arenas(rh_ctx *c)
{
	(void)hipStreamSynchronize(c->stream);
	DevBuf *all[] = {&c->raw, &c->off, &c->cal_off, &c->cal_scale, &c->act[0], &c->act[1], &c->zbuf, &c->t1buf, &c->t2buf, &c->n_norm, &c->peaks, &c->n_peaks, &c->ev, &c->n_ev, &c->skip, &c->sx, &c->sy,
	                 &c->n_seed, &c->m_val, &c->m_n, &c->m_meta, &c->m_pref, &c->n_match, &c->n_new, &c->rep_len, &c->a_off, &c->anc, &c->raw_anc, &c->zs, &c->n_z, &c->need_exact, &c->need_exact2, &c->prev_stage,
	                 &c->carry[0], &c->carry[1], &c->carry_off, &c->a_off_slice, &c->u, &c->n_u, &c->n_v, &c->ws, &c->sort_ws, &c->rec,
	                 &c->events, &c->dtw_ws, &c->dtw_n, &c->dtw_off, &c->dtw_rec, &c->dtw_dec};
	for (DevBuf *b : all) b->release();
	for (DevBuf &b : c->st) b.release();
	c->arena_room = 0;
	for (rh_ctx *sc : c->subs) release_arenas(sc);
}
// device bytes of the per-batch buffers of ONE context (not its sub-batch contexts)
size_t ctx_bytes_held(rh_ctx *c)
{
	DevBuf *all[] = {&c->raw, &c->off, &c->cal_off, &c->cal_scale, &c->act[0], &c->act[1], &c->zbuf, &c->t1buf, &c->t2buf, &c->n_norm, &c->peaks, &c->n_peaks, &c->ev, &c->n_ev, &c->skip, &c->sx, &c->sy,
	                 &c->n_seed, &c->m_val, &c->m_n, &c->m_meta, &c->m_pref, &c->n_match, &c->n_new, &c->rep_len, &c->a_off, &c->anc, &c->raw_anc, &c->zs, &c->n_z, &c->need_exact, &c->need_exact2, &c->prev_stage,
	                 &c->carry[0], &c->carry[1], &c->carry_off, &c->a_off_slice, &c->u, &c->n_u, &c->n_v, &c->ws, &c->sort_ws, &c->rec,
	                 &c->events, &c->dtw_ws, &c->dtw_n, &c->dtw_off, &c->dtw_rec, &c->dtw_dec, &c->res_len, &c->cnt_res, &c->new_len, &c->lsig_given};
	size_t b = 0;
	for (DevBuf *d : all) b += d->cap;
	for (DevBuf &d : c->st) b += d.cap;
	return b;
}
bool holds_arenas(const rh_ctx *c)
{
	if (c->anc.cap || c->zbuf.cap || c->raw.cap) return true;
	for (const rh_ctx *sc : c->subs) if (holds_arenas(sc)) return true;
	return false;
}

int need_index(rh_ctx *c) { if (!c->have_index) { rh_set_error("no index resident on this context (rh_index_upload first)"); return -1; } return 0; }

} // namespace

// =================================================================================================== context
extern "C" int rh_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" int rh_ctx_create(rh_ctx **out, int device_id)
{
	*out = nullptr;
	const int n = rh_device_count();
	if (n <= 0) { rh_set_error("no HIP device visible: the mapping path has no CPU fallback"); return -1; }
	if (device_id < 0 || device_id >= n) { rh_set_error("device %d out of range (%d visible)", device_id, n); return -1; }
	RH_HIP(hipSetDevice(device_id));
	struct CtxDel { void operator()(rh_ctx *p) const { rh_ctx_destroy(p); } };   // failure paths release stream / events / buffers too
	std::unique_ptr<rh_ctx, CtxDel> c(new rh_ctx());
	c->device = device_id;
	RH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	RH_HIP(hipEventCreate(&c->e0));
	RH_HIP(hipEventCreate(&c->e1));
	// logf() of small integers from the HOST libm: hit.c:525-533 feeds integer scores to logf and truncates, so MAPQ
	// parity needs the host's roundings (device logf may differ in the last ulp)
	std::vector<float> tab(RH_LOGF_N);
	for (uint32_t i = 0; i < RH_LOGF_N; ++i) tab[i] = logf((float)i);
	if (c->logf_tab.ensure((size_t)RH_LOGF_N * 4)) return -1;
	RH_HIP(hipMemcpy(c->logf_tab.p, tab.data(), (size_t)RH_LOGF_N * 4, hipMemcpyHostToDevice));
	if (const char *e = getenv("RH_SUB_BATCHES")) c->n_sub = atoi(e) > 0 ? atoi(e) : 1; else c->n_sub = 3;   // measured on MI355X (100 k reads): 1: 585 k, 2: 656 k, 3: 690 k, 4: 517 k (636 k with GPU_MAX_HW_QUEUES=8: HIP multiplexes streams onto 4 hardware queues by default), 6: 544 k (653 k) reads/s
	*out = c.release();
	return 0;
}

extern "C" void rh_ctx_destroy(rh_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (auto &f : c->flight) { if (f.th.joinable()) f.th.join(); if (f.ctx) rh_ctx_destroy(f.ctx); f.ctx = nullptr; }
	for (rh_ctx *sc : c->subs) rh_ctx_destroy(sc);
	c->subs.clear();
	DevBuf *all[] = {&c->logf_tab, &c->raw, &c->off, &c->cal_off, &c->cal_scale, &c->act[0], &c->act[1], &c->n_act_dev, &c->zbuf, &c->t1buf, &c->t2buf, &c->n_norm, &c->peaks, &c->n_peaks, &c->ev, &c->n_ev, &c->skip, &c->sx, &c->sy,
	                 &c->n_seed, &c->m_val, &c->m_n, &c->m_meta, &c->m_pref, &c->n_match, &c->n_new, &c->rep_len, &c->a_off, &c->anc, &c->raw_anc, &c->zs, &c->n_z, &c->need_exact, &c->need_exact2, &c->prev_stage, &c->carry[0], &c->carry[1], &c->carry_off, &c->a_off_slice, &c->u,
	                 &c->n_u, &c->n_v, &c->ws, &c->sort_ws, &c->counters, &c->rec, &c->sy_samples, &c->sy_off, &c->sy_cal_off, &c->sy_cal_scale, &c->sy_levels, &c->name_rank, &c->t_rank, &c->rec_off,
	                 &c->events, &c->dtw_ws, &c->dtw_n, &c->dtw_off, &c->dtw_rec, &c->dtw_dec};
	for (DevBuf *b : all) b->release();
	for (DevBuf &b : c->st) b.release();
	if (c->blob_owned) c->blob.release();
	for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
	if (c->pin) (void)hipHostFree(c->pin);
	if (c->e0) (void)hipEventDestroy(c->e0);
	if (c->e1) (void)hipEventDestroy(c->e1);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

// =================================================================================================== index residency
// Blob layout: [table: nb*8 slots of 16 B][pos: n_pos u64][seq_len: n_seq u32]; the 256-byte header describes it so
// that another rank can adopt a broadcast copy.
namespace {
typedef rh_blob_header BlobHeader;
int set_target_ranks_from(rh_ctx *c, const rh_index *ix);
const uint64_t kBlobMagic = RH_BLOB_MAGIC;

int bind_blob(rh_ctx *c, const BlobHeader &h)
{
	unsigned char *base = c->blob.as<unsigned char>();
	c->dix.table = (const rh_tslot*)(base + h.table_off);
	c->dix.pos = (const uint64_t*)(base + h.pos_off);
	c->dix.seq_len = (const uint32_t*)(base + h.len_off);
	c->dix.lg_buckets = h.lg_buckets; c->dix.n_seq = h.n_seq; c->dix.flag = h.flag; c->dix.sp = h.sp;
	c->dix.t_rank = nullptr;                                        // (rh_index_set_target_ranks)
	c->dix.sig_off = h.sig_off ? (const uint64_t*)(base + h.sig_off) : nullptr;
	c->dix.sig = h.sig_off ? (const float*)(base + h.sig_off + ((uint64_t)2 * h.n_seq + 1) * 8) : nullptr;
	// anchor keys fit 32 bits when strand + target id + position do (they do up to a few hundred Mbp in a few targets)
	uint32_t lo = 0, mid = 0;
	while (lo < 32 && (1ull << lo) <= (uint64_t)h.max_len) ++lo;
	while (mid < 32 && (1ull << mid) < (uint64_t)(h.n_seq ? h.n_seq : 1)) ++mid;
	c->akey_on = h.max_len > 0 && lo <= 30 && mid <= 24 && lo + mid + 1 <= 32;
	c->akey_lo = (uint8_t)lo; c->akey_mid = (uint8_t)mid;
	memset(c->header, 0, sizeof(c->header));
	memcpy(c->header, &h, sizeof(h));
	c->have_index = true;
	return 0;
}
} // namespace

// the resident blob may only be replaced while no rh_map_submit batch is mapping against it (the borrowed contexts alias it)
static int index_replaceable(rh_ctx *c, const char *who)
{
	for (auto &f : c->flight) if (f.busy) { rh_set_error("%s: a batch is still in flight on this context (rh_map_wait first): the resident index cannot be replaced", who); return -1; }
	return 0;
}

extern "C" int rh_index_upload(rh_ctx *c, const rh_index *ix)
{
	if (index_replaceable(c, "rh_index_upload")) return -1;
	RH_HIP(hipSetDevice(c->device));
	if (ix->e > 16 || ix->w > RH_DEV_MAXW) { rh_set_error("index parameters e=%d w=%d exceed the device sketch limits (e<=16, w<=%d)", ix->e, ix->w, RH_DEV_MAXW); return -1; }
	std::vector<rh_tslot> slots;
	const int lg = rh_index_make_table(*ix, slots);
	BlobHeader h{};
	h.magic = kBlobMagic;
	h.table_off = 0;
	h.pos_off = slots.size() * sizeof(rh_tslot);
	h.n_pos = ix->pos.size();
	h.len_off = h.pos_off + (h.n_pos ? h.n_pos : 1) * 8;
	h.bytes = h.len_off + (ix->lens.size() ? ix->lens.size() : 1) * 4;
	std::vector<uint64_t> so;
	if ((ix->flag & RH_I_STORE_SIG) && !ix->sigF.empty()) {	// --store-sig: [u64 so[2 n + 1] | floats] behind the lengths (DTW re-scoring aligns with them)
		h.bytes = (h.bytes + 7) & ~7ull;
		h.sig_off = h.bytes;
		so.assign(2 * ix->lens.size() + 1, 0);
		for (size_t i = 0; i < ix->lens.size(); ++i) {
			so[2 * i + 1] = so[2 * i] + (i < ix->sigF.size() ? ix->sigF[i].size() : 0);
			so[2 * i + 2] = so[2 * i + 1] + (i < ix->sigR.size() ? ix->sigR[i].size() : 0);
		}
		h.bytes += so.size() * 8 + (so.back() + 2) * 4;              // (+ two floats of zero padding: the DTW's global border reads one element past a signal, as the reference does)
	}
	h.lg_buckets = lg; h.n_seq = (uint32_t)ix->lens.size(); h.flag = ix->flag;
	h.max_len = 0;
	for (uint32_t L : ix->lens) if (L > h.max_len) h.max_len = L;
	h.sp = rh_sketch_par{ix->e, ix->w, ix->q, ix->k, ix->diff, ix->fine_min, ix->fine_max, ix->fine_range};
	if (!c->blob_owned) { c->blob.p = nullptr; c->blob.cap = 0; c->blob_owned = true; }
	if (c->blob.ensure(h.bytes, false)) return -1;                  // (an index is as large as it is: no growth margin)
	unsigned char *base = c->blob.as<unsigned char>();
	RH_HIP(hipMemcpy(base + h.table_off, slots.data(), slots.size() * sizeof(rh_tslot), hipMemcpyHostToDevice));
	if (h.n_pos) RH_HIP(hipMemcpy(base + h.pos_off, ix->pos.data(), h.n_pos * 8, hipMemcpyHostToDevice));
	if (h.n_seq) RH_HIP(hipMemcpy(base + h.len_off, ix->lens.data(), (size_t)h.n_seq * 4, hipMemcpyHostToDevice));
	if (h.sig_off) {
		RH_HIP(hipMemcpy(base + h.sig_off, so.data(), so.size() * 8, hipMemcpyHostToDevice));
		unsigned char *data = base + h.sig_off + so.size() * 8;
		RH_HIP(hipMemset(data + so.back() * 4, 0, 8));
		for (size_t i = 0; i < ix->lens.size(); ++i) {
			if (i < ix->sigF.size() && !ix->sigF[i].empty()) RH_HIP(hipMemcpy(data + so[2 * i] * 4, ix->sigF[i].data(), ix->sigF[i].size() * 4, hipMemcpyHostToDevice));
			if (i < ix->sigR.size() && !ix->sigR[i].empty()) RH_HIP(hipMemcpy(data + so[2 * i + 1] * 4, ix->sigR[i].data(), ix->sigR[i].size() * 4, hipMemcpyHostToDevice));
		}
	}
	if (bind_blob(c, h)) return -1;
	if (ix->flag & RH_I_SIG_TARGET) return set_target_ranks_from(c, ix);   // all-vs-all: the device compares name ranks (rmap.cpp:86)
	return 0;
}

extern "C" int rh_index_device_blob(rh_ctx *c, void **dev_ptr, uint64_t *bytes, void *header_out)
{
	if (need_index(c)) return -1;
	BlobHeader h; memcpy(&h, c->header, sizeof(h));
	*dev_ptr = c->blob.p; *bytes = h.bytes;
	if (header_out) memcpy(header_out, c->header, 256);
	return 0;
}

extern "C" int rh_index_copy_blob(rh_ctx *c, void *dst)
{
	if (need_index(c)) return -1;
	RH_HIP(hipSetDevice(c->device));
	BlobHeader h; memcpy(&h, c->header, sizeof(h));
	RH_HIP(hipMemcpy(dst, c->blob.p, h.bytes, hipMemcpyDeviceToDevice));
	return 0;
}

extern "C" int rh_index_adopt_blob(rh_ctx *c, const rh_index *, void *dev_ptr, uint64_t bytes, const void *header, int take_ownership)
{
	if (index_replaceable(c, "rh_index_adopt_blob")) return -1;
	BlobHeader h; memcpy(&h, header, sizeof(h));
	if (h.magic != kBlobMagic || h.bytes != bytes) { rh_set_error("index blob header mismatch"); return -1; }
	{	// the offsets must describe three aligned, non-overlapping arrays inside the blob
		const uint64_t tb = h.lg_buckets >= 0 && h.lg_buckets < 40 ? ((uint64_t)RH_TB_SLOTS << h.lg_buckets) * sizeof(rh_tslot) : ~0ull;
		const bool ok = tb != ~0ull && (h.table_off & 15) == 0 && (h.pos_off & 7) == 0 && (h.len_off & 3) == 0 &&
		                h.table_off + tb <= h.pos_off && h.n_pos <= (h.bytes >> 3) && h.pos_off + h.n_pos * 8 <= h.len_off && h.len_off + (uint64_t)h.n_seq * 4 <= h.bytes;
		if (!ok) { rh_set_error("index blob header is inconsistent (offsets / sizes)"); return -1; }
		// --store-sig signals behind the lengths: [u64 so[2 n + 1] | floats]; 0 = none (a header from a build before the field existed holds 0 there)
		if (h.sig_off != 0 && !((h.sig_off & 7) == 0 && h.sig_off >= h.len_off + (uint64_t)h.n_seq * 4 && h.sig_off + ((uint64_t)2 * h.n_seq + 1) * 8 <= h.bytes)) {
			rh_set_error("index blob header is inconsistent (stored-signal offset)"); return -1;
		}
	}
	if (c->blob_owned) c->blob.release();
	c->blob.p = dev_ptr; c->blob.cap = bytes; c->blob_owned = take_ownership != 0;
	return bind_blob(c, h);
}

// =================================================================================================== index built on the device
extern "C" rh_index *rh_index_build_device(rh_ctx *c, uint32_t n_seq, const char *const *names, const char *const *seqs, const uint32_t *lens,
                                           const char *pore_model_path, const rh_idxopt_t *io, int n_threads)
{
	if (index_replaceable(c, "rh_index_build_device")) return nullptr;
	if (hipSetDevice(c->device) != hipSuccess) { rh_set_error("hipSetDevice failed"); return nullptr; }
	if (io->flag & RH_I_SIG_TARGET) { rh_set_error("signal-target (Rawsamble) indexes are built by rh_index_build_signals"); return nullptr; }
	if (io->w < 0 || io->w > RH_DEV_MAXW) { rh_set_error("minimiser window w = %d outside what the device sketch maps with (0..%d)", io->w, RH_DEV_MAXW); return nullptr; }
	if (io->e < 1 || io->e > 16 || io->q < 1 || io->q * io->e > 64 || io->k < 1 || io->k > 12) { rh_set_error("unsupported index parameters e=%d q=%d k=%d", io->e, io->q, io->k); return nullptr; }
	std::unique_ptr<rh_index_s> ix(new rh_index_s());
	ix->w = io->w; ix->e = io->e; ix->n = io->n; ix->q = io->q; ix->k = io->k; ix->flag = io->flag;
	ix->diff = io->diff; ix->fine_min = io->fine_min; ix->fine_max = io->fine_max; ix->fine_range = io->fine_range;
	if (!rh_load_model(pore_model_path, io->k, io->lev_col, ix->pore_vals)) return nullptr;
	ix->n_pore_vals = (uint32_t)ix->pore_vals.size(); ix->pore_k = (int16_t)io->k;
	rh_make_pore_inds(ix->pore_vals, io->k, ix->pore_inds);
	for (uint32_t i = 0; i < n_seq; ++i) { ix->names.push_back(names && names[i] ? names[i] : ""); ix->lens.push_back(lens[i]); }
	// the resident index of this context is replaced
	if (c->blob_owned) c->blob.release(); else { c->blob.p = nullptr; c->blob.cap = 0; }
	c->have_index = false;
	BlobHeader h{};
	void *blob = nullptr;
	uint64_t n_keys = 0;
	if (rhk_index_build_device(c->stream, n_seq, seqs, lens, ix->pore_vals, io, &h, &blob, ix->occ_hist, &n_keys, n_threads)) return nullptr;
	c->blob.p = blob; c->blob.cap = h.bytes; c->blob.owned = true; c->blob_owned = true;
	if (bind_blob(c, h)) return nullptr;
	ix->dev_n_keys = n_keys; ix->dev_n_pos = h.n_pos;
	return ix.release();
}

extern "C" rh_index *rh_index_build_device_fasta(rh_ctx *c, const char *fasta_path, const char *pore_model_path, const rh_idxopt_t *io, int n_threads)
{
	std::vector<std::string> names, seqs;
	if (!rh_read_fasta(fasta_path, names, seqs)) return nullptr;
	std::vector<const char*> np, sp; std::vector<uint32_t> ln;
	for (size_t i = 0; i < seqs.size(); ++i) {
		if (seqs[i].size() >= (1ull << 31)) { rh_set_error("%s: sequence %s is too long", fasta_path, names[i].c_str()); return nullptr; }
		np.push_back(names[i].c_str()); sp.push_back(seqs[i].data()); ln.push_back((uint32_t)seqs[i].size());
	}
	return rh_index_build_device(c, (uint32_t)seqs.size(), np.data(), sp.data(), ln.data(), pore_model_path, io, n_threads);
}

// ---------------------------------------------------------------------------------------------------- all-vs-all: name ranks
// strcmp(qname, tname) >= 0 (rmap.cpp:86) as an integer comparison: the distinct target names in strcmp order get the odd
// ranks 1, 3, 5, ...; a query name that is a target name takes its rank, any other the even rank just below the first greater
// target.  Then strcmp(q, t) >= 0  <=>  rank(q) >= rank(t).
namespace {
void sorted_unique_names(const std::vector<std::string> &names, std::vector<const std::string*> &u)
{
	u.clear();
	for (const std::string &n : names) u.push_back(&n);
	std::sort(u.begin(), u.end(), [](const std::string *a, const std::string *b) { return strcmp(a->c_str(), b->c_str()) < 0; });
	u.erase(std::unique(u.begin(), u.end(), [](const std::string *a, const std::string *b) { return strcmp(a->c_str(), b->c_str()) == 0; }), u.end());
}
uint32_t rank_of(const std::vector<const std::string*> &u, const char *q)
{
	size_t lo = 0, hi = u.size();                                  // first target name >= q
	while (lo < hi) { const size_t mid = (lo + hi) / 2; if (strcmp(u[mid]->c_str(), q) < 0) lo = mid + 1; else hi = mid; }
	if (lo < u.size() && strcmp(u[lo]->c_str(), q) == 0) return (uint32_t)(2 * lo + 1);
	return (uint32_t)(2 * lo);
}
}
extern "C" int rh_index_name_ranks(const rh_index *ix, const char *const *names, uint32_t n, uint32_t *query_ranks, uint32_t *target_ranks)
{
	std::vector<const std::string*> u;
	sorted_unique_names(ix->names, u);
	if (query_ranks) for (uint32_t i = 0; i < n; ++i) query_ranks[i] = rank_of(u, names[i] ? names[i] : "");
	if (target_ranks) for (size_t t = 0; t < ix->names.size(); ++t) target_ranks[t] = rank_of(u, ix->names[t].c_str());
	return 0;
}
extern "C" int rh_index_set_target_ranks(rh_ctx *c, const uint32_t *target_ranks, uint32_t n)
{
	if (need_index(c)) return -1;
	RH_HIP(hipSetDevice(c->device));
	if (n != c->dix.n_seq) { rh_set_error("%u target ranks for an index of %u targets", n, c->dix.n_seq); return -1; }
	if (c->t_rank.ensure((size_t)(n ? n : 1) * 4)) return -1;
	if (n) RH_HIP(hipMemcpy(c->t_rank.p, target_ranks, (size_t)n * 4, hipMemcpyHostToDevice));
	c->dix.t_rank = c->t_rank.as<uint32_t>();
	return 0;
}
namespace {
int set_target_ranks_from(rh_ctx *c, const rh_index *ix)
{
	std::vector<uint32_t> tr(ix->names.size());
	rh_index_name_ranks(ix, nullptr, 0, nullptr, tr.data());
	return rh_index_set_target_ranks(c, tr.data(), (uint32_t)tr.size());
}
}

// keys and positions of the resident index back into the host object (hash order), for rh_index_get / rh_index_write
extern "C" int rh_index_download(rh_ctx *c, rh_index *ix, int n_threads)
{
	if (need_index(c)) return -1;
	RH_HIP(hipSetDevice(c->device));
	BlobHeader h; memcpy(&h, c->header, sizeof(h));
	const uint64_t n_slots = (uint64_t)RH_TB_SLOTS << h.lg_buckets;
	std::vector<rh_tslot> slots(n_slots);
	RH_HIP(hipMemcpy(slots.data(), c->blob.as<unsigned char>() + h.table_off, n_slots * sizeof(rh_tslot), hipMemcpyDeviceToHost));
	ix->pos.resize(h.n_pos);
	if (h.n_pos) RH_HIP(hipMemcpy(ix->pos.data(), c->blob.as<unsigned char>() + h.pos_off, h.n_pos * 8, hipMemcpyDeviceToHost));
	// non-empty slots by the top byte of the hash, each range sorted concurrently
	if (n_threads < 1) n_threads = 1;
	std::vector<uint64_t> cnt(257, 0);
	for (const rh_tslot &sl : slots) if (sl.n) ++cnt[(sl.hash >> 24) + 1];
	for (int i = 0; i < 256; ++i) cnt[i + 1] += cnt[i];
	std::vector<rh_tslot> ent(cnt[256]);
	{ std::vector<uint64_t> w(cnt.begin(), cnt.end() - 1); for (const rh_tslot &sl : slots) if (sl.n) ent[w[sl.hash >> 24]++] = sl; }
	std::vector<rh_tslot>().swap(slots);
	{
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t)
			th.emplace_back([&, t]() { for (int b = t; b < 256; b += n_threads) std::sort(ent.begin() + cnt[b], ent.begin() + cnt[b + 1], [](const rh_tslot &a, const rh_tslot &b2) { return a.hash < b2.hash; }); });
		for (auto &t : th) t.join();
	}
	const size_t nk = ent.size();
	ix->key_hash.resize(nk); ix->key_n.resize(nk); ix->key_val.resize(nk);
	for (size_t i = 0; i < nk; ++i) { ix->key_hash[i] = ent[i].hash; ix->key_n[i] = ent[i].n; ix->key_val[i] = ent[i].val; }
	if (h.sig_off && (ix->flag & RH_I_STORE_SIG)) {	// --store-sig: the targets' signals, for rh_index_write (rindex.c:590-598)
		std::vector<uint64_t> so((size_t)2 * h.n_seq + 1);
		const unsigned char *sb = c->blob.as<unsigned char>() + h.sig_off;
		RH_HIP(hipMemcpy(so.data(), sb, so.size() * 8, hipMemcpyDeviceToHost));
		const bool rev = !(ix->flag & RH_I_NO_REV_TARGET);
		ix->sigF.assign(h.n_seq, std::vector<float>()); ix->sigR.clear();
		if (rev) ix->sigR.assign(h.n_seq, std::vector<float>());
		for (uint32_t i = 0; i < h.n_seq; ++i) {
			ix->sigF[i].resize(so[2 * (size_t)i + 1] - so[2 * (size_t)i]);
			if (!ix->sigF[i].empty()) RH_HIP(hipMemcpy(ix->sigF[i].data(), sb + so.size() * 8 + so[2 * (size_t)i] * 4, ix->sigF[i].size() * 4, hipMemcpyDeviceToHost));
			if (rev) {
				ix->sigR[i].resize(so[2 * (size_t)i + 2] - so[2 * (size_t)i + 1]);
				if (!ix->sigR[i].empty()) RH_HIP(hipMemcpy(ix->sigR[i].data(), sb + so.size() * 8 + so[2 * (size_t)i + 1] * 4, ix->sigR[i].size() * 4, hipMemcpyDeviceToHost));
			}
		}
	}
	return 0;
}

// =================================================================================================== the hot path
extern "C" uint64_t rh_map_max_records(const rh_read_batch_t *in, const rh_mapopt_t *) { return in->n_reads; }

extern "C" const char *rh_stage_name(int i) { return (i >= 0 && i < 24) ? kStageName[i] : ""; }

extern "C" int rh_map_last_stats(rh_ctx *c, rh_map_stats_t *out) { *out = c->stats; return 0; }

namespace {
// RH_DEBUG_ROUNDS=1: per chunk round, the anchor-count distribution of the active reads and how many needed the exact sort
bool debug_rounds() { static const bool on = RH_DEVENV("RH_DEBUG_ROUNDS") != nullptr; return on; }
void dump_round(rh_ctx *c, uint32_t chunk, uint32_t n_act, const rh_dev_round &rr)
{
	std::vector<uint64_t> off((size_t)n_act + 1);
	std::vector<uint8_t> ex(n_act);
	(void)hipStreamSynchronize(c->stream);
	(void)hipMemcpy(off.data(), rr.a_off, off.size() * 8, hipMemcpyDeviceToHost);
	(void)hipMemcpy(ex.data(), rr.need_exact, n_act, hipMemcpyDeviceToHost);
	const uint32_t edges[] = {0, 1, 64, 512, 1024, 2048, 3072, 4096, 6144, 8192, 1u << 30};
	uint32_t h[10] = {0}, hx[10] = {0};
	for (uint32_t a = 0; a < n_act; ++a) {
		const uint64_t n = off[a + 1] - off[a];
		for (int b = 0; b < 10; ++b) if (n >= edges[b] && n < edges[b + 1]) { ++h[b]; hx[b] += ex[a]; }
	}
	fprintf(stderr, "[round %u] n_act %u anchors %llu :", chunk, n_act, (unsigned long long)off[n_act]);
	for (int b = 0; b < 10; ++b) fprintf(stderr, " <%u:%u(%u)", edges[b + 1], h[b], hx[b]);
	fprintf(stderr, "\n");
}

void dump_round2(rh_ctx *c, uint32_t chunk, uint32_t n_act, const rh_dev_round &rr)
{
	std::vector<uint32_t> nu(n_act), nz(n_act), nv(n_act);
	(void)hipStreamSynchronize(c->stream);
	(void)hipMemcpy(nu.data(), rr.n_u, (size_t)n_act * 4, hipMemcpyDeviceToHost);
	(void)hipMemcpy(nz.data(), rr.n_z, (size_t)n_act * 4, hipMemcpyDeviceToHost);
	(void)hipMemcpy(nv.data(), rr.n_v, (size_t)n_act * 4, hipMemcpyDeviceToHost);
	const uint32_t edges[] = {0, 1, 8, 64, 256, 512, 768, 1024, 1536, 1u << 30};
	uint32_t h[9] = {0};
	uint64_t su = 0, sz = 0, sv = 0;
	for (uint32_t a = 0; a < n_act; ++a) { su += nu[a]; sz += nz[a]; sv += nv[a]; for (int b = 0; b < 9; ++b) if (nu[a] >= edges[b] && nu[a] < edges[b + 1]) ++h[b]; }
	fprintf(stderr, "[round %u] chains: mean n_u %.1f n_z %.1f n_v %.1f :", chunk, (double)su / n_act, (double)sz / n_act, (double)sv / n_act);
	for (int b = 0; b < 9; ++b) fprintf(stderr, " <%u:%u", edges[b + 1], h[b]);
	fprintf(stderr, "\n");
}

// the whole path for one (sub-)batch on one context's stream
// strides of the per-read rows: a chunk's worth, or (whole-read rounds) what the longest read of the batch needs
int set_row_strides(rh_ctx *c, const rh_mapopt_t *mo, const rh_read_batch_t *in)
{
	c->whole = (mo->flag & RH_M_NO_ADAPTIVE) ? 1 : 0;
	c->cs_stride = (c->whole ? 1u : (mo->max_num_chunk ? mo->max_num_chunk : 1u)) + 1u;
	c->ev_row = RH_CHUNK_MAX + 64; c->ev_cap = RH_EV_CAP;
	if (!c->whole && mo->chunk_size > RH_CHUNK_MAX) {	// chunks beyond the LDS-resident event kernels: the rows-in-HBM kernels of the whole-read rounds, a chunk at a time
		c->whole = 1;
		c->ev_row = (uint32_t)(((uint64_t)mo->chunk_size + 64 + 63) & ~63ull);
		c->ev_cap = c->ev_row / 2 + 2;
		return 0;
	}
	if (!c->whole) return 0;
	const uint32_t R = in->n_reads;
	std::vector<uint64_t> off_h;
	const uint64_t *off = in->offsets;
	if (in->samples_on_device && R) { off_h.resize((size_t)R + 1); RH_HIP(hipMemcpy(off_h.data(), in->offsets, ((size_t)R + 1) * 8, hipMemcpyDeviceToHost)); off = off_h.data(); }
	uint64_t mx = 0;
	for (uint32_t r = 0; r < R; ++r) if (off[r + 1] - off[r] > mx) mx = off[r + 1] - off[r];
	if (mx >= (1ull << 26)) { rh_set_error("whole-read rounds: reads of 2^26 samples or more are not supported"); return -1; }
	c->ev_row = (uint32_t)((mx + 64 + 63) & ~63ull);
	c->ev_cap = c->ev_row / 2 + 2;                                  // peaks are >= 2 samples apart (revent.c:140)
	if (c->ev_cap < RH_EV_CAP) c->ev_cap = RH_EV_CAP;
	return 0;
}

// 8-byte records through the sorters where key and payload fit one word (rh_rec_fmt, rh_kernels.h): the candidates of the backtrack
// (score | anchor), the chain-order keys (first anchor's strand | target | position, chain number) and - when strand, target, position,
// tandem flag and query position fit - the anchors themselves.  What does not fit keeps its 16-byte records.
void set_rec8_formats(const rh_ctx *c, const rh_mapopt_t *mo, const rh_dev_opt &o, rh_dev_round *rr)
{
	rr->afmt = rh_rec_fmt{0, 0, 0, 0}; rr->cfmt = rh_rec_fmt{0, 0, 0, 0}; rr->aq_bits = 0; rr->z8 = 0; rr->a_span = 0;
	static const bool off = RH_DEVENV("RH_REC8") && atoi(RH_DEVENV("RH_REC8")) == 0;   // RH_REC8=0: 16-byte records everywhere (comparison runs)
	if (off) return;
	rr->z8 = o.min_sc >= 0 ? 1 : 0;
	rh_blob_header h;
	memcpy(&h, c->header, sizeof(h));
	const uint32_t lo = c->akey_lo, mid = c->akey_mid;
	if (h.max_len == 0 || lo > 30 || mid > 31) return;             // (x keeps position bit 30 a second time at bit 31: zero below 2^30)
	const uint32_t kb = 1u + lo + mid;
	const uint32_t ib = 64u - kb < 31u ? 64u - kb : 31u;            // chain number below the key (checked against the slice's largest read)
	if (ib >= 8u) rr->cfmt = rh_rec_fmt{1, (uint8_t)ib, (uint8_t)lo, (uint8_t)mid};
	if (mo->flag & RH_M_ALL_CHAINS) return;                        // (k_expand_ava writes 16-byte anchors)
	// the serial / RMQ chaining kernels and the DTW alignment read 16-byte anchors: opt-in modes, left as they are
	if ((mo->flag & (RH_M_RMQ | RH_M_DTW_EVALUATE_CHAINS)) || mo->bw_long > mo->bw || o.max_iter > 255) return;
	const uint32_t span = (uint32_t)(c->dix.sp.k + c->dix.sp.e - 1);
	if (span > 63u) return;
	const uint64_t q_max = (mo->flag & RH_M_NO_ADAPTIVE) ? (uint64_t)c->ev_cap + 1u : ((uint64_t)mo->max_num_chunk + 1u) * c->ev_cap;   // query positions = events so far
	uint32_t qb = 1;
	while (qb < 32u && (1ull << qb) <= q_max) ++qb;
	if (kb + 1u + qb <= 64u) { rr->afmt = rh_rec_fmt{1, (uint8_t)(qb + 1u), (uint8_t)lo, (uint8_t)mid}; rr->aq_bits = (uint8_t)qb; rr->a_span = (uint8_t)span; }
}

// rec_off != null: all-vs-all, a read may have several records (rec_off[r] .. rec_off[r + 1], n_reads + 1 offsets)
// Consumed-prefix staging: before round `chunk`, every active read must hold the chunk's raw samples in HBM (k_need's condition).  The reads that
// lack them are extended by the device itself from the caller's page-locked samples - two and a quarter chunks ahead in rounds 0 and 1 (a read
// that maps is done by then), the whole rest from round 2 on (what is still active then runs to max_num_chunk) - and re-ranked by k_prefilter.
int ensure_resident(rh_ctx *c, hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const uint32_t *act, uint32_t n_act, uint32_t chunk)
{
	const uint64_t ahead = 2ull * o.chunk_size + o.chunk_size / 4 + 64;
	uint32_t grow = (chunk < 2 && ahead < c->lazy_maxlen) ? (uint32_t)ahead : 0xFFFFFFFFu;
	uint32_t *n_need = c->n_act_dev.as<uint32_t>() + 8;
	for (;;) {
		RH_HIP(hipMemsetAsync(n_need, 0, 4, s));
		rhk_need(s, o, rd, act, n_act, chunk, grow, c->new_len.as<uint32_t>(), n_need);
		RH_HIP(hipMemcpyAsync(c->pin + 6, n_need, 4, hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		const uint32_t needy = (uint32_t)c->pin[6];
		if (!needy) return 0;
		rhk_fetch(s, rd, c->lazy_host, act, n_act, c->new_len.as<uint32_t>(), grow < c->lazy_maxlen ? grow : c->lazy_maxlen);
		rhk_prefilter(s, o, rd, act, n_act, 0, c->counters.as<unsigned long long>() + 8);
		grow = 0xFFFFFFFFu;                                           // (a stretch that was not enough: many samples outside 30 .. 200 pA - the whole read next)
	}
}

int map_batch_once(rh_ctx *c, const rh_mapopt_t *mo, const rh_read_batch_t *in, rh_map_record_t *out, uint64_t out_cap, uint64_t *n_out, uint64_t *rec_off = nullptr)
{
	*n_out = 0;
	if (need_index(c)) return -1;
	RH_HIP(hipSetDevice(c->device));
	rh_dev_opt o;
	if (fill_dev_opt(c, mo, &o)) return -1;
	const uint32_t R = in->n_reads;
	const bool ava = (mo->flag & RH_M_ALL_CHAINS) != 0;
	if (ava && !rec_off) { rh_set_error("all-vs-all mapping returns several records per read: call rh_map_batch_multi"); return -1; }
	if (ava && !in->name_rank) { rh_set_error("all-vs-all mapping needs the name ranks of the reads (rh_read_batch_t::name_rank, see rh_index_name_ranks)"); return -1; }
	if (out_cap < R) { rh_set_error("output capacity %llu < %u reads", (unsigned long long)out_cap, R); return -1; }
	if (rec_off) rec_off[0] = 0;
	if (R == 0) return 0;
	if (set_row_strides(c, mo, in)) return -1;
	memset(&c->stats, 0, sizeof(c->stats));
	c->dtw_dev_reads = 0; c->dtw_host_reads = 0;
	c->ev_used = 0; c->ev_stage.clear();
	const auto t_begin = std::chrono::steady_clock::now();
	hipStream_t s = c->stream;
	rh_dev_reads rd;
	{ StageTimer t(c, ST_H2D); if (stage_reads(c, in, &rd, !(mo->flag & RH_M_NO_ADAPTIVE))) return -1; }
	const bool lazy = rd.res_len != nullptr;
	const bool dtw = (mo->flag & RH_M_DTW_EVALUATE_CHAINS) != 0;
	if (dtw) {	// every read keeps the events of all its processed chunks
		rd.ev_stride = (mo->flag & RH_M_NO_ADAPTIVE) ? c->ev_cap : mo->max_num_chunk * c->ev_cap;
		if (c->events.ensure((size_t)R * rd.ev_stride * 4)) return -1;
		rd.events = c->events.as<float>();
	}
	if (c->act[0].ensure((size_t)R * 4) || c->act[1].ensure((size_t)R * 4) || c->n_act_dev.ensure(64) || c->carry[0].ensure(16) || c->carry[1].ensure(16) || c->counters.ensure(16 * 8) || c->rec.ensure((size_t)R * sizeof(rh_map_record_t))) return -1;
	RH_HIP(hipMemsetAsync(c->counters.p, 0, 16 * 8, s));
	{ StageTimer t(c, ST_PREFILTER); rhk_prefilter(s, o, rd, nullptr, 0, 1, c->counters.as<unsigned long long>() + 8); }   // (consumed-prefix staging: nothing resident yet - state set up, lengths taken from the caller)
	int cur = 0;
	{ StageTimer t(c, ST_COMPACT); rhk_compact_active(s, o, rd, nullptr, R, 0, c->act[cur].as<uint32_t>(), c->n_act_dev.as<uint32_t>()); }
	if (!c->pin) RH_HIP(hipHostMalloc((void**)&c->pin, 256, 0));
	uint32_t n_act = 0;
	RH_HIP(hipMemcpyAsync(c->pin, c->n_act_dev.p, 4, hipMemcpyDeviceToHost, s));
	RH_HIP(hipStreamSynchronize(s));
	n_act = (uint32_t)c->pin[0];
	int which = 0;
	uint64_t carry_used = 0;                                       // anchors in carry[which] so far this round
	for (uint32_t chunk = 0; chunk < mo->max_num_chunk && n_act > 0; ++chunk) {
		rh_dev_round rr{};
		if (stage_round(c, n_act, &rr)) return -1;
		rr.act = c->act[cur].as<uint32_t>(); rr.chunk = chunk;
		if (lazy) { StageTimer t(c, ST_H2D); if (ensure_resident(c, s, o, rd, rr.act, n_act, chunk)) return -1; }
		rr.akey_on = c->akey_on ? 1 : 0; rr.akey_lo = c->akey_lo; rr.akey_mid = c->akey_mid;
		set_rec8_formats(c, mo, o, &rr);
		rr.prev_in = c->carry[which ^ 1].as<rh_mm128_t>();
		{ StageTimer t(c, ST_EV_NORM); rhk_events_norm(s, o, rd, rr); }
		{ StageTimer t(c, ST_EV_PEAKS); rhk_events_peaks(s, o, rr); }
		{ StageTimer t(c, ST_EV_MEANS); rhk_events_means(s, o, rr); }
		if (dtw) rhk_events_append(s, rd, rr);                          // reg->events (rmap.cpp:237-241)
		{ StageTimer t(c, ST_SKETCH); rhk_sketch(s, o, c->dix, rd, rr); }
		{ StageTimer t(c, ST_PROBE); rhk_probe(s, o, c->dix, rd, rr); }
		uint64_t total = 0;
		{ StageTimer t(c, ST_SCAN); rhk_scan_anchors(s, rd, rr); RH_HIP(hipMemcpyAsync(c->pin + 1, rr.a_off + n_act, 16, hipMemcpyDeviceToHost, s)); }
		RH_HIP(hipStreamSynchronize(s));
		total = c->pin[1];
		const uint32_t max_all = (uint32_t)c->pin[2];
		// The anchor-sized stages run over slices of the active reads whose anchors fit the device together (one slice unless
		// the index is large and the batch big); the event / seeding stages above ran for all of them at once.
		const uint64_t budget = slice_budget(c);
		if (budget == 0) { rh_set_error("not enough free device memory for the anchor arenas of a batch (the index and other batches in flight hold it)"); return -1; }
		std::vector<uint32_t> cuts;                                // slice boundaries in the active list
		std::vector<uint64_t> a_off_h;
		cuts.push_back(0);
		if (total > budget) {
			a_off_h.resize((size_t)n_act + 1);
			RH_HIP(hipMemcpy(a_off_h.data(), rr.a_off, ((size_t)n_act + 1) * 8, hipMemcpyDeviceToHost));
			uint32_t lo = 0;
			for (uint32_t a = 1; a <= n_act; ++a)
				if (a_off_h[a] - a_off_h[lo] > budget && a - 1 > lo) { cuts.push_back(a - 1); lo = a - 1; }
		}
		cuts.push_back(n_act);
		static const bool trace_rounds = RH_DEVENV("RH_TRACE_ROUNDS") != nullptr;   // development aid
		if (trace_rounds) fprintf(stderr, "[ctx %p] round %u: %u reads, %llu anchors, budget %llu, %zu slice(s), %.1f ms since the call began\n", (void*)c, chunk, n_act,
		                          (unsigned long long)total, (unsigned long long)budget, cuts.size() - 1, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
		carry_used = 0;
		if (c->carry_off.ensure((size_t)(n_act + 1) * 8) || c->a_off_slice.ensure((size_t)(n_act + 2) * 8)) return -1;
		for (size_t si = 0; si + 1 < cuts.size(); ++si) {
			const uint32_t lo = cuts[si], n = cuts[si + 1] - lo;
			rh_dev_round rs = rr;
			uint64_t stotal = total;
			rs.max_anchors = max_all;
			if (cuts.size() > 2) {
				rhk_rebase_offsets(s, rr.a_off + lo, n, c->a_off_slice.as<uint64_t>());
				rs = slice_view(rr, lo, n, c->a_off_slice.as<uint64_t>());
				stotal = a_off_h[lo + n] - a_off_h[lo];
				uint32_t mx = 0;
				for (uint32_t a = lo; a < lo + n; ++a) { const uint64_t m = a_off_h[a + 1] - a_off_h[a]; if (m > mx) mx = (uint32_t)m; }
				rs.max_anchors = mx;
			}
			if (stage_anchors(c, stotal, &rs, cuts.size() > 2 ? budget : 0)) return -1;
			if (rs.cfmt.rec8 && (rs.max_anchors == 0 || (uint64_t)rs.max_anchors >= (1ull << rs.cfmt.shift))) rs.cfmt = rh_rec_fmt{0, 0, 0, 0};   // (a chain number must fit below the key)
			if (cuts.size() > 2) c->arena_room = budget;
			// compact_a's copy of the chains back over the anchors is left out where nothing reads it: the region stage takes the chains' ends from where
			// they were gathered (not: DTW - it aligns along the chains -, re-chaining, all-vs-all, the serial region kernels' own sort, chains of one anchor)
			rs.lazy_reorder = (!dtw && !ava && !(o.bw_long > o.bw) && o.min_cnt >= 2 && rhk_regions_fast_ok(o) && !debug_rounds()) ? 1 : 0;
			// the chained anchors the reads carry into their next chunk: staging arena -> dense carry buffer (all-vs-all: the
			// reported chains, which the region stage leaves there)
			auto pack_carry = [&]() -> int {
				StageTimer t(c, ST_COMPACT);
				rhk_carry_scan(s, rd, rs.act, n, carry_used, c->carry_off.as<uint64_t>(), c->n_act_dev.as<uint64_t>() + 2);
				RH_HIP(hipMemcpyAsync(c->pin + 4, c->n_act_dev.as<uint64_t>() + 2, 8, hipMemcpyDeviceToHost, s));
				RH_HIP(hipStreamSynchronize(s));
				const uint64_t add = c->pin[4];
				// (one-word anchors are carried as words: 8 bytes each.  The buffer is sized once a round, from what the slices so far say about the rest of them - a growth
				// step of a half with the old buffer still there was what large calls ran out of memory on: 262 144 reads a call, round 0, 12 GB beside 12 GB)
				const size_t esz = (rs.afmt.rec8 && !ava) ? 8 : 16;
				const uint64_t seen = carry_used + add, n_sl = cuts.size() - 1;
				const uint64_t est = n_sl > si + 1 ? (uint64_t)((double)seen * (double)n_sl / (double)(si + 1) * 1.1) + 4096 : seen + 1;
				if (c->carry[which].ensure_keep_sized((size_t)(seen + 1) * esz, (size_t)(est + 1) * esz, (size_t)carry_used * esz, s)) return -1;
				rhk_carry_copy(s, rd, rs.act, n, rs.prev_out, c->carry_off.as<uint64_t>(), c->carry[which].as<rh_mm128_t>(), (rs.afmt.rec8 && !ava) ? 1 : 0);
				carry_used += add;
				if (si + 2 == cuts.size() && R) { const size_t per = (size_t)(carry_used * esz / R) + 1; if (per > c->carry_per_read) c->carry_per_read = per; }   // (the round's last slice: bytes carried per read of the call)
				return 0;
			};
			{ StageTimer t(c, ST_EXPAND); rhk_expand(s, o, c->dix, rd, rs); }
			{	// (all-vs-all keeps the exact passes for all: k_expand_ava has no second run)
				StageTimer t(c, ST_SORT);
				std::function<int(const uint8_t*)> again;
				if (!ava) again = [&](const uint8_t *mask) { rhk_expand(s, o, c->dix, rd, rs, mask); return 0; };
				if (rhk_sort(s, c->dix, rs, again)) return -1;
			}
			if (debug_rounds()) dump_round(c, chunk, n, rs);
			if (chain_stages(c, s, o, rd, rs, true)) return -1;
			if (!ava && pack_carry()) return -1;                      // (before the region sort: it borrows the staging arena)
			if (dtw) { StageTimer t(c, ST_REGIONS); if (dtw_regions_stage(c, s, o, mo, rd, rs, n, chunk)) return -1; }
			else {
				{ StageTimer t(c, ST_RSORT); if (rhk_regions_sort(s, o, rd, rs)) return -1; }
				{ StageTimer t(c, ST_REGIONS); rhk_regions(s, o, rd, rs, c->logf_tab.as<float>()); }
			}
			if (debug_rounds()) dump_round2(c, chunk, n, rs);
			if (ava && pack_carry()) return -1;
		}
		{ StageTimer t(c, ST_COMPACT); rhk_compact_active(s, o, rd, rr.act, n_act, chunk + 1, c->act[cur ^ 1].as<uint32_t>(), c->n_act_dev.as<uint32_t>()); }
		RH_HIP(hipMemcpyAsync(c->pin, c->n_act_dev.p, 4, hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		n_act = (uint32_t)c->pin[0];
		cur ^= 1; which ^= 1;
	}
	uint64_t n_rec = R;
	if (!ava) { StageTimer t(c, ST_FINALIZE); rhk_finalize(s, o, c->dix, rd, c->rec.as<rh_map_record_t>()); }
	else {	// the reported chains wait in the dense carry buffer of the (only) round
		StageTimer t(c, ST_FINALIZE);
		if (c->rec_off.ensure((size_t)(R + 1) * 8)) return -1;
		rhk_ava_rec_scan(s, rd, c->rec_off.as<uint64_t>());
		RH_HIP(hipMemcpyAsync(rec_off, c->rec_off.p, (size_t)(R + 1) * 8, hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		n_rec = rec_off[R];
		if (n_rec > out_cap) { rh_set_error("output capacity %llu < %llu records", (unsigned long long)out_cap, (unsigned long long)n_rec); return -1; }
		if (c->rec.ensure((size_t)n_rec * sizeof(rh_map_record_t))) return -1;
		rhk_finalize_ava(s, o, c->dix, rd, c->carry[which ^ 1].as<rh_mm128_t>(), c->rec_off.as<uint64_t>(), c->rec.as<rh_map_record_t>());
	}
	{
		StageTimer t(c, ST_D2H);
		RH_HIP(hipMemcpyAsync(out, c->rec.p, (size_t)n_rec * sizeof(rh_map_record_t), hipMemcpyDeviceToHost, s));
		uint64_t cnt[16];
		RH_HIP(hipMemcpyAsync(cnt, c->counters.p, sizeof(cnt), hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		c->stats.n_events = cnt[0]; c->stats.n_seeds = cnt[1]; c->stats.n_hits = cnt[2]; c->stats.n_anchors = cnt[3]; c->stats.n_chained = cnt[4];
		c->stats.n_samples_used = cnt[5]; c->stats.n_chunks = cnt[6];
		for (int q = 0; q < 4; ++q) c->stats.n_rmq_class[q] = cnt[9 + q];
		c->stats.n_dtw_device = c->dtw_dev_reads; c->stats.n_dtw_host = c->dtw_host_reads;
		if (cnt[7]) { rh_set_error("%llu chunk(s) hold more than %d event boundaries: beyond the per-chunk arrays of the device path", (unsigned long long)cnt[7], RH_EV_CAP); return -1; }
		if (cnt[8]) { rh_set_error("rh_read_batch_t::n_filtered is wrong for %llu read(s): not the number of samples the pA filter (rsig.c:496-503) leaves of them", (unsigned long long)cnt[8]); return -1; }
	}
	RH_HIP(hipStreamSynchronize(s));                                // the last stop event
	stage_timers_collect(c);
	RH_HIP(hipGetLastError());
	c->stats.n_reads = R;
	c->stats.n_samples_raw = in->samples_on_device ? 0 : in->offsets[R] - in->offsets[0];
	c->stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
	*n_out = n_rec;
	return 0;
}
void add_stats(rh_map_stats_t &tot, const rh_map_stats_t &q)
{
	tot.n_reads += q.n_reads; tot.n_chunks += q.n_chunks; tot.n_samples_raw += q.n_samples_raw; tot.n_samples_used += q.n_samples_used;
	tot.n_events += q.n_events; tot.n_seeds += q.n_seeds; tot.n_hits += q.n_hits; tot.n_anchors += q.n_anchors; tot.n_chained += q.n_chained;
	tot.ms_total += q.ms_total;
	for (int i = 0; i < 4; ++i) tot.n_rmq_class[i] += q.n_rmq_class[i];
	tot.n_dtw_device += q.n_dtw_device; tot.n_dtw_host += q.n_dtw_host;
	for (int i = 0; i < 24; ++i) { tot.ms_kernel[i] += q.ms_kernel[i]; tot.n_launch[i] += q.n_launch[i]; }
}

// One (sub-)batch on one context.  The per-anchor arenas are sized by what the reads turn out to need (tens of thousands
// of anchors per chunk on a large index); when they do not fit the device, the batch is mapped in consecutive slices,
// halving the slice until it fits.  Reads are independent, so the records are the same.
int map_batch_single(rh_ctx *c, const rh_mapopt_t *mo, const rh_read_batch_t *in, rh_map_record_t *out, uint64_t out_cap, uint64_t *n_out)
{
	const uint32_t R = in->n_reads;
	*n_out = 0;
	if (out_cap < R) { rh_set_error("output capacity %llu < %u reads", (unsigned long long)out_cap, R); return -1; }
	g_oom = false;
	uint64_t n = 0;
	uint32_t slice = R / 2, done = 0, call_cap = 0xFFFFFFFFu;
	// The event / seeding stages hold rows for every active read of a call (~140 KB each: z / t1 / t2 rows, peaks, events, seeds,
	// matches): a call takes as many reads as fit a third of this context's share of the free memory (a million-read batch is
	// mapped in a few consecutive calls; the anchor-sized stages have their own slices inside a call).
	if (!(mo->flag & RH_M_NO_ADAPTIVE)) {
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
			size_t per_read = (size_t)(RH_CHUNK_MAX + 64) * 12 + (size_t)RH_EV_CAP * 44 + 4096;
			if (mo->flag & RH_M_DTW_EVALUATE_CHAINS) per_read += (size_t)mo->max_num_chunk * RH_EV_CAP * 4 * 5 + 256;   // reg->events of every chunk + the DP buffers (4 x the events so far), see dtw_regions_stage
			// ... and what else a read keeps on the device for the whole call: its signal when the batch comes from the host, and its chained anchors in the two
			// dense carry buffers (round 6: with 87 000 reads a sub-batch - a 262 144-read call - these were what the anchor arenas, sized before them, left no room for;
			// the carry per read is what this context saw in its last calls, 128 KB before it has seen any)
			if (!in->samples_on_device && R) per_read += (size_t)((in->offsets[R] - in->offsets[0]) / R) * 2 + 64;
			per_read += 2 * (c->carry_per_read ? c->carry_per_read : ((size_t)128 << 10));
			size_t mine = holds_arenas(c) ? c->zbuf.cap + c->t1buf.cap + c->t2buf.cap + c->sx.cap + c->sy.cap + c->m_val.cap : 0;
			const uint64_t lim = c->mem_allow ? (uint64_t)((double)c->mem_allow / 3.0 / (double)per_read)
			                                  : (uint64_t)(((double)free_b / (c->share > 0 ? c->share : 1) + (double)mine) / 3.0 / (double)per_read);
			uint32_t cap = lim > 0xFFFFFFFFull ? 0xFFFFFFFFu : (lim < 4096 ? 4096u : (uint32_t)lim);
			if (const char *e = getenv("RH_CALL_READS_MAX")) { const uint32_t m = (uint32_t)strtoul(e, nullptr, 10); if (m && m < cap) cap = m; }   // (tests)
			call_cap = cap;                                             // one hipMemGetInfo snapshot: a limit of this call only, never remembered
		}
	}
	const uint32_t hint = c->slice_hint && c->slice_hint < call_cap ? c->slice_hint : call_cap;
	bool halved = false;
	if (R <= hint) {
		if (map_batch_once(c, mo, in, out, out_cap, &n) == 0) { *n_out = n; return 0; }
		if (!g_oom || R < 2) return -1;
		fprintf(stderr, "[rawhash_amd] device memory exhausted while mapping %u reads (%s): retrying in halves\n", R, rh_last_error());
		DevBuf *big[] = {&c->anc, &c->raw_anc, &c->zs, &c->prev_stage, &c->u, &c->ws, &c->sort_ws};
		(void)hipStreamSynchronize(c->stream);
		for (DevBuf *d : big) d->release();
		c->arena_room = 0;
		halved = true;
	} else slice = hint;
	rh_map_stats_t tot{};
	while (done < R) {
		const uint32_t m = R - done < slice ? R - done : slice;
		rh_read_batch_t b = *in;
		b.n_reads = m;
		b.offsets = in->offsets + done;
		if (in->cal_offset) b.cal_offset = in->cal_offset + done;
		if (in->cal_scale) b.cal_scale = in->cal_scale + done;
		if (in->name_rank) b.name_rank = in->name_rank + done;
		if (in->n_filtered) b.n_filtered = in->n_filtered + done;
		g_oom = false;
		if (map_batch_once(c, mo, &b, out + done, m, &n)) {
			if (!g_oom || slice < 2) return -1;
			slice /= 2;                                             // try smaller, from empty per-anchor arenas
			DevBuf *big[] = {&c->anc, &c->raw_anc, &c->zs, &c->prev_stage, &c->u, &c->ws, &c->sort_ws};
			(void)hipStreamSynchronize(c->stream);
			for (DevBuf *d : big) d->release();
			c->arena_room = 0;
			halved = true;
			continue;
		}
		for (uint32_t i = 0; i < m; ++i) out[done + i].read_idx += done;
		add_stats(tot, c->stats);
		done += m;
	}
	if (halved || (c->slice_hint && slice < c->slice_hint)) c->slice_hint = slice;   // only what an out-of-memory retry taught us sticks
	c->stats = tot;
	*n_out = R;
	return 0;
}

} // namespace

// Reads are independent, and after the first chunk round only the hard reads remain (latency-bound kernels that cannot
// fill the GPU).  Large batches are therefore split into contiguous sub-batches that run the identical per-round pipeline
// concurrently, each on its own HIP stream with its own arenas, so the rounds of different sub-batches overlap.
extern "C" int rh_map_batch(rh_ctx *c, const rh_mapopt_t *mo, const rh_read_batch_t *in, rh_map_record_t *out, uint64_t out_cap, uint64_t *n_out)
{
	for (auto &f : c->flight) if (f.ctx && !f.busy && holds_arenas(f.ctx)) release_arenas(f.ctx);   // idle batch slots give their arenas back
	const uint32_t R = in->n_reads;
	int n_sub = c->n_sub;
	while (n_sub > 1 && R / (uint32_t)n_sub < 2048u) --n_sub;
	// RMQ chaining: a launch lasts as long as its longest read (one wavefront per read, k_chain_rmq) and three streams of such launches did not overlap
	// (D. mel scale, 8 000 / 48 000 reads a call: 1 stream 2.11 k / 7.38 k reads/s, 2: 2.15 k / 7.02 k, 3: 1.18 k / 4.52 k; E. coli scale, 20 000: 29.0 k, 31.1 k, 18.0 k)
	if (((mo->flag & RH_M_RMQ) || mo->bw_long > mo->bw) && n_sub > 2 && !getenv("RH_SUB_BATCHES")) n_sub = 2;
	if (n_sub <= 1 || c->is_sub) { if (!c->is_sub) c->share = c->flight_mult; return map_batch_single(c, mo, in, out, out_cap, n_out); }
	*n_out = 0;
	if (need_index(c)) return -1;
	if (out_cap < R) { rh_set_error("output capacity %llu < %u reads", (unsigned long long)out_cap, R); return -1; }
	RH_HIP(hipSetDevice(c->device));
	while ((int)c->subs.size() < n_sub - 1) {
		struct CtxDel { void operator()(rh_ctx *p) const { rh_ctx_destroy(p); } };
		std::unique_ptr<rh_ctx, CtxDel> sc(new rh_ctx());
		sc->device = c->device; sc->is_sub = true; sc->n_sub = 1; sc->blob_owned = false; sc->logf_tab.owned = false;
		RH_HIP(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
		RH_HIP(hipEventCreate(&sc->e0));
		RH_HIP(hipEventCreate(&sc->e1));
		c->subs.push_back(sc.release());
	}
	for (rh_ctx *sc : c->subs) {	// borrow the resident index and the logf table
		sc->dix = c->dix; sc->have_index = true; sc->blob_owned = false;
		sc->akey_on = c->akey_on; sc->akey_lo = c->akey_lo; sc->akey_mid = c->akey_mid;
		sc->logf_tab.p = c->logf_tab.p; sc->logf_tab.cap = c->logf_tab.cap; sc->logf_tab.owned = false;
	}
	// what the sub-batches of this call may hold, in equal parts: the free memory and what their buffers hold already, less a reserve for the
	// runtime (kernel scratch, queues: it allocates at dispatch time and aborts the process when that fails)
	size_t allow = 0;
	{
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
			size_t held = 0;
			for (int g = 0; g < n_sub; ++g) held += ctx_bytes_held(g == 0 ? c : c->subs[g - 1]);
			const size_t reserve = total_b / 24 > ((size_t)3 << 30) ? total_b / 24 : ((size_t)3 << 30);
			const size_t avail = (free_b > reserve ? free_b - reserve : 0) / (size_t)(c->flight_mult > 0 ? c->flight_mult : 1) + held;   // (batches in flight share the free part)
			allow = avail / (size_t)n_sub;
		}
	}
	// (what an out-of-memory retry of an earlier call taught a sub-batch context - "map in slices of so many reads" - came from that call's sizes and from
	// what the other sub-batches happened to hold at that moment: it does not carry over, or one tight warm-up call leaves every later call mapping in crumbs)
	c->slice_hint = 0;
	for (rh_ctx *sc : c->subs) sc->slice_hint = 0;
	const auto t_begin = std::chrono::steady_clock::now();
	std::vector<int> rc(n_sub, 0);
	std::vector<std::string> err(n_sub);
	std::vector<std::thread> th;
	std::vector<uint32_t> lo(n_sub + 1);
	for (int g = 0; g <= n_sub; ++g) lo[g] = (uint32_t)((uint64_t)R * g / n_sub);
	for (int g = 0; g < n_sub; ++g)
		th.emplace_back([&, g]() {
			rh_ctx *lc = g == 0 ? c : c->subs[g - 1];
			rh_read_batch_t b = *in;
			b.n_reads = lo[g + 1] - lo[g];
			b.offsets = in->offsets + lo[g];
			if (in->cal_offset) b.cal_offset = in->cal_offset + lo[g];
			if (in->cal_scale) b.cal_scale = in->cal_scale + lo[g];
			if (in->name_rank) b.name_rank = in->name_rank + lo[g];
			if (in->n_filtered) b.n_filtered = in->n_filtered + lo[g];
			uint64_t n = 0;
			const bool saved = lc->is_sub;
			lc->is_sub = true;                                      // no further splitting
			lc->share = n_sub * c->flight_mult;                     // the device's memory is shared by the sub-batches (of every batch in flight)
			lc->mem_allow = allow;
			rc[g] = map_batch_single(lc, mo, &b, out + lo[g], b.n_reads, &n);
			lc->is_sub = saved; lc->mem_allow = 0;
			if (rc[g]) err[g] = rh_last_error();
			else for (uint32_t i = 0; i < b.n_reads; ++i) out[lo[g] + i].read_idx += lo[g];
		});
	for (auto &t : th) t.join();
	for (int g = 0; g < n_sub; ++g) if (rc[g]) { rh_set_error("sub-batch %d: %s", g, err[g].c_str()); return -1; }
	// merge the statistics of the sub-batches into this context's
	rh_map_stats_t tot = c->stats;
	for (int g = 1; g < n_sub; ++g) {	// (only the sub-batches of THIS call: earlier, larger batches may have created more)
		const rh_map_stats_t &q = c->subs[g - 1]->stats;
		tot.n_reads += q.n_reads; tot.n_chunks += q.n_chunks; tot.n_samples_raw += q.n_samples_raw; tot.n_samples_used += q.n_samples_used;
		tot.n_events += q.n_events; tot.n_seeds += q.n_seeds; tot.n_hits += q.n_hits; tot.n_anchors += q.n_anchors; tot.n_chained += q.n_chained;
		for (int i = 0; i < 24; ++i) { tot.ms_kernel[i] += q.ms_kernel[i]; tot.n_launch[i] += q.n_launch[i]; }
		for (int i = 0; i < 4; ++i) tot.n_rmq_class[i] += q.n_rmq_class[i];
		tot.n_dtw_device += q.n_dtw_device; tot.n_dtw_host += q.n_dtw_host;
	}
	tot.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
	c->stats = tot;
	*n_out = R;
	return 0;
}

// =================================================================================================== batches in flight
namespace {
// a context with its own stream and arenas that serves reads from `c`'s resident index
rh_ctx *borrow_ctx(rh_ctx *c)
{
	rh_ctx *b = new rh_ctx();
	b->device = c->device; b->blob_owned = false; b->logf_tab.owned = false; b->n_sub = (c->n_sub > 2 && !getenv("RH_SUB_BATCHES")) ? 2 : c->n_sub; b->flight_mult = RH_MAX_IN_FLIGHT;   // (two sub-batch streams per batch in flight: four streams together - measured, 12 500-read calls on the human index, upload-inclusive: 2 x 3 streams 21.2 k reads/s, 2 x 2 29.8 k)
	if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&b->e0) != hipSuccess || hipEventCreate(&b->e1) != hipSuccess) {
		rh_set_error("cannot create the stream of a batch slot"); rh_ctx_destroy(b); return nullptr;
	}
	return b;
}
void lend_index(rh_ctx *c, rh_ctx *b)
{
	b->dix = c->dix; b->have_index = c->have_index; b->blob_owned = false;
	b->akey_on = c->akey_on; b->akey_lo = c->akey_lo; b->akey_mid = c->akey_mid;
	b->logf_tab.p = c->logf_tab.p; b->logf_tab.cap = c->logf_tab.cap; b->logf_tab.owned = false;
	memcpy(b->header, c->header, sizeof(b->header));
}
} // namespace

extern "C" int rh_map_submit(rh_ctx *c, const rh_mapopt_t *mo, const rh_read_batch_t *in, rh_map_record_t *out, uint64_t out_cap, rh_ticket_t *ticket)
{
	if (need_index(c)) return -1;
	int slot = -1;
	for (int i = 0; i < RH_MAX_IN_FLIGHT; ++i) if (!c->flight[i].busy) { slot = i; break; }
	if (slot < 0) { rh_set_error("%d batches are in flight on this context already: rh_map_wait first", RH_MAX_IN_FLIGHT); return -1; }
	auto &f = c->flight[slot];
	if (f.th.joinable()) f.th.join();
	if (holds_arenas(c)) release_arenas(c);                         // arenas of earlier synchronous calls: the batches in flight need the memory
	if (!f.ctx && !(f.ctx = borrow_ctx(c))) return -1;
	lend_index(c, f.ctx);
	f.busy = true; f.rc = 0; f.n_out = 0; f.err.clear(); ++f.serial;
	const rh_mapopt_t mo_copy = *mo;
	const rh_read_batch_t in_copy = *in;
	rh_ctx *fc = f.ctx;
	auto *fp = &f;
	f.th = std::thread([fc, fp, mo_copy, in_copy, out, out_cap]() {
		uint64_t n = 0;
		fp->rc = rh_map_batch(fc, &mo_copy, &in_copy, out, out_cap, &n);
		fp->n_out = n;
		if (fp->rc) fp->err = rh_last_error();
	});
	ticket->slot = slot; ticket->serial = f.serial;
	return 0;
}

extern "C" int rh_map_wait(rh_ctx *c, rh_ticket_t t, uint64_t *n_out)
{
	if (t.slot < 0 || t.slot >= RH_MAX_IN_FLIGHT || !c->flight[t.slot].busy || c->flight[t.slot].serial != t.serial) { rh_set_error("rh_map_wait: no such batch in flight"); return -1; }
	auto &f = c->flight[t.slot];
	if (f.th.joinable()) f.th.join();
	f.busy = false;
	if (n_out) *n_out = f.n_out;
	c->stats = f.ctx->stats;
	if (f.rc) { rh_set_error("%s", f.err.c_str()); return -1; }
	return 0;
}

// a device-resident batch (rh_synth_reads_device) copied into host arrays (n_reads + 1 offsets)
extern "C" int rh_read_batch_to_host(rh_ctx *c, const rh_read_batch_t *dev, int16_t *samples, uint64_t *offsets, double *cal_offset, float *cal_scale)
{
	if (!dev->samples_on_device) { rh_set_error("rh_read_batch_to_host: the batch is not on the device"); return -1; }
	RH_HIP(hipSetDevice(c->device));
	const uint32_t n = dev->n_reads;
	RH_HIP(hipMemcpy(offsets, dev->offsets, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
	if (offsets[n]) RH_HIP(hipMemcpy(samples, dev->samples, (size_t)offsets[n] * 2, hipMemcpyDeviceToHost));
	if (n) { RH_HIP(hipMemcpy(cal_offset, dev->cal_offset, (size_t)n * 8, hipMemcpyDeviceToHost)); RH_HIP(hipMemcpy(cal_scale, dev->cal_scale, (size_t)n * 4, hipMemcpyDeviceToHost)); }
	return 0;
}

extern "C" void *rh_pinned_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, 0) != hipSuccess) { (void)hipGetLastError(); rh_set_error("cannot page-lock %zu bytes", bytes); return nullptr; }
	return p;
}
extern "C" void rh_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

// ---- RCCL, in process (north star: "RCCL over xGMI only to broadcast the index at load").  librccl.so.1 is loaded at run time, so the
// library has no link-time dependency on it; the entry points used are the NCCL API as rccl.h declares it (ncclCommInitAll,
// ncclGroupStart / End, ncclBroadcast, ncclCommDestroy).  One communicator per device of the contexts, one broadcast per <= 1 GiB piece
// (a 45 GB blob exceeds what a single collective takes as an element count on some builds), every rank's call inside one group.
namespace {
struct Rccl {
	void *h = nullptr;
	int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
	int (*CommDestroy)(void *comm) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Broadcast)(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t s) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	bool ok = false;
	Rccl()
	{
		for (const char *nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
		if (!h) return;
		CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
		GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
		Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast"); GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
		ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
	}
	const char *err(int rc) const { return GetErrorString ? GetErrorString(rc) : "?"; }
};
Rccl &rccl() { static Rccl r; return r; }
const int kNcclUint8 = 1;                                          // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

// broadcast bytes of buf[0] (device devs[0]) into buf[i] (device devs[i]) - distinct devices.  0 done, 1 RCCL not available (the caller
// falls back to peer copies), -1 failed (error set)
int rccl_bcast(int n, const int *devs, void *const *buf, uint64_t bytes)
{
	Rccl &R = rccl();
	if (!R.ok) return 1;
	std::vector<void*> comms((size_t)n, nullptr);
	int rc = R.CommInitAll(comms.data(), n, devs);
	if (rc) { rh_set_error("ncclCommInitAll over %d devices failed: %s", n, R.err(rc)); return -1; }
	std::vector<hipStream_t> st((size_t)n, nullptr);
	int bad = 0;
	for (int i = 0; i < n && !bad; ++i) if (hipSetDevice(devs[i]) != hipSuccess || hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) bad = 1;
	uint64_t piece = 1ull << 30;
	if (const char *e = getenv("RH_BCAST_PIECE_BYTES")) { const uint64_t v = strtoull(e, nullptr, 10); if (v) piece = v; }
	for (uint64_t at = 0; at < bytes && !bad; at += piece) {
		const uint64_t len = bytes - at < piece ? bytes - at : piece;
		if ((rc = R.GroupStart())) { bad = 1; break; }
		for (int i = 0; i < n; ++i) {
			(void)hipSetDevice(devs[i]);
			unsigned char *p = (unsigned char*)buf[i] + at;
			const int r2 = R.Broadcast(p, p, (size_t)len, kNcclUint8, 0, comms[i], st[i]);
			if (r2) rc = r2;
		}
		const int r3 = R.GroupEnd();
		if (rc || r3) { if (!rc) rc = r3; bad = 1; }
	}
	for (int i = 0; i < n; ++i) if (st[i]) { (void)hipSetDevice(devs[i]); if (hipStreamSynchronize(st[i]) != hipSuccess) bad = 1; (void)hipStreamDestroy(st[i]); }
	for (int i = 0; i < n; ++i) if (comms[i]) (void)R.CommDestroy(comms[i]);
	if (bad) { rh_set_error("RCCL broadcast of the index blob failed: %s", rc ? R.err(rc) : hipGetErrorString(hipGetLastError())); return -1; }
	return 0;
}
} // namespace

// what the last rh_index_bcast of this process went through: 0 nothing yet, 1 RCCL (ncclBroadcast), 2 peer copies (hipMemcpyPeerAsync tree)
static std::atomic<int> g_bcast_path{0};
extern "C" int rh_index_bcast_path(void) { return g_bcast_path.load(); }

// librccl.so.1 loads, its entry points resolve and a one-rank communicator broadcasts in place on this context's device: what a one-GPU
// box can check of the RCCL path (returns 0, or -1 with the reason)
extern "C" int rh_rccl_selftest(rh_ctx *c)
{
	if (!rccl().ok) { rh_set_error("librccl.so.1 could not be loaded (or lacks the NCCL entry points)"); return -1; }
	RH_HIP(hipSetDevice(c->device));
	void *buf = nullptr;
	RH_HIP(hipMalloc(&buf, 1 << 20));
	RH_HIP(hipMemset(buf, 0x5A, 1 << 20));
	const int dev = c->device;
	const int rc = rccl_bcast(1, &dev, &buf, 1 << 20);
	unsigned char probe[2] = {0, 0};
	(void)hipMemcpy(probe, (unsigned char*)buf + (1 << 20) - 2, 2, hipMemcpyDeviceToHost);
	(void)hipFree(buf);
	if (rc) { if (rc > 0) rh_set_error("RCCL not available"); return -1; }
	if (probe[0] != 0x5A || probe[1] != 0x5A) { rh_set_error("RCCL self-test: the buffer changed under a one-rank broadcast"); return -1; }
	return 0;
}

extern "C" int rh_index_bcast(rh_ctx *const *ctxs, int n)
{
	if (n < 1 || need_index(ctxs[0])) return -1;
	for (int i = 0; i < n; ++i) if (index_replaceable(ctxs[i], "rh_index_bcast")) return -1;
	rh_ctx *src = ctxs[0];
	BlobHeader h; memcpy(&h, src->header, sizeof(h));
	for (int i = 1; i < n; ++i) {
		rh_ctx *d = ctxs[i];
		RH_HIP(hipSetDevice(d->device));
		if (d->blob_owned) d->blob.release(); else { d->blob.p = nullptr; d->blob.cap = 0; }
		d->blob_owned = true; d->have_index = false;
		if (d->blob.ensure(h.bytes, false)) return -1;
	}
	// RCCL first (one ncclBroadcast per piece over the ring / tree RCCL builds on the xGMI links) when every context sits on its own
	// device; RH_BCAST=peer, contexts sharing a device (tests on a one-GPU box) or a missing librccl take the peer-copy tree below
	{
		bool distinct = n >= 2;
		for (int i = 0; i < n && distinct; ++i) for (int j = 0; j < i; ++j) if (ctxs[i]->device == ctxs[j]->device) { distinct = false; break; }
		const char *mode = getenv("RH_BCAST");
		if (distinct && !(mode && !strcmp(mode, "peer"))) {
			std::vector<int> devs((size_t)n); std::vector<void*> bufs((size_t)n);
			for (int i = 0; i < n; ++i) { devs[i] = ctxs[i]->device; bufs[i] = ctxs[i]->blob.p; }
			const int r = rccl_bcast(n, devs.data(), bufs.data(), h.bytes);
			if (r < 0) return -1;
			if (r == 0) {
				for (int i = 1; i < n; ++i) { RH_HIP(hipSetDevice(ctxs[i]->device)); if (bind_blob(ctxs[i], h)) return -1; }
				RH_HIP(hipSetDevice(src->device));
				g_bcast_path.store(1);
				return 0;
			}
		}
	}
	// doubling tree 0 -> 1, {0,1} -> {2,3}, {0..3} -> {4..7}: every GPU that holds the blob feeds one that does not, each copy over
	// its own point-to-point xGMI link, ceil(log2 n) rounds of one blob time instead of n - 1 copies out of GPU 0
	g_bcast_path.store(2);
	std::vector<hipStream_t> st((size_t)n, nullptr);
	int rc = 0;
	for (int have = 1; have < n && !rc; have *= 2) {
		const int m = have < n - have ? have : n - have;
		for (int i = 0; i < m && !rc; ++i) {
			rh_ctx *a = ctxs[i], *b = ctxs[have + i];
			if (hipSetDevice(a->device) != hipSuccess) { rc = -1; break; }
			if (!st[i] && hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) { rc = -1; break; }
			if (hipMemcpyPeerAsync(b->blob.p, b->device, a->blob.p, a->device, h.bytes, st[i]) != hipSuccess) rc = -1;
		}
		for (int i = 0; i < m; ++i) if (st[i]) { (void)hipSetDevice(ctxs[i]->device); if (hipStreamSynchronize(st[i]) != hipSuccess) rc = -1; }
	}
	for (int i = 0; i < n; ++i) if (st[i]) { (void)hipSetDevice(ctxs[i]->device); (void)hipStreamDestroy(st[i]); }
	if (rc) { rh_set_error("rh_index_bcast: a peer copy of the index blob failed: %s", hipGetErrorString(hipGetLastError())); return -1; }
	for (int i = 1; i < n; ++i) { RH_HIP(hipSetDevice(ctxs[i]->device)); if (bind_blob(ctxs[i], h)) return -1; }
	RH_HIP(hipSetDevice(src->device));
	return 0;
}

// =================================================================================================== stage-level calls
namespace {

// identity active list 0..n-1 on the device
int make_identity(rh_ctx *c, uint32_t n, int slot)
{
	std::vector<uint32_t> id(n ? n : 1);
	for (uint32_t i = 0; i < n; ++i) id[i] = i;
	if (c->act[slot].ensure((size_t)(n ? n : 1) * 4)) return -1;
	if (n) RH_HIP(hipMemcpy(c->act[slot].p, id.data(), (size_t)n * 4, hipMemcpyHostToDevice));
	return 0;
}

template <class T> int d2h(std::vector<T> &dst, const void *src, size_t n) { dst.resize(n ? n : 1); if (n) RH_HIP(hipMemcpy(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost)); return 0; }
template <class T> int h2d(void *dst, const T *src, size_t n) { if (n) RH_HIP(hipMemcpy(dst, src, n * sizeof(T), hipMemcpyHostToDevice)); return 0; }

// state arrays for stage calls that do not start from raw signal
int stage_state_only(rh_ctx *c, uint32_t R, rh_dev_reads *rd)
{
	rh_read_batch_t empty{};
	std::vector<uint64_t> off((size_t)R + 1, 0);
	empty.n_reads = R; empty.offsets = off.data(); empty.samples = nullptr;
	if (stage_reads(c, &empty, rd)) return -1;
	const size_t n = R ? R : 1;
	RH_HIP(hipMemset(rd->ev_off, 0, n * 4)); RH_HIP(hipMemset(rd->n_prev, 0, n * 4)); RH_HIP(hipMemset(rd->prev_off, 0, n * 8));
	RH_HIP(hipMemset(rd->done, 0, n)); RH_HIP(hipMemset(rd->ls_ncregs, 0, n * 4));
	return 0;
}

} // namespace

extern "C" int rh_events_batch(rh_ctx *c, const rh_mapopt_t *mo, const rh_read_batch_t *in, uint32_t chunk, float *events, uint64_t events_cap, uint64_t *ev_offsets, uint32_t *l_sig)
{
	RH_HIP(hipSetDevice(c->device));
	rh_mapopt_t m2 = *mo;
	m2.max_num_chunk = chunk + 1 > 32u ? chunk + 1 : 32u; m2.flag = 0; m2.bw_long = 0;
	c->cs_stride = m2.max_num_chunk + 1;
	rh_dev_opt o;
	if (fill_dev_opt(c, &m2, &o)) return -1;
	// (this stage-level call runs the LDS-resident chunk kernels with a chunk's row strides, whatever the context mapped last)
	if (m2.chunk_size > RH_CHUNK_MAX) { rh_set_error("rh_events_batch: chunks of more than %d samples are only supported through rh_map_batch", RH_CHUNK_MAX); return -1; }
	if (set_row_strides(c, &m2, in)) return -1;
	o.min_events = 0;
	const uint32_t R = in->n_reads;
	hipStream_t s = c->stream;
	rh_dev_reads rd;
	if (stage_reads(c, in, &rd)) return -1;
	if (c->act[0].ensure((size_t)(R ? R : 1) * 4) || c->act[1].ensure((size_t)(R ? R : 1) * 4) || c->n_act_dev.ensure(64) || c->counters.ensure(16 * 8)) return -1;
	rhk_prefilter(s, o, rd);
	std::vector<uint32_t> act_h; std::vector<uint32_t> nev_h; std::vector<float> ev_h;
	uint32_t n_act = 0;
	for (uint32_t cc = 0; cc <= chunk; ++cc) {
		rhk_compact_active(s, o, rd, nullptr, R, cc, c->act[0].as<uint32_t>(), c->n_act_dev.as<uint32_t>());
		RH_HIP(hipMemcpyAsync(&n_act, c->n_act_dev.p, 4, hipMemcpyDeviceToHost, s));
		RH_HIP(hipStreamSynchronize(s));
		if (n_act == 0) break;
		rh_dev_round rr{};
		if (stage_round(c, n_act, &rr)) return -1;
		rr.act = c->act[0].as<uint32_t>(); rr.chunk = cc;
		rhk_events(s, o, rd, rr);
		RH_HIP(hipStreamSynchronize(s));
		if (cc == chunk) { if (d2h(act_h, rr.act, n_act) || d2h(nev_h, rr.n_ev, n_act) || d2h(ev_h, rr.ev, (size_t)n_act * RH_EV_CAP)) return -1; }
	}
	RH_HIP(hipGetLastError());
	if (l_sig && R) RH_HIP(hipMemcpy(l_sig, rd.l_sig, (size_t)R * 4, hipMemcpyDeviceToHost));
	std::vector<int64_t> slot(R, -1);
	if (act_h.size() >= n_act) for (uint32_t a = 0; a < n_act && !nev_h.empty(); ++a) slot[act_h[a]] = a;
	uint64_t k = 0;
	ev_offsets[0] = 0;
	for (uint32_t r = 0; r < R; ++r) {
		if (slot[r] >= 0) {
			const uint32_t ne = nev_h[slot[r]];
			if (k + ne > events_cap) { rh_set_error("events buffer too small"); return -1; }
			memcpy(events + k, ev_h.data() + (size_t)slot[r] * RH_EV_CAP, (size_t)ne * 4);
			k += ne;
		}
		ev_offsets[r + 1] = k;
	}
	return 0;
}

extern "C" int rh_sketch_batch(rh_ctx *c, uint32_t R, const float *events, const uint64_t *ev_offsets, rh_mm128_t *seeds, uint64_t seeds_cap, uint64_t *seed_offsets)
{
	if (need_index(c)) return -1;
	RH_HIP(hipSetDevice(c->device));
	rh_dev_opt o{};
	rh_dev_reads rd;
	if (stage_state_only(c, R, &rd)) return -1;
	rh_dev_round rr{};
	if (stage_round(c, R, &rr) || make_identity(c, R, 0)) return -1;
	rr.act = c->act[0].as<uint32_t>();
	std::vector<float> evp((size_t)(R ? R : 1) * RH_EV_CAP, 0.0f); std::vector<uint32_t> nev(R ? R : 1, 0); std::vector<uint8_t> skip(R ? R : 1, 0);
	for (uint32_t r = 0; r < R; ++r) {
		const uint64_t ne = ev_offsets[r + 1] - ev_offsets[r];
		if (ne > RH_EV_CAP) { rh_set_error("read %u has %llu events (> %d per chunk)", r, (unsigned long long)ne, RH_EV_CAP); return -1; }
		memcpy(&evp[(size_t)r * RH_EV_CAP], events + ev_offsets[r], ne * 4); nev[r] = (uint32_t)ne;
	}
	if (h2d(rr.ev, evp.data(), evp.size()) || h2d(rr.n_ev, nev.data(), R) || h2d(rr.skip, skip.data(), R)) return -1;
	RH_HIP(hipMemset(rr.counters, 0, 16 * 8));
	rhk_sketch(c->stream, o, c->dix, rd, rr);
	RH_HIP(hipStreamSynchronize(c->stream));
	RH_HIP(hipGetLastError());
	std::vector<uint64_t> sx, sy; std::vector<uint32_t> ns;
	if (d2h(sx, rr.sx, (size_t)R * RH_EV_CAP) || d2h(sy, rr.sy, (size_t)R * RH_EV_CAP) || d2h(ns, rr.n_seed, R)) return -1;
	uint64_t k = 0;
	seed_offsets[0] = 0;
	for (uint32_t r = 0; r < R; ++r) {
		if (k + ns[r] > seeds_cap) { rh_set_error("seed buffer too small"); return -1; }
		for (uint32_t i = 0; i < ns[r]; ++i) { seeds[k].x = sx[(size_t)r * RH_EV_CAP + i]; seeds[k].y = sy[(size_t)r * RH_EV_CAP + i]; ++k; }
		seed_offsets[r + 1] = k;
	}
	return 0;
}

extern "C" int rh_seed_batch(rh_ctx *c, const rh_mapopt_t *mo, uint32_t R, const rh_mm128_t *seeds, const uint64_t *seed_offsets, const uint32_t *q_offset,
                             const rh_mm128_t *prev, const uint64_t *prev_offsets, rh_mm128_t *anchors, uint64_t anchors_cap, uint64_t *anchor_offsets, int32_t *rep_len)
{
	if (need_index(c)) return -1;
	RH_HIP(hipSetDevice(c->device));
	rh_mapopt_t m2 = *mo; m2.flag = 0; m2.bw_long = 0;
	rh_dev_opt o;
	if (fill_dev_opt(c, &m2, &o)) return -1;
	rh_dev_reads rd;
	if (stage_state_only(c, R, &rd)) return -1;
	rh_dev_round rr{};
	if (stage_round(c, R, &rr) || make_identity(c, R, 0)) return -1;
	rr.act = c->act[0].as<uint32_t>();
	const size_t n = R ? R : 1;
	std::vector<uint64_t> sx(n * RH_EV_CAP, 0), sy(n * RH_EV_CAP, 0); std::vector<uint32_t> ns(n, 0); std::vector<uint8_t> skip(n, 0);
	for (uint32_t r = 0; r < R; ++r) {
		const uint64_t m = seed_offsets[r + 1] - seed_offsets[r];
		if (m > RH_EV_CAP) { rh_set_error("read %u has %llu seeds (> %d per chunk)", r, (unsigned long long)m, RH_EV_CAP); return -1; }
		for (uint64_t i = 0; i < m; ++i) { sx[(size_t)r * RH_EV_CAP + i] = seeds[seed_offsets[r] + i].x; sy[(size_t)r * RH_EV_CAP + i] = seeds[seed_offsets[r] + i].y; }
		ns[r] = (uint32_t)m;
	}
	if (h2d(rr.sx, sx.data(), sx.size()) || h2d(rr.sy, sy.data(), sy.size()) || h2d(rr.n_seed, ns.data(), R) || h2d(rr.skip, skip.data(), R)) return -1;
	if (q_offset && h2d(rd.ev_off, q_offset, R)) return -1;
	uint64_t n_prev_total = 0;
	if (prev_offsets) {
		std::vector<uint32_t> np(n, 0);
		for (uint32_t r = 0; r < R; ++r) np[r] = (uint32_t)(prev_offsets[r + 1] - prev_offsets[r]);
		n_prev_total = prev_offsets[R];
		if (c->carry[1].ensure((n_prev_total ? n_prev_total : 1) * 16)) return -1;
		if (h2d(c->carry[1].p, prev, n_prev_total) || h2d(rd.n_prev, np.data(), R) || h2d(rd.prev_off, prev_offsets, R)) return -1;
	} else if (c->carry[1].ensure(16)) return -1;
	rr.prev_in = c->carry[1].as<rh_mm128_t>();
	RH_HIP(hipMemset(rr.counters, 0, 16 * 8));
	hipStream_t s = c->stream;
	rhk_probe(s, o, c->dix, rd, rr);
	rhk_scan_anchors(s, rd, rr);
	uint64_t total = 0;
	RH_HIP(hipMemcpyAsync(&total, rr.a_off + R, 8, hipMemcpyDeviceToHost, s));
	RH_HIP(hipStreamSynchronize(s));
	if (stage_anchors(c, total, &rr)) return -1;
	rr.prev_in = c->carry[1].as<rh_mm128_t>();
	rhk_expand(s, o, c->dix, rd, rr);
	if (rhk_sort(s, c->dix, rr, [&](const uint8_t *mask) { rhk_expand(s, o, c->dix, rd, rr, mask); return 0; })) return -1;
	RH_HIP(hipStreamSynchronize(s));
	RH_HIP(hipGetLastError());
	if (total > anchors_cap) { rh_set_error("anchor buffer too small (%llu needed)", (unsigned long long)total); return -1; }
	RH_HIP(hipMemcpy(anchor_offsets, rr.a_off, (size_t)(R + 1) * 8, hipMemcpyDeviceToHost));
	if (total) RH_HIP(hipMemcpy(anchors, rr.anc, total * 16, hipMemcpyDeviceToHost));
	if (rep_len && R) RH_HIP(hipMemcpy(rep_len, rr.rep_len, (size_t)R * 4, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int rh_chain_batch(rh_ctx *c, const rh_mapopt_t *mo, uint32_t R, const rh_mm128_t *anchors, const uint64_t *anchor_offsets, rh_mm128_t *chained, uint64_t chained_cap,
                              uint64_t *chained_offsets, uint64_t *u, uint64_t u_cap, uint64_t *u_offsets, rh_mm128_t *prev_out)
{
	if (need_index(c)) return -1;
	RH_HIP(hipSetDevice(c->device));
	rh_mapopt_t m2 = *mo; m2.flag = mo->flag & RH_M_RMQ;   // (the chaining variant stays: --rmq / --bw-long at stage level)
	rh_dev_opt o;
	if (fill_dev_opt(c, &m2, &o)) return -1;
	rh_dev_reads rd;
	if (stage_state_only(c, R, &rd)) return -1;
	rh_dev_round rr{};
	if (stage_round(c, R, &rr) || make_identity(c, R, 0)) return -1;
	rr.act = c->act[0].as<uint32_t>();
	const uint64_t total = anchor_offsets[R];
	for (uint32_t r = 0; r < R; ++r) { const uint64_t m = anchor_offsets[r + 1] - anchor_offsets[r]; if (m > rr.max_anchors) rr.max_anchors = (uint32_t)m; }
	if (stage_anchors(c, total, &rr)) return -1;
	std::vector<uint8_t> skip(R ? R : 1, 0);
	if (h2d(rr.a_off, anchor_offsets, (size_t)R + 1) || h2d(rr.anc, anchors, total) || h2d(rr.skip, skip.data(), R)) return -1;
	RH_HIP(hipMemset(rr.counters, 0, 16 * 8));
	RH_HIP(hipMemset(rr.n_u, 0, (size_t)(R ? R : 1) * 4)); RH_HIP(hipMemset(rr.n_v, 0, (size_t)(R ? R : 1) * 4));
	hipStream_t s = c->stream;
	if (chain_stages(c, s, o, rd, rr, false)) return -1;
	RH_HIP(hipStreamSynchronize(s));
	RH_HIP(hipGetLastError());
	std::vector<uint32_t> nu, nv; std::vector<rh_mm128_t> an, pv; std::vector<uint64_t> uu;
	if (d2h(nu, rr.n_u, R) || d2h(nv, rr.n_v, R) || d2h(an, rr.anc, total) || d2h(pv, rr.prev_out, total) || d2h(uu, rr.u, total)) return -1;
	uint64_t k = 0, ku = 0;
	chained_offsets[0] = 0; u_offsets[0] = 0;
	for (uint32_t r = 0; r < R; ++r) {
		const uint64_t b = anchor_offsets[r];
		if (k + nv[r] > chained_cap || ku + nu[r] > u_cap) { rh_set_error("chain output buffers too small"); return -1; }
		for (uint32_t i = 0; i < nv[r]; ++i) { chained[k + i] = an[b + i]; if (prev_out) prev_out[k + i] = pv[b + i]; }
		for (uint32_t i = 0; i < nu[r]; ++i) u[ku + i] = uu[b + i];
		k += nv[r]; ku += nu[r];
		chained_offsets[r + 1] = k; u_offsets[r + 1] = ku;
	}
	return 0;
}

extern "C" int rh_regions_batch(rh_ctx *c, const rh_mapopt_t *mo, uint32_t R, const rh_mm128_t *anchors, const uint64_t *anchor_offsets,
                                const int32_t *rep_len, const uint32_t *qlen, int32_t *summary /* R x 10 */,
                                int32_t *regs /* 18 per kept region, may be NULL */, uint64_t regs_cap, uint64_t *reg_offsets /* R + 1 */)
{
	// chain -> backtrack -> compact (as rh_chain_batch) and then the region stage of a round: keys + sort (hit.c:111-126), parents,
	// secondaries dropped, MAPQ (hit.c:195-367, 502-539); summary[r] = {n_cregs, cnt, score, mapq, qs, qe, rs, re, rid, rev} of region 0
	if (need_index(c)) return -1;
	RH_HIP(hipSetDevice(c->device));
	rh_mapopt_t m2 = *mo; m2.flag = mo->flag & RH_M_RMQ;   // (the chaining variant stays: --rmq / --bw-long at stage level)
	rh_dev_opt o;
	if (fill_dev_opt(c, &m2, &o)) return -1;
	o.flag |= mo->flag & (RH_M_ALL_CHAINS | RH_M_HARD_MLEVEL);   // region-stage switches: mm_select_sub skipped (rmap.cpp:353), hit.c:136's hard mask level
	rh_dev_reads rd;
	if (stage_state_only(c, R, &rd)) return -1;
	rh_dev_round rr{};
	if (stage_round(c, R, &rr) || make_identity(c, R, 0)) return -1;
	rr.act = c->act[0].as<uint32_t>();
	const uint64_t total = anchor_offsets[R];
	for (uint32_t r = 0; r < R; ++r) { const uint64_t m = anchor_offsets[r + 1] - anchor_offsets[r]; if (m > rr.max_anchors) rr.max_anchors = (uint32_t)m; }
	if (stage_anchors(c, total, &rr)) return -1;
	std::vector<uint8_t> skip(R ? R : 1, 0);
	if (h2d(rr.a_off, anchor_offsets, (size_t)R + 1) || h2d(rr.anc, anchors, total) || h2d(rr.skip, skip.data(), R)) return -1;
	if (h2d(rr.rep_len, rep_len, R) || h2d(rr.n_ev, qlen, R)) return -1;       // (ev_off = 0: the hash seed is offset + n_events, rmap.cpp:346)
	RH_HIP(hipMemset(rr.counters, 0, 16 * 8));
	RH_HIP(hipMemset(rr.n_u, 0, (size_t)(R ? R : 1) * 4)); RH_HIP(hipMemset(rr.n_v, 0, (size_t)(R ? R : 1) * 4));
	hipStream_t s = c->stream;
	if (chain_stages(c, s, o, rd, rr, false)) return -1;
	if (rhk_regions_sort(s, o, rd, rr)) return -1;
	DevBuf reg_all;                                                  // every kept region, where the kernels leave it: read r's k-th at (anchor_offsets[r] + k) * 18
	struct Rel { DevBuf &b; ~Rel() { b.release(); } } rel{reg_all};
	if (regs) { if (reg_all.ensure((size_t)(total ? total : 1) * 18 * 4, false)) return -1; rr.reg_out = reg_all.as<int32_t>(); }
	rhk_regions(s, o, rd, rr, c->logf_tab.as<float>());
	RH_HIP(hipStreamSynchronize(s));
	RH_HIP(hipGetLastError());
	std::vector<int32_t> f[10];
	const int32_t *src[10] = { rd.ls_ncregs, rd.ls_cnt, rd.ls_score, rd.ls_mapq, rd.ls_qs, rd.ls_qe, rd.ls_rs, rd.ls_re, rd.ls_rid, rd.ls_rev };
	for (int k = 0; k < 10; ++k) if (d2h(f[k], src[k], R)) return -1;
	for (uint32_t r = 0; r < R; ++r) for (int k = 0; k < 10; ++k) summary[(size_t)r * 10 + k] = f[k][r];
	if (regs) {
		std::vector<int32_t> all;
		if (d2h(all, reg_all.as<int32_t>(), (size_t)total * 18)) return -1;
		uint64_t n = 0;
		for (uint32_t r = 0; r < R; ++r) {
			reg_offsets[r] = n;
			const uint64_t k = (uint64_t)(f[0][r] > 0 ? f[0][r] : 0);
			if (n + k > regs_cap) { rh_set_error("rh_regions_batch: more than %llu kept regions", (unsigned long long)regs_cap); return -1; }
			if (k) memcpy(regs + n * 18, all.data() + anchor_offsets[r] * 18, (size_t)k * 18 * 4);
			n += k;
		}
		reg_offsets[R] = n;
	}
	return 0;
}

extern "C" int rh_sort128x_batch(rh_ctx *c, uint32_t n_seg, rh_mm128_t *a, const uint64_t *offsets)
{
	// runs the production sort stage (rhk_sort: LDS block sort fast/exact passes + oversized fallback) on free-standing segments
	RH_HIP(hipSetDevice(c->device));
	const uint64_t total = n_seg ? offsets[n_seg] : 0;
	rh_dev_round rr{};
	if (stage_round(c, n_seg, &rr) || stage_anchors(c, total, &rr)) return -1;
	std::vector<uint8_t> skip(n_seg ? n_seg : 1, 0);
	if (h2d(rr.raw, a, total) || h2d(rr.a_off, offsets, (size_t)n_seg + 1) || h2d(rr.skip, skip.data(), n_seg)) return -1;
	// (as the round loop runs it: long segments in any order first, those that hold equal keys again - from the caller's records - with the exact passes)
	if (rhk_sort(c->stream, c->dix, rr, [&](const uint8_t *) { return h2d(rr.raw, a, total); })) return -1;
	RH_HIP(hipStreamSynchronize(c->stream));
	RH_HIP(hipGetLastError());
	if (total) RH_HIP(hipMemcpy(a, rr.anc, total * 16, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int rh_sort128x_packed_batch(rh_ctx *c, uint32_t n_seg, rh_mm128_t *a, const uint64_t *offsets, uint32_t lo_bits, uint32_t mid_bits, int any_order)
{
	// the anchor sort of the round loop on one-word records (rh_rec_fmt): key' << shift | payload
	RH_HIP(hipSetDevice(c->device));
	const uint64_t total = n_seg ? offsets[n_seg] : 0;
	const uint32_t kb = 1u + lo_bits + mid_bits;
	if (lo_bits > 30u || mid_bits > 24u || kb > 56u) { rh_set_error("packed sort: %u + %u key bits do not leave a payload", lo_bits, mid_bits); return -1; }
	const uint32_t shift = 64u - kb;
	std::vector<uint64_t> w((size_t)(total ? total : 1));
	for (uint64_t i = 0; i < total; ++i) {
		if ((a[i].x & 0x7FFFFFFF00000000ull) >> 32 >> mid_bits || (a[i].x & 0xFFFFFFFFull) >> lo_bits || a[i].y >> shift) { rh_set_error("packed sort: record %llu does not fit %u / %u key bits and a %u-bit payload", (unsigned long long)i, lo_bits, mid_bits, shift); return -1; }
		w[i] = rh_rec8_pack_key(a[i].x, lo_bits, mid_bits) << shift | a[i].y;
	}
	rh_dev_round rr{};
	uint32_t mx = 0;
	for (uint32_t s = 0; s < n_seg; ++s) if (offsets[s + 1] - offsets[s] > mx) mx = (uint32_t)(offsets[s + 1] - offsets[s]);
	rr.max_anchors = mx;
	if (stage_round(c, n_seg, &rr) || stage_anchors(c, total, &rr)) return -1;
	rr.afmt = rh_rec_fmt{1, (uint8_t)shift, (uint8_t)lo_bits, (uint8_t)mid_bits, 0};
	rr.akey_on = kb <= 32u ? 1 : 0; rr.akey_lo = (uint8_t)lo_bits; rr.akey_mid = (uint8_t)mid_bits;
	std::vector<uint8_t> skip(n_seg ? n_seg : 1, 0);
	uint64_t *d_raw = reinterpret_cast<uint64_t*>(rr.raw);
	if (h2d(d_raw, w.data(), total) || h2d(rr.a_off, offsets, (size_t)n_seg + 1) || h2d(rr.skip, skip.data(), n_seg)) return -1;
	std::function<int(const uint8_t*)> again;
	if (any_order) again = [&](const uint8_t *) { return h2d(d_raw, w.data(), total); };
	if (rhk_sort(c->stream, c->dix, rr, again)) return -1;
	RH_HIP(hipStreamSynchronize(c->stream));
	RH_HIP(hipGetLastError());
	if (total) RH_HIP(hipMemcpy(w.data(), rr.anc, total * 8, hipMemcpyDeviceToHost));
	for (uint64_t i = 0; i < total; ++i) { a[i].x = rh_rec8_key(w[i], shift, lo_bits, mid_bits); a[i].y = w[i] & ((1ull << shift) - 1ull); }
	return 0;
}

extern "C" int rh_sort128x_any_batch(rh_ctx *c, uint32_t n_seg, rh_mm128_t *a, const uint64_t *offsets, uint8_t *has_ties)
{
	// the sorter's path for keys that are almost never equal (region keys, hit.c:111-126): long segments are placed level by level in
	// any order, then checked for equal neighbours; has_ties[s] = 1 for the long segments the caller would redo with the exact passes
	RH_HIP(hipSetDevice(c->device));
	const uint64_t total = n_seg ? offsets[n_seg] : 0;
	rh_dev_round rr{};
	if (stage_round(c, n_seg, &rr) || stage_anchors(c, total, &rr)) return -1;
	std::vector<uint8_t> skip(n_seg ? n_seg : 1, 0);
	if (h2d(rr.raw, a, total) || h2d(rr.a_off, offsets, (size_t)n_seg + 1) || h2d(rr.skip, skip.data(), n_seg)) return -1;
	rh_sort_job jb = { rr.n_act, rr.skip, rr.a_off, nullptr, rr.raw, rr.anc, rr.need_exact2, rr.ws, RH_WS_PER_ANCHOR, 0, 0, 0, 0, 0, rr.max_anchors };
	jb.big_alt = rr.zs; jb.big_ws = rr.sort_ws; jb.big_ws_bytes = rr.sort_ws_bytes; jb.big_pin = rr.sort_pin; jb.big_total = rr.sort_total;
	uint32_t n_redo = 0;
	jb.any_order = 1; jb.redo_skip = rr.need_exact; jb.n_redo = &n_redo;
	RH_HIP(hipMemset(rr.need_exact, 1, n_seg ? n_seg : 1));
	if (rhk_sort_job(c->stream, jb, false, 0u)) return -1;
	RH_HIP(hipStreamSynchronize(c->stream));
	RH_HIP(hipGetLastError());
	if (total) RH_HIP(hipMemcpy(a, rr.anc, total * 16, hipMemcpyDeviceToHost));
	std::vector<uint8_t> rs;
	if (d2h(rs, rr.need_exact, n_seg)) return -1;
	uint32_t cnt = 0;
	for (uint32_t i = 0; i < n_seg; ++i) { has_ties[i] = rs[i] ? 0 : 1; cnt += has_ties[i]; }
	if (cnt != n_redo) { rh_set_error("any-order sort: %u segments flagged, %u counted", cnt, n_redo); return -1; }
	return 0;
}

// =================================================================================================== synthetic batch in HBM
extern "C" int rh_synth_reads_device(rh_ctx *c, const rh_synth_cfg_t *cfg, const char *model_path, uint64_t first, uint32_t n, rh_read_batch_t *out)
{
	RH_HIP(hipSetDevice(c->device));
	std::vector<int32_t> level16;
	if (rh_synth_level_table(cfg, model_path, level16) < 0) return -1;
	const size_t nn = n ? n : 1;
	if (c->sy_samples.ensure(nn * cfg->n_samples * 2) || c->sy_off.ensure((nn + 1) * 8) || c->sy_cal_off.ensure(nn * 8) || c->sy_cal_scale.ensure(nn * 4) || c->sy_levels.ensure(level16.size() * 4)) return -1;
	RH_HIP(hipMemcpy(c->sy_levels.p, level16.data(), level16.size() * 4, hipMemcpyHostToDevice));
	rhk_synth_reads(c->stream, *cfg, c->sy_levels.as<int32_t>(), rh_synth_model_k(level16.size()), first, n, c->sy_samples.as<int16_t>(), c->sy_off.as<uint64_t>(), c->sy_cal_off.as<double>(), c->sy_cal_scale.as<float>());
	RH_HIP(hipStreamSynchronize(c->stream));
	RH_HIP(hipGetLastError());
	memset(out, 0, sizeof(*out));
	out->n_reads = n;
	out->samples = c->sy_samples.as<int16_t>(); out->offsets = c->sy_off.as<uint64_t>();
	out->cal_offset = c->sy_cal_off.as<double>(); out->cal_scale = c->sy_cal_scale.as<float>();
	out->samples_on_device = 1;
	return 0;
}

// All-vs-all (RH_M_ALL_CHAINS) returns as many records per read as chains it reports (rmap.cpp:557-586), else one: the records
// of read r are out[rec_offsets[r] .. rec_offsets[r + 1]).  One round over whole reads on the context's stream.
extern "C" int rh_map_batch_multi(rh_ctx *c, const rh_mapopt_t *mo, const rh_read_batch_t *in, rh_map_record_t *out, uint64_t out_cap,
                                  uint64_t *rec_offsets, uint64_t *n_out)
{
	for (auto &f : c->flight) if (f.ctx && !f.busy && holds_arenas(f.ctx)) release_arenas(f.ctx);
	c->share = c->flight_mult;
	std::vector<uint64_t> tmp;
	if (!rec_offsets) { tmp.resize((size_t)in->n_reads + 1); rec_offsets = tmp.data(); }
	if (!(mo->flag & RH_M_ALL_CHAINS)) {	// one record per read: the ordinary path, offsets 0 .. n
		if (rh_map_batch(c, mo, in, out, out_cap, n_out)) return -1;
		for (uint32_t r = 0; r <= in->n_reads; ++r) rec_offsets[r] = r;
		return 0;
	}
	// A whole-read round holds rows as long as the longest read of the call for every read of the call (~34 bytes per sample of
	// row): the batch goes through in consecutive groups of reads whose rows fit a quarter of the free memory (one group unless
	// the lengths are very uneven or the batch is huge).  Reads are independent: the records are the same.
	const uint32_t R = in->n_reads;
	*n_out = 0; rec_offsets[0] = 0;
	if (R == 0) return 0;
	if (hipSetDevice(c->device) != hipSuccess) { rh_set_error("hipSetDevice failed"); return -1; }
	std::vector<uint64_t> off_h;
	const uint64_t *off = in->offsets;
	if (in->samples_on_device) { off_h.resize((size_t)R + 1); RH_HIP(hipMemcpy(off_h.data(), in->offsets, ((size_t)R + 1) * 8, hipMemcpyDeviceToHost)); off = off_h.data(); }
	size_t free_b = 0, total_b = 0;
	RH_HIP(hipMemGetInfo(&free_b, &total_b));
	uint64_t budget = (uint64_t)((double)free_b / 4.0 / 34.0);       // samples of row the groups may hold
	if (const char *e = getenv("RH_WHOLE_ROWS_MAX_SAMPLES")) budget = strtoull(e, nullptr, 10);   // (tests)
	uint32_t done = 0;
	uint64_t n_total = 0;
	std::vector<uint64_t> tmp_off;
	rh_map_stats_t tot{};
	while (done < R) {
		uint32_t m = 0;
		uint64_t mx = 0;
		while (done + m < R) {
			const uint64_t len = off[done + m + 1] - off[done + m] + 64, nmx = len > mx ? len : mx;
			if (m > 0 && nmx * (m + 1) > budget) break;
			mx = nmx; ++m;
		}
		rh_read_batch_t b = *in;
		b.n_reads = m;
		b.offsets = in->offsets + done;
		if (in->cal_offset) b.cal_offset = in->cal_offset + done;
		if (in->cal_scale) b.cal_scale = in->cal_scale + done;
		b.name_rank = in->name_rank ? in->name_rank + done : nullptr;
		b.n_filtered = in->n_filtered ? in->n_filtered + done : nullptr;
		tmp_off.assign((size_t)m + 1, 0);
		uint64_t n = 0;
		if (map_batch_once(c, mo, &b, out + n_total, out_cap - n_total, &n, tmp_off.data())) return -1;
		for (uint64_t k = 0; k < n; ++k) out[n_total + k].read_idx += done;
		for (uint32_t i = 0; i <= m; ++i) rec_offsets[done + i] = n_total + tmp_off[i];
		add_stats(tot, c->stats);
		n_total += n; done += m;
	}
	c->stats = tot;
	*n_out = n_total;
	return 0;
}

// ---------------------------------------------------------------------------------------------------- signal-target index
// ri_idx_siggen (rindex.c:927) on the device: every read is a target (name, filtered length); its whole signal goes through
// the event kernels and the sketch with the read's id (worker_sig_pipeline rindex.c:283-287: no min_events filter here), and
// the seeds of all reads, in read order, through the same sort / key / table kernels as the FASTA index.
extern "C" rh_index *rh_index_build_signals_device(rh_ctx *c, const rh_read_batch_t *in, const char *const *names, const char *pore_model_path,
                                                   const rh_idxopt_t *io, const rh_mapopt_t *mo)
{
	if (hipSetDevice(c->device) != hipSuccess) { rh_set_error("hipSetDevice failed"); return nullptr; }
	if (index_replaceable(c, "rh_index_build_signals_device")) return nullptr;
	if (!(io->flag & RH_I_SIG_TARGET)) { rh_set_error("rh_index_build_signals_device builds signal-target indexes (RH_I_SIG_TARGET, the ava presets)"); return nullptr; }
	if (io->e < 1 || io->e > 16 || io->q < 1 || io->q * io->e > 64 || io->w < 0 || io->w > RH_DEV_MAXW) { rh_set_error("unsupported index parameters e=%d q=%d w=%d", io->e, io->q, io->w); return nullptr; }
	const uint32_t R = in->n_reads;
	std::unique_ptr<rh_index_s> ix(new rh_index_s());
	ix->w = io->w; ix->e = io->e; ix->n = io->n; ix->q = io->q; ix->k = io->k; ix->flag = io->flag;
	ix->diff = io->diff; ix->fine_min = io->fine_min; ix->fine_max = io->fine_max; ix->fine_range = io->fine_range;
	if (!rh_load_model(pore_model_path, io->k, io->lev_col, ix->pore_vals)) return nullptr;
	ix->n_pore_vals = (uint32_t)ix->pore_vals.size(); ix->pore_k = (int16_t)io->k;
	rh_make_pore_inds(ix->pore_vals, io->k, ix->pore_inds);
	for (uint32_t i = 0; i < R; ++i) ix->names.push_back(names && names[i] ? names[i] : "");
	// the resident index of this context is replaced
	if (c->blob_owned) c->blob.release(); else { c->blob.p = nullptr; c->blob.cap = 0; }
	c->have_index = false; c->dix.t_rank = nullptr;
	rh_mapopt_t mw = *mo;
	mw.flag = RH_M_NO_ADAPTIVE; mw.min_events = 0;
	rh_dev_opt o;
	if (fill_dev_opt(c, &mw, &o)) return nullptr;
	c->dix.sp = rh_sketch_par{io->e, io->w, io->q, io->k, io->diff, io->fine_min, io->fine_max, io->fine_range};
	hipStream_t s = c->stream;
	void *dH = nullptr, *dY = nullptr;
	uint64_t n_seeds = 0, seed_cap = 0;
	std::vector<uint32_t> lens(R ? R : 1, 0);
	auto fail = [&]() -> rh_index* { if (dH) (void)hipFree(dH); if (dY) (void)hipFree(dY); return nullptr; };
	// room for `need` seeds in the two arrays (geometric growth, contents kept)
	auto seeds_reserve = [&](uint64_t need) -> int {
		if (need + 1 <= seed_cap) return 0;
		const uint64_t nc = need + 1 > 2 * seed_cap ? need + 1 : 2 * seed_cap;
		void *nh = nullptr, *ny = nullptr;
		if (hipMalloc(&nh, nc * 4) != hipSuccess || hipMalloc(&ny, nc * 8) != hipSuccess) { if (nh) (void)hipFree(nh); rh_set_error("index build: out of device memory for %llu seeds", (unsigned long long)nc); return -1; }
		if (n_seeds) { (void)hipMemcpyAsync(nh, dH, n_seeds * 4, hipMemcpyDeviceToDevice, s); (void)hipMemcpyAsync(ny, dY, n_seeds * 8, hipMemcpyDeviceToDevice, s); }
		if (hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(nh); (void)hipFree(ny); rh_set_error("index build: device error"); return -1; }
		if (dH) (void)hipFree(dH);
		if (dY) (void)hipFree(dY);
		dH = nh; dY = ny; seed_cap = nc;
		return 0;
	};
	if (seeds_reserve(0)) return fail();
	// the reads go through in consecutive groups whose rows (as long as the group's longest read, ~34 bytes per sample) fit a
	// quarter of the free memory; one group unless the lengths are very uneven or the read set is huge
	std::vector<uint64_t> off_h;
	const uint64_t *off = in->offsets;
	if (R && in->samples_on_device) { off_h.resize((size_t)R + 1); if (hipMemcpy(off_h.data(), in->offsets, ((size_t)R + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) { rh_set_error("index build: device error"); return fail(); } off = off_h.data(); }
	size_t free_b = 0, total_b = 0;
	(void)hipMemGetInfo(&free_b, &total_b);
	uint64_t budget = (uint64_t)((double)free_b / 4.0 / 34.0);
	if (const char *e = getenv("RH_WHOLE_ROWS_MAX_SAMPLES")) budget = strtoull(e, nullptr, 10);   // (tests)
	for (uint32_t done = 0; done < R;) {
		uint32_t m = 0;
		uint64_t mx = 0;
		while (done + m < R) {
			const uint64_t len = off[done + m + 1] - off[done + m] + 64, nmx = len > mx ? len : mx;
			if (m > 0 && nmx * (m + 1) > budget) break;
			mx = nmx; ++m;
		}
		rh_read_batch_t b = *in;
		b.n_reads = m; b.offsets = in->offsets + done; b.name_rank = nullptr;
		if (in->cal_offset) b.cal_offset = in->cal_offset + done;
		if (in->cal_scale) b.cal_scale = in->cal_scale + done;
		rh_dev_reads rd;
		if (set_row_strides(c, &mw, &b) || stage_reads(c, &b, &rd)) return fail();
		if (c->act[0].ensure((size_t)m * 4) || c->n_act_dev.ensure(64) || c->counters.ensure(16 * 8) || c->rec_off.ensure((size_t)(m + 2) * 8)) return fail();
		if (hipMemsetAsync(c->counters.p, 0, 16 * 8, s) != hipSuccess) return fail();
		rhk_prefilter(s, o, rd);
		rhk_compact_active(s, o, rd, nullptr, m, 0, c->act[0].as<uint32_t>(), c->n_act_dev.as<uint32_t>());
		uint32_t n_act = 0;
		if (hipMemcpyAsync(&n_act, c->n_act_dev.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rh_set_error("index build: device error"); return fail(); }
		rh_dev_round rr{};
		if (stage_round(c, n_act, &rr)) return fail();
		rr.act = c->act[0].as<uint32_t>(); rr.chunk = 0;
		uint64_t n_g = 0;
		if (n_act) {
			rhk_events_norm(s, o, rd, rr); rhk_events_peaks(s, o, rr); rhk_events_means(s, o, rr);
			rhk_sketch(s, o, c->dix, rd, rr);
			rhk_seed_scan(s, rr, c->rec_off.as<uint64_t>());
			uint64_t cnt[16];
			if (hipMemcpyAsync(&n_g, c->rec_off.as<uint64_t>() + n_act, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
			    hipMemcpyAsync(cnt, c->counters.p, sizeof(cnt), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rh_set_error("index build: device error"); return fail(); }
			if (cnt[7]) { rh_set_error("index build: %llu read(s) hold more event boundaries than their arrays", (unsigned long long)cnt[7]); return fail(); }
			if (seeds_reserve(n_seeds + n_g)) return fail();
			rhk_seed_pack(s, rr, c->rec_off.as<uint64_t>(), done, (uint32_t*)dH + n_seeds, (uint64_t*)dY + n_seeds);
		}
		if (hipMemcpyAsync(lens.data() + done, rd.l_sig, (size_t)m * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rh_set_error("index build: device error"); return fail(); }
		n_seeds += n_g; done += m;
	}
	uint32_t max_len = 0;
	for (uint32_t i = 0; i < R; ++i) { ix->lens.push_back(lens[i]); if (lens[i] > max_len) max_len = lens[i]; }
	release_arenas(c);                                              // the rows of whole reads are large: back to the device before the sort
	BlobHeader h{};
	void *blob = nullptr;
	uint64_t n_keys = 0;
	if (rhk_index_assemble(s, dH, dY, n_seeds, R, lens.data(), max_len, io, &h, &blob, ix->occ_hist, &n_keys)) return nullptr;
	c->blob.p = blob; c->blob.cap = h.bytes; c->blob.owned = true; c->blob_owned = true;
	if (bind_blob(c, h)) return nullptr;
	ix->dev_n_keys = n_keys; ix->dev_n_pos = h.n_pos;
	if (set_target_ranks_from(c, ix.get())) return nullptr;
	return ix.release();
}

