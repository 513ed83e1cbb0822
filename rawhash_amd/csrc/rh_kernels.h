// Device-side data structures and kernel launch wrappers of the mapping path (gfx950).
// Layout: batch-of-reads, structure-of-arrays, CSR offsets; no per-read allocation anywhere.
#pragma once
#include <cstdlib>
#include "rh_gpu.h"
#include "rh_core.h"
#include "rh_index.h"
#include "rh_synth_core.h"
#include <vector>
#include <functional>

#define RH_CHUNK_MAX   4096          // samples of one chunk held in LDS by the event kernel
#define RH_EV_CAP      2048          // events per chunk: peaks are >= 2 samples apart (revent.c:140)
#define RH_WS_PER_ANCHOR 64          // bytes of per-anchor scratch shared by DP / backtrack / compaction / regions (>= 128 B per chain: a chain has >= 2 anchors)
// Layout of a read's scratch during DP and backtrack (n anchors): {f, p} pairs [0, 8 n) | v [8 n, 12 n) | "used" marks, one byte each [16 n, 17 n) |
// claim stamps [20 n, 24 n).  Three arrays on purpose.  Measured on MI355X (human-scale step, k_backtrack_spec 234 ms): the marks inside a 16-byte
// {f, p, claim, used} record -> 404 ms (every candidate starts with a look at its own mark, most are used already, and 64 byte marks share a sector
// where 4 records do); the claim stamp inside a {f, p, claim, v} record with the marks apart -> 1051 ms (the stamps are L2 atomics: a read-modify-write
// on the line every walker of the neighbourhood is loading f / p from).
#define RH_LOGF_N      (1u << 20)    // host-libm logf() table for integer arguments (MAPQ parity, hit.c:525-533)
#define RH_DEV_MAXW    16            // largest minimiser window the device sketch supports

// index resident in HBM
struct rh_dev_index {
	const rh_tslot *table;           // (1 << lg_buckets) buckets x RH_TB_SLOTS slots
	const uint64_t *pos;             // concatenated position lists
	const uint32_t *seq_len;
	const uint32_t *t_rank;          // all-vs-all: rank of every target's name (see rd.name_rank); null otherwise
	const uint64_t *sig_off; const float *sig;   // RH_I_STORE_SIG: expected signals of the targets (DTW re-scoring); null otherwise
	int32_t lg_buckets;
	uint32_t n_seq;
	int32_t flag;
	rh_sketch_par sp;
};

// Resident index blob [table | positions | target lengths] and the 256-byte header that describes it (what RCCL
// broadcasts to the other GPUs).
struct rh_blob_header {
	uint64_t magic, bytes, table_off, pos_off, len_off, n_pos;
	int32_t lg_buckets; uint32_t n_seq; int32_t flag;
	rh_sketch_par sp;
	uint32_t max_len;
	uint32_t pad_;
	uint64_t sig_off;                // RH_I_STORE_SIG: offset of [u64 so[2 n_seq + 1] | float data] - target i's forward signal = data[so[2i] .. so[2i+1]), reverse = [so[2i+1] .. so[2i+2]); 0 = none
};
#define RH_BLOB_MAGIC 0x3130424958444952ULL   // "RIDXIB01"
// index construction on the device (rh_index_device.hip)
int rhk_index_build_device(hipStream_t s, uint32_t n_seq, const char *const *seqs, const uint32_t *lens, const std::vector<float> &model,
                           const rh_idxopt_t *io, rh_blob_header *hdr, void **blob_out, std::vector<uint32_t> &occ_hist, uint64_t *n_keys_out, int n_threads);

int rhk_index_assemble(hipStream_t s, void *seed_hash, void *seed_pos, uint64_t n_seeds, uint32_t n_seq, const uint32_t *lens, uint32_t max_len,
                       const rh_idxopt_t *io, rh_blob_header *hdr, void **blob_out, std::vector<uint32_t> &occ_hist, uint64_t *n_keys_out, bool sort_pos = false, uint64_t extra_bytes = 0);   // sort_pos: the seeds are not in position order yet; extra_bytes: room left behind the blob

// scalar parameters every kernel may need
struct rh_dev_opt {
	uint32_t chunk_size, max_num_chunk, min_events;
	uint32_t w1, w2; float thr1, thr2, peak_height;
	int32_t mid_occ;
	int32_t max_dist_t, max_dist_q, bw, max_skip, max_iter, min_cnt, min_sc, min_sc2;
	int32_t bw_long, rmq_inner_dist, rmq_size_cap;      // RH_M_RMQ chaining / bw_long > bw re-chaining (lchain.c:606, rmap.cpp:336)
	uint32_t dtw_border, dtw_fill; float dtw_band_frac, dtw_match_bonus, dtw_min_score;   // RH_M_DTW_EVALUATE_CHAINS (rmap.cpp:128-208)
	float pen_gap, pen_skip;
	float mask_level; int32_t mask_len; float pri_ratio; int32_t best_n; int32_t min_strand_sc;
	float w_bestq, w_bestmq, w_bestmc, w_threshold, w_bestma;
	int32_t min_mapq;
	float sample_per_base;
	int64_t flag;
	int32_t sig_target;
};

// per-batch read state (all arrays have n_reads entries unless noted)
struct rh_dev_reads {
	uint32_t n_reads;
	uint32_t fast5;                  // raw -> pA the way the FAST5 reader does it (float arithmetic, value truncated to int16: rsig.c:363-374)
	const int16_t *raw; const uint64_t *off; const double *cal_off; const float *cal_scale;
	const uint32_t *name_rank;       // all-vs-all: rank of the read's name among the target names (strcmp(q, t) >= 0 <=> name_rank >= t_rank[t])
	uint32_t *l_sig;                 // filtered length (sl:i tag)
	// consumed-prefix staging (rh_read_batch_t::n_filtered given, samples in page-locked host memory): only res_len[r] raw samples of read r are in
	// HBM yet (k_fetch brings more, k_need says who lacks the round's chunk); the read's filtered length comes from the caller, cnt_res[r] counts the
	// survivors of the resident stretch.  All null / 0 when the whole batch was uploaded or handed over on the device.
	uint32_t *res_len, *cnt_res; const uint32_t *l_sig_given; int16_t *raw_w;
	uint32_t *chunk_start; uint32_t cs_stride;   // n_reads x cs_stride (= max_num_chunk + 1): raw index of the first sample of chunk c
	double *sum, *sum2; uint32_t *n_sum;   // running normalisation sums (rmap.cpp:412-413)
	uint32_t *ev_off;                // events accepted so far (reg->offset)
	uint32_t *n_prev; uint64_t *prev_off;  // carried chain anchors (reg->prev_anchors)
	uint8_t *done;                   // 1 once a mapping decision stopped the read
	uint32_t *stop_chunk;            // chunk index at which it stopped
	// summary of the regions of the last processed chunk (what the record is built from)
	int32_t *ls_ncregs, *ls_cnt, *ls_score, *ls_mapq, *ls_qs, *ls_qe, *ls_rs, *ls_re, *ls_rid, *ls_rev;
	float *events; uint32_t ev_stride;   // RH_M_DTW_EVALUATE_CHAINS: the events of every processed chunk of every read (reg->events, rmap.cpp:237-241), ev_stride floats per read
};

// Records of a sort job: 16-byte rh_mm128_t (key = x, payload = y), or - when key and payload fit one word - 8-byte words
//   key' << shift | payload,   key' = hi << (lo + mid) | mid << lo | lo   (the bit fields of the ORIGINAL 64-bit key  hi << 63 | mid << 32 | lo),
// which halves every byte the sorters move.  The order is still the reference's radix_sort_128x permutation of the original keys: the
// digits of a level are taken from the key rebuilt at its original bit positions (rh_rec8_key).
// up != 0 (the multi-workgroup levels of any-order jobs only, rh_sort_job::any_up): the sorter orders by the packed key with its upper fields
// (strand, target) moved up to bit 63 and the position left at the bottom, (k >> lo) << up | (k & low mask), k = w >> shift - the same order as the
// original key's (the fields keep their significance), but the first level's byte then holds strand AND target: a human-scale chunk's anchors split 48
// ways in ONE placement instead of 2 ways (byte 7) and then 24 (byte 4), into the same ~1.9 k-record buckets the block sorter likes (measured: moving two
// position bits up as well - 192 buckets of ~460 - costs more in the block sorter than the placement saves).  Exact jobs never set it: their levels have
// to be the reference's bytes.
struct rh_rec_fmt { uint8_t rec8, shift, lo, mid, up; };
RH_HD inline uint64_t rh_rec8_key(uint64_t w, uint32_t shift, uint32_t lo, uint32_t mid)
{
	const uint64_t k = w >> shift;
	return (k & ((1ull << lo) - 1ull)) | ((k >> lo) & ((1ull << mid) - 1ull)) << 32 | ((k >> (lo + mid)) & 1ull) << 63;
}
RH_HD inline uint64_t rh_rec8_pack_key(uint64_t x, uint32_t lo, uint32_t mid)   // key' of the original key x (whose fields fit lo / mid bits)
{
	return (x & ((1ull << lo) - 1ull)) | ((x >> 32) & ((1ull << mid) - 1ull)) << lo | (x >> 63) << (lo + mid);
}

// per-round work arrays indexed by active slot a in [0, n_act)
struct rh_dev_round {
	uint32_t max_anchors;                    // anchors of the largest active read this round (0 = unknown)
	uint8_t akey_on, akey_lo, akey_mid;      // anchor x = rev << 63 | rid << 32 | pos with pos < 2^akey_lo, rid < 2^akey_mid (0 = unknown)
	const uint32_t *act;             // read ids active this round
	uint32_t n_act; uint32_t chunk;
	uint32_t ev_row, ev_cap;         // strides of the per-read rows: samples (zbuf / t1buf / t2buf) and events (peaks, ev, seeds, matches)
	uint32_t whole;                  // RH_M_NO_ADAPTIVE: a round covers whole reads (rows in HBM only, 32-bit peak positions)
	float *zbuf, *t1buf, *t2buf; uint32_t *n_norm;   // n_act rows of (RH_CHUNK_MAX + 64): normalised signal, both t-statistics
	uint16_t *peaks; uint32_t *n_peaks;              // n_act x RH_EV_CAP peak positions
	float *ev; uint32_t *n_ev;       // n_act x RH_EV_CAP
	uint8_t *skip;                   // chunk produced < min_events events (rmap.cpp:232)
	uint64_t *sx, *sy; uint32_t *n_seed;     // n_act x RH_EV_CAP seeds
	uint64_t *m_val; uint32_t *m_n, *m_meta, *m_pref; uint32_t *n_match, *n_new; int32_t *rep_len;   // kept seed matches
	uint64_t *a_off;                 // n_act+1 anchor offsets (exclusive scan of n_new + n_prev)
	rh_mm128_t *raw;                 // anchors as expanded (unsorted)
	rh_mm128_t *anc;                 // anchors in the reference's sorted order
	uint8_t *need_exact;             // sort: read has equal keys, needs the exact-permutation pass (also reused as a per-read redo flag)
	uint8_t *need_exact2;
	rh_mm128_t *chn;                 // chained anchors, chains ordered by target position
	const rh_mm128_t *prev_in; rh_mm128_t *prev_out;
	uint64_t *u; uint32_t *n_u, *n_v;        // chains: score<<32 | count
	rh_mm128_t *zs; uint32_t *n_z;           // backtrack candidates (score, anchor) in radix_sort_128x order
	unsigned char *ws; uint32_t ws_stride;   // ws_stride bytes of scratch per anchor: RH_WS_PER_ANCHOR, twice that when chains may have a single anchor (min_num_anchors < 2: the region stage needs 128 B per chain)
	unsigned char *sort_ws; size_t sort_ws_bytes; void *sort_pin; uint64_t sort_total;   // tables of the multi-workgroup segment sorter (rh_bigsort.hip); its second record array is an arena idle at the time
	float *dtw_ws; uint32_t dtw_stride;      // RH_M_DTW_EVALUATE_CHAINS: DP buffers, dtw_stride floats per active read
	uint32_t *dtw_n; float *dtw_rec; const uint64_t *dtw_off; const int32_t *dtw_dec;   // regions per read; packed per-region values for the host's MAPQ; their offsets; the host's decisions (3 int32 per read)
	uint64_t *counters;              // [0] events [1] seeds [2] hits [3] anchors [4] chained [5] samples used [6] chunks [7] chunks with more peaks than RH_EV_CAP (an error)
	// 8-byte records (rh_rec_fmt; all zero = 16-byte records everywhere: the stage-level calls):
	//   afmt  the ANCHORS OF THE ROUND are one word each,  key' << shift | tandem << aq_bits | q_pos  (shift = aq_bits + 1; a_span = the index's constant
	//         span, seg_id = 0: nothing is lost): k_expand writes them so, the sort moves them so, and raw / anc / prev_in / prev_out / the carry
	//         buffers hold them so - the kernels that need x and y take them apart in registers (rh_an_ld);
	//   cfmt  chain-order keys  key' << shift | chain;   z8  candidates  score << 32 | anchor
	rh_rec_fmt afmt, cfmt; uint8_t aq_bits, z8, a_span;
	uint64_t arena_n;                // anchors the 16-byte-per-anchor arenas (raw, anc, zs, prev_out) hold
	uint8_t lazy_reorder;            // the mapping path's default mode: compact_a's last copy (chains back over the anchor slice in sorted order) is left out - all the region stage reads of it are the first and last anchor of every chain, which it takes from the gathered chains in the carry staging (k_chain_reorder leaves where each sorted chain starts there)
	int32_t *reg_out;                // stage-level call only (rh_regions_batch): 18 int32 per kept region, region k of active read a at (a_off[a] + k) * 18; null on the mapping path
};

// anchor i of an anchor array of the round (rr.anc, rr.prev_in, rr.prev_out: absolute element index), whichever way the round keeps them
RH_HD inline rh_mm128_t rh_anchor_unpack(uint64_t w, const rh_rec_fmt &f, uint32_t qb, uint32_t span)
{
	rh_mm128_t p;
	p.x = rh_rec8_key(w, f.shift, f.lo, f.mid);
	p.y = (uint64_t)span << 32 | (w & ((1ull << qb) - 1ull)) | ((w >> qb) & 1ull) << 38;
	return p;
}
RH_HD inline uint64_t rh_anchor_pack(const rh_mm128_t &p, const rh_rec_fmt &f, uint32_t qb)
{
	return rh_rec8_pack_key(p.x, f.lo, f.mid) << f.shift | ((p.y >> 38) & 1ull) << qb | (uint64_t)(uint32_t)p.y;
}
RH_HD inline rh_mm128_t rh_an_ld(const rh_dev_round &rr, const rh_mm128_t *arr, uint64_t i)
{
	if (rr.afmt.rec8) return rh_anchor_unpack(reinterpret_cast<const uint64_t*>(arr)[i], rr.afmt, rr.aq_bits, rr.a_span);
	return arr[i];
}
RH_HD inline void rh_an_cp(const rh_dev_round &rr, rh_mm128_t *dst, uint64_t di, const rh_mm128_t *src, uint64_t si)   // dst[di] = src[si]
{
	if (rr.afmt.rec8) reinterpret_cast<uint64_t*>(dst)[di] = reinterpret_cast<const uint64_t*>(src)[si];
	else dst[di] = src[si];
}

// an anchor as the round keeps it, moved through a register without taking it apart (one-word anchors: the word in .x)
RH_HD inline rh_mm128_t rh_an_raw_ld(const rh_dev_round &rr, const rh_mm128_t *arr, uint64_t i)
{
	if (rr.afmt.rec8) { rh_mm128_t r; r.x = reinterpret_cast<const uint64_t*>(arr)[i]; r.y = 0; return r; }
	return arr[i];
}
RH_HD inline void rh_an_raw_st(const rh_dev_round &rr, rh_mm128_t *arr, uint64_t i, const rh_mm128_t &r)
{
	if (rr.afmt.rec8) reinterpret_cast<uint64_t*>(arr)[i] = r.x;
	else arr[i] = r;
}
RH_HD inline uint64_t rh_an_raw_x(const rh_dev_round &rr, const rh_mm128_t &r) { return rr.afmt.rec8 ? rh_rec8_key(r.x, rr.afmt.shift, rr.afmt.lo, rr.afmt.mid) : r.x; }

// ---- LDS size classes of the block sorter (rh_sort.hip)
// Size classes = LDS footprints (12 B per record + ~4.5 KB) chosen for whole workgroups per CU; the allocation granularity
// means a class must stay clearly below 160 KB / k to get k workgroups resident (measured: 54.0 KB gives 2, 52.4 KB gives 3).
#ifndef RH_SORT_CAP0
#define RH_SORT_CAP0 512      // ~10 KB: candidate / chain-key sorts and short anchor lists
#endif
#ifndef RH_SORT_CAP1
#define RH_SORT_CAP1 2816     // ~38 KB: four workgroups per CU (a typical chunk's anchors)
#endif
#ifndef RH_SORT_CAP2
#define RH_SORT_CAP2 3968     // ~52 KB: three
#endif
#ifndef RH_SORT_CAP3
#define RH_SORT_CAP3 6144     // ~78 KB: two (unmapped reads accumulate carried anchors)
#endif
#ifndef RH_SORT_CAP4
#define RH_SORT_CAP4 8192     // ~103 KB: one
#endif

// the same classes when the job's keys fit 32-bit words: 8 B of LDS per record
#ifndef RH_SORT32_CAPH
#define RH_SORT32_CAPH 2048     // ~21 KB: six workgroups per CU (the strand x target buckets of a human-scale chunk: ~1.9 k anchors)
#endif
#ifndef RH_SORT32_CAP1
#define RH_SORT32_CAP1 4096     // ~38 KB: four workgroups per CU
#endif
#ifndef RH_SORT32_CAP2
#define RH_SORT32_CAP2 5632     // ~51 KB: three
#endif
#ifndef RH_SORT32_CAP3
#define RH_SORT32_CAP3 8192     // ~73 KB: two
#endif

#define RH_SORT_LDS_MIN_TOP (RH_SORT32_CAP3 < RH_SORT_CAP4 ? RH_SORT32_CAP3 : RH_SORT_CAP4)   // segments longer than this may need rh_bigsort.hip

// a batch of independent record segments to be put into radix_sort_128x order (rh_sort.hip)
struct rh_sort_job {
	uint32_t n_seg; const uint8_t *skip; const uint64_t *off; const uint32_t *cnt;   // segment a = [off[a], off[a] + (cnt ? cnt[a] : off[a+1]-off[a]))
	const rh_mm128_t *src; rh_mm128_t *dst; uint8_t *need_exact;
	unsigned char *scratch; uint32_t scratch_stride, scratch_skip;   // 2 KB per oversized segment at scratch + off*stride + skip*len
	// keys known to be  hi << 63 | mid << 32 | lo  with lo < 2^kc_lo, mid < 2^kc_mid (kc_lo + kc_mid + kc_hi <= 32, kc_on set):
	// the LDS sorter keeps them as 32-bit words (8 instead of 12 bytes of LDS per record -> more workgroups per CU)
	uint8_t kc_on, kc_lo, kc_mid, kc_hi;
	uint32_t n_max;                          // no segment is longer than this (0 = unknown): size classes above it are not launched
	// segments beyond the LDS classes (rh_bigsort.hip): a second record array the size of src (src itself is overwritten),
	// scratch of rhk_bigsort_ws_bytes(big_total, ...) bytes, 32 pinned host bytes for the per-level read-backs
	rh_mm128_t *big_alt; unsigned char *big_ws; size_t big_ws_bytes; void *big_pin; uint64_t big_total;
	// keys that are almost never equal (hashed): a sorted order without ties is unique, so the segments beyond the LDS classes are
	// placed level by level in ANY order (no token walk); afterwards every segment is checked for equal neighbours:
	// redo_skip[a] (set to 1 by the caller) = 0 for the segments that hold equal keys (the caller redoes them with any_order = 0 and skip = redo_skip),
	// 1 for all others; *n_redo (host) = their number
	uint8_t any_order; uint8_t *redo_skip; uint32_t *n_redo;
	uint8_t any_up;                          // 8-byte records of an any-order job: rh_rec_fmt::up of its multi-workgroup levels (0: the levels are the original key's bytes)
	// cnt[] may be rewritten (the bucket lists of rh_bigsort.hip): the wavefront-per-segment sorter zeroes the count of a segment it
	// has finished, so that the LDS classes launched after it pass over it
	uint32_t *cnt_rw;
	// which sort of the path this is (1 anchors, 2 chain candidates, 3 chains, 4 regions, 0 anything else): the multi-workgroup sorter
	// remembers per kind the byte its first level split on (rh_bigsort.hip: k_bs_hist0)
	uint8_t kind;
	// the exact re-run of the segments an any-order job found equal keys in (skip = its redo_skip): that job also left, in the sorter's scratch, WHERE in the
	// sorted order the equal keys lie, and the re-run takes its exact passes only along the way to them - a range of a level (an interval of the sorted order)
	// that holds no equal keys has one sorted order and is placed in any order again.  Only right after that job, on the same scratch.
	uint8_t tie_path;
	// dead_cnt[a] records of segment a - its lowest keys, all equal - are never looked at in the sorted order (the backtrack candidates without a
	// predecessor: k_zbuild): they take part in the permutation like any other record, but the multi-workgroup sorter does not move them once their
	// bucket is final (their stretch of dst stays unwritten).  Null: none.
	const uint32_t *dead_cnt;
	// 8-byte records (see rh_rec_fmt): src / dst / big_alt then point to uint64_t arrays (same record offsets)
	rh_rec_fmt rf;
	// the caller redoes every segment reported in need_exact from its input (the buckets of an any-order job): the LDS sorter reports equal keys and leaves it at that
	uint8_t no_redo;
	// the LDS sorter's path for segments without equal keys (rh_sort.hip: sort_fast); rhk_sort_job sets it (development knob RH_SORT_FAST=0: off)
	uint8_t fast_on;
};
// record access of the sorters, by record type
template <class REC> struct rh_rec_ops;
template <> struct rh_rec_ops<rh_mm128_t> { static RH_HD inline uint64_t key(const rh_mm128_t &r, const rh_rec_fmt &) { return r.x; } };
template <> struct rh_rec_ops<uint64_t> { static RH_HD inline uint64_t key(const uint64_t &r, const rh_rec_fmt &f) { if (!f.up) return rh_rec8_key(r, f.shift, f.lo, f.mid); const uint64_t k = r >> f.shift; return (k >> f.lo) << f.up | (k & ((1ull << f.lo) - 1ull)); } };
// Unused dynamic LDS handed to the one-wavefront-per-read kernels (development knob RH_WAVE_LDS, bytes): their wavefronts live long, and
// without a cap per CU they end up holding every wave slot while the other streams' bandwidth-bound kernels wait
inline uint32_t rh_wave_lds() { static const uint32_t v = RH_DEVENV("RH_WAVE_LDS") ? (uint32_t)strtoul(RH_DEVENV("RH_WAVE_LDS"), nullptr, 10) : 0u; return v; }
int rhk_sort_job(hipStream_t s, const rh_sort_job &jb, bool all_exact, uint32_t min_n);   // segments with <= min_n records are left alone
uint32_t rhk_sort_lds_max(const rh_sort_job &jb);                        // longest segment the LDS classes take for this job's keys
size_t rhk_bigsort_ws_bytes(uint64_t total, uint32_t n_lo);
int rhk_bigsort(hipStream_t s, const rh_sort_job &jb, bool all_exact, uint32_t n_lo);

// kernel launchers (rh_kernels.hip)
void rhk_prefilter(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const uint32_t *act = nullptr, uint32_t n = 0, int init = 1, unsigned long long *bad = nullptr);
void rhk_need(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const uint32_t *act, uint32_t n, uint32_t chunk, uint32_t grow, uint32_t *new_len, uint32_t *n_need);
void rhk_fetch(hipStream_t s, const rh_dev_reads &rd, const int16_t *host_samples, const uint32_t *act, uint32_t n, const uint32_t *new_len, uint32_t max_span);
void rhk_events_norm(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r);
void rhk_events_peaks(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r);
void rhk_events_means(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r);
inline void rhk_events(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r) { rhk_events_norm(s, o, rd, r); rhk_events_peaks(s, o, r); rhk_events_means(s, o, r); }
void rhk_sketch(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r);
void rhk_probe(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r);
void rhk_scan_anchors(hipStream_t s, const rh_dev_reads &rd, const rh_dev_round &r);
void rhk_expand(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r, const uint8_t *skip2 = nullptr);   // skip2: only the reads with skip2[a] == 0 (and not r.skip[a]), again
int rhk_sort(hipStream_t s, const rh_dev_index &ix, const rh_dev_round &r, const std::function<int(const uint8_t*)> &reexpand = {});   // (r.afmt: one-word anchors, raw -> anc as such)
void rhk_chain(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r);
void rhk_chain_rmq(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r, const uint32_t *counts, int32_t max_dist, int32_t max_dist_inner, int32_t cap_rmq_size);   // mg_lchain_rmq (lchain.c:606); o.bw = the bandwidth of this pass
int rhk_zsort(hipStream_t s, const rh_dev_opt &o, const rh_dev_round &r);
int rhk_backtrack(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r);
int rhk_regions_sort(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r);
void rhk_regions(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r, const float *logf_tab);
// RH_M_DTW_EVALUATE_CHAINS: regions + DTW scores (device), MAPQ and decision (host: logf of a float), commit (device)
void rhk_events_append(hipStream_t s, const rh_dev_reads &rd, const rh_dev_round &r);
void rhk_regions_dtw(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_dev_round &r);
bool rhk_regions_fast_ok(const rh_dev_opt &o);   // the primaries-only region kernels apply (default selection: secondaries dropped, not all-chains)
void rhk_dtw_pack(hipStream_t s, const rh_dev_round &r);
void rhk_dtw_decide(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r, const float *logf_tab, uint32_t *n_host);   // MAPQ + decision on the device where the host's logf cannot matter; *n_host (device) += reads left to the host
void rhk_dtw_commit(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const rh_dev_round &r);
void rhk_compact_active(hipStream_t s, const rh_dev_opt &o, const rh_dev_reads &rd, const uint32_t *act_in, uint32_t n_in, uint32_t next_chunk,
                        uint32_t *act_out, uint32_t *n_out);
void rhk_rebase_offsets(hipStream_t s, const uint64_t *a_off, uint32_t n, uint64_t *out);
void rhk_carry_scan(hipStream_t s, const rh_dev_reads &rd, const uint32_t *act, uint32_t n, uint64_t used, uint64_t *dst_off, uint64_t *total_out);
void rhk_carry_copy(hipStream_t s, const rh_dev_reads &rd, const uint32_t *act, uint32_t n, const rh_mm128_t *staging, const uint64_t *dst_off, rh_mm128_t *carry, int words8);   // words8: the anchors are one word each
void rhk_finalize(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, rh_map_record_t *rec);
void rhk_seed_scan(hipStream_t s, const rh_dev_round &r, uint64_t *off);
void rhk_seed_pack(hipStream_t s, const rh_dev_round &r, const uint64_t *off, uint32_t id0, uint32_t *hash_out, uint64_t *pos_out);   // target ids = id0 + read index
void rhk_ava_rec_scan(hipStream_t s, const rh_dev_reads &rd, uint64_t *rec_off);
void rhk_finalize_ava(hipStream_t s, const rh_dev_opt &o, const rh_dev_index &ix, const rh_dev_reads &rd, const rh_mm128_t *maps, const uint64_t *rec_off, rh_map_record_t *rec);
void rhk_synth_reads(hipStream_t s, const rh_synth_cfg_t &c, const int32_t *level16, uint32_t k, uint64_t first, uint32_t n, int16_t *samples, uint64_t *off, double *cal_off, float *cal_scale);
