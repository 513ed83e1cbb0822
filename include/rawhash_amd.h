/* rawhash_amd -- MI355X-native raw-signal mapping path: C ABI.
 *
 * Drop-in boundary for ONE path of CMU-SAFARI/RawHash (RawHash2 v2.1): the per-read mapping pipeline that
 * `kt_for(p->n_threads, map_worker_for, step_mt*, n_sig)` runs (reference src/rmap.cpp:700), i.e.
 *   raw signal -> pA+filter -> normalise -> event segmentation -> quantise+hash sketch -> seed lookup
 *   -> anchor sort -> chaining DP -> regions/MAPQ -> mapping decision -> PAF record.
 * The reference has no FFI of its own (single C/C++ program); every entry point below names the reference
 * function(s) it replaces as file:line relative to the reference's src/.  INTEGRATION.md shows the binding a
 * RawHash2 maintainer would add in rmap.cpp.
 *
 * Conventions: plain C, opaque handles, `int` status (0 ok, -1 error; rh_last_error() gives the text),
 * caller-owned buffers, no torch/HIP types in any signature.  One rh_ctx per GPU; calls on one context must be
 * serialised by the caller (contexts on different GPUs are independent).  The library has NO CPU fallback:
 * every compute entry point fails with -1 when no gfx950 device is present.
 */
#ifndef RAWHASH_AMD_H
#define RAWHASH_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RH_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------ types */

typedef struct { uint64_t x, y; } rh_mm128_t;      /* = mm128_t, rutils.h:20 */

/* index flags, roptions.h:8-16 */
#define RH_I_MIN            0x2
#define RH_I_STORE_SIG      0x10
#define RH_I_SIG_TARGET     0x20
#define RH_I_NO_REV_TARGET  0x40
/* map flags, roptions.h:18-34 */
#define RH_M_RMQ            0x2
#define RH_M_HARD_MLEVEL    0x4
#define RH_M_NO_ADAPTIVE    0x20
#define RH_M_DTW_EVALUATE_CHAINS 0x40
#define RH_M_ALL_CHAINS     0x2000

/* Indexing options: the fields of ri_idxopt_t (roptions.h:50-67) that reach the path. */
typedef struct rh_idxopt_s {
	int32_t b, w, e, n, q, k, flag, lev_col;
	float diff, fine_min, fine_max, fine_range;
} rh_idxopt_t;

/* Mapping options: the fields of ri_mapopt_t (roptions.h:69-143) that reach the path. */
typedef struct rh_mapopt_s {
	uint32_t bp_per_sec, sample_rate, chunk_size;
	float sample_per_base;
	float mid_occ_frac;
	int32_t min_mid_occ, max_mid_occ;
	int32_t mid_occ, max_max_occ, occ_dist;
	uint32_t min_events;
	int32_t bw, bw_long, max_target_gap_length, max_query_gap_length, max_chain_iter;
	int32_t max_num_skips, min_num_anchors, min_chaining_score, min_chaining_score2;
	float chain_gap_scale, chain_skip_scale;
	float w_bestq, w_bestmq, w_bestmc, w_threshold;
	float mask_level; int32_t mask_len;
	float pri_ratio; int32_t best_n;
	float alt_drop;
	uint32_t max_num_chunk;
	int32_t min_mapq;
	int64_t flag;
	uint32_t window_length1, window_length2;
	float threshold1, threshold2, peak_height;
	int32_t rmq_inner_dist, rmq_size_cap;   /* --rmq-inner-dist [1000], --rmq-size-cap [100000] (roptions.c:65-66): RH_M_RMQ chaining and the bw_long > bw re-chaining */
	/* RH_M_DTW_EVALUATE_CHAINS (--dtw-evaluate-chains; the index must carry the target signals: RH_I_STORE_SIG), roptions.c:84, 96-101 */
	uint32_t dtw_border_constraint;         /* RH_DTW_BORDER_GLOBAL 0 | RH_DTW_BORDER_SPARSE 1 (between consecutive anchors) [sparse] */
	uint32_t dtw_fill_method;               /* RH_DTW_FILL_FULL 0 | RH_DTW_FILL_BANDED 1 [banded] */
	float dtw_band_radius_frac, dtw_match_bonus, dtw_min_score;   /* [0.10, 0.4, 20.0] */
	float w_bestma;                         /* --w-bestma [0.2]: weight of the alignment score in the mapping decision (rmap.cpp:474) */
} rh_mapopt_t;
#define RH_DTW_BORDER_GLOBAL 0u
#define RH_DTW_BORDER_SPARSE 1u
#define RH_DTW_FILL_FULL     0u
#define RH_DTW_FILL_BANDED   1u

/* One output record = ri_map_t (rmap.h:12-22) + the integer tag values rmap.cpp:523-571 formats.
 * Unmapped reads get exactly one record with mapped=0 (rmap.cpp:521-556). */
typedef struct rh_map_record_s {
	uint32_t read_idx;                /* index in the submitted batch */
	uint32_t read_length;             /* ri_map_t::read_length */
	uint32_t ref_id;
	uint32_t read_start_position, read_end_position;
	uint32_t fragment_start_position, fragment_length;
	uint8_t  mapq, rev, mapped, _pad;
	int32_t  tag_ci, tag_sl, tag_cm, tag_nc, tag_s1;   /* ci:i sl:i cm:i nc:i s1:i */
} rh_map_record_t;

/* A batch of reads in structure-of-arrays / CSR form (replaces step_mt::sig[], rmap.h:60-67 + ri_sig_t
 * rsig.h:20-28).  Samples are the RAW int16 ADC values; the pA conversion + 30<pA<200 filter of
 * rsig.c:496-503 happens on the device. */
typedef struct rh_read_batch_s {
	uint32_t n_reads;
	const int16_t  *samples;          /* concatenated raw samples */
	const uint64_t *offsets;          /* n_reads+1 sample offsets into samples[] */
	const double   *cal_offset;       /* per read: slow5 `offset`                (may be NULL -> 0) */
	const float    *cal_scale;        /* per read: (float)(range/digitisation)   (may be NULL -> 1) */
	const uint32_t *name_rank;        /* only for RH_M_ALL_CHAINS (host array): rank of the read's name among the target
	                                     names, such that strcmp(qname, tname) >= 0  <=>  name_rank[q] >= rank of the
	                                     target (rh_index_name_ranks computes both sides); NULL otherwise */
	int samples_on_device;            /* 1: samples/offsets/cal_* are device pointers already resident in HBM */
	int fast5_ingest;                 /* 1: raw -> pA as the FAST5 reader does it (rsig.c:363-374): offset and scale are floats there, so
	                                     (raw + (float)cal_offset) * cal_scale is float arithmetic, and a sample that passes the
	                                     30 < pA < 200 test is truncated to int16 before it becomes the float signal.  0: the
	                                     SLOW5 / POD5 readers (rsig.c:452, :497: double offset, no truncation) */
	const uint32_t *n_filtered;       /* optional (host array, NULL = not known): per read, how many samples pass the reader's 30 < pA < 200 filter -
	                                     what ri_read_sig leaves as l_sig (rsig.c:496-503) and the sl:i tag prints.  A reader that counts while it
	                                     decodes (rh_reads_batch does; rh_count_filtered for other sources) lets the device FETCH ONLY THE SIGNAL THE
	                                     ROUNDS CONSUME when samples[] is page-locked (rh_pinned_alloc, rh_reads_*): a read that maps stops after
	                                     one or two chunks (rmap.cpp:425/498), the rest of its signal never crosses PCIe.  Values that are not the
	                                     true counts fail the call loudly where the device sees the whole read - and ONLY there: for a read that maps after one or two
	                                     chunks the count is UNCHECKED INPUT (it still bounds the chunk loop and is what sl:i prints), and with samples[]
	                                     in pageable memory every count is checked.  Like every per-read array it has to
	                                     be offset together with `offsets` when a caller slices a batch by hand. */
} rh_read_batch_t;

typedef struct rh_index_s rh_index;  /* host-side parsed .ind (flattened) */
typedef struct rh_ctx_s   rh_ctx;    /* one GPU: streams, arenas, resident index */

/* ------------------------------------------------------------------------------------------- utilities */
RH_API const char *rh_last_error(void);
RH_API const char *rh_version(void);
RH_API int rh_device_count(void);                       /* number of visible HIP devices (0 if none) */

/* ------------------------------------------------------------------------------------------- options */
RH_API void rh_idxopt_init(rh_idxopt_t *io);             /* ri_idxopt_init roptions.c:4  */
RH_API void rh_mapopt_init(rh_mapopt_t *mo);             /* ri_mapopt_init roptions.c:34 */
RH_API int  rh_set_preset(const char *preset, rh_idxopt_t *io, rh_mapopt_t *mo); /* ri_set_opt main.cpp:111; preset NULL = defaults */

/* ------------------------------------------------------------------------------------------- index (.ind) */
RH_API rh_index *rh_index_load(const char *ind_path);    /* ri_idx_load rindex.c:650 (format: rindex.c:545) */
/* ri_idx_gen rindex.c:900 + ri_idx_dump rindex.c:545: build from FASTA (plain or .gz is NOT supported: plain only)
 * + k-mer model (load_pore rutils.c:133) and optionally write `out_ind` (may be NULL). */
RH_API rh_index *rh_index_build(const char *fasta_path, const char *pore_model_path, const rh_idxopt_t *io,
                                const char *out_ind, int n_threads);
RH_API void      rh_index_destroy(rh_index *idx);
RH_API void      rh_mapopt_update(rh_mapopt_t *mo, const rh_index *idx); /* ri_mapopt_update rindex.c:1041 (+ ri_idx_cal_max_occ :1018) */
RH_API uint32_t  rh_index_n_seq(const rh_index *idx);
RH_API const char *rh_index_seq_name(const rh_index *idx, uint32_t i);
RH_API uint32_t  rh_index_seq_len(const rh_index *idx, uint32_t i);
RH_API void      rh_index_params(const rh_index *idx, rh_idxopt_t *out);  /* w,e,n,q,k,flag,diff,fine_* stored in the header */
RH_API uint64_t  rh_index_n_keys(const rh_index *idx);
RH_API uint64_t  rh_index_n_positions(const rh_index *idx);
/* ri_idx_get rindex.c:497 on the host copy (used by tests and by the CLI's sanity checks) */
RH_API const uint64_t *rh_index_get(const rh_index *idx, uint64_t hashval, int *n);

/* ------------------------------------------------------------------------------------------- device context */
RH_API int  rh_ctx_create(rh_ctx **out, int device_id);
RH_API void rh_ctx_destroy(rh_ctx *ctx);
/* Flatten + upload the index into this GPU's HBM (bucketed open-addressing table + positions array). */
RH_API int  rh_index_upload(rh_ctx *ctx, const rh_index *idx);
/* Multi-GPU replication over RCCL is done by the caller on the raw device blob (torch.distributed broadcast
 * in bench.py / rawhash_amd.dist): rank 0 uploads, every rank allocates `bytes`, broadcasts, then adopts. */
RH_API int  rh_index_device_blob(rh_ctx *ctx, void **dev_ptr, uint64_t *bytes, void *header_out /* >= 256 B */);
RH_API int  rh_index_copy_blob(rh_ctx *ctx, void *dst_dev_ptr);   /* device-to-device copy of the resident blob */
RH_API int  rh_index_adopt_blob(rh_ctx *ctx, const rh_index *idx_meta /* may be NULL */, void *dev_ptr, uint64_t bytes,
                                const void *header /* from rank 0 */, int take_ownership);
/* Single-process replication (a multi-threaded C/C++ host driving all GPUs of a node, one rh_ctx each): the resident index of
 * ctxs[0] is copied device-to-device (hipMemcpyPeer: xGMI where the GPUs are linked) into every other context, which adopts
 * it.  Contexts on distinct devices: ONE collective, ncclBroadcast over RCCL (librccl.so.1, loaded at run time) in pieces of <= 1 GiB
 * - the north star's "RCCL over xGMI only to broadcast the index at load", reachable from a single C/C++ process; RH_BCAST=peer, contexts
 * that share a device, or a missing librccl fall back to a doubling tree of hipMemcpyPeerAsync copies.  One process per GPU replicates
 * with RCCL through torch.distributed instead (rawhash_amd.dist / bench.py). */
RH_API int  rh_index_bcast(rh_ctx *const *ctxs, int n);
RH_API int  rh_index_bcast_path(void);            /* what the last rh_index_bcast went through: 1 RCCL, 2 peer copies, 0 none yet */
RH_API int  rh_rccl_selftest(rh_ctx *ctx);         /* librccl loads, resolves, and a one-rank communicator broadcasts in place (one-GPU boxes) */

/* ri_idx_gen rindex.c:900 on the GPU: sketches the targets, sorts and groups the seeds and fills the
 * HBM-resident table of this context directly (as rh_index_upload would), in seconds for a human-sized reference.
 * seqs[i] = the bases of target i in host memory (ACGTU in either case; anything else is an ambiguous base,
 * ri_seq_to_sig rsig.c:13-41), lens[i] < 2^31.  Plain (w = 0) and minimiser (w > 0, ri_sketch_min rsketch.c:55-141) indexes are both
 * built on the device; signal-target indexes by rh_index_build_signals_device below.  The returned host object carries the header, target names/lengths and the occupancy
 * statistics rh_mapopt_update needs; rh_index_download fetches keys and positions (for rh_index_get or to write a .ind
 * with rh_index_write). */
RH_API rh_index *rh_index_build_device(rh_ctx *ctx, uint32_t n_seq, const char *const *names, const char *const *seqs, const uint32_t *lens,
                                       const char *pore_model_path, const rh_idxopt_t *io, int n_threads);
/* ri_idx_siggen rindex.c:927 (Rawsamble: `rawhash2 -x ava -p model -d out.ind reads`): every read of the batch becomes a
 * target (name, filtered signal length); event detection over the whole signal + sketch + bucketing run on the device.  `mo`
 * supplies the segmentation parameters (window lengths, thresholds, peak height).  The index is resident on `ctx` afterwards
 * (target name ranks included); rh_index_download + rh_index_write give the reference's .ind file. */
RH_API rh_index *rh_index_build_signals_device(rh_ctx *ctx, const rh_read_batch_t *reads, const char *const *names, const char *pore_model_path,
                                               const rh_idxopt_t *io, const rh_mapopt_t *mo);
/* All-vs-all drops the hits on targets whose name is not greater than the read's (strcmp(qname, tname) >= 0, rmap.cpp:86).
 * Names stay on the host: rh_index_name_ranks turns the comparison into integers (query_ranks[n] for `names`, target_ranks
 * [n_seq] for the index's targets; either may be NULL), rh_index_set_target_ranks hands the target side to the device
 * (rh_index_upload and rh_index_build_signals_device do it themselves; a broadcast / adopted blob needs the call). */
RH_API int  rh_index_name_ranks(const rh_index *idx, const char *const *names, uint32_t n, uint32_t *query_ranks, uint32_t *target_ranks);
RH_API int  rh_index_set_target_ranks(rh_ctx *ctx, const uint32_t *target_ranks, uint32_t n);
RH_API rh_index *rh_index_build_device_fasta(rh_ctx *ctx, const char *fasta_path, const char *pore_model_path, const rh_idxopt_t *io, int n_threads);
RH_API int  rh_index_download(rh_ctx *ctx, rh_index *idx, int n_threads);
RH_API int  rh_index_write(const rh_index *idx, const char *out_ind);          /* ri_idx_dump rindex.c:545 */

/* ------------------------------------------------------------------------------------------- the hot path */
/* = kt_for(n_threads, map_worker_for, step, n_sig)  rmap.cpp:700 / map_worker_for rmap.cpp:389.
 * out must hold at least rh_map_max_records(batch, mo) records; *n_out receives the count.  Records are ordered by
 * read_idx (then chain order), the order step 2 prints them in (rmap.cpp:740). */
RH_API uint64_t rh_map_max_records(const rh_read_batch_t *in, const rh_mapopt_t *mo);
RH_API int  rh_map_batch(rh_ctx *ctx, const rh_mapopt_t *mo, const rh_read_batch_t *in,
                         rh_map_record_t *out, uint64_t out_cap, uint64_t *n_out);
/* The same when a read may have several records: all-vs-all overlapping (the ava presets, RI_M_ALL_CHAINS | RI_M_NO_ADAPTIVE)
 * reports every chain whose score reaches min_chaining_score2 (rmap.cpp:421-500, records :557-586).  One round over the whole
 * reads; the records of read r are out[rec_offsets[r] .. rec_offsets[r + 1]) (n_reads + 1 offsets, may be NULL), a read
 * without a reported chain has its one unmapped record.  Needs `in->name_rank` and the targets' ranks on the context. */
RH_API int  rh_map_batch_multi(rh_ctx *ctx, const rh_mapopt_t *mo, const rh_read_batch_t *in,
                               rh_map_record_t *out, uint64_t out_cap, uint64_t *rec_offsets, uint64_t *n_out);

/* The reference keeps up to two mini-batches in flight (kt_pipeline rmap.cpp:852 with pl_threads = 2, rmap.cpp:831): step 0
 * reads batch k+1 while step 1 maps batch k.  rh_map_submit starts mapping a batch and returns at once; rh_map_wait blocks
 * until that batch's records are in `out` (same contents as rh_map_batch).  Up to RH_MAX_IN_FLIGHT batches may be in flight on
 * one context (each has its own stream and arenas; the upload of one overlaps the kernels of the other); `in`, its arrays and
 * `out` must stay valid until rh_map_wait returns.  submit/wait calls on one context come from one thread at a time. */
#define RH_MAX_IN_FLIGHT 2
typedef struct rh_ticket_s { int32_t slot; uint32_t serial; } rh_ticket_t;
RH_API int  rh_map_submit(rh_ctx *ctx, const rh_mapopt_t *mo, const rh_read_batch_t *in, rh_map_record_t *out, uint64_t out_cap, rh_ticket_t *ticket);
RH_API int  rh_map_wait(rh_ctx *ctx, rh_ticket_t ticket, uint64_t *n_out);

/* Page-locked host memory for read batches (the int16 staging buffers of SURVEY 8 f3): uploads from it run at PCIe speed and
 * asynchronously, overlapped with the kernels of the other sub-batches / batches in flight. */
RH_API void *rh_pinned_alloc(size_t bytes);
/* copy of a device-resident batch (rh_synth_reads_device) in host arrays: samples[offsets[n]], offsets[n + 1], cal_*[n] */
RH_API int   rh_read_batch_to_host(rh_ctx *ctx, const rh_read_batch_t *dev, int16_t *samples, uint64_t *offsets, double *cal_offset, float *cal_scale);
RH_API void  rh_pinned_free(void *p);

/* Counters of the last rh_map_batch call (for the roofline model). */
typedef struct rh_map_stats_s {
	uint64_t n_reads, n_chunks, n_samples_raw, n_samples_used, n_events, n_seeds, n_hits, n_anchors, n_chained;
	double   ms_total;                /* device time of the whole call (hipEvents on the context's stream) */
	double   ms_kernel[24];           /* per-stage device time (HIP events on the context's stream), see rh_stage_name() */
	uint32_t n_launch[24];
	uint64_t n_rmq_class[4];          /* RH_M_RMQ / bw_long: (read, chunk) pairs chained with the trees in LDS rings of 64 / 128 / 512 nodes, or in HBM */
	uint64_t n_dtw_device, n_dtw_host; /* RH_M_DTW_EVALUATE_CHAINS: (read, chunk) pairs whose MAPQ and mapping decision the device settled / that needed the host's logf */
} rh_map_stats_t;
RH_API int  rh_map_last_stats(rh_ctx *ctx, rh_map_stats_t *out);
RH_API const char *rh_stage_name(int i);

/* Stage-level entry points (same kernels as rh_map_batch, one stage at a time) for the parity tests.
 * All buffers are HOST buffers; CSR offsets have n+1 entries. */
/* detect_events revent.c:257 over chunk `chunk` (0-based) of every read, with the running sums carried from
 * chunks 0..chunk-1 (rmap.cpp:412-421).  Also returns l_sig (filtered length, rsig.c:496-503). */
RH_API int  rh_events_batch(rh_ctx *ctx, const rh_mapopt_t *mo, const rh_read_batch_t *in, uint32_t chunk,
                            float *events, uint64_t events_cap, uint64_t *ev_offsets, uint32_t *l_sig);
/* ri_sketch rsketch.c:271 on per-read event arrays */
RH_API int  rh_sketch_batch(rh_ctx *ctx, uint32_t n_reads, const float *events, const uint64_t *ev_offsets,
                            rh_mm128_t *seeds, uint64_t seeds_cap, uint64_t *seed_offsets);
/* collect_seed_hits rmap.cpp:51 (lookup, mid_occ filter, rep_len, expansion, + carried anchors, exact sort) */
RH_API int  rh_seed_batch(rh_ctx *ctx, const rh_mapopt_t *mo, uint32_t n_reads, const rh_mm128_t *seeds, const uint64_t *seed_offsets,
                          const uint32_t *q_offset /* reg->offset per read, may be NULL */,
                          const rh_mm128_t *prev, const uint64_t *prev_offsets /* may be NULL */,
                          rh_mm128_t *anchors, uint64_t anchors_cap, uint64_t *anchor_offsets, int32_t *rep_len);
/* mg_lchain_dp lchain.c:385 (+ backtrack :95, compact_a :214): returns chained anchors and u[] per read */
RH_API int  rh_chain_batch(rh_ctx *ctx, const rh_mapopt_t *mo, uint32_t n_reads, const rh_mm128_t *anchors, const uint64_t *anchor_offsets,
                           rh_mm128_t *chained, uint64_t chained_cap, uint64_t *chained_offsets,
                           uint64_t *u, uint64_t u_cap, uint64_t *u_offsets,
                           rh_mm128_t *prev_out /* the *_a copy = next chunk's prev_anchors, may be NULL */);
/* mm_gen_regs + mm_set_parent + mm_select_sub + mm_set_mapq (hit.c:100-367, 502-539) on top of rh_chain_batch's chains, as ri_map_frag
   runs them (rmap.cpp:346-377); qlen[r] = reg->offset + n_events seeds the region hash.  summary: 10 int32 per read =
   {n_cregs, cnt, score, mapq, qs, qe, rs, re, rid, rev} of creg[0] (what the mapping decision and the record are built from).
   regs (may be NULL; then regs_cap / reg_offsets are ignored): EVERY kept region of every read, 18 int32 each in the order of the reference's
   mm_reg1_t dump {id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0, mapq, rev, hash} (chain.h:27-44; `as` relative to the
   read's chained anchors), read r's regions at reg_offsets[r] .. reg_offsets[r + 1].  Of mo->flag the stage honours RH_M_RMQ, RH_M_ALL_CHAINS
   (mm_select_sub skipped, rmap.cpp:353) and RH_M_HARD_MLEVEL; mo->best_n > 0 keeps secondaries (hit.c:338-367). */
RH_API int  rh_regions_batch(rh_ctx *ctx, const rh_mapopt_t *mo, uint32_t n_reads, const rh_mm128_t *anchors, const uint64_t *anchor_offsets,
                             const int32_t *rep_len, const uint32_t *qlen, int32_t *summary,
                             int32_t *regs, uint64_t regs_cap, uint64_t *reg_offsets);
/* radix_sort_128x ksort.h:101-151 (exact, unstable permutation) on independent segments */
RH_API int  rh_sort128x_batch(rh_ctx *ctx, uint32_t n_seg, rh_mm128_t *a, const uint64_t *offsets);
/* the same sort the way the region keys of mm_gen_regs (hit.c:111-126: score << 32 | count ^ 32-bit hash, practically never equal) take
   it: segments beyond the LDS classes are placed level by level in any order (a sorted order without equal keys is unique), then
   checked; has_ties[s] = 1 for the long segments that do hold equal keys and have to be redone with rh_sort128x_batch's exact passes */
RH_API int  rh_sort128x_any_batch(rh_ctx *ctx, uint32_t n_seg, rh_mm128_t *a, const uint64_t *offsets, uint8_t *has_ties);
/* rh_sort128x_batch with the records travelling as ONE 8-byte word each, the way the mapping path moves anchors whose fields fit: keys
   x = strand << 63 | target << 32 | position with position < 2^lo_bits and target < 2^mid_bits, payloads y < 2^(63 - lo_bits - mid_bits); the result is
   the same radix_sort_128x permutation (ksort.h:101-151).  any_order != 0: as the round loop runs it - long segments level by level in any order first, the
   segments that hold equal keys again with the exact passes */
RH_API int  rh_sort128x_packed_batch(rh_ctx *ctx, uint32_t n_seg, rh_mm128_t *a, const uint64_t *offsets, uint32_t lo_bits, uint32_t mid_bits, int any_order);

/* ------------------------------------------------------------------------------------------- PAF (host) */
/* One PAF line per record exactly as rmap.cpp:740-783 prints it; `mt_ms` fills the mt:f: tag (wall clock in the
 * reference, excluded from parity).  Returns bytes written (excluding NUL) or -1 if cap is too small. */
RH_API int  rh_paf_format(const rh_index *idx, const rh_map_record_t *rec, const char *read_name, double mt_ms,
                          char *buf, size_t cap);

/* ------------------------------------------------------------------------------------------- read container */
/* Reads from a file into the SoA batch (raw int16 + calibration: what step 0, ri_sig_read_frag rmap.cpp:601, hands to step 1):
 *   BLOW5 (binary SLOW5, hasindu2008/slow5lib file format 1.0.0 / 0.2.0 / 0.1.0; replaces ri_read_sig_slow5 rsig.c:478-533 +
 *          slow5lib): records uncompressed, zlib- or zstd-compressed (libzstd.so.1 is loaded at run time: a missing library is an
 *          error), signal raw or svb-zd (StreamVByte of zig-zag deltas; ONE layout: u64 compressed byte count | u32 values | block -
 *          anything else is refused).  Every length field is checked against the record / file before a buffer grows for it.
 *   RHR1  (own minimal container used by tests and the reference harness): magic, u32 n; per read: u32 name_len, name,
 *          u32 n_samples, f64 digitisation, f64 range, f64 offset, i16[n].
 * The format is recognised by its magic. */
typedef struct rh_reads_s rh_reads;
RH_API rh_reads *rh_reads_load(const char *path);
RH_API void      rh_reads_destroy(rh_reads *r);
RH_API uint32_t  rh_reads_n(const rh_reads *r);
RH_API const char *rh_reads_name(const rh_reads *r, uint32_t i);
RH_API int       rh_reads_batch(const rh_reads *r, rh_read_batch_t *out);   /* views into r; valid until destroy */
RH_API int       rh_reads_pinned(const rh_reads *r);                          /* 1: the samples sit in page-locked memory (uploads at PCIe speed, no staging copy) */
/* the reader's pA filter as a count (rsig.c:496-503; FAST5 arithmetic rsig.c:363-374 with in->fast5_ingest): out[r] = l_sig of read r of a HOST
   batch - what rh_read_batch_t::n_filtered wants - for callers whose own reader does not count.  n_threads <= 0: up to 32 host threads */
RH_API int       rh_count_filtered(const rh_read_batch_t *in, uint32_t *out, int n_threads);
RH_API int       rh_reads_write(const char *path, uint32_t n, const char *const *names, const int16_t *samples,
                                const uint64_t *offsets, double digitisation, double range, double offset);
RH_API int       rh_reads_write_blow5(const char *path, uint32_t n, const char *const *names, const int16_t *samples, const uint64_t *offsets,
                                      double digitisation, double range, double offset, double sampling_rate, int compression);
/* (compression & 0xFF = record compression: 0 none, 1 zlib, 2 zstd; bit 8 (0x100) = svb-zd signal compression) */

/* ------------------------------------------------------------------------------------------- synthetic workload */
/* Deterministic, integer-only generator (same bytes on any host): i.i.d. genome, 6-mer pore model ~N(90,12) pA,
 * R9.4-like reads (dwell ~Gamma(2) mean ~8.9 samples/base, gaussian-ish noise), SURVEY §8d. */
typedef struct rh_synth_cfg_s {
	uint64_t model_seed, genome_seed, read_seed;
	uint32_t n_chrom, chrom_len;      /* n_chrom sequences "chr<i>" of chrom_len bases each */
	uint32_t n_samples;               /* raw samples per read (fixed) */
	uint32_t junk_per_1024;           /* reads whose bases are random (unmappable), per 1024 */
	uint32_t noise_q24;               /* noise scale; 0 -> default (sigma ~1.5 pA) */
	double   digitisation, range, offset;
} rh_synth_cfg_t;
RH_API void rh_synth_cfg_init(rh_synth_cfg_t *c);
RH_API int  rh_synth_write_model(const rh_synth_cfg_t *c, const char *path);
RH_API int  rh_synth_write_model_k(const rh_synth_cfg_t *c, const char *path, int k);   /* a model of 4^k levels (4 .. 12; 6 = rh_synth_write_model; R10: 9); the read generators take k from the file */
RH_API int  rh_synth_write_fasta(const rh_synth_cfg_t *c, const char *path);
/* bases of chromosome `chrom` ("chr<chrom+1>") into out[0 .. chrom_len): the same sequence rh_synth_write_fasta writes */
RH_API int  rh_synth_genome(const rh_synth_cfg_t *c, uint32_t chrom, char *out, int n_threads);
/* reads [first, first+n): samples must hold n*n_samples int16; names (optional) n*64 chars */
RH_API int  rh_synth_reads(const rh_synth_cfg_t *c, const char *model_path, uint64_t first, uint32_t n,
                           int16_t *samples, char *names64, int n_threads);
/* Same reads generated directly into this GPU's HBM (bench: "inputs already resident").  out gets device pointers
 * (samples_on_device = 1) owned by the context and valid until the next call or rh_ctx_destroy. */
RH_API int  rh_synth_reads_device(rh_ctx *ctx, const rh_synth_cfg_t *c, const char *model_path, uint64_t first, uint32_t n,
                                  rh_read_batch_t *out);
/* true origin of read `idx` (for "maps to origin" sanity checks): chrom, 0-based start base, strand, junk flag */
RH_API int  rh_synth_origin(const rh_synth_cfg_t *c, uint64_t idx, uint32_t *chrom, uint32_t *pos, uint32_t *strand, uint32_t *junk);

#ifdef __cplusplus
}
#endif
#endif /* RAWHASH_AMD_H */
