/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the rawhash_amd mapping path.
 *
 * Plain-C restatement of the reference algorithm (RawHash2 v2.1, /root/reference/src); every function in
 * rh_oracle.c cites the reference file:line it follows.  Parity is PINNED: tests/test_oracle_vs_reference.py
 * checks it stage by stage and end to end (PAF) against oracle/_ref (the unmodified reference sources compiled
 * with -ffp-contract=off, see oracle/Makefile) in this container, and tests/golden/ holds the resulting vectors
 * for the GPU box.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (rawhash_amd/) never does.
 *
 * Types (options, records, batches) are shared with the public C ABI header so both sides take identical inputs.
 */
#ifndef RH_ORACLE_H
#define RH_ORACLE_H

#include "../include/rawhash_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ro_index_s ro_index;

ro_index *ro_index_load(const char *ind_path);                 /* rindex.c:650 */
void      ro_index_free(ro_index *ix);
const uint64_t *ro_index_get(const ro_index *ix, uint64_t hashval, int *n);   /* rindex.c:497 */
void      ro_mapopt_update(rh_mapopt_t *mo, const ro_index *ix);              /* rindex.c:1041,1018 */
uint32_t  ro_index_n_seq(const ro_index *ix);
const char *ro_index_seq_name(const ro_index *ix, uint32_t i);
uint32_t  ro_index_seq_len(const ro_index *ix, uint32_t i);
void      ro_index_params(const ro_index *ix, rh_idxopt_t *out);
uint64_t  ro_index_n_keys(const ro_index *ix);
/* canonical listing for index parity tests: keys sorted by hash */
uint64_t  ro_index_list(const ro_index *ix, uint64_t *hashes, uint32_t *counts, uint64_t cap);

void ro_idxopt_init(rh_idxopt_t *io);                           /* roptions.c:4 */
void ro_mapopt_init(rh_mapopt_t *mo);                           /* roptions.c:34 */
int  ro_set_preset(const char *preset, rh_idxopt_t *io, rh_mapopt_t *mo);  /* main.cpp:111 */

/* single-read stage functions */
uint32_t ro_pa_filter(const int16_t *raw, uint64_t n, double cal_offset, float cal_scale, int fast5, float *out);  /* rsig.c:496-503; fast5: rsig.c:346-374 */
float   *ro_detect_events(uint32_t s_len, const float *sig, uint32_t w1, uint32_t w2, float thr1, float thr2, float peak_height,
                          double *mean_sum, double *std_dev_sum, uint32_t *n_events_sum, uint32_t *n_events);   /* revent.c:257; caller frees */
uint64_t ro_sketch(const float *ev, uint32_t len, uint32_t id, int strand, const rh_idxopt_t *ip, rh_mm128_t *out, uint64_t cap); /* rsketch.c:271 */
void     ro_radix_sort_128x(rh_mm128_t *beg, rh_mm128_t *end);   /* ksort.h:146 */

/* batch stage functions: same contracts as the rh_*_batch entry points of include/rawhash_amd.h */
int ro_events_batch(const rh_mapopt_t *mo, const rh_read_batch_t *in, uint32_t chunk,
                    float *events, uint64_t events_cap, uint64_t *ev_offsets, uint32_t *l_sig);
int ro_sketch_batch(const ro_index *ix, uint32_t n_reads, const float *events, const uint64_t *ev_offsets,
                    rh_mm128_t *seeds, uint64_t seeds_cap, uint64_t *seed_offsets);
int ro_seed_batch(const ro_index *ix, const rh_mapopt_t *mo, uint32_t n_reads, const rh_mm128_t *seeds, const uint64_t *seed_offsets,
                  const uint32_t *q_offset, const rh_mm128_t *prev, const uint64_t *prev_offsets,
                  rh_mm128_t *anchors, uint64_t anchors_cap, uint64_t *anchor_offsets, int32_t *rep_len);
int ro_chain_batch(const ro_index *ix, const rh_mapopt_t *mo, uint32_t n_reads, const rh_mm128_t *anchors, const uint64_t *anchor_offsets,
                   rh_mm128_t *chained, uint64_t chained_cap, uint64_t *chained_offsets,
                   uint64_t *u, uint64_t u_cap, uint64_t *u_offsets, rh_mm128_t *prev_out);
int ro_regions_batch(const rh_mapopt_t *mo, uint32_t n_reads, const rh_mm128_t *chained, const uint64_t *chained_offsets, const uint64_t *u, const uint64_t *u_offsets,
                     const int32_t *rep_len, const uint32_t *qlen, int32_t *regs_out, uint64_t regs_cap, uint64_t *reg_offsets);   /* hit.c:100-367, 502-539 as a stage */
int ro_sort128x_batch(uint32_t n_seg, rh_mm128_t *a, const uint64_t *offsets);

/* the whole path: kt_for(map_worker_for) rmap.cpp:700 */
int ro_map_batch(const ro_index *ix, const rh_mapopt_t *mo, const rh_read_batch_t *in, const char *const *names /* only for ava */,
                 rh_map_record_t *out, uint64_t out_cap, uint64_t *n_out, int n_threads);
/* counters accumulated by the last ro_map_batch: chunks, samples used, events, seeds, hits, anchors, chained */
void ro_last_counters(uint64_t c[8]);

int ro_paf_format(const ro_index *ix, const rh_map_record_t *rec, const char *read_name, double mt_ms, char *buf, size_t cap); /* rmap.cpp:740-783 */

#ifdef __cplusplus
}
#endif
#endif
