/* TEST INFRASTRUCTURE ONLY -- see rh_oracle.h.  CPU restatement of the RawHash2 mapping path.
 * Compile with -ffp-contract=off (the reference's output depends on FMA contraction).
 * All reference citations are relative to /root/reference/src/.
 */
#define _GNU_SOURCE
#include "rh_oracle.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef rh_mm128_t a128;

/* ================================================================== options (roptions.c:4-138, main.cpp:111-210) */
void ro_idxopt_init(rh_idxopt_t *io)
{
	memset(io, 0, sizeof(*io));
	io->e = 8; io->w = 0; io->q = 4; io->n = 0; io->k = 6; io->lev_col = 1; io->b = 14;
	io->diff = 0.35f; io->fine_min = -2.0f; io->fine_max = 2.0f; io->fine_range = 0.4;
}

void ro_mapopt_init(rh_mapopt_t *mo)
{
	memset(mo, 0, sizeof(*mo));
	mo->bp_per_sec = 450; mo->sample_rate = 4000; mo->chunk_size = 4000;
	mo->sample_per_base = (float)mo->sample_rate / mo->bp_per_sec;
	mo->mid_occ_frac = 1e-2f; mo->min_mid_occ = 50; mo->max_mid_occ = 500000;
	mo->max_max_occ = 32767; mo->occ_dist = 500;
	mo->bw = 500; mo->bw_long = 0; mo->max_target_gap_length = 2500; mo->max_query_gap_length = 2500;
	mo->max_chain_iter = 200; mo->max_num_skips = 5; mo->min_num_anchors = 2;
	mo->min_chaining_score = 15; mo->min_chaining_score2 = 0;
	mo->chain_gap_scale = 0.8f; mo->chain_skip_scale = 0.0f;
	mo->rmq_inner_dist = 1000; mo->rmq_size_cap = 100000;	/* roptions.c:65-66 */
	mo->dtw_border_constraint = RH_DTW_BORDER_SPARSE; mo->dtw_fill_method = RH_DTW_FILL_BANDED;	/* roptions.c:96-101, :84 */
	mo->dtw_band_radius_frac = 0.10f; mo->dtw_match_bonus = 0.4f; mo->dtw_min_score = 20.0f; mo->w_bestma = 0.2f;
	mo->mask_level = 0.5f; mo->mask_len = INT_MAX; mo->pri_ratio = 0.3f; mo->best_n = 0; mo->alt_drop = 0.15f;
	mo->w_bestmq = 0.05f; mo->w_bestmc = 0.6f; mo->w_bestq = 0.35f; mo->w_threshold = 0.45f;
	mo->min_events = 50; mo->max_num_chunk = 10; mo->min_mapq = 2;
	mo->window_length1 = 3; mo->window_length2 = 9; mo->threshold1 = 4.0f; mo->threshold2 = 3.5f; mo->peak_height = 0.4f;
}

static void ava_common(rh_idxopt_t *io, rh_mapopt_t *mo, int w, int sc, int sc2, int n_anch, int mapq, int bw)
{
	io->w = w; io->diff = 0.45f;
	mo->min_chaining_score = sc; mo->min_chaining_score2 = sc2; mo->min_num_anchors = n_anch; mo->min_mapq = mapq;
	mo->bw = bw; mo->max_target_gap_length = 2500; mo->max_query_gap_length = 2500;
	io->flag |= RH_I_SIG_TARGET; mo->flag |= RH_M_ALL_CHAINS | RH_M_NO_ADAPTIVE;
	mo->pri_ratio = 0.0f;
}

int ro_set_preset(const char *preset, rh_idxopt_t *io, rh_mapopt_t *mo)
{
	if (preset == 0) { ro_idxopt_init(io); ro_mapopt_init(mo); return 0; }
	if (!strcmp(preset, "sensitive") || !strcmp(preset, "sequence-until")) return 0;
	if (!strcmp(preset, "viral")) {
		io->e = 6; mo->bw = 100; mo->max_target_gap_length = 500; mo->max_query_gap_length = 500;
		mo->max_num_chunk = 5; mo->min_chaining_score = 10; mo->chain_gap_scale = 1.2f; mo->chain_skip_scale = 0.3f;
	} else if (!strcmp(preset, "fast")) {
		io->fine_range = 0.6; mo->min_mapq = 5; mo->min_chaining_score = 10; mo->chain_gap_scale = 0.6f;
	} else if (!strcmp(preset, "faster")) {
		io->e = 11; io->w = 3; io->fine_range = 0.6;
		mo->max_num_chunk = 5; mo->min_mapq = 5; mo->min_chaining_score = 10; mo->chain_gap_scale = 0.6f;
	} else if (!strcmp(preset, "ava-viral")) {
		io->e = 6; mo->chain_gap_scale = 1.2f; mo->chain_skip_scale = 0.3f;
		ava_common(io, mo, 0, 20, 30, 5, 5, 1000);
	} else if (!strcmp(preset, "ava")) {
		ava_common(io, mo, 3, 40, 75, 5, 5, 5000);
	} else if (!strcmp(preset, "ava-sensitive")) {
		ava_common(io, mo, 0, 75, 100, 5, 5, 1000);
	} else if (!strcmp(preset, "ava-large")) {
		io->fine_range = 0.6; mo->chain_gap_scale = 0.6f;
		ava_common(io, mo, 5, 20, 50, 2, 2, 5000);
	} else return -1;
	return 0;
}

/* ================================================================== exact radix_sort_128x (ksort.h:101-151) */
static void ins_sort128(a128 *beg, a128 *end)
{
	for (a128 *i = beg + 1; i < end; ++i) {
		if (i->x < (i - 1)->x) {
			a128 t = *i, *j = i;
			while (j > beg && t.x < (j - 1)->x) { *j = *(j - 1); --j; }
			*j = t;
		}
	}
}

/* In-place MSD "American flag" pass on byte (s/8) followed by recursion; the unstable permutation it produces among
 * equal keys is observable downstream, hence restated step for step. */
static void af_sort128(a128 *beg, a128 *end, int s)
{
	a128 *head[256], *tail[256];
	size_t cnt[256];
	memset(cnt, 0, sizeof(cnt));
	for (a128 *i = beg; i != end; ++i) ++cnt[(i->x >> s) & 255];
	a128 *p = beg;
	for (int c = 0; c < 256; ++c) { head[c] = p; p += cnt[c]; tail[c] = p; }
	for (int c = 0; c < 256;) {
		if (head[c] == tail[c]) { ++c; continue; }
		int d = (int)((head[c]->x >> s) & 255);
		if (d == c) { ++head[c]; continue; }
		a128 carry = *head[c];
		do {
			a128 evicted = *head[d];
			*head[d]++ = carry;
			carry = evicted;
			d = (int)((carry.x >> s) & 255);
		} while (d != c);
		*head[c]++ = carry;
	}
	if (s == 0) return;
	int ns = s > 8 ? s - 8 : 0;
	p = beg;
	for (int c = 0; c < 256; ++c) {
		a128 *b = p, *e = p + cnt[c];
		p = e;
		if (cnt[c] > 64) af_sort128(b, e, ns);
		else if (cnt[c] > 1) ins_sort128(b, e);
	}
}

void ro_radix_sort_128x(a128 *beg, a128 *end)
{
	if (end - beg <= 64) ins_sort128(beg, end);
	else af_sort128(beg, end, 56);
}

static int cmp_u64(const void *a, const void *b)
{
	uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return (x > y) - (x < y);
}

/* ================================================================== index (.ind: rindex.c:545-648 dump, :650-776 load) */
struct ro_index_s {
	int32_t w, e, n, q, k, flag;
	float diff, fine_min, fine_max, fine_range;
	uint32_t n_seq;
	char **name; uint32_t *len;
	uint64_t n_keys, n_pos;
	uint64_t *pos;            /* all bucket p[] arrays concatenated */
	/* open-addressing table over the 32-bit seed hash: slot = {hash+1 (0 = empty), n, val} */
	uint64_t tmask;
	uint32_t *t_hash; uint32_t *t_n; uint64_t *t_val;   /* n==1: val = position word; n>1: val = offset into pos[] */
	float **F, **R; uint32_t *fl, *rl;                  /* RH_I_STORE_SIG: expected signal of every target, forward / reverse (rindex.c:716-726) */
};

static int rd(void *dst, size_t sz, size_t n, FILE *fp) { return fread(dst, sz, n, fp) == n ? 0 : -1; }

ro_index *ro_index_load(const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) return 0;
	char magic[2];
	uint32_t pars[7];
	ro_index *ix = (ro_index*)calloc(1, sizeof(*ix));
	if (rd(magic, 1, 2, fp) || magic[0] != 'R' || magic[1] != 'I' || rd(pars, 4, 7, fp)) goto fail;
	ix->w = pars[0]; ix->e = pars[1]; ix->n = pars[2]; ix->q = pars[3]; ix->k = pars[4]; ix->n_seq = pars[5]; ix->flag = pars[6];
	if (rd(&ix->diff, 4, 1, fp) || rd(&ix->fine_min, 4, 1, fp) || rd(&ix->fine_max, 4, 1, fp) || rd(&ix->fine_range, 4, 1, fp)) goto fail;
	{	/* raw ri_pore_t (32 bytes on x86-64: 2 pointers, u32 n_pore_vals, i16 k, pad, 2 floats), then the two tables */
		unsigned char pore[32];
		uint32_t npv;
		if (rd(pore, 1, 32, fp)) goto fail;
		memcpy(&npv, pore + 16, 4);
		if (fseek(fp, (long)npv * 4 + (long)npv * 12, SEEK_CUR)) goto fail;
	}
	ix->name = (char**)calloc(ix->n_seq ? ix->n_seq : 1, sizeof(char*));
	ix->len = (uint32_t*)calloc(ix->n_seq ? ix->n_seq : 1, 4);
	for (uint32_t i = 0; i < ix->n_seq; ++i) {
		uint8_t l;
		if (rd(&l, 1, 1, fp)) goto fail;
		ix->name[i] = (char*)calloc((size_t)l + 1, 1);
		if (l && rd(ix->name[i], 1, l, fp)) goto fail;
		if (rd(&ix->len[i], 4, 1, fp)) goto fail;
		if (ix->flag & RH_I_STORE_SIG) {
			if (!ix->F) { ix->F = (float**)calloc(ix->n_seq, sizeof(float*)); ix->R = (float**)calloc(ix->n_seq, sizeof(float*)); ix->fl = (uint32_t*)calloc(ix->n_seq, 4); ix->rl = (uint32_t*)calloc(ix->n_seq, 4); }
			if (rd(&ix->fl[i], 4, 1, fp)) goto fail;
			ix->F[i] = (float*)malloc(((size_t)ix->fl[i] + 1) * 4);
			if (ix->fl[i] && rd(ix->F[i], 4, ix->fl[i], fp)) goto fail;
			if (!(ix->flag & RH_I_NO_REV_TARGET)) {
				if (rd(&ix->rl[i], 4, 1, fp)) goto fail;
				ix->R[i] = (float*)malloc(((size_t)ix->rl[i] + 1) * 4);
				if (ix->rl[i] && rd(ix->R[i], 4, ix->rl[i], fp)) goto fail;
			}
		}
	}
	{	/* pass 1: sizes */
		long here = ftell(fp);
		uint64_t nk = 0, np = 0;
		for (int b = 0; b < (1 << 14); ++b) {
			int32_t n; uint32_t size;
			if (rd(&n, 4, 1, fp) || fseek(fp, (long)n * 8, SEEK_CUR) || rd(&size, 4, 1, fp) || fseek(fp, (long)size * 16, SEEK_CUR)) goto fail;
			np += n; nk += size;
		}
		ix->n_keys = nk; ix->n_pos = np;
		uint64_t cap = 16;
		while (cap < nk * 2) cap <<= 1;
		ix->tmask = cap - 1;
		ix->t_hash = (uint32_t*)calloc(cap, 4); ix->t_n = (uint32_t*)calloc(cap, 4); ix->t_val = (uint64_t*)calloc(cap, 8);
		ix->pos = (uint64_t*)malloc((np ? np : 1) * 8);
		fseek(fp, here, SEEK_SET);
		uint64_t base = 0;
		for (int b = 0; b < (1 << 14); ++b) {
			int32_t n; uint32_t size;
			if (rd(&n, 4, 1, fp) || (n && rd(ix->pos + base, 8, n, fp)) || rd(&size, 4, 1, fp)) goto fail;
			for (uint32_t j = 0; j < size; ++j) {
				uint64_t kv[2];
				if (rd(kv, 8, 2, fp)) goto fail;
				uint32_t hash = (uint32_t)((kv[0] >> 1) << 14 | (uint64_t)b);
				uint64_t s = (hash * 0x9E3779B1u) & ix->tmask;
				while (ix->t_n[s]) s = (s + 1) & ix->tmask;
				ix->t_hash[s] = hash;
				if (kv[0] & 1) { ix->t_n[s] = 1; ix->t_val[s] = kv[1]; }
				else { ix->t_n[s] = (uint32_t)kv[1]; ix->t_val[s] = base + (kv[1] >> 32); }
			}
			base += n;
		}
	}
	fclose(fp);
	return ix;
fail:
	fclose(fp);
	ro_index_free(ix);
	return 0;
}

void ro_index_free(ro_index *ix)
{
	if (!ix) return;
	if (ix->name) for (uint32_t i = 0; i < ix->n_seq; ++i) free(ix->name[i]);
	free(ix->name); free(ix->len); free(ix->pos); free(ix->t_hash); free(ix->t_n); free(ix->t_val); free(ix);
}

/* ri_idx_get rindex.c:497-514: (positions, n) of a 32-bit seed hash; n=0 if absent */
const uint64_t *ro_index_get(const ro_index *ix, uint64_t hashval, int *n)
{
	uint32_t h = (uint32_t)hashval;
	uint64_t s = (h * 0x9E3779B1u) & ix->tmask;
	*n = 0;
	if ((hashval >> 32) != 0) return 0;   /* keys only ever hold 32 bits (rsketch.c:7: mask 2^32-1) */
	while (ix->t_n[s]) {
		if (ix->t_hash[s] == h) {
			*n = (int)ix->t_n[s];
			return ix->t_n[s] == 1 ? &ix->t_val[s] : &ix->pos[ix->t_val[s]];
		}
		s = (s + 1) & ix->tmask;
	}
	return 0;
}

uint32_t ro_index_n_seq(const ro_index *ix) { return ix->n_seq; }
const char *ro_index_seq_name(const ro_index *ix, uint32_t i) { return ix->name[i]; }
uint32_t ro_index_seq_len(const ro_index *ix, uint32_t i) { return ix->len[i]; }
uint64_t ro_index_n_keys(const ro_index *ix) { return ix->n_keys; }
void ro_index_params(const ro_index *ix, rh_idxopt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->b = 14; o->w = ix->w; o->e = ix->e; o->n = ix->n; o->q = ix->q; o->k = ix->k; o->flag = ix->flag;
	o->diff = ix->diff; o->fine_min = ix->fine_min; o->fine_max = ix->fine_max; o->fine_range = ix->fine_range;
}

static int cmp_u32(const void *a, const void *b)
{
	uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
	return (x > y) - (x < y);
}

uint64_t ro_index_list(const ro_index *ix, uint64_t *hashes, uint32_t *counts, uint64_t cap)
{
	uint64_t k = 0;
	uint32_t *hs = (uint32_t*)malloc((ix->n_keys ? ix->n_keys : 1) * 4);
	for (uint64_t s = 0; s <= ix->tmask; ++s) if (ix->t_n[s]) hs[k++] = ix->t_hash[s];
	qsort(hs, k, 4, cmp_u32);
	for (uint64_t i = 0; i < k && i < cap; ++i) { int n; ro_index_get(ix, hs[i], &n); hashes[i] = hs[i]; counts[i] = n; }
	free(hs);
	return k;
}

/* ri_idx_cal_max_occ rindex.c:1018-1039 + ri_mapopt_update :1041-1054.  ks_ksmall returns the kk-th smallest
 * (0-based) element, which a full sort reproduces. */
void ro_mapopt_update(rh_mapopt_t *mo, const ro_index *ix)
{
	if (mo->mid_occ <= 0) {
		int32_t thres = INT32_MAX;
		if (mo->mid_occ_frac > 0.) {
			uint64_t n = 0;
			uint32_t *a = (uint32_t*)malloc((ix->n_keys ? ix->n_keys : 1) * 4);
			for (uint64_t s = 0; s <= ix->tmask; ++s) if (ix->t_n[s]) a[n++] = ix->t_n[s];
			qsort(a, n, 4, cmp_u32);
			thres = a[(uint32_t)((1. - mo->mid_occ_frac) * n)] + 1;
			free(a);
		}
		mo->mid_occ = thres;
		if (mo->mid_occ < mo->min_mid_occ) mo->mid_occ = mo->min_mid_occ;
		if (mo->max_mid_occ > mo->min_mid_occ && mo->mid_occ > mo->max_mid_occ) mo->mid_occ = mo->max_mid_occ;
	}
	if (mo->bw_long < mo->bw) mo->bw_long = mo->bw;
}

/* ================================================================== a0: raw -> pA + filter (rsig.c:494-503) */
uint32_t ro_pa_filter(const int16_t *raw, uint64_t n, double cal_offset, float cal_scale, int fast5, float *out)
{
	uint32_t l = 0;
	if (fast5) {	/* rsig.c:346-374: dig / ran / offset are floats, the kept value goes back into the int16_t vector first */
		const float offset = (float)cal_offset;
		for (uint64_t i = 0; i < n; ++i) {
			float pa = (raw[i] + offset) * cal_scale;
			if (pa > 30.0f && pa < 200.0f) { const int16_t t = (int16_t)pa; out[l++] = (float)t; }
		}
		return l;
	}
	for (uint64_t i = 0; i < n; ++i) {
		float pa = (raw[i] + cal_offset) * cal_scale;
		if (pa > 30.0f && pa < 200.0f) out[l++] = pa;
	}
	return l;
}

/* ================================================================== a4-a8: events (revent.c) */
/* revent.c:221-255: running z-score with the sums carried since the start of the read, |z| >= 3 dropped */
static float *normalise(const float *sig, uint32_t s_len, double *mean_sum, double *std_dev_sum, uint32_t *n_sum, uint32_t *n_out)
{
	double sum = *mean_sum, sum2 = *std_dev_sum;
	float *z = (float*)calloc(s_len ? s_len : 1, sizeof(float));
	for (uint32_t i = 0; i < s_len; ++i) { sum += sig[i]; sum2 += sig[i] * sig[i]; }
	*n_sum += s_len; *mean_sum = sum; *std_dev_sum = sum2;
	double mean = sum / (*n_sum);
	double sd = sqrt(sum2 / (*n_sum) - mean * mean);
	uint32_t k = 0;
	for (uint32_t i = 0; i < s_len; ++i) {
		float v = (sig[i] - mean) / sd;
		if (v < 3 && v > -3) z[k++] = v;
	}
	*n_out = k;
	return z;
}

/* revent.c:38-74: two-window t-statistic from serial fp32 prefix sums (:23-36); zero at both borders */
static float *tstat(const float *ps, const float *pss, uint32_t n, uint32_t w)
{
	float *t = (float*)calloc((size_t)n + 1, sizeof(float));
	if (n < 2 * w || w < 2) return t;
	for (uint32_t i = w; i <= n - w; ++i) {
		float s1 = ps[i], q1 = pss[i];
		if (i > w) { s1 -= ps[i - w]; q1 -= pss[i - w]; }
		float s2 = ps[i + w] - ps[i], q2 = pss[i + w] - pss[i];
		float m1 = s1 / w, m2 = s2 / w;
		float var = (q1 / w - m1 * m1 + q2 / w - m2 * m2) / w;
		var = fmaxf(var, FLT_MIN);
		float dm = m2 - m1;
		t[i] = fabsf(dm) / sqrtf(var);
	}
	return t;
}

typedef struct { const float *sig; float thr; uint32_t win, masked_to; int peak_pos; float peak_val; int valid; } detector_t;

/* revent.c:91-150: two coupled peak detectors (short masks long) */
static uint32_t find_peaks(detector_t *d, int nd, uint32_t n, float peak_height, uint32_t *peaks)
{
	uint32_t np = 0;
	for (uint32_t i = 0; i < n; ++i) {
		for (int k = 0; k < nd; ++k) {
			detector_t *q = &d[k];
			if (q->masked_to >= i) continue;
			float cur = q->sig[i];
			if (q->peak_pos == -1) {
				if (cur < q->peak_val) q->peak_val = cur;
				else if (cur - q->peak_val > peak_height) { q->peak_val = cur; q->peak_pos = (int)i; }
			} else {
				if (cur > q->peak_val) { q->peak_val = cur; q->peak_pos = (int)i; }
				if (q->peak_val > q->thr)
					for (int m = k + 1; m < nd; ++m) {
						d[m].masked_to = (uint32_t)q->peak_pos + d[0].win;
						d[m].peak_pos = -1; d[m].peak_val = FLT_MAX; d[m].valid = 0;
					}
				if (q->peak_val - cur > peak_height && q->peak_val > q->thr) q->valid = 1;
				if (q->valid && (i - (uint32_t)q->peak_pos) > q->win / 2) {
					peaks[np++] = (uint32_t)q->peak_pos;
					q->peak_pos = -1; q->peak_val = cur; q->valid = 0;
				}
			}
		}
	}
	return np;
}

static int cmp_f32(const void *a, const void *b)
{
	float x = *(const float*)a, y = *(const float*)b;
	return (x > y) - (x < y);
}

/* revent.c:158-180: mean of the segment after dropping values outside [Q1-IQR, Q3+IQR] (sorts the segment in place) */
static float seg_mean(float *seg, uint32_t len)
{
	qsort(seg, len, sizeof(float), cmp_f32);
	float q1 = seg[len / 4], q3 = seg[3 * len / 4], iqr = q3 - q1, lo = q1 - iqr, hi = q3 + iqr;
	float sum = 0.0f; uint32_t cnt = 0;
	for (uint32_t i = 0; i < len; ++i) if (seg[i] >= lo && seg[i] <= hi) { sum += seg[i]; ++cnt; }
	return cnt > 0 ? sum / cnt : 0;
}

/* revent.c:257-316 */
float *ro_detect_events(uint32_t s_len, const float *sig, uint32_t w1, uint32_t w2, float thr1, float thr2, float peak_height,
                        double *mean_sum, double *std_dev_sum, uint32_t *n_events_sum, uint32_t *n_events)
{
	uint32_t n = 0;
	*n_events = 0;
	float *z = normalise(sig, s_len, mean_sum, std_dev_sum, n_events_sum, &n);
	if (n == 0) { free(z); return 0; }
	float *ps = (float*)calloc((size_t)n + 1, 4), *pss = (float*)calloc((size_t)n + 1, 4);
	for (uint32_t i = 0; i < n; ++i) { ps[i + 1] = ps[i] + z[i]; pss[i + 1] = pss[i] + z[i] * z[i]; }
	float *t1 = tstat(ps, pss, n, w1), *t2 = tstat(ps, pss, n, w2);
	detector_t d[2] = { { t1, thr1, w1, 0, -1, FLT_MAX, 0 }, { t2, thr2, w2, 0, -1, FLT_MAX, 0 } };
	uint32_t *peaks = (uint32_t*)malloc((size_t)n * 4);
	uint32_t np = find_peaks(d, 2, n, peak_height, peaks);
	free(t1); free(t2); free(ps); free(pss);
	float *ev = 0;
	if (np > 0) {	/* revent.c:193-219 */
		uint32_t ne = 0, start = 0, i = 0;
		for (uint32_t p = 0; p < np; ++p) if (peaks[p] > 0 && peaks[p] < n) ++ne;
		ev = (float*)malloc((ne ? ne : 1) * sizeof(float));
		for (uint32_t p = 0; p < np && i < ne; ++p) {
			if (!(peaks[p] > 0 && peaks[p] < n)) continue;
			ev[i++] = seg_mean(z + start, peaks[p] - start);
			start = peaks[p];
		}
		*n_events = ne;
	}
	free(z); free(peaks);
	return ev;
}

/* ================================================================== a9: sketch (rsketch.c) */
static inline uint64_t hash64m(uint64_t key, uint64_t mask)	/* rsketch.c:7-16 */
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

static uint32_t quantise(float s, float fine_min, float fine_max, float fine_range, uint32_t n_buckets)	/* rsketch.c:18-53 */
{
	float lo = -3.0, hi = 3.0, range = hi - lo;
	float c1 = (1 - fine_range) / 2, c2 = fine_range + c1;
	float nrm = (s - lo) / range;
	float a = (fine_min - lo) / range, b = (fine_max - lo) / range;
	float q;
	if (s >= fine_min && s <= fine_max) q = fine_range * ((nrm - a) / (b - a));
	else if (nrm < 0.5) q = fine_range + c1 * nrm;
	else q = c2 + c1 * nrm;
	return (uint32_t)(q * (n_buckets - 1));
}

typedef struct { a128 *a; uint64_t n, cap; int overflow; } outv_t;
static inline void ov_push(outv_t *o, a128 v) { if (o->n < o->cap) o->a[o->n++] = v; else o->overflow = 1; }

/* rsketch.c:143-204 (w == 0) and :55-141 (w > 0, minimizers); a seed = (hash of the last e kept events) with the
 * position of the FIRST of them */
uint64_t ro_sketch(const float *ev, uint32_t len, uint32_t id, int strand, const rh_idxopt_t *ip, a128 *out, uint64_t cap)
{
	const int e = ip->e, w = ip->w; const uint32_t qb = ip->q;
	const uint32_t span = ip->k + e - 1, n_buckets = 1u << qb;
	const uint64_t id_shift = (uint64_t)id << 32, mask = (1ULL << 32) - 1;
	const uint64_t mask_events = (qb * e >= 64) ? ~0ULL : (1ULL << (qb * e)) - 1, mask_q = (1ULL << qb) - 1;
	outv_t o = { out, 0, cap, 0 };
	a128 ring[64];
	memset(ring, 0, sizeof(ring));
	int full = 0; uint32_t rp = 0, last = 0; uint64_t qv = 0;
	if (len == 0) return 0;
	if (w == 0) {
		for (uint32_t f = 0; f < len; ++f) {
			if (f > 0 && fabsf(ev[f] - ev[last]) < ip->diff) continue;
			last = f;
			uint32_t code = quantise(ev[f], ip->fine_min, ip->fine_max, ip->fine_range, n_buckets) & mask_q;
			ring[rp].y = id_shift | (uint64_t)(uint32_t)(f << 1) | (uint64_t)strand;
			if (++rp == (uint32_t)e) { full = 1; rp = 0; }
			qv = f == 0 ? (code & mask_events) : ((qv << qb | code) & mask_events);
			ring[rp].x = (hash64m(qv, mask) << 6) | span;
			if (full) ov_push(&o, ring[rp]);
		}
		return o.overflow ? UINT64_MAX : o.n;
	}
	/* minimizer variant */
	a128 buf[256], min = { UINT64_MAX, UINT64_MAX };
	memset(buf, 0xff, (size_t)w * 16);
	int buf_pos = 0, min_pos = 0, j; uint32_t l = 0;
	for (uint32_t f = 0; f < len; ++f) {
		if (f > 0 && fabsf(ev[f] - ev[last]) < ip->diff) continue;
		++l;
		a128 info = { UINT64_MAX, UINT64_MAX };
		last = f;
		uint32_t code = quantise(ev[f], ip->fine_min, ip->fine_max, ip->fine_range, n_buckets) & mask_q;
		qv = (qv << qb | code) & mask_events;
		ring[rp].y = id_shift | (uint64_t)(uint32_t)(f << 1) | (uint64_t)strand;
		if (++rp == (uint32_t)e) { full = 1; rp = 0; }
		ring[rp].x = hash64m(qv, mask) << 6 | span;
		if (!full) continue;
		info = ring[rp];
		buf[buf_pos] = info;
		if (l == (uint32_t)(w + e - 1) && min.x != UINT64_MAX) {
			for (j = buf_pos + 1; j < w; ++j) if (min.x == buf[j].x && buf[j].y != min.y) ov_push(&o, buf[j]);
			for (j = 0; j < buf_pos; ++j) if (min.x == buf[j].x && buf[j].y != min.y) ov_push(&o, buf[j]);
		}
		if (info.x <= min.x) {
			if (l >= (uint32_t)(w + e) && min.x != UINT64_MAX) ov_push(&o, min);
			min = info; min_pos = buf_pos;
		} else if (buf_pos == min_pos) {
			if (l >= (uint32_t)(w + e - 1) && min.x != UINT64_MAX) ov_push(&o, min);
			for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j) if (min.x >= buf[j].x) { min = buf[j]; min_pos = j; }
			for (j = 0; j <= buf_pos; ++j) if (min.x >= buf[j].x) { min = buf[j]; min_pos = j; }
			if (l >= (uint32_t)(w + e - 1) && min.x != UINT64_MAX) {
				for (j = buf_pos + 1; j < w; ++j) if (min.x == buf[j].x && min.y != buf[j].y) ov_push(&o, buf[j]);
				for (j = 0; j <= buf_pos; ++j) if (min.x == buf[j].x && min.y != buf[j].y) ov_push(&o, buf[j]);
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (min.x != UINT64_MAX) ov_push(&o, min);
	return o.overflow ? UINT64_MAX : o.n;
}

/* ================================================================== a10-a13: seeding (rseed.c:60-154, rmap.cpp:51-126) */
typedef struct { uint32_t n, q_pos, q_span, seg_id; int flt, tandem; const uint64_t *cr; } seedm_t;

/* Returns the sorted anchor array (malloc) and its length; consumes prev (appended before the sort). */
static a128 *collect_anchors(const ro_index *ix, const rh_mapopt_t *mo, const a128 *sd, uint64_t n_sd, uint32_t q_offset,
                             const a128 *prev, uint64_t n_prev, int ava, const char *qname, uint32_t name_rank,
                             int64_t *n_out, int *rep_len, uint64_t *n_hits)
{
	seedm_t *m = (seedm_t*)malloc((n_sd ? n_sd : 1) * sizeof(seedm_t));
	uint64_t nm0 = 0, nm = 0;
	int64_t npos = 0;
	for (uint64_t i = 0; i < n_sd; ++i) {	/* ri_seed_collect_all rseed.c:60-85 */
		int t;
		const uint64_t *cr = ro_index_get(ix, sd[i].x >> 6, &t);
		if (t == 0) continue;
		seedm_t *q = &m[nm0++];
		q->q_pos = (uint32_t)sd[i].y; q->q_span = sd[i].x & 63; q->cr = cr; q->n = t; q->seg_id = (uint32_t)(sd[i].y >> 32);
		q->tandem = q->flt = 0;
		if (i > 0 && sd[i].x >> 6 == sd[i - 1].x >> 6) q->tandem = 1;
		if (i < n_sd - 1 && sd[i].x >> 6 == sd[i + 1].x >> 6) q->tandem = 1;
	}
	int rep_st = 0, rep_en = 0;
	*rep_len = 0;
	for (uint64_t i = 0; i < nm0; ++i) {	/* ri_collect_matches rseed.c:105-154 */
		seedm_t *q = &m[i];
		if (q->n > (uint32_t)mo->mid_occ) q->flt = 1;
		if (q->flt) {
			int st = (int)(q->q_pos >> 1) + 1, en = st + (int)q->q_span + 1;
			if (st > rep_en) { *rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
			else rep_en = en;
		} else { npos += q->n; m[nm++] = *q; }
	}
	*rep_len += rep_en - rep_st;
	a128 *a = (a128*)malloc(((size_t)npos + n_prev + 1) * sizeof(a128));
	int64_t k = 0;
	const uint64_t mask_id_shift = 0x7FFFFFFF80000000ULL;
	for (uint64_t i = 0; i < nm; ++i) {	/* rmap.cpp:74-107 */
		const seedm_t *q = &m[i];
		for (uint32_t j = 0; j < q->n; ++j) {
			uint64_t hit = q->cr[j];
			uint32_t ref_pos = (uint32_t)(hit >> 1) & 0x7FFFFFFFu;
			if (ava) {
				uint32_t tid = (uint32_t)(hit >> 32);
				if (qname ? strcmp(qname, ix->name[tid]) >= 0 : name_rank >= tid) continue;
			}
			a128 *p = &a[k++];
			p->x = (hit & mask_id_shift) | ref_pos;
			if (hit & 1) p->x |= 1ULL << 63;
			p->y = (uint64_t)q->seg_id << 40 | (uint64_t)q->q_span << 32 | (uint32_t)((q->q_pos >> 1) + q_offset);
			if (q->tandem) p->y |= 1ULL << 38;
		}
	}
	free(m);
	*n_hits = (uint64_t)k;
	if (n_prev) { memcpy(a + k, prev, n_prev * sizeof(a128)); k += (int64_t)n_prev; }
	ro_radix_sort_128x(a, a + k);
	*n_out = k;
	return a;
}

/* ================================================================== a14-a16: chaining (lchain.c) */
static inline float log2_approx(float x)	/* lchain.c:23-31 */
{
	union { float f; uint32_t i; } z = { x };
	float l = ((z.i >> 23) & 255) - 128;
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	l += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return l;
}

static inline int32_t pair_score(const a128 *ai, const a128 *aj, int32_t max_dist_t, int32_t max_dist_q, int32_t bw, float pen_gap, float pen_skip)	/* lchain.c:297-356 */
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	if (dq <= 0 || dq > max_dist_q) return INT32_MIN;
	dr = (int32_t)(ai->x - aj->x);
	if (dr == 0 || dr > max_dist_t) return INT32_MIN;
	dd = dr > dq ? dr - dq : dq - dr;
	if (dd > bw || dr > max_dist_q) return INT32_MIN;
	dg = dr < dq ? dr : dq;
	q_span = (int32_t)((aj->y >> 32) & 63);
	sc = q_span < dg ? q_span : dg;
	if (dd || dg > q_span) {
		float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		float lg = dd >= 1 ? log2_approx((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

/* lchain.c:47-75 */
static int64_t bk_end(int32_t max_drop, const a128 *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k)
{
	int64_t i = (int64_t)z[k].y, end_i = -1, max_i = i;
	int32_t max_s = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		int32_t s;
		t[i] = 2;
		end_i = i = p[i];
		s = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (s > max_s) { max_s = s; max_i = i; }
		else if (max_s - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = (int64_t)z[k].y; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
	return max_i;
}

/* mg_chain_backtrack (lchain.c:95-194) + compact_a (:214-281) on filled f / p / v / t (all freed here, and a); what both
 * mg_lchain_dp and mg_lchain_rmq end with */
static a128 *chain_finish(int64_t n, a128 *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t, int min_cnt, int min_sc, int32_t max_drop,
                          int64_t *n_io, a128 **prev_out, int *n_u_out, uint64_t **u_out)
{
	/* backtrack: v[] is reused for the chain members */
	int64_t n_z = 0, n_v = 0; int32_t n_u = 0;
	uint64_t *u = 0;
	for (int64_t i = 0; i < n; ++i) if (f[i] >= min_sc) ++n_z;
	if (n_z > 0) {
		a128 *z = (a128*)malloc(n_z * sizeof(a128));
		int64_t k = 0;
		for (int64_t i = 0; i < n; ++i) if (f[i] >= min_sc) { z[k].x = (uint64_t)(int64_t)f[i]; z[k++].y = (uint64_t)i; }
		ro_radix_sort_128x(z, z + n_z);
		u = (uint64_t*)malloc(n_z * 8);
		memset(t, 0, n * 4);
		for (k = n_z - 1; k >= 0; --k) {
			if (t[z[k].y] != 0) continue;
			int64_t n_v0 = n_v, i;
			int64_t end_i = bk_end(max_drop, z, f, p, t, k);
			for (i = (int64_t)z[k].y; i != end_i; i = p[i]) { v[n_v++] = (int32_t)i; t[i] = 1; }
			int32_t sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
			if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
			else n_v = n_v0;
		}
		free(z);
	}
	free(p); free(f); free(t);
	*n_u_out = n_u; *u_out = u;
	if (n_u == 0) { free(a); free(v); free(u); *u_out = 0; *n_io = 0; return 0; }
	/* compact (lchain.c:214-281) */
	a128 *b = (a128*)malloc(n_v * sizeof(a128)), *pa = (a128*)malloc(n_v * sizeof(a128));
	int64_t k = 0;
	for (int32_t i = 0; i < n_u; ++i) {
		int64_t k0 = k; int32_t ni = (int32_t)u[i];
		for (int32_t j = 0; j < ni; ++j) { b[k] = a[v[k0 + (ni - j - 1)]]; pa[k] = b[k]; ++k; }
	}
	free(v);
	a128 *w = (a128*)malloc(n_u * sizeof(a128));
	k = 0;
	for (int32_t i = 0; i < n_u; ++i) { w[i].x = b[k].x; w[i].y = (uint64_t)k << 32 | (uint64_t)i; k += (int32_t)u[i]; }
	ro_radix_sort_128x(w, w + n_u);
	uint64_t *u2 = (uint64_t*)malloc(n_u * 8);
	a128 *res = (a128*)malloc(n_v * sizeof(a128));
	k = 0;
	for (int32_t i = 0; i < n_u; ++i) {
		int32_t j = (int32_t)w[i].y, cnt = (int32_t)u[j];
		u2[i] = u[j];
		memcpy(&res[k], &b[w[i].y >> 32], cnt * sizeof(a128));
		k += cnt;
	}
	memcpy(u, u2, n_u * 8);
	free(u2); free(w); free(b); free(a);
	*prev_out = pa;
	*n_io = k;
	return res;
}


/* mg_lchain_dp lchain.c:385-530 with mg_chain_backtrack :95-194 and compact_a :214-281.
 * in : a[n] sorted anchors (freed here).  out: returned array of chained anchors (*n updated), *u_out (malloc, n_u),
 *      *prev_out = copy of the chained anchors in pre-sort chain order (next chunk's prev_anchors). */
static a128 *chain_dp(const rh_mapopt_t *mo, float pen_gap, float pen_skip, int64_t *n_io, a128 *a, a128 **prev_out, int *n_u_out, uint64_t **u_out)
{
	int max_dist_t = mo->max_target_gap_length, max_dist_q = mo->max_query_gap_length, bw = mo->bw;
	const int max_skip = mo->max_num_skips, max_iter = mo->max_chain_iter, min_cnt = mo->min_num_anchors, min_sc = mo->min_chaining_score;
	const int32_t max_drop = bw;
	int64_t n = *n_io;
	*u_out = 0; *n_u_out = 0;
	free(*prev_out); *prev_out = 0;
	if (n == 0 || a == 0) { free(a); *n_io = 0; return 0; }
	if (max_dist_t < bw) max_dist_t = bw;
	if (max_dist_q < bw) max_dist_q = bw;
	int64_t *p = (int64_t*)malloc(n * 8);
	int32_t *f = (int32_t*)malloc(n * 4), *v = (int32_t*)malloc(n * 4), *t = (int32_t*)calloc(n, 4);
	int64_t st = 0, max_ii = -1;
	for (int64_t i = 0; i < n; ++i) {
		int64_t max_j = -1, end_j, j;
		int32_t max_f = (int32_t)((a[i].y >> 32) & 63), n_skip = 0;
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist_t)) ++st;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int32_t sc = pair_score(&a[i], &a[j], max_dist_t, max_dist_q, bw, pen_gap, pen_skip);
			if (sc == INT32_MIN) continue;
			sc += f[j];
			if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
			else if (t[j] == (int32_t)i) { if (++n_skip > max_skip) break; }
			if (p[j] >= 0) t[p[j]] = (int32_t)i;
		}
		end_j = j;
		if (max_ii < 0 || a[i].x - a[max_ii].x > (uint64_t)(int64_t)max_dist_t) {
			int32_t mx = INT32_MIN;
			max_ii = -1;
			for (j = i - 1; j >= st; --j) if (mx < f[j]) { mx = f[j]; max_ii = j; }
		}
		if (max_ii >= 0 && max_ii < end_j) {
			int32_t tmp = pair_score(&a[i], &a[max_ii], max_dist_t, max_dist_q, bw, pen_gap, pen_skip);
			if (tmp != INT32_MIN && max_f < tmp + f[max_ii]) { max_f = tmp + f[max_ii]; max_j = max_ii; }
		}
		f[i] = max_f; p[i] = max_j;
		v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : max_f;
		if (max_ii < 0 || (a[i].x - a[max_ii].x <= (uint64_t)(int64_t)max_dist_t && f[max_ii] < f[i])) max_ii = i;
	}
	return chain_finish(n, a, f, p, v, t, min_cnt, min_sc, max_drop, n_io, prev_out, n_u_out, u_out);
}

/* ================================================================== f4: RMQ chaining (lchain.c:532-756 over krmq.h)
 * krmq.h (klib, vendored in the reference) is an AVL tree keyed by (y, i) whose nodes also carry the size of their subtree and a
 * pointer `s` to the element of least `pri` in it.  Equal `pri` values are told apart by position in the tree - an ancestor wins over
 * its descendants, left over right - so the anchor mg_lchain_rmq picks depends on the tree's SHAPE, i.e. on the exact insertion /
 * deletion / rotation history.  Restated operation by operation. */
typedef struct rq_node_s {
	int32_t y; int64_t i; double pri;
	struct rq_node_s *p[2], *s;
	signed char balance; unsigned size;
} rq_node;
#define RQ_DEPTH 64
static inline int rq_cmp(const rq_node *a, const rq_node *b) { return a->y < b->y ? -1 : a->y > b->y ? 1 : (a->i > b->i) - (a->i < b->i); }	/* lc_elem_cmp lchain.c:539 */
#define rq_lt2(a, b) ((a)->pri < (b)->pri)																/* lc_elem_lt2 :540 */
#define rq_size(p) ((p) ? (p)->size : 0u)
#define rq_size_child(q, k) ((q)->p[(k)] ? (q)->p[(k)]->size : 0u)

static rq_node *rq_find(const rq_node *root, const rq_node *x)	/* krmq.h:81-96 */
{
	const rq_node *p = root;
	while (p) { int c = rq_cmp(x, p); if (c < 0) p = p->p[0]; else if (c > 0) p = p->p[1]; else break; }
	return (rq_node*)p;
}
static void rq_interval(const rq_node *root, const rq_node *x, rq_node **lower, rq_node **upper)	/* krmq.h:97-108 */
{
	const rq_node *p = root, *l = 0, *u = 0;
	while (p) { int c = rq_cmp(x, p); if (c < 0) { u = p; p = p->p[0]; } else if (c > 0) { l = p; p = p->p[1]; } else { l = u = p; break; } }
	*lower = (rq_node*)l; *upper = (rq_node*)u;
}
static rq_node *rq_rmq(const rq_node *root, const rq_node *lo, const rq_node *up)	/* krmq.h:110-150: closed interval */
{
	const rq_node *p = root, *path[2][RQ_DEPTH], *min;
	int plen[2] = {0, 0}, pcmp[2][RQ_DEPTH], i, c, lca;
	if (!root) return 0;
	while (p) { c = rq_cmp(lo, p); path[0][plen[0]] = p; pcmp[0][plen[0]++] = c; if (c < 0) p = p->p[0]; else if (c > 0) p = p->p[1]; else break; }
	p = root;
	while (p) { c = rq_cmp(up, p); path[1][plen[1]] = p; pcmp[1][plen[1]++] = c; if (c < 0) p = p->p[0]; else if (c > 0) p = p->p[1]; else break; }
	for (i = 0; i < plen[0] && i < plen[1]; ++i) if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
	if (i == plen[0] || i == plen[1]) return 0;
	lca = i; min = path[0][lca];
	for (i = lca + 1; i < plen[0]; ++i) if (pcmp[0][i] <= 0) {
		if (rq_lt2(path[0][i], min)) min = path[0][i];
		if (path[0][i]->p[1] && rq_lt2(path[0][i]->p[1]->s, min)) min = path[0][i]->p[1]->s;
	}
	for (i = lca + 1; i < plen[1]; ++i) if (pcmp[1][i] >= 0) {
		if (rq_lt2(path[1][i], min)) min = path[1][i];
		if (path[1][i]->p[0] && rq_lt2(path[1][i]->p[0]->s, min)) min = path[1][i]->p[0]->s;
	}
	return (rq_node*)min;
}
static inline void rq_update_min(rq_node *p, const rq_node *q, const rq_node *r)	/* krmq.h:154-157 */
{
	p->s = !q || rq_lt2(p, q->s) ? p : q->s;
	p->s = !r || rq_lt2(p->s, r->s) ? p->s : r->s;
}
static rq_node *rq_rotate1(rq_node *p, int dir)	/* krmq.h:159-170 */
{
	int opp = 1 - dir;
	rq_node *q = p->p[opp], *s = p->s;
	unsigned size_p = p->size;
	p->size -= q->size - rq_size_child(q, dir);
	q->size = size_p;
	rq_update_min(p, p->p[dir], q->p[dir]);
	q->s = s;
	p->p[opp] = q->p[dir];
	q->p[dir] = p;
	return q;
}
static rq_node *rq_rotate2(rq_node *p, int dir)	/* krmq.h:172-192 */
{
	int b1, opp = 1 - dir;
	rq_node *q = p->p[opp], *r = q->p[dir], *s = p->s;
	unsigned size_x_dir = rq_size_child(r, dir);
	r->size = p->size;
	p->size -= q->size - size_x_dir;
	q->size -= size_x_dir + 1;
	rq_update_min(p, p->p[dir], r->p[dir]);
	rq_update_min(q, q->p[opp], r->p[opp]);
	r->s = s;
	p->p[opp] = r->p[dir]; r->p[dir] = p;
	q->p[dir] = r->p[opp]; r->p[opp] = q;
	b1 = dir == 0 ? +1 : -1;
	if (r->balance == b1) { q->balance = 0; p->balance = (signed char)-b1; }
	else if (r->balance == 0) q->balance = p->balance = 0;
	else { q->balance = (signed char)b1; p->balance = 0; }
	r->balance = 0;
	return r;
}
static rq_node *rq_insert(rq_node **root_, rq_node *x)	/* krmq.h:194-242 */
{
	unsigned char stack[RQ_DEPTH];
	rq_node *path[RQ_DEPTH], *bp, *bq, *p, *q, *r = 0;
	int i, which = 0, top, b1, path_len;
	bp = *root_; bq = 0;
	for (p = bp, q = bq, top = path_len = 0; p; q = p, p = p->p[which]) {
		int c = rq_cmp(x, p);
		if (c == 0) return p;
		if (p->balance != 0) { bq = q; bp = p; top = 0; }
		stack[top++] = (unsigned char)(which = (c > 0));
		path[path_len++] = p;
	}
	x->balance = 0; x->size = 1; x->p[0] = x->p[1] = 0; x->s = x;
	if (q == 0) *root_ = x; else q->p[which] = x;
	if (bp == 0) return x;
	for (i = 0; i < path_len; ++i) ++path[i]->size;
	for (i = path_len - 1; i >= 0; --i) { rq_update_min(path[i], path[i]->p[0], path[i]->p[1]); if (path[i]->s != x) break; }
	for (p = bp, top = 0; p != x; p = p->p[stack[top]], ++top) { if (stack[top] == 0) --p->balance; else ++p->balance; }
	if (bp->balance > -2 && bp->balance < 2) return x;
	which = (bp->balance < 0);
	b1 = which == 0 ? +1 : -1;
	q = bp->p[1 - which];
	if (q->balance == b1) { r = rq_rotate1(bp, which); q->balance = bp->balance = 0; }
	else r = rq_rotate2(bp, which);
	if (bq == 0) *root_ = r; else bq->p[bp != bq->p[0]] = r;
	return x;
}
static rq_node *rq_erase(rq_node **root_, const rq_node *x)	/* krmq.h:244-327 (x != NULL) */
{
	rq_node *p, *path[RQ_DEPTH], fake;
	unsigned char dir[RQ_DEPTH];
	int i, d = 0, c;
	fake = **root_; fake.p[0] = *root_; fake.p[1] = 0;
	for (c = -1, p = &fake; c; c = rq_cmp(x, p)) {
		int which = (c > 0);
		dir[d] = (unsigned char)which; path[d++] = p;
		p = p->p[which];
		if (p == 0) return 0;
	}
	for (i = 1; i < d; ++i) --path[i]->size;
	if (p->p[1] == 0) path[d - 1]->p[dir[d - 1]] = p->p[0];
	else {
		rq_node *q = p->p[1];
		if (q->p[0] == 0) {
			q->p[0] = p->p[0]; q->balance = p->balance;
			path[d - 1]->p[dir[d - 1]] = q;
			path[d] = q; dir[d++] = 1;
			q->size = p->size - 1;
		} else {
			rq_node *r;
			int e = d++;
			for (;;) { dir[d] = 0; path[d++] = q; r = q->p[0]; if (r->p[0] == 0) break; q = r; }
			r->p[0] = p->p[0]; q->p[0] = r->p[1]; r->p[1] = p->p[1];
			r->balance = p->balance;
			path[e - 1]->p[dir[e - 1]] = r;
			path[e] = r; dir[e] = 1;
			for (i = e + 1; i < d; ++i) --path[i]->size;
			r->size = p->size - 1;
		}
	}
	for (i = d - 1; i >= 0; --i) rq_update_min(path[i], path[i]->p[0], path[i]->p[1]);	/* (includes the fake root, as the original) */
	while (--d > 0) {
		rq_node *q = path[d];
		int which, other, b1 = 1, b2 = 2;
		which = dir[d]; other = 1 - which;
		if (which) { b1 = -b1; b2 = -b2; }
		q->balance = (signed char)(q->balance + b1);
		if (q->balance == b1) break;
		else if (q->balance == b2) {
			rq_node *r = q->p[other];
			if (r->balance == -b1) path[d - 1]->p[dir[d - 1]] = rq_rotate2(q, which);
			else {
				path[d - 1]->p[dir[d - 1]] = rq_rotate1(q, which);
				if (r->balance == 0) { r->balance = (signed char)-b1; q->balance = (signed char)b1; break; }
				else r->balance = q->balance = 0;
			}
		}
	}
	*root_ = fake.p[0];
	return p;
}
typedef struct { const rq_node *stack[RQ_DEPTH], **top; } rq_itr;
static int rq_itr_find(const rq_node *root, const rq_node *x, rq_itr *it)	/* krmq.h:355-367 */
{
	const rq_node *p = root;
	it->top = it->stack - 1;
	while (p) { int c; *++it->top = p; c = rq_cmp(x, p); if (c < 0) p = p->p[0]; else if (c > 0) p = p->p[1]; else break; }
	return p ? 1 : 0;
}
static int rq_itr_prev(rq_itr *it)	/* krmq_itr_next_bidir(itr, 0), krmq.h:368-384 */
{
	const rq_node *p;
	if (it->top < it->stack) return 0;
	p = (*it->top)->p[0];
	if (p) { for (; p; p = p->p[1]) *++it->top = p; return 1; }
	do { p = *it->top--; } while (it->top >= it->stack && p == (*it->top)->p[0]);
	return it->top < it->stack ? 0 : 1;
}
#define rq_at(it) ((it)->top < (it)->stack ? 0 : *(it)->top)

static inline int32_t sc_simple(const a128 *ai, const a128 *aj, float pen_gap, float pen_skip, int32_t *exact, int32_t *width)	/* comput_sc_simple lchain.c:557-581 */
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	dr = (int32_t)(ai->x - aj->x);
	*width = dd = dr > dq ? dr - dq : dq - dr;
	dg = dr < dq ? dr : dq;
	q_span = (int32_t)((aj->y >> 32) & 63);
	sc = q_span < dg ? q_span : dg;
	if (exact) *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		float lg = dd >= 1 ? log2_approx((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

/* mg_lchain_rmq lchain.c:606-756.  Same contract as chain_dp. */
/* Diagnostic (RO_RMQ_TIE_STATS=1; round 6): how often does krmq_rmq's answer depend on the SHAPE of the tree - two or more live nodes in the query's range
 * sharing the least priority?  [0] queries answered, [1] of them with such a tie, [2] chain_rmq calls (read x chunk), [3] of them with at least one tie. */
static unsigned long long g_rmq_tie[4];
void ro_debug_rmq_ties(unsigned long long *out, int reset) { for (int k = 0; k < 4; ++k) { out[k] = __atomic_load_n(&g_rmq_tie[k], __ATOMIC_RELAXED); if (reset) __atomic_store_n(&g_rmq_tie[k], 0ull, __ATOMIC_RELAXED); } }
static a128 *chain_rmq(int max_dist, int max_dist_inner, int bw, int max_skip, int cap_rmq_size, int min_cnt, int min_sc, float pen_gap, float pen_skip,
                       int64_t *n_io, a128 *a, a128 **prev_out, int *n_u_out, uint64_t **u_out)
{
	const int tie_stats = getenv("RO_RMQ_TIE_STATS") != 0;
	unsigned long long ts_q = 0, ts_t = 0;
	const int32_t max_drop = bw;
	int64_t n = *n_io, i, i0, st = 0, st_inner = 0;
	*u_out = 0; *n_u_out = 0;
	if (n == 0 || a == 0) { free(a); free(*prev_out); *prev_out = 0; *n_io = 0; return 0; }
	free(*prev_out); *prev_out = 0;	/* (compact_a overwrites *_a; every exit below frees or replaces it) */
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	int64_t *p = (int64_t*)malloc(n * 8);
	int32_t *f = (int32_t*)malloc(n * 4), *v = (int32_t*)malloc(n * 4), *t = (int32_t*)calloc(n, 4);
	rq_node *pool = (rq_node*)malloc((size_t)n * 2 * sizeof(rq_node));	/* node j of the outer tree = pool[2j], of the inner tree = pool[2j + 1] */
	rq_node *root = 0, *root_inner = 0;
	for (i = i0 = 0; i < n; ++i) {
		int64_t max_j = -1;
		int32_t q_span = (int32_t)((a[i].y >> 32) & 63), max_f = q_span;
		rq_node s, *q, *r, lo, hi;
		if (i0 < i && a[i0].x != a[i].x) {	/* add in-range anchors */
			for (int64_t j = i0; j < i; ++j) {
				q = &pool[2 * j];
				q->y = (int32_t)a[j].y; q->i = j; q->pri = -(f[j] + 0.5 * pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y));
				rq_insert(&root, q);
				if (max_dist_inner > 0) { r = &pool[2 * j + 1]; *r = *q; rq_insert(&root_inner, r); }
			}
			i0 = i;
		}
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist || rq_size(root) > (unsigned)cap_rmq_size)) {
			s.y = (int32_t)a[st].y; s.i = st;
			if ((q = rq_find(root, &s)) != 0) rq_erase(&root, q);
			++st;
		}
		if (max_dist_inner > 0) {
			while (st_inner < i && (a[i].x >> 32 != a[st_inner].x >> 32 || a[i].x > a[st_inner].x + (uint64_t)max_dist_inner || rq_size(root_inner) > (unsigned)cap_rmq_size)) {
				s.y = (int32_t)a[st_inner].y; s.i = st_inner;
				if ((q = rq_find(root_inner, &s)) != 0) rq_erase(&root_inner, q);
				++st_inner;
			}
		}
		lo.i = INT32_MAX; lo.y = (int32_t)a[i].y - max_dist;
		hi.i = 0; hi.y = (int32_t)a[i].y;
		if ((q = rq_rmq(root, &lo, &hi)) != 0) {
			int32_t sc, exact, width, n_skip = 0;
			int64_t j = q->i;
			if (tie_stats) {	/* the live nodes are the anchors [st, i0) */
				int same = 0;
				for (int64_t k = st; k < i0; ++k) { const rq_node *z = &pool[2 * k]; if (rq_cmp(z, &lo) >= 0 && rq_cmp(z, &hi) <= 0 && z->pri == q->pri) ++same; }
				++ts_q; if (same > 1) ++ts_t;
			}
			sc = f[j] + sc_simple(&a[i], &a[j], pen_gap, pen_skip, &exact, &width);
			if (width <= bw && sc > max_f) { max_f = sc; max_j = j; }
			if (!exact && root_inner && (int32_t)a[i].y > 0) {
				rq_node *lo2, *hi2;
				s.y = (int32_t)a[i].y - 1; s.i = n;
				rq_interval(root_inner, &s, &lo2, &hi2);
				if (lo2) {
					const rq_node *qq;
					int32_t width2;
					rq_itr itr;
					rq_itr_find(root_inner, lo2, &itr);
					while ((qq = rq_at(&itr)) != 0) {
						if (qq->y < (int32_t)a[i].y - max_dist_inner) break;
						j = qq->i;
						sc = f[j] + sc_simple(&a[i], &a[j], pen_gap, pen_skip, 0, &width2);
						if (width2 <= bw) {
							if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
							else if (t[j] == (int32_t)i) { if (++n_skip > max_skip) break; }
							if (p[j] >= 0) t[p[j]] = (int32_t)i;
						}
						if (!rq_itr_prev(&itr)) break;
					}
				}
			}
		}
		f[i] = max_f; p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
	}
	free(pool);
	if (tie_stats) { __atomic_fetch_add(&g_rmq_tie[0], ts_q, __ATOMIC_RELAXED); __atomic_fetch_add(&g_rmq_tie[1], ts_t, __ATOMIC_RELAXED); __atomic_fetch_add(&g_rmq_tie[2], 1ull, __ATOMIC_RELAXED); if (ts_t) __atomic_fetch_add(&g_rmq_tie[3], 1ull, __ATOMIC_RELAXED); }
	return chain_finish(n, a, f, p, v, t, min_cnt, min_sc, max_drop, n_io, prev_out, n_u_out, u_out);
}

/* ================================================================== a17-a19: regions, parents, MAPQ (hit.c) */
typedef struct {
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0;
	uint32_t mapq, rev, hash, strand_retained;
	float alignment_score;	/* RH_M_DTW_EVALUATE_CHAINS (rmap.cpp:128-208) */
} reg_t;

static inline uint64_t hash64u(uint64_t key)	/* hit.c:73-83 */
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

static inline uint32_t wang32(uint32_t key)	/* khash.h:400-409 */
{
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3); key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
	return key;
}

static void reg_coords(reg_t *r, const a128 *a)	/* hit.c:40-64 + :10-29 */
{
	int32_t k = r->as;
	r->rev = (uint32_t)(a[k].x >> 63);
	r->rid = (int32_t)(a[k].x << 1 >> 33);
	r->rs = (int32_t)a[k].x;
	r->re = (int32_t)a[k + r->cnt - 1].x + 1;
	r->qs = (int32_t)a[k].y;
	r->qe = (int32_t)a[k + r->cnt - 1].y + 1;
	r->mlen = r->blen = 0;
	if (r->cnt <= 0) return;
	r->mlen = r->blen = (int32_t)((a[k].y >> 32) & 63);
	for (int i = k + 1; i < k + r->cnt; ++i) {
		int span = (int)((a[i].y >> 32) & 63);
		int tl = (int32_t)a[i].x - (int32_t)a[i - 1].x, ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
		r->blen += tl > ql ? tl : ql;
		r->mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
		r->mlen += tl < ql ? tl : ql;
	}
}

static reg_t *gen_regs(uint32_t hash, int n_u, const uint64_t *u, const a128 *a)	/* hit.c:100-150 */
{
	if (n_u == 0) return 0;
	a128 *z = (a128*)malloc(n_u * sizeof(a128));
	int k = 0;
	for (int i = 0; i < n_u; ++i) {
		uint32_t h = (uint32_t)hash64u((hash64u(a[k].x) + hash64u(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	ro_radix_sort_128x(z, z + n_u);
	for (int i = 0; i < n_u >> 1; ++i) { a128 t = z[i]; z[i] = z[n_u - 1 - i]; z[n_u - 1 - i] = t; }
	reg_t *r = (reg_t*)calloc(n_u, sizeof(reg_t));
	for (int i = 0; i < n_u; ++i) {
		reg_t *q = &r[i];
		q->id = i; q->parent = -1;
		q->score = q->score0 = (int32_t)(z[i].x >> 32);
		q->hash = (uint32_t)z[i].x;
		q->cnt = (int32_t)z[i].y;
		q->as = (int32_t)(z[i].y >> 32);
		reg_coords(q, a);
	}
	free(z);
	return r;
}

static void set_parent(float mask_level, int mask_len, int n, reg_t *r, int hard_mask_level)	/* hit.c:195-263 (no alt contigs on this path) */
{
	if (n <= 0) return;
	for (int i = 0; i < n; ++i) r[i].id = i;
	uint64_t *cov = (uint64_t*)malloc(n * 8);
	int *w = (int*)malloc(n * sizeof(int));
	int k = 1;
	w[0] = 0; r[0].parent = 0;
	for (int i = 1; i < n; ++i) {
		reg_t *ri = &r[i];
		int si = ri->qs, ei = ri->qe, n_cov = 0, uncov = 0, j;
		if (!hard_mask_level) {
			for (j = 0; j < k; ++j) {
				const reg_t *rp = &r[w[j]];
				int sj = rp->qs, ej = rp->qe;
				if (ej <= si || sj >= ei) continue;
				if (sj < si) sj = si;
				if (ej > ei) ej = ei;
				cov[n_cov++] = (uint64_t)sj << 32 | (uint32_t)ej;
			}
			if (n_cov == 0) { j = k; goto decided; }
			qsort(cov, n_cov, 8, cmp_u64);
			int x = si;
			for (j = 0; j < n_cov; ++j) {
				if ((int)(cov[j] >> 32) > x) uncov += (int)(cov[j] >> 32) - x;
				x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
			}
			if (ei > x) uncov += ei - x;
		}
		for (j = 0; j < k; ++j) {
			reg_t *rp = &r[w[j]];
			int sj = rp->qs, ej = rp->qe, mn, mx, ol;
			if (ej <= si || sj >= ei) continue;
			mn = ej - sj < ei - si ? ej - sj : ei - si;
			mx = ej - sj > ei - si ? ej - sj : ei - si;
			ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
			if ((float)ol / mn - (float)uncov / mx > mask_level && uncov <= mask_len) {
				int sci = ri->score;
				ri->parent = rp->parent;
				rp->subsc = rp->subsc > sci ? rp->subsc : sci;
				if (ri->cnt >= rp->cnt) ++rp->n_sub;
				break;
			}
		}
decided:
		if (j == k) { w[k++] = i; ri->parent = i; ri->n_sub = 0; }
	}
	free(cov); free(w);
}

static void sync_regs(int n, reg_t *r)	/* hit.c:312-336 */
{
	if (n <= 0) return;
	int max_id = -1;
	for (int i = 0; i < n; ++i) max_id = max_id > r[i].id ? max_id : r[i].id;
	int n_tmp = max_id + 1;
	int *tmp = (int*)malloc((n_tmp > 0 ? n_tmp : 1) * sizeof(int));
	for (int i = 0; i < n_tmp; ++i) tmp[i] = -1;
	for (int i = 0; i < n; ++i) if (r[i].id >= 0) tmp[r[i].id] = i;
	for (int i = 0; i < n; ++i) {
		reg_t *q = &r[i];
		q->id = i;
		if (q->parent == -2) q->parent = i;
		else if (q->parent >= 0 && tmp[q->parent] >= 0) q->parent = tmp[q->parent];
		else q->parent = -1;
	}
	free(tmp);
}

static void select_sub(float pri_ratio, int best_n, int check_strand, int min_strand_sc, int *n_, reg_t *r)	/* hit.c:338-367 */
{
	if (pri_ratio <= 0.0f || *n_ <= 0) return;
	int k = 0, n = *n_, n_2nd = 0;
	for (int i = 0; i < n; ++i) {
		int p = r[i].parent;
		if (p == i) r[k++] = r[i];
		else if ((r[i].score >= r[p].score * pri_ratio) && n_2nd < best_n) {
			if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) { r[k++] = r[i]; ++n_2nd; }
		} else if (check_strand && n_2nd < best_n && r[i].score > min_strand_sc && r[i].rev != r[p].rev) {
			r[i].strand_retained = 1;
			r[k++] = r[i]; ++n_2nd;
		}
	}
	if (k != n) sync_regs(k, r);
	*n_ = k;
}

static void set_mapq(int n, reg_t *r, int min_chain_sc, int rep_len, int is_dtw)	/* hit.c:502-539 */
{
	if (n == 0) return;
	int64_t sum_sc = 0;
	for (int i = 0; i < n; ++i) if (r[i].parent == r[i].id) sum_sc += r[i].score;
	float uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
	for (int i = 0; i < n; ++i) {
		reg_t *q = &r[i];
		int mapq, subsc;
		float pen_s1 = (q->score > 100 ? 1.0f : 0.01 * q->score) * uniq_ratio;
		float pen_cm = q->cnt > 10 ? 1.0f : 0.1f * q->cnt;
		pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
		subsc = q->subsc > min_chain_sc ? q->subsc : min_chain_sc;
		float x = (float)subsc / q->score0;
		mapq = 0;
		if (is_dtw && q->alignment_score > 0) mapq = (int)(pen_cm * 40.0f * (1.0f - x) * 2 * logf(q->alignment_score));
		else if (!is_dtw) mapq = (int)(pen_cm * 40.0f * (1.0f - x) * logf(q->score));
		mapq -= (int)(4.343f * logf(q->n_sub + 1) + .499f);
		mapq = mapq > 0 ? mapq : 0;
		q->mapq = mapq < 60 ? mapq : 60;
	}
}

/* ================================================================== f4: DTW re-scoring of chains (rmap.cpp:128-208, dtw.cpp) */
#define DTW_DIST(A, B) fabsf((A) - (B))	/* DISTANCE, dtw.cpp:12 (float abs) */
static inline float fmin2(float a, float b) { return b < a ? b : a; }	/* std::min */

static float dtw_global(const float *a, uint32_t a_len, const float *b, uint32_t b_len, int exclude_last)	/* DTW_global dtw.cpp:37-66 */
{
	float *dp = (float*)malloc((a_len ? a_len : 1) * sizeof(float));
	dp[0] = DTW_DIST(a[0], b[0]);
	for (uint32_t j = 1; j < a_len; j++) dp[j] = dp[j - 1] + DTW_DIST(a[j], b[0]);
	for (uint32_t i = 1; i < b_len; i++) {
		float old_left = dp[0];
		dp[0] = dp[0] + DTW_DIST(a[0], b[i]);
		for (uint32_t j = 1; j < a_len; j++) {
			float top = dp[j - 1], left = dp[j], topleft = old_left;
			float center = fmin2(fmin2(top, left), topleft) + DTW_DIST(a[j], b[i]);
			dp[j] = center;
			old_left = left;
		}
	}
	float res = exclude_last ? dp[a_len - 1] - DTW_DIST(a[a_len - 1], b[b_len - 1]) : dp[a_len - 1];
	free(dp);
	return res;
}

/* DTW_global_slantedbanded_antidiagonalwise dtw.cpp:273-523: three rotating anti-diagonal buffers; cells outside the clipped range of
 * an anti-diagonal keep whatever an earlier one left there, and the neighbours read them - the buffers are kept exactly as there */
static float dtw_banded(const float *a, uint32_t a_length, const float *b, uint32_t b_length, int band_radius, int exclude_last)
{
	if (a_length < b_length) { const float *tv = a; uint32_t tl = a_length; a = b; a_length = b_length; b = tv; b_length = tl; }
	int extra = (int)(((a_length - b_length) * band_radius + a_length - 1) / a_length);	/* (unsigned arithmetic, as there) */
	band_radius += extra;
	const int plen = band_radius + (band_radius % 2 == 0 ? 1 : 0), slen = band_radius + (band_radius % 2 == 1 ? 1 : 0);
	const int primary_larger = plen > slen;
	const int dpsize = plen > slen ? plen : slen;
	float *store = (float*)malloc((size_t)dpsize * 3 * sizeof(float));
	float *dp0 = store, *dp1 = store + dpsize, *dp2 = store + dpsize * 2, *tmp;
	for (int i = 0; i < dpsize * 3; i++) store[i] = 1e10;
	int center_row = 0;
	{
		int off = plen / 2, i = (0 + plen / 2) - off, j = (center_row - plen / 2) + off;
		if (j >= 0 && j < (int)b_length && i >= 0 && i < (int)a_length) {
			if (primary_larger) dp2[off] = DTW_DIST(a[i], b[j]); else dp2[off + 1] = DTW_DIST(a[i], b[j]);
		}
		tmp = dp0; dp0 = dp1; dp1 = dp2; dp2 = tmp;
	}
	int prev_inc = 0;
	for (int it = 1; (uint32_t)it < a_length; it++) {
		const int center_column = it;
		int inc = 0;
		if ((int64_t)(center_row + 1) * (int64_t)a_length <= (int64_t)b_length * (int64_t)center_column) { center_row++; inc = 1; }
		if (inc) {
			const int si = center_column + slen / 2 - 1, sj = center_row - slen / 2;
			int o0 = 0; if (si - (int)a_length + 1 > o0) o0 = si - (int)a_length + 1; if (-sj > o0) o0 = -sj;
			int o1 = slen; if (si + 1 < o1) o1 = si + 1; if ((int)b_length - sj < o1) o1 = (int)b_length - sj;
			for (int off = o0; off < o1; off++) {
				const int i = si - off, j = sj + off;
				float top, topleft, left;
				if (primary_larger) { top = dp1[off]; topleft = dp0[off]; left = dp1[off + 1]; }
				else {
					const int is_first = off == 0, is_last = off == slen - 1;
					top = is_first ? 1e10f : dp1[off];
					topleft = is_first && !prev_inc ? 1e10f : dp0[off];
					left = is_last ? 1e10f : dp1[off + 1];
				}
				dp2[off] = fmin2(fmin2(top, left), topleft) + DTW_DIST(a[i], b[j]);
			}
			tmp = dp0; dp0 = dp1; dp1 = dp2; dp2 = tmp;
		}
		const int si = center_column + plen / 2, sj = center_row - plen / 2;
		int o0 = 0; if (si - (int)a_length + 1 > o0) o0 = si - (int)a_length + 1; if (-sj > o0) o0 = -sj;
		int o1 = plen; if (si + 1 < o1) o1 = si + 1; if ((int)b_length - sj < o1) o1 = (int)b_length - sj;
		for (int off = o0; off < o1; off++) {
			const int i = si - off, j = sj + off;
			float top, topleft, left;
			const int is_first = off == 0, is_last = off == plen - 1;
			if (primary_larger) {
				if (inc) { top = is_first ? 1e10f : dp1[off - 1]; topleft = dp0[off]; left = is_last ? 1e10f : dp1[off]; }
				else { top = is_first ? 1e10f : dp1[off - 1]; topleft = is_first ? 1e10f : dp0[off - 1]; left = dp1[off]; }
				dp2[off] = fmin2(fmin2(top, left), topleft) + DTW_DIST(a[i], b[j]);
			} else {
				if (inc) { top = dp1[off]; topleft = dp0[off + 1]; left = dp1[off + 1]; }
				else { top = is_first ? 1e10f : dp1[off]; topleft = is_first && !prev_inc ? 1e10f : dp0[off]; left = dp1[off + 1]; }
				dp2[off + 1] = fmin2(fmin2(top, left), topleft) + DTW_DIST(a[i], b[j]);
			}
		}
		tmp = dp0; dp0 = dp1; dp1 = dp2; dp2 = tmp;
		prev_inc = inc;
	}
	float res = primary_larger ? dp1[plen / 2] : dp1[plen / 2 + 1];
	if (exclude_last) res -= DTW_DIST(a[a_length - 1], b[b_length - 1]);
	free(store);
	return res;
}

/* align_chain rmap.cpp:128-208 */
static void align_chain(reg_t *c, const a128 *anchors, const ro_index *ix, const float *read_events, const rh_mapopt_t *mo, float min_score)
{
	const float *ref = c->rev ? ix->R[c->rid] : ix->F[c->rid];
	float dtw_cost = 0.0f;
	uint32_t n_aligned = 0;
	if (mo->dtw_border_constraint == RH_DTW_BORDER_GLOBAL) {
		const float *revents = ref + c->rs; const uint32_t rlen = (uint32_t)(c->re - c->rs + 1);
		const float *qevents = read_events + c->qs; const uint32_t qlen = (uint32_t)(c->qe - c->qs + 1);
		float max_attainable = qlen * mo->dtw_match_bonus;
		if (max_attainable < min_score) { c->alignment_score = -1e10; return; }
		if (mo->dtw_fill_method == RH_DTW_FILL_FULL) dtw_cost = dtw_global(qevents, qlen, revents, rlen, 0);
		else { int band = (int)(qlen * mo->dtw_band_radius_frac); if (band < 1) band = 1; dtw_cost = dtw_banded(qevents, qlen, revents, rlen, band, 0); }
		n_aligned = qlen;
	} else {
		const uint32_t parts = (uint32_t)c->cnt - 1;
		const uint32_t qfull = (uint32_t)(c->qe - c->qs + 1);
		float cur_max = qfull * mo->dtw_match_bonus;
		for (uint32_t part = 0; part < parts; part++) {
			const a128 *sa = &anchors[part], *ea = &anchors[part + 1];
			const float *revents = ref + (uint32_t)sa->x; const uint32_t rlen = (uint32_t)ea->x - (uint32_t)sa->x + 1;
			const float *qevents = read_events + (uint32_t)sa->y; const uint32_t qlen = (uint32_t)ea->y - (uint32_t)sa->y + 1;
			if (cur_max < min_score) { c->alignment_score = -1e10; return; }
			const int excl = part != parts - 1;
			float sub;
			if (mo->dtw_fill_method == RH_DTW_FILL_FULL) sub = dtw_global(qevents, qlen, revents, rlen, excl);
			else { int band = (int)(qlen * mo->dtw_band_radius_frac); if (band < 1) band = 1; sub = dtw_banded(qevents, qlen, revents, rlen, band, excl); }
			dtw_cost += sub;
			cur_max -= sub;
			n_aligned += qlen;
		}
	}
	c->alignment_score = n_aligned * mo->dtw_match_bonus - dtw_cost;
}

/* ================================================================== a2/a3: per-read driver (rmap.cpp:210-387, :389-599) */
typedef struct {
	uint32_t offset;          /* events accepted so far (reg->offset) */
	a128 *prev; int64_t n_prev;
	reg_t *creg; int n_cregs;
	double mean_sum, std_dev_sum; uint32_t n_sum;
	float *events;            /* RH_M_DTW_EVALUATE_CHAINS: the events of every processed chunk (reg->events, rmap.cpp:237-241) */
} rstate_t;

static uint64_t g_cnt[8];
static pthread_mutex_t g_cnt_mx = PTHREAD_MUTEX_INITIALIZER;

static void map_chunk(const ro_index *ix, const rh_mapopt_t *mo, const rh_idxopt_t *ip, const float *sig, uint32_t s_len, rstate_t *st,
                      const char *qname, uint32_t name_rank, uint64_t cnt[8])
{
	uint32_t n_events = 0;
	float *ev = ro_detect_events(s_len, sig, mo->window_length1, mo->window_length2, mo->threshold1, mo->threshold2, mo->peak_height,
	                             &st->mean_sum, &st->std_dev_sum, &st->n_sum, &n_events);
	cnt[0]++; cnt[1] += s_len; cnt[2] += n_events;
	if (n_events < mo->min_events) { free(ev); return; }
	if (mo->flag & RH_M_DTW_EVALUATE_CHAINS) {
		st->events = (float*)realloc(st->events, ((size_t)st->offset + n_events) * sizeof(float));
		memcpy(st->events + st->offset, ev, n_events * sizeof(float));
	}
	uint64_t cap = (uint64_t)n_events * 2 + 16;
	a128 *sd = (a128*)malloc(cap * sizeof(a128));
	uint64_t n_sd = ro_sketch(ev, n_events, 0, 0, ip, sd, cap);
	free(ev);
	cnt[3] += n_sd;
	int rep_len; int64_t n_a; uint64_t n_hits;
	a128 *a = collect_anchors(ix, mo, sd, n_sd, st->offset, st->prev, (uint64_t)st->n_prev, (mo->flag & RH_M_ALL_CHAINS) ? 1 : 0, qname, name_rank, &n_a, &rep_len, &n_hits);
	free(sd);
	cnt[4] += n_hits; cnt[5] += (uint64_t)n_a;
	float pen_gap = mo->chain_gap_scale * 0.01 * (ip->e + ip->k - 1), pen_skip = mo->chain_skip_scale * 0.01 * (ip->e + ip->k - 1);
	uint64_t *u = 0; int n_u = 0;
	{	/* rmap.cpp:317-342: DP or RMQ chaining, then the optional RMQ re-chaining of the chained anchors with the long bandwidth */
		const int max_gap = mo->max_target_gap_length > mo->max_query_gap_length ? mo->max_target_gap_length : mo->max_query_gap_length;
		if (!(mo->flag & RH_M_RMQ)) a = chain_dp(mo, pen_gap, pen_skip, &n_a, a, &st->prev, &n_u, &u);
		else a = chain_rmq(max_gap, mo->rmq_inner_dist, mo->bw, mo->max_num_skips, mo->rmq_size_cap, mo->min_num_anchors, mo->min_chaining_score, pen_gap, pen_skip, &n_a, a, &st->prev, &n_u, &u);
		if (mo->bw_long > mo->bw) {
			free(u); u = 0; n_u = 0;	/* (the reference leaks the first u[]; its contents are not used again) */
			a = chain_rmq(max_gap, mo->rmq_inner_dist, mo->bw_long, mo->max_num_skips, mo->rmq_size_cap, mo->min_num_anchors, mo->min_chaining_score, pen_gap, pen_skip, &n_a, a, &st->prev, &n_u, &u);
		}
	}
	st->n_prev = n_a > 0 ? n_a : 0;
	if (n_a <= 0) { free(st->prev); st->prev = 0; }
	cnt[6] += (uint64_t)st->n_prev;
	uint32_t hash = 0;
	hash ^= wang32(st->offset + n_events) + wang32(11);
	hash = wang32(hash);
	st->n_cregs = n_u;
	st->creg = gen_regs(hash, n_u, u, a);
	set_parent(mo->mask_level, mo->mask_len, st->n_cregs, st->creg, (mo->flag & RH_M_HARD_MLEVEL) ? 1 : 0);
	if (!(mo->flag & RH_M_ALL_CHAINS)) select_sub(mo->pri_ratio, mo->best_n, 1, mo->max_target_gap_length * 0.8, &st->n_cregs, st->creg);
	if (mo->flag & RH_M_DTW_EVALUATE_CHAINS) {	/* rmap.cpp:355-374 */
		float best_found = 0.0f;
		for (int i = 0; i < st->n_cregs; ++i) {
			reg_t *c = &st->creg[i];
			align_chain(c, a + c->as, ix, st->events, mo, best_found);
			if (c->alignment_score >= mo->dtw_min_score) { if (c->alignment_score > best_found) best_found = c->alignment_score; }
			else if (c->alignment_score < mo->dtw_min_score && c->alignment_score < 0) c->alignment_score = (mo->dtw_min_score > 0) ? 0 : mo->dtw_min_score;
		}
	}
	set_mapq(st->n_cregs, st->creg, mo->min_chaining_score, rep_len, (mo->flag & RH_M_DTW_EVALUATE_CHAINS) ? 1 : 0);
	free(a); free(u);
	st->offset += n_events;
}

/* map_worker_for rmap.cpp:389-599; returns the number of records written to out (>= 1) */
static uint32_t map_read(const ro_index *ix, const rh_mapopt_t *mo, const rh_idxopt_t *ip, uint32_t read_idx, const float *sig, uint32_t qlen,
                         const char *qname, uint32_t name_rank, rh_map_record_t *out, uint32_t out_cap, uint64_t cnt[8])
{
	rstate_t st;
	memset(&st, 0, sizeof(st));
	uint32_t l_chunk = (mo->chunk_size > qlen || (mo->flag & RH_M_NO_ADAPTIVE)) ? qlen : mo->chunk_size;
	uint32_t max_chunk = (mo->flag & RH_M_NO_ADAPTIVE) ? 1 : mo->max_num_chunk;
	uint32_t s_qs, s_qe, c_count, n_maps = 0;
	uint32_t *c_ids = (uint32_t*)malloc((out_cap ? out_cap : 1) * 4);
	const int sig_target = (ip->flag & RH_I_SIG_TARGET) != 0;
	for (s_qs = c_count = 0; s_qs < qlen && c_count < max_chunk; s_qs += l_chunk, ++c_count) {
		s_qe = s_qs + l_chunk;
		if (s_qe > qlen) s_qe = qlen;
		free(st.creg); st.creg = 0; st.n_cregs = 0;
		map_chunk(ix, mo, ip, sig + s_qs, s_qe - s_qs, &st, qname, name_rank, cnt);
		int n_chains = ((mo->flag & RH_M_ALL_CHAINS) || st.n_cregs < 1) ? st.n_cregs : 1;
		const int dtw = (mo->flag & RH_M_DTW_EVALUATE_CHAINS) != 0;
		if (st.n_cregs == 1 && ((int)st.creg[0].mapq >= mo->min_mapq || (dtw && st.creg[0].alignment_score >= mo->dtw_min_score))) { c_ids[n_maps++] = 0; break; }
		float meanC = 0, meanQ = 0;
		for (int c = 0; c < st.n_cregs; ++c) { meanC += st.creg[c].score; meanQ += st.creg[c].mapq; }
		if (st.n_cregs > 0) { meanC /= st.n_cregs; meanQ /= st.n_cregs; }
		for (int ic = 0; ic < n_chains; ++ic) {
			float r_bestmq = 0.0f, r_bestmc = 0.0f, r_bestq = 0.0f, weighted = 0.0f;
			float bestQ = st.creg[ic].mapq, bestC = st.creg[ic].score;
			if (!(mo->flag & RH_M_ALL_CHAINS) && dtw) {	/* rmap.cpp:458-478 */
				float bestA = st.creg[ic].alignment_score, r_bestma;
				if (n_chains == 1) {
					int best_ind = 0;
					for (int i = 1; i < st.n_cregs; ++i) if (st.creg[i].alignment_score > bestA) { bestA = st.creg[i].alignment_score; best_ind = i; }
					ic = best_ind;
					bestQ = st.creg[ic].mapq; bestC = st.creg[ic].score;
				}
				if (bestA >= mo->dtw_min_score) {
					r_bestma = (bestA > 0) ? (bestA / 50.0f) : 0.0f; if (r_bestma < 0) r_bestma = 0.0f;
					r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
					r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
					weighted = mo->w_bestma * r_bestma + mo->w_bestmq * r_bestmq + mo->w_bestmc * r_bestmc;
				}
			} else if (!(mo->flag & RH_M_ALL_CHAINS)) {
				r_bestq = (bestQ > 0) ? (bestQ / 30.0f) : 0.0f; if (r_bestq > 1) r_bestq = 1.0f;
				r_bestmq = (bestQ > 0) ? (1.0f - (meanQ / bestQ)) : 0.0f; if (r_bestmq < 0) r_bestmq = 0.0f;
				r_bestmc = (bestC > 0) ? (1.0f - (meanC / bestC)) : 0.0f; if (r_bestmc < 0) r_bestmc = 0.0f;
				weighted = mo->w_bestq * r_bestq + mo->w_bestmq * r_bestmq + mo->w_bestmc * r_bestmc;
			}
			if (weighted >= mo->w_threshold || ((mo->flag & RH_M_ALL_CHAINS) && st.creg[ic].score >= mo->min_chaining_score2))
				if (n_maps < out_cap) c_ids[n_maps++] = (uint32_t)ic;
		}
		if (n_maps > 0) break;
	}
	if (c_count > 0 && (s_qs >= qlen || c_count == max_chunk)) --c_count;
	float scale = (st.offset == 0) ? 0.0f : (mo->sample_per_base == 0) ? 0.0f : ((float)(c_count + 1) * l_chunk / st.offset) / mo->sample_per_base;
	if (!st.creg) st.n_cregs = 0;
	if (n_maps == 0 && st.creg && (int)st.creg[0].mapq > mo->min_mapq) c_ids[n_maps++] = 0;
	uint32_t n_rec;
	if (n_maps == 0) {
		rh_map_record_t *r = &out[0];
		memset(r, 0, sizeof(*r));
		r->read_idx = read_idx;
		r->read_length = sig_target ? st.offset : (uint32_t)(scale * st.offset);
		r->tag_ci = (int32_t)c_count + 1; r->tag_sl = (int32_t)qlen;
		if (st.n_cregs >= 1) { r->tag_cm = st.creg[0].cnt; r->tag_nc = st.n_cregs; r->tag_s1 = st.creg[0].score; }
		n_rec = 1;
	} else {
		for (uint32_t m = 0; m < n_maps; ++m) {
			const reg_t *c = &st.creg[c_ids[m]];
			rh_map_record_t *r = &out[m];
			memset(r, 0, sizeof(*r));
			r->read_idx = read_idx;
			r->tag_ci = (int32_t)c_count + 1; r->tag_sl = (int32_t)qlen; r->tag_cm = c->cnt; r->tag_nc = st.n_cregs; r->tag_s1 = c->score;
			r->read_length = sig_target ? st.offset : (uint32_t)(scale * c->qe);
			r->ref_id = (uint32_t)c->rid;
			r->read_start_position = sig_target ? (uint32_t)c->qs : (uint32_t)(scale * c->qs);
			r->read_end_position = sig_target ? (uint32_t)c->qe : (uint32_t)(scale * c->qe);
			r->fragment_start_position = c->rev ? (uint32_t)(ix->len[c->rid] + 1 - c->re) : (uint32_t)c->rs;
			r->fragment_length = (uint32_t)(c->re - c->rs + 1);
			r->mapq = (uint8_t)c->mapq; r->rev = c->rev == 1; r->mapped = 1;
		}
		n_rec = n_maps;
	}
	free(st.prev); free(st.creg); free(c_ids); free(st.events);
	return n_rec;
}

/* ================================================================== batch drivers */
static inline const float *batch_scale(const rh_read_batch_t *in, uint32_t r, float *tmp) { *tmp = in->cal_scale ? in->cal_scale[r] : 1.0f; return tmp; }

typedef struct {
	const ro_index *ix; const rh_mapopt_t *mo; rh_idxopt_t ip; const rh_read_batch_t *in; const char *const *names;
	rh_map_record_t **recs; uint32_t *n_recs; uint32_t max_rec;
	uint32_t next; pthread_mutex_t mx;
} mapjob_t;

static void *map_thread(void *arg)
{
	mapjob_t *jb = (mapjob_t*)arg;
	uint64_t cnt[8]; memset(cnt, 0, sizeof(cnt));
	for (;;) {
		pthread_mutex_lock(&jb->mx);
		uint32_t r0 = jb->next; jb->next += 16;
		pthread_mutex_unlock(&jb->mx);
		if (r0 >= jb->in->n_reads) break;
		uint32_t r1 = r0 + 16 < jb->in->n_reads ? r0 + 16 : jb->in->n_reads;
		for (uint32_t r = r0; r < r1; ++r) {
			uint64_t n = jb->in->offsets[r + 1] - jb->in->offsets[r];
			float *sig = (float*)malloc((n ? n : 1) * sizeof(float));
			uint32_t l = ro_pa_filter(jb->in->samples + jb->in->offsets[r], n, jb->in->cal_offset ? jb->in->cal_offset[r] : 0.0,
			                          jb->in->cal_scale ? jb->in->cal_scale[r] : 1.0f, jb->in->fast5_ingest, sig);
			jb->recs[r] = (rh_map_record_t*)malloc(jb->max_rec * sizeof(rh_map_record_t));
			jb->n_recs[r] = map_read(jb->ix, jb->mo, &jb->ip, r, sig, l, jb->names ? jb->names[r] : 0,
			                         jb->in->name_rank ? jb->in->name_rank[r] : 0, jb->recs[r], jb->max_rec, cnt);
			free(sig);
		}
	}
	pthread_mutex_lock(&g_cnt_mx);
	for (int i = 0; i < 8; ++i) g_cnt[i] += cnt[i];
	pthread_mutex_unlock(&g_cnt_mx);
	return 0;
}

int ro_map_batch(const ro_index *ix, const rh_mapopt_t *mo, const rh_read_batch_t *in, const char *const *names,
                 rh_map_record_t *out, uint64_t out_cap, uint64_t *n_out, int n_threads)
{
	mapjob_t jb;
	memset(&jb, 0, sizeof(jb));
	jb.ix = ix; jb.mo = mo; jb.in = in; jb.names = names;
	ro_index_params(ix, &jb.ip);
	jb.max_rec = (mo->flag & RH_M_ALL_CHAINS) ? 4096 : 1;
	jb.recs = (rh_map_record_t**)calloc(in->n_reads ? in->n_reads : 1, sizeof(void*));
	jb.n_recs = (uint32_t*)calloc(in->n_reads ? in->n_reads : 1, 4);
	pthread_mutex_init(&jb.mx, 0);
	memset(g_cnt, 0, sizeof(g_cnt));
	if (n_threads < 1) n_threads = 1;
	pthread_t *th = (pthread_t*)malloc(n_threads * sizeof(pthread_t));
	for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], 0, map_thread, &jb);
	for (int t = 0; t < n_threads; ++t) pthread_join(th[t], 0);
	free(th);
	uint64_t k = 0; int rc = 0;
	for (uint32_t r = 0; r < in->n_reads; ++r) {
		for (uint32_t m = 0; m < jb.n_recs[r]; ++m) { if (k < out_cap) out[k++] = jb.recs[r][m]; else rc = -1; }
		free(jb.recs[r]);
	}
	free(jb.recs); free(jb.n_recs);
	*n_out = k;
	return rc;
}

void ro_last_counters(uint64_t c[8]) { memcpy(c, g_cnt, sizeof(g_cnt)); }

int ro_events_batch(const rh_mapopt_t *mo, const rh_read_batch_t *in, uint32_t chunk, float *events, uint64_t events_cap, uint64_t *ev_offsets, uint32_t *l_sig)
{
	uint64_t k = 0;
	ev_offsets[0] = 0;
	for (uint32_t r = 0; r < in->n_reads; ++r) {
		uint64_t n = in->offsets[r + 1] - in->offsets[r];
		float *sig = (float*)malloc((n ? n : 1) * sizeof(float));
		uint32_t qlen = ro_pa_filter(in->samples + in->offsets[r], n, in->cal_offset ? in->cal_offset[r] : 0.0, in->cal_scale ? in->cal_scale[r] : 1.0f, in->fast5_ingest, sig);
		if (l_sig) l_sig[r] = qlen;
		uint32_t l_chunk = (mo->chunk_size > qlen || (mo->flag & RH_M_NO_ADAPTIVE)) ? qlen : mo->chunk_size;
		double ms = 0, ss = 0; uint32_t ns = 0;
		for (uint32_t c = 0; c <= chunk; ++c) {
			uint64_t s0 = (uint64_t)c * l_chunk;
			if (s0 >= qlen || l_chunk == 0) break;
			uint32_t s1 = s0 + l_chunk > qlen ? qlen : (uint32_t)(s0 + l_chunk), ne = 0;
			float *ev = ro_detect_events(s1 - (uint32_t)s0, sig + s0, mo->window_length1, mo->window_length2, mo->threshold1, mo->threshold2, mo->peak_height, &ms, &ss, &ns, &ne);
			if (c == chunk) {
				if (k + ne > events_cap) { free(ev); free(sig); return -1; }
				if (ne) memcpy(events + k, ev, ne * sizeof(float));
				k += ne;
			}
			free(ev);
		}
		free(sig);
		ev_offsets[r + 1] = k;
	}
	return 0;
}

int ro_sketch_batch(const ro_index *ix, uint32_t n_reads, const float *events, const uint64_t *ev_offsets, a128 *seeds, uint64_t seeds_cap, uint64_t *seed_offsets)
{
	rh_idxopt_t ip; ro_index_params(ix, &ip);
	uint64_t k = 0;
	seed_offsets[0] = 0;
	for (uint32_t r = 0; r < n_reads; ++r) {
		uint64_t n = ro_sketch(events + ev_offsets[r], (uint32_t)(ev_offsets[r + 1] - ev_offsets[r]), 0, 0, &ip, seeds + k, seeds_cap - k);
		if (n == UINT64_MAX) return -1;
		k += n;
		seed_offsets[r + 1] = k;
	}
	return 0;
}

int ro_seed_batch(const ro_index *ix, const rh_mapopt_t *mo, uint32_t n_reads, const a128 *seeds, const uint64_t *seed_offsets,
                  const uint32_t *q_offset, const a128 *prev, const uint64_t *prev_offsets,
                  a128 *anchors, uint64_t anchors_cap, uint64_t *anchor_offsets, int32_t *rep_len)
{
	uint64_t k = 0;
	anchor_offsets[0] = 0;
	for (uint32_t r = 0; r < n_reads; ++r) {
		int64_t n; int rl; uint64_t nh;
		uint64_t np = prev_offsets ? prev_offsets[r + 1] - prev_offsets[r] : 0;
		a128 *a = collect_anchors(ix, mo, seeds + seed_offsets[r], seed_offsets[r + 1] - seed_offsets[r], q_offset ? q_offset[r] : 0,
		                          np ? prev + prev_offsets[r] : 0, np, 0, 0, 0, &n, &rl, &nh);
		if (k + (uint64_t)n > anchors_cap) { free(a); return -1; }
		if (n) memcpy(anchors + k, a, n * sizeof(a128));
		free(a);
		k += n;
		anchor_offsets[r + 1] = k;
		if (rep_len) rep_len[r] = rl;
	}
	return 0;
}

int ro_chain_batch(const ro_index *ix, const rh_mapopt_t *mo, uint32_t n_reads, const a128 *anchors, const uint64_t *anchor_offsets,
                   a128 *chained, uint64_t chained_cap, uint64_t *chained_offsets, uint64_t *u, uint64_t u_cap, uint64_t *u_offsets, a128 *prev_out)
{
	rh_idxopt_t ip; ro_index_params(ix, &ip);
	float pen_gap = mo->chain_gap_scale * 0.01 * (ip.e + ip.k - 1), pen_skip = mo->chain_skip_scale * 0.01 * (ip.e + ip.k - 1);
	uint64_t k = 0, ku = 0;
	chained_offsets[0] = 0; u_offsets[0] = 0;
	for (uint32_t r = 0; r < n_reads; ++r) {
		int64_t n = (int64_t)(anchor_offsets[r + 1] - anchor_offsets[r]);
		a128 *a = (a128*)malloc((n ? n : 1) * sizeof(a128)), *pv = 0; uint64_t *uu = 0; int n_u = 0;
		memcpy(a, anchors + anchor_offsets[r], n * sizeof(a128));
		if (n == 0) { free(a); a = 0; }
		a128 *res;
		{	/* rmap.cpp:317-342 */
			const int max_gap = mo->max_target_gap_length > mo->max_query_gap_length ? mo->max_target_gap_length : mo->max_query_gap_length;
			if (!(mo->flag & RH_M_RMQ)) res = chain_dp(mo, pen_gap, pen_skip, &n, a, &pv, &n_u, &uu);
			else res = chain_rmq(max_gap, mo->rmq_inner_dist, mo->bw, mo->max_num_skips, mo->rmq_size_cap, mo->min_num_anchors, mo->min_chaining_score, pen_gap, pen_skip, &n, a, &pv, &n_u, &uu);
			if (mo->bw_long > mo->bw) {
				free(uu); uu = 0; n_u = 0;
				res = chain_rmq(max_gap, mo->rmq_inner_dist, mo->bw_long, mo->max_num_skips, mo->rmq_size_cap, mo->min_num_anchors, mo->min_chaining_score, pen_gap, pen_skip, &n, res, &pv, &n_u, &uu);
			}
		}
		if (k + (uint64_t)n > chained_cap || ku + (uint64_t)n_u > u_cap) { free(res); free(pv); free(uu); return -1; }
		if (n > 0) { memcpy(chained + k, res, n * sizeof(a128)); if (prev_out) memcpy(prev_out + k, pv, n * sizeof(a128)); }
		if (n_u > 0) memcpy(u + ku, uu, n_u * 8);
		k += n > 0 ? n : 0; ku += n_u;
		free(res); free(pv); free(uu);
		chained_offsets[r + 1] = k; u_offsets[r + 1] = ku;
	}
	return 0;
}

/* a17-a19 as a stage: mm_gen_regs + mm_set_parent + mm_select_sub + mm_set_mapq (hit.c:100-367, 502-539) of the chains of every read,
   as ri_map_frag calls them (rmap.cpp:346-377).  chained / u: output of ro_chain_batch; qlen[r] = reg->offset + n_events (hash seed).
   regs_out: 18 int32 per kept region in the order {id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0,
   mapq, rev, hash}; reg_offsets[n_reads + 1] */
int ro_regions_batch(const rh_mapopt_t *mo, uint32_t n_reads, const a128 *chained, const uint64_t *chained_offsets, const uint64_t *u, const uint64_t *u_offsets,
                     const int32_t *rep_len, const uint32_t *qlen, int32_t *regs_out, uint64_t regs_cap, uint64_t *reg_offsets)
{
	uint64_t k = 0;
	reg_offsets[0] = 0;
	for (uint32_t r = 0; r < n_reads; ++r) {
		int n_u = (int)(u_offsets[r + 1] - u_offsets[r]);
		uint32_t hash = 0;
		hash ^= wang32(qlen[r]) + wang32(11);
		hash = wang32(hash);
		int n = n_u;
		reg_t *g = gen_regs(hash, n_u, u + u_offsets[r], chained + chained_offsets[r]);
		set_parent(mo->mask_level, mo->mask_len, n, g, (mo->flag & RH_M_HARD_MLEVEL) ? 1 : 0);
		if (!(mo->flag & RH_M_ALL_CHAINS)) select_sub(mo->pri_ratio, mo->best_n, 1, mo->max_target_gap_length * 0.8, &n, g);
		set_mapq(n, g, mo->min_chaining_score, rep_len[r], 0);
		if (k + (uint64_t)n > regs_cap) { free(g); return -1; }
		for (int i = 0; i < n; ++i) {
			const reg_t *q = &g[i];
			int32_t *o = regs_out + (k + i) * 18;
			o[0] = q->id; o[1] = q->cnt; o[2] = q->rid; o[3] = q->score; o[4] = q->qs; o[5] = q->qe; o[6] = q->rs; o[7] = q->re; o[8] = q->parent;
			o[9] = q->subsc; o[10] = q->as; o[11] = q->mlen; o[12] = q->blen; o[13] = q->n_sub; o[14] = q->score0; o[15] = (int32_t)q->mapq;
			o[16] = (int32_t)q->rev; o[17] = (int32_t)q->hash;
		}
		k += n;
		reg_offsets[r + 1] = k;
		free(g);
	}
	return 0;
}

int ro_sort128x_batch(uint32_t n_seg, a128 *a, const uint64_t *offsets)
{
	for (uint32_t s = 0; s < n_seg; ++s) ro_radix_sort_128x(a + offsets[s], a + offsets[s + 1]);
	return 0;
}

/* ================================================================== a20: PAF line (rmap.cpp:523-571 tags, :740-783 columns) */
int ro_paf_format(const ro_index *ix, const rh_map_record_t *r, const char *name, double mt_ms, char *buf, size_t cap)
{
	char tags[256];
	int n;
	if (r->mapped || r->tag_nc >= 1)
		snprintf(tags, sizeof(tags), "mt:f:%.6f\tci:i:%d\tsl:i:%d\tcm:i:%d\tnc:i:%d\ts1:i:%d\tsm:f:%.2f", mt_ms, r->tag_ci, r->tag_sl, r->tag_cm, r->tag_nc, r->tag_s1, 0.0);
	else
		snprintf(tags, sizeof(tags), "mt:f:%.6f\tci:i:%d\tsl:i:%d\tcm:i:0\tnc:i:0\ts1:i:0\tsm:f:0", mt_ms, r->tag_ci, r->tag_sl);
	if (r->mapped) {
		if (r->ref_id >= ix->n_seq) { if (cap) buf[0] = 0; return 0; }
		n = snprintf(buf, cap, "%s\t%u\t%u\t%u\t%c\t%s\t%u\t%u\t%u\t%u\t%u\t%u\t%s\n", name, r->read_length, r->read_start_position, r->read_end_position,
		             r->rev ? '-' : '+', ix->name[r->ref_id], ix->len[r->ref_id], r->fragment_start_position, r->fragment_start_position + r->fragment_length,
		             r->read_end_position - r->read_start_position - 1, r->fragment_length, (unsigned)r->mapq, tags);
	} else
		n = snprintf(buf, cap, "%s\t%u\t*\t*\t*\t*\t*\t*\t*\t*\t*\t%u\t%s\n", name, r->read_length, (unsigned)r->mapq, tags);
	return (n < 0 || (size_t)n >= cap) ? -1 : n;
}
