// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Harness around the *unmodified* reference sources under /root/reference/src.
// It is compiled together with them (see oracle/Makefile, outputs go to
// oracle/_ref/) and gives the tests three things the reference cannot do in
// this image on its own (no HDF5 / POD5 / slow5lib, so `rawhash2` cannot open a
// read file):
//
//   ref_harness index <preset> <ref.fa> <pore.model> <out.ind> [threads]
//        = `rawhash2 -x <preset> -p <pore.model> -d <out.ind> <ref.fa>`   (main.cpp:568)
//   ref_harness sigindex <preset> <reads.rhr> <pore.model> <out.ind> [threads]
//        = `rawhash2 -x <preset> -p <pore.model> -d <out.ind> <reads>` for the signal-target presets (ava*)
//   ref_harness map <preset> <ref.ind> <reads.rhr> [threads]  > out.paf
//        = `rawhash2 -x <preset> <ref.ind> <reads>`: feeds an in-memory batch
//          to the reference's own step-1/step-2 pipeline callbacks
//          (map_worker_pipeline, rmap.cpp:661) so the PAF text is produced by
//          the reference's printer, not by us.
//   ref_harness dump <preset> <ref.ind> <reads.rhr> <out.bin>
//        per-chunk stage dumps (events, seeds, sorted anchors, chains, regions)
//        obtained by calling the reference's stage functions in the order
//        ri_map_frag (rmap.cpp:210) calls them.
//   ref_harness idxdump <ref.ind> <out.bin>
//        canonical (hash, n, positions[]) listing of a loaded index.
//
// The two reference translation units with file-static entry points are
// reached by #including them where they lie (nothing is copied into the repo).
//
// Reads come from our own trivial container ("RHR1", see rawhash_amd/csrc/rh_reads.h);
// the raw->pA conversion below restates ri_read_sig_slow5 (rsig.c:496-503)
// because the reader itself is compiled out with -DNSLOW5RH.

#define main ref_rawhash2_main
#include "main.cpp"      // ri_set_opt presets (main.cpp:111), option defaults
#undef main
#include "rmap.cpp"      // map_worker_pipeline / map_worker_for / collect_seed_hits (static)

#include <vector>
#include <string>
#include <algorithm>
#include <fcntl.h>
#include <unistd.h>

struct rhr_read { std::string name; std::vector<int16_t> raw; double dig, range, offset; };

static bool load_rhr(const char *fn, std::vector<rhr_read> &out)
{
	FILE *fp = fopen(fn, "rb");
	if (!fp) { fprintf(stderr, "cannot open %s\n", fn); return false; }
	char magic[4]; uint32_t n;
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "RHR1", 4) != 0 || fread(&n, 4, 1, fp) != 1) { fclose(fp); return false; }
	out.resize(n);
	for (uint32_t i = 0; i < n; ++i) {
		uint32_t l, ns;
		if (fread(&l, 4, 1, fp) != 1) return false;
		out[i].name.resize(l);
		if (l && fread(&out[i].name[0], 1, l, fp) != l) return false;
		if (fread(&ns, 4, 1, fp) != 1) return false;
		if (fread(&out[i].dig, 8, 1, fp) != 1 || fread(&out[i].range, 8, 1, fp) != 1 || fread(&out[i].offset, 8, 1, fp) != 1) return false;
		out[i].raw.resize(ns);
		if (ns && fread(out[i].raw.data(), 2, ns, fp) != ns) return false;
	}
	fclose(fp);
	return true;
}

// rsig.c:496-503 (slow5 record -> float pA, keep 30 < pA < 200)
static ri_sig_t *to_sig(const rhr_read &r, uint32_t rid)
{
	ri_sig_t *s = (ri_sig_t*)calloc(1, sizeof(ri_sig_t));
	s->name = strdup(r.name.c_str());
	s->rid = rid;
	static const bool fast5 = getenv("RH_FAST5_INGEST") && atoi(getenv("RH_FAST5_INGEST"));
	if (fast5) {	// rsig.c:346-374 (the FAST5 reader, compiled out here with HDF5): float dig / ran / offset, the kept pA value goes back
		// into the int16_t signal vector - truncated - and only then becomes the float signal
		float dig = (float)r.dig, ran = (float)r.range, offset = (float)r.offset;
		std::vector<int16_t> sig(r.raw.begin(), r.raw.end());
		uint32_t l_sig = 0;
		float scale = ran / dig;
		float pa = 0;
		for (size_t i = 0; i < sig.size(); i++) {
			pa = (sig[i] + offset) * scale;
			if (pa > 30.0f && pa < 200.0f) sig[l_sig++] = pa;
		}
		s->sig = (float*)calloc(l_sig ? l_sig : 1, sizeof(float));
		s->l_sig = l_sig;
		std::copy(sig.begin(), sig.begin() + l_sig, s->sig);
		return s;
	}
	float *sigF = (float*)malloc((r.raw.size() + 1) * sizeof(float));
	uint32_t l_sig = 0;
	float pa = 0.0f;
	float scale = r.range / r.dig;
	for (size_t i = 0; i < r.raw.size(); ++i) {
		pa = (r.raw[i] + r.offset) * scale;
		if (pa > 30.0f && pa < 200.0f) sigF[l_sig++] = pa;
	}
	s->sig = (float*)calloc(l_sig ? l_sig : 1, sizeof(float));
	s->l_sig = l_sig;
	memcpy(s->sig, sigF, l_sig * sizeof(float));
	free(sigF);
	return s;
}

static void apply_overrides(ri_idxopt_t *ipt, ri_mapopt_t *opt)
{	// optional env overrides so tests can reach non-preset corners without a CLI parser
	const char *s;
	if ((s = getenv("RH_MAX_CHUNKS"))) opt->max_num_chunk = atoi(s);
	if ((s = getenv("RH_CHUNK_SIZE"))) opt->chunk_size = atoi(s);
	if ((s = getenv("RH_MIN_MAPQ"))) opt->min_mapq = atoi(s);
	if ((s = getenv("RH_MIN_ANCHORS"))) opt->min_num_anchors = atoi(s);              // --min-anchors (main.cpp:19)
	if ((s = getenv("RH_MID_OCC"))) opt->mid_occ = atoi(s);
	if ((s = getenv("RH_W"))) ipt->w = atoi(s);
	if ((s = getenv("RH_E"))) ipt->e = atoi(s);
	if ((s = getenv("RH_NO_ADAPTIVE")) && atoi(s)) opt->flag |= RI_M_NO_ADAPTIVE;   // --disable-adaptive (main.cpp:369)
	if ((s = getenv("RH_RMQ")) && atoi(s)) opt->flag |= RI_M_RMQ;                    // --rmq (main.cpp:330)
	if ((s = getenv("RH_RMQ_INNER_DIST"))) opt->rmq_inner_dist = atoi(s);           // --rmq-inner-dist (:331)
	if ((s = getenv("RH_RMQ_SIZE_CAP"))) opt->rmq_size_cap = atoi(s);               // --rmq-size-cap (:332)
	if ((s = getenv("RH_BW_LONG"))) opt->bw_long = atoi(s);                         // --bw-long (:333)
	if ((s = getenv("RH_STORE_SIG")) && atoi(s)) ipt->flag |= RI_I_STORE_SIG;       // --store-sig (:367)
	if ((s = getenv("RH_DTW")) && atoi(s)) opt->flag |= RI_M_DTW_EVALUATE_CHAINS;   // --dtw-evaluate-chains (:372)
	if ((s = getenv("RH_DTW_BORDER"))) opt->dtw_border_constraint = (uint32_t)atoi(s);   // --dtw-border-constraint global = 0 | sparse = 1 (:374)
	if ((s = getenv("RH_DTW_FILL"))) opt->dtw_fill_method = (uint32_t)atoi(s);      // --dtw-fill-method full = 0 | banded = 1 (:384)
	if ((s = getenv("RH_DTW_BAND_FRAC"))) opt->dtw_band_radius_frac = (float)atof(s);
	if ((s = getenv("RH_DTW_MIN_SCORE"))) opt->dtw_min_score = (float)atof(s);      // --dtw-min-score (:390)
	if ((s = getenv("RH_R10")) && atoi(s)) {                                       // --r10 (main.cpp:396-406), field for field
		ipt->k = 9;
		ipt->window_length1 = 3; ipt->window_length2 = 6; ipt->threshold1 = 6.5f; ipt->threshold2 = 4.0f; ipt->peak_height = 0.2f;
		opt->window_length1 = 3; opt->window_length2 = 6; opt->threshold1 = 6.5f; opt->threshold2 = 4.0f; opt->peak_height = 0.2f;
		opt->chain_gap_scale = 1.2f;
	}
}

static int set_presets(const char *preset, ri_idxopt_t *ipt, ri_mapopt_t *opt)
{
	ri_set_opt(0, ipt, opt);
	if (strcmp(preset, "default") != 0 && ri_set_opt(preset, ipt, opt) < 0) {
		fprintf(stderr, "unknown preset %s\n", preset);
		return -1;
	}
	apply_overrides(ipt, opt);
	return 0;
}

static int cmd_index(int argc, char **argv)
{
	if (argc < 6) return 2;
	ri_idxopt_t ipt; ri_mapopt_t opt;
	if (set_presets(argv[2], &ipt, &opt) < 0) return 1;
	int n_threads = argc > 6 ? atoi(argv[6]) : 3;
	ri_idx_reader_t *rdr = ri_idx_reader_open(argv[3], &ipt, argv[5]);
	if (!rdr) { fprintf(stderr, "cannot open %s\n", argv[3]); return 1; }
	ri_pore_t pore; pore.pore_vals = NULL; pore.pore_inds = NULL; pore.max_val = -5000.0; pore.min_val = 5000.0;
	load_pore(argv[4], ipt.k, ipt.lev_col, &pore);
	if (!pore.pore_vals) { fprintf(stderr, "cannot parse pore model\n"); return 1; }
	ri_idx_t *ri;
	while ((ri = ri_idx_reader_read(rdr, &pore, n_threads, 1)) != 0) {
		ri_idx_stat(ri);
		ri_idx_destroy(ri);
	}
	ri_idx_reader_close(rdr);
	return 0;
}

// ref_harness sigindex <preset> <reads.rhr> <pore.model> <out.ind> [threads]
//   = `rawhash2 -x <preset> -p <pore.model> -d <out.ind> <reads>` for the signal-target (Rawsamble, RI_I_SIG_TARGET) presets.
// ri_idx_siggen (rindex.c:927) cannot open a read file in this image, so the harness walks the reads itself and makes the
// calls of worker_sig_pipeline (rindex.c:239-310) in its order: register the read as a target (step 0), detect_events over
// the whole signal + ri_sketch with the read's id (step 1), ri_idx_add (step 2); then ri_idx_sort (= ri_idx_post) and
// ri_idx_dump.  Every number in the file comes from the reference's own functions.
void ri_idx_sort(ri_idx_t *ri, int n_threads);   // rindex.c:496 (defined there, not declared in rindex.h)
static int cmd_sigindex(int argc, char **argv)
{
	if (argc < 6) return 2;
	ri_idxopt_t ipt; ri_mapopt_t opt;
	if (set_presets(argv[2], &ipt, &opt) < 0) return 1;
	if (!(ipt.flag & RI_I_SIG_TARGET)) { fprintf(stderr, "preset %s does not build a signal-target index\n", argv[2]); return 1; }
	int n_threads = argc > 6 ? atoi(argv[6]) : 3;
	std::vector<rhr_read> reads;
	if (!load_rhr(argv[3], reads)) { fprintf(stderr, "bad reads file\n"); return 1; }
	ri_pore_t pore; pore.pore_vals = NULL; pore.pore_inds = NULL; pore.max_val = -5000.0; pore.min_val = 5000.0;
	load_pore(argv[4], ipt.k, ipt.lev_col, &pore);
	if (!pore.pore_vals) { fprintf(stderr, "cannot parse pore model\n"); return 1; }
	ri_idx_t *ri = ri_idx_init(ipt.diff, ipt.b, ipt.w, ipt.e, ipt.n, ipt.q, ipt.k, ipt.fine_min, ipt.fine_max, ipt.fine_range, ipt.flag);
	ri->pore = (ri_pore_t*)ri_kmalloc(ri->km, sizeof(ri_pore_t));
	memcpy(ri->pore, &pore, sizeof(ri_pore_t));
	ri->pore->pore_vals = (float*)ri_kmalloc(ri->km, pore.n_pore_vals * sizeof(float));
	memcpy(ri->pore->pore_vals, pore.pore_vals, pore.n_pore_vals * sizeof(float));
	ri->pore->pore_inds = (ri_porei_t*)ri_kmalloc(ri->km, pore.n_pore_vals * sizeof(ri_porei_t));
	memcpy(ri->pore->pore_inds, pore.pore_inds, pore.n_pore_vals * sizeof(ri_porei_t));
	ri->window_length1 = ipt.window_length1; ri->window_length2 = ipt.window_length2;
	ri->threshold1 = ipt.threshold1; ri->threshold2 = ipt.threshold2; ri->peak_height = ipt.peak_height;
	uint64_t sum_len = 0;
	ri->sig = (ri_sig_t*)ri_kcalloc(ri->km, reads.size() ? reads.size() : 1, sizeof(ri_sig_t));
	for (size_t i = 0; i < reads.size(); ++i) {
		ri_sig_t *t = to_sig(reads[i], 0);
		ri_sig_t *sig = &ri->sig[ri->n_seq];
		sig->name = (char*)ri_kmalloc(ri->km, strlen(t->name) + 1);
		strcpy(sig->name, t->name);
		sig->l_sig = t->l_sig; sig->offset = sum_len; sum_len += t->l_sig;
		t->rid = ri->n_seq++;
		if (t->l_sig > 0) {
			uint32_t s_len = 0, n_events_sum = 0;
			double s_sum = 0, s_std = 0;
			mm128_v a = {0, 0, 0};
			float *s_values = detect_events(0, t->l_sig, t->sig, ri->window_length1, ri->window_length2, ri->threshold1, ri->threshold2, ri->peak_height, &s_sum, &s_std, &n_events_sum, &s_len);
			ri_sketch(0, s_values, t->rid, 0, s_len, ri->diff, ri->w, ri->e, ri->n, ri->q, ri->k, ri->fine_min, ri->fine_max, ri->fine_range, &a, 0);
			if (s_values) free(s_values);
			ri_idx_add(ri, a.n, a.a);
			ri_kfree(0, a.a);
		}
		free(t->sig); free(t->name); free(t);
	}
	ri_idx_sort(ri, n_threads);
	FILE *fp = fopen(argv[5], "wb");
	if (!fp) { fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
	ri_idx_dump(fp, ri);
	fclose(fp);
	return 0;
}

static ri_idx_t *load_index(const char *fn, ri_idxopt_t *ipt)
{
	ri_idx_reader_t *rdr = ri_idx_reader_open(fn, ipt, 0);
	if (!rdr || !rdr->is_idx) { fprintf(stderr, "%s is not an index\n", fn); return 0; }
	ri_pore_t pore; pore.pore_vals = NULL; pore.pore_inds = NULL;
	ri_idx_t *ri = ri_idx_reader_read(rdr, &pore, 1, 1);
	ri_idx_reader_close(rdr);
	return ri;
}

static int cmd_map(int argc, char **argv)
{
	if (argc < 5) return 2;
	ri_idxopt_t ipt; ri_mapopt_t opt;
	if (set_presets(argv[2], &ipt, &opt) < 0) return 1;
	// thread counts: one number, or a comma-separated list (the index is loaded once, the reads are mapped once per
	// count; the PAF of the first run goes to stdout, the others are timed only)
	std::vector<int> threads;
	{
		const char *p = argc > 5 ? argv[5] : "1";
		while (*p) { int v = atoi(p); threads.push_back(v > 0 ? v : 1); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
		if (threads.empty()) threads.push_back(1);
	}
	double t_load = ri_realtime();
	ri_idx_t *ri = load_index(argv[3], &ipt);
	if (!ri) return 1;
	t_load = ri_realtime() - t_load;
	ri_mapopt_update(&opt, ri);
	std::vector<rhr_read> reads;
	if (!load_rhr(argv[4], reads)) { fprintf(stderr, "bad reads file\n"); return 1; }
	fprintf(stderr, "[ref_harness] index loaded in %.3f s, %zu reads\n", t_load, reads.size());

	int saved_stdout = -1;
	for (size_t run = 0; run < threads.size(); ++run) {
		pipeline_mt pl;
		memset(&pl, 0, sizeof(pl));
		pl.n_threads = threads[run];
		pl.opt = &opt; pl.ri = ri; pl.su_stop = 0;
		if (run == 1) { fflush(stdout); saved_stdout = dup(1); int nul = open("/dev/null", O_WRONLY); dup2(nul, 1); close(nul); }
		double t_map = 0;
		size_t b0 = 0;
		while (b0 < reads.size()) {
			// what step 0 does (rmap.cpp:601-690): reads are taken until their filtered signal adds up to mini_batch_size (-K, 500 M samples)
			std::vector<ri_sig_t*> sigs;
			int64_t sum = 0;
			while (b0 < reads.size() && sum < opt.mini_batch_size) { ri_sig_t *sg = to_sig(reads[b0], pl.n_processed++); sum += sg->l_sig; sigs.push_back(sg); ++b0; }
			const size_t n = sigs.size();
			step_mt *s = (step_mt*)calloc(1, sizeof(step_mt));
			s->n_sig = (int)n;
			s->sig = (ri_sig_t**)calloc(n, sizeof(ri_sig_t*));
			for (size_t i = 0; i < n; ++i) s->sig[i] = sigs[i];
			s->p = &pl;
			s->buf = (ri_tbuf_t**)calloc(pl.n_threads, sizeof(ri_tbuf_t*));
			for (int i = 0; i < pl.n_threads; ++i) s->buf[i] = ri_tbuf_init();
			s->reg = (ri_reg1_t**)calloc(n, sizeof(ri_reg1_t*));
			for (size_t i = 0; i < n; ++i) s->reg[i] = (ri_reg1_t*)calloc(1, sizeof(ri_reg1_t));
			double t0 = ri_realtime();
			map_worker_pipeline(&pl, 1, s);   // kt_for(map_worker_for)
			t_map += ri_realtime() - t0;
			map_worker_pipeline(&pl, 2, s);   // PAF printer, frees the batch
		}
		fflush(stdout);
		fprintf(stderr, "[ref_harness] mapped %zu reads, map phase %.6f s, threads %d, mid_occ %d\n", reads.size(), t_map, pl.n_threads, opt.mid_occ);
	}
	if (saved_stdout >= 0) { fflush(stdout); dup2(saved_stdout, 1); close(saved_stdout); }
	ri_idx_destroy(ri);
	return 0;
}

// ---------------------------------------------------------------- stage dumps
static FILE *g_dump;
static void dump_rec(uint32_t tag, uint32_t read, uint32_t chunk, uint32_t count, uint32_t esz, const void *data)
{
	uint32_t h[5] = {tag, read, chunk, count, esz};
	fwrite(h, 4, 5, g_dump);
	if (count && data) fwrite(data, esz, count, g_dump);
}
enum { T_EVENTS = 1, T_SEEDS = 2, T_ANCHORS = 3, T_CHAIN_U = 4, T_CHAIN_A = 5, T_REGS = 6, T_SCALARS = 7, T_FINAL = 8, T_SIG = 9 };

static void dump_regs(uint32_t rd, uint32_t c, int n, const mm_reg1_t *r)
{
	std::vector<int32_t> v;
	for (int i = 0; i < n; ++i) {
		const mm_reg1_t *q = &r[i];
		int32_t a[18] = {q->id, q->cnt, q->rid, q->score, q->qs, q->qe, q->rs, q->re, q->parent, q->subsc, q->as,
						 q->mlen, q->blen, q->n_sub, q->score0, (int32_t)q->mapq, (int32_t)q->rev, (int32_t)q->hash};
		v.insert(v.end(), a, a + 18);
	}
	dump_rec(T_REGS, rd, c, n, 18 * 4, v.data());
}

// Same call order as ri_map_frag (rmap.cpp:210-387); returns like the original via reg
static void frag_dump(const ri_idx_t *ri, uint32_t s_len, const float *sig, ri_reg1_t *reg, ri_tbuf_t *b, const ri_mapopt_t *opt,
					  const char *qname, double *mean_sum, double *std_dev_sum, uint32_t *n_events_sum, uint32_t rd, uint32_t c)
{
	uint32_t n_events = 0;
	float *events = detect_events(b->km, s_len, sig, opt->window_length1, opt->window_length2, opt->threshold1, opt->threshold2,
								  opt->peak_height, mean_sum, std_dev_sum, n_events_sum, &n_events);
	dump_rec(T_EVENTS, rd, c, n_events, 4, events);
	if (n_events < opt->min_events) { if (events) ri_kfree(b->km, events); return; }
	mm128_v riv = {0, 0, 0};
	ri_sketch(b->km, events, 0, 0, n_events, ri->diff, ri->w, ri->e, ri->n, ri->q, ri->k, ri->fine_min, ri->fine_max, ri->fine_range, &riv, 0);
	if (events) ri_kfree(b->km, events);
	dump_rec(T_SEEDS, rd, c, riv.n, 16, riv.a);
	int rep_len; int64_t n_seed_pos; mm128_t *seed_hits; uint64_t *u; uint32_t hash;
	seed_hits = collect_seed_hits(b->km, (opt->flag & RI_M_ALL_CHAINS) ? 1 : 0, opt->mid_occ, opt->max_max_occ, opt->occ_dist, ri, qname, reg, &riv, n_events, &n_seed_pos, &rep_len);
	if (riv.a) ri_kfree(b->km, riv.a);
	dump_rec(T_ANCHORS, rd, c, (uint32_t)n_seed_pos, 16, seed_hits);
	float chn_pen_gap = opt->chain_gap_scale * 0.01 * (ri->e + ri->k - 1), chn_pen_skip = opt->chain_skip_scale * 0.01 * (ri->e + ri->k - 1);
	seed_hits = mg_lchain_dp(opt->max_target_gap_length, opt->max_query_gap_length, opt->bw, opt->max_num_skips, opt->max_chain_iter, opt->min_num_anchors,
							 opt->min_chaining_score, chn_pen_gap, chn_pen_skip, &n_seed_pos, seed_hits, &(reg->prev_anchors), &(reg->n_cregs), &u, b->km);
	reg->n_prev_anchors = 0;
	if (n_seed_pos > 0) reg->n_prev_anchors = n_seed_pos;
	else if (reg->prev_anchors) { ri_kfree(b->km, reg->prev_anchors); reg->prev_anchors = NULL; }
	dump_rec(T_CHAIN_U, rd, c, reg->n_cregs, 8, u);
	dump_rec(T_CHAIN_A, rd, c, (uint32_t)n_seed_pos, 16, seed_hits);
	hash = 0;
	hash ^= __ac_Wang_hash(reg->offset + n_events) + __ac_Wang_hash(11);
	hash = __ac_Wang_hash(hash);
	reg->creg = mm_gen_regs(b->km, hash, reg->offset + n_events, reg->n_cregs, u, seed_hits);
	mm_set_parent(b->km, opt->mask_level, opt->mask_len, reg->n_cregs, reg->creg, opt->flag & RI_M_HARD_MLEVEL, opt->alt_drop);
	if (!(opt->flag & RI_M_ALL_CHAINS))
		mm_select_sub(b->km, opt->pri_ratio, opt->best_n, 1, opt->max_target_gap_length * 0.8, &(reg->n_cregs), reg->creg);
	mm_set_mapq(b->km, reg->n_cregs, reg->creg, opt->min_chaining_score, rep_len, 0);
	int32_t sc[3] = {rep_len, (int32_t)n_events, (int32_t)reg->offset};
	dump_rec(T_SCALARS, rd, c, 3, 4, sc);
	dump_regs(rd, c, reg->n_cregs, reg->creg);
	if (seed_hits) ri_kfree(b->km, seed_hits);
	if (u) ri_kfree(b->km, u);
	reg->offset += n_events;
}

static int cmd_dump(int argc, char **argv)
{
	if (argc < 6) return 2;
	ri_idxopt_t ipt; ri_mapopt_t opt;
	if (set_presets(argv[2], &ipt, &opt) < 0) return 1;
	ri_idx_t *ri = load_index(argv[3], &ipt);
	if (!ri) return 1;
	ri_mapopt_update(&opt, ri);
	std::vector<rhr_read> reads;
	if (!load_rhr(argv[4], reads)) return 1;
	g_dump = fopen(argv[5], "wb");
	if (!g_dump) return 1;
	ri_tbuf_t *b = ri_tbuf_init();
	for (size_t r = 0; r < reads.size(); ++r) {
		ri_sig_t *sig = to_sig(reads[r], r);
		dump_rec(T_SIG, r, 0, sig->l_sig, 4, sig->sig);
		ri_reg1_t reg0; memset(&reg0, 0, sizeof(reg0));
		// chunk loop of map_worker_for (rmap.cpp:402-501), all chunks dumped, decision logic not needed here
		uint32_t qlen = sig->l_sig;
		uint32_t l_chunk = (opt.chunk_size > qlen || (opt.flag & RI_M_NO_ADAPTIVE)) ? qlen : opt.chunk_size;
		uint32_t max_chunk = (opt.flag & RI_M_NO_ADAPTIVE) ? 1 : opt.max_num_chunk;
		uint32_t s_qs, s_qe, c_count;
		double mean_sum = 0, std_dev_sum = 0; uint32_t n_events_sum = 0;
		for (s_qs = c_count = 0; s_qs < qlen && c_count < max_chunk; s_qs += l_chunk, ++c_count) {
			s_qe = s_qs + l_chunk; if (s_qe > qlen) s_qe = qlen;
			if (reg0.creg) { free(reg0.creg); reg0.creg = NULL; reg0.n_cregs = 0; }
			frag_dump(ri, s_qe - s_qs, &sig->sig[s_qs], &reg0, b, &opt, sig->name, &mean_sum, &std_dev_sum, &n_events_sum, r, c_count);
		}
		if (reg0.creg) free(reg0.creg);
		free(sig->sig); free(sig->name); free(sig);
		ri_km_destroy(b->km); b->km = ri_km_init();
	}
	fclose(g_dump);
	ri_tbuf_destroy(b);
	ri_idx_destroy(ri);
	return 0;
}

// The bucket hash tables are opaque (void*) outside rindex.c; to walk them we need the same khash
// instantiation rindex.c:17-19 makes (a macro invocation, layout-identical by construction).
#define idx_hash(a) ((a)>>1)
#define idx_eq(a, b) ((a)>>1 == (b)>>1)
KHASH_INIT(idx, uint64_t, uint64_t, 1, idx_hash, idx_eq)
typedef khash_t(idx) idxhash_t;

static int cmd_idxdump(int argc, char **argv)
{
	if (argc < 4) return 2;
	ri_idxopt_t ipt; ri_mapopt_t opt;
	set_presets("default", &ipt, &opt);
	ri_idx_t *ri = load_index(argv[2], &ipt);
	if (!ri) return 1;
	ri_mapopt_update(&opt, ri);
	struct ent { uint64_t hash; uint32_t n; const uint64_t *p; };
	std::vector<ent> v;
	for (int i = 0; i < 1 << ri->b; ++i) {
		idxhash_t *h = (idxhash_t*)ri->B[i].h;
		if (!h) continue;
		for (khint_t k = 0; k < kh_end(h); ++k) {
			if (!kh_exist(h, k)) continue;
			ent e; e.hash = (kh_key(h, k) >> 1) << ri->b | (uint64_t)i;
			int n; e.p = ri_idx_get(ri, e.hash, &n); e.n = n;
			v.push_back(e);
		}
	}
	std::sort(v.begin(), v.end(), [](const ent &a, const ent &b) { return a.hash < b.hash; });
	FILE *fp = fopen(argv[3], "wb");
	uint64_t nk = v.size(); int32_t hdr[9] = {ri->w, ri->e, ri->n, ri->q, ri->k, (int32_t)ri->n_seq, ri->flag, opt.mid_occ, 0};
	fwrite(hdr, 4, 9, fp); fwrite(&nk, 8, 1, fp);
	for (size_t i = 0; i < v.size(); ++i) { fwrite(&v[i].hash, 8, 1, fp); fwrite(&v[i].n, 4, 1, fp); fwrite(v[i].p, 8, v[i].n, fp); }
	fclose(fp);
	ri_idx_destroy(ri);
	return 0;
}

// RH_REF_INTERLEAVE=1: every page this process touches from here on is interleaved over all NUMA nodes (what `numactl --interleave=all`
// does; the image has no numactl).  The index loader is single-threaded: without it the whole table is first-touched on the loader's
// node and every worker thread of the other socket reads it remotely.  Returns the number of nodes found (0 = policy not set).
#include <sys/syscall.h>
#include <dirent.h>
static int numa_interleave_all()
{
	int n_nodes = 0;
	if (DIR *d = opendir("/sys/devices/system/node")) {
		while (struct dirent *e = readdir(d)) if (!strncmp(e->d_name, "node", 4) && e->d_name[4] >= '0' && e->d_name[4] <= '9') { const int k = atoi(e->d_name + 4) + 1; if (k > n_nodes) n_nodes = k; }
		closedir(d);
	}
	if (n_nodes < 2 || n_nodes > 1024) return n_nodes < 2 ? n_nodes : 0;
	unsigned long mask[16] = {0};
	for (int i = 0; i < n_nodes; ++i) mask[i / (8 * sizeof(long))] |= 1ul << (i % (8 * sizeof(long)));
	if (syscall(SYS_set_mempolicy, 3 /* MPOL_INTERLEAVE */, mask, (unsigned long)(sizeof(mask) * 8)) != 0) { perror("[ref_harness] set_mempolicy"); return 0; }
	return n_nodes;
}

int main(int argc, char **argv)
{
	ri_verbose = 1; ri_realtime0 = ri_realtime();
	if (const char *e = getenv("RH_REF_INTERLEAVE")) if (atoi(e)) fprintf(stderr, "[ref_harness] memory interleaved over %d NUMA node(s)\n", numa_interleave_all());
	int ret = 2;
	if (argc >= 2) {
		if (!strcmp(argv[1], "index")) ret = cmd_index(argc, argv);
		else if (!strcmp(argv[1], "sigindex")) ret = cmd_sigindex(argc, argv);
		else if (!strcmp(argv[1], "map")) ret = cmd_map(argc, argv);
		else if (!strcmp(argv[1], "dump")) ret = cmd_dump(argc, argv);
		else if (!strcmp(argv[1], "idxdump")) ret = cmd_idxdump(argc, argv);
	}
	if (ret == 2) fprintf(stderr, "usage: ref_harness index|map|dump|idxdump ... (see header of oracle/ref_harness.cpp)\n");
	return ret;
}
