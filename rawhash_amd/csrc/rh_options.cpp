// Option defaults and presets of the mapping path.
// Values follow ri_idxopt_init / ri_mapopt_init (reference roptions.c:4-32, :34-138) and the `-x` preset table
// ri_set_opt (main.cpp:111-210); only the fields that reach the path are kept (rh_idxopt_t / rh_mapopt_t).
#include "rh_common.h"
#include <climits>

extern "C" void rh_idxopt_init(rh_idxopt_t *io)
{
	*io = rh_idxopt_t{};
	io->b = 14; io->w = 0; io->e = 8; io->n = 0; io->q = 4; io->k = 6; io->lev_col = 1;
	io->diff = 0.35f;
	io->fine_min = -2.0f; io->fine_max = 2.0f; io->fine_range = 0.4;   // double literal narrowed, as in the reference
}

extern "C" void rh_mapopt_init(rh_mapopt_t *mo)
{
	*mo = rh_mapopt_t{};
	mo->bp_per_sec = 450; mo->sample_rate = 4000; mo->chunk_size = 4000;
	mo->sample_per_base = (float)mo->sample_rate / mo->bp_per_sec;
	// seeding
	mo->mid_occ_frac = 1e-2f; mo->min_mid_occ = 50; mo->max_mid_occ = 500000;
	mo->max_max_occ = 32767; mo->occ_dist = 500;
	// chaining
	mo->bw = 500; mo->bw_long = 0;
	mo->max_target_gap_length = 2500; mo->max_query_gap_length = 2500;
	mo->max_chain_iter = 200; mo->max_num_skips = 5; mo->min_num_anchors = 2;
	mo->min_chaining_score = 15; mo->min_chaining_score2 = 0;
	mo->chain_gap_scale = 0.8f; mo->chain_skip_scale = 0.0f;
	mo->rmq_inner_dist = 1000; mo->rmq_size_cap = 100000;
	mo->dtw_border_constraint = RH_DTW_BORDER_SPARSE; mo->dtw_fill_method = RH_DTW_FILL_BANDED;
	mo->dtw_band_radius_frac = 0.10f; mo->dtw_match_bonus = 0.4f; mo->dtw_min_score = 20.0f; mo->w_bestma = 0.2f;
	// regions
	mo->mask_level = 0.5f; mo->mask_len = INT_MAX; mo->pri_ratio = 0.3f; mo->best_n = 0; mo->alt_drop = 0.15f;
	// decision
	mo->w_bestq = 0.35f; mo->w_bestmq = 0.05f; mo->w_bestmc = 0.6f; mo->w_threshold = 0.45f;
	mo->min_events = 50; mo->max_num_chunk = 10; mo->min_mapq = 2;
	// segmentation
	mo->window_length1 = 3; mo->window_length2 = 9; mo->threshold1 = 4.0f; mo->threshold2 = 3.5f; mo->peak_height = 0.4f;
}

namespace {
struct AvaPreset { const char *name; int w, min_sc, min_sc2, min_anchors, min_mapq, bw; };
const AvaPreset kAva[] = {
	{"ava-viral", 0, 20, 30, 5, 5, 1000},
	{"ava", 3, 40, 75, 5, 5, 5000},
	{"ava-sensitive", 0, 75, 100, 5, 5, 1000},
	{"ava-large", 5, 20, 50, 2, 2, 5000},
};
}

extern "C" int rh_set_preset(const char *preset, rh_idxopt_t *io, rh_mapopt_t *mo)
{
	if (!preset) { rh_idxopt_init(io); rh_mapopt_init(mo); return 0; }
	const std::string p(preset);
	if (p == "sensitive" || p == "sequence-until") return 0;   // == defaults
	if (p == "viral") {
		io->e = 6;
		mo->bw = 100; mo->max_target_gap_length = 500; mo->max_query_gap_length = 500;
		mo->max_num_chunk = 5; mo->min_chaining_score = 10; mo->chain_gap_scale = 1.2f; mo->chain_skip_scale = 0.3f;
		return 0;
	}
	if (p == "fast" || p == "faster") {
		io->fine_range = 0.6;
		mo->min_mapq = 5; mo->min_chaining_score = 10; mo->chain_gap_scale = 0.6f;
		if (p == "faster") { io->e = 11; io->w = 3; mo->max_num_chunk = 5; }
		return 0;
	}
	for (const AvaPreset &a : kAva) {
		if (p != a.name) continue;
		if (p == "ava-viral") { io->e = 6; mo->chain_gap_scale = 1.2f; mo->chain_skip_scale = 0.3f; }
		if (p == "ava-large") { io->fine_range = 0.6; mo->chain_gap_scale = 0.6f; }
		io->w = a.w; io->diff = 0.45f;
		mo->min_chaining_score = a.min_sc; mo->min_chaining_score2 = a.min_sc2;
		mo->min_num_anchors = a.min_anchors; mo->min_mapq = a.min_mapq; mo->bw = a.bw;
		mo->max_target_gap_length = 2500; mo->max_query_gap_length = 2500;
		io->flag |= RH_I_SIG_TARGET;
		mo->flag |= RH_M_ALL_CHAINS | RH_M_NO_ADAPTIVE;
		mo->pri_ratio = 0.0f;
		return 0;
	}
	rh_set_error("unknown preset '%s'", preset);
	return -1;
}
