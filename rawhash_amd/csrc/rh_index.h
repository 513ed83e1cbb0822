// Host-side index: what ri_idx_t (reference rindex.h:29-60) holds for the mapping path, flattened.
// Instead of 2^14 khash buckets (rindex.c:17-19) the keys live in one array sorted by the 32-bit seed hash, with the
// position lists of multi-occurrence keys concatenated in `pos` -- the layout the device table is built from.
#pragma once
#include "rh_common.h"

struct rh_index_s {
	int32_t w = 0, e = 0, n = 0, q = 0, k = 0, flag = 0;
	float diff = 0, fine_min = 0, fine_max = 0, fine_range = 0;
	std::vector<std::string> names;
	std::vector<uint32_t> lens;
	// pore model blob as stored in the file header (needed to write a loadable .ind back)
	uint32_t n_pore_vals = 0; int16_t pore_k = 0;
	std::vector<float> pore_vals;
	std::vector<unsigned char> pore_inds;      // n_pore_vals x {float, u32, u32}
	// keys sorted by hash
	std::vector<uint32_t> key_hash;            // 32-bit seed hash (= x >> 6 of a sketch entry)
	std::vector<uint32_t> key_n;               // occurrences
	std::vector<uint64_t> key_val;             // n == 1: the position word itself; n > 1: offset into pos[]
	std::vector<uint64_t> pos;                 // position words id<<32 | pos<<1 | strand, ascending per key
	// an index built on the device (rh_index_build_device) keeps its keys in HBM: the host copy holds the header fields, the
	// per-key occupancy histogram (occ_hist[n] = keys with n occurrences, last bin = that many or more; mid_occ calibration)
	// and the totals, until rh_index_download fetches the keys
	std::vector<uint32_t> occ_hist;
	// RH_I_STORE_SIG (--store-sig): the expected signal of every target, one level per k-mer (ri_seq_to_sig rsig.c:13), forward and -
	// unless RH_I_NO_REV_TARGET - reverse strand; what DTW re-scoring (--dtw-evaluate-chains) aligns the read's events with
	std::vector<std::vector<float>> sigF, sigR;
	uint64_t dev_n_keys = 0, dev_n_pos = 0;
};
bool rh_load_model(const char *path, int k, int lev_col, std::vector<float> &vals);
void rh_make_pore_inds(const std::vector<float> &vals, int k, std::vector<unsigned char> &blob);
bool rh_read_fasta(const char *path, std::vector<std::string> &names, std::vector<std::string> &seqs);

// Device table geometry (see rh_device.hip): buckets of RH_TB_SLOTS 16-byte slots = one 128-byte line.
#define RH_TB_SLOTS 8
struct rh_tslot { uint32_t hash; uint32_t n; uint64_t val; };   // n == 0: empty

// Build the bucketed open-addressing table for upload.  Returns log2(#buckets); slots has (1<<log2)*RH_TB_SLOTS entries.
int rh_index_make_table(const rh_index_s &ix, std::vector<rh_tslot> &slots);
