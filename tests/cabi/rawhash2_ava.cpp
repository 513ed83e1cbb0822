// TEST INFRASTRUCTURE: the Rawsamble binding of INTEGRATION.md section 3 as a stand-alone C++ program against the C ABI only.
//
//   rawhash2_ava <preset> <pore.model> <reads.rhr> <out.ind>   > overlaps.paf
//
// = `rawhash2 -x <preset> -p <pore.model> -d <out.ind> <reads>` (ri_idx_siggen rindex.c:927 + ri_idx_dump :545) followed by
//   `rawhash2 -x <preset> <out.ind> <reads>` (all-vs-all mapping; step 2 prints reg->maps[0 .. n_maps), rmap.cpp:740-783).
// Built by tests/test_cabi.py with g++ (no HIP headers: only include/rawhash_amd.h).
#include "rawhash_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int fail(const char *what) { fprintf(stderr, "%s: %s\n", what, rh_last_error()); return 1; }

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: %s <preset> <pore.model> <reads.rhr> <out.ind>\n", argv[0]); return 2; }
	rh_idxopt_t io; rh_mapopt_t mo;
	if (rh_set_preset(nullptr, &io, &mo) || rh_set_preset(argv[1], &io, &mo)) return fail("preset");
	rh_reads *reads = rh_reads_load(argv[3]);
	if (!reads) return fail("reads");
	rh_read_batch_t batch;
	rh_reads_batch(reads, &batch);
	const uint32_t n = batch.n_reads;
	std::vector<const char*> names(n);
	for (uint32_t r = 0; r < n; ++r) names[r] = rh_reads_name(reads, r);
	rh_ctx *ctx = nullptr;
	if (rh_ctx_create(&ctx, 0)) return fail("device");
	// -d: every read becomes a target; the index stays resident on ctx, the file is what ri_idx_dump writes
	rh_index *idx = rh_index_build_signals_device(ctx, &batch, names.data(), argv[2], &io, &mo);
	if (!idx) return fail("rh_index_build_signals_device");
	if (rh_index_download(ctx, idx, 4) || rh_index_write(idx, argv[4])) return fail("index file");
	// mapping: the strcmp of rmap.cpp:86 on name ranks, one round over whole reads, every reported chain a record
	rh_mapopt_update(&mo, idx);
	std::vector<uint32_t> qrank(n ? n : 1);
	if (rh_index_name_ranks(idx, names.data(), n, qrank.data(), nullptr)) return fail("rh_index_name_ranks");
	batch.name_rank = qrank.data();
	std::vector<rh_map_record_t> rec((size_t)64 * n + 1024);
	std::vector<uint64_t> off((size_t)n + 1);
	uint64_t n_rec = 0;
	if (rh_map_batch_multi(ctx, &mo, &batch, rec.data(), rec.size(), off.data(), &n_rec)) return fail("rh_map_batch_multi");
	char line[4096];
	for (uint32_t r = 0; r < n; ++r)
		for (uint64_t m = off[r]; m < off[r + 1]; ++m) {
			const int len = rh_paf_format(idx, &rec[m], names[r], 0.0, line, sizeof(line));
			if (len < 0) return fail("rh_paf_format");
			if (len) fwrite(line, 1, (size_t)len, stdout);
		}
	rh_reads_destroy(reads); rh_ctx_destroy(ctx); rh_index_destroy(idx);
	return 0;
}
