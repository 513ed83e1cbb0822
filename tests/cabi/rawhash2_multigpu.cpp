// TEST INFRASTRUCTURE: a compiled C++ consumer of the C ABI - ONE process driving every visible GPU: what a
// single-process `rawhash2` (the reference is one process, main.cpp:588) does with a node of MI355X: the index is loaded once, uploaded
// to GPU 0 and replicated with rh_index_bcast (one RCCL broadcast over xGMI when the contexts sit on distinct devices; the peer-copy
// tree otherwise), the reads of the mini-batch are sharded in contiguous blocks over one host thread + context per GPU (no
// collective on the data path), and the PAF is printed in read order (step 2, rmap.cpp:736-783).
//
//   rawhash2_multigpu <preset> <ref.ind> <reads.rhr> [contexts (0 = one per visible GPU)]   > out.paf
//
// With more contexts than GPUs the extra ones share devices round-robin (lets a one-GPU box run the sharded flow).
// Built by tests/test_cabi.py with g++ (no HIP headers: only include/rawhash_amd.h).
#include "rawhash_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static int fail(const char *what) { fprintf(stderr, "%s: %s\n", what, rh_last_error()); return 1; }

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s <preset> <ref.ind> <reads.rhr> [contexts]\n", argv[0]); return 2; }
	rh_idxopt_t io; rh_mapopt_t mo;
	if (rh_set_preset(nullptr, &io, &mo) || (strcmp(argv[1], "default") && rh_set_preset(argv[1], &io, &mo))) return fail("preset");
	rh_index *idx = rh_index_load(argv[2]);
	if (!idx) return fail("index");
	rh_mapopt_update(&mo, idx);
	const int n_dev = rh_device_count();
	if (n_dev < 1) return fail("no GPU");
	int n_ctx = argc > 4 ? atoi(argv[4]) : 0;
	if (n_ctx <= 0) n_ctx = n_dev;
	std::vector<rh_ctx*> ctx((size_t)n_ctx, nullptr);
	for (int g = 0; g < n_ctx; ++g) if (rh_ctx_create(&ctx[g], g % n_dev)) return fail("rh_ctx_create");
	if (rh_index_upload(ctx[0], idx)) return fail("rh_index_upload");
	if (n_ctx > 1 && rh_index_bcast(ctx.data(), n_ctx)) return fail("rh_index_bcast");   // the one collective of the design
	fprintf(stderr, "%d context(s) on %d GPU(s); index replicated through %s\n", n_ctx, n_dev,
	        n_ctx == 1 ? "nothing (one context)" : rh_index_bcast_path() == 1 ? "RCCL (ncclBroadcast)" : "peer copies");
	rh_reads *reads = rh_reads_load(argv[3]);
	if (!reads) return fail("reads");
	rh_read_batch_t all;
	rh_reads_batch(reads, &all);
	const uint32_t n = all.n_reads;
	std::vector<rh_map_record_t> rec(n);
	std::vector<int> rc((size_t)n_ctx, 0);
	std::vector<std::string> err((size_t)n_ctx);
	std::vector<std::thread> th;
	for (int g = 0; g < n_ctx; ++g)
		th.emplace_back([&, g]() {                                  // shard g: reads [lo, hi), a view into the read set (absolute CSR offsets kept)
			const uint32_t lo = (uint32_t)((uint64_t)n * g / n_ctx), hi = (uint32_t)((uint64_t)n * (g + 1) / n_ctx);
			if (hi == lo) return;
			rh_read_batch_t b = all;
			b.n_reads = hi - lo; b.offsets = all.offsets + lo; b.cal_offset = all.cal_offset + lo; b.cal_scale = all.cal_scale + lo;
			if (all.n_filtered) b.n_filtered = all.n_filtered + lo;
			uint64_t got = 0;
			rc[g] = rh_map_batch(ctx[g], &mo, &b, rec.data() + lo, hi - lo, &got);
			if (rc[g]) err[g] = rh_last_error();                     // (the error text is per thread)
			else for (uint64_t k = 0; k < got; ++k) rec[lo + k].read_idx += lo;
		});
	for (auto &t : th) t.join();
	for (int g = 0; g < n_ctx; ++g) if (rc[g]) { fprintf(stderr, "shard %d: %s\n", g, err[g].c_str()); return 1; }
	char line[4096];
	for (uint32_t k = 0; k < n; ++k) {
		const int len = rh_paf_format(idx, &rec[k], rh_reads_name(reads, rec[k].read_idx), 0.0, line, sizeof(line));
		if (len < 0) return fail("rh_paf_format");
		if (len) fwrite(line, 1, (size_t)len, stdout);
	}
	rh_reads_destroy(reads);
	for (rh_ctx *c : ctx) rh_ctx_destroy(c);
	rh_index_destroy(idx);
	return 0;
}
