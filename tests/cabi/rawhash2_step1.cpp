// TEST INFRASTRUCTURE: a compiled C++ consumer of the C ABI - the binding of INTEGRATION.md section 1 as a stand-alone
// program.  What a RawHash2 host does around step 1 of its pipeline (rmap.cpp:695-703) with librawhash_amd.so in place of
// kt_for(map_worker_for): load the .ind, take the preset's options, keep two mini-batches in flight (kt_pipeline with
// pl_threads = 2, rmap.cpp:831,852) and print PAF in read order (step 2, rmap.cpp:736-783).
//
//   rawhash2_step1 <preset> <ref.ind> <reads.rhr> [reads per mini-batch]   > out.paf
//
// Built by tests/test_cabi.py with g++ (no HIP headers: only include/rawhash_amd.h).
#include "rawhash_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int fail(const char *what) { fprintf(stderr, "%s: %s\n", what, rh_last_error()); return 1; }

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s <preset> <ref.ind> <reads.rhr> [batch]\n", argv[0]); return 2; }
	setenv("GPU_MAX_HW_QUEUES", "8", 0);                          // two mini-batches in flight = four streams: the HIP runtime's default of 4 hardware queues costs a fifth of the rate; before the first HIP call
	rh_idxopt_t io; rh_mapopt_t mo;
	if (rh_set_preset(nullptr, &io, &mo) || (strcmp(argv[1], "default") && rh_set_preset(argv[1], &io, &mo))) return fail("preset");
	rh_index *idx = rh_index_load(argv[2]);                       // ri_idx_load (rindex.c:650)
	if (!idx) return fail("index");
	rh_mapopt_update(&mo, idx);                                   // ri_mapopt_update (rindex.c:1041)
	rh_ctx *ctx = nullptr;
	if (rh_ctx_create(&ctx, 0) || rh_index_upload(ctx, idx)) return fail("device");
	rh_reads *reads = rh_reads_load(argv[3]);                     // step 0 (our container; see rh_reads_load_blow5 for BLOW5)
	if (!reads) return fail("reads");
	rh_read_batch_t all;
	rh_reads_batch(reads, &all);
	const uint32_t n = all.n_reads, per = argc > 4 ? (uint32_t)atoi(argv[4]) : (n + 1) / 2;
	std::vector<rh_map_record_t> rec(n);
	// mini-batches: views into the read set (absolute CSR offsets are kept, as rh_map_batch's slices do)
	struct InFlight { rh_read_batch_t b; rh_ticket_t t; uint32_t first; bool on; } fl[RH_MAX_IN_FLIGHT] = {};
	uint32_t next = 0, done = 0;
	int cur = 0;
	while (done < n) {
		InFlight &f = fl[cur];
		if (f.on) {                                               // oldest batch: wait, rebase read_idx (records are per batch)
			uint64_t got = 0;
			if (rh_map_wait(ctx, f.t, &got)) return fail("rh_map_wait");
			for (uint64_t k = 0; k < got; ++k) rec[f.first + k].read_idx += f.first;
			done += (uint32_t)got; f.on = false;
		}
		if (next < n) {
			const uint32_t m = n - next < per ? n - next : per;
			f.b = all; f.b.n_reads = m; f.b.offsets = all.offsets + next;
			f.b.cal_offset = all.cal_offset + next; f.b.cal_scale = all.cal_scale + next;
			if (all.n_filtered) f.b.n_filtered = all.n_filtered + next;   // (the reader's l_sig per read: the device then fetches only the signal the rounds consume)
			f.first = next;
			if (rh_map_submit(ctx, &mo, &f.b, rec.data() + next, m, &f.t)) return fail("rh_map_submit");
			f.on = true; next += m;
		}
		cur = (cur + 1) % RH_MAX_IN_FLIGHT;
	}
	char line[4096];
	for (uint32_t k = 0; k < n; ++k) {                             // step 2: PAF in read order
		const int len = rh_paf_format(idx, &rec[k], rh_reads_name(reads, rec[k].read_idx), 0.0, line, sizeof(line));
		if (len < 0) return fail("rh_paf_format");
		if (len) fwrite(line, 1, (size_t)len, stdout);                // (the line ends with its newline, as fprintf'd by rmap.cpp:751)
	}
	rh_reads_destroy(reads); rh_ctx_destroy(ctx); rh_index_destroy(idx);
	return 0;
}
