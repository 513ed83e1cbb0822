// Synthetic workload generator (bench/test support, not on the mapping path): SURVEY §8d.
//
// Everything after the pore-model text file is integer arithmetic on counter-based hashes, so the same
// (cfg, model file) gives the same int16 samples on any host, with any thread count, in any order.
//   genome : i.i.d. uniform ACGT, n_chrom sequences of chrom_len bases
//   model  : 4^6 levels ~ N(90, 12) pA (sum of four 16-bit uniforms), written as an ONT-style k-mer table
//            (header line starting with "kmer", level in column 1: what load_pore rutils.c:133-178 parses)
//   reads  : uniform start, 50/50 strand, per-base dwell ~ Gamma(2) with mean ~8.9 samples (4000 Hz / 450 bp/s),
//            additive noise ~ N(0, 1.5 pA), digitised with (digitisation, range, offset) like an R9.4 MinION.
#include "rh_common.h"
#include <cmath>
#include <cstdlib>
#include <thread>

static const int SY_K = 6;

extern "C" void rh_synth_cfg_init(rh_synth_cfg_t *c)
{
	memset(c, 0, sizeof(*c));
	c->model_seed = 1; c->genome_seed = 2; c->read_seed = 3;
	c->n_chrom = 1; c->chrom_len = 4600000;
	c->n_samples = 40000;
	c->junk_per_1024 = 0;
	c->noise_q24 = 0;
	c->digitisation = 8192.0; c->range = 1402.882; c->offset = 6.0;
}

static inline uint32_t genome_base(const rh_synth_cfg_t *c, uint32_t chrom, uint32_t pos)
{
	return (uint32_t)(rh_rand3(c->genome_seed, chrom, pos >> 5) >> ((pos & 31) * 2)) & 3;
}

static double model_level(const rh_synth_cfg_t *c, uint32_t kmer)
{
	uint64_t h = rh_rand3(c->model_seed, kmer, 0);
	int64_t s = (int64_t)(h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + ((h >> 48) & 0xFFFF) - 131070;
	return 90.0 + 12.0 * (double)s / 37837.2;
}

extern "C" int rh_synth_write_model(const rh_synth_cfg_t *c, const char *path)
{
	FILE *fp = fopen(path, "w");
	if (!fp) { rh_set_error("cannot write %s", path); return -1; }
	fprintf(fp, "kmer\tlevel_mean\tlevel_stdv\n");
	for (uint32_t i = 0; i < (1u << (2 * SY_K)); ++i) {
		char km[SY_K + 1];
		for (int j = 0; j < SY_K; ++j) km[j] = "ACGT"[(i >> (2 * (SY_K - 1 - j))) & 3];
		km[SY_K] = 0;
		fprintf(fp, "%s\t%.4f\t1.5000\n", km, model_level(c, i));
	}
	fclose(fp);
	return 0;
}

extern "C" int rh_synth_write_fasta(const rh_synth_cfg_t *c, const char *path)
{
	FILE *fp = fopen(path, "w");
	if (!fp) { rh_set_error("cannot write %s", path); return -1; }
	std::vector<char> line(81);
	for (uint32_t ch = 0; ch < c->n_chrom; ++ch) {
		fprintf(fp, ">chr%u\n", ch + 1);
		for (uint32_t p = 0; p < c->chrom_len; p += 80) {
			uint32_t n = c->chrom_len - p < 80 ? c->chrom_len - p : 80;
			for (uint32_t j = 0; j < n; ++j) line[j] = "ACGT"[genome_base(c, ch, p + j)];
			line[n] = '\n';
			fwrite(line.data(), 1, n + 1, fp);
		}
	}
	fclose(fp);
	return 0;
}

static inline uint32_t read_span(const rh_synth_cfg_t *c) { return c->n_samples / 4 + 16; }

extern "C" int rh_synth_origin(const rh_synth_cfg_t *c, uint64_t idx, uint32_t *chrom, uint32_t *pos, uint32_t *strand, uint32_t *junk)
{
	uint32_t span = read_span(c);
	if (c->chrom_len <= span + 1 || c->n_chrom == 0) { rh_set_error("chrom_len too small for n_samples"); return -1; }
	uint64_t h = rh_rand3(c->read_seed, idx, 0);
	if (junk) *junk = (h & 1023) < c->junk_per_1024;
	if (chrom) *chrom = (uint32_t)((h >> 10) % c->n_chrom);
	if (strand) *strand = (uint32_t)(h >> 40) & 1;
	if (pos) *pos = (uint32_t)(rh_rand3(c->read_seed, idx, 1) % (c->chrom_len - span));
	return 0;
}

// -log2(u / 65536) in Q8 for u in [1, 65536] (piecewise-linear mantissa), integer only
static inline uint32_t neg_log2_q8(uint32_t u)
{
	int i = 31 - __builtin_clz(u);
	uint32_t frac = ((u << (16 - i)) & 0xFFFF) >> 8;
	return (16u << 8) - (((uint32_t)i << 8) + frac);
}

static int load_model_levels(const char *path, std::vector<float> &lev)
{
	FILE *fp = fopen(path, "r");
	if (!fp) { rh_set_error("cannot open %s", path); return -1; }
	char line[1024];
	lev.clear();
	while (fgets(line, sizeof(line), fp)) {
		if (!strncmp(line, "kmer", 4)) continue;
		char *t = strchr(line, '\t');
		if (!t) continue;
		lev.push_back(strtof(t + 1, 0));
	}
	fclose(fp);
	if (lev.size() != (1u << (2 * SY_K))) { rh_set_error("model %s: expected %u k-mers, got %zu", path, 1u << (2 * SY_K), lev.size()); return -1; }
	return 0;
}

static void synth_one(const rh_synth_cfg_t *c, const int32_t *level16, uint64_t idx, int16_t *out, char *name64)
{
	uint32_t chrom, pos, strand, junk, span = read_span(c);
	rh_synth_origin(c, idx, &chrom, &pos, &strand, &junk);
	if (name64) {
		if (junk) snprintf(name64, 64, "r%llu_junk", (unsigned long long)idx);
		else snprintf(name64, 64, "r%llu_chr%u_%u_%c", (unsigned long long)idx, chrom + 1, pos, strand ? '-' : '+');
	}
	const uint32_t noise_q24 = c->noise_q24 ? c->noise_q24 : 62152u;
	const uint32_t kmask = (1u << (2 * SY_K)) - 1;
	uint32_t kmer = 0, s = 0;
	for (uint32_t j = 0; j < span && s < c->n_samples; ++j) {
		uint32_t b;
		if (junk) b = (uint32_t)(rh_rand3(c->read_seed ^ 0x6A756E6BULL, idx, j >> 5) >> ((j & 31) * 2)) & 3;
		else if (!strand) b = genome_base(c, chrom, pos + j);
		else b = 3 - genome_base(c, chrom, pos + span - 1 - j);
		kmer = ((kmer << 2) | b) & kmask;
		if (j + 1 < (uint32_t)SY_K) continue;
		uint64_t hd = rh_rand3(c->read_seed + 2, idx, j);
		uint32_t e = neg_log2_q8((uint32_t)(hd & 0xFFFF) + 1) + neg_log2_q8((uint32_t)((hd >> 16) & 0xFFFF) + 1);
		uint32_t dwell = (e * 790u + (1u << 15)) >> 16;
		if (dwell < 1) dwell = 1;
		if (j + 1 == span) dwell = c->n_samples; // ran out of bases (cannot happen in practice): hold the last level
		for (uint32_t d = 0; d < dwell && s < c->n_samples; ++d, ++s) {
			uint64_t hn = rh_rand3(c->read_seed + 3, idx, s);
			int64_t u = (int64_t)(hn & 0xFFFF) + ((hn >> 16) & 0xFFFF) + ((hn >> 32) & 0xFFFF) + ((hn >> 48) & 0xFFFF) - 131070;
			int64_t n16 = (u * (int64_t)noise_q24) >> 24;
			int64_t v = ((int64_t)level16[kmer] + n16 + 8) >> 4;
			if (v > 32767) v = 32767;
			if (v < -32768) v = -32768;
			out[s] = (int16_t)v;
		}
	}
}

extern "C" int rh_synth_reads(const rh_synth_cfg_t *c, const char *model_path, uint64_t first, uint32_t n,
                              int16_t *samples, char *names64, int n_threads)
{
	std::vector<float> lev;
	if (load_model_levels(model_path, lev) < 0) return -1;
	if (c->chrom_len <= read_span(c) + 1) { rh_set_error("chrom_len too small for n_samples"); return -1; }
	std::vector<int32_t> level16(lev.size());
	const double scale = c->range / c->digitisation;
	for (size_t i = 0; i < lev.size(); ++i) level16[i] = (int32_t)floor(((double)lev[i] / scale - c->offset) * 16.0 + 0.5);
	if (n_threads < 1) n_threads = 1;
	if ((uint32_t)n_threads > n) n_threads = n ? n : 1;
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t)
		th.emplace_back([=, &level16]() {
			for (uint32_t i = t; i < n; i += n_threads)
				synth_one(c, level16.data(), first + i, samples + (size_t)i * c->n_samples, names64 ? names64 + (size_t)i * 64 : 0);
		});
	for (auto &t : th) t.join();
	return 0;
}
