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
#include "rh_synth_core.h"
#include <cmath>
#include <cstdlib>
#include <thread>

static const int SY_K = RH_SY_K;

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

static inline uint32_t genome_base(const rh_synth_cfg_t *c, uint32_t chrom, uint32_t pos) { return rh_sy_genome_base(c->genome_seed, chrom, pos); }

static double model_level(const rh_synth_cfg_t *c, uint32_t kmer)
{
	uint64_t h = rh_sy_rand3(c->model_seed, kmer, 0);
	int64_t s = (int64_t)(h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + ((h >> 48) & 0xFFFF) - 131070;
	return 90.0 + 12.0 * (double)s / 37837.2;
}

extern "C" int rh_synth_write_model_k(const rh_synth_cfg_t *c, const char *path, int k)
{
	if (k < 4 || k > 12) { rh_set_error("synthetic pore model: k = %d (4 .. 12)", k); return -1; }
	FILE *fp = fopen(path, "w");
	if (!fp) { rh_set_error("cannot write %s", path); return -1; }
	fprintf(fp, "kmer\tlevel_mean\tlevel_stdv\n");
	for (uint32_t i = 0; i < (1u << (2 * k)); ++i) {
		char km[16];
		for (int j = 0; j < k; ++j) km[j] = "ACGT"[(i >> (2 * (k - 1 - j))) & 3];
		km[k] = 0;
		fprintf(fp, "%s\t%.4f\t1.5000\n", km, model_level(c, i));
	}
	fclose(fp);
	return 0;
}
extern "C" int rh_synth_write_model(const rh_synth_cfg_t *c, const char *path) { return rh_synth_write_model_k(c, path, SY_K); }

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

extern "C" int rh_synth_genome(const rh_synth_cfg_t *c, uint32_t chrom, char *out, int n_threads)
{
	if (chrom >= c->n_chrom) { rh_set_error("chromosome %u out of range", chrom); return -1; }
	if (n_threads < 1) n_threads = 1;
	std::vector<std::thread> th;
	const uint64_t per = ((uint64_t)c->chrom_len + n_threads - 1) / n_threads;
	for (int t = 0; t < n_threads; ++t)
		th.emplace_back([=]() {
			const uint64_t b = (uint64_t)t * per, e = b + per < c->chrom_len ? b + per : c->chrom_len;
			for (uint64_t p = b; p < e; ++p) out[p] = "ACGT"[genome_base(c, chrom, (uint32_t)p)];
		});
	for (auto &t : th) t.join();
	return 0;
}

static inline uint32_t read_span(const rh_synth_cfg_t *c) { return rh_sy_span(c->n_samples); }

extern "C" int rh_synth_origin(const rh_synth_cfg_t *c, uint64_t idx, uint32_t *chrom, uint32_t *pos, uint32_t *strand, uint32_t *junk)
{
	uint32_t span = read_span(c);
	if (c->chrom_len <= span + 1 || c->n_chrom == 0) { rh_set_error("chrom_len too small for n_samples"); return -1; }
	const rh_sy_origin o = rh_sy_read_origin(*c, idx);
	if (junk) *junk = o.junk;
	if (chrom) *chrom = o.chrom;
	if (strand) *strand = o.strand;
	if (pos) *pos = o.pos;
	return 0;
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
	int k = 0;
	while (k <= 12 && ((size_t)1 << (2 * k)) < lev.size()) ++k;
	if (k < 4 || k > 12 || ((size_t)1 << (2 * k)) != lev.size()) { rh_set_error("model %s: %zu k-mers is not 4^k for a k of 4 .. 12", path, lev.size()); return -1; }
	return 0;
}

static void synth_one(const rh_synth_cfg_t *c, const int32_t *level16, uint32_t k, uint64_t idx, int16_t *out, char *name64)
{
	if (name64) {
		const rh_sy_origin o = rh_sy_read_origin(*c, idx);
		if (o.junk) snprintf(name64, 64, "r%llu_junk", (unsigned long long)idx);
		else snprintf(name64, 64, "r%llu_chr%u_%u_%c", (unsigned long long)idx, o.chrom + 1, o.pos, o.strand ? '-' : '+');
	}
	rh_sy_generate(*c, level16, k, idx, out);
}

uint32_t rh_synth_model_k(size_t n_levels) { uint32_t k = 0; while (((size_t)1 << (2 * k)) < n_levels) ++k; return k; }   // (a table of 4^k levels)

int rh_synth_level_table(const rh_synth_cfg_t *c, const char *model_path, std::vector<int32_t> &level16)
{
	std::vector<float> lev;
	if (load_model_levels(model_path, lev) < 0) return -1;
	if (c->chrom_len <= read_span(c) + 1 || c->n_chrom == 0) { rh_set_error("chrom_len too small for n_samples"); return -1; }
	level16.resize(lev.size());
	const double scale = c->range / c->digitisation;
	for (size_t i = 0; i < lev.size(); ++i) level16[i] = (int32_t)floor(((double)lev[i] / scale - c->offset) * 16.0 + 0.5);
	return 0;
}

extern "C" int rh_synth_reads(const rh_synth_cfg_t *c, const char *model_path, uint64_t first, uint32_t n,
                              int16_t *samples, char *names64, int n_threads)
{
	std::vector<int32_t> level16;
	if (rh_synth_level_table(c, model_path, level16) < 0) return -1;
	if (n_threads < 1) n_threads = 1;
	if ((uint32_t)n_threads > n) n_threads = n ? n : 1;
	const uint32_t k = rh_synth_model_k(level16.size());
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t)
		th.emplace_back([=, &level16]() {
			for (uint32_t i = t; i < n; i += n_threads)
				synth_one(c, level16.data(), k, first + i, samples + (size_t)i * c->n_samples, names64 ? names64 + (size_t)i * 64 : 0);
		});
	for (auto &t : th) t.join();
	return 0;
}
