// Synthetic R9.4-like read generator shared by the host (rh_synth.cpp) and the device (k_synth_reads): integer-only
// arithmetic on counter-based hashes, so both produce the same int16 samples for the same (cfg, level table).
#pragma once
#include "rh_gpu.h"
#include "rawhash_amd.h"
#include <cstdint>

#define RH_SY_K 6                    // k-mers of the default pore model (R9.4); a model file of 4^k lines sets its own k (R10: 9)

RH_HD inline uint64_t rh_sy_mix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ULL;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
	return x ^ (x >> 31);
}
RH_HD inline uint64_t rh_sy_rand3(uint64_t seed, uint64_t a, uint64_t b)
{
	return rh_sy_mix64(rh_sy_mix64(seed ^ (a * 0xD6E8FEB86659FD93ULL)) + b * 0xA24BAED4963EE407ULL);
}
RH_HD inline uint32_t rh_sy_genome_base(uint64_t genome_seed, uint32_t chrom, uint32_t pos)
{
	return (uint32_t)(rh_sy_rand3(genome_seed, chrom, pos >> 5) >> ((pos & 31) * 2)) & 3;
}
RH_HD inline uint32_t rh_sy_span(uint32_t n_samples) { return n_samples / 4 + 16; }

struct rh_sy_origin { uint32_t chrom, pos, strand, junk; };
RH_HD inline rh_sy_origin rh_sy_read_origin(const rh_synth_cfg_t &c, uint64_t idx)
{
	rh_sy_origin o;
	const uint64_t h = rh_sy_rand3(c.read_seed, idx, 0);
	o.junk = (h & 1023) < c.junk_per_1024;
	o.chrom = (uint32_t)((h >> 10) % c.n_chrom);
	o.strand = (uint32_t)(h >> 40) & 1;
	o.pos = (uint32_t)(rh_sy_rand3(c.read_seed, idx, 1) % (c.chrom_len - rh_sy_span(c.n_samples)));
	return o;
}

// -log2(u / 65536) in Q8 for u in [1, 65536] (piecewise-linear mantissa)
RH_HD inline uint32_t rh_sy_neg_log2_q8(uint32_t u)
{
	const int i = 31 - __builtin_clz(u);
	const uint32_t frac = ((u << (16 - i)) & 0xFFFF) >> 8;
	return (16u << 8) - (((uint32_t)i << 8) + frac);
}

// level16[kmer] = model level in raw ADC units x 16
RH_HD inline void rh_sy_generate(const rh_synth_cfg_t &c, const int32_t *level16, uint32_t k, uint64_t idx, int16_t *out)
{
	const rh_sy_origin o = rh_sy_read_origin(c, idx);
	const uint32_t span = rh_sy_span(c.n_samples);
	const uint32_t noise_q24 = c.noise_q24 ? c.noise_q24 : 62152u;
	const uint32_t kmask = (1u << (2 * k)) - 1;
	uint32_t kmer = 0, s = 0;
	for (uint32_t j = 0; j < span && s < c.n_samples; ++j) {
		uint32_t b;
		if (o.junk) b = (uint32_t)(rh_sy_rand3(c.read_seed ^ 0x6A756E6BULL, idx, j >> 5) >> ((j & 31) * 2)) & 3;
		else if (!o.strand) b = rh_sy_genome_base(c.genome_seed, o.chrom, o.pos + j);
		else b = 3 - rh_sy_genome_base(c.genome_seed, o.chrom, o.pos + span - 1 - j);
		kmer = ((kmer << 2) | b) & kmask;
		if (j + 1 < k) continue;
		const uint64_t hd = rh_sy_rand3(c.read_seed + 2, idx, j);
		const uint32_t e = rh_sy_neg_log2_q8((uint32_t)(hd & 0xFFFF) + 1) + rh_sy_neg_log2_q8((uint32_t)((hd >> 16) & 0xFFFF) + 1);
		uint32_t dwell = (e * 790u + (1u << 15)) >> 16;
		if (dwell < 1) dwell = 1;
		if (j + 1 == span) dwell = c.n_samples;   // ran out of bases (cannot happen in practice): hold the last level
		for (uint32_t d = 0; d < dwell && s < c.n_samples; ++d, ++s) {
			const uint64_t hn = rh_sy_rand3(c.read_seed + 3, idx, s);
			const int64_t u = (int64_t)(hn & 0xFFFF) + (int64_t)((hn >> 16) & 0xFFFF) + (int64_t)((hn >> 32) & 0xFFFF) + (int64_t)((hn >> 48) & 0xFFFF) - 131070;
			const int64_t n16 = (u * (int64_t)noise_q24) >> 24;
			int64_t v = ((int64_t)level16[kmer] + n16 + 8) >> 4;
			if (v > 32767) v = 32767;
			if (v < -32768) v = -32768;
			out[s] = (int16_t)v;
		}
	}
}
