// Shared host-side helpers of librawhash_amd (error reporting, small utilities).
#pragma once
#include "rawhash_amd.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

void rh_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// splitmix64 finaliser: the only mixing primitive of the synthetic generator (integer-only => same bytes everywhere)
static inline uint64_t rh_mix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ULL;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
	return x ^ (x >> 31);
}
static inline uint64_t rh_rand3(uint64_t seed, uint64_t a, uint64_t b)
{
	return rh_mix64(rh_mix64(seed ^ (a * 0xD6E8FEB86659FD93ULL)) + b * 0xA24BAED4963EE407ULL);
}
