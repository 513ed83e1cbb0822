// Shared host-side helpers of librawhash_amd (error reporting, small utilities).
#pragma once
#include "rawhash_amd.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

void rh_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

int rh_synth_level_table(const rh_synth_cfg_t *c, const char *model_path, std::vector<int32_t> &level16);
uint32_t rh_synth_model_k(size_t n_levels);

// Environment variables the library reads.  Supported options (documented in INTEGRATION.md) go through getenv as usual: RH_SUB_BATCHES,
// RH_ARENA_MAX_BYTES, RH_CALL_READS_MAX, RH_WHOLE_ROWS_MAX_SAMPLES, RH_BCAST, RH_BCAST_PIECE_BYTES, RH_READS_NO_PIN, RH_TSTAT_CB, RH_BS_TOK_ADV.
// DEVELOPMENT knobs (profiling aids, A/B switches of the sorter and of the record formats, traces) are read through RH_DEVENV and exist only in
// builds with -DRH_DEV (RH_HIPCC_EXTRA=-DRH_DEV python -m rawhash_amd.build --force): the shipped launch paths do not look at them.
#include <cstdlib>
#ifdef RH_DEV
#define RH_DEVENV(name) getenv(name)
#else
#define RH_DEVENV(name) ((const char*)nullptr)
#endif
