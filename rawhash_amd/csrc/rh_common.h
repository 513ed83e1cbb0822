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
