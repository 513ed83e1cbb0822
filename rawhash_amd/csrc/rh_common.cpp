#include "rh_common.h"

static thread_local char g_err[1024] = "";

void rh_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

extern "C" const char *rh_last_error(void) { return g_err; }
extern "C" const char *rh_version(void) { return "rawhash_amd 0.1 (path of RawHash2 v2.1)"; }

