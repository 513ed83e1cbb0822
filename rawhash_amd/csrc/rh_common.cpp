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

// The HIP runtime multiplexes a process's streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and reads it when it
// initialises.  A call's sub-batches and the calls in flight are 4 - 6 streams whose kernels are meant to overlap (measured on MI355X, human-scale
// index: four sub-batch streams 29.2 k reads/s on 4 queues, 36.9 k on 8; 12 500-read calls, two in flight: 18.6 k -> 29.8 k upload-inclusive), so the
// library asks for 8 when it is loaded - unless the variable is set already, and to no effect if the runtime was initialised before.
__attribute__((constructor)) static void rh_default_hw_queues(void) { setenv("GPU_MAX_HW_QUEUES", "8", 0); }
