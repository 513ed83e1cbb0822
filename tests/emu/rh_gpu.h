// TEST INFRASTRUCTURE ONLY: SIMT logic emulator that shadows rawhash_amd/csrc/rh_gpu.h when the *unchanged* kernel
// and host sources are compiled with g++ into tests/emu/_build/librawhash_emu.so (see tests/emu/build_emu.py).
// It lets the `not gpu` tests run the real kernel code (block barriers, wave ballots, shuffles, LDS) on the CPU so
// indexing/ordering bugs are caught without a GPU.  Each thread of a block is a ucontext fiber; blocks run one after
// another.  Nothing here is compiled into, or loaded by, the product library.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define RH_HD
#define RH_DEV static inline
#define RH_WAVE 64
#define RH_INLINE_LAMBDA
#define RH_VALUE_READY(x) ((void)0)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

void rh_set_error(const char *fmt, ...);

struct int2 { int x, y; };
struct float4 { float x, y, z, w; };
struct uint4 { unsigned int x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

// ---- device-side intrinsics
void emu_syncthreads();
unsigned long long emu_ballot(int pred);
unsigned long long emu_shfl_down_bits(unsigned long long bits, unsigned delta);
unsigned long long emu_shfl_bits(unsigned long long bits, int op, unsigned par);   // op: 2 down, 3 up, 4 xor, 5 idx
#define __syncthreads() emu_syncthreads()
static inline void __threadfence() {}
static inline void __threadfence_block() {}
#define RH_WG_FENCE() ((void)0)
#define __ballot(p) emu_ballot((p) ? 1 : 0)
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline double __shfl_down(double v, int d) { unsigned long long b; memcpy(&b, &v, 8); b = emu_shfl_down_bits(b, (unsigned)d); memcpy(&v, &b, 8); return v; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline unsigned __shfl_up(unsigned v, int d) { return (unsigned)emu_shfl_bits(v, 3, (unsigned)d); }
static inline unsigned __shfl_xor(unsigned v, int d) { return (unsigned)emu_shfl_bits(v, 4, (unsigned)d); }
static inline unsigned long long __shfl_xor(unsigned long long v, int d) { return emu_shfl_bits(v, 4, (unsigned)d); }
static inline unsigned long __shfl_xor(unsigned long v, int d) { return (unsigned long)emu_shfl_bits(v, 4, (unsigned)d); }
static inline unsigned long __shfl_up(unsigned long v, int d) { return (unsigned long)emu_shfl_bits(v, 3, (unsigned)d); }
static inline unsigned long __shfl(unsigned long v, int lane) { return (unsigned long)emu_shfl_bits(v, 5, (unsigned)lane); }
static inline unsigned __shfl(unsigned v, int lane) { return (unsigned)emu_shfl_bits(v, 5, (unsigned)lane); }
static inline int __shfl(int v, int lane) { return (int)emu_shfl_bits((unsigned)v, 5, (unsigned)lane); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p += v; return o; }
static inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p |= v; return o; }
static inline unsigned long long atomicAnd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p &= v; return o; }
static inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p |= v; return o; }
static inline unsigned atomicAnd(unsigned *p, unsigned v) { unsigned o = *p; *p &= v; return o; }
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long c, unsigned long long v) { unsigned long long o = *p; if (o == c) *p = v; return o; }
static inline unsigned atomicMax(unsigned *p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }

static inline unsigned rh_readlane(unsigned v, unsigned l) { return (unsigned)emu_shfl_bits(v, 5, l); }
static inline unsigned rh_writelane(unsigned v, unsigned val, unsigned l) { return (threadIdx.x & 63u) == l ? val : v; }
static inline void rh_writelane2(unsigned &a, unsigned &b, unsigned va, unsigned vb, unsigned l) { if ((threadIdx.x & 63u) == l) { a = va; b = vb; } }
static inline unsigned rh_uniform(unsigned v) { return v; }
static inline unsigned rh_and_or(unsigned a, unsigned m, unsigned o) { return (a & m) | o; }
static inline void rh_tok_advance(unsigned &jr, unsigned &head, const unsigned char *ring, unsigned, unsigned l, unsigned lane) { if (lane == l) { jr += 1u; head = ring[jr & 63u]; } }
static inline void rh_lds_wait(unsigned &) {}
static inline unsigned rh_lds_addr(const void *) { return 0u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline unsigned rh_wave_shr1(unsigned v, unsigned first) { const unsigned up = (unsigned)emu_shfl_bits(v, 3, 1u); return (threadIdx.x & 63u) == 0 ? first : up; }
static inline int rh_quad_perm_0022(int v) { return (int)emu_shfl_bits((unsigned)v, 5, (threadIdx.x & 63u) & ~1u); }
static inline int rh_quad_perm_1133(int v) { return (int)emu_shfl_bits((unsigned)v, 5, (threadIdx.x & 63u) | 1u); }
#define RH_WAVE_SYNC() ((void)emu_ballot(1))
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __ATOMIC_RELAXED_DEFINED 1

// ---- host runtime subset
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1 };
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xCD, n); return *p ? hipSuccess : 2; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)1 << 46; *t = (size_t)1 << 36; return hipSuccess; }
enum { hipErrorOutOfMemory = 2 };
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
enum { hipMemoryTypeHost = 0 };
struct hipPointerAttribute_t { int type; void *devicePointer, *hostPointer; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) { a->type = hipMemoryTypeHost; a->devicePointer = (void*)p; a->hostPointer = (void*)p; return hipSuccess; }   // (the emulator's "device" reads any host memory)
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeer(void *d, int, const void *s, int, size_t n) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, int) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }

#define RH_HIP(call)                                                                                   \
	do {                                                                                               \
		hipError_t e_ = (call);                                                                        \
		if (e_ != hipSuccess) { rh_set_error("%s:%d: %s failed", __FILE__, __LINE__, #call); return -1; } \
	} while (0)

void emu_launch(unsigned grid, unsigned block, const std::function<void()> &body);
#define RH_LAUNCH(kernel, grid, block, lds, stream, ...) emu_launch((unsigned)(grid), (unsigned)(block), [&]() { kernel(__VA_ARGS__); })

#define RH_HIP_VOID(call) do { (void)(call); } while (0)
