// HIP runtime glue for gfx950: error checks, launch macro, host/device qualifier.
// (tests/emu/ holds a same-named header that shadows this one when the kernel sources are compiled for the
//  CPU-side SIMT logic emulator used by the `not gpu` tests; the product build never sees it.)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define RH_HD __host__ __device__
#define RH_DEV __device__ __forceinline__
#define RH_WAVE 64
#define RH_INLINE_LAMBDA __attribute__((always_inline))   // a device lambda that captures register arrays by reference: out of line they would live in scratch

void rh_set_error(const char *fmt, ...);

#define RH_HIP(call)                                                                                   \
	do {                                                                                               \
		hipError_t e_ = (call);                                                                        \
		if (e_ != hipSuccess) {                                                                        \
			rh_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));    \
			return -1;                                                                                 \
		}                                                                                              \
	} while (0)

// kernel<<<grid, block, lds, stream>>>(args...)
#define RH_LAUNCH(kernel, grid, block, lds, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, stream, __VA_ARGS__)

// wave-uniform register file tricks: lane `l` of a VGPR read / written with a wave-uniform lane index
// (the amdgcn builtins only exist in the device pass of hipcc; the host pass just needs the declarations to parse)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t rh_readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// v with lane l := val; val and l must be wave-uniform values produced by scalar instructions (the s_and / s_add of the
// callers), which keeps clear of the "VALU-written SGPR as lane select" hazard the assembler cannot see inside asm
__device__ __forceinline__ uint32_t rh_writelane(uint32_t v, uint32_t val, uint32_t l) { asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(val), "s"(l) : "m0"); return v; }   // (one SGPR + M0: constant-bus limit)
__device__ __forceinline__ void rh_writelane2(uint32_t &a, uint32_t &b, uint32_t va, uint32_t vb, uint32_t l) { asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(a), "+v"(b) : "s"(va), "s"(vb), "s"(l) : "m0"); }   // two registers, same lane: one M0 load
__device__ __forceinline__ uint32_t rh_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the value is in its register from here on: a load that produces it is waited for at this point, not at its first use (a loop that keeps
// other loads in flight must not be the place where the compiler waits for memory)
#define RH_VALUE_READY(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ uint32_t rh_and_or(uint32_t a, uint32_t m, uint32_t o) { uint32_t r; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(m), "v"(o)); return r; }   // (a & m) | o in one instruction
// The token walker's per-pop advance of ONE lane (l: wave-uniform, in an SGPR): jr += 1, head = LDS byte at ringa | (jr & 63).
// v_cmpx selects the lane (EXEC written by the compare itself); EXEC is saved before and restored after, whatever it was - where
// `if (lane == l)` compiles to v_cmp + s_and_saveexec + s_cbranch_execz + s_or: the walk is bound by the number of instructions a SIMD
// issues per pop, not by their kind.
// The LDS read is ASYNCHRONOUS and the compiler does not know it: `head` is only defined after rh_lds_wait(head), which the walker
// calls before the next v_readlane of it.  Nothing may read `head` in between - the compiler has no reason to (head is not used between
// the two statements), and tests/test_abi.py::test_token_walker_isa_keeps_lds_read_private checks the generated code for it after every
// build (a register copy or spill of `head` inside that window would read stale data); RH_BS_TOK_ADV=0 selects the compiler-scheduled pop.
__device__ __forceinline__ void rh_tok_advance(uint32_t &jr, uint32_t &head, const uint8_t *, uint32_t ringa, uint32_t l, uint32_t lane)
{
	uint32_t ad;
	uint64_t sv;
	asm volatile("s_mov_b64 %[sv], exec\n\t"
	             "v_cmpx_eq_u32_e32 vcc, %[l], %[ln]\n\t"
	             "v_add_u32_e32 %[jr], 1, %[jr]\n\t"
	             "v_and_or_b32 %[ad], %[jr], 63, %[ra]\n\t"
	             "ds_read_u8 %[hd], %[ad]\n\t"
	             "s_mov_b64 exec, %[sv]"
	             : [jr] "+v"(jr), [hd] "+v"(head), [ad] "=&v"(ad), [sv] "=&s"(sv)
	             : [l] "s"(l), [ln] "v"(lane), [ra] "v"(ringa) : "memory", "vcc");
}
__device__ __forceinline__ void rh_lds_wait(uint32_t &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); }
__device__ __forceinline__ uint32_t rh_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }   // byte offset of a __shared__ object in the workgroup's LDS
// value of lane (l & ~1) / (l | 1) of each lane pair (DPP quad permutes: VALU speed, no LDS)
__device__ __forceinline__ int32_t rh_quad_perm_0022(int32_t v) { return __builtin_amdgcn_mov_dpp(v, 0xA0, 0xF, 0xF, true); }
__device__ __forceinline__ int32_t rh_quad_perm_1133(int32_t v) { return __builtin_amdgcn_mov_dpp(v, 0xF5, 0xF, 0xF, true); }
// value of the lane below (lane 0 keeps `first`): a DPP whole-wave shift, VALU speed (a shuffle goes through the LDS crossbar)
__device__ __forceinline__ uint32_t rh_wave_shr1(uint32_t v, uint32_t first) { return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xF, 0xF, false); }
#else
__device__ uint32_t rh_wave_shr1(uint32_t v, uint32_t first);
__device__ void rh_tok_advance(uint32_t &jr, uint32_t &head, const uint8_t *ring, uint32_t ringa, uint32_t l, uint32_t lane);
__device__ void rh_lds_wait(uint32_t &v);
__device__ uint32_t rh_lds_addr(const void *p);
__device__ int32_t rh_quad_perm_0022(int32_t v);
__device__ int32_t rh_quad_perm_1133(int32_t v);
__device__ uint32_t rh_readlane(uint32_t v, uint32_t l);
__device__ uint32_t rh_writelane(uint32_t v, uint32_t val, uint32_t l);
__device__ void rh_writelane2(uint32_t &a, uint32_t &b, uint32_t va, uint32_t vb, uint32_t l);
__device__ uint32_t rh_uniform(uint32_t v);
#define RH_VALUE_READY(x) ((void)0)
__device__ uint32_t rh_and_or(uint32_t a, uint32_t m, uint32_t o);
#endif

// all lanes of the wavefront have executed everything above (lock step on the GPU: only a compiler-level barrier)
#define RH_WAVE_SYNC() __builtin_amdgcn_wave_barrier()

// order this wavefront's / workgroup's memory operations without touching the caches (a device-scope fence writes the L2 back)
#define RH_WG_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")

#define RH_HIP_VOID(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) rh_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); } while (0)
