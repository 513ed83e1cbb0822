// TEST INFRASTRUCTURE ONLY: fiber scheduler behind tests/emu/rh_gpu.h (see there).
#include "rh_gpu.h"
#include <ucontext.h>
#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {
enum { RUN = 0, AT_BARRIER, AT_WAVE, DONE };
enum { OP_BALLOT = 1, OP_SHFL_DOWN = 2, OP_SHFL_UP = 3, OP_SHFL_XOR = 4, OP_SHFL_IDX = 5 };
struct Fiber {
	ucontext_t ctx; char *stack = nullptr; int state = DONE; unsigned tid = 0;
	int op = 0; unsigned long long in = 0, out = 0; unsigned par = 0;
};
const size_t kStack = 512 << 10;
std::vector<Fiber> g_f;
ucontext_t g_main;
Fiber *g_cur = nullptr;
const std::function<void()> *g_body = nullptr;

void trampoline() { (*g_body)(); g_cur->state = DONE; swapcontext(&g_cur->ctx, &g_main); }
void yield_to_main() { swapcontext(&g_cur->ctx, &g_main); }
}

void emu_syncthreads() { g_cur->state = AT_BARRIER; yield_to_main(); }
unsigned long long emu_ballot(int pred) { g_cur->state = AT_WAVE; g_cur->op = OP_BALLOT; g_cur->in = pred ? 1 : 0; yield_to_main(); return g_cur->out; }
unsigned long long emu_shfl_bits(unsigned long long bits, int op, unsigned par) { g_cur->state = AT_WAVE; g_cur->op = op; g_cur->in = bits; g_cur->par = par; yield_to_main(); return g_cur->out; }
unsigned long long emu_shfl_down_bits(unsigned long long bits, unsigned delta) { return emu_shfl_bits(bits, OP_SHFL_DOWN, delta); }

void emu_launch(unsigned grid, unsigned block, const std::function<void()> &body)
{
	if (g_f.size() < block) g_f.resize(block);
	g_body = &body;
	gridDim = dim3(grid); blockDim = dim3(block);
	for (unsigned b = 0; b < grid; ++b) {
		blockIdx = dim3(b);
		for (unsigned t = 0; t < block; ++t) {
			Fiber &f = g_f[t];
			if (!f.stack) f.stack = (char*)malloc(kStack);
			getcontext(&f.ctx);
			f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = &g_main;
			makecontext(&f.ctx, trampoline, 0);
			f.state = RUN; f.tid = t;
		}
		for (;;) {
			bool progressed = false;
			for (unsigned t = 0; t < block; ++t) {
				Fiber &f = g_f[t];
				if (f.state != RUN) continue;
				g_cur = &f; threadIdx = dim3(t);
				swapcontext(&g_main, &f.ctx);
				progressed = true;
			}
			// wave collectives complete once no lane of the wave can still run
			unsigned n_done = 0, n_bar = 0, n_wave = 0;
			for (unsigned w0 = 0; w0 < block; w0 += 64) {
				const unsigned w1 = w0 + 64 < block ? w0 + 64 : block;
				bool any_run = false, any_wave = false;
				for (unsigned t = w0; t < w1; ++t) { any_run |= g_f[t].state == RUN; any_wave |= g_f[t].state == AT_WAVE; }
				if (any_run || !any_wave) continue;
				unsigned long long mask = 0;
				for (unsigned t = w0; t < w1; ++t) if (g_f[t].state == AT_WAVE && g_f[t].op == OP_BALLOT && g_f[t].in) mask |= 1ull << (t - w0);
				for (unsigned t = w0; t < w1; ++t) {
					Fiber &f = g_f[t];
					if (f.state != AT_WAVE) continue;
					if (f.op == OP_BALLOT) f.out = mask;
					else {
						const unsigned l = t - w0;
						long src = -1;
						if (f.op == OP_SHFL_DOWN) src = (long)l + f.par;
						else if (f.op == OP_SHFL_UP) src = (long)l - (long)f.par;
						else if (f.op == OP_SHFL_XOR) src = (long)(l ^ f.par);
						else if (f.op == OP_SHFL_IDX) src = (long)(f.par & 63);
						const long st = src + w0;
						f.out = (src >= 0 && src < 64 && st < (long)w1 && g_f[st].state == AT_WAVE && g_f[st].op == f.op) ? g_f[st].in : f.in;
					}
				}
				for (unsigned t = w0; t < w1; ++t) if (g_f[t].state == AT_WAVE) { g_f[t].state = RUN; progressed = true; }
			}
			for (unsigned t = 0; t < block; ++t) { n_done += g_f[t].state == DONE; n_bar += g_f[t].state == AT_BARRIER; n_wave += g_f[t].state == AT_WAVE; }
			if (n_done == block) break;
			if (n_bar + n_done == block) { for (unsigned t = 0; t < block; ++t) if (g_f[t].state == AT_BARRIER) g_f[t].state = RUN; progressed = true; }
			if (!progressed) { fprintf(stderr, "emu: deadlock in block %u (barrier %u wave %u done %u of %u)\n", b, n_bar, n_wave, n_done, block); abort(); }
		}
	}
}
