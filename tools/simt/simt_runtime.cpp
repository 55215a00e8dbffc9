// Fiber scheduler behind tools/simt/hip/hip_runtime.h.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <vector>

float* feta_lds = nullptr;

namespace simt {

dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;

namespace {
struct Barrier {
  int n = 0;
  int count = 0;
  uint64_t gen = 0;
};
struct Lane {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  Barrier* wait = nullptr;
  uint64_t wait_gen = 0;
  int tid = 0;
};
constexpr size_t kStack = 256 * 1024;
std::vector<Lane> lanes;
std::vector<Barrier> wave_bars;
std::vector<WaveScratch> scratch;
Barrier block_bar;
ucontext_t main_ctx;
Lane* cur = nullptr;
const std::function<void()>* body_ = nullptr;

void trampoline() {
  (*body_)();
  cur->done = true;
  swapcontext(&cur->ctx, &main_ctx);
}

void arrive(Barrier& b) {
  uint64_t g = b.gen;
  if (++b.count == b.n) {
    b.count = 0;
    b.gen++;
    return;
  }
  Lane* me = cur;
  me->wait = &b;
  me->wait_gen = g;
  swapcontext(&me->ctx, &main_ctx);
  me->wait = nullptr;
}
}  // namespace

WaveScratch& wave_scratch() { return scratch[cur->tid >> 6]; }
void wave_barrier() { arrive(wave_bars[cur->tid >> 6]); }
void block_barrier() { arrive(block_bar); }

void run_grid(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes) {
  // dynamic LDS of exactly the requested size; the guard behind it is NaN-poisoned so that a
  // kernel touching more LDS than the launch asked for reads NaN (and is reported below)
  constexpr size_t kGuard = 64 * 1024 / 4;
  const size_t lds_n = (lds_bytes + 3) / 4;
  if (lds_bytes > (size_t)kLdsBytes) {
    fprintf(stderr, "simt: %zu bytes of LDS requested, CU has %d\n", lds_bytes, kLdsBytes);
    abort();
  }
  std::vector<float> lds_store(lds_n + kGuard);
  feta_lds = lds_store.data();
  const int nt = (int)block.x;
  if (block.y != 1 || block.z != 1 || grid.z != 1) {
    fprintf(stderr, "simt: only 1-D blocks and 2-D grids are emulated\n");
    abort();
  }
  if ((int)lanes.size() < nt) {
    size_t old = lanes.size();
    lanes.resize(nt);
    for (size_t i = old; i < lanes.size(); ++i) lanes[i].stack = (char*)malloc(kStack);
  }
  const int nw = (nt + 63) / 64;
  wave_bars.assign(nw, Barrier());
  scratch.resize(nw);
  body_ = &body;
  gridDim_ = grid;
  blockDim_ = block;
  for (unsigned bxy = 0; bxy < grid.x * grid.y; ++bxy) {
    for (size_t i = 0; i < lds_n + kGuard; ++i) lds_store[i] = __builtin_nanf("");
    const unsigned bx = bxy % grid.x;
    blockIdx_ = dim3(bx, bxy / grid.x, 0);
    for (int w = 0; w < nw; ++w) {
      wave_bars[w] = Barrier();
      wave_bars[w].n = (w == nw - 1) ? nt - 64 * w : 64;
    }
    block_bar = Barrier();
    block_bar.n = nt;
    for (int t = 0; t < nt; ++t) {
      Lane& l = lanes[t];
      l.done = false;
      l.wait = nullptr;
      l.tid = t;
      getcontext(&l.ctx);
      l.ctx.uc_stack.ss_sp = l.stack;
      l.ctx.uc_stack.ss_size = kStack;
      l.ctx.uc_link = &main_ctx;
      makecontext(&l.ctx, trampoline, 0);
    }
    int remaining = nt;
    while (remaining > 0) {
      bool progress = false;
      for (int t = 0; t < nt; ++t) {
        Lane& l = lanes[t];
        if (l.done) continue;
        if (l.wait && l.wait->gen == l.wait_gen) continue;
        cur = &l;
        threadIdx_ = dim3(t, 0, 0);
        swapcontext(&main_ctx, &l.ctx);
        progress = true;
        if (l.done) --remaining;
      }
      if (remaining == 0) {
        for (size_t i = lds_n; i < lds_n + kGuard; ++i)
          if (lds_store[i] == lds_store[i]) {
            fprintf(stderr, "simt: block %u wrote LDS word %zu, beyond the %zu bytes the launch requested\n",
                    bx, i, lds_bytes);
            abort();
          }
      }
      if (!progress) {
        fprintf(stderr, "simt: deadlock in block %u (divergent collective or early exit before a barrier)\n", bx);
        abort();
      }
    }
  }
  cur = nullptr;
  feta_lds = nullptr;
}

}  // namespace simt
