// TEST TOOL (not product): fiber-based SIMT emulator runtime.  See emu_rt.h.
#include "emu_rt.h"


#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {
enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
constexpr size_t kStack = 256 * 1024;

// Minimal x86-64 System V context switch (callee-saved registers + stack pointer).  ucontext's
// swapcontext issues a sigprocmask syscall per switch, which made shuffle-heavy kernels crawl.
extern "C" void emu_swap(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_swap
.type emu_swap,@function
emu_swap:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_swap,.-emu_swap
)");

struct Fiber {
  void* sp = nullptr;
  State st = DONE;
  dim3 tid;
};

void* g_main_sp = nullptr;
std::vector<Fiber> g_fibers;
std::vector<unsigned char> g_stacks;
int g_cur = -1;
const std::function<void()>* g_body = nullptr;
float g_xchg[1024 * 2];  // per-thread exchange slots (a, b)

void fiber_entry() {
  (*g_body)();
  g_fibers[g_cur].st = DONE;
  emu_swap(&g_fibers[g_cur].sp, g_main_sp);
  abort();  // a finished fiber is never resumed
}

void yield_as(State s) {
  int me = g_cur;
  g_fibers[me].st = s;
  emu_swap(&g_fibers[me].sp, g_main_sp);
  threadIdx = g_fibers[me].tid;
}

void run_block(int nthreads) {
  if ((int)g_fibers.size() < nthreads) g_fibers.resize(nthreads);
  if (g_stacks.size() < kStack * (size_t)nthreads) g_stacks.resize(kStack * (size_t)nthreads);
  for (int i = 0; i < nthreads; ++i) {
    Fiber& f = g_fibers[i];
    // initial frame: six callee-saved slots, then fiber_entry as the return address of emu_swap
    uintptr_t top = reinterpret_cast<uintptr_t>(g_stacks.data() + kStack * (size_t)(i + 1));
    top &= ~(uintptr_t)15;
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;                                   // fake return address of fiber_entry (keeps rsp = 16n+8)
    *--sp = reinterpret_cast<void*>(&fiber_entry);
    for (int r = 0; r < 6; ++r) *--sp = nullptr;
    f.sp = sp;
    f.st = RUNNABLE;
    f.tid = dim3(i % blockDim.x, (i / blockDim.x) % blockDim.y, i / (blockDim.x * blockDim.y));
  }
  for (;;) {
    bool ran = false;
    for (int i = 0; i < nthreads; ++i) {
      if (g_fibers[i].st != RUNNABLE) continue;
      g_cur = i;
      threadIdx = g_fibers[i].tid;
      emu_swap(&g_main_sp, g_fibers[i].sp);
      ran = true;
    }
    // release barriers
    bool released = false;
    int nwaves = (nthreads + 63) / 64;
    for (int w = 0; w < nwaves; ++w) {
      int lo = w * 64, hi = lo + 64 > nthreads ? nthreads : lo + 64;
      int waiting = 0, live = 0;
      for (int i = lo; i < hi; ++i) {
        if (g_fibers[i].st != DONE) ++live;
        if (g_fibers[i].st == WAIT_WAVE) ++waiting;
      }
      if (live > 0 && waiting == live) {
        for (int i = lo; i < hi; ++i)
          if (g_fibers[i].st == WAIT_WAVE) g_fibers[i].st = RUNNABLE;
        released = true;
      }
    }
    int live = 0, waiting = 0;
    for (int i = 0; i < nthreads; ++i) {
      if (g_fibers[i].st != DONE) ++live;
      if (g_fibers[i].st == WAIT_BLOCK) ++waiting;
    }
    if (live == 0) break;
    if (waiting == live) {
      for (int i = 0; i < nthreads; ++i)
        if (g_fibers[i].st == WAIT_BLOCK) g_fibers[i].st = RUNNABLE;
      released = true;
    }
    if (!ran && !released) {
      fprintf(stderr, "emu: deadlock (divergent barrier) in block (%u,%u)\n", blockIdx.x, blockIdx.y);
      abort();
    }
  }
}
}  // namespace

void emu_sync_block() { yield_as(WAIT_BLOCK); }

float emu_shfl(float v, int src_lane, int width) {
  int me = g_cur;
  int lane = me & 63;
  (void)width;
  g_xchg[me * 2] = v;
  yield_as(WAIT_WAVE);
  int src = (me - lane) + (src_lane & 63);
  float r = g_xchg[src * 2];
  yield_as(WAIT_WAVE);
  return r;
}

emu_f32x4 emu_mfma_16x16x4(float a, float b, emu_f32x4 c) {
  // A[i][k]: lane k*16+i ; B[k][j]: lane k*16+j ; D[row=(lane>>4)*4+r][col=lane&15]
  int me = g_cur;
  int lane = me & 63, base = me - lane;
  g_xchg[me * 2] = a;
  g_xchg[me * 2 + 1] = b;
  yield_as(WAIT_WAVE);
  int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(g_xchg[(base + k * 16 + row) * 2], g_xchg[(base + k * 16 + col) * 2 + 1], acc);
    c[r] = acc;
  }
  yield_as(WAIT_WAVE);
  return c;
}

namespace {
uint16_t g_xchg16[1024][16];  // per-thread bf16 operands (a[8] | b[8])
float bf16f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
}  // namespace

emu_f32x16 emu_mfma_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
  // A[m][k]: lane (k / 8) * 32 + m, element k % 8 ; B[k][n]: lane (k / 8) * 32 + n ; D register v: row 8 (v/4) + 4 (lane>>5) + v%4
  int me = g_cur;
  int lane = me & 63, base = me - lane;
  for (int e = 0; e < 8; ++e) {
    g_xchg16[me][e] = a.v[e];
    g_xchg16[me][8 + e] = b.v[e];
  }
  yield_as(WAIT_WAVE);
  int col = lane & 31;
  for (int v = 0; v < 16; ++v) {
    int row = 8 * (v / 4) + 4 * (lane >> 5) + v % 4;
    float acc = c[v];
    for (int k = 0; k < 16; ++k)
      acc = fmaf(bf16f(g_xchg16[base + (k / 8) * 32 + row][k % 8]), bf16f(g_xchg16[base + (k / 8) * 32 + col][8 + k % 8]), acc);
    c[v] = acc;
  }
  yield_as(WAIT_WAVE);
  return c;
}

emu_f32x4 emu_mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
  // A[m][k]: lane (k / 8) * 16 + m, element k % 8 ; B[k][n]: lane (k / 8) * 16 + n ; D register r: row 4 (lane >> 4) + r
  int me = g_cur;
  int lane = me & 63, base = me - lane;
  for (int e = 0; e < 8; ++e) {
    g_xchg16[me][e] = a.v[e];
    g_xchg16[me][8 + e] = b.v[e];
  }
  yield_as(WAIT_WAVE);
  int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (lane >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k)
      acc = fmaf(bf16f(g_xchg16[base + (k / 8) * 16 + row][k % 8]), bf16f(g_xchg16[base + (k / 8) * 16 + col][8 + k % 8]), acc);
    c[r] = acc;
  }
  yield_as(WAIT_WAVE);
  return c;
}

namespace {
const uint16_t* g_trp[1024];  // per-thread run addresses of a transposing read
}
void emu_lds_tr16(const uint16_t* p, uint16_t out[4]) {
  int me = g_cur;
  int lane = me & 63, base = me - lane, grp = lane & ~15, i = lane & 15;
  g_trp[me] = p;
  yield_as(WAIT_WAVE);
  for (int j = 0; j < 4; ++j) out[j] = g_trp[base + grp + 4 * j + (i >> 2)][i & 3];
  yield_as(WAIT_WAVE);
}

void emu_run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
  g_body = &body;
  gridDim = grid;
  blockDim = block;
  int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads > 1024) {
    fprintf(stderr, "emu: block too large\n");
    abort();
  }
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        run_block(nthreads);
      }
  g_body = nullptr;
}
