// TEST INFRASTRUCTURE ONLY — runtime of the CPU stand-in for HIP (see hip/hip_runtime.h in this directory).
// Fibers (hand-rolled x86-64 context switch), the wavefront/block scheduler, and the few HIP runtime
// calls the engine's host code makes, as plain host memory operations.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <mutex>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <vector>

#if !defined(__x86_64__)
#error "the test emulator's context switch is written for x86-64"
#endif

namespace emu {

Idx3 g_threadIdx, g_blockIdx;
dim3 g_blockDim, g_gridDim;
unsigned char* g_dynLds = nullptr;

extern "C" void emu_switch(void** saveSp, void* newSp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

enum State { ST_READY = 0, ST_WAIT_WAVE, ST_WAIT_BLOCK, ST_DONE };

struct Fiber {
  void* sp = nullptr;
  unsigned char* stack = nullptr;
  int state = ST_DONE;
  int op = 0;
  int arg = 0;
  int width = 64;
  void* site = nullptr;
  uint64_t value = 0;
  uint64_t result = 0;
};

static constexpr size_t STACK_BYTES = 96 * 1024;
static std::vector<Fiber> g_fibers;      // of the block being executed
static void* g_schedSp = nullptr;
static Fiber* g_cur = nullptr;
static Thunk g_thunk;
static std::vector<unsigned char> g_ldsBuf;

static void yield_to_scheduler() {
  Fiber* f = g_cur;
  emu_switch(&f->sp, g_schedSp);
}

static void fiber_main() {
  g_thunk.fn(g_thunk.ctx);
  g_cur->state = ST_DONE;
  yield_to_scheduler();
  abort();  // a finished fiber is never resumed
}

static void prepare(Fiber& f) {
  if (!f.stack) f.stack = (unsigned char*)malloc(STACK_BYTES);
  uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
  void** s = (void**)top;
  *--s = nullptr;               // fake return address of fiber_main (keeps rsp % 16 == 8 at its entry)
  *--s = (void*)&fiber_main;    // popped by emu_switch's ret
  for (int k = 0; k < 6; k++) *--s = nullptr;  // rbp rbx r12..r15
  f.sp = (void*)s;
  f.state = ST_READY;
}

static void resume(unsigned tid) {
  Fiber& f = g_fibers[tid];
  g_cur = &f;
  g_threadIdx.x = tid;
  g_threadIdx.y = g_threadIdx.z = 0;
  emu_switch(&g_schedSp, f.sp);
}

uint64_t wave_collective(int op, uint64_t value, int arg, int width, void* site) {
  Fiber* f = g_cur;
  f->op = op;
  f->value = value;
  f->arg = arg;
  f->width = width;
  f->site = site;
  f->state = ST_WAIT_WAVE;
  yield_to_scheduler();
  return f->result;
}

void block_barrier() {
  g_cur->state = ST_WAIT_BLOCK;
  yield_to_scheduler();
}

// every lane of wave [lo, hi) that waits in a wave collective gets its result; lanes at the same call
// site form one group
static void complete_wave(unsigned lo, unsigned hi) {
  if (getenv("WG_EMU_TRACE_GROUPS")) {  // debugging aid: report a wavefront whose waiting lanes stand at different call sites
    void* first = nullptr;
    bool split = false;
    for (unsigned i = lo; i < hi; i++) {
      const Fiber& f = g_fibers[i];
      if (f.state != ST_WAIT_WAVE) continue;
      if (!first) first = f.site;
      else if (f.site != first) split = true;
    }
    if (split) {
      Dl_info di;
      if (dladdr(first, &di)) fprintf(stderr, "emu: base %p ", di.dli_fbase);
      fprintf(stderr, "emu: split collective:");
      for (unsigned i = lo; i < hi; i++)
        if (g_fibers[i].state == ST_WAIT_WAVE) fprintf(stderr, " %u:%p/%d", i - lo, g_fibers[i].site, g_fibers[i].op);
      fprintf(stderr, "\n");
    }
  }
  for (unsigned i = lo; i < hi; i++) {
    Fiber& f = g_fibers[i];
    if (f.state != ST_WAIT_WAVE) continue;
    const int lane = (int)(i - lo);
    auto peer = [&](int src) -> const Fiber* {
      if (src < 0 || lo + (unsigned)src >= hi) return nullptr;
      const Fiber& p = g_fibers[lo + src];
      return (p.state == ST_WAIT_WAVE && p.site == f.site && p.op == f.op) ? &p : nullptr;
    };
    const int w = f.width;
    switch (f.op) {
      case OP_BALLOT: {
        uint64_t m = 0;
        for (int s = 0; s < (int)(hi - lo); s++) {
          const Fiber* p = peer(s);
          if (p && p->value) m |= 1ULL << s;
        }
        f.result = m;
        break;
      }
      case OP_SHFL: {
        int src = (lane & ~(w - 1)) | (f.arg & (w - 1));
        const Fiber* p = peer(src);
        f.result = p ? p->value : f.value;
        break;
      }
      case OP_SHFL_UP: {
        const Fiber* p = (lane & (w - 1)) >= f.arg ? peer(lane - f.arg) : nullptr;
        f.result = p ? p->value : f.value;
        break;
      }
      case OP_SHFL_XOR: {
        int src = lane ^ f.arg;
        const Fiber* p = (src & ~(w - 1)) == (lane & ~(w - 1)) ? peer(src) : nullptr;
        f.result = p ? p->value : f.value;
        break;
      }
      default: f.result = 0;
    }
  }
  for (unsigned i = lo; i < hi; i++)
    if (g_fibers[i].state == ST_WAIT_WAVE) g_fibers[i].state = ST_READY;
}

static void run_block(unsigned nThreads) {
  if (g_fibers.size() < nThreads) g_fibers.resize(nThreads);
  for (unsigned i = 0; i < nThreads; i++) prepare(g_fibers[i]);
  for (;;) {
    bool anyLive = false, allAtBlockBarrier = true;
    for (unsigned lo = 0; lo < nThreads; lo += 64) {
      const unsigned hi = std::min(nThreads, lo + 64);
      for (;;) {
        bool ran = false;
        for (unsigned i = lo; i < hi; i++)
          if (g_fibers[i].state == ST_READY) {
            resume(i);
            ran = true;
          }
        bool waitWave = false;
        for (unsigned i = lo; i < hi; i++) waitWave |= g_fibers[i].state == ST_WAIT_WAVE;
        if (waitWave)
          complete_wave(lo, hi);
        else if (!ran)
          break;
      }
      for (unsigned i = lo; i < hi; i++) {
        if (g_fibers[i].state == ST_DONE) continue;
        anyLive = true;
        if (g_fibers[i].state != ST_WAIT_BLOCK) allAtBlockBarrier = false;
      }
    }
    if (!anyLive) return;
    if (!allAtBlockBarrier) {
      fprintf(stderr, "emu: scheduler stuck (a live thread is neither ready nor at a barrier)\n");
      abort();
    }
    for (unsigned i = 0; i < nThreads; i++)
      if (g_fibers[i].state == ST_WAIT_BLOCK) g_fibers[i].state = ST_READY;
  }
}

// one launch at a time: the emulated device state (fibers, LDS, the built-in index variables, the kernels' static
// __shared__ storage) is global, and the sharded-engine tests drive several engines from several host threads
static std::mutex g_launchMutex;

void launch(dim3 grid, dim3 block, size_t lds, Thunk t) {
  std::lock_guard<std::mutex> guard(g_launchMutex);
  if (block.y != 1 || block.z != 1 || grid.z != 1) {
    fprintf(stderr, "emu: only 1-D blocks and 2-D grids are supported\n");
    abort();
  }
  g_thunk = t;
  g_blockDim = block;
  g_gridDim = grid;
  if (g_ldsBuf.size() < lds + 64) g_ldsBuf.resize(lds + 64);
  g_dynLds = g_ldsBuf.data();
  for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
      g_blockIdx.x = bx;
      g_blockIdx.y = by;
      g_blockIdx.z = 0;
      memset(g_dynLds, 0xA5, lds);
      run_block(block.x);
    }
}

}  // namespace emu

// ---- HIP runtime calls, as host memory operations -------------------------------------------------
struct emuStream {
  int unused;
};
struct emuEvent {
  std::chrono::steady_clock::time_point t;
};

hipError_t hipMalloc(void** p, size_t n) {
  void* m = malloc(n ? n : 16);
  if (!m) return hipErrorInvalidValue;
  memset(m, 0xCB, n);  // device memory is not zeroed: make reads of uninitialised memory visible
  *p = m;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
  memmove(dst, src, n);
  return hipSuccess;
}
// ---- stream capture / graphs (see hip_runtime.h)
struct emuGraph {
  std::vector<std::function<void()>> ops;
};
static emuGraph* g_capture = nullptr;
namespace emu {
bool capturing() { return g_capture != nullptr; }
void record(std::function<void()> f) { g_capture->ops.push_back(std::move(f)); }
}  // namespace emu
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
  if (g_capture) return hipErrorInvalidValue;
  g_capture = new emuGraph();
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
  *g = g_capture;
  g_capture = nullptr;
  return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) {
  *e = new emuGraph(*g);
  return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
  if (getenv("EMU_TRACE_GRAPH")) fprintf(stderr, "emu: hipGraphLaunch of %zu recorded operations\n", e->ops.size());
  for (auto& f : e->ops) f();
  return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) {
  delete g;
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t) {
  if (g_capture) return hipErrorInvalidValue;  // (the engine captures no copies)
  return hipMemcpy(dst, src, n, k);
}
hipError_t hipMemset(void* dst, int v, size_t n) {
  memset(dst, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t) {
  if (g_capture) {
    g_capture->ops.push_back([=]() { memset(dst, v, n); });
    return hipSuccess;
  }
  return hipMemset(dst, v, n);
}
hipError_t hipStreamCreate(hipStream_t* s) {
  *s = new emuStream();
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
// WG_EMU_DEVICES=<k>: the emulator answers as a box of k devices (they all are this process's memory) — multi-rank
// host logic that pins rank r to device r (bench.py --gpus N over gloo) runs unchanged
static int emu_devices() {
  const char* v = getenv("WG_EMU_DEVICES");
  const int k = v ? atoi(v) : 1;
  return k > 0 ? k : 1;
}
static thread_local int emuCurDevice = 0;
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= emu_devices()) return hipErrorInvalidValue;
  emuCurDevice = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) {
  *d = emuCurDevice;
  return hipSuccess;
}
hipError_t hipGetDeviceCount(int* n) {
  *n = emu_devices();
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu: error"; }
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new emuEvent();
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipMemGetInfo(size_t* freeB, size_t* totalB) {
  *freeB = *totalB = (size_t)32 << 30;
  return hipSuccess;
}
