// TEST INFRASTRUCTURE ONLY — a CPU stand-in for <hip/hip_runtime.h>.
//
// tests/emu builds the PRODUCT's kernel sources (wittgenstein_amd/csrc/*.hip*) with g++ against this
// header so that the engine's logic can be checked against the oracle in the GPU-less build container
// (`pytest -m "not gpu"`). It is never part of libwittgpu.so and the package never loads it: only
// tests/conftest.py's `emu` fixture does, explicitly. Performance claims and the parity claims proper
// come from the real library on an MI355X (`pytest -m gpu`).
//
// Execution model reproduced: a launch runs its blocks one after the other; the threads of a block are
// fibers; a wavefront is 64 consecutive threads. Wave collectives (__ballot, __shfl*, wave barrier,
// __threadfence_block) complete when every live lane of the wavefront waits in a collective, lanes
// waiting at the same source call site forming one group (= the exec mask of that instruction on the GPU);
// __syncthreads completes when every live thread of the block waits in it. Lanes of a wavefront do NOT
// run in lock-step between collectives (lane 0 runs until it blocks, then lane 1, ...), so code that
// relies on lock-step without a wave barrier fails here — stricter than the hardware, on purpose.
#pragma once
#include <chrono>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <type_traits>
#include <utility>
#include <functional>
#include <tuple>

#define WG_EMU 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define WG_DYN_LDS(T, name) T* name = (T*)::emu::g_dynLds
#define WG_GRID_DIV 128

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
constexpr hipError_t hipErrorNotReady = 600;
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

hipError_t hipMalloc(void** p, size_t n);
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s = nullptr);
hipError_t hipMemset(void* dst, int v, size_t n);
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s = nullptr);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
constexpr unsigned hipEventDisableTiming = 2;
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);  // (launches run at once here: nothing to wait for)
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemGetInfo(size_t* freeB, size_t* totalB);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipStreamQuery(hipStream_t s);
inline void __threadfence_system() {}
// the constant-rate device clock (s_memrealtime) the engine's graph-mode profiler stamps with, and its rate in kHz
inline unsigned long long wall_clock64() {
  return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
  *v = 100000;
  return hipSuccess;
}
// stream capture / graphs: while a stream is capturing, launches and async memsets are recorded instead of executed;
// hipGraphLaunch replays the record (what the engine's WG_GRAPH=1 path relies on)
typedef struct emuGraph* hipGraph_t;
typedef struct emuGraph* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
hipError_t hipGraphDestroy(hipGraph_t g);

namespace emu {
struct Idx3 {
  unsigned x, y, z;
};
extern Idx3 g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
extern unsigned char* g_dynLds;

enum Op { OP_BALLOT = 0, OP_SHFL, OP_SHFL_UP, OP_SHFL_XOR, OP_BARRIER };
// blocks the calling lane in a wave collective; returns the op's result for this lane
uint64_t wave_collective(int op, uint64_t value, int arg, int width, void* site);
void block_barrier();

struct Thunk {
  void (*fn)(void*);
  void* ctx;
};
void launch(dim3 grid, dim3 block, size_t lds, Thunk t);

template <class T>
inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  uint64_t b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <class T>
inline T from_bits(uint64_t b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}
}  // namespace emu

#define threadIdx (::emu::g_threadIdx)
#define blockIdx (::emu::g_blockIdx)
#define blockDim (::emu::g_blockDim)
#define gridDim (::emu::g_gridDim)

// ---- wave / block collectives. A call site is identified by a static marker at the place of the call
// in the SOURCE (the return address would not do: the optimiser duplicates code paths).
#define EMU_SITE ([]() -> void* { static char marker; return (void*)&marker; }())
namespace emu {
inline uint64_t ballot(void* site, int pred) { return wave_collective(OP_BALLOT, pred ? 1 : 0, 0, 64, site); }
template <class T>
inline T shfl(void* site, T v, int src, int width = 64) {
  return from_bits<T>(wave_collective(OP_SHFL, to_bits(v), src, width, site));
}
template <class T>
inline T shfl_up(void* site, T v, unsigned delta, int width = 64) {
  return from_bits<T>(wave_collective(OP_SHFL_UP, to_bits(v), (int)delta, width, site));
}
template <class T>
inline T shfl_xor(void* site, T v, int mask, int width = 64) {
  return from_bits<T>(wave_collective(OP_SHFL_XOR, to_bits(v), mask, width, site));
}
inline void wave_barrier(void* site) { (void)wave_collective(OP_BARRIER, 0, 0, 64, site); }
}  // namespace emu
#define __ballot(...) ::emu::ballot(EMU_SITE, __VA_ARGS__)
#define __shfl(...) ::emu::shfl(EMU_SITE, __VA_ARGS__)
#define __shfl_up(...) ::emu::shfl_up(EMU_SITE, __VA_ARGS__)
#define __shfl_xor(...) ::emu::shfl_xor(EMU_SITE, __VA_ARGS__)
// the product's SGPR hints: lane l's value / the (wave-uniform) value every lane holds
#define WG_READLANE(v, l) ((uint32_t)::emu::shfl(EMU_SITE, (uint32_t)(v), (l), 64))
#define WG_READFIRST(v) ((uint32_t)(v))
#define __builtin_amdgcn_wave_barrier() ::emu::wave_barrier(EMU_SITE)
#define __threadfence_block() ::emu::wave_barrier(EMU_SITE)
inline void __threadfence() {}
inline void __syncthreads() { emu::block_barrier(); }

// ---- bit intrinsics
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffsll(unsigned long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline float __fsqrt_rn(float v) { return sqrtf(v); }

// ---- atomics: one host thread runs every fiber, so plain read-modify-write is atomic
template <class T, class U>
inline T atomicAdd(T* p, U v) {
  T o = *p;
  *p = (T)(o + (T)v);
  return o;
}
template <class T, class U>
inline T atomicSub(T* p, U v) {
  T o = *p;
  *p = (T)(o - (T)v);
  return o;
}
template <class T, class U>
inline T atomicOr(T* p, U v) {
  T o = *p;
  *p = (T)(o | (T)v);
  return o;
}
template <class T, class U>
inline T atomicAnd(T* p, U v) {
  T o = *p;
  *p = (T)(o & (T)v);
  return o;
}
template <class T, class U>
inline T atomicExch(T* p, U v) {
  T o = *p;
  *p = (T)v;
  return o;
}
template <class T, class U>
inline T atomicMax(T* p, U v) {
  T o = *p;
  if ((T)v > o) *p = (T)v;
  return o;
}
template <class T, class U>
inline T atomicMin(T* p, U v) {
  T o = *p;
  if ((T)v < o) *p = (T)v;
  return o;
}
template <class T, class U, class V>
inline T atomicCAS(T* p, U cmp, V v) {
  T o = *p;
  if (o == (T)cmp) *p = (T)v;
  return o;
}
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))

// HIP's device-side min/max accept mixed integer types
template <class A, class B>
inline typename std::common_type<A, B>::type min(A a, B b) {
  typedef typename std::common_type<A, B>::type C;
  return (C)a < (C)b ? (C)a : (C)b;
}
template <class A, class B>
inline typename std::common_type<A, B>::type max(A a, B b) {
  typedef typename std::common_type<A, B>::type C;
  return (C)a > (C)b ? (C)a : (C)b;
}

// ---- launch
namespace emu {
bool capturing();
void record(std::function<void()> f);
template <class... KArgs, class... Args>
inline void launch_kernel(void (*k)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
  if (capturing()) {  // copy the arguments: the record outlives this call
    std::tuple<std::decay_t<KArgs>...> copy(static_cast<KArgs>(args)...);
    record([=]() {
      std::apply([&](auto&... a) {
        auto call = [&]() { k(a...); };
        typedef decltype(call) L;
        Thunk t;
        t.fn = [](void* c) { (*(L*)c)(); };
        t.ctx = &call;
        launch(grid, block, lds, t);
      }, const_cast<std::tuple<std::decay_t<KArgs>...>&>(copy));
    });
    return;
  }
  auto call = [&]() { k(static_cast<KArgs>(args)...); };
  typedef decltype(call) L;
  Thunk t;
  t.fn = [](void* c) { (*(L*)c)(); };
  t.ctx = &call;
  launch(grid, block, lds, t);
}
}  // namespace emu
#define hipLaunchKernelGGL(...) ::emu::launch_kernel(__VA_ARGS__)
