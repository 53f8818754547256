// Generic (protocol-independent) CDNA4 kernels of the time-stepped engine. gfx950 only: 64-wide
// wavefronts are hard-coded (ballot masks are 64-bit, one simulated node per wavefront in the
// delivery kernels, lanes = 64-bit words of that node's bitsets).
//
// One simulated millisecond t (C/Network.java:533-637) =
//   [conditional-task phase — protocol kernels, see proto_handel.hip.h]
//   expand   bucket t in LIFO order, MultipleDestEnvelope runs unrolled -> events[0..E); the same scan
//            hands every event a private outbox slice (protocol emission bound) and threads the event
//            onto its destination node's inbox list (atomicExch on the node's head: distinct addresses)
//   deliver  one wavefront per node with >=1 event applies them in event order  -> per-event outbox slices
//   order    scan of per-event record/draw counts (+ statistics)               -> global push order + draw index
//   resolve  seed = rd.nextInt() by LCG jump-ahead, latency, arrival, drops     -> ordered outbox + tile histograms
//   append   stable multisplit of the ordered outbox by arrival-ms (per-1024-record tile histogram,
//            wave-ballot match for the in-tile rank) onto the tail of each arrival bucket
// No kernel of the per-ms pipeline performs a same-address atomic per event or per record: on MI355X
// one L2 atomic unit retires ~88 same-address atomics/us, which at 10^4..10^5 events per ms was the
// whole cost of the first version of this engine (profiles/r01a_kernel_stats.md).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "engine.h"

namespace wg {

#define WG_LANE (threadIdx.x & 63)

// Investigation builds only (-DWG_KPROF): lane 0 of a wavefront adds the cycles since its previous mark
// to Globals::kprof[slot]. Marks sit where the code first consumes loaded data, so a region's cycles
// are mostly the wait for its loads.
#ifdef WG_KPROF
// (per-wavefront rows of Globals::kprofBuf — distinct addresses, no-return atomics: a same-address counter would make
// the instrumentation the run time, one L2 atomic unit retires ~88 same-address atomics/us)
#define KPROF_WAVE ((((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & (KPROF_WAVES - 1)) * 32)
#define KPROF_DECL unsigned long long _kp = __builtin_readcyclecounter()
#define KPROF_MARK(g, slot)                                                            \
  do {                                                                                 \
    const unsigned long long _n = __builtin_readcyclecounter();                        \
    if (WG_LANE == 0) atomicAdd(F(&(g)->kprofBuf[KPROF_WAVE + (slot)]), _n - _kp);        \
    _kp = _n;                                                                          \
  } while (0)
#define KPROF_COUNT(g, slot) do { if (WG_LANE == 0) atomicAdd(F(&(g)->kprofBuf[KPROF_WAVE + (slot)]), 1ULL); } while (0)
#define KPROF_ADD(g, slot, v) do { if (WG_LANE == 0) atomicAdd(F(&(g)->kprofBuf[KPROF_WAVE + (slot)]), (unsigned long long)(v)); } while (0)
#else
#define KPROF_DECL
#define KPROF_MARK(g, slot)
#define KPROF_COUNT(g, slot)
#define KPROF_ADD(g, slot, v)
#endif

// dynamic LDS of a kernel (a macro so that tests/emu, which builds these kernels for its CPU wave
// emulator, can bind the name to its own buffer)
#ifndef WG_DYN_LDS
#define WG_DYN_LDS(T, name) extern __shared__ T name[]
#endif

#ifndef WG_READLANE
#define WG_READLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (l)))  // wave-uniform result (an SGPR)
#define WG_READFIRST(v) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(v)))
#endif

// whole records to / from global memory (a struct's copy constructor takes a generic reference: the copy goes through
// memcpy, which the compiler turns into the widest global_load / global_store the alignment allows)
template <class T>
__device__ __forceinline__ T gld(const T WG_G* p) {
  T v;
  __builtin_memcpy(&v, p, sizeof(T));
  return v;
}
template <class T>
__device__ __forceinline__ void gst(T WG_G* p, const T& v) {
  __builtin_memcpy(p, &v, sizeof(T));
}
// generic view of a global pointer, for the atomic builtins (the compiler still sees where it came from: global_atomic_*)
template <class T>
__device__ __forceinline__ T* F(T WG_G* p) {
  return (T*)p;
}

__device__ __forceinline__ void set_err(Globals WG_G* g, uint32_t bit) { atomicOr(F(&g->err), bit); }

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int delta) {
  uint32_t lo = __shfl_up((uint32_t)v, delta, 64), hi = __shfl_up((uint32_t)(v >> 32), delta, 64);
  return ((uint64_t)hi << 32) | lo;
}
// ---- wave-wide reductions and scans on the DPP path ------------------------------------------------------------
// `__shfl*` lowers to ds_bpermute_b32: every step of a 6-step butterfly is an LDS-crossbar round trip (~100 cycles) the
// next step depends on — with four wavefronts a SIMD that latency is the cost of a node visit's reductions (DESIGN.md
// §3.1). The DPP row operations move data between lanes inside the VALU (quad_perm / row_ror inside a row of 16,
// row_bcast:15 / :31 across rows — gfx9-family encodings), a few cycles a step; the total is read from lane 63 with
// v_readlane, i.e. it lands in an SGPR. tests/emu (no DPP) takes the generic branch.
struct OpAdd {
  template <class T>
  __device__ static T f(T a, T b) { return a + b; }
};
struct OpMin {
  template <class T>
  __device__ static T f(T a, T b) { return a < b ? a : b; }
};
struct OpMax {
  template <class T>
  __device__ static T f(T a, T b) { return a > b ? a : b; }
};
#if defined(__HIPCC__) && !defined(WG_NO_DPP)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWMASK, 0xf, false);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint64_t dpp_mov64(uint64_t old, uint64_t v) {
  const uint32_t lo = dpp_mov<CTRL, ROWMASK>((uint32_t)old, (uint32_t)v);
  const uint32_t hi = dpp_mov<CTRL, ROWMASK>((uint32_t)(old >> 32), (uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// the six steps: pairs, quads, row_ror:4, row_ror:8 (every lane then holds its row's total), row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3 — lane 63 ends with the wave's total. `id` is the operation's identity (what a lane
// without a source contributes).
template <class OP>
__device__ __forceinline__ uint32_t dpp_reduce32(uint32_t v, uint32_t id) {
  v = OP::f(v, dpp_mov<0xb1, 0xf>(id, v));
  v = OP::f(v, dpp_mov<0x4e, 0xf>(id, v));
  v = OP::f(v, dpp_mov<0x124, 0xf>(id, v));
  v = OP::f(v, dpp_mov<0x128, 0xf>(id, v));
  v = OP::f(v, dpp_mov<0x142, 0xa>(id, v));
  v = OP::f(v, dpp_mov<0x143, 0xc>(id, v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_reduce_add32(uint32_t v) { return dpp_reduce32<OpAdd>(v, 0u); }
__device__ __forceinline__ uint64_t wave_reduce_add64(uint64_t v) {
  v = v + dpp_mov64<0xb1, 0xf>(0ull, v);
  v = v + dpp_mov64<0x4e, 0xf>(0ull, v);
  v = v + dpp_mov64<0x124, 0xf>(0ull, v);
  v = v + dpp_mov64<0x128, 0xf>(0ull, v);
  v = v + dpp_mov64<0x142, 0xa>(0ull, v);
  v = v + dpp_mov64<0x143, 0xc>(0ull, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int32_t wave_reduce_min_i32(int32_t v) {  // (order-preserving map to unsigned)
  return (int32_t)(dpp_reduce32<OpMin>((uint32_t)v ^ 0x80000000u, 0xFFFFFFFFu) ^ 0x80000000u);
}
__device__ __forceinline__ int32_t wave_reduce_max_i32(int32_t v) {
  return (int32_t)(dpp_reduce32<OpMax>((uint32_t)v ^ 0x80000000u, 0u) ^ 0x80000000u);
}
#else
__device__ __forceinline__ uint32_t wave_reduce_add32(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint64_t wave_reduce_add64(uint64_t v) {
  for (int o = 32; o > 0; o >>= 1) v += shfl64(v, WG_LANE ^ o);
  return v;
}
__device__ __forceinline__ int32_t wave_reduce_min_i32(int32_t v) {
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int32_t wave_reduce_max_i32(int32_t v) {
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
#endif
__device__ __forceinline__ int wave_sum(int v) { return (int)wave_reduce_add32((uint32_t)v); }
// the value lane `src` holds, src wave-uniform: v_readlane (an SGPR) instead of a ds_bpermute broadcast
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int src) { return WG_READLANE(v, src); }
__device__ __forceinline__ uint64_t lane_bcast64(uint64_t v, int src) {
  return ((uint64_t)WG_READLANE((uint32_t)(v >> 32), src) << 32) | WG_READLANE((uint32_t)v, src);
}
// inclusive prefix sums over the 64 lanes: Hillis-Steele inside each row of 16 with row_shr (zero fill), then the row
// totals carried over with row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3)
#if defined(__HIPCC__) && !defined(WG_NO_DPP)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_mov0(uint32_t v) {  // lanes without a source read 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan32(uint32_t v) {
  v += dpp_mov0<0x111, 0xf>(v);
  v += dpp_mov0<0x112, 0xf>(v);
  v += dpp_mov0<0x114, 0xf>(v);
  v += dpp_mov0<0x118, 0xf>(v);
  v += dpp_mov<0x142, 0xa>(0u, v);
  v += dpp_mov<0x143, 0xc>(0u, v);
  return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
  auto mv0 = [](uint64_t x, auto tag) {
    constexpr int C = decltype(tag)::value;
    return ((uint64_t)dpp_mov0<C, 0xf>((uint32_t)(x >> 32)) << 32) | dpp_mov0<C, 0xf>((uint32_t)x);
  };
  v += mv0(v, std::integral_constant<int, 0x111>());
  v += mv0(v, std::integral_constant<int, 0x112>());
  v += mv0(v, std::integral_constant<int, 0x114>());
  v += mv0(v, std::integral_constant<int, 0x118>());
  v += dpp_mov64<0x142, 0xa>(0ull, v);
  v += dpp_mov64<0x143, 0xc>(0ull, v);
  return v;
}
#else
__device__ __forceinline__ uint32_t wave_incl_scan32(uint32_t v) {
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t u = __shfl_up(v, o, 64);
    if ((int)WG_LANE >= o) v += u;
  }
  return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
  for (int o = 1; o < 64; o <<= 1) {
    uint64_t u = shfl_up64(v, o);
    if ((int)WG_LANE >= o) v += u;
  }
  return v;
}
#endif
__device__ __forceinline__ uint64_t lanes_lt() { return (1ULL << WG_LANE) - 1ULL; }

__device__ __forceinline__ Rec* rec_ptr(const EngineDev& d, uint32_t bucket, uint32_t i) {
  uint32_t page = d.pagetab[(size_t)bucket * d.maxPagesPerBucket + (i >> PAGE_SHIFT)];
  return d.pool + (((size_t)page << PAGE_SHIFT) | (i & (PAGE_RECS - 1)));
}

// Entry base + k of a ring of `cap` entries whose base is already reduced and whose step is far below cap: a compare and
// a subtract. (A 64-bit % by a run-time value is emulated — ~ 100 instructions; a 10-destination list paid thirty of them
// in k_resolve, every chain hop one in expand. The second test keeps the function exact for any base.)
__device__ __forceinline__ unsigned long long ring_at(unsigned long long base, unsigned long long k, unsigned long long cap) {
  unsigned long long i = base + k;
  if (i >= cap) {
    i -= cap;
    if (i >= cap) i %= cap;
  }
  return i;
}

__device__ __forceinline__ int32_t dev_latency(const EngineDev& d, int32_t from, int32_t to, int32_t seed) {
  const NodeArrays& n = d.nodes;
  return latency_of(d.lat, from, to, n.x[from], n.y[from], n.extraLatency[from], n.x[to], n.y[to],
                    n.extraLatency[to], pseudo_delta(to, seed));
}

// arrival of the j-th destination of a chain (MultipleDestEnvelope.arrivalTime C/Envelope.java:107-113,
// MultipleDestWithDelayEnvelope.nextArrivalTime :186-188)
// An envelope created by k_resolve keeps each destination's LATENCY in the upper half of its destination word
// (Chain::flags & CHAIN_LAT: id | latency << 16 — engines of <= 65 536 nodes whose destination words carry no protocol tag):
// arrival j = sendTime + latency j is what MultipleDestEnvelope.nextArrivalTime recomputes at every hop (C/Envelope.java:
// 107-113) from both ends' positions and the jitter table — four dependent scattered loads per hop, three times per hop
// (the expand scan's run length, its write, the re-push's arrival in k_resolve), which was 136 bytes fetched per delivered
// message in GSFSignature's k_scan1<ExpandF>. The value stored is the one createMessageArrivals sorted by (:449-467).
constexpr uint32_t CHAIN_LAT = 4u;
__device__ __forceinline__ int32_t chain_dest_word(const EngineDev& d, const Chain& c, int j) {  // (id | tag << 16: EngineDev::destTagged; the id alone otherwise)
  const int32_t w = d.dests[ring_at(c.destOff, (unsigned long long)j, d.chainDests)];
  return (c.flags & CHAIN_LAT) ? (int32_t)((uint32_t)w & 0xFFFFu) : w;
}
__device__ __forceinline__ int32_t chain_dest(const EngineDev& d, const Chain& c, int j) { return dest_id(d, chain_dest_word(d, c, j)); }
__device__ __forceinline__ int32_t chain_arrival(const EngineDev& d, const Chain& c, int j) {
  if (c.flags & CHAIN_LAT) return c.sendTime + (int32_t)((uint32_t)d.dests[ring_at(c.destOff, (unsigned long long)j, d.chainDests)] >> 16);
  if (c.flags & 2u) return d.dests[ring_at(c.destOff, (unsigned long long)c.ndest + j, d.chainDests)];
  return c.sendTime + dev_latency(d, c.from, chain_dest(d, c, j), c.seed);
}

// ------------------------------------------------------------------------------------------------
// Device-wide exclusive scan of F(i), i in [0, F.count()): every block owns a contiguous chunk. Values are uint64 so
// a pair of 32-bit counters can be scanned at once.
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_GRID = 240 / WG_GRID_DIV > 2 ? 240 / WG_GRID_DIV : 2;  // blocks per ENGINE (grid.x; a batch multiplies it by its members in grid.y): a multiple of
                                               // the 8 XCDs that keeps a block's chunk of a typical ms's events at a few SCAN_BLOCK rounds

__device__ __forceinline__ void scan_range(uint32_t n, uint32_t bx, uint32_t gx, uint32_t& lo, uint32_t& hi) {  // (bx of gx blocks: WG_ENGINE's)
  uint32_t chunk = (n + gx - 1) / gx;
  chunk = (chunk + SCAN_BLOCK - 1) / SCAN_BLOCK * SCAN_BLOCK;
  lo = min(n, bx * chunk);
  hi = min(n, lo + chunk);
}

__device__ __forceinline__ uint64_t block_sum64(uint64_t v, uint64_t* sh) {
  v = wave_reduce_add64(v);
  int w = threadIdx.x >> 6;
  if (WG_LANE == 0) sh[w] = v;
  __syncthreads();
  uint64_t t = 0;
  for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += sh[k];
  __syncthreads();
  return t;
}

// exclusive scan of one uint32 per thread over a block of up to 1024 threads (sh16: 16 words); *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan32_1024(uint32_t v, uint32_t* sh16, uint32_t* total) {
  const uint32_t incl = wave_incl_scan32(v);
  int w = threadIdx.x >> 6;
  if (WG_LANE == 63) sh16[w] = incl;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
  for (int k = 0; k < (int)(blockDim.x >> 6); k++) {
    uint32_t x = sh16[k];
    if (k < w) woff += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return woff + incl - v;
}

// the "some node still ..." flag of a continuation predicate (newContIf): every wavefront that finds such a node used to
// atomicOr the same word — 12 k same-address atomics per call at 24 copies of 32 768 nodes, 100 us of every tenth ms
// (one L2 atomic unit retires ~ 88 of them per us). A plain store of the same value by whoever finds it first is enough.
__device__ __forceinline__ bool cont_if_known(const uint32_t* out) {
  return __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
__device__ __forceinline__ void cont_if_set(uint32_t* out) {
  if (!cont_if_known(out)) __hip_atomic_store(out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every kernel of the pipeline is launched over a TABLE of engines — grid (blocks per engine, engines) —:
// a batch of independent simulations (the reference's RunMultipleTimes copies, C/RunMultipleTimes.java:
// 44-48) advances one simulated ms per launch. A stand-alone engine is a table of one.
//
// WHICH block works on which engine is XCD-aware. The dispatcher hands workgroup L (linear id, x fastest) to XCD L % 8,
// each XCD has a private 4 MB L2 (not coherent with the others) and its own TLBs, and an engine is a closed world: its
// kernels read and write that engine's arrays only. With the plain (blockIdx.x, blockIdx.y) mapping an engine's blocks
// are dealt round all eight XCDs, so what one kernel of a ms wrote (inbox lines, work lists, queue records, event results)
// is in another XCD's L2 seven times out of eight when the next kernel reads it, and every XCD's TLBs walk all the batch's
// 280 GB. wg_place re-deals the launch's blocks so that engine e is worked on by XCD e % 8 only: the blocks that land on
// XCD k (L % 8 == k) are split evenly over the engines k, k + 8, k + 16, ... — wgBy is the engine, wgBx / wgGx the block's
// index / count WITHIN that engine (every kernel here is a grid-stride loop over wgGx blocks, so results never depend on
// the split). A block left over by the division does nothing. Placement is a speed matter only: nothing relies on the
// dispatcher really behaving this way (MI355X_MICROARCH.md: observed, not contracted). Batches of fewer than 8 engines,
// and grids of one block per engine, keep the plain mapping (EngineDev::xcdPlace == 0 forces it: WG_XCD_PLACE=0).
struct WgPos {
  uint32_t bx, by, gx;
};
__device__ __forceinline__ bool wg_place(const EngineDev* __restrict__ tab, WgPos& w) {
  const uint32_t gx = gridDim.x, R = gridDim.y;
  w.bx = blockIdx.x;
  w.by = blockIdx.y;
  w.gx = gx;
  if (R < 8u || gx < 2u || !tab[0].xcdPlace) return true;
  const uint32_t L = blockIdx.x + gx * blockIdx.y, T = gx * R;
  const uint32_t xcd = L & 7u, j = L >> 3;
  const uint32_t blocksHere = (T - xcd + 7u) >> 3;    // blocks of this launch with L % 8 == xcd
  const uint32_t enginesHere = (R - xcd + 7u) >> 3;   // engines xcd, xcd + 8, ...
  uint32_t per = blocksHere / enginesHere;            // blocks each of them gets (>= 1: T >= 2 R)
  if (per > gx) per = gx;                             // (never more than the launch's own width: per-block scratch is sized by it)
  const uint32_t ei = j / per;
  if (ei >= enginesHere) return false;                // a left-over block
  w.by = xcd + 8u * ei;
  w.bx = j - ei * per;
  w.gx = per;
  return true;
}
#define WG_ENGINE(tab)                                                               \
  WgPos wgPos_;                                                                      \
  if (!wg_place((tab), wgPos_)) return;                                              \
  [[maybe_unused]] const uint32_t wgBx = wgPos_.bx, wgBy = wgPos_.by, wgGx = wgPos_.gx; \
  const EngineDev& d = (tab)[wgBy];                                                  \
  if (d.halted) return

// k_scan2 re-evaluates F::value(i); a functor whose value is expensive may take a `first` flag and reuse what the
// first evaluation (k_scan1) left behind (ExpandF: chain run lengths)
template <class F>
__device__ __forceinline__ auto scan_value_again(const F& f, uint32_t i) -> decltype(f.value(i, false)) {
  return f.value(i, false);
}
template <class F, class... X>
__device__ __forceinline__ uint64_t scan_value_again(const F& f, uint32_t i, X...) {
  return f.value(i);
}

// Two launches with no inter-block communication: (1) per-block sums of a contiguous chunk, (2) every block re-sums the
// sums before it, scans its chunk and calls F.write(i, exclusive_prefix). (Measured against ONE launch whose blocks
// wait for their predecessors' sums through device-scope atomics — profiles/r07d_*: twice the time. Launches on one
// stream run back to back on this chip, a kernel boundary costs nothing by itself; a block that waits holds its CU.)
template <class F>
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan1(const EngineDev* __restrict__ tab, const typename F::Aux* atab) {
  WG_ENGINE(tab);
  const F f(d, atab ? atab + wgBy : nullptr);
  unsigned long long WG_G* partials = d.scanPartials;
  __shared__ uint64_t sh[SCAN_BLOCK / 64];
  uint32_t n = f.count(), lo, hi;
  scan_range(n, wgBx, wgGx, lo, hi);
  uint64_t acc = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += SCAN_BLOCK) acc += f.value(i);
  uint64_t tot = block_sum64(acc, sh);
  if (threadIdx.x == 0) partials[wgBx] = tot;
  f.tally(lo, hi);
}

// f.write(i, exclusive_prefix, valid) is called by every thread of the block at the same point (valid
// = i is in range), so write() may use wave-level collectives.
template <class F>
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan2(const EngineDev* __restrict__ tab, const typename F::Aux* atab) {
  WG_ENGINE(tab);
  const F f(d, atab ? atab + wgBy : nullptr);
  const unsigned long long WG_G* partials = d.scanPartials;
  __shared__ uint64_t sh[SCAN_BLOCK / 64];
  __shared__ uint64_t shw[SCAN_BLOCK / 64];
  uint32_t n = f.count(), lo, hi;
  scan_range(n, wgBx, wgGx, lo, hi);
  uint64_t before = 0, all = 0;
  for (uint32_t b = threadIdx.x; b < wgGx; b += SCAN_BLOCK) {
    uint64_t p = partials[b];
    all += p;
    if (b < wgBx) before += p;
  }
  uint64_t prefix = block_sum64(before, sh);
  uint64_t total = block_sum64(all, sh);
  if (wgBx == 0 && threadIdx.x == 0) f.total(total);
  int w = threadIdx.x >> 6;
  for (uint32_t base = lo; base < hi; base += SCAN_BLOCK) {
    uint32_t i = base + threadIdx.x;
    uint64_t v = i < hi ? scan_value_again(f, i) : 0;
    uint64_t incl = wave_incl_scan64(v);
    if (WG_LANE == 63) shw[w] = incl;
    __syncthreads();
    uint64_t woff = 0, tile = 0;
    for (int k = 0; k < SCAN_BLOCK / 64; k++) {
      uint64_t x = shw[k];
      if (k < w) woff += x;
      tile += x;
    }
    f.write(i, prefix + woff + incl - v, i < hi);
    prefix += tile;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// expand: bucket t, LIFO, chain runs unrolled. Pair scan: low = events, high = outbox slots.
struct ExpandF {
  typedef int Aux;
  const EngineDev& d;
  __device__ ExpandF(const EngineDev& d_, const Aux*) : d(d_) {}
  __device__ int32_t now() const { return d.g->now; }
  __device__ uint32_t bucket() const { return (uint32_t)now() & (uint32_t)(d.horizon - 1); }
  __device__ uint32_t count() const { return d.bcnt[bucket()]; }
  __device__ uint32_t runlen(const Rec& r, int32_t t) const {
    if (rec_kind(r) != K_CHAIN) return 1;
    const Chain c = d.chains[r.w1];
    // The destinations of an envelope are sorted by arrival when it is created (C/Network.java:464), so arrivals
    // never decrease along the chain and the run of hops arriving in t, which starts at r.w2, ends before the first
    // hop arriving later: gallop to bracket that hop, then bisect. One probe for the usual run of one hop (as the
    // plain walk), ~2 log2(len) for the N/300-hop runs of a sendAll.
    const int p = (int)r.w2, n = c.ndest;
    int lo = p + 1, hi = n, step = 1;  // arrival(j) == t for p <= j < lo; arrival(hi) > t or hi == n
    while (lo < n) {
      const int probe = min(lo + step - 1, n - 1);
      if (chain_arrival(d, c, probe) == t) {
        lo = probe + 1;
        step <<= 1;
      } else {
        hi = probe;
        break;
      }
    }
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (chain_arrival(d, c, mid) == t)
        lo = mid + 1;
      else
        hi = mid;
    }
    return (uint32_t)(lo - p);
  }
  __device__ uint32_t task_bound(const Rec& r) const {
    return d.boundTask[r.w2 < 3u ? r.w2 : 3u] + (rec_kind(r) == K_PERIODIC ? 1u : 0u);
  }
  // The run length of chain record i is needed three times (the chunk sum, the scan and its write): the
  // first evaluation parks it in recEv[i] — scratch of the later `order` phase, idle during expand — for the other two.
  __device__ uint32_t runlen_first(uint32_t i, const Rec& r) const {
    const uint32_t len = runlen(r, now());
    if (i < d.maxOut) d.recEv[i] = len;
    return len;
  }
  __device__ uint32_t runlen_again(uint32_t i, const Rec& r) const { return i < d.maxOut ? d.recEv[i] : runlen(r, now()); }
  __device__ uint64_t value(uint32_t i, bool first = true) const {
    const Rec r = *rec_ptr(d, bucket(), count() - 1 - i);
    const uint32_t k = rec_kind(r);
    if (k == K_MSG) return ((uint64_t)d.boundMsg << 32) | 1u;
    if (k != K_CHAIN) return ((uint64_t)task_bound(r) << 32) | 1u;
    const uint32_t len = first ? runlen_first(i, r) : runlen_again(i, r);
    return ((uint64_t)(len * d.boundMsg + 1u) << 32) | len;  // + the re-push after the run (:629-632)
  }
  __device__ void tally(uint32_t, uint32_t) const {}
  __device__ void total(uint64_t tot) const {
    uint32_t ne = (uint32_t)tot, ns = (uint32_t)(tot >> 32);
    if (ne > d.maxEvents) {
      set_err(d.g, ERR_EVENTS);
      ne = 0;
    }
    if (ns > d.maxOut) {  // no delivery kernel may see outbox slices beyond the outbox: the ms is dropped, loudly
      set_err(d.g, ERR_OUTBOX);
      ne = 0;
    }
    d.g->nEvents = ne;
    d.g->outSlots = ns;
  }
  // A protocol that delivers one kind of message one lane per EVENT (Casper's attestations, k_casper_attestations) has no
  // use for those events' inbox lists: they are not threaded at expand (EngineDev::laneMsgPlus1) — one L2 atomic and one
  // store less per event, and a node that only receives such messages never becomes a node visit. The protocol's kernel
  // threads the few that fall to a node k_deliver visits after all (a node with another kind of event in the same ms).
  __device__ bool lane_only(uint32_t msgWord) const { return d.laneMsgPlus1 != 0 && msgWord + 1u == d.laneMsgPlus1; }
  // thread the event onto its node's inbox list; returns true if it is the node's first event
  // `inLine`: the event's record went into the node's inbox line (nothing will look for it in ev[])
  __device__ bool link(uint32_t e, int32_t to, const Rec& r, bool chainHop, uint32_t ob, bool& inLine) const {
    inLine = false;
    if (d.sharded) {  // the event list is replicated on every shard; a shard applies the events of its own nodes
      d.evRes[e] = EvRes{0u, 0u};  // ... and reports zeros for the others (summed across shards before `order`)
      if (!shard_owns(d, to)) return false;
    }
    if (d.inbox) {  // the node's inbox line: the event itself lands in the node's 64 bytes, no list to chase at delivery
      const uint32_t k = atomicAdd(&d.icnt[to], 1u);
      if (k < (uint32_t)INBOX_SLOTS) {
        inLine = true;
        // (a task's `from` is the node itself: its w0 carries the event's first outbox slot instead, so that a visit
        // needs no EvAux read for it; a message's action() gets its slice from EvAux only when boundMsg > 0 or it is a hop)
        InboxEntry ie;
        ie.e = e;
        ie.w0 = rec_kind(r) == K_MSG ? (r.w0 | (chainHop ? INBOX_CHAIN : 0u)) : ((r.w0 & 0xF0000000u) | ob);
        ie.w2 = r.w2;
        ie.w3 = r.w3;
        gst(d.inbox + ((size_t)to * INBOX_SLOTS + k), ie);
      } else {  // beyond the line: the overflow list
        d.evNext[e] = atomicExch(&d.head[to], (int32_t)e);
      }
      return k == 0;
    }
    int32_t prev = atomicExch(&d.head[to], (int32_t)e);
    d.evNext[e] = prev;
    return prev < 0;
  }
  __device__ void write(uint32_t i, uint64_t excl, bool valid) const {
    bool first = false;
    int32_t firstNode = 0;
    if (valid) {
      const Rec r = *rec_ptr(d, bucket(), count() - 1 - i);
      uint32_t e = (uint32_t)excl, ob = (uint32_t)(excl >> 32);
      const uint32_t k = rec_kind(r);
      if (k != K_CHAIN) {
        if (e < d.maxEvents) {
          bool inLine = false;
          if (!d.hostMode && !(k == K_MSG && lane_only(r.w2))) first = link(e, (int32_t)r.w1, r, false, ob, inLine);
          // The event's 16-byte record + its 16 bytes of side data are what a visit reads when it walks a LIST. An event in its
          // node's inbox line is read from the line; what is still looked up by event index is the outbox slice of an event
          // that can emit records (k_resolve: a task's; a message's where the protocol's action() sends, boundMsg > 0) and, on
          // a sharded engine, the receiver (k_resolve<true>, the snapshot scans). A plain message in a line of an unsharded
          // engine whose deliveries emit nothing — nine events in ten of Handel and GSFSignature — writes neither.
          if (!(inLine && k == K_MSG && d.boundMsg == 0 && !d.sharded)) {
            d.ev[e] = r;
            EvAux a;
            a.chain = -1;
            a.cpos = 0;
            a.outBase = ob;
            a.outCap = k == K_MSG ? d.boundMsg : task_bound(r);
            d.evAux[e] = a;
          }
          firstNode = (int32_t)r.w1;
        }
      } else {
        const Chain c = d.chains[r.w1];
        const uint32_t len = runlen_again(i, r);
        if (d.runMin && len >= d.runMin) {  // a long run: one wavefront unrolls it (k_expand_runs)
          const uint32_t k = atomicAdd(&d.g->nRuns, 1u);
          if (k < d.maxRuns) {
            RunDesc rd;
            rd.chain = r.w1;
            rd.pos = r.w2;
            rd.e = e;
            rd.ob = ob;
            rd.len = len;
            rd.pad0 = rd.pad1 = rd.pad2 = 0;
            d.runs[k] = rd;
          } else {
            set_err(d.g, ERR_EVENTS);
          }
        } else
        for (uint32_t q = 0; q < len; q++, e++) {
          if (e >= d.maxEvents) break;
          const int32_t tw = chain_dest_word(d, c, (int)r.w2 + (int)q), to = dest_id(d, tw);
          const Rec hop = make_rec(K_MSG, c.from, (uint32_t)to, dest_msg(d, c.msg, tw), c.payload);
          const bool last = q + 1 == len;
          EvAux a;
          a.chain = (int32_t)r.w1;
          a.cpos = (int32_t)(r.w2 + q) | (last ? (int32_t)0x80000000 : 0);
          a.outBase = ob + q * d.boundMsg;
          a.outCap = d.boundMsg + (last ? 1u : 0u);
          d.evAux[e] = a;  // (a hop's side data is read by its visit: the envelope, its position, the re-push's slot)
          bool inLine = false;
          if (!d.hostMode && !lane_only(c.msg) && link(e, to, hop, true, ob + q * d.boundMsg, inLine)) d.active[atomicAdd(&d.g->nActive, 1u)] = (uint32_t)to;  // chains are rare
          if (!inLine || d.sharded) d.ev[e] = hop;
        }
        // sharded: the envelope's slot is released by every shard (deliver_event's release runs on one shard only)
        if (d.sharded && (int)(r.w2 + len) >= c.ndest) d.chains[r.w1].flags = 0;
      }
    }
    // wave-aggregated append to the active list: one atomic per wavefront, not per node
    const uint64_t m = __ballot(first);
    if (m) {
      uint32_t base = 0;
      const int leader = __ffsll((unsigned long long)m) - 1;
      if ((int)WG_LANE == leader) base = atomicAdd(&d.g->nActive, (uint32_t)__popcll(m));
      base = lane_bcast(base, leader);
      if (first) d.active[base + __popcll(m & lanes_lt())] = (uint32_t)firstNode;
    }
  }
};

// The long chain runs ExpandF::write listed (EngineDev::runs): one wavefront per run, a lane per hop — destination ids
// read and events written coalesced, the inbox links and the active-list append as in ExpandF::write. Event indices
// and outbox slices are the scan's, so the result is what the in-place unrolling would have written.
__global__ void __launch_bounds__(256) k_expand_runs(const EngineDev* __restrict__ tab) {
  WG_ENGINE(tab);
  const uint32_t nRuns = min(d.g->nRuns, d.maxRuns);
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t lane = WG_LANE;
  const ExpandF f(d, nullptr);
  for (uint32_t k = wave; k < nRuns; k += nWaves) {
    const RunDesc rd = d.runs[k];
    const Chain c = d.chains[rd.chain];
    for (uint32_t q0 = 0; q0 < rd.len; q0 += 64) {
      const uint32_t q = q0 + lane, e = rd.e + q;
      bool first = false;
      int32_t to = 0;
      if (q < rd.len && e < d.maxEvents) {
        const int32_t tw = chain_dest_word(d, c, (int)(rd.pos + q));
        to = dest_id(d, tw);
        const Rec hop = make_rec(K_MSG, c.from, (uint32_t)to, dest_msg(d, c.msg, tw), c.payload);
        const bool last = q + 1 == rd.len;
        EvAux a;
        a.chain = (int32_t)rd.chain;
        a.cpos = (int32_t)(rd.pos + q) | (last ? (int32_t)0x80000000 : 0);
        a.outBase = rd.ob + q * d.boundMsg;
        a.outCap = d.boundMsg + (last ? 1u : 0u);
        d.evAux[e] = a;
        bool inLine = false;
        if (!d.hostMode && !f.lane_only(c.msg)) first = f.link(e, to, hop, true, rd.ob + q * d.boundMsg, inLine);
        if (!inLine || d.sharded) d.ev[e] = hop;  // (as ExpandF::write: an event in its node's line is read from the line)
      }
      const uint64_t m = __ballot(first);
      if (m) {
        uint32_t base = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if ((int)lane == leader) base = atomicAdd(&d.g->nActive, (uint32_t)__popcll(m));
        base = lane_bcast(base, leader);
        if (first) d.active[base + __popcll(m & lanes_lt())] = (uint32_t)to;
      }
    }
    // sharded: the envelope's slot is released by every shard (as ExpandF::write does for the runs it unrolls in place)
    if (d.sharded && lane == 0 && (int)(rd.pos + rd.len) >= c.ndest) d.chains[rd.chain].flags = 0;
  }
}

// order: per-event (records, draws) -> offsets in the global push order / draw order, the event of
// every ordered outbox position, and the run statistics (block-aggregated).
struct RecsF {
  typedef int Aux;
  const EngineDev& d;
  __device__ RecsF(const EngineDev& d_, const Aux*) : d(d_) {}
  __device__ uint32_t count() const { return d.g->nEvents; }
  __device__ uint64_t value(uint32_t i) const {
    const EvRes r = d.evRes[i];
    return ((uint64_t)r.ndraw << 32) | (r.nrec & EV_NREC_MASK);
  }
  __device__ void tally(uint32_t lo, uint32_t hi) const {
    __shared__ uint32_t shStat[35];  // [0..31] delivered by level, [32] delivered, [33] tasks, [34] snapshot rows (sharded)
    for (int k = threadIdx.x; k < 35; k += SCAN_BLOCK) shStat[k] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += SCAN_BLOCK) {
      const uint32_t f = d.evRes[i].nrec;
      if (f & EV_DELIVERED) {
        atomicAdd(&shStat[(f >> 24) & 31u], 1u);
        atomicAdd(&shStat[32], 1u);
      }
      if (f & EV_TASK_RUN) atomicAdd(&shStat[33], 1u);
      if (f & EV_SNAP_MASK) atomicAdd(&shStat[34], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 34 && shStat[threadIdx.x]) {
      unsigned long long* dst = threadIdx.x < 32 ? &d.g->deliveredByLevel[threadIdx.x]
                                                 : (threadIdx.x == 32 ? &d.g->delivered : &d.g->tasks);
      atomicAdd(dst, (unsigned long long)shStat[threadIdx.x]);
    }
    if (threadIdx.x == 34 && shStat[34]) atomicAdd(&d.g->nSnapEv, shStat[34]);
  }
  __device__ void total(uint64_t tot) const {
    uint32_t n = (uint32_t)tot;
    if (n > d.maxOut) {
      set_err(d.g, ERR_OUTBOX);
      n = 0;
    }
    d.g->nOut = n;
    d.g->nDraws = (uint32_t)(tot >> 32);
  }
  __device__ void write(uint32_t i, uint64_t excl, bool valid) const {
    if (!valid) return;
    const uint32_t off = (uint32_t)excl, n = d.evRes[i].nrec & EV_NREC_MASK;
    d.evRecOff[i] = off;
    d.evDrawOff[i] = (uint32_t)(excl >> 32);
    for (uint32_t k = 0; k < n && off + k < d.maxOut; k++) d.recEv[off + k] = i;
  }
};

// exchange 1 of a sharded engine: EvRes (8 bytes) <-> one packed int32 per event in the exchange image
template <bool PACK>
__global__ void __launch_bounds__(256) k_shard_evres(const EngineDev* __restrict__ tab) {
  WG_ENGINE(tab);
  const uint32_t n = d.g->nEvents;
  if (PACK && wgBx == 0 && threadIdx.x == 0) d.g->nSnapEv = 0;  // (counted afresh by this ms's order scan)
  for (uint32_t e = wgBx * blockDim.x + threadIdx.x; e < n; e += wgGx * blockDim.x) {
    if (PACK) {
      const EvRes r = d.evRes[e];
      if (!evres_packable(r)) set_err(d.g, ERR_SHARD_EVENT);
      d.xev[e] = (int32_t)evres_pack(r);
    } else {
      d.evRes[e] = evres_unpack((uint32_t)d.xev[e]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// resolve: unordered outbox -> ordered outbox (fin, arr). One thread per record.
// Network.send / createMessageArrival(s) (C/Network.java:369-382,418-487).
__device__ __forceinline__ int32_t draw_next_int(const EngineDev& d, uint32_t drawIdx) {
  uint64_t s = lcg_skip(d.g->rng, (uint64_t)drawIdx + 1);
  return (int32_t)(int64_t)(s >> 16);
}

// a node's NodeGeo record as one 16-byte load
struct GeoRegs {
  int32_t x, y, extra;
  uint32_t down, part;
};
__device__ __forceinline__ GeoRegs node_geo(const EngineDev& d, int32_t node) {
  const U4 q = gld((const U4 WG_G*)(const NodeGeo WG_G*)d.nodes.geo + node);
  GeoRegs g;
  g.x = (int32_t)(int16_t)(q.x & 0xFFFFu);
  g.y = (int32_t)(int16_t)(q.x >> 16);
  g.extra = (int32_t)q.y;
  g.down = q.z & 0xFFu;
  g.part = (q.z >> 8) & 0xFFu;
  return g;
}
__device__ __forceinline__ bool arrival_of_send(const EngineDev& d, int32_t from, int32_t to, int32_t sendTime,
                                               int32_t seed, int32_t& arrival) {
  const GeoRegs f = node_geo(d, from), t = node_geo(d, to);  // (both ends in one round of loads)
  if (f.part != t.part || f.down || t.down) return false;
  int32_t nt = latency_of(d.lat, from, to, f.x, f.y, f.extra, t.x, t.y, t.extra, pseudo_delta(to, seed));
  if (nt >= d.discardTime) return false;
  arrival = sendTime + nt;
  return true;
}

constexpr int TILE = 1024;

// createMessageArrivals of a device-side multi-destination send (delaysBetweenMessage == 0): arrivals of the
// destinations that are reachable, stable-sorted by arrival (C/Network.java:449-467). Returns their number.
// IN PLACE: the unsorted list lies in the scratch destination ring at o.destOff (EngineDev::sdests — on an unsharded
// engine the chain ring itself); the sorted destinations end up in the same entries [0, m), their arrivals in the same
// entries of the parallel ring EngineDev::arvTmp. Entry j is read before anything at or beyond j is written (the sorted
// prefix is never longer than the entries consumed), so no copy of the list is needed — and no per-lane array, which the
// compiler could only have kept in scratch memory (528 bytes a lane until round 3).
__device__ __forceinline__ unsigned long long multi_idx(const EngineDev& d, const Out& o, int k) {
  return ring_at(o.destOff, (unsigned long long)k, d.sdestCap);
}
__device__ __forceinline__ int resolve_multi(const EngineDev& d, const Out& o, int32_t from, int32_t seed) {
  const int nd = o.to;
  const int32_t step = (o.pad & OUT_DELAYED) ? (int32_t)(o.pad >> 8) + 1 : 0;  // sendTime += delay + 1 per destination (:459)
  int m = 0;
  for (int j = 0; j < nd && j < 64; j++) {
    const int32_t to = d.sdests[multi_idx(d, o, j)];  // (the whole word moves: id | tag << 16 on a tagged engine)
    int32_t a;
    if (!arrival_of_send(d, from, dest_id(d, to), o.t + j * step, seed, a)) continue;
    int k = m++;
    while (k > 0 && d.arvTmp[multi_idx(d, o, k - 1)] > a) {  // insertion keeps equal arrivals in caller order
      d.arvTmp[multi_idx(d, o, k)] = d.arvTmp[multi_idx(d, o, k - 1)];
      d.sdests[multi_idx(d, o, k)] = d.sdests[multi_idx(d, o, k - 1)];
      k--;
    }
    d.arvTmp[multi_idx(d, o, k)] = a;
    d.sdests[multi_idx(d, o, k)] = to;
  }
  return m;
}
// Lists of up to MULTI_LDS destinations (Handel's fast path and GSFSignature's accelerated calls send 10) are sorted in LDS
// instead — entry k of thread t at sh[k][t]: no bank conflicts, no memory round trip per insertion step — and written to
// the same entries of the two rings at the end; longer lists take the in-place form above.
constexpr int MULTI_LDS = 10;  // (20 KB a block: eight blocks a CU still fit)
__device__ __forceinline__ int resolve_multi_lds(const EngineDev& d, const Out& o, int32_t from, int32_t seed, int32_t* shDst,
                                                 int32_t* shArv, bool packLat) {
  const int nd = o.to, T = (int)blockDim.x, t = (int)threadIdx.x;
  const int32_t step = (o.pad & OUT_DELAYED) ? (int32_t)(o.pad >> 8) + 1 : 0;
  // The list's arrivals in three ROUNDS of independent loads — every destination, then every destination's flags and
  // coordinates, then every latency-table entry — instead of one chain of three dependent round trips per destination
  // (ten destinations: thirty round trips one behind the other; the kernel is one wave-round of lanes, so its duration
  // IS its longest chain). Static indices throughout: the arrays live in registers.
  int32_t to_[MULTI_LDS], tw_[MULTI_LDS], arv[MULTI_LDS];  // (tw_: the destination words as they travel, to_: the ids)
  bool ok[MULTI_LDS];
#pragma unroll
  for (int j = 0; j < MULTI_LDS; j++) {
    tw_[j] = j < nd ? d.sdests[multi_idx(d, o, j)] : from;
    to_[j] = j < nd ? dest_id(d, tw_[j]) : from;
  }
  const GeoRegs gf = node_geo(d, from);
  const uint32_t pf = gf.part, df = gf.down;
  const int32_t xf = gf.x, yf = gf.y, ef = gf.extra;
  uint32_t pt[MULTI_LDS], dt[MULTI_LDS];
  int32_t xt[MULTI_LDS], yt[MULTI_LDS], et[MULTI_LDS];
#pragma unroll
  for (int j = 0; j < MULTI_LDS; j++) {
    const GeoRegs g = node_geo(d, to_[j]);
    pt[j] = g.part;
    dt[j] = g.down;
    xt[j] = g.x;
    yt[j] = g.y;
    et[j] = g.extra;
  }
#pragma unroll
  for (int j = 0; j < MULTI_LDS; j++) {  // arrival_of_send, with the loads above
    ok[j] = j < nd && pf == pt[j] && !df && !dt[j];
    arv[j] = latency_of(d.lat, from, to_[j], xf, yf, ef, xt[j], yt[j], et[j], pseudo_delta(to_[j], seed));
  }
  int m = 0;
#pragma unroll
  for (int j = 0; j < MULTI_LDS; j++) {
    if (!ok[j] || arv[j] >= d.discardTime) continue;
    const int32_t to = tw_[j];
    const int32_t a = o.t + j * step + arv[j];
    int k = m++;
    while (k > 0 && shArv[(k - 1) * T + t] > a) {
      shArv[k * T + t] = shArv[(k - 1) * T + t];
      shDst[k * T + t] = shDst[(k - 1) * T + t];
      k--;
    }
    shArv[k * T + t] = a;
    shDst[k * T + t] = to;
  }
  // (packLat: the envelope's destination words keep the latency beside the id — CHAIN_LAT; one destination is a plain send)
  for (int k = 0; k < m; k++) {
    const int32_t a = shArv[k * T + t];
    d.sdests[multi_idx(d, o, k)] = packLat && m > 1 ? (int32_t)((uint32_t)shDst[k * T + t] | ((uint32_t)(a - o.t) << 16)) : shDst[k * T + t];
    d.arvTmp[multi_idx(d, o, k)] = a;
  }
  return m;
}
// ... only how many are reachable and the first arrival (a shard's k_resolve: the envelope itself is created from the
// exchanged image, k_shard_multi_fill sorts the list)
__device__ __forceinline__ int count_multi(const EngineDev& d, const Out& o, int32_t from, int32_t seed, int32_t& first) {
  const int nd = o.to;
  const int32_t step = (o.pad & OUT_DELAYED) ? (int32_t)(o.pad >> 8) + 1 : 0;
  int m = 0;
  first = INT32_MAX;
  for (int j = 0; j < nd && j < 64; j++) {
    int32_t a;
    if (!arrival_of_send(d, from, dest_id(d, d.sdests[multi_idx(d, o, j)]), o.t + j * step, seed, a)) continue;
    m++;
    first = a < first ? a : first;
  }
  return m;
}

// A multi-destination send whose action() shuffled the list with the shared rd just before sending
// (Collections.shuffle(newList, rd) — P/SanFerminHelper.java:144, C/messages/FloodMessage.java:52): the permutation
// only decides the order of the destinations, so it is deferred to here, where the rd index of the event's draws is
// known: for (i = n; i > 1; i--) swap(list, i-1, rd.nextInt(i)), in the scratch ring, then the seed draw. Returns the
// draws consumed (n - 1). A nextInt(bound) rejection (bound not a power of two, p < 2^-30 per draw) would shift every
// later draw of the ms: flagged, the run fails loudly rather than diverge.
__device__ __forceinline__ uint32_t shuffle_dests(const EngineDev& d, const Out& o, uint32_t drawIdx) {
  const int n = o.to < 64 ? o.to : 64;
  uint64_t st = lcg_skip(d.g->rng, (uint64_t)drawIdx);
  for (int i = n; i > 1; i--) {
    int consumed;
    const int32_t j = lcg_next_int_bounded(st, i, &consumed);
    if (consumed != 1) set_err(d.g, ERR_PROTOCOL);
    const unsigned long long a = ring_at(o.destOff, (unsigned long long)(i - 1), d.sdestCap);
    const unsigned long long b = ring_at(o.destOff, (unsigned long long)j, d.sdestCap);
    const int32_t x = d.sdests[a];
    d.sdests[a] = d.sdests[b];
    d.sdests[b] = x;
  }
  return n > 1 ? (uint32_t)(n - 1) : 0u;
}

// an arrival beyond the bucket ring: parked for the host (FarRec, Engine::collect_far) if the engine keeps a far buffer and
// the arrival is at least two rings ahead; false = it cannot be held (ERR_HORIZON)
__device__ __forceinline__ bool park_far(const EngineDev& d, int32_t t, uint32_t p, const Rec& fin, int32_t arrival) {
  uint32_t k = 0xFFFFFFFFu;
  if (d.farBuf && arrival - t >= 2 * d.horizon) k = atomicAdd(&d.g->nFar, 1u);
  if (k >= d.farCap) return false;
  FarRec fr;
  fr.ms = t;
  fr.p = p;
  fr.rec = fin;
  fr.arrival = arrival;
  fr.pad = 0;
  d.farBuf[k] = fr;
  return true;
}

// SH (sharded engine, wg_shard_configure): a record is resolved by the shard that owns the node whose action()
// emitted it; the result goes to the exchange image xbuf (zeros for records of other shards), which the host sums
// across shards before k_shard_unpack rebuilds fin / arr / the tile histograms on every shard.
template <bool SH>
__global__ void __launch_bounds__(256) k_resolve(const EngineDev* __restrict__ tab) {
  __shared__ int32_t shMulti[2][MULTI_LDS][256];  // resolve_multi_lds: destinations / arrivals being sorted, [entry][thread]
  WG_ENGINE(tab);
  const int32_t t = d.g->now;
  const uint32_t n = d.g->nOut;
  const uint32_t D = (uint32_t)d.horizon;
  for (uint32_t p = wgBx * blockDim.x + threadIdx.x; p < n; p += wgGx * blockDim.x) {
    const uint32_t e = d.recEv[p];
    if (SH && !shard_owns(d, (int32_t)d.ev[e].w1)) {
      for (int k = 0; k < 5; k++) d.xbuf[(size_t)p * 5 + k] = 0;
      continue;
    }
    const Out o = d.outTmp[d.evAux[e].outBase + (p - d.evRecOff[e])];
    uint32_t kind = o.kindfrom >> 28;
    int32_t from = (int32_t)(o.kindfrom & 0x0FFFFFFFu);
    Rec fin = make_rec(K_MSG, from, 0, 0, 0);
    int32_t arrival = -1;
    switch (kind) {
      case O_SEND: {
        int32_t seed = draw_next_int(d, d.evDrawOff[e] + o.drawsub);
        fin = make_rec(K_MSG, from, (uint32_t)o.to, o.a, o.b);
        if (!arrival_of_send(d, from, o.to, o.t, seed, arrival)) arrival = -1;
        break;
      }
      case O_MULTI: {  // delaysBetweenMessage == 0 only (device actions); stable sort by arrival (:464)
        uint32_t drawIdx = d.evDrawOff[e] + o.drawsub;
        if (o.pad & OUT_SHUFFLE) drawIdx += shuffle_dests(d, o, drawIdx);  // Collections.shuffle(dests, rd) first
        int32_t seed = draw_next_int(d, drawIdx);
        int32_t first = 0;
        // (latencies beside the ids: undelayed envelopes of an untagged engine whose ids and latencies fit 16 bits each —
        // a latency is < discardTime and < horizon, or the send is dropped / refused below)
        const bool packLat = !SH && !d.destTagged && !(o.pad & OUT_DELAYED) && d.nodes.n <= 65536 && d.horizon <= 65536;
        const int m = SH ? count_multi(d, o, from, seed, first)
                         : (o.to <= MULTI_LDS ? resolve_multi_lds(d, o, from, seed, &shMulti[0][0][0], &shMulti[1][0][0], packLat)
                                              : resolve_multi(d, o, from, seed));
        if (m == 1 && !SH) {
          const int32_t tw = d.sdests[multi_idx(d, o, 0)];
          fin = make_rec(K_MSG, from, (uint32_t)dest_id(d, tw), dest_msg(d, o.a, tw), o.b);
          arrival = d.arvTmp[multi_idx(d, o, 0)];
        } else if (m == 1) {  // (the one reachable destination: found again, the list is left as it is)
          const int32_t step = (o.pad & OUT_DELAYED) ? (int32_t)(o.pad >> 8) + 1 : 0;
          for (int j = 0; j < o.to && j < 64; j++) {
            int32_t a;
            const int32_t tw = d.sdests[multi_idx(d, o, j)], to = dest_id(d, tw);
            if (arrival_of_send(d, from, to, o.t + j * step, seed, a)) {
              fin = make_rec(K_MSG, from, (uint32_t)to, dest_msg(d, o.a, tw), o.b);
              arrival = a;
              break;
            }
          }
        } else if (m > 1 && SH) {
          // the envelope (slot, sorted destinations, explicit arrivals if any) is replicated state: it is created on every
          // shard from the exchanged image by k_shard_multi_fill / k_shard_multi_create; here only its place in the push
          // order (w1 = the entries it takes in the destination ring: MultiF sums them)
          fin = make_rec(K_CHAIN, from, (uint32_t)((o.pad & OUT_DELAYED) ? 2 * m : m), 0, MULTI_FRESH);
          arrival = first;
          atomicAdd(&d.xbuf[-XB_HEAD], 1);  // header word of the exchange image: envelopes to create (rare)
        } else if (m > 1) {
          uint32_t slot = atomicAdd(&d.g->chainHead, 1u) % d.chainSlots;
          if (d.chains[slot].flags & 1u) {
            set_err(d.g, ERR_CHAIN_SLOTS);
            break;
          }
          // (unsharded: sdests IS the chain ring — the sorted destinations are where the envelope reads them)
          if (o.pad & OUT_DELAYED)  // explicit arrivals follow the destinations (the action reserved 2 n entries)
            for (int j = 0; j < m; j++) d.dests[ring_at(o.destOff, (unsigned long long)(m + j), d.chainDests)] = d.arvTmp[multi_idx(d, o, j)];
          arrival = d.arvTmp[multi_idx(d, o, 0)];
          bool lat = packLat && d.arvTmp[multi_idx(d, o, m - 1)] - o.t < 65536;  // (sorted: the last is the largest)
          if (lat && o.to > MULTI_LDS)  // (the in-place sort left plain ids: the latencies go beside them now)
            for (int j = 0; j < m; j++) {
              const unsigned long long at = multi_idx(d, o, j);
              d.sdests[at] = (int32_t)((uint32_t)d.sdests[at] | ((uint32_t)(d.arvTmp[at] - o.t) << 16));
            }
          if (packLat && !lat && o.to <= MULTI_LDS)  // (never in practice — a latency of 65 536 ms —: back to plain ids)
            for (int j = 0; j < m; j++) {
              const unsigned long long at = multi_idx(d, o, j);
              d.sdests[at] = (int32_t)((uint32_t)d.sdests[at] & 0xFFFFu);
            }
          Chain c;
          c.from = from;
          c.seed = seed;
          c.sendTime = o.t;
          c.ndest = m;
          c.destOff = o.destOff;
          c.msg = o.a;
          c.payload = o.b;
          c.flags = 1u | ((o.pad & OUT_DELAYED) ? 2u : 0u) | (lat ? CHAIN_LAT : 0u);
          d.chains[slot] = c;
          fin = make_rec(K_CHAIN, from, slot, 0, 0);
        }
        break;
      }
      case O_TASK:
        fin = make_rec(K_TASK, from, (uint32_t)o.to, o.a, o.b);
        arrival = o.t;
        break;
      case O_PERIODIC:
        fin = make_rec(K_PERIODIC, from, (uint32_t)o.to, o.a, o.b);
        arrival = o.t;
        break;
      case O_SENDALL: {  // Network.sendAll(m, sendTime, from) inside an action(): N destinations, resolved by k_sendall_*
        if (d.maxSendAll == 0) {
          set_err(d.g, ERR_MULTI_TOO_BIG);
          break;
        }
        if (SH) {
          // sharded: the envelope (slot, destination slice, descriptor) is replicated state — every shard creates it from
          // the exchanged image (k_shard_multi_fill / k_shard_multi_create, numbered in push order by MultiF) and then
          // resolves the destinations itself (k_sendall_*: a pure function of (from, seed)); here only its place in the
          // push order. No arrival yet: k_sendall_scan files the first hop's.
          fin = make_rec(K_CHAIN, from, (uint32_t)d.nodes.n, 0, SENDALL_FRESH);
          arrival = -1;
          atomicAdd(&d.xbuf[-XB_HEAD], 1);
          break;
        }
        int32_t seed = draw_next_int(d, d.evDrawOff[e] + o.drawsub);
        const uint32_t k = atomicAdd(&d.g->nSendAll, 1u);
        if (k >= d.maxSendAll) {
          set_err(d.g, ERR_MULTI_TOO_BIG);
          break;
        }
        const uint32_t slot = atomicAdd(&d.g->chainHead, 1u) % d.chainSlots;
        if (d.chains[slot].flags & 1u) {
          set_err(d.g, ERR_CHAIN_SLOTS);
          break;
        }
        SendAllDesc sd;
        sd.p = p;
        sd.from = from;
        sd.seed = seed;
        sd.sendTime = o.t;
        sd.slot = slot;
        sd.msg = o.a;
        sd.payload = o.b;
        sd.pad = 0;
        sd.destOff = atomicAdd(&d.g->destHead, (unsigned long long)d.nodes.n) % d.chainDests;
        d.saDesc[k] = sd;
        d.fin[p] = make_rec(K_CHAIN, from, slot, 0, 0);
        d.arr[p] = -1;  // k_sendall_scan files the arrival of the first hop (and the histogram entry)
        continue;
      }
      default: {  // O_CHAINCONT: msgs.addMsg(m) after markRead (C/Network.java:629-632)
        const Chain c = d.chains[o.to];
        fin = make_rec(K_CHAIN, from, (uint32_t)o.to, o.a, 0);
        arrival = chain_arrival(d, c, (int)o.a);
      }
    }
    if (arrival >= 0) {
      if (arrival < t) {
        set_err(d.g, ERR_ARRIVAL_PAST);
        arrival = -1;
      } else if (arrival == t) {
        set_err(d.g, ERR_SAME_MS);
        arrival = -1;
      } else if (!SH && arrival - t >= d.horizon) {  // (sharded: k_shard_unpack parks it, on every shard alike)
        if (!park_far(d, t, p, fin, arrival)) set_err(d.g, ERR_HORIZON);
        arrival = -1;
      }
    }
    if (SH) {
      int32_t* x = d.xbuf + (size_t)p * 5;
      x[0] = (int32_t)fin.w0;
      x[1] = (int32_t)fin.w1;
      x[2] = (int32_t)fin.w2;
      x[3] = (int32_t)fin.w3;
      x[4] = arrival + 1;  // 0 = dropped at send time
      continue;
    }
    d.fin[p] = fin;
    d.arr[p] = arrival;
    // per-tile arrival histogram of the multisplit (rows are zero on entry: k_scatter re-zeroes them)
    if (arrival >= 0) atomicAdd(&d.tileHist[(size_t)(p / TILE) * D + ((uint32_t)arrival & (D - 1))], 1u);
  }
}

// sharded engine: the summed exchange image -> ordered outbox + tile histograms, identically on every shard
__global__ void __launch_bounds__(256) k_shard_unpack(const EngineDev* __restrict__ tab) {
  WG_ENGINE(tab);
  const int32_t t = d.g->now;
  const uint32_t n = d.g->nOut;
  const uint32_t D = (uint32_t)d.horizon;
  for (uint32_t p = wgBx * blockDim.x + threadIdx.x; p < n; p += wgGx * blockDim.x) {
    const int32_t* x = d.xbuf + (size_t)p * 5;
    Rec fin;
    fin.w0 = (uint32_t)x[0];
    fin.w1 = (uint32_t)x[1];
    fin.w2 = (uint32_t)x[2];
    fin.w3 = (uint32_t)x[3];
    int32_t arrival = x[4] - 1;
    if (arrival >= 0 && arrival - t >= d.horizon) {  // every shard holds the same far envelopes (the scheduler is replicated)
      if (!park_far(d, t, p, fin, arrival)) set_err(d.g, ERR_HORIZON);
      arrival = -1;
    }
    d.fin[p] = fin;
    d.arr[p] = arrival;
    if (arrival >= 0) atomicAdd(&d.tileHist[(size_t)(p / TILE) * D + ((uint32_t)arrival & (D - 1))], 1u);
  }
}

// sharded engine, multi-destination envelopes emitted in this phase (Network.send(m, from, dests) inside an
// action(), C/Network.java:418-447). MultiF numbers them in push order; the owner of the sender writes the
// envelope's image {seed, sendTime, msg, payload, ndest, destinations sorted by arrival}; after the sum across
// shards every shard creates the same envelope in the same slot of its (replicated) envelope table.
struct MultiF {
  typedef int Aux;
  const EngineDev& d;
  __device__ MultiF(const EngineDev& d_, const Aux*) : d(d_) {}
  __device__ uint32_t count() const { return d.g->nOut; }
  __device__ bool fresh(uint32_t p) const {
    const Rec r = d.fin[p];
    if (rec_kind(r) != K_CHAIN) return false;
    return (d.arr[p] >= 0 && r.w3 == MULTI_FRESH) || (d.arr[p] < 0 && r.w3 == SENDALL_FRESH);
  }
  __device__ uint64_t value(uint32_t p) const { return fresh(p) ? (((uint64_t)d.fin[p].w1 << 32) | 1u) : 0; }
  __device__ void tally(uint32_t, uint32_t) const {}
  __device__ void total(uint64_t tot) const {
    if ((uint32_t)tot > d.maxMulti) set_err(d.g, ERR_CHAIN_SLOTS);
    d.g->nMulti = min((uint32_t)tot, d.maxMulti);
    d.g->nMultiDests = (uint32_t)(tot >> 32);
  }
  __device__ void write(uint32_t p, uint64_t excl, bool valid) const {
    if (!valid) return;
    d.multiK[p] = (uint32_t)excl;
    d.multiOff[p] = (uint32_t)(excl >> 32);
  }
};

__global__ void __launch_bounds__(256) k_shard_multi_fill(const EngineDev* __restrict__ tab) {
  WG_ENGINE(tab);
  const uint32_t n = d.g->nOut;
  const MultiF f(d, nullptr);
  for (uint32_t p = wgBx * blockDim.x + threadIdx.x; p < n; p += wgGx * blockDim.x) {
    if (!f.fresh(p) || d.multiK[p] >= d.maxMulti) continue;
    const uint32_t e = d.recEv[p];
    if (!shard_owns(d, (int32_t)d.ev[e].w1)) continue;  // (the image is zero on entry)
    const Out o = d.outTmp[d.evAux[e].outBase + (p - d.evRecOff[e])];
    const int32_t from = (int32_t)(o.kindfrom & 0x0FFFFFFFu);
    if ((o.kindfrom >> 28) == O_SENDALL) {  // Network.sendAll: the descriptor; destinations are resolved on every shard
      int32_t* x = d.xmulti + (size_t)d.multiK[p] * XM_WORDS;
      x[0] = draw_next_int(d, d.evDrawOff[e] + o.drawsub);
      x[1] = o.t;
      x[2] = (int32_t)o.a;
      x[3] = (int32_t)o.b;
      x[4] = -1;
      continue;
    }
    // (a shuffled list was permuted in the scratch ring by k_resolve<true> already; its draws precede the seed)
    const uint32_t shuffled = (o.pad & OUT_SHUFFLE) && o.to > 1 ? (uint32_t)((o.to < 64 ? o.to : 64) - 1) : 0u;
    const int32_t seed = draw_next_int(d, d.evDrawOff[e] + o.drawsub + shuffled);
    const int m = resolve_multi(d, o, from, seed);  // (in the shard's private list ring)
    int32_t* x = d.xmulti + (size_t)d.multiK[p] * XM_WORDS;
    x[0] = seed;
    x[1] = o.t;
    x[2] = (int32_t)o.a;
    x[3] = (int32_t)o.b;
    x[4] = m;
    x[5] = (o.pad & OUT_DELAYED) ? 1 : 0;
    for (int j = 0; j < m; j++) x[6 + j] = d.sdests[multi_idx(d, o, j)];
    if (o.pad & OUT_DELAYED)  // MultipleDestWithDelayEnvelope: explicit arrivals (C/Network.java:449-467)
      for (int j = 0; j < m; j++) x[6 + 64 + j] = d.arvTmp[multi_idx(d, o, j)];
  }
}

__global__ void __launch_bounds__(256) k_shard_multi_create(const EngineDev* __restrict__ tab) {
  WG_ENGINE(tab);
  const uint32_t n = d.g->nOut;
  const MultiF f(d, nullptr);
  for (uint32_t p = wgBx * blockDim.x + threadIdx.x; p < n; p += wgGx * blockDim.x) {
    if (!f.fresh(p) || d.multiK[p] >= d.maxMulti) continue;
    const int32_t* x = d.xmulti + (size_t)d.multiK[p] * XM_WORDS;
    const Rec r = d.fin[p];
    const uint32_t slot = (d.g->chainHead + d.multiK[p]) % d.chainSlots;
    if (d.chains[slot].flags & 1u) {
      set_err(d.g, ERR_CHAIN_SLOTS);
      continue;
    }
    const unsigned long long off = (d.g->destHead + d.multiOff[p]) % d.chainDests;
    const int m = x[4];
    if (m < 0) {  // a Network.sendAll: descriptor for k_sendall_* (which create the envelope and file its first arrival)
      const uint32_t k = atomicAdd(&d.g->nSendAll, 1u);
      if (k >= d.maxSendAll) {
        set_err(d.g, ERR_MULTI_TOO_BIG);
        continue;
      }
      SendAllDesc sd;
      sd.p = p;
      sd.from = rec_from(r);
      sd.seed = x[0];
      sd.sendTime = x[1];
      sd.slot = slot;
      sd.msg = (uint32_t)x[2];
      sd.payload = (uint32_t)x[3];
      sd.pad = 0;
      sd.destOff = off;
      d.saDesc[k] = sd;
      d.fin[p] = make_rec(K_CHAIN, sd.from, slot, 0, 0);
      continue;
    }
    for (int j = 0; j < m; j++) d.dests[ring_at(off, (unsigned long long)j, d.chainDests)] = x[6 + j];
    if (x[5])  // explicit arrivals follow the destinations (as k_resolve<false> lays them out; Chain::flags bit 1)
      for (int j = 0; j < m; j++) d.dests[ring_at(off, (unsigned long long)(m + j), d.chainDests)] = x[6 + 64 + j];
    Chain c;
    c.from = rec_from(r);
    c.seed = x[0];
    c.sendTime = x[1];
    c.ndest = m;
    c.destOff = (uint32_t)off;
    c.msg = (uint32_t)x[2];
    c.payload = (uint32_t)x[3];
    c.flags = 1u | (x[5] ? 2u : 0u);
    d.chains[slot] = c;
    d.fin[p] = make_rec(K_CHAIN, c.from, slot, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------
// append: stable multisplit of the ordered outbox by arrival bucket.

// in-tile stable rank of each element among equal bins: wave-level ballot match, then waves in order
// through an LDS running count. `hist` is this block's LDS histogram [D]; must be zero on entry and
// holds the per-bin tile totals on exit.
__device__ __forceinline__ uint32_t tile_rank(uint32_t* hist, int bin, bool valid, int binBits) {
  uint64_t m = __ballot(valid);
  if (valid) {
    for (int b = 0; b < binBits; b++) {
      uint64_t bb = __ballot((bin >> b) & 1);
      m &= ((bin >> b) & 1) ? bb : ~bb;
    }
  }
  uint32_t rankInWave = __popcll(m & lanes_lt());
  uint32_t group = __popcll(m);
  uint32_t rank = 0;
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int k = 0; k < nw; k++) {
    if (k == w && valid) {
      rank = hist[bin] + rankInWave;
    }
    __syncthreads();
    if (k == w && valid && rankInWave + 1 == group) hist[bin] += group;
    __syncthreads();
  }
  return rank;
}

// (host-staged envelopes only: the device pipeline builds the histogram inside k_resolve / cond_a2)
__global__ void __launch_bounds__(TILE) k_tile_hist(const EngineDev* __restrict__ tab, int binBits) {
  WG_ENGINE(tab);
  WG_DYN_LDS(uint32_t, hist);
  uint32_t n = d.g->nOutKeep + d.g->nOut;
  uint32_t nTiles = (n + TILE - 1) / TILE;
  uint32_t D = (uint32_t)d.horizon;
  for (uint32_t tile = wgBx; tile < nTiles; tile += wgGx) {
    for (uint32_t b = threadIdx.x; b < D; b += TILE) hist[b] = 0;
    __syncthreads();
    uint32_t i = tile * TILE + threadIdx.x;
    if (i < n) {
      int32_t a = d.arr[i];
      if (a >= 0) atomicAdd(&hist[(uint32_t)a & (D - 1)], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < D; b += TILE) d.tileHist[(size_t)tile * D + b] = hist[b];
    __syncthreads();
  }
}

// One block: per-bin exclusive prefix over tiles (in place), then reserve pages for every bucket that grows
// (MessageStorage.ensureSize analogue) and publish each bucket's append base.
__device__ __forceinline__ void col_reserve_body(const EngineDev& d) {
  __shared__ uint32_t shNeed[16];
  __shared__ uint32_t shBase;
  uint32_t n = d.g->nOutKeep + d.g->nOut;  // (a drain's outbox kept over the edge's conditional phase + that phase's records)
  uint32_t nTiles = (n + TILE - 1) / TILE;
  uint32_t D = (uint32_t)d.horizon;
  for (uint32_t b0 = 0; b0 < D; b0 += blockDim.x) {
    uint32_t b = b0 + threadIdx.x;
    uint32_t add = 0;
    if (b < D) {
      // in-place exclusive prefix over the tiles of this bin, eight tiles a round: the loads of a round are issued
      // together (one at a time, each iteration waited a memory round trip — the millisecond in which every node
      // disseminates has hundreds of tiles, and this is one block)
      uint32_t tile = 0;
      for (; tile + 8 <= nTiles; tile += 8) {
        uint32_t h[8];
#pragma unroll
        for (int q = 0; q < 8; q++) h[q] = d.tileHist[(size_t)(tile + q) * D + b];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          d.tileHist[(size_t)(tile + q) * D + b] = add;
          add += h[q];
        }
      }
      for (; tile < nTiles; tile++) {
        uint32_t h = d.tileHist[(size_t)tile * D + b];
        d.tileHist[(size_t)tile * D + b] = add;
        add += h;
      }
    }
    uint32_t have = 0, need = 0, cur = 0;
    if (b < D) {
      cur = d.bcnt[b];
      have = (cur + PAGE_RECS - 1) >> PAGE_SHIFT;
      need = ((cur + add + PAGE_RECS - 1) >> PAGE_SHIFT) - have;
      if (have + need > (uint32_t)d.maxPagesPerBucket) {
        set_err(d.g, ERR_BUCKET_PAGES);
        need = 0;
        add = 0;
      }
    }
    uint32_t totalNeed;
    uint32_t before = block_excl_scan32_1024(need, shNeed, &totalNeed);
    if (threadIdx.x == 0) shBase = d.g->freeTop;
    __syncthreads();
    bool ok = totalNeed <= shBase;
    if (!ok && threadIdx.x == 0) set_err(d.g, ERR_BUCKET_POOL);
    if (b < D) {
      if (ok) {
        for (uint32_t k = 0; k < need; k++)
          d.pagetab[(size_t)b * d.maxPagesPerBucket + have + k] = d.freeStack[shBase - 1 - (before + k)];
        d.binBase[b] = cur;
        d.bcnt[b] = cur + add;
      } else {
        d.binBase[b] = cur;  // nothing appended
      }
    }
    __syncthreads();
    if (threadIdx.x == 0 && ok) d.g->freeTop = shBase - totalNeed;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(1024) k_col_reserve(const EngineDev* __restrict__ tab) {
  WG_ENGINE(tab);
  col_reserve_body(d);
}
__global__ void __launch_bounds__(TILE) k_scatter(const EngineDev* __restrict__ tab, int binBits, int saved) {
  WG_ENGINE(tab);
  WG_DYN_LDS(uint32_t, hist);
  uint32_t n = saved ? d.g->nScatter : d.g->nOutKeep + d.g->nOut;  // (saved: the phase's counters were reset by k_col_reserve_end already)
  uint32_t nTiles = (n + TILE - 1) / TILE;
  uint32_t D = (uint32_t)d.horizon;
  if (d.g->err & (ERR_BUCKET_POOL | ERR_BUCKET_PAGES)) nTiles = 0;
  for (uint32_t tile = wgBx; tile < nTiles; tile += wgGx) {
    for (uint32_t b = threadIdx.x; b < D; b += TILE) hist[b] = 0;
    __syncthreads();
    uint32_t i = tile * TILE + threadIdx.x;
    int32_t a = i < n ? d.arr[i] : -1;
    bool valid = a >= 0;
    int bin = valid ? (int)((uint32_t)a & (D - 1)) : 0;
    uint32_t rank = tile_rank(hist, bin, valid, binBits);
    if (valid) {
      uint32_t pos = d.binBase[bin] + d.tileHist[(size_t)tile * D + bin] + rank;
      *rec_ptr(d, (uint32_t)bin, pos) = d.fin[i];
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < D; b += TILE) d.tileHist[(size_t)tile * D + b] = 0;  // for the next phase
  }
}

// ------------------------------------------------------------------------------------------------
// send expand: createMessageArrivals + its stable sort (C/Network.java:449-467) and the destination array of the
// MultipleDestEnvelope (C/Envelope.java:66-81) for ONE host-issued multi-destination send (sendAll of a protocol's
// init(), or any list send in host-callback mode) with many destinations: latency per destination (drops for
// partitions / down nodes / msgDiscardTime as createMessageArrival :469-487), then a stable counting sort by
// latency — the same tile histogram + wave-ballot rank multisplit as `append`, aimed at the envelope's slice of the
// destination ring instead of the buckets. Equal arrivals keep the caller's order, as Java's stable sort does.
struct SendExpand {
  const int32_t* in;   // [n] destination ids in caller order
  int32_t* lat;        // [n] scratch: latency, -1 = dropped
  uint32_t* hist;      // [tiles][D] scratch
  int32_t* result;     // [2]: reachable destinations m, smallest latency
  int32_t n, from, seed;
  unsigned long long destOff;
};
__global__ void __launch_bounds__(TILE) k_send_expand_lat(const EngineDev* __restrict__ tab, SendExpand x) {
  const EngineDev& d = tab[0];
  WG_DYN_LDS(uint32_t, hist);
  const uint32_t D = (uint32_t)d.horizon;
  const uint32_t nTiles = ((uint32_t)x.n + TILE - 1) / TILE;
  for (uint32_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    for (uint32_t b = threadIdx.x; b < D; b += TILE) hist[b] = 0;
    __syncthreads();
    const uint32_t j = tile * TILE + threadIdx.x;
    if (j < (uint32_t)x.n) {
      const int32_t to = x.in[j];
      int32_t nt = -1;
      const NodeArrays& n = d.nodes;
      if (n.part[x.from] == n.part[to] && !n.down[x.from] && !n.down[to]) {
        nt = dev_latency(d, x.from, to, x.seed);
        if (nt >= d.discardTime) nt = -1;
      }
      if (nt >= (int32_t)D) {  // (a hop that far ahead cannot be filed in the bucket ring either)
        set_err(d.g, ERR_HORIZON);
        nt = -1;
      }
      x.lat[j] = nt;
      if (nt >= 0) atomicAdd(&hist[nt], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < D; b += TILE) x.hist[(size_t)tile * D + b] = hist[b];
    __syncthreads();
  }
}
// single block: per latency value the exclusive prefix over tiles, then over latency values
__global__ void __launch_bounds__(1024) k_send_expand_scan(const EngineDev* __restrict__ tab, SendExpand x) {
  const EngineDev& d = tab[0];
  __shared__ uint32_t sh16[16];
  __shared__ uint32_t shCarry;
  __shared__ int32_t shMin;
  const uint32_t D = (uint32_t)d.horizon;
  const uint32_t nTiles = ((uint32_t)x.n + TILE - 1) / TILE;
  if (threadIdx.x == 0) {
    shCarry = 0;
    shMin = INT32_MAX;
  }
  __syncthreads();
  for (uint32_t b0 = 0; b0 < D; b0 += 1024) {
    const uint32_t b = b0 + threadIdx.x;
    uint32_t tot = 0;
    if (b < D)
      for (uint32_t t = 0; t < nTiles; t++) {
        const uint32_t h = x.hist[(size_t)t * D + b];
        x.hist[(size_t)t * D + b] = tot;
        tot += h;
      }
    uint32_t all;
    const uint32_t before = block_excl_scan32_1024(tot, sh16, &all);
    const uint32_t carry = shCarry;
    if (b < D) {
      for (uint32_t t = 0; t < nTiles; t++) x.hist[(size_t)t * D + b] += carry + before;
      if (tot) atomicMin(&shMin, (int32_t)b);
    }
    __syncthreads();
    if (threadIdx.x == 0) shCarry = carry + all;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    x.result[0] = (int32_t)shCarry;
    x.result[1] = shMin;
  }
}
__global__ void __launch_bounds__(TILE) k_send_expand_scatter(const EngineDev* __restrict__ tab, SendExpand x, int binBits) {
  const EngineDev& d = tab[0];
  WG_DYN_LDS(uint32_t, hist);
  const uint32_t D = (uint32_t)d.horizon;
  const uint32_t nTiles = ((uint32_t)x.n + TILE - 1) / TILE;
  for (uint32_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    for (uint32_t b = threadIdx.x; b < D; b += TILE) hist[b] = 0;
    __syncthreads();
    const uint32_t j = tile * TILE + threadIdx.x;
    const int32_t nt = j < (uint32_t)x.n ? x.lat[j] : -1;
    const bool valid = nt >= 0;
    const int bin = valid ? nt : 0;
    const uint32_t rank = tile_rank(hist, bin, valid, binBits);
    if (valid) d.dests[ring_at(x.destOff, (unsigned long long)x.hist[(size_t)tile * D + bin] + rank, d.chainDests)] = x.in[j];
    __syncthreads();
  }
}

// The same three steps for the Network.sendAll calls an action() made in this phase (O_SENDALL): destinations are all
// nodes in id order (C/Network.java:341-347); one descriptor per call, blockIdx.y strides over the descriptors. The scan
// step also creates the envelope and files the arrival of its first hop in the ordered outbox (+ tile histogram).
__global__ void __launch_bounds__(TILE) k_sendall_lat(const EngineDev* __restrict__ tab) {
  WG_ENGINE(tab);
  WG_DYN_LDS(uint32_t, hist);
  // D: the histograms span the latencies the model can return (saBins <= horizon), not the whole bucket ring
  const uint32_t D = d.saBins, N = (uint32_t)d.nodes.n;
  const uint32_t nTiles = (N + TILE - 1) / TILE, nSA = min(d.g->nSendAll, d.maxSendAll);
  for (uint32_t wi = wgBx; wi < nSA * nTiles; wi += wgGx) {  // work item = (descriptor, tile)
    const uint32_t k = wi / nTiles, tile = wi % nTiles;
    const SendAllDesc sd = d.saDesc[k];
    {
      for (uint32_t b = threadIdx.x; b < D; b += TILE) hist[b] = 0;
      __syncthreads();
      const uint32_t to = tile * TILE + threadIdx.x;
      if (to < N) {
        int32_t nt = -1;
        const NodeArrays& n = d.nodes;
        if (n.part[sd.from] == n.part[to] && !n.down[sd.from] && !n.down[to]) {
          nt = dev_latency(d, sd.from, (int32_t)to, sd.seed);
          if (nt >= d.discardTime) nt = -1;
        }
        if (nt >= (int32_t)D) {  // (beyond the model's own bound or the ring: loud)
          set_err(d.g, ERR_HORIZON);
          nt = -1;
        }
        d.saLat[(size_t)k * N + to] = nt;
        if (nt >= 0) atomicAdd(&hist[nt], 1u);
      }
      __syncthreads();
      for (uint32_t b = threadIdx.x; b < D; b += TILE) d.saHist[((size_t)k * nTiles + tile) * D + b] = hist[b];
      __syncthreads();
    }
  }
}
__global__ void __launch_bounds__(1024) k_sendall_scan(const EngineDev* __restrict__ tab) {
  WG_ENGINE(tab);
  __shared__ uint32_t sh16[16];
  __shared__ uint32_t shCarry;
  __shared__ int32_t shMin;
  const uint32_t D = d.saBins, N = (uint32_t)d.nodes.n;  // latency bins (k_sendall_lat)
  const uint32_t RING = (uint32_t)d.horizon;
  const uint32_t nTiles = (N + TILE - 1) / TILE, nSA = min(d.g->nSendAll, d.maxSendAll);
  const int32_t t = d.g->now;
  for (uint32_t k = wgBx; k < nSA; k += wgGx) {
    const SendAllDesc sd = d.saDesc[k];
    uint32_t* H = d.saHist + (size_t)k * nTiles * D;
    if (threadIdx.x == 0) {
      shCarry = 0;
      shMin = INT32_MAX;
    }
    __syncthreads();
    for (uint32_t b0 = 0; b0 < D; b0 += 1024) {
      const uint32_t b = b0 + threadIdx.x;
      uint32_t tot = 0;
      if (b < D)
        for (uint32_t tl = 0; tl < nTiles; tl++) {
          const uint32_t h = H[(size_t)tl * D + b];
          H[(size_t)tl * D + b] = tot;
          tot += h;
        }
      uint32_t all;
      const uint32_t before = block_excl_scan32_1024(tot, sh16, &all);
      const uint32_t carry = shCarry;
      if (b < D) {
        for (uint32_t tl = 0; tl < nTiles; tl++) H[(size_t)tl * D + b] += carry + before;
        if (tot) atomicMin(&shMin, (int32_t)b);
      }
      __syncthreads();
      if (threadIdx.x == 0) shCarry = carry + all;
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const int32_t m = (int32_t)shCarry;
      int32_t arrival = -1;
      if (m > 0) {
        arrival = sd.sendTime + shMin;
        if (arrival <= t) {  // (sendAll(m, sendTime, from) with sendTime <= time throws in the reference, :471)
          set_err(d.g, arrival == t ? ERR_SAME_MS : ERR_ARRIVAL_PAST);
          arrival = -1;
        } else if (arrival - t >= d.horizon) {
          set_err(d.g, ERR_HORIZON);
          arrival = -1;
        }
      }
      if (arrival >= 0) {
        Chain c;
        c.from = sd.from;
        c.seed = sd.seed;
        c.sendTime = sd.sendTime;
        c.ndest = m;
        c.destOff = (uint32_t)sd.destOff;
        c.msg = sd.msg;
        c.payload = sd.payload;
        c.flags = 1u;
        d.chains[sd.slot] = c;  // (a single reachable destination stays a one-hop envelope: same delivery)
        d.arr[sd.p] = arrival;
        atomicAdd(&d.tileHist[(size_t)(sd.p / TILE) * RING + ((uint32_t)arrival & (RING - 1))], 1u);
      }
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(TILE) k_sendall_scatter(const EngineDev* __restrict__ tab, int binBits) {
  WG_ENGINE(tab);
  WG_DYN_LDS(uint32_t, hist);
  const uint32_t D = d.saBins, N = (uint32_t)d.nodes.n;  // latency bins (k_sendall_lat)
  const uint32_t nTiles = (N + TILE - 1) / TILE, nSA = min(d.g->nSendAll, d.maxSendAll);
  for (uint32_t wi = wgBx; wi < nSA * nTiles; wi += wgGx) {  // work item = (descriptor, tile)
    const uint32_t k = wi / nTiles, tile = wi % nTiles;
    const SendAllDesc sd = d.saDesc[k];
    {
      for (uint32_t b = threadIdx.x; b < D; b += TILE) hist[b] = 0;
      __syncthreads();
      const uint32_t to = tile * TILE + threadIdx.x;
      const int32_t nt = to < N ? d.saLat[(size_t)k * N + to] : -1;
      const bool valid = nt >= 0;
      const int bin = valid ? nt : 0;
      const uint32_t rank = tile_rank(hist, bin, valid, binBits);
      if (valid) d.dests[ring_at(sd.destOff, (unsigned long long)d.saHist[((size_t)k * nTiles + tile) * D + bin] + rank, d.chainDests)] = (int32_t)to;
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// end of a phase: advance rd by the draws consumed, reset scratch counters; after a drain also
// release the bucket's pages and bump the nextMessage() epoch if anything was polled.
__device__ __forceinline__ void end_phase_body(const EngineDev& d, int drained) {
  __shared__ uint32_t shTop;
  Globals WG_G* g = d.g;
  const int32_t t = g->now;
  if (threadIdx.x == 0) {
    shTop = g->freeTop;
    g->chainHead += g->nMulti;  // (sharded engines only: envelopes created by k_shard_multi_create; 0 otherwise)
    g->destHead += g->nMultiDests;
    g->nMulti = 0;
    g->nMultiDests = 0;
    g->nSendAll = 0;
    g->nRuns = 0;
  }
  __syncthreads();
  if (drained) {
    uint32_t b = (uint32_t)t & (uint32_t)(d.horizon - 1);
    uint32_t cnt = d.bcnt[b];
    uint32_t pages = (cnt + PAGE_RECS - 1) >> PAGE_SHIFT;
    for (uint32_t k = threadIdx.x; k < pages; k += blockDim.x)
      d.freeStack[shTop + k] = d.pagetab[(size_t)b * d.maxPagesPerBucket + k];
    __syncthreads();
    if (threadIdx.x == 0) {
      g->freeTop = shTop + pages;
      d.bcnt[b] = 0;
      if (g->nEvents > 0) {
        g->epoch++;
        g->anyEvent = 1;
        g->events += g->nEvents;
      }
      unsigned long long oldD = d.destHeadAt[b], oldP = d.payloadHeadAt[b];
      if (g->destHead - oldD > d.chainDests) set_err(g, ERR_CHAIN_DESTS);
      if (g->payloadHead - oldP > d.payloadWords) set_err(g, ERR_PAYLOAD);
      d.destHeadAt[b] = g->destHead;
      d.payloadHeadAt[b] = g->payloadHead;
      g->now = t + 1;  // nextMessage(): time++
    }
  }
  if (threadIdx.x == 0) {
    g->rng = lcg_skip(g->rng, g->nDraws);
    g->draws += g->nDraws;
    g->nEvents = 0;
    g->nActive = 0;
    g->nActiveB = 0;
    g->outSlots = 0;
    g->nOut = 0;
    g->nDraws = 0;
    g->rejectSeen = 0;
  }
}
// keep: the phase's ordered outbox is NOT filed yet — the conditional-task phase of the edge appends its records behind it
// and one k_col_reserve_end + k_scatter files both (a launch less per simulated ms; the buckets' push order is the same:
// the drain's records, then the edge's). (Folding this one-block launch into k_resolve as "the last block to finish ends the
// phase" needs a device-scope fence per block — an L2 write-back on this chip: 563 -> 483 M msgs/s, profiles/r20q_*.)
__global__ void __launch_bounds__(256) k_end_phase(const EngineDev* __restrict__ tab, int drained, int keep) {
  WG_ENGINE(tab);
  const uint32_t nOut = d.g->nOut;
  __syncthreads();
  end_phase_body(d, drained);
  if (threadIdx.x == 0) d.g->nOutKeep = keep ? nOut : 0u;
}
// k_col_reserve and k_end_phase as ONE launch of one block per engine (both are a single block's short chain of dependent
// accesses, ~ 5 us of launch + latency each, twice per simulated ms): the end-of-phase bookkeeping touches nothing k_scatter
// reads — except nOut, which is handed over in nScatter — and the drained bucket's pages go back on the free stack after
// this phase's reservations were taken from it, as before.
__global__ void __launch_bounds__(1024) k_col_reserve_end(const EngineDev* __restrict__ tab, int drained) {
  WG_ENGINE(tab);
  col_reserve_body(d);
  __syncthreads();
  if (threadIdx.x == 0) {
    d.g->nScatter = d.g->nOutKeep + d.g->nOut;
    d.g->nOutKeep = 0;
  }
  __syncthreads();
  end_phase_body(d, drained);
}

// Idle stretches (protocols without conditional tasks): an empty bucket's ms does nothing (nextMessage just moves the
// clock, C/Network.java:533-570), yet its launch sequence costs the host ~15 launches. k_next_busy reports how many ms
// from `now` the first non-empty bucket is (horizon if none); k_skip_idle moves the clock over n empty ms, leaving the
// per-ms ring bookkeeping as their k_end_phase would have.
__global__ void __launch_bounds__(256) k_next_busy(const EngineDev* __restrict__ tab, int32_t* __restrict__ out) {
  const EngineDev& d = tab[blockIdx.y];
  __shared__ uint32_t shMin;
  if (threadIdx.x == 0) shMin = d.halted ? 0x7FFFFFFFu : (uint32_t)d.horizon;
  __syncthreads();
  if (!d.halted) {
    const uint32_t D = (uint32_t)d.horizon, now = (uint32_t)d.g->now;
    for (uint32_t k = threadIdx.x; k < D; k += blockDim.x)
      if (d.bcnt[(now + k) & (D - 1)] != 0) {
        atomicMin(&shMin, k);
        break;
      }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.y] = (int32_t)shMin;
}
__global__ void __launch_bounds__(256) k_skip_idle(const EngineDev* __restrict__ tab, int32_t n) {
  WG_ENGINE(tab);
  Globals WG_G* g = d.g;
  const uint32_t D = (uint32_t)d.horizon;
  const int32_t t = g->now;
  const unsigned long long dh = g->destHead, ph = g->payloadHead;
  for (int32_t k = threadIdx.x; k < n; k += blockDim.x) {
    const uint32_t b = (uint32_t)(t + k) & (D - 1);
    d.destHeadAt[b] = dh;
    d.payloadHeadAt[b] = ph;
  }
  __syncthreads();
  if (threadIdx.x == 0) g->now = t + n;
}

__global__ void k_latency_probe(const EngineDev* __restrict__ tab, int n, const int32_t* from, const int32_t* to, const int32_t* delta,
                                int32_t* out) {
  const EngineDev& d = tab[0];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const NodeArrays& nd = d.nodes;
  int f = from[i], t = to[i];
  out[i] = latency_of(d.lat, f, t, nd.x[f], nd.y[f], nd.extraLatency[f], nd.x[t], nd.y[t], nd.extraLatency[t],
                      delta[i]);
}

// ------------------------------------------------------------------------------------------------
// Delivery context handed to protocol action() code. One wavefront per destination node; every
// lane runs the same control flow, scalars are wave-uniform, lane 0 performs the scalar stores.
struct Ctx {
  const EngineDev& d;
  int32_t t;       // network.time
  int32_t node;    // the node whose action() runs (`to`)
  uint32_t ev;     // event index in the global order
  uint32_t sub;    // records emitted by this event so far
  uint32_t draws;  // rd.nextInt() calls by this event so far
  uint32_t outBase, outCap;      // the event's private outbox slice
  long long msgSent, bytesSent;  // accumulated Node counters (C/Network.java:476-477)
  uint32_t evFlags;              // OR-ed into the event's EvRes::nrec by whoever writes it (EV_SNAP_* code of a snapshot)

  __device__ void put(uint32_t kind, int32_t to, uint32_t a, uint32_t b, int32_t tt, uint32_t destOff, bool draw,
                      uint32_t pad = 0, uint32_t extraDraws = 0) {
    if (sub < outCap && outBase + sub < d.maxOut) {
      // the 32-byte record as ONE memory instruction: lanes 0 and 1 store 16 bytes each
      U4 q;
      if (WG_LANE == 0) {
        q.x = (kind << 28) | (uint32_t)node;
        q.y = (uint32_t)to;
        q.z = a;
        q.w = b;
      } else {
        q.x = (uint32_t)tt;
        q.y = destOff;
        q.z = draws;
        q.w = pad;
      }
      if (WG_LANE < 2) ((U4*)&d.outTmp[outBase + sub])[WG_LANE] = q;
    } else if (WG_LANE == 0) {
      set_err(d.g, ERR_OUTBOX);  // the protocol's emission bound (EngineDev::boundMsg/boundTask) is wrong
    }
    if (sub < outCap) sub++;
    if (draw) draws += 1 + extraDraws;
  }
  // Network.send(m, this, to): sendTime = time + 1, one rd.nextInt() (C/Network.java:364-382)
  __device__ void send(int32_t to, uint32_t msg, uint32_t payload, int size) {
    msgSent++;
    bytesSent += size;
    put(O_SEND, to, msg, payload, t + 1, 0, true);
  }
  // n single-destination Network.send calls of one action(), written by n lanes at once: lane `active` with
  // `rank` (0..n-1, the order the reference would issue them in) writes the rank-th record; every lane
  // calls this with the same n / bytesTotal
  __device__ void send_many(bool active, int rank, int n, int32_t to, uint32_t msg, uint32_t payload, long long bytesTotal) {
    if (active) {
      const uint32_t idx = sub + (uint32_t)rank;
      if (idx < outCap && outBase + idx < d.maxOut) {
        Out o;
        o.kindfrom = (O_SEND << 28) | (uint32_t)node;
        o.to = to;
        o.a = msg;
        o.b = payload;
        o.t = t + 1;
        o.destOff = 0;
        o.drawsub = draws + (uint32_t)rank;
        o.pad = 0;
        d.outTmp[outBase + idx] = o;
      } else {
        set_err(d.g, ERR_OUTBOX);
      }
    }
    sub = min(sub + (uint32_t)n, outCap);
    draws += (uint32_t)n;
    msgSent += n;
    bytesSent += bytesTotal;
  }
  // Network.send(m, this, dests) (:353-362): empty -> nothing, one -> single send, else multi-dest.
  // The destination ids must already be in the dest ring at destOff (dest_reserve).
  __device__ uint32_t dest_reserve(int n) {
    unsigned long long off = 0;
    if (WG_LANE == 0) off = atomicAdd(d.sharded ? &d.g->localDestHead : &d.g->destHead, (unsigned long long)n);
    off = lane_bcast64(off, 0);
    return (uint32_t)(off % d.sdestCap);
  }
  __device__ void dest_put(uint32_t destOff, int j, int32_t id) {
    d.sdests[ring_at(destOff, (unsigned long long)j, d.sdestCap)] = id;
  }
  __device__ void send_list(uint32_t destOff, int n, uint32_t msg, uint32_t payload, int size) {
    if (n == 0) return;
    msgSent += n;
    bytesSent += (long long)n * size;
    if (n == 1) {
      int32_t to = __hip_atomic_load(&d.sdests[ring_at(destOff, 0, d.sdestCap)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      put(O_SEND, dest_id(d, to), dest_msg(d, msg, to), payload, t + 1, 0, true);
    } else {
      if (n > 64) {
        if (WG_LANE == 0) set_err(d.g, ERR_MULTI_TOO_BIG);
        n = 64;
      }
      put(O_MULTI, n, msg, payload, t + 1, destOff, true);
    }
  }
  // Collections.shuffle(dests, rd); network.send(m, this, dests): the shuffle's n - 1 draws and the permutation are
  // deferred to `resolve` (shuffle_dests); the list at destOff is in the order the action built it
  __device__ void send_list_shuffled(uint32_t destOff, int n, uint32_t msg, uint32_t payload, int size) {
    if (n == 0) return;
    msgSent += n;
    bytesSent += (long long)n * size;
    if (n == 1) {
      int32_t to = __hip_atomic_load(&d.sdests[ring_at(destOff, 0, d.sdestCap)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      put(O_SEND, dest_id(d, to), dest_msg(d, msg, to), payload, t + 1, 0, true);
      return;
    }
    if (n > 64) {
      if (WG_LANE == 0) set_err(d.g, ERR_MULTI_TOO_BIG);
      n = 64;
    }
    put(O_MULTI, n, msg, payload, t + 1, destOff, true, OUT_SHUFFLE, (uint32_t)(n - 1));
  }
  // Collections.shuffle(dests, rd); network.send(m, sendTime, this, dests, delaysBetweenMessage) (:418-447, FloodMessage
  // :52-54): the list overload — it draws its seed even for an empty or one-element list; the list at destOff must have
  // been reserved with 2 n entries (destinations, then their explicit arrivals)
  __device__ void send_list_delayed_shuffled(uint32_t destOff, int n, uint32_t msg, uint32_t payload, int32_t sendTime,
                                             int delay, int size) {
    msgSent += n;
    bytesSent += (long long)n * size;
    if (n > 64) {
      if (WG_LANE == 0) set_err(d.g, ERR_MULTI_TOO_BIG);
      n = 64;
    }
    put(O_MULTI, n, msg, payload, sendTime, destOff, true, OUT_SHUFFLE | OUT_DELAYED | ((uint32_t)delay << 8),
        n > 1 ? (uint32_t)(n - 1) : 0u);
  }
  // Network.sendAll(m, sendTime, this) (:341-347): every node of the network is a destination, one rd draw
  __device__ void send_all(uint32_t msg, uint32_t payload, int32_t sendTime, int size) {
    msgSent += d.nodes.n;
    bytesSent += (long long)d.nodes.n * size;
    put(O_SENDALL, d.nodes.n, msg, payload, sendTime, 0, true);
  }
  // Network.registerTask(r, startAt, this) (:505-508)
  __device__ void register_task(int32_t startAt, uint32_t word, uint32_t arg) {
    put(O_TASK, node, word, arg, startAt, 0, false);
  }
  // engine payload ring (irregular senders): `words` 64-bit words, returns the word offset
  __device__ uint32_t alloc_payload(int words) {
    unsigned long long off = 0;
    if (WG_LANE == 0) off = atomicAdd(&d.g->payloadHead, (unsigned long long)words);
    off = lane_bcast64(off, 0);
    // keep an allocation contiguous: skip the tail of the ring if it does not fit
    unsigned long long pos = off % d.payloadWords;
    if (pos + words > d.payloadWords) {
      if (WG_LANE == 0) off = atomicAdd(&d.g->payloadHead, (unsigned long long)words);
      off = lane_bcast64(off, 0);
      pos = off % d.payloadWords;
      if (pos + words > d.payloadWords) {
        if (WG_LANE == 0) set_err(d.g, ERR_PAYLOAD);
        pos = 0;
      }
    }
    return (uint32_t)pos;
  }
};

// The delivery kernel: receiveUntil's loop body (C/Network.java:594-635) for all events of ms t.
// A protocol P provides:
//   State                      device pointers of its SoA state (kernel argument)
//   WaveShared                 per-wavefront LDS scratch
//   NodeRegs                   wave-uniform registers holding the node's scalars between its events
//   node_begin / node_end      load / store NodeRegs
//   msg_size(State, word)      Message.size()
//   msg_level(word)            statistics bucket (Handel level) 0..31
//   on_message / on_task       Message.action()
template <class P>
__device__ __forceinline__ void deliver_event(const EngineDev& d, const typename P::State& ps, Ctx& c,
                                              typename P::NodeRegs& r, uint32_t e, const Rec rec, const EvAux aux,
                                              bool toDown, uint8_t toPart, bool moreEvents, long long& nRecv,
                                              long long& bRecv) {
  const uint32_t kind = rec_kind(rec);
  const int32_t from = rec_from(rec);
  c.ev = e;
  c.sub = 0;
  c.draws = 0;
  c.evFlags = 0;
  c.outBase = aux.outBase;
  c.outCap = aux.outCap;
  uint32_t flags = 0;
  KPROF_DECL;
  // :606 — partitions are rare: without cuts every node is in partition 0 and the lookup is skipped
  if (!toDown && (d.nparts == 0 || d.nodes.part[from] == toPart)) {
    KPROF_MARK(d.g, 14);  // event prologue
    if (kind == K_MSG) {
      nRecv++;
      bRecv += P::msg_size(ps, rec.w2);
      flags = EV_DELIVERED | ((uint32_t)P::msg_level(rec.w2) << 24);
      P::on_message(c, ps, r, from, rec.w2, rec.w3);
    } else {
      flags = EV_TASK_RUN;
      P::on_task(c, ps, r, rec.w2, rec.w3);
      if (kind == K_PERIODIC)  // PeriodicTask.action re-arm (C/messages/PeriodicTask.java:39-47)
        c.put(O_PERIODIC, c.node, rec.w2, rec.w3, c.t + (int32_t)rec.w3, 0, false);
    }
  }
  if (aux.chain >= 0 && aux.cpos < 0) {  // last hop of the run: markRead(); if (hasNextReader()) msgs.addMsg(m)  :629-632
    const int32_t next = (aux.cpos & 0x7FFFFFFF) + 1;
    if (next < d.chains[aux.chain].ndest)
      c.put(O_CHAINCONT, aux.chain, (uint32_t)next, 0, 0, 0, false);
    else if (WG_LANE == 0)
      d.chains[aux.chain].flags = 0;  // envelope fully delivered
  }
  KPROF_MARK(d.g, 30);  // the action() (+ a periodic task's re-arm, a chain's continuation)
  if (WG_LANE == 0) {
    EvRes res;
    res.nrec = c.sub | flags | c.evFlags;
    res.ndraw = c.draws;
    gst(d.evRes + e, res);
  }
  // the node's next event reads what this one wrote (other lanes, same wavefront)
  if (moreEvents) __threadfence_block();
  KPROF_MARK(d.g, 15);  // event epilogue (result record, fence)
}

// A node visit is a chain of dependent HBM round trips and the kernel is bound by that latency
// (DESIGN.md §3.1), so the loop is written to keep the chain short:
//   1. active[a]                                  -> node
//   2. head[node] + everything node_begin loads   -> newest event e0, node registers
//   3. evNext[e0] + ev[e0] + evAux[e0]            -> (usually) "e0 is the only event" and the event itself
//   4. whatever action() loads (hoisted inside P) -> stores; counters by no-return atomics
// A protocol may deliver some nodes' events itself before this kernel runs (Casper: nodes whose events of the ms are
// all attestations, applied one lane per event with atomics — they commute); P::visit_skip(d, ps, node) then tells the
// visit to leave the node alone (its events already have their EvRes). Detected at compile time: nothing changes for
// the protocols without the hook.
template <class P, class = void>
struct HasVisitSkip {
  static constexpr bool value = false;
};
template <class P>
struct HasVisitSkip<P, decltype((void)&P::visit_skip)> {
  static constexpr bool value = true;
};

// One node visit: receiveUntil's body for the node's events of this ms, in event order. The node registers are loaded
// (P::node_begin / node_begin_pre) by the caller.
template <class P>
__device__ __forceinline__ void deliver_visit(const EngineDev& d, const typename P::State& ps, Ctx& c,
                                              typename P::NodeRegs& r, const VisitDesc& vd, uint32_t* shSortW) {
  const int lane = WG_LANE;
  const int32_t node = vd.node;
  const int32_t e0 = vd.e0;
  const bool toDown = (vd.flags & VD_DOWN) != 0;
  const uint8_t toPart = (uint8_t)(vd.flags >> 8);
  long long nRecv = 0, bRecv = 0;
  const int32_t next0 = vd.next0;
  const Rec rec0 = vd.rec0;
  const EvAux aux0 = vd.aux0;
  // mode 0: e0 is the only event (usual); 1: <= 64 events, sorted into shSort; 2: more (the PingPong
  // origin): repeated minimum search over the list. One call site of deliver_event for all three.
  int mode = 0;
  uint32_t cnt = 1;
  if (next0 >= 0) {
    int32_t cur = e0;
    uint32_t mine = 0xFFFFFFFFu;
    cnt = 0;
    while (cur >= 0 && cnt < 64) {
      if ((uint32_t)lane == cnt) mine = (uint32_t)cur;
      cur = d.evNext[cur];
      cnt++;
    }
    if (cur < 0) {
      mode = 1;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < cnt; j++) rank += lane_bcast(mine, (int)j) < mine;
      if ((uint32_t)lane < cnt) shSortW[rank] = mine;
      __builtin_amdgcn_wave_barrier();
    } else {
      mode = 2;
      cnt = 0xFFFFFFFFu;
    }
  }
  bool have = false;
  uint32_t last = 0;
  KPROF_DECL;
  KPROF_MARK(d.g, 1);  // descriptor + node_begin (+ the inbox sort of multi-event nodes)
  for (uint32_t k = 0; k < cnt; k++) {
    uint32_t e = (uint32_t)e0;
    if (mode == 1) {
      e = shSortW[k];
    } else if (mode == 2) {
      uint32_t best = 0xFFFFFFFFu;
      for (int32_t q = e0; q >= 0; q = d.evNext[q])
        if ((!have || (uint32_t)q > last) && (uint32_t)q < best) best = (uint32_t)q;
      if (best == 0xFFFFFFFFu) break;
      e = best;
      last = best;
      have = true;
    }
    Rec rec = rec0;
    EvAux aux = aux0;
    if (mode != 0) {
      rec = d.ev[e];
      aux = d.evAux[e];
    }
    deliver_event<P>(d, ps, c, r, e, rec, aux, toDown, toPart, mode == 2 || k + 1 < cnt, nRecv, bRecv);
  }
  __builtin_amdgcn_wave_barrier();
  KPROF_MARK(d.g, 2);  // the events' action()s
  P::node_end(c, ps, r);
  if (lane == 0) {
    // Node counters (C/Node.java:69-79): this wavefront is the node's only writer in this launch;
    // atomics without a return value do not stall the wave the way a load-add-store would
    if (nRecv) {
      atomicAdd((unsigned long long*)&d.nodes.msgReceived[node], (unsigned long long)nRecv);
      atomicAdd((unsigned long long*)&d.nodes.bytesReceived[node], (unsigned long long)bRecv);
    }
    if (c.msgSent) {
      atomicAdd((unsigned long long*)&d.nodes.msgSent[node], (unsigned long long)c.msgSent);
      atomicAdd((unsigned long long*)&d.nodes.bytesSent[node], (unsigned long long)c.bytesSent);
    }
    d.head[node] = -1;
  }
  __builtin_amdgcn_wave_barrier();
  KPROF_MARK(d.g, 3);  // node_end + counters
}

template <class P, class = void>
struct HasNodeCounters {
  static constexpr bool value = false;
};
template <class P>
struct HasNodeCounters<P, decltype((void)&P::node_counters)> {
  static constexpr bool value = true;
};
// ... the same visit for a protocol whose nodes keep their events in inbox lines (EngineDev::inbox): lane k < INBOX_SLOTS
// of the wavefront holds entry k of the node's line (`mine`; beyond `cnt` it is not looked at), the events beyond the
// line hang on the node's overflow list. The usual node has 1..4 events: they are ranked by event index among the four
// lanes and handed to deliver_event straight from the registers — no event record, no EvAux (a task's outbox slice is
// in its entry) and no list is read from memory. `skip`: the node's first events (in event order) that another kernel
// has applied already (a lane-per-node kernel that handed the rest of the visit over).
template <class P>
__device__ __forceinline__ void deliver_visit_inbox(const EngineDev& d, const typename P::State& ps, Ctx& c,
                                                    typename P::NodeRegs& r, int32_t node, uint32_t cnt, uint32_t vflags,
                                                    const InboxEntry& mine, uint32_t skip) {
  const int lane = WG_LANE;
  const bool toDown = (vflags & VD_DOWN) != 0;
  const uint8_t toPart = (uint8_t)(vflags >> 8);
  long long nRecv = 0, bRecv = 0;
  if (cnt <= (uint32_t)INBOX_SLOTS) {
    const uint32_t myE = (uint32_t)lane < cnt ? mine.e : 0xFFFFFFFFu;
    uint32_t rank = 0;
#pragma unroll
    for (int j = 0; j < INBOX_SLOTS; j++) rank += lane_bcast(myE, j) < myE;
    for (uint32_t k = skip; k < cnt; k++) {
      const int src = __ffsll((unsigned long long)__ballot((uint32_t)lane < cnt && rank == k)) - 1;
      const uint32_t e = lane_bcast(mine.e, src), w0 = lane_bcast(mine.w0, src);
      const uint32_t w2 = lane_bcast(mine.w2, src), w3 = lane_bcast(mine.w3, src);
      const uint32_t kind = (w0 >> 28) & 3u;
      Rec rec;
      EvAux aux;
      aux.chain = -1;
      aux.cpos = 0;
      if (kind == K_MSG) {
        rec = make_rec(K_MSG, (int32_t)(w0 & 0x0FFFFFFFu), (uint32_t)node, w2, w3);
        aux.outBase = 0;
        aux.outCap = 0;
        if ((w0 & INBOX_CHAIN) || d.boundMsg) aux = gld(d.evAux + e);
      } else {
        rec = make_rec(kind, node, (uint32_t)node, w2, w3);
        aux.outBase = w0 & 0x0FFFFFFFu;
        aux.outCap = d.boundTask[w2 < 3u ? w2 : 3u] + (kind == K_PERIODIC ? 1u : 0u);
      }
      deliver_event<P>(d, ps, c, r, e, rec, aux, toDown, toPart, k + 1 < cnt, nRecv, bRecv);
    }
  } else {
    // more events than the line holds: the line's four and the overflow list, in event order — by repeated minimum
    // search (the list is unordered; such nodes are rare: a PingPong-style origin), records from the event arrays.
    // (`skip` is honoured in the branch above only: a lane-per-node kernel hands a visit over only for nodes whose events
    // fit the line — anything else here would apply an event twice, so it is an error, not a guess)
    if (skip != 0) {
      if (lane == 0) set_err(d.g, ERR_PROTOCOL);
      return;
    }
    // (the line's four come from the line — expand wrote no event record for them —, the others from the event arrays)
    const uint32_t l0 = lane_bcast(mine.e, 0), l1 = lane_bcast(mine.e, 1), l2 = lane_bcast(mine.e, 2), l3 = lane_bcast(mine.e, 3);
    const int32_t listHead = d.head[node];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) d.head[node] = -1;
    bool have = false;
    uint32_t last = 0;
    for (uint32_t k = 0; k < cnt; k++) {
      uint32_t best = 0xFFFFFFFFu;
      auto consider = [&](uint32_t q) {
        if ((!have || q > last) && q < best) best = q;
      };
      consider(l0);
      consider(l1);
      consider(l2);
      consider(l3);
      for (int32_t q = listHead; q >= 0; q = d.evNext[q]) consider((uint32_t)q);
      if (best == 0xFFFFFFFFu) break;
      last = best;
      have = true;
      Rec rec;
      EvAux aux;
      const int src = best == l0 ? 0 : best == l1 ? 1 : best == l2 ? 2 : best == l3 ? 3 : -1;
      if (src >= 0) {  // (as the branch above builds an event from its line entry)
        const uint32_t w0 = lane_bcast(mine.w0, src), w2 = lane_bcast(mine.w2, src), w3 = lane_bcast(mine.w3, src);
        const uint32_t kind = (w0 >> 28) & 3u;
        aux.chain = -1;
        aux.cpos = 0;
        if (kind == K_MSG) {
          rec = make_rec(K_MSG, (int32_t)(w0 & 0x0FFFFFFFu), (uint32_t)node, w2, w3);
          aux.outBase = 0;
          aux.outCap = 0;
          if ((w0 & INBOX_CHAIN) || d.boundMsg) aux = gld(d.evAux + best);
        } else {
          rec = make_rec(kind, node, (uint32_t)node, w2, w3);
          aux.outBase = w0 & 0x0FFFFFFFu;
          aux.outCap = d.boundTask[w2 < 3u ? w2 : 3u] + (kind == K_PERIODIC ? 1u : 0u);
        }
      } else {
        rec = d.ev[best];
        aux = d.evAux[best];
      }
      deliver_event<P>(d, ps, c, r, best, rec, aux, toDown, toPart, true, nRecv, bRecv);
    }
  }
  __builtin_amdgcn_wave_barrier();
  if constexpr (HasNodeCounters<P>::value) {  // the protocol keeps the Node counters with its own per-node record
    P::node_counters(c, ps, r, nRecv, bRecv);
    P::node_end(c, ps, r);
  } else {
    P::node_end(c, ps, r);
    if (lane == 0) {
      if (nRecv) {
        atomicAdd((unsigned long long*)&d.nodes.msgReceived[node], (unsigned long long)nRecv);
        atomicAdd((unsigned long long*)&d.nodes.bytesReceived[node], (unsigned long long)bRecv);
      }
      if (c.msgSent) {
        atomicAdd((unsigned long long*)&d.nodes.msgSent[node], (unsigned long long)c.msgSent);
        atomicAdd((unsigned long long*)&d.nodes.bytesSent[node], (unsigned long long)c.bytesSent);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// WPE: waves per SIMD the register allocation must admit
template <class P, int WPE>
__global__ void __launch_bounds__(256, WPE) k_deliver(const EngineDev* __restrict__ tab,
                                                      const typename P::State* __restrict__ stab) {
  WG_ENGINE(tab);
  const typename P::State& ps = stab[wgBy];
  __shared__ typename P::WaveShared shP[4];
  __shared__ uint32_t shSort[4][64];
  const int lane = WG_LANE, w = threadIdx.x >> 6;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nActive = d.g->nActive;
  const int32_t t = d.g->now;
  for (uint32_t a = wave; a < nActive; a += nWaves) {
    KPROF_DECL;
    KPROF_COUNT(d.g, 0);
    VisitDesc vd;
    vd.node = (int32_t)d.active[a];
    vd.e0 = d.head[vd.node];  // newest event of the node (always >= 0 for a listed node)
    vd.flags = (d.nodes.down[vd.node] ? VD_DOWN : 0u) | (d.nparts ? (uint32_t)d.nodes.part[vd.node] << 8 : 0u);
    const int32_t node = vd.node;
    if constexpr (HasVisitSkip<P>::value) {
      if (P::visit_skip(d, ps, node)) {
        if (lane == 0) d.head[node] = -1;
        continue;
      }
    }
    Ctx c{d, t, node, 0, 0, 0, 0, 0, 0, 0};
    typename P::NodeRegs r;
    P::node_begin(c, ps, r, &shP[w]);
    vd.next0 = d.evNext[vd.e0];
    vd.rec0 = d.ev[vd.e0];
    vd.aux0 = d.evAux[vd.e0];
    deliver_visit<P>(d, ps, c, r, vd, shSort[w]);
  }
}

// ... the same visit from the node's INBOX LINE (EngineDev::inbox, protocols that ask for it: Engine::wantInbox): the
// node's <= 4 events are one 64-byte read, not a walk of head[node] -> evNext -> ev / evAux gathers — three dependent
// round trips per event of the visit's serial chain. The next node's line and event count are in flight during the
// current visit.
// LISTB: the nodes to visit are the ones a lean kernel of the protocol listed in EngineDev::activeB (32-bit node ids,
// Globals::nActiveB of them) — GSFSignature's k_gsf_lane lists what it and the kernels before it did not take; without the
// list every active node cost this kernel a wavefront's look at its (zeroed) inbox count: 49 of its 221 us per ordinary ms at 256 copies.
template <class P, int WPE, bool LISTB = false>
__global__ void __launch_bounds__(256, WPE) k_deliver_inbox(const EngineDev* __restrict__ tab,
                                                            const typename P::State* __restrict__ stab) {
  WG_ENGINE(tab);
  const typename P::State& ps = stab[wgBy];
  __shared__ typename P::WaveShared shP[4];
  const int lane = WG_LANE, w = threadIdx.x >> 6;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nActive = LISTB ? d.g->nActiveB : d.g->nActive;
  const uint32_t WG_G* nodes = LISTB ? (const uint32_t WG_G*)(VisitDesc WG_G*)d.activeB : (const uint32_t WG_G*)(uint32_t WG_G*)d.active;
  const int32_t t = d.g->now;
  if (wave >= nActive) return;
  auto line_of = [&](int32_t node) -> InboxEntry {  // lane k < 4: entry k of the node's line
    return gld(d.inbox + ((size_t)node * INBOX_SLOTS + (lane < INBOX_SLOTS ? lane : 0)));
  };
  int32_t node = (int32_t)nodes[wave];
  InboxEntry in = line_of(node);
  uint32_t cnt = d.icnt[node];
  for (uint32_t a = wave; a < nActive; a += nWaves) {
    const bool haveNext = a + nWaves < nActive;
    int32_t nodeN = node;
    InboxEntry inN = in;
    uint32_t cntN = cnt;
    if (haveNext) {
      nodeN = (int32_t)nodes[a + nWaves];
      inN = line_of(nodeN);
      cntN = d.icnt[nodeN];
    }
    __builtin_amdgcn_wave_barrier();  // every lane has read the node's count (and the next node's) before lane 0 clears it
    if (cnt != 0) {  // (0: a lane-per-node kernel of the protocol has delivered the node's events already — k_gsf_lane)
      if (lane == 0) d.icnt[node] = 0;  // the line is consumed by this visit
      const uint32_t vflags = (d.nodes.down[node] ? VD_DOWN : 0u) | (d.nparts ? (uint32_t)d.nodes.part[node] << 8 : 0u);
      Ctx c{d, t, node, 0, 0, 0, 0, 0, 0, 0};
      typename P::NodeRegs r;
      KPROF_DECL;
      KPROF_COUNT(d.g, 0);
      P::node_begin(c, ps, r, &shP[w]);
      KPROF_MARK(d.g, 1);  // node_begin
      deliver_visit_inbox<P>(d, ps, c, r, node, cnt, vflags, in, 0u);
      KPROF_MARK(d.g, 2);  // the visit (events + node_end)
    } else {
      KPROF_COUNT(d.g, 3);  // a node another kernel delivered
    }
    node = nodeN;
    in = inN;
    cnt = cntN;
  }
}

// ------------------------------------------------------------------------------------------------
// A payload copy a lane-per-node kernel hands to its whole wavefront (k_handel_lane: SendSigs payloads wider than one
// word are copied coalesced after the lanes' scalar work).
#ifndef WG_COPY_UNROLL
#define WG_COPY_UNROLL 4  // words a lane has in flight per round of the wide-payload copy
#endif
struct CopyJob {
  const uint64_t WG_G* src;
  uint64_t WG_G* dst;
  int32_t nw;
  int32_t pad;
};

// ------------------------------------------------------------------------------------------------
// RunMultipleTimes' inner loop (C/RunMultipleTimes.java:50-64) kept on the device, so that a batch runs
// chunk after chunk without a host round trip per runMs:
//   do { didSomething = runMs(chunk); } while ((maxTime == 0 || time < maxTime) && (!didSomething || contIf(p)));
// k_chunk_begin = the head of Network.runMs (:318-338) for every member still running; k_chunk_end = the
// loop condition (cont[] comes from the protocol's predicate kernel); a member whose loop ended gets
// `halted` in the device table and every later kernel returns at once for it.
__global__ void k_chunk_begin(EngineDev* tab, int32_t ms) {
  EngineDev& d = tab[blockIdx.x];
  if (d.halted || threadIdx.x != 0) return;
  Globals WG_G* g = d.g;
  const int32_t time = g->until;  // Network.time after the previous runMs
  const int32_t endAt = (int32_t)((uint32_t)time + (uint32_t)ms);
  if (endAt <= 0) {  // "Maximum time reached!" (:333) — stop the member; the host reports it
    set_err(g, ERR_ARRIVAL_PAST);
    d.halted = 1;
    return;
  }
  g->epoch++;  // a new receiveUntil() starts with a fresh nextMessage() call
  g->anyEvent = 0;
  g->now = time;
  g->until = endAt;
}
__global__ void k_chunk_end(EngineDev* tab, const uint32_t* cont, int32_t maxTime, uint32_t* running) {
  EngineDev& d = tab[blockIdx.x];
  if (d.halted || threadIdx.x != 0) return;
  const Globals WG_G* g = d.g;
  const bool goOn = (maxTime == 0 || g->until < maxTime) && (!g->anyEvent || cont[blockIdx.x] != 0) && g->err == 0;
  if (!goOn)
    d.halted = 1;
  else
    atomicAdd(running, 1u);
}

}  // namespace wg
