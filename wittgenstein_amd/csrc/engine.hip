// Host runtime of the engine + the PingPong resident protocol (P/PingPong.java).
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <map>
#include <dlfcn.h>
#include <type_traits>
#include "engine_host.h"
#include "engine_kernels.hip.h"

namespace wg {

// ------------------------------------------------------------------------------------------------
// PingPong (P/PingPong.java:20-32,60-74): Ping -> send(Pong) to the sender; Pong -> pong++.
struct PingPongProto {
  struct State {
    GP<int32_t> pong;
  };
  struct WaveShared {
    int unused;
  };
  struct NodeRegs {};
  enum : uint32_t { MSG_PING = 0, MSG_PONG = 1 };
  __device__ static int msg_size(const State&, uint32_t) { return 1; }  // Message.size() default
  __device__ static int msg_level(uint32_t) { return 0; }
  __device__ static void node_begin(Ctx&, const State&, NodeRegs&, WaveShared*) {}
  __device__ static void node_end(Ctx&, const State&, NodeRegs&) {}
  __device__ static void on_message(Ctx& c, const State& s, NodeRegs&, int32_t from, uint32_t msg, uint32_t) {
    if (msg == MSG_PING) {
      c.send(from, MSG_PONG, 0, 1);  // onPing :67-69
    } else if (WG_LANE == 0) {
      s.pong[c.node]++;  // onPong :71-73
    }
  }
  __device__ static void on_task(Ctx&, const State&, NodeRegs&, uint32_t, uint32_t) {}
};

struct PingPongHost : ProtoHost {
  PingPongProto::State st{};
  explicit PingPongHost(Engine& e) {
    st.pong = e.dalloc<int32_t>(e.dev.nodes.n);
    e.dev.boundMsg = 1;  // a delivered Ping emits one Pong
    for (int k = 0; k < 4; k++) e.dev.boundTask[k] = 0;
  }
  void launch_deliver(const Group& g) override {
    hipLaunchKernelGGL((k_deliver<PingPongProto, 4>), dim3(GRID_DELIVER_SMALL, g.R), dim3(256), 0, g.stream, g.tab,
                       (const PingPongProto::State*)g.stab);
  }
  size_t state_size() const override { return sizeof(st); }
  const void* state_host() const override { return &st; }
  bool supports_shards() const override { return true; }  // pong[] is touched for the visited node only
  bool read_i64(Engine& e, int32_t field, int64_t* dst, int32_t n) override {
    if (field != WG_F_PONG) return false;
    std::vector<int32_t> h(n);
    WG_HIP(hipMemcpy(h.data(), st.pong, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) dst[i] = h[i];
    return true;
  }
};
ProtoHost* make_pingpong_host(Engine& e) { return new PingPongHost(e); }

// ------------------------------------------------------------------------------------------------
static uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// ---- the engine's own RCCL communicator --------------------------------------------------------------------------
// librccl is loaded at run time (dlopen), from the ROCm installation this library was built against first: the
// collective must run on the HIP runtime that owns the engine's stream and buffers, and a host process may carry a
// second RCCL (PyTorch bundles one). Only the five entry points used are bound; the types are RCCL's ABI
// (rccl.h: ncclUniqueId = 128 opaque bytes, ncclInt32 = 2, ncclSum = 0, ncclSuccess = 0).
namespace {
struct NcclUniqueId {
  char internal[128];
};
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};
Rccl& rccl() {
  static Rccl r;
  if (r.lib || !r.err.empty()) return r;
  // WG_RCCL_LIB names THE library to load (no fallback: a host that pins its RCCL must not silently get another one)
  std::vector<std::string> names;
  if (const char* p = getenv("WG_RCCL_LIB")) {
    names.push_back(p);
  } else {
    if (const char* p = getenv("ROCM_PATH"))
      if (*p) names.push_back(std::string(p) + "/lib/librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("librccl.so.1");
  }
  std::string why;
  for (const std::string& n : names) {
    r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
    const char* m = dlerror();  // (one call per failure: dlerror() clears the message it returns)
    why += (why.empty() ? "" : "; ") + n + ": " + (m ? m : "?");
  }
  if (!r.lib) {
    r.err = "librccl.so.1 could not be loaded: " + why;
    return r;
  }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");  // (the all-to-all of shard_alltoallv: optional, has_alltoall())
  r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) r.err = "librccl.so.1 lacks the nccl* entry points";
  return r;
}
void rccl_check(int rc, const char* what) {
  if (rc == 0) return;
  Rccl& r = rccl();
  throw WgError(WG_EHIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}
}  // namespace

void rccl_unique_id(uint8_t* id128) {
  Rccl& r = rccl();
  if (!r.err.empty()) throw WgError(WG_EHIP, r.err);
  NcclUniqueId id;
  rccl_check(r.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, id.internal, 128);
}

Engine::Engine(const wg_config& c) : cfg(c) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) throw WgError(WG_EHIP, "no HIP device visible (libwittgpu.so has no CPU fallback)");
  if (cfg.device < 0 || cfg.device >= count) throw WgError(WG_EINVAL, "bad device ordinal");
  WG_HIP(hipSetDevice(cfg.device));
  WG_HIP(hipStreamCreate(&stream));
  gh.rng = lcg_scramble(0);  // new Random(0)  C/Network.java:32
  set_latency(WG_LAT_IC3, nullptr, 0);
  if (cfg.nshards != 0) {
    if (!cfg.allreduce && cfg.rccl_id)
      configure_shard_rccl(cfg.shard, cfg.nshards, cfg.rccl_id);
    else
{
      configure_shard(cfg.shard, cfg.nshards, cfg.allreduce, cfg.allreduce_ctx);
      if (cfg.alltoallv) set_alltoallv(cfg.alltoallv, cfg.alltoallv_ctx);
    }
    cfg.rccl_id = nullptr;  // (the caller's buffer is not kept)
  }
}

Engine::~Engine() {
  delete proto;
  if (rcclComm) (void)rccl().CommDestroy(rcclComm);
  if (mailbox) (void)hipHostFree((void*)mailbox);
  if (sentBuf) (void)hipFree(sentBuf);
  if (snap) {
    if (snap->arena) (void)hipFree(snap->arena);
    delete snap;
  }
  for (void* p : allocs) (void)hipFree(p);
  if (dLut) (void)hipFree(dLut);
  if (dTabDelta) (void)hipFree(dTabDelta);
  if (dTabDist) (void)hipFree(dTabDist);
  for (void* p : {dCity, dCityTab, dCityPing, dCityJit})
    if (p) (void)hipFree(p);
  for (auto& sp : profSpans) {
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  for (auto ev : profFree) (void)hipEventDestroy(ev);
  if (profOwnRef) (void)hipEventDestroy(profOwnRef);
  if (dStamps) (void)hipFree(dStamps);
  if (dStampCnt) (void)hipFree(dStampCnt);
  if (stream) (void)hipStreamDestroy(stream);
}

hipEvent_t Engine::prof_event() {
  if (!profFree.empty()) {
    hipEvent_t ev = profFree.back();
    profFree.pop_back();
    return ev;
  }
  hipEvent_t ev;
  WG_HIP(hipEventCreate(&ev));
  return ev;
}
__global__ void k_prof_stamp(uint64_t* __restrict__ ring, uint32_t* __restrict__ cnt, uint32_t cap, uint32_t tag) {
  const uint32_t i = atomicAdd(cnt, 1u) % cap;
  ring[2 * (size_t)i] = tag;
  ring[2 * (size_t)i + 1] = wall_clock64();
}
void Engine::prof_stamp(int tag) {
  if (!dStamps) {
    WG_HIP(hipMalloc((void**)&dStamps, sizeof(uint64_t) * 2 * PROF_STAMP_CAP));
    WG_HIP(hipMalloc((void**)&dStampCnt, sizeof(uint32_t)));
    WG_HIP(hipMemset(dStampCnt, 0, sizeof(uint32_t)));
  }
  hipLaunchKernelGGL(k_prof_stamp, dim3(1), dim3(1), 0, stream, dStamps, dStampCnt, PROF_STAMP_CAP, (uint32_t)tag);
}
void Engine::prof_collect() {
  if (dStampCnt) {  // the records of the replayed graphs: a begin stamp of a class, then its end stamp
    uint32_t n = 0;
    WG_HIP(hipMemcpy(&n, dStampCnt, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n > PROF_STAMP_CAP) n = PROF_STAMP_CAP;  // (the ring wrapped: the oldest records are gone, the rest still pair up)
    std::vector<uint64_t> h(2 * (size_t)n);
    if (n) WG_HIP(hipMemcpy(h.data(), dStamps, sizeof(uint64_t) * 2 * n, hipMemcpyDeviceToHost));
    WG_HIP(hipMemset(dStampCnt, 0, sizeof(uint32_t)));
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, cfg.device);
    const double nsPerTick = 1e6 / (double)(khz > 0 ? khz : 100000);
    for (uint32_t i = 0; i + 1 < n; i++) {
      const uint64_t a = h[2 * (size_t)i], b = h[2 * (size_t)i + 2];
      if ((a & 1) || b != a + 1 || a / 2 >= PC_COUNT) continue;
      profNs[a / 2] += (double)(h[2 * (size_t)i + 3] - h[2 * (size_t)i + 1]) * nsPerTick;
      profLaunches[a / 2]++;
      i++;
    }
  }
  for (auto& sp : profSpans) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
      profNs[sp.cls] += (double)ms * 1e6;
      profLaunches[sp.cls]++;
      float t0 = 0;
      if (profRef && hipEventElapsedTime(&t0, profRef, sp.a) == hipSuccess)
        profTimes[sp.cls].push_back({(double)t0 * 1e6, ((double)t0 + (double)ms) * 1e6});
    }
    profFree.push_back(sp.a);
    profFree.push_back(sp.b);
  }
  profSpans.clear();
}

void Engine::add_nodes(int32_t n, const int32_t* x, const int32_t* y, const int32_t* extra, const uint8_t* down,
                       const uint8_t* byz, const double* speed) {
  if (allocated) throw WgError(WG_ESTATE, "nodes cannot be added after the first run / protocol load");
  if (n < 0 || !x || !y) throw WgError(WG_EINVAL, "wg_add_nodes: n/x/y");
  for (int i = 0; i < n; i++) {
    if (x[i] <= 0 || x[i] > 2000) throw WgError(WG_EINVAL, "bad x=" + std::to_string(x[i]));  // C/Node.java:256-261
    if (y[i] <= 0 || y[i] > 1112) throw WgError(WG_EINVAL, "bad y=" + std::to_string(y[i]));
    if (speed && speed[i] <= 0) throw WgError(WG_EINVAL, "speedRatio");
    hx.push_back(x[i]);
    hy.push_back(y[i]);
    hextra.push_back(extra ? extra[i] : 0);
    hdown.push_back(down ? down[i] : 0);
    hbyz.push_back(byz ? byz[i] : 0);
    hspeed.push_back(speed ? speed[i] : 1.0);
  }
}

// ---- latency models -> tables (C/NetworkLatency.java). All FP happens here, once, on the host.
static double gpd_inverseF(double y) {  // GeneralizedParetoDistribution(1.4, -0.3, 0.35).inverseF
  const double shape = 1.4, location = -0.3, scale = 0.35;
  if (y < 0.000001) return location;
  if (y > 0.999999) return INFINITY;
  return location + scale / shape * (-1 + std::pow(1 - y, -shape));
}
static const int kMaxDist = 1144;  // Node.MAX_DIST = (int) sqrt(1000^2 + 556^2)

void Engine::set_latency(int32_t kind, const int32_t* params, int32_t nparams) {
  if (queue_size() != 0)
    throw WgError(WG_ESTATE, "You can't change the latency while the system as on going messages");
  lutDist.clear();
  tabDelta.clear();
  tabDist.clear();
  latParam = 0;
  switch (kind) {
    case WG_LAT_BY_DISTANCE_WJITTER: {  // :49-73
      lutDist.resize((kMaxDist + 1) * 100);
      const double pointValue = (24860.0 / 2) / kMaxDist;
      for (int dist = 0; dist <= kMaxDist; dist++)
        for (int delta = 0; delta < 100; delta++) {
          double raw = (pointValue * dist) * 0.022 + 4.862 + gpd_inverseF(delta / 100.0);
          int v = (int)(raw / 2);
          if (v < 0 || v > 255) throw WgError(WG_ESTATE, "latency LUT out of u8 range");
          lutDist[dist * 100 + delta] = (uint8_t)v;
        }
      break;
    }
    case WG_LAT_FIXED:
      if (nparams < 1) throw WgError(WG_EINVAL, "NetworkFixedLatency needs 1 parameter");
      latParam = std::max(1, params[0]);
      break;
    case WG_LAT_UNIFORM: {
      if (nparams < 1) throw WgError(WG_EINVAL, "NetworkUniformLatency needs 1 parameter");
      latParam = std::max(1, params[0]);
      tabDelta.resize(100);
      for (int d = 0; d < 100; d++) tabDelta[d] = (int)((d / 99.0) * latParam);
      break;
    }
    case WG_LAT_NONE: break;
    case WG_LAT_MEASURED:
      if (nparams != 100) throw WgError(WG_EINVAL, "MeasuredNetworkLatency needs longDistrib[100]");
      tabDelta.assign(params, params + 100);
      break;
    case WG_LAT_ETHSCAN: {  // :366-384 over MeasuredNetworkLatency.setLatency :284-302
      static const int prop[] = {16, 18, 17, 12, 8, 5, 4, 3, 3, 1, 1, 2, 1, 1, 8};
      static const int val[] = {250, 500, 1000, 1250, 1500, 1750, 2000, 2250, 2500, 2750, 4500, 6000, 8500, 9750, 10000};
      int cur = 0;
      for (int i = 0; i < 15; i++) {
        int step = (val[i] - cur) / prop[i];
        for (int k = 0; k < prop[i]; k++) {
          cur += step;
          tabDelta.push_back(cur);
        }
      }
      break;
    }
    case WG_LAT_IC3: {  // :399-417
      tabDist.resize(kMaxDist + 1);
      for (int dist = 0; dist <= kMaxDist; dist++) {
        double dd = dist;
        double surface = dd * dd * M_PI;
        double totalSurface = 2000 * 1112;
        int position = (int)((surface * 100) / totalSurface);
        int v;
        if (position <= 10) v = 92 / 2;
        else if (position <= 33) v = 125 / 2;
        else if (position <= 50) v = 152 / 2;
        else if (position <= 67) v = 200 / 2;
        else if (position <= 90) v = 276 / 2;
        else v = 350 / 2;
        tabDist[dist] = v;
      }
      break;
    }
    default: throw WgError(WG_EINVAL, "unknown latency kind");
  }
  latKind = kind;
  if (allocated) {
    upload_latency();
    dev.saBins = sendall_bins();
  }
}

void Engine::set_latency_city(int32_t mode, int32_t nC, const int32_t* cityOfNode, const int32_t* tab, const float* ping,
                              const double* jit) {
  if (queue_size() != 0) throw WgError(WG_ESTATE, "You can't change the latency while the system as on going messages");
  const size_t n = hx.size();
  if (mode < 0 || mode > 2 || nC <= 0 || nC > 65535 || !cityOfNode || n == 0)
    throw WgError(WG_EINVAL, "wg_set_latency_city: mode / n_cities / city_of_node (add the nodes first)");
  if ((mode != WG_CITY_BY_CITY_WJITTER && !tab) || (mode == WG_CITY_BY_CITY_WJITTER && !ping) || (mode != WG_CITY_BY_CITY && !jit))
    throw WgError(WG_EINVAL, "wg_set_latency_city: a table this mode needs is NULL");
  cityOf.resize(n);
  for (size_t i = 0; i < n; i++) {
    if (cityOfNode[i] < 0 || cityOfNode[i] >= nC) throw WgError(WG_EINVAL, "wg_set_latency_city: city index out of range");
    cityOf[i] = (uint16_t)cityOfNode[i];
  }
  const size_t cc = (size_t)nC * nC;
  cityTab.assign(tab ? tab : nullptr, tab ? tab + cc : nullptr);
  cityPing.assign(ping ? ping : nullptr, ping ? ping + cc : nullptr);
  cityJit.assign(jit ? jit : nullptr, jit ? jit + 100 : nullptr);
  nCities = nC;
  lutDist.clear();
  tabDelta.clear();
  tabDist.clear();
  latKind = LAT_CITY;
  latParam = mode;
  if (allocated) {
    upload_latency();
    dev.saBins = sendall_bins();
  }
}

void Engine::set_latency_by_name(const char* name) {  // C/RegistryNetworkLatencies.java:42-58
  std::string s = name ? name : "";
  int32_t p;
  if (s.empty() || s == "NetworkLatencyByDistanceWJitter") return set_latency(WG_LAT_BY_DISTANCE_WJITTER, nullptr, 0);
  if (s.rfind("NetworkFixedLatency(", 0) == 0) {
    p = atoi(s.c_str() + 20);
    return set_latency(WG_LAT_FIXED, &p, 1);
  }
  if (s.rfind("NetworkUniformLatency(", 0) == 0) {
    p = atoi(s.c_str() + 22);
    return set_latency(WG_LAT_UNIFORM, &p, 1);
  }
  if (s == "NetworkNoLatency") return set_latency(WG_LAT_NONE, nullptr, 0);
  if (s == "IC3NetworkLatency") return set_latency(WG_LAT_IC3, nullptr, 0);
  if (s == "EthScanNetworkLatency") return set_latency(WG_LAT_ETHSCAN, nullptr, 0);
  throw WgError(WG_EINVAL, "latency model not available on the device engine: " + s);
}

void Engine::upload_latency() {
  auto up = [&](void*& dptr, const void* src, size_t bytes) {
    if (dptr) {
      WG_HIP(hipFree(dptr));
      dptr = nullptr;
    }
    if (bytes == 0) return;
    WG_HIP(hipMalloc(&dptr, bytes));
    WG_HIP(hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice));
  };
  up(dLut, lutDist.data(), lutDist.size());
  up(dTabDelta, tabDelta.data(), tabDelta.size() * sizeof(int32_t));
  up(dTabDist, tabDist.data(), tabDist.size() * sizeof(int32_t));
  dev.lat.kind = latKind;
  dev.lat.param = latParam;
  dev.lat.lutDist = (const uint8_t*)dLut;
  dev.lat.tabDelta = (const int32_t*)dTabDelta;
  dev.lat.tabDist = (const int32_t*)dTabDist;
  up(dCity, cityOf.data(), cityOf.size() * sizeof(uint16_t));
  up(dCityTab, cityTab.data(), cityTab.size() * sizeof(int32_t));
  up(dCityPing, cityPing.data(), cityPing.size() * sizeof(float));
  up(dCityJit, cityJit.data(), cityJit.size() * sizeof(double));
  dev.lat.city = (const uint16_t*)dCity;
  dev.lat.cityTab = (const int32_t*)dCityTab;
  dev.lat.cityPing = (const float*)dCityPing;
  dev.lat.cityJit = (const double*)dCityJit;
  dev.lat.nCities = nCities;
}

int32_t Engine::host_latency(int32_t from, int32_t to, int32_t seed) const {
  LatencyModel m;
  m.kind = latKind;
  m.param = latParam;
  m.lutDist = lutDist.data();
  m.tabDelta = tabDelta.data();
  m.tabDist = tabDist.data();
  m.city = cityOf.data();
  m.cityTab = cityTab.data();
  m.cityPing = cityPing.data();
  m.cityJit = cityJit.data();
  m.nCities = nCities;
  return latency_of(m, from, to, hx[from], hy[from], hextra[from], hx[to], hy[to], hextra[to], pseudo_delta(to, seed));
}

int32_t Engine::part_of(int32_t x) const {  // partitionId  C/Network.java:639-649
  int p = 0;
  for (int c : cuts) {
    if (c > x) return p;
    p++;
  }
  return p;
}

void Engine::rebuild_partitions() {
  dev.nparts = (uint32_t)cuts.size();
  if (!allocated) return;
  std::vector<uint8_t> part(hx.size());
  for (size_t i = 0; i < hx.size(); i++) part[i] = (uint8_t)part_of(hx[i]);
  WG_HIP(hipMemcpy(dev.nodes.part, part.data(), part.size(), hipMemcpyHostToDevice));
  geoStale = true;
  upload_geo();
}

// NodeArrays::geo from the host's copies of x / y / extraLatency / down and the cuts
void Engine::upload_geo() {
  if (!geoStale || !allocated) return;
  std::vector<NodeGeo> g(hx.size());
  for (size_t i = 0; i < hx.size(); i++) {
    NodeGeo q;
    memset(&q, 0, sizeof(q));
    q.x = (int16_t)hx[i];
    q.y = (int16_t)hy[i];
    q.extraLatency = hextra[i];
    q.down = hdown[i] ? 1 : 0;
    q.part = (uint8_t)part_of(hx[i]);
    g[i] = q;
  }
  WG_HIP(hipMemcpyAsync(dev.nodes.geo, g.data(), g.size() * sizeof(NodeGeo), hipMemcpyHostToDevice, stream));
  WG_HIP(hipStreamSynchronize(stream));
  geoStale = false;
}

void Engine::set_partitions(const int32_t* c, int32_t k) {
  if (k < 0 || k > MAX_CUTS) throw WgError(WG_EINVAL, "too many partitions");
  cuts.assign(c, c + k);
  std::sort(cuts.begin(), cuts.end());
  rebuild_partitions();
}

void Engine::set_node_down(int32_t id, bool down) {
  if (id < 0 || id >= (int)hx.size()) throw WgError(WG_EINVAL, "node id");
  hdown[id] = down;
  downDirty = true;  // uploaded as one array before the next launch (a stop()ped population is thousands of calls)
}
void Engine::upload_down() {
  if (downDirty && allocated) {
    WG_HIP(hipMemcpyAsync(dev.nodes.down, hdown.data(), hdown.size(), hipMemcpyHostToDevice, stream));
    WG_HIP(hipStreamSynchronize(stream));
    downDirty = false;
    geoStale = true;
  }
  upload_geo();
}

// the largest latency the current model can return for these nodes (sizes the bucket ring and the sendAll histograms)
int32_t Engine::max_latency() const {
  int32_t maxExtra = 0;
  for (int v : hextra) maxExtra = std::max(maxExtra, v);
  int32_t maxLat = 1;
  switch (latKind) {
    case LAT_BYDIST: maxLat = *std::max_element(lutDist.begin(), lutDist.end()) + 2 * maxExtra; break;
    case LAT_FIXED:
    case LAT_UNIFORM: maxLat = latParam + 2 * maxExtra; break;
    case LAT_NONE: maxLat = 1 + 2 * maxExtra; break;
    case LAT_MEASURED: maxLat = *std::max_element(tabDelta.begin(), tabDelta.end()) + 2 * maxExtra; break;
    case LAT_ETHSCAN: maxLat = *std::max_element(tabDelta.begin(), tabDelta.end()) + 4 * maxExtra; break;
    case LAT_IC3: maxLat = 175 + 2 * maxExtra; break;
    case LAT_CITY: {  // the largest table entry (+ the largest jitter where the mode adds one)
      double jmax = 0;
      for (double j : cityJit) jmax = std::max(jmax, j);
      double m = 1;
      for (int32_t v : cityTab) m = std::max(m, (double)v + (latParam == 0 ? jmax : 0.0));
      for (float v : cityPing) m = std::max(m, 0.5 * ((double)v + jmax) + 1.0);
      if (latParam == 2) m = std::max(m, 0.5 * (10.0 + jmax) + 1.0);
      maxLat = (int32_t)std::ceil(m) + 2 * maxExtra;
      break;
    }
  }
  return maxLat;
}
// latency bins of the sendAll histograms: every latency the model can return, not the whole bucket ring
uint32_t Engine::sendall_bins() const {
  const uint32_t want = ((uint32_t)std::max(1, max_latency()) + 1u + 63u) & ~63u;
  return std::min<uint32_t>((uint32_t)dev.horizon, want);
}

// ---- device allocation
void Engine::ensure_device() {
  if (allocated) return;
  const int32_t n = (int32_t)hx.size();
  if (n == 0) throw WgError(WG_ESTATE, "no nodes in the network");
  const int32_t maxLat = max_latency();
  uint32_t D = cfg.horizon_ms > 0 ? (uint32_t)cfg.horizon_ms
                                  : next_pow2((uint32_t)std::max({256, maxLat + 8 + horizonExtra, horizonFloor}));
  if ((D & (D - 1)) != 0 || D > 32768) throw WgError(WG_EINVAL, "horizon_ms must be a power of two <= 32768");
  binBits = 0;
  while ((1u << binBits) < D) binBits++;

  uint64_t poolRecs = cfg.bucket_pool_records > 0 ? (uint64_t)cfg.bucket_pool_records
                                                   : std::max<uint64_t>(1u << 20, 256ull * n);
  uint32_t nPages = (uint32_t)((poolRecs + PAGE_RECS - 1) / PAGE_RECS) + D;  // + one partial page per bucket
  uint32_t maxOut = cfg.outbox_records > 0 ? (uint32_t)cfg.outbox_records : (uint32_t)std::max<uint64_t>(1u << 16, 24ull * n);
  uint32_t chainSlots = cfg.chain_slots > 0 ? (uint32_t)cfg.chain_slots : (uint32_t)std::max<uint64_t>(4096, 16ull * n);
  uint64_t chainDests = cfg.chain_dests > 0 ? (uint64_t)cfg.chain_dests : std::max<uint64_t>(1u << 20, 256ull * n);
  uint64_t payloadWords = cfg.payload_words > 0 ? (uint64_t)cfg.payload_words : (1u << 16);

  dev.g = dalloc<Globals>(1);
#ifdef WG_KPROF
  gh.kprofBuf = dalloc<unsigned long long>((size_t)KPROF_WAVES * 32);
#endif
  NodeArrays& nd = dev.nodes;
  nd.n = n;
  nd.x = dalloc<int16_t>(n);
  nd.y = dalloc<int16_t>(n);
  nd.extraLatency = dalloc<int32_t>(n);
  nd.down = dalloc<uint8_t>(n);
  nd.part = dalloc<uint8_t>(n);
  nd.msgReceived = dalloc<long long>(n);
  nd.msgSent = dalloc<long long>(n);
  nd.bytesSent = dalloc<long long>(n);
  nd.bytesReceived = dalloc<long long>(n);
  nd.doneAt = dalloc<long long>(n);
  std::vector<int16_t> x16(n), y16(n);
  for (int i = 0; i < n; i++) {
    x16[i] = (int16_t)hx[i];
    y16[i] = (int16_t)hy[i];
  }
  WG_HIP(hipMemcpy(nd.x, x16.data(), n * 2, hipMemcpyHostToDevice));
  WG_HIP(hipMemcpy(nd.y, y16.data(), n * 2, hipMemcpyHostToDevice));
  WG_HIP(hipMemcpy(nd.extraLatency, hextra.data(), n * 4, hipMemcpyHostToDevice));
  WG_HIP(hipMemcpy(nd.down, hdown.data(), n, hipMemcpyHostToDevice));
  nd.geo = dalloc<NodeGeo>(n);
  geoStale = true;  // (uploaded by upload_geo once `allocated` is set: rebuild_partitions / upload_down / self())

  dev.discardTime = discardTime;
  dev.horizon = (int32_t)D;
  dev.nPages = nPages;
  dev.maxPagesPerBucket = (int32_t)std::min<uint32_t>(nPages, 4096);
  dev.pool = dalloc<Rec>((size_t)nPages * PAGE_RECS, false);
  dev.freeStack = dalloc<uint32_t>(nPages, false);
  dev.pagetab = dalloc<uint32_t>((size_t)D * dev.maxPagesPerBucket);
  dev.bcnt = dalloc<uint32_t>(D);
  {
    std::vector<uint32_t> fs(nPages);
    for (uint32_t i = 0; i < nPages; i++) fs[i] = nPages - 1 - i;
    WG_HIP(hipMemcpy(dev.freeStack, fs.data(), sizeof(uint32_t) * nPages, hipMemcpyHostToDevice));
  }
  gh.freeTop = nPages;
  dev.chainSlots = chainSlots;
  dev.chains = dalloc<Chain>(chainSlots);
  dev.chainDests = chainDests;
  dev.dests = dalloc<int32_t>(chainDests, false);
  dev.payloadWords = payloadWords;
  dev.payload = dalloc<uint64_t>(payloadWords, false);
  dev.destHeadAt = dalloc<unsigned long long>(D);
  dev.payloadHeadAt = dalloc<unsigned long long>(D);

  dev.maxEvents = maxOut;
  dev.ev = dalloc<Rec>(maxOut, false, AC_SCRATCH);
  dev.evAux = dalloc<EvAux>(maxOut, false, AC_SCRATCH);
  dev.evRes = dalloc<EvRes>(maxOut, true, AC_SCRATCH);   // (per-ms scratch holds nothing between milliseconds:
  dev.evRecOff = dalloc<uint32_t>(maxOut, true, AC_SCRATCH);  //  not part of the init() image, wg_snapshot_bytes)

  dev.evDrawOff = dalloc<uint32_t>(maxOut, true, AC_SCRATCH);
  dev.evNext = dalloc<int32_t>(maxOut, false, AC_SCRATCH);
  dev.head = dalloc<int32_t>(n, false);
  WG_HIP(hipMemsetAsync(dev.head, 0xFF, sizeof(int32_t) * (size_t)n, stream));
  dev.inbox = nullptr;
  dev.icnt = nullptr;
  if (wantInbox) {  // (per-ms scratch: empty between milliseconds — the delivery pass resets the counts it consumes)
    dev.inbox = dalloc<InboxEntry>((size_t)n * INBOX_SLOTS, false, AC_SCRATCH);
    dev.icnt = dalloc<uint32_t>(n, true);
  }
  dev.active = dalloc<uint32_t>(n, true, AC_SCRATCH);
  dev.activeB = dalloc<VisitDesc>(n, true, AC_SCRATCH);
  dev.maxOut = maxOut;
  dev.outTmp = dalloc<Out>(maxOut, false, AC_SCRATCH);
  dev.recEv = dalloc<uint32_t>(maxOut, false, AC_SCRATCH);
  dev.fin = dalloc<Rec>(maxOut, false, AC_SCRATCH);
  dev.arr = dalloc<int32_t>(maxOut, false, AC_SCRATCH);
  maxTiles = (maxOut + TILE - 1) / TILE;
  dev.tileHist = dalloc<uint32_t>((size_t)maxTiles * D);  // zero between phases (k_scatter re-zeroes its rows)
  dev.binBase = dalloc<uint32_t>(D, true, AC_SCRATCH);
  dev.scanPartials = dalloc<unsigned long long>(SCAN_GRID, true, AC_SCRATCH);
  dev.farBuf = nullptr;
  dev.farCap = 0;
  if (farCapacity > 0) {
    dev.farCap = (uint32_t)farCapacity;
    dev.farBuf = dalloc<FarRec>(dev.farCap, false);
  }
  dev.saDesc = nullptr;
  dev.saLat = nullptr;
  dev.saHist = nullptr;
  dev.maxSendAll = 0;
  if (sendAllCapacity > 0) {  // (a resident protocol whose action()s call Network.sendAll asked for it)
    dev.maxSendAll = (uint32_t)sendAllCapacity;
    dev.saDesc = dalloc<SendAllDesc>(dev.maxSendAll, true, AC_SCRATCH);
    dev.saLat = dalloc<int32_t>((size_t)dev.maxSendAll * n, false, AC_SCRATCH);
    dev.saHist = dalloc<uint32_t>((size_t)dev.maxSendAll * ((n + TILE - 1) / TILE) * D, false, AC_SCRATCH);
  }
  dev.saBins = sendall_bins();
  // long chain runs (EngineDev::runs): worth a wavefront each where envelopes reach every node — the resident
  // protocols that call Network.sendAll, from 8 hops on (WG_RUN_MIN=<hops> overrides, 0 = unroll every run in place)
  dev.runs = nullptr;
  dev.maxRuns = 0;
  dev.runMin = sendAllCapacity > 0 ? 8u : 0u;  // (profiles/r02i_sweep_run_min_casper.txt: 4..16 within 2 %, 64 is 12 % slower)
  if (const char* rm = getenv("WG_RUN_MIN")) dev.runMin = (uint32_t)std::max(0, atoi(rm));
  if (dev.runMin) {
    dev.maxRuns = dev.chainSlots;
    dev.runs = dalloc<RunDesc>(dev.maxRuns, false, AC_SCRATCH);
  }
  dev.xcdPlace = !(getenv("WG_XCD_PLACE") && atoi(getenv("WG_XCD_PLACE")) == 0);  // (read once per engine; the A/B switch of wg_place)
  dev.sharded = shardCount > 0 ? 1u : 0u;
  dev.shardLo = 0;
  dev.shardHi = INT32_MAX;
  dev.xbuf = nullptr;
  dev.xmulti = nullptr;
  dev.xev = nullptr;
  dev.maxMulti = 0;
  dev.multiK = dev.multiOff = nullptr;
  dev.sdests = dev.dests;
  dev.sdestCap = dev.chainDests;
  dev.arvTmp = dalloc<int32_t>(chainDests, false, AC_SCRATCH);
  if (shardCount > 0) {
    dev.shardLo = (int32_t)((int64_t)n * shardIndex / shardCount);
    dev.shardHi = (int32_t)((int64_t)n * (shardIndex + 1) / shardCount);
    dev.xbuf = dalloc<int32_t>((size_t)maxOut * 5 + XB_HEAD) + XB_HEAD;
    dev.maxMulti = std::max<uint32_t>(1024, maxOut / 16);
    dev.xmulti = dalloc<int32_t>((size_t)dev.maxMulti * XM_WORDS);
    dev.xev = dalloc<int32_t>(maxOut, false, AC_SCRATCH);
    dev.multiK = dalloc<uint32_t>(maxOut, false, AC_SCRATCH);
    dev.multiOff = dalloc<uint32_t>(maxOut, false, AC_SCRATCH);
    dev.sdests = dalloc<int32_t>(chainDests, false);  // private scratch for unsorted destination lists
  }
  allocated = true;
  upload_latency();
  rebuild_partitions();
  globalsDirty = true;
  WG_HIP(hipStreamSynchronize(stream));
}

void Engine::sync_globals_to_device() {
  upload_down();
  if (!globalsDirty) return;
  WG_HIP(hipMemcpyAsync(dev.g, &gh, sizeof(Globals), hipMemcpyHostToDevice, stream));
  WG_HIP(hipStreamSynchronize(stream));
  globalsDirty = false;
}
void Engine::sync_globals_to_host() {
#ifdef WG_KPROF
  if (gh.kprofBuf) {
    std::vector<unsigned long long> rows((size_t)KPROF_WAVES * 32);
    WG_HIP(hipMemcpy(rows.data(), gh.kprofBuf, 8 * rows.size(), hipMemcpyDeviceToHost));
    WG_HIP(hipMemset(gh.kprofBuf, 0, 8 * rows.size()));
    for (size_t i = 0; i < rows.size(); i++) kprofSum[i & 31] += rows[i];
  }
#endif
  WG_HIP(hipMemcpyAsync(&gh, dev.g, sizeof(Globals), hipMemcpyDeviceToHost, stream));
  WG_HIP(hipStreamSynchronize(stream));
}

// ---- wg_snapshot / wg_restore: the engine as init() left it, kept on the device --------------------------------
// RunMultipleTimes re-creates and re-initialises the protocol for every run (`p.copy(); rd.setSeed(i); init()`,
// C/RunMultipleTimes.java:44-48); init() is sequential host work (Handel: N cumulative shuffles, P/Handel.java:940-948).
// For a seed that was initialised once, restoring the image is the same state at the cost of a device-to-device copy.
// Valid between init() and the first event (nothing polled yet): the protocol hosts rely on that to leave their
// queues / snapshot rings / payload slabs — empty at that point — out of the image (AC_SCRATCH).
void Engine::snapshot() {
  if (!proto) throw WgError(WG_ESTATE, "no resident protocol loaded");
  if (dev.hostMode) throw WgError(WG_EUNSUPPORTED, "host-callback mode keeps its envelopes with the caller");
  ensure_device();
  flush_staged(time, false);
  sync_globals_to_device();
  WG_HIP(hipStreamSynchronize(stream));
  sync_globals_to_host();
  if (gh.events != 0 || gh.delivered != 0 || gh.tasks != 0)
    throw WgError(WG_ESTATE, "wg_snapshot is taken after init() and before the first event is polled");
  if (!stagedChains.empty() || !pendingSent.empty()) throw WgError(WG_ESTATE, "host-staged sends left after a flush");
  if (snap) {
    if (snap->arena) (void)hipFree(snap->arena);
    delete snap;
    snap = nullptr;
  }
  Snapshot* sn = new Snapshot();
  sn->nAllocs = allocs.size();
  sn->offs.assign(allocs.size(), (size_t)-1);
  size_t total = 0;
  // the bucket pool by its pages in use: every bucket's page list is its first ceil(count / PAGE_RECS) table entries
  {
    const uint32_t D = (uint32_t)dev.horizon;
    std::vector<uint32_t> cnt(D), tab((size_t)D * dev.maxPagesPerBucket);
    WG_HIP(hipMemcpy(cnt.data(), dev.bcnt, 4 * (size_t)D, hipMemcpyDeviceToHost));
    WG_HIP(hipMemcpy(tab.data(), dev.pagetab, 4 * tab.size(), hipMemcpyDeviceToHost));
    for (uint32_t b = 0; b < D; b++)
      for (uint32_t k = 0; k < (cnt[b] + PAGE_RECS - 1) / PAGE_RECS; k++) {
        sn->poolPages.push_back({tab[(size_t)b * dev.maxPagesPerBucket + k], total});
        total += sizeof(Rec) * (size_t)PAGE_RECS;
      }
  }
  sn->chainsZero = gh.chainHead == 0 && gh.destHead == 0;
  for (size_t i = 0; i < allocs.size(); i++) {
    if (allocInfo[i].cls != AC_STATE) continue;
    if (allocs[i] == (void*)dev.payload && gh.payloadHead == 0) continue;  // nothing allocated in the ring yet
    if (allocs[i] == (void*)dev.pool) continue;                             // (by pages, above)
    if (sn->chainsZero && (allocs[i] == (void*)dev.chains || allocs[i] == (void*)dev.dests)) continue;  // no envelope yet
    sn->offs[i] = total;
    total += (allocInfo[i].bytes + 255) & ~(size_t)255;
  }
  sn->bytes = total;
  if (hipMalloc((void**)&sn->arena, total > 0 ? total : 16) != hipSuccess) {
    delete sn;
    throw WgError(WG_ENOMEM, "wg_snapshot: no device memory for the image (" + std::to_string(total >> 20) + " MiB)");
  }
  for (size_t i = 0; i < allocs.size(); i++)
    if (sn->offs[i] != (size_t)-1)
      WG_HIP(hipMemcpyAsync(sn->arena + sn->offs[i], allocs[i], allocInfo[i].bytes, hipMemcpyDeviceToDevice, stream));
  for (const auto& pg : sn->poolPages)
    WG_HIP(hipMemcpyAsync(sn->arena + pg.second, dev.pool + ((size_t)pg.first << PAGE_SHIFT), sizeof(Rec) * (size_t)PAGE_RECS,
                          hipMemcpyDeviceToDevice, stream));
  WG_HIP(hipStreamSynchronize(stream));
  sn->gh = gh;
  sn->time = time;
  sn->discardTime = discardTime;
  sn->stagedMin = stagedMin;
  sn->hdown = hdown;
  sn->cuts = cuts;
  sn->staged = staged;
  snap = sn;
}

void Engine::restore() {
  if (!snap) throw WgError(WG_ESTATE, "wg_restore without a wg_snapshot");
  Snapshot& sn = *snap;
  WG_HIP(hipStreamSynchronize(stream));
  for (size_t i = 0; i < sn.nAllocs; i++)
    if (sn.offs[i] != (size_t)-1)
      WG_HIP(hipMemcpyAsync(allocs[i], sn.arena + sn.offs[i], allocInfo[i].bytes, hipMemcpyDeviceToDevice, stream));
  for (const auto& pg : sn.poolPages)  // (the page table, the free stack and the bucket counts came back above: the same pages are in use)
    WG_HIP(hipMemcpyAsync(dev.pool + ((size_t)pg.first << PAGE_SHIFT), sn.arena + pg.second, sizeof(Rec) * (size_t)PAGE_RECS,
                          hipMemcpyDeviceToDevice, stream));
  if (sn.chainsZero)  // no envelope existed at the image: none is busy (a run that was cut short leaves busy slots behind)
    WG_HIP(hipMemsetAsync(dev.chains, 0, sizeof(Chain) * (size_t)dev.chainSlots, stream));
  const uint32_t notes = gh.notes;  // (what the run being undone left: the protocol's on_restore may need it)
  gh = sn.gh;
  gh.notes = notes;
  time = sn.time;
  discardTime = sn.discardTime;
  dev.discardTime = sn.discardTime;
  stagedMin = sn.stagedMin;
  hdown = sn.hdown;
  downDirty = false;  // (the device array comes back with the image)
  if (cuts != sn.cuts) {
    cuts = sn.cuts;
    rebuild_partitions();
  }
  staged = sn.staged;
  stagedChains.clear();
  pendingSent.clear();
  dev.halted = 0;
  if (proto) proto->on_restore(*this);
  gh.notes = sn.gh.notes;
  globalsDirty = true;
  sync_globals_to_device();
  lastError.clear();
}

void Engine::latency_probe(int32_t n, const int32_t* from, const int32_t* to, const int32_t* delta, int32_t* out) {
  ensure_device();
  for (int i = 0; i < n; i++)
    if (from[i] < 0 || from[i] >= dev.nodes.n || to[i] < 0 || to[i] >= dev.nodes.n || delta[i] < 0 || delta[i] > 99)
      throw WgError(WG_EINVAL, "delta=" + std::to_string(delta[i]));  // checkDelta  C/NetworkLatency.java:21-25
  int32_t *df, *dt, *dd, *dout;
  WG_HIP(hipMalloc((void**)&df, 4 * n));
  WG_HIP(hipMalloc((void**)&dt, 4 * n));
  WG_HIP(hipMalloc((void**)&dd, 4 * n));
  WG_HIP(hipMalloc((void**)&dout, 4 * n));
  WG_HIP(hipMemcpy(df, from, 4 * n, hipMemcpyHostToDevice));
  WG_HIP(hipMemcpy(dt, to, 4 * n, hipMemcpyHostToDevice));
  WG_HIP(hipMemcpy(dd, delta, 4 * n, hipMemcpyHostToDevice));
  Group g = self();
  hipLaunchKernelGGL(k_latency_probe, dim3((n + 255) / 256), dim3(256), 0, stream, g.tab, n, df, dt, dd, dout);
  WG_HIP(hipStreamSynchronize(stream));
  WG_HIP(hipMemcpy(out, dout, 4 * n, hipMemcpyDeviceToHost));
  (void)hipFree(df);
  (void)hipFree(dt);
  (void)hipFree(dd);
  (void)hipFree(dout);
}

// ---- host-side Network API (init() code paths)
void Engine::send(uint32_t msg, uint32_t payload, int32_t sendTime, int32_t from, const int32_t* dests, int32_t n,
                  int32_t delayBetween) {
  const int32_t N = (int32_t)hx.size();
  if (from < 0 || from >= N) throw WgError(WG_EINVAL, "The from node is not in the network.");  // :371,:427
  for (int i = 0; i < n; i++)
    if (dests[i] < 0 || dests[i] >= N) throw WgError(WG_EINVAL, "The to node is not in the network.");  // :374
  if (n <= 0) {  // the list overload draws its seed before it looks at the destinations (:430): an empty list costs a draw
    JavaRandom rd0;
    rd0.s = gh.rng;
    (void)rd0.nextInt();
    gh.rng = rd0.s;
    gh.draws++;
    globalsDirty = true;
    return;
  }
  if (sendTime <= time) throw WgError(WG_ESTATE, "sendTime=" + std::to_string(sendTime) + ", time=" + std::to_string(time));  // :471
  if (!proto) throw WgError(WG_ESTATE, "load a protocol before sending (message sizes are protocol-defined)");
  JavaRandom rd;
  rd.s = gh.rng;
  int32_t seed = rd.nextInt();  // :377 / :430 — drawn before arrivals are computed
  gh.rng = rd.s;
  gh.draws++;
  globalsDirty = true;
  send_seeded(msg, payload, sendTime, from, dests, n, delayBetween, seed);
}

// ... with the seed already drawn (by send() above, or by the caller of a batched step, who holds rd while it applies the
// step's deliveries: wg_step_end)
void Engine::send_seeded(uint32_t msg, uint32_t payload, int32_t sendTime, int32_t from, const int32_t* dests, int32_t n,
                         int32_t delayBetween, int32_t seed) {
  const int32_t N = (int32_t)hx.size();
  if (from < 0 || from >= N) throw WgError(WG_EINVAL, "The from node is not in the network.");
  for (int i = 0; i < n; i++)
    if (dests[i] < 0 || dests[i] >= N) throw WgError(WG_EINVAL, "The to node is not in the network.");
  if (n <= 0) return;
  if (sendTime <= time) throw WgError(WG_ESTATE, "sendTime=" + std::to_string(sendTime) + ", time=" + std::to_string(time));  // :471
  if (!proto) throw WgError(WG_ESTATE, "load a protocol before sending (message sizes are protocol-defined)");
  if (n >= sendExpandMin && delayBetween == 0) return send_expanded(msg, payload, sendTime, from, dests, n, seed);
  // sender statistics are applied on the device at flush time; host keeps them in staged counters
  struct Arr {
    int32_t dest, arrival;
  };
  std::vector<Arr> da;
  int32_t st = sendTime;
  int64_t sentMsgs = 0;
  for (int i = 0; i < n; i++) {
    int32_t to = dests[i];
    sentMsgs++;
    if (part_of(hx[from]) == part_of(hx[to]) && !hdown[from] && !hdown[to]) {
      int32_t nt = host_latency(from, to, seed);
      if (nt < discardTime) da.push_back({to, st + nt});
    }
    st += delayBetween + (delayBetween > 0 ? 1 : 0);  // :459
  }
  if (da.size() > 1) {  // every refusal BEFORE anything is counted or allocated: a caller that catches it finds the engine as it was
    ensure_device();
    if (da.size() * (delayBetween == 0 ? 1 : 2) > dev.chainDests) throw WgError(WG_ENOMEM, "chain_dests too small for this multi-destination send");
    const uint32_t slot = gh.chainHead % dev.chainSlots;
    if (dev.hostMode && hostChains.size() > slot && hostChains[slot].live)  // (the ring has gone round onto an envelope that still has destinations to reach)
      throw WgError(WG_ENOMEM, "chain_slots (" + std::to_string(dev.chainSlots) + ") too small: more multi-destination envelopes are in flight; raise wg_config.chain_slots");
  }
  pendingSent.push_back({from, sentMsgs, msg});
  if (n > 1) std::stable_sort(da.begin(), da.end(), [](const Arr& a, const Arr& b) { return a.arrival < b.arrival; });
  if (da.empty()) {
    if (dev.hostMode) hcReleased.push_back(msg);  // no destination is reachable: the envelope ends here
    return;
  }
  if (da.size() == 1) {
    hc_push(da[0].arrival, make_rec(K_MSG, from, (uint32_t)da[0].dest, msg, payload));
    return;
  }
  StagedChain sc;
  sc.slot = gh.chainHead++ % dev.chainSlots;
  sc.c.from = from;
  sc.c.seed = seed;
  sc.c.sendTime = sendTime;
  sc.c.ndest = (int32_t)da.size();
  size_t words = da.size() * (delayBetween == 0 ? 1 : 2);
  sc.c.destOff = (uint32_t)(gh.destHead % dev.chainDests);
  gh.destHead += words;
  sc.c.msg = msg;
  sc.c.payload = payload;
  sc.c.flags = 1u | (delayBetween == 0 ? 0u : 2u);
  for (auto& a : da) sc.words.push_back(a.dest);
  if (delayBetween != 0)
    for (auto& a : da) sc.words.push_back(a.arrival);
  if (dev.hostMode) {  // the host hands the hops out itself: keep the envelope
    if (hostChains.size() < dev.chainSlots) hostChains.resize(dev.chainSlots);
    hostChains[sc.slot].live = true;
    hostChains[sc.slot].c = sc.c;
    hostChains[sc.slot].words = sc.words;
  }
  hc_push(da[0].arrival, make_rec(K_CHAIN, from, sc.slot, 0, 0));
  stagedChains.push_back(std::move(sc));
}

// A list send with many destinations (a protocol's sendAll): latencies, drops and the stable sort by arrival on the
// device (k_send_expand_*), straight into the envelope's slice of the destination ring. The seed has been drawn.
void Engine::send_expanded(uint32_t msg, uint32_t payload, int32_t sendTime, int32_t from, const int32_t* dests, int32_t n,
                           int32_t seed) {
  ensure_device();
  if (!allocated || (unsigned long long)n > dev.chainDests)
    throw WgError(WG_ENOMEM, "chain_dests too small for this multi-destination send");
  if (dev.hostMode && hostChains.size() > gh.chainHead % dev.chainSlots && hostChains[gh.chainHead % dev.chainSlots].live)  // (before anything is counted)
    throw WgError(WG_ENOMEM, "chain_slots (" + std::to_string(dev.chainSlots) + ") too small: more multi-destination envelopes are in flight; raise wg_config.chain_slots");
  pendingSent.push_back({from, (long long)n, msg});
  Group g = self();
  const uint32_t D = (uint32_t)dev.horizon;
  const uint32_t nTiles = ((uint32_t)n + TILE - 1) / TILE;
  if (expCap < (size_t)n) {  // scratch grows with the largest send seen
    expCap = (size_t)n;
    expIn = dalloc<int32_t>(expCap, false);
    expLat = dalloc<int32_t>(expCap, false);
    expHist = dalloc<uint32_t>((size_t)((expCap + TILE - 1) / TILE) * D, false);
    if (!expResult) expResult = dalloc<int32_t>(2);
  }
  SendExpand x;
  x.in = expIn;
  x.lat = expLat;
  x.hist = expHist;
  x.result = expResult;
  x.n = n;
  x.from = from;
  x.seed = seed;
  x.destOff = gh.destHead % dev.chainDests;
  WG_HIP(hipMemcpyAsync(expIn, dests, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, stream));
  const size_t lds = sizeof(uint32_t) * (size_t)D;
  hipLaunchKernelGGL(k_send_expand_lat, dim3(std::min<uint32_t>(nTiles, GRID_TILES)), dim3(TILE), lds, stream, g.tab, x);
  hipLaunchKernelGGL(k_send_expand_scan, dim3(1), dim3(1024), 0, stream, g.tab, x);
  hipLaunchKernelGGL(k_send_expand_scatter, dim3(std::min<uint32_t>(nTiles, GRID_TILES)), dim3(TILE), lds, stream, g.tab, x, binBits);
  int32_t res[2] = {0, 0};
  WG_HIP(hipMemcpyAsync(res, expResult, sizeof(res), hipMemcpyDeviceToHost, stream));
  WG_HIP(hipStreamSynchronize(stream));
  {  // (ERR_HORIZON from a latency beyond the bucket ring)
    uint32_t err = 0;
    WG_HIP(hipMemcpy(&err, (const char*)dev.g.raw + offsetof(Globals, err), 4, hipMemcpyDeviceToHost));
    if (err) {
      gh.err |= err;
      check_device_errors();
    }
  }
  const int32_t m = res[0];
  if (m <= 0) {
    if (dev.hostMode) hcReleased.push_back(msg);
    return;
  }
  if (sendTime <= time) throw WgError(WG_ESTATE, "sendTime=" + std::to_string(sendTime) + ", time=" + std::to_string(time));  // :471
  const int32_t firstArrival = sendTime + res[1];
  const bool needIds = m == 1 || dev.hostMode;
  std::vector<int32_t> sorted;
  if (needIds) {  // the host hands the hops out itself (host-callback mode), or the envelope degenerates to one destination
    sorted.resize((size_t)m);
    const unsigned long long off = x.destOff;
    const size_t first = (size_t)std::min<unsigned long long>((unsigned long long)m, dev.chainDests - off);
    WG_HIP(hipMemcpy(sorted.data(), dev.dests + off, 4 * first, hipMemcpyDeviceToHost));
    if (first < (size_t)m) WG_HIP(hipMemcpy(sorted.data() + first, dev.dests, 4 * ((size_t)m - first), hipMemcpyDeviceToHost));
  }
  if (m == 1) {
    hc_push(firstArrival, make_rec(K_MSG, from, (uint32_t)sorted[0], msg, payload));
    return;
  }
  StagedChain sc;
  sc.slot = gh.chainHead++ % dev.chainSlots;
  sc.c.from = from;
  sc.c.seed = seed;
  sc.c.sendTime = sendTime;
  sc.c.ndest = m;
  sc.c.destOff = (uint32_t)x.destOff;
  gh.destHead += (unsigned long long)m;
  sc.c.msg = msg;
  sc.c.payload = payload;
  sc.c.flags = 1u;
  globalsDirty = true;
  if (dev.hostMode) {
    if (hostChains.size() < dev.chainSlots) hostChains.resize(dev.chainSlots);
    hostChains[sc.slot].live = true;
    hostChains[sc.slot].c = sc.c;
    hostChains[sc.slot].words = sorted;
  }
  hc_push(firstArrival, make_rec(K_CHAIN, from, sc.slot, 0, 0));
  stagedChains.push_back(std::move(sc));  // (no words: the destinations are in the ring already)
}

void Engine::send_arrive_at(uint32_t msg, uint32_t payload, int32_t arriveAt, int32_t from, int32_t to) {  // :384-390
  const int32_t N = (int32_t)hx.size();
  if (from < 0 || from >= N || to < 0 || to >= N) throw WgError(WG_EINVAL, "node id");
  if (arriveAt <= time)
    throw WgError(WG_EINVAL, "wrong arrival time: arriveAt=" + std::to_string(arriveAt) + ", time=" + std::to_string(time));
  hc_push(arriveAt, make_rec(K_MSG, from, (uint32_t)to, msg, payload));
}

void Engine::register_task(uint32_t task, uint32_t arg, int32_t startAt, int32_t node) {  // :505-508
  if (node < 0 || node >= (int)hx.size()) throw WgError(WG_EINVAL, "node id");
  if (startAt < time) throw WgError(WG_ESTATE, "Arriving in the past: arrival=" + std::to_string(startAt));  // :249-252
  hc_push(startAt, make_rec(K_TASK, node, (uint32_t)node, task, arg));
}
void Engine::register_periodic_task(uint32_t task, int32_t startAt, int32_t period, int32_t node) {  // :510-513
  if (node < 0 || node >= (int)hx.size()) throw WgError(WG_EINVAL, "node id");
  if (period <= 0) throw WgError(WG_EINVAL, "period");
  if (dev.hostMode)
    throw WgError(WG_EUNSUPPORTED, "host-callback mode: a PeriodicTask re-arms itself from its action() (C/messages/PeriodicTask.java:39-47)");
  if (startAt < time) throw WgError(WG_ESTATE, "Arriving in the past: arrival=" + std::to_string(startAt));
  staged.push_back({startAt, make_rec(K_PERIODIC, node, (uint32_t)node, task, (uint32_t)period)});
  if (!periodicUnknown) {  // (Group::periodic_may_fire: a handful of phases with a synchronised start, `period` of them without)
    const int32_t ph = startAt % period;
    bool seen = false;
    for (const PeriodicReg& r : periodicRegs) seen = seen || (r.task == task && r.period == period && r.phase % period == ph && r.phase <= startAt);
    if (!seen) {
      if (periodicRegs.size() >= 64)
        periodicUnknown = true;
      else
        periodicRegs.push_back({task, period, startAt});
    }
  }
}

template <class F>
void Engine::scan(const Group& g, const typename F::Aux* atab) {
  constexpr int total = 1024;
  const int gx = grid_per_engine(SCAN_GRID, g.R, total, g.nodes, 1024);
  hipLaunchKernelGGL(k_scan1<F>, dim3(gx, g.R), dim3(SCAN_BLOCK), 0, g.stream, g.tab, atab);
  hipLaunchKernelGGL(k_scan2<F>, dim3(gx, g.R), dim3(SCAN_BLOCK), 0, g.stream, g.tab, atab);
}
template void Engine::scan<ExpandF>(const Group&, const int*);
// expand: bucket `now` -> events (the pair scan), then the long chain runs it set aside, one wavefront each
void Engine::expand(const Group& g) {
  scan<ExpandF>(g, nullptr);
  if (dev.runMin) hipLaunchKernelGGL(k_expand_runs, dim3(GRID_EXPAND_RUNS, g.R), dim3(256), 0, g.stream, g.tab);
}
template void Engine::scan<RecsF>(const Group&, const int*);
template void Engine::scan<MultiF>(const Group&, const int*);

// multisplit of the ordered outbox (fin/arr, g->nOut) into the buckets. The per-tile histograms are
// built by the producer of the outbox (k_resolve / the protocol's conditional-task kernel); only
// host-staged envelopes need the standalone histogram kernel.
void Engine::append_phase(const Group& g, bool needHist) {
  constexpr int total = 512;
  const int gx = grid_per_engine(GRID_TILES, g.R, total, g.nodes, 2048);
  if (needHist) hipLaunchKernelGGL(k_tile_hist, dim3(gx, g.R), dim3(TILE), g.histLds, g.stream, g.tab, g.binBits);
  hipLaunchKernelGGL(k_col_reserve, dim3(1, g.R), dim3(1024), 0, g.stream, g.tab);
  hipLaunchKernelGGL(k_scatter, dim3(gx, g.R), dim3(TILE), g.histLds, g.stream, g.tab, g.binBits, 0);
}
// append + end of the phase in two launches instead of three (k_col_reserve_end, then the scatter)
void Engine::append_end_phase(const Group& g, bool drained) {
  constexpr int total = 512;
  const int gx = grid_per_engine(GRID_TILES, g.R, total, g.nodes, 2048);
  hipLaunchKernelGGL(k_col_reserve_end, dim3(1, g.R), dim3(1024), 0, g.stream, g.tab, drained ? 1 : 0);
  hipLaunchKernelGGL(k_scatter, dim3(gx, g.R), dim3(TILE), g.histLds, g.stream, g.tab, g.binBits, 1);
}
void Engine::end_phase(const Group& g, bool drained, bool keep) {
  hipLaunchKernelGGL(k_end_phase, dim3(1, g.R), dim3(256), 0, g.stream, g.tab, drained ? 1 : 0, keep ? 1 : 0);
}

// WG_MERGE_APPEND=0: two appends per simulated ms (rounds 1-4). Read when a group is made — once per runMs of an engine or a
// batch, so a test can still toggle it between runs — and carried in the Group: the per-ms enqueue path reads no environment.
static bool merge_append_setting() {
  const char* v = getenv("WG_MERGE_APPEND");
  return !(v && atoi(v) == 0);
}
Group Engine::self() {
  ensure_device();
  upload_down();
  dev.halted = 0;
  bool sync = false;
  if (!dTab || memcmp(&tabShadow, &dev, sizeof(EngineDev)) != 0) {
    if (!dTab) dTab = dalloc<EngineDev>(1);
    WG_HIP(hipMemcpyAsync(dTab, &dev, sizeof(EngineDev), hipMemcpyHostToDevice, stream));
    tabShadow = dev;
    sync = true;
  }
  if (proto) {
    const size_t sz = proto->state_size();
    if (!dStab || stabShadow.size() != sz || memcmp(stabShadow.data(), proto->state_host(), sz) != 0) {
      if (!dStab) {
        WG_HIP(hipMalloc(&dStab, sz));
        allocs.push_back(dStab);
        allocInfo.push_back({sz, AC_STATE});
      }
      WG_HIP(hipMemcpyAsync(dStab, proto->state_host(), sz, hipMemcpyHostToDevice, stream));
      stabShadow.assign((const char*)proto->state_host(), (const char*)proto->state_host() + sz);
      sync = true;
    }
  }
  if (sync) WG_HIP(hipStreamSynchronize(stream));
  Group g;
  g.tab = dTab;
  g.stab = dStab;
  g.R = 1;
  g.stream = stream;
  g.binBits = binBits;
  g.histLds = sizeof(uint32_t) * (size_t)dev.horizon;
  g.mergeAppend = merge_append_setting();
  g.nodes = dev.nodes.n;
  g.periodic = periodicUnknown ? nullptr : &periodicRegs;
  return g;
}

__global__ void k_apply_sent(NodeArrays nd, int n, const int32_t* node, const long long* msgs, const long long* bytes) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  atomicAdd((unsigned long long*)&nd.msgSent[node[i]], (unsigned long long)msgs[i]);
  atomicAdd((unsigned long long*)&nd.bytesSent[node[i]], (unsigned long long)bytes[i]);
}
__global__ void k_set_nout(Globals* g, uint32_t n) { g->nOut = n; }

// Push host-staged envelopes (in push order) through the same append pipeline the device uses.
// Envelopes arriving beyond the bucket ring's horizon stay on the host until they come into range;
// they are injected at the start of ms (arrival - horizon + 1), before any device push of that ms can
// target the same bucket, so the push order inside the bucket is unchanged.
void Engine::flush_staged(int32_t t, bool inRun) {
  if (staged.empty() && pendingSent.empty()) return;
  ensure_device();
  if (!inRun) sync_globals_to_device();
  for (auto& sc : stagedChains) {
    if (sc.slot >= dev.chainSlots) throw WgError(WG_ENOMEM, "chain_slots");
    {  // the envelope's words into the destination ring: one copy, two where the slice wraps
      const size_t n = sc.words.size(), at = sc.c.destOff % dev.chainDests;
      const size_t first = std::min(n, (size_t)(dev.chainDests - at));
      if (first) WG_HIP(hipMemcpy(dev.dests + at, sc.words.data(), 4 * first, hipMemcpyHostToDevice));
      if (n > first) WG_HIP(hipMemcpy((int32_t*)dev.dests, sc.words.data() + first, 4 * (n - first), hipMemcpyHostToDevice));
    }
    WG_HIP(hipMemcpy(dev.chains + sc.slot, &sc.c, sizeof(Chain), hipMemcpyHostToDevice));
  }
  stagedChains.clear();
  if (!pendingSent.empty()) {
    // Node.msgSent / bytesSent for host-side sends (C/Network.java:476-477)
    std::vector<int32_t> nodes;
    std::vector<long long> msgs, bytes;
    for (auto& p : pendingSent) {
      if (!shard_owns(dev, p.node)) continue;  // sharded: Node counters live with the owner of the node
      nodes.push_back(p.node);
      msgs.push_back(p.msgs);
      bytes.push_back(p.msgs * (long long)proto->host_msg_size(p.msg));
    }
    const int n = (int)nodes.size();
    if (n > 0) {  // one grow-only staging buffer {msgs[n], bytes[n], nodes[n]} and one copy (a hipMalloc / hipFree per flush before)
      const size_t need = 20 * (size_t)n;
      if (need > sentBufBytes) {
        if (sentBuf) (void)hipFree(sentBuf);
        sentBuf = nullptr;
        sentBufBytes = 0;
        WG_HIP(hipMalloc((void**)&sentBuf, 2 * need));
        sentBufBytes = 2 * need;
      }
      std::vector<char> host(need);
      memcpy(host.data(), msgs.data(), 8 * (size_t)n);
      memcpy(host.data() + 8 * (size_t)n, bytes.data(), 8 * (size_t)n);
      memcpy(host.data() + 16 * (size_t)n, nodes.data(), 4 * (size_t)n);
      WG_HIP(hipMemcpyAsync(sentBuf, host.data(), need, hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k_apply_sent, dim3((n + 255) / 256), dim3(256), 0, stream, dev.nodes, n, (int32_t*)(sentBuf + 16 * (size_t)n),
                         (long long*)sentBuf, (long long*)(sentBuf + 8 * (size_t)n));
      WG_HIP(hipStreamSynchronize(stream));  // (`host` is pageable: the copy has left it by now)
    }
    pendingSent.clear();
  }
  std::vector<Staged> near, far;
  for (auto& st : staged) (st.arrival - t < dev.horizon ? near : far).push_back(st);
  staged.swap(far);
  stagedMin = INT32_MAX;
  for (auto& st : staged) stagedMin = std::min(stagedMin, st.arrival);
  size_t done = 0;
  while (done < near.size()) {
    size_t n = std::min<size_t>(near.size() - done, dev.maxOut);
    std::vector<Rec> recs(n);
    std::vector<int32_t> arr(n);
    for (size_t i = 0; i < n; i++) {
      recs[i] = near[done + i].rec;
      arr[i] = near[done + i].arrival;
    }
    WG_HIP(hipMemcpyAsync(dev.fin, recs.data(), sizeof(Rec) * n, hipMemcpyHostToDevice, stream));
    WG_HIP(hipMemcpyAsync(dev.arr, arr.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_set_nout, dim3(1), dim3(1), 0, stream, dev.g, (uint32_t)n);
    append_phase(self(), true);
    hipLaunchKernelGGL(k_set_nout, dim3(1), dim3(1), 0, stream, dev.g, 0u);
    WG_HIP(hipStreamSynchronize(stream));
    done += n;
  }
  if (!inRun) {
    sync_globals_to_host();
    check_device_errors();
  }
}

void Engine::check_device_errors() {
  uint32_t e = gh.err;
  if (!e) return;
  std::string m;
  int32_t code = WG_ENOMEM;
  if (e & ERR_BUCKET_POOL) m += "bucket page pool exhausted (raise wg_config.bucket_pool_records); ";
  if (e & ERR_BUCKET_PAGES) m += "one ms bucket exceeded the per-bucket page table; ";
  if (e & ERR_OUTBOX) m += "more records emitted in one ms than wg_config.outbox_records; ";
  if (e & ERR_EVENTS) m += "more events in one ms than wg_config.outbox_records; ";
  if (e & ERR_HORIZON) m += "arrival beyond wg_config.horizon_ms; ";
  if (e & ERR_CHAIN_SLOTS) m += "wg_config.chain_slots exhausted; ";
  if (e & ERR_CHAIN_DESTS) m += "wg_config.chain_dests ring overrun; ";
  if (e & ERR_PAYLOAD) m += "wg_config.payload_words ring overrun; ";
  if (e & ERR_QUEUE_CAP) m += "toVerify list (Handel toVerifyAgg / GSFSignature toVerify) exceeded wg_config.queue_cap (Handel levels of 1024 ids and more: queue_cap_wide); ";
  if (e & ERR_PENDING) m += "pending-verification table full; ";
  if (e & ERR_RANK_BUMPS) m += "a Handel node bumped the reception rank of more distinct senders than its table holds: raise wg_config.rank_bump_cap "
                               "(default 512 beyond 4096 nodes; nodeCount = a direct-indexed table that cannot overflow) or keep the N x N matrix "
                               "with WG_HANDEL_RANKS=matrix; ";
  if (e & ERR_MULTI_TOO_BIG) {
    m += "device-side multi-destination send with more than 64 destinations; ";
    code = WG_EUNSUPPORTED;
  }
  if (e & ERR_SHARD_MULTI) {
    m += "sharded engine: an action() emitted a multi-destination envelope (not sharded yet); ";
    code = WG_EUNSUPPORTED;
  }
  if (e & ERR_SAME_MS) {
    m += "an action() registered an envelope for the millisecond being drained; ";
    code = WG_EUNSUPPORTED;
  }
  if (e & ERR_SHARD_EVENT) {
    m += "sharded engine: one event emitted 1024 or more records or made 2048 or more rd draws (the packed exchange word's limits, include/wittgpu.h); ";
    code = WG_EUNSUPPORTED;
  }
  if (e & ERR_ARRIVAL_PAST) {
    m += "Arriving in the past; ";
    code = WG_ESTATE;
  }
  if (e & ERR_SAME_MS_BLOCKS)
    throw WgError(WG_EUNSUPPORTED, "Casper IMD resident: two blocks were created in one simulated ms (the reference allows it; "
                                   "block-id order across concurrent wavefronts is not resident) — run it on the host-callback mode");
  if (e & ERR_PROTOCOL) {
    m += "protocol invariant violated (reference IllegalStateException site); ";
    code = WG_ESTATE;
  }
  throw WgError(code, m);
}

void Engine::load_protocol(int32_t id, const void* params, const void* initState) {
  if (proto) throw WgError(WG_ESTATE, "a protocol is already resident");
  if (id == WG_PROTO_HOST) {
    if (shardCount > 0) throw WgError(WG_EUNSUPPORTED, "host-callback mode on a sharded engine");
    ensure_device();
    proto = make_host_proto(*this);
    dev.hostMode = 1;
  } else if (id == WG_PROTO_PINGPONG) {
    ensure_device();
    proto = make_pingpong_host(*this);
  } else if (id == WG_PROTO_HANDEL) {
    if (!params || !initState) throw WgError(WG_EINVAL, "Handel needs wg_handel_params and wg_handel_init_state");
    proto = make_handel_host(*this, *(const wg_handel_params*)params, *(const wg_handel_init_state*)initState);
  } else if (id == WG_PROTO_GSF) {
    if (!params || !initState) throw WgError(WG_EINVAL, "GSFSignature needs wg_gsf_params and wg_gsf_init_state");
    proto = make_gsf_host(*this, *(const wg_gsf_params*)params, *(const wg_gsf_init_state*)initState);
  } else if (id == WG_PROTO_SANFERMIN) {
    if (!params) throw WgError(WG_EINVAL, "San Fermin needs wg_sanfermin_params");
    proto = make_sanfermin_host(*this, *(const wg_sanfermin_params*)params);
  } else if (id == WG_PROTO_CASPER) {
    if (!params) throw WgError(WG_EINVAL, "Casper IMD needs wg_casper_params");
    proto = make_casper_host(*this, *(const wg_casper_params*)params);
  } else if (id == WG_PROTO_P2PFLOOD) {
    if (!params || !initState) throw WgError(WG_EINVAL, "P2PFlood needs wg_p2pflood_params and wg_p2pflood_init_state");
    proto = make_p2pflood_host(*this, *(const wg_p2pflood_params*)params, *(const wg_p2pflood_init_state*)initState);
  } else {
    throw WgError(WG_EINVAL, "unknown protocol id");
  }
}

// Network.runMs (C/Network.java:318-338) + receiveUntil/nextMessage as a time-stepped loop.
void Engine::begin_run(int32_t ms, int32_t* endAtOut) {
  if (ms <= 0) throw WgError(WG_EINVAL, "Should be greater than 0. ms=" + std::to_string(ms));
  if (!proto) throw WgError(WG_ESTATE, "no resident protocol loaded");
  int32_t endAt = (int32_t)((uint32_t)time + (uint32_t)ms);
  if (endAt <= 0) throw WgError(WG_ESTATE, "Maximum time reached!");
  ensure_device();
  flush_staged(time, false);
  // (time == 0: Node.start() on non-down nodes only re-asserts down == false, :323-329)
  gh.epoch++;  // a new receiveUntil() starts with a fresh nextMessage() call
  gh.anyEvent = 0;
  gh.now = time;
  gh.until = endAt;
  globalsDirty = true;
  sync_globals_to_device();
  *endAtOut = endAt;
}

void Engine::run_ms(int32_t ms, uint8_t* didSomething, wg_run_stats* stats) {
  if (ms <= 0) throw WgError(WG_EINVAL, "Should be greater than 0. ms=" + std::to_string(ms));
  if (!proto) throw WgError(WG_ESTATE, "no resident protocol loaded");
  if (dev.hostMode) throw WgError(WG_ESTATE, "host-callback mode: drive the run with wg_next_delivery");
  if (shardCount > 0) return run_ms_sharded(ms, didSomething, stats);
  Engine* me = this;
  Group g = self();
  run_group(&me, 1, nullptr, g, ms, didSomething, stats);
}

// Envelopes the device parked beyond the bucket ring (FarRec) become host-held envelopes, in push order (ms, then
// position in that ms's ordered outbox); flush_staged injects them when their bucket comes into range.
void Engine::collect_far() {
  if (!dev.farBuf) return;
  WG_HIP(hipStreamSynchronize(stream));
  uint32_t n = 0;
  WG_HIP(hipMemcpy(&n, (const char*)dev.g.raw + offsetof(Globals, nFar), 4, hipMemcpyDeviceToHost));
  if (!n) return;
  if (n > dev.farCap) throw WgError(WG_ENOMEM, "more envelopes registered beyond horizon_ms than the far buffer holds");
  std::vector<FarRec> recs(n);
  WG_HIP(hipMemcpy(recs.data(), dev.farBuf, sizeof(FarRec) * n, hipMemcpyDeviceToHost));
  const uint32_t zero = 0;
  WG_HIP(hipMemcpy((char*)dev.g.raw + offsetof(Globals, nFar), &zero, 4, hipMemcpyHostToDevice));
  std::sort(recs.begin(), recs.end(), [](const FarRec& a, const FarRec& b) { return a.ms != b.ms ? a.ms < b.ms : a.p < b.p; });
  for (const FarRec& r : recs) {
    staged.push_back({r.arrival, r.rec});
    stagedMin = std::min(stagedMin, r.arrival);
  }
}

// One simulated ms: drain(now) [k_end_phase: now++] + the conditional-task phase of the edge to the new `now`.
static void enqueue_one_ms(Engine& lead, const Group& g0, int32_t tNow) {
  Group g = g0;
  g.now = tNow;
  ProtoHost* proto = lead.proto;
  const bool cond = proto->has_cond();
  typedef Engine::ProfScope ProfScope;
    {
      ProfScope ps(lead, Engine::PC_EXPAND);
      lead.expand(g);
    }
    {
      ProfScope ps(lead, Engine::PC_DELIVER);
      proto->launch_deliver(g);
    }
    {
      ProfScope ps(lead, Engine::PC_ORDER);
      Engine::scan<RecsF>(g, nullptr);
    }
    {
      ProfScope ps(lead, Engine::PC_RESOLVE);
      // (a block that finds no record still costs its launch — ~ 7 ns each, 25 us of GSFSignature's every ms at 4096 blocks,
      // profiles/r14j —: the wide grid only in a ms in which a periodic task may fire)
      constexpr int total = 4096, totalQuiet = 512;
      const int gx = grid_per_engine(GRID_RESOLVE, g.R, g.any_periodic_may_fire() ? total : totalQuiet, g.nodes, g.any_periodic_may_fire() ? 256 : 2048);
      hipLaunchKernelGGL(k_resolve<false>, dim3(gx, g.R), dim3(256), 0, g.stream, g.tab);
    }
    if (lead.dev.maxSendAll) {  // Network.sendAll calls of this ms's action()s: destinations, envelopes, first arrivals
      ProfScope ps(lead, Engine::PC_RESOLVE);
      hipLaunchKernelGGL(k_sendall_lat, dim3(GRID_TILES, g.R), dim3(TILE), g.histLds, g.stream, g.tab);
      hipLaunchKernelGGL(k_sendall_scan, dim3(GRID_DELIVER_SMALL / 8 > 0 ? GRID_DELIVER_SMALL / 8 : 1, g.R), dim3(1024), 0, g.stream, g.tab);
      hipLaunchKernelGGL(k_sendall_scatter, dim3(GRID_TILES, g.R), dim3(TILE), g.histLds, g.stream, g.tab, g.binBits);
    }
    {
      ProfScope ps(lead, Engine::PC_APPEND);
      // with a conditional-task phase behind it the drain only ENDS here (clock, rd, the drained bucket's pages): its ordered
      // outbox stays in fin / arr, the edge's records follow it there, and the phase's append files both — one
      // k_col_reserve_end + k_scatter per simulated ms instead of two (WG_MERGE_APPEND=0: two)
      if (cond && g.mergeAppend)
        Engine::end_phase(g, true, true);
      else
        Engine::append_end_phase(g, true);
    }
    if (cond) {
      // (running the phase's first kernels on a second stream beside the drain's tail was measured: no gain — the short
      // kernels' latency doubles under the other stream's memory load, profiles/r07e_*)
      proto->launch_cond(lead, g);
      ProfScope ps(lead, Engine::PC_APPEND);
      Engine::append_end_phase(g, false);
    }
  }

void Engine::enqueue_ms_sequence(Engine& lead, const Group& g, int32_t ms, int32_t tStart) {
  for (int32_t k = 0; k <= ms; k++) enqueue_one_ms(lead, g, tStart == INT32_MIN ? INT32_MIN : tStart + k);
}

// One simulated ms = drain(now) [k_end_phase: now++] + the conditional-task phase of the edge to the
// new `now` (:543-566). runMs(ms) is ms + 1 of those: the first drain re-visits the current ms
// (envelopes the host registered for it), the last conditional phase is the edge to until + 1, which
// still runs tasks whose minStartTime <= until. Kernels read `now` from device globals,
// so the launch sequence does not depend on the time and serves every member of the group at once.
void Engine::run_group(Engine** es, int R, const uint8_t* active, const Group& g, int32_t ms, uint8_t* did,
                       wg_run_stats* stats) {
  Engine& lead = *es[0];
  std::vector<Globals> before(R);
  std::vector<int32_t> endAt(R, 0);
  std::vector<uint8_t> on(R, 1);
  for (int r = 0; r < R; r++) {
    on[r] = !active || active[r];
    if (on[r]) {
      before[r] = es[r]->gh;  // statistics deltas are taken against the state before begin_run
      es[r]->begin_run(ms, &endAt[r]);
    }
  }
  auto t0 = std::chrono::steady_clock::now();
  ProtoHost* proto = lead.proto;
  const bool cond = proto->has_cond();
  // Host-held envelopes beyond the bucket ring come into range as `now` advances (arrival - t <
  // horizon). They must precede every device push into their bucket: drain(t - 1) reaches at most
  // t + horizon - 2 and the conditional-task phase of the edge to t is held to the same bound
  // (ERR_HORIZON at horizon - 1), so injecting before drain(t) keeps the bucket's push order.
  auto host_envelopes = [&](int32_t k) {
    for (int r = 0; r < R; r++) {
      Engine& e = *es[r];
      const int32_t t = e.time + k;
      // (a parked envelope is >= 2 * horizon ms ahead of its push: collecting every horizon ms is early enough)
      if (on[r] && e.dev.farBuf && (t % e.dev.horizon) == 0) e.collect_far();
      if (on[r] && e.stagedMin - t < e.dev.horizon) {
        WG_HIP(hipStreamSynchronize(g.stream));
        e.flush_staged(t, true);
      }
    }
  };
  // Idle stretches are skipped instead of enqueued (k_next_busy / k_skip_idle) where an empty ms does nothing: no
  // conditional tasks. Every SKIP_EVERY ms the host asks how far the first non-empty bucket is; the skip stops at
  // the next multiple of horizon (far collection), at the run's last ms, and cannot pass a host-held envelope (those
  // are >= horizon ahead after host_envelopes(k)). WG_SKIP_IDLE=0 enqueues every ms.
  const bool skipIdle = !(getenv("WG_SKIP_IDLE") && atoi(getenv("WG_SKIP_IDLE")) == 0);  // (read per run: tests toggle it)
  const bool canSkip = skipIdle && !cond && !lead.dev.hostMode;
  const int32_t SKIP_EVERY = 16;
  int32_t* dNb = nullptr;
  std::vector<int32_t> hNb((size_t)R);
  if (canSkip && ms > SKIP_EVERY) {  // (kept by the lead engine: no allocation on the per-chunk path)
    if (lead.skipCap < R) {
      lead.skipBuf = lead.dalloc<int32_t>((size_t)R);
      lead.skipCap = R;
    }
    dNb = lead.skipBuf;
  }
  int32_t sinceCheck = 0;
  int32_t tCommon = INT32_MIN;  // Network.time of the members, if they share it (Group::now)
  for (int r = 0; r < R; r++)
    if (on[r]) {
      if (tCommon == INT32_MIN)
        tCommon = es[r]->time;
      else if (tCommon != es[r]->time) {
        tCommon = INT32_MIN;
        break;
      }
    }
  {
    for (int32_t k = 0; k <= ms; k++) {
      if (k > 0) host_envelopes(k);
      if (dNb && k > 0 && k < ms && ++sinceCheck >= SKIP_EVERY) {
        sinceCheck = 0;
        hipLaunchKernelGGL(k_next_busy, dim3(1, R), dim3(256), 0, g.stream, g.tab, dNb);
        WG_HIP(hipMemcpyAsync(hNb.data(), dNb, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost, g.stream));
        WG_HIP(hipStreamSynchronize(g.stream));
        int32_t n = ms - k;
        for (int r = 0; r < R; r++)
          if (on[r]) {
            const int32_t D = es[r]->dev.horizon, t = es[r]->time + k;
            n = std::min(n, std::min(hNb[r], D - (int32_t)((uint32_t)t & (uint32_t)(D - 1))));
          }
        if (n > 0) {
          hipLaunchKernelGGL(k_skip_idle, dim3(1, R), dim3(256), 0, g.stream, g.tab, n);
          k += n;
          host_envelopes(k);
        }
      }
      enqueue_one_ms(lead, g, tCommon == INT32_MIN ? INT32_MIN : tCommon + k);
    }
  }
  WG_HIP(hipStreamSynchronize(g.stream));
  auto t1 = std::chrono::steady_clock::now();
  if (lead.profiling) lead.prof_collect();
  const int64_t wall = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
  std::string firstErr;
  int32_t firstCode = WG_OK;
  for (int r = 0; r < R; r++) {
    if (!on[r]) {
      if (did) did[r] = 0;
      if (stats) memset(&stats[r], 0, sizeof(wg_run_stats));
      continue;
    }
    Engine& e = *es[r];
    e.collect_far();
    e.sync_globals_to_host();
    e.time = endAt[r];
    if (did) did[r] = e.gh.anyEvent ? 1 : 0;
    if (stats) {
      wg_run_stats& st = stats[r];
      st.delivered = (int64_t)(e.gh.delivered - before[r].delivered);
      st.tasks = (int64_t)(e.gh.tasks - before[r].tasks);
      st.events = (int64_t)(e.gh.events - before[r].events);
      st.draws = (int64_t)(e.gh.draws - before[r].draws);
      st.simulated_ms = ms;
      st.wall_ns = wall;
      int64_t pb = 0;
      for (int l = 0; l < 32; l++)
        pb += (int64_t)(e.gh.deliveredByLevel[l] - before[r].deliveredByLevel[l]) * e.proto->payload_bytes_of_level(l);
      st.payload_bytes = pb;
    }
    try {
      e.check_device_errors();
    } catch (const WgError& x) {
      e.lastError = x.what();
      if (firstCode == WG_OK) {
        firstCode = x.code;
        firstErr = x.what();
      }
    }
  }
  if (firstCode != WG_OK) throw WgError(firstCode, firstErr);
}

// ------------------------------------------------------------------------------------------------
// Node-range sharding of ONE simulation (wg_shard_configure, include/wittgpu.h). Every shard runs the whole
// per-ms pipeline on a replicated scheduler state; only the delivery (Message.action(), node and protocol state)
// and the send resolution (rd draw -> latency -> arrival) are split, by owner of the acting node. The two places
// where a shard needs what the others computed are summed across shards (non-owners contribute zeros):
//   evRes[0..nEvents)   (records emitted, draws) per event  -> `order` gives every shard the global push order
//   xbuf[0..nOut)       resolved record + arrival           -> `append` files the same records everywhere
// The counts are replicated too, so every shard issues the same collectives without negotiating sizes.
void Engine::configure_shard(int32_t shard, int32_t nshards, wg_allreduce_fn fn, void* ctx) {
  if (nshards <= 0 || shard < 0 || shard >= nshards || !fn) throw WgError(WG_EINVAL, "shard / nshards / allreduce");
  if (allocated) throw WgError(WG_ESTATE, "wg_shard_configure must precede the first call that allocates the engine");
  shardIndex = shard;
  shardCount = nshards;
  xfn = fn;
  xctx = ctx;
}

void Engine::configure_shard_rccl(int32_t shard, int32_t nshards, const uint8_t* id128) {
  if (nshards <= 0 || shard < 0 || shard >= nshards || !id128) throw WgError(WG_EINVAL, "shard / nshards / unique id");
  if (allocated) throw WgError(WG_ESTATE, "wg_shard_configure_rccl must precede the first call that allocates the engine");
  Rccl& r = rccl();
  if (!r.err.empty()) throw WgError(WG_EHIP, r.err);
  NcclUniqueId id;
  memcpy(id.internal, id128, 128);
  WG_HIP(hipSetDevice(cfg.device));
  rccl_check(r.CommInitRank(&rcclComm, nshards, id, shard), "ncclCommInitRank");
  shardIndex = shard;
  shardCount = nshards;
  xfn = nullptr;
  xctx = nullptr;
}

// counts the host needs to size the next collective: published into pinned host memory in stream order, awaited by
// polling (a few microseconds instead of a stream synchronisation and a blocking copy per count)
__global__ void k_publish(const uint32_t* a, const uint32_t* b, Engine::Mailbox* mb, uint32_t seq) {
  mb->v[0] = a ? *a : 0u;
  mb->v[1] = b ? *b : 0u;
  __threadfence_system();
  mb->seq = seq;
}
uint32_t Engine::publish_counts(const uint32_t* a, const uint32_t* b) {
  if (!mailbox) {
    WG_HIP(hipHostMalloc((void**)&mailbox, sizeof(Mailbox) * MAILBOXES, 0));
    memset((void*)mailbox, 0, sizeof(Mailbox) * MAILBOXES);
  }
  const uint32_t seq = ++mailSeq;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, stream, a, b, mailbox + (seq % MAILBOXES), seq);
  return seq;
}
void Engine::wait_counts(uint32_t seq, uint32_t* va, uint32_t* vb) {
  volatile Mailbox* mb = mailbox + (seq % MAILBOXES);
  // acquire on the sequence word: the counts the device wrote before it are read after it, whatever the compiler would
  // like to hoist; the spin yields its pipeline slots
  uint64_t spins = 0;
  while (__atomic_load_n(&mb->seq, __ATOMIC_ACQUIRE) != seq) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    if ((++spins & 0xFFFF) == 0) {  // a failed launch would never publish: ask the runtime now and then
      const hipError_t q = hipStreamQuery(stream);
      if (q != hipSuccess && q != hipErrorNotReady) WG_HIP(q);
      if (q == hipSuccess && __atomic_load_n(&mb->seq, __ATOMIC_ACQUIRE) != seq)
        throw WgError(WG_EHIP, "the count mailbox was not written");
    }
  }
  if (va) *va = __atomic_load_n(&mb->v[0], __ATOMIC_RELAXED);
  if (vb) *vb = __atomic_load_n(&mb->v[1], __ATOMIC_RELAXED);
}

void Engine::set_alltoallv(wg_alltoallv_fn fn, void* ctx) {
  if (shardCount <= 0) throw WgError(WG_ESTATE, "wg_shard_set_alltoallv: configure the shard first (wg_shard_configure)");
  if (proto) throw WgError(WG_ESTATE, "wg_shard_set_alltoallv must precede wg_protocol_load");
  xa2a = fn;
  xa2aCtx = ctx;
}
bool Engine::has_alltoall() const {
  if (rcclComm) {
    const Rccl& r = rccl();
    return r.Send && r.Recv && r.GroupStart && r.GroupEnd;
  }
  return xa2a != nullptr;
}
void Engine::shard_alltoallv(const void* sendbuf, const int64_t* sc, const int64_t* so, void* recvbuf, const int64_t* rc, const int64_t* ro, int kind) {
  int64_t got = 0;
  for (int r = 0; r < shardCount; r++) got += r == shardIndex ? 0 : rc[r];
  if (rcclComm) {  // grouped point-to-point calls on the engine's stream: RCCL's all-to-all
    Rccl& R = rccl();
    rccl_check(R.GroupStart(), "ncclGroupStart");
    for (int p = 0; p < shardCount; p++) {
      if (p == shardIndex) continue;
      if (sc[p] > 0) rccl_check(R.Send((const int32_t*)sendbuf + so[p], (size_t)sc[p], /*ncclInt32*/ 2, p, rcclComm, stream), "ncclSend");
      if (rc[p] > 0) rccl_check(R.Recv((int32_t*)recvbuf + ro[p], (size_t)rc[p], /*ncclInt32*/ 2, p, rcclComm, stream), "ncclRecv");
    }
    rccl_check(R.GroupEnd(), "ncclGroupEnd");
  } else {
    if (!xa2a) throw WgError(WG_ESTATE, "no all-to-all transport (wg_shard_set_alltoallv)");
    WG_HIP(hipStreamSynchronize(stream));
    const int32_t rcod = xa2a(xa2aCtx, sendbuf, sc, so, recvbuf, rc, ro);
    if (rcod != 0) throw WgError(WG_EHIP, "the shard all-to-all callback failed with " + std::to_string(rcod));
  }
  shardCollectives++;
  shardWords += got;  // (words RECEIVED by this shard: what the exchange costs it — an all-reduce counts its whole buffer)
  shardCallsBy[kind]++;
  shardWordsBy[kind] += got;
}

void Engine::shard_allreduce(void* buf, int64_t count, int kind) {
  if (count <= 0) return;
  if (rcclComm) {  // in stream order with the producers and consumers of `buf`: nothing to wait for on the host
    rccl_check(rccl().AllReduce(buf, buf, (size_t)count, /*ncclInt32*/ 2, /*ncclSum*/ 0, rcclComm, stream), "ncclAllReduce");
  } else {
    WG_HIP(hipStreamSynchronize(stream));
    const int32_t rc = xfn(xctx, buf, count);
    if (rc != 0) throw WgError(WG_EHIP, "the shard all-reduce callback failed with " + std::to_string(rc));
  }
  shardCollectives++;
  shardWords += count;
  shardCallsBy[kind]++;
  shardWordsBy[kind] += count;
}

// the exchange image of a phase's ordered outbox (xbuf, written by the owners) -> fin / arr / tile histograms on
// every shard, then the multi-destination envelopes among them
void Engine::exchange_outbox(uint32_t nOut) {
  Group g = self();
  shard_allreduce(dev.xbuf - XB_HEAD, XB_HEAD + 5 * (int64_t)nOut, XK_OUTBOX);
  // the header word arrived with the records: how many of them are multi-destination envelopes still to be created
  // (the collective has synchronised; no further stream synchronisation is needed to read it)
  uint32_t nMulti = 0;
  const uint32_t seq = publish_counts((const uint32_t*)(dev.xbuf - XB_HEAD), nullptr);
  WG_HIP(hipMemsetAsync(dev.xbuf - XB_HEAD, 0, sizeof(int32_t) * XB_HEAD, stream));  // for the next phase's producers
  hipLaunchKernelGGL(k_shard_unpack, dim3(GRID_SHARD_SMALL, 1), dim3(256), 0, stream, g.tab);
  wait_counts(seq, &nMulti, nullptr);  // (read while k_shard_unpack runs)
  if (!nMulti) return;
  scan<MultiF>(g, nullptr);
  nMulti = std::min(nMulti, dev.maxMulti);
  WG_HIP(hipMemsetAsync(dev.xmulti, 0, sizeof(int32_t) * (size_t)nMulti * XM_WORDS, stream));
  hipLaunchKernelGGL(k_shard_multi_fill, dim3(GRID_RESOLVE, 1), dim3(256), 0, stream, g.tab);
  shard_allreduce(dev.xmulti, (int64_t)nMulti * XM_WORDS, XK_ENVELOPES);
  hipLaunchKernelGGL(k_shard_multi_create, dim3(GRID_RESOLVE, 1), dim3(256), 0, stream, g.tab);
}

void Engine::run_ms_sharded(int32_t ms, uint8_t* didSomething, wg_run_stats* stats) {
  if (!proto->supports_shards())
    throw WgError(WG_EUNSUPPORTED, "this resident protocol does not run on a sharded engine yet (PingPong, Handel and GSFSignature do)");
  const Globals before = gh;
  int32_t endAt = 0;
  begin_run(ms, &endAt);
  Group g = self();
  auto t0 = std::chrono::steady_clock::now();
  auto gfield = [&](uint32_t Globals::*field) {
    return (const uint32_t*)((const char*)dev.g.raw + ((const char*)&(gh.*field) - (const char*)&gh));
  };
  // host-held envelopes (see run_group) — the far ones were parked by k_shard_unpack on every shard alike, so every shard
  // stages the same
  auto host_envelopes = [&](int32_t k) {
    const int32_t t = time + k;
    if (dev.farBuf && (t % dev.horizon) == 0) collect_far();
    if (k > 0 && stagedMin - t < dev.horizon) {
      WG_HIP(hipStreamSynchronize(stream));
      flush_staged(t, true);
    }
  };
  // Idle stretches are skipped as run_group skips them (k_next_busy / k_skip_idle: the bucket counts are replicated, every
  // shard skips alike) — asked after a ms that held no event, so a protocol that is busy every ms never pays the question
  const bool canSkip = !(getenv("WG_SKIP_IDLE") && atoi(getenv("WG_SKIP_IDLE")) == 0) && !proto->has_cond();
  if (canSkip && skipCap < 1) {
    skipBuf = dalloc<int32_t>(1);
    skipCap = 1;
  }
  bool idlePrev = false;
  for (int32_t k = 0; k <= ms; k++) {
    host_envelopes(k);
    if (canSkip && idlePrev && k > 0 && k < ms) {
      int32_t nb = 0;
      hipLaunchKernelGGL(k_next_busy, dim3(1, 1), dim3(256), 0, stream, g.tab, skipBuf);
      WG_HIP(hipMemcpyAsync(&nb, skipBuf, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
      WG_HIP(hipStreamSynchronize(stream));
      const int32_t D = dev.horizon, t = time + k;
      const int32_t n = std::min(ms - k, std::min(nb, D - (int32_t)((uint32_t)t & (uint32_t)(D - 1))));
      if (n > 0) {
        hipLaunchKernelGGL(k_skip_idle, dim3(1, 1), dim3(256), 0, stream, g.tab, n);
        k += n;
        host_envelopes(k);
      }
    }
    expand(g);
    const uint32_t seqE = publish_counts(gfield(&Globals::nEvents), nullptr);  // (known after expand: read behind the delivery pass)
    {
      ProfScope ps(*this, PC_DELIVER);
      if (!proto->shard_deliver(*this, g)) proto->launch_deliver(g);  // (a pass with collectives of its own: Casper's randomOnTies)
    }
    uint32_t nEvents = 0;
    wait_counts(seqE, &nEvents, nullptr);
    idlePrev = nEvents == 0;
    // exchange 1: the events' results, one packed word each (records, draws, the two flags, the level, the snapshot code)
    if (nEvents) {
      hipLaunchKernelGGL((k_shard_evres<true>), dim3(GRID_SHARD_SMALL, 1), dim3(256), 0, stream, g.tab);
      shard_allreduce(dev.xev, (int64_t)nEvents, XK_EVENTS);
      hipLaunchKernelGGL((k_shard_evres<false>), dim3(GRID_SHARD_SMALL, 1), dim3(256), 0, stream, g.tab);
    }
    scan<RecsF>(g, nullptr);
    // the payload snapshots of this ms (Handel's disseminations, GSF's doCycles): which events wrote one, and how wide, came
    // with exchange 1 and the order scan counted them (nSnapEv); only a ms that has any numbers its rows (a second scan)
    const bool snapScan = proto->shard_snap_is_scan(), snapDirected = proto->shard_snap_directed();
    const uint32_t* dSecond = !nEvents ? nullptr : (snapScan || snapDirected) ? gfield(&Globals::nSnapEv) : proto->shard_snap_enqueue(g);
    uint32_t nOut = 0, nSnap = 0;
    const uint32_t seqO = publish_counts(gfield(&Globals::nOut), dSecond);
    // (k_resolve takes the record count from the device: enqueued BEFORE the host learns it, so that the poll is not a gap)
    if (nEvents) hipLaunchKernelGGL(k_resolve<true>, dim3(GRID_RESOLVE, 1), dim3(256), 0, stream, g.tab);
    wait_counts(seqO, &nOut, &nSnap);
    if (dSecond && nSnap) {
      if (snapDirected) {  // the sub-rows other shards' nodes will read, to those shards (all-to-all)
        proto->shard_snap_exchange(*this, g, nSnap);
      } else if (snapScan) {  // rows packed back to back: their total width sizes the collective
        uint32_t words = 0;
        await_counts(proto->shard_snap_enqueue(g), nullptr, &words, nullptr);
        if (words) proto->shard_snap_exchange(*this, g, words);
      } else {
        proto->shard_snap_exchange(*this, g, nSnap);
      }
    }
    if (nOut) {
      exchange_outbox(nOut);
      if (dev.maxSendAll) {  // the Network.sendAll calls among them: destinations, envelopes, first arrivals — on every shard
        hipLaunchKernelGGL(k_sendall_lat, dim3(GRID_TILES, 1), dim3(TILE), g.histLds, stream, g.tab);
        hipLaunchKernelGGL(k_sendall_scan, dim3(GRID_DELIVER_SMALL / 8 > 0 ? GRID_DELIVER_SMALL / 8 : 1, 1), dim3(1024), 0, stream, g.tab);
        hipLaunchKernelGGL(k_sendall_scatter, dim3(GRID_TILES, 1), dim3(TILE), g.histLds, stream, g.tab, g.binBits);
      }
    }
    append_end_phase(g, true);
    if (proto->has_cond()) {  // the conditional tasks of the edge to the new `now` (C/Network.java:543-566)
      const uint32_t nCond = proto->shard_cond(*this, g);
      if (nCond) exchange_outbox(nCond);
      append_end_phase(g, false);
    }
  }
  WG_HIP(hipStreamSynchronize(stream));
  auto t1 = std::chrono::steady_clock::now();
  if (profiling) prof_collect();
  collect_far();  // (as run_group: what the last ms parked counts in msgs.size() from now on)
  sync_globals_to_host();
  time = endAt;
  if (didSomething) *didSomething = gh.anyEvent ? 1 : 0;
  if (stats) {
    stats->delivered = (int64_t)(gh.delivered - before.delivered);
    stats->tasks = (int64_t)(gh.tasks - before.tasks);
    stats->events = (int64_t)(gh.events - before.events);
    stats->draws = (int64_t)(gh.draws - before.draws);
    stats->simulated_ms = ms;
    stats->wall_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
    int64_t pb = 0;
    for (int l = 0; l < 32; l++)
      pb += (int64_t)(gh.deliveredByLevel[l] - before.deliveredByLevel[l]) * proto->payload_bytes_of_level(l);
    stats->payload_bytes = pb;
  }
  check_device_errors();
}

// ------------------------------------------------------------------------------------------------
// Host-callback mode (WG_PROTO_HOST): the queue, its ordering, latency sampling and rd live in the engine,
// Message.action() stays with the caller. The device expands bucket `time` exactly as for a resident
// protocol (LIFO, chain runs unrolled: the same k_scan<ExpandF>); the events come to the host and are
// handed out one by one; what the caller pushes meanwhile is staged in push order and appended to the
// buckets by the same multisplit the device pipeline uses.
struct HostProto : ProtoHost {
  int dummy[4] = {0, 0, 0, 0};
  void launch_deliver(const Group&) override { throw WgError(WG_ESTATE, "host-callback mode has no device action()"); }
  size_t state_size() const override { return sizeof(dummy); }
  const void* state_host() const override { return dummy; }
  int host_msg_size(uint32_t) const override { return 0; }  // Message.size() is the caller's business here
};
ProtoHost* make_host_proto(Engine&) { return new HostProto(); }

// msgs.addMsg of a host push (C/Network.java:247-255): an envelope for the ms being handed out goes on top of
// what is left of it (LIFO: it is the next one delivered); everything else is staged for its bucket.
void Engine::hc_push(int32_t arrival, const Rec& rec) {
  if (dev.hostMode && hcLoaded && arrival == time) {
    HostEv ev;
    ev.rec = rec;
    ev.aux.chain = -1;
    ev.aux.cpos = 0;
    ev.aux.outBase = ev.aux.outCap = 0;
    if (rec_kind(rec) == K_CHAIN) throw WgError(WG_EUNSUPPORTED, "multi-destination envelope arriving in the millisecond being delivered");
    hcEvents.insert(hcEvents.begin() + (long)hcCursor, ev);
    return;
  }
  staged.push_back({arrival, rec});
}

void Engine::host_set_time(int32_t t) {
  if (!dev.hostMode) throw WgError(WG_ESTATE, "wg_set_time is for host-callback mode");
  // (nextMessage leaves time at until + 1; runMs then assigns endAt = until, C/Network.java:336)
  if (hcLoaded && hcCursor < hcEvents.size()) throw WgError(WG_ESTATE, "deliveries of the current ms are pending");
  if (hcLoaded) hc_finish_ms();
  time = t;
}

// m.markRead(); if (m.hasNextReader()) msgs.addMsg(m)  (C/Network.java:629-632), after the action()'s own pushes
void Engine::hc_stage_continuation() {
  if (hcContSlot < 0) return;
  const StagedChainKeep& k = hostChains[hcContSlot];
  const Chain& c = k.c;
  const int j = hcContPos;
  int32_t arrival;
  if (c.flags & 2u)
    arrival = k.words[(size_t)c.ndest + j];
  else
    arrival = c.sendTime + host_latency(c.from, k.words[j], c.seed);
  const int32_t slot = hcContSlot;
  hcContSlot = -1;
  if (arrival < time) throw WgError(WG_ESTATE, "Arriving in the past");
  if (arrival == time) throw WgError(WG_ESTATE, "chain run was not unrolled");  // expand unrolls same-ms hops
  staged.push_back({arrival, make_rec(K_CHAIN, c.from, (uint32_t)slot, (uint32_t)j, 0)});
}

// the ms handed out is exhausted: append what was pushed meanwhile, release the bucket, time++
void Engine::hc_finish_ms() {
  flush_staged(time, false);
  gh.now = time;
  globalsDirty = true;
  sync_globals_to_device();
  end_phase(self(), true);
  sync_globals_to_host();
  check_device_errors();
  hcLoaded = false;
  hcEvents.clear();
  hcCursor = 0;
}

bool Engine::next_delivery(int32_t until, int32_t condTime, wg_delivery* out) {
  if (!dev.hostMode) throw WgError(WG_ESTATE, "load WG_PROTO_HOST first");
  if (hcStepOpen) throw WgError(WG_ESTATE, "a batched step is open (wg_step_end)");
  for (;;) {
    hc_stage_continuation();  // owed by the delivery the caller has just applied
    while (hcLoaded && hcCursor < hcEvents.size()) {
      const HostEv ev = hcEvents[hcCursor++];
      if (ev.aux.chain >= 0 && ev.aux.cpos < 0) {  // last hop of a run: the envelope is re-pushed after action()
        const int next = (ev.aux.cpos & 0x7FFFFFFF) + 1;
        if (next < hostChains[ev.aux.chain].c.ndest) {
          hcContSlot = ev.aux.chain;
          hcContPos = next;
        } else {
          hostChains[ev.aux.chain].live = false;
          hcReleased.push_back(hostChains[ev.aux.chain].c.msg);  // the envelope's last hop
        }
      }
      if (ev.aux.chain < 0) hcReleased.push_back(ev.rec.w2);  // a single-destination envelope / a task ends with this event, delivered or consumed
      const int32_t from = rec_from(ev.rec), to = (int32_t)ev.rec.w1;
      if (hdown[to] || part_of(hx[from]) != part_of(hx[to])) {  // :606 — consumed, not delivered
        hc_stage_continuation();
        continue;
      }
      out->kind = rec_kind(ev.rec) == K_MSG ? 0 : 1;
      out->time = time;
      out->from = from;
      out->to = to;
      out->msg = ev.rec.w2;
      out->payload = ev.rec.w3;
      return true;
    }
    if (hcLoaded) {  // nextMessage(): poll returned null -> time++ and the conditional-task edge
      hc_finish_ms();
      time++;
      if (time >= condTime) {
        out->kind = 2;
        out->time = time;
        out->from = out->to = -1;
        out->msg = out->payload = 0;
        return true;
      }
      continue;
    }
    if (time > until) return false;
    hc_load_ms(until);
  }
}

// bucket `time` -> the host: expanded on the device as for a resident protocol, copied in two transfers
void Engine::hc_load_ms(int32_t until) {
  flush_staged(time, false);
  gh.now = time;
  gh.until = until;
  globalsDirty = true;
  sync_globals_to_device();
  Group g = self();
  expand(g);
  sync_globals_to_host();
  check_device_errors();
  const uint32_t n = gh.nEvents;
  hcEvents.resize(n);
  if (n) {
    std::vector<Rec> recs(n);
    std::vector<EvAux> aux(n);
    WG_HIP(hipMemcpy(recs.data(), dev.ev, sizeof(Rec) * n, hipMemcpyDeviceToHost));
    WG_HIP(hipMemcpy(aux.data(), dev.evAux, sizeof(EvAux) * n, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) hcEvents[i] = {recs[i], aux[i]};
  }
  hcCursor = 0;
  hcLoaded = true;
}

// ---- batched host-callback steps (wg_step_begin / wg_step_end) ---------------------------------------------------
// The same walk as next_delivery, but every deliverable envelope of the ms goes out in ONE call, and what the
// caller's action()s pushed comes back in ONE call, tagged with the delivery that issued it — the shape of
// External.receive(ei) -> List<SendMessage> (C/Network.java:616-623). step_end replays the pushes in the order the
// reference would have made them: delivery i's pushes, then the re-push of i's multi-destination envelope
// (markRead / hasNextReader / addMsg :629-632), then whatever an undeliverable envelope between i and i + 1 owes.
int32_t Engine::step_begin(int32_t until, int32_t condTime, wg_delivery* out, int32_t cap) {
  if (!dev.hostMode) throw WgError(WG_ESTATE, "load WG_PROTO_HOST first");
  if (hcStepOpen) throw WgError(WG_ESTATE, "wg_step_end has not closed the previous step");
  if (cap <= 0) throw WgError(WG_EINVAL, "cap");
  hcPlan.clear();
  for (;;) {
    hc_stage_continuation();  // (owed by a wg_next_delivery the caller mixed in)
    int32_t nOut = 0;
    while (hcLoaded && hcCursor < hcEvents.size() && nOut < cap) {
      const HostEv ev = hcEvents[hcCursor++];
      StepPlan pl{-1, -1, 0};
      if (ev.aux.chain >= 0 && ev.aux.cpos < 0) {  // last hop of a run: the envelope is re-pushed after action()
        const int next = (ev.aux.cpos & 0x7FFFFFFF) + 1;
        if (next < hostChains[ev.aux.chain].c.ndest) {
          pl.contSlot = ev.aux.chain;
          pl.contPos = next;
        } else {
          hostChains[ev.aux.chain].live = false;
          hcReleased.push_back(hostChains[ev.aux.chain].c.msg);
        }
      }
      if (ev.aux.chain < 0) hcReleased.push_back(ev.rec.w2);
      const int32_t from = rec_from(ev.rec), to = (int32_t)ev.rec.w1;
      if (!(hdown[to] || part_of(hx[from]) != part_of(hx[to]))) {  // :606
        wg_delivery& o = out[nOut];
        o.kind = rec_kind(ev.rec) == K_MSG ? 0 : 1;
        o.time = time;
        o.from = from;
        o.to = to;
        o.msg = ev.rec.w2;
        o.payload = ev.rec.w3;
        pl.outIdx = nOut++;
      }
      if (pl.outIdx >= 0 || pl.contSlot >= 0) hcPlan.push_back(pl);
    }
    if (nOut > 0) {
      hcStepOpen = true;
      return nOut;
    }
    if (!hcPlan.empty()) {  // only undeliverable envelopes were left in the ms: their re-pushes are filed here
      hcStepOpen = true;
      step_end(nullptr, 0, nullptr);
      continue;
    }
    if (hcLoaded) {  // nextMessage(): poll returned null -> time++ and the conditional-task edge
      hc_finish_ms();
      time++;
      if (time >= condTime) {
        out[0].kind = 2;
        out[0].time = time;
        out[0].from = out[0].to = -1;
        out[0].msg = out[0].payload = 0;
        hcStepOpen = true;  // (an edge is closed by wg_step_end too: the conditional tasks' pushes are its ops, after = 0)
        hcPlan.push_back({0, -1, 0});
        return 1;
      }
      continue;
    }
    if (time > until) return 0;
    hc_load_ms(until);
  }
}

int32_t Engine::host_released(uint32_t* msgs, int32_t cap) {
  if (!dev.hostMode) throw WgError(WG_ESTATE, "load WG_PROTO_HOST first");
  if (cap < 0 || (cap > 0 && !msgs)) throw WgError(WG_EINVAL, "msgs / cap");
  const size_t k = std::min((size_t)cap, hcReleased.size());
  // (oldest first; the handles of a delivery just handed out are reported too: the caller applies the delivery — it holds
  // the object — and forgets it afterwards)
  for (size_t i = 0; i < k; i++) msgs[i] = hcReleased[i];
  hcReleased.erase(hcReleased.begin(), hcReleased.begin() + (long)k);
  return (int32_t)k;
}

void Engine::step_end(const wg_step_op* ops, int32_t nops, const int32_t* dests) {
  if (!hcStepOpen) throw WgError(WG_ESTATE, "no step is open (wg_step_begin)");
  if (nops < 0 || (nops > 0 && !ops)) throw WgError(WG_EINVAL, "ops");
  // First pass: everything that can refuse an op is checked BEFORE anything is applied, and the step stays open when
  // it does — the ms has been handed out already, so a half-applied step (earlier ops staged, the multi-destination
  // envelopes' owed re-pushes dropped with the plan) could not be repaired by the caller. After a refusal the caller
  // resubmits (corrected ops, or none: the deliveries' own pushes are then lost, the envelopes' re-pushes are not).
  {
    const int32_t N = (int32_t)hx.size();
    int32_t k = 0;
    for (size_t pi = 0; pi < hcPlan.size(); pi++) {
      const StepPlan& pl = hcPlan[pi];
      if (pl.outIdx < 0) continue;
      if (k < nops && ops[k].after < pl.outIdx) throw WgError(WG_EINVAL, "wg_step_end: ops must be ordered by `after`");
      const bool last = pi + 1 == hcPlan.size();
      for (; k < nops && ops[k].after == pl.outIdx; k++) {
        const wg_step_op& op = ops[k];
        switch (op.kind) {
          case WG_OP_SEND: {
            if (op.n < 0) throw WgError(WG_EINVAL, "wg_step_op.n");
            if (op.n > 1 && !dests) throw WgError(WG_EINVAL, "dests");
            if (op.n > 1 && op.to < 0) throw WgError(WG_EINVAL, "wg_step_op.to (offset into dests)");
            if (op.from < 0 || op.from >= N) throw WgError(WG_EINVAL, "The from node is not in the network.");
            const int32_t* dd = op.n == 1 ? &op.to : dests + op.to;
            for (int i = 0; i < op.n; i++)
              if (dd[i] < 0 || dd[i] >= N) throw WgError(WG_EINVAL, "The to node is not in the network.");
            if (op.n > 0 && op.time <= time)
              throw WgError(WG_ESTATE, "sendTime=" + std::to_string(op.time) + ", time=" + std::to_string(time));  // :471
            break;
          }
          case WG_OP_SEND_ARRIVE_AT:
            if (op.from < 0 || op.from >= N || op.to < 0 || op.to >= N) throw WgError(WG_EINVAL, "node id");
            if (op.time <= time)
              throw WgError(WG_EINVAL, "wrong arrival time: arriveAt=" + std::to_string(op.time) + ", time=" + std::to_string(time));
            break;
          case WG_OP_TASK:
            if (op.from < 0 || op.from >= N) throw WgError(WG_EINVAL, "node id");
            if (op.time < time) throw WgError(WG_ESTATE, "Arriving in the past: arrival=" + std::to_string(op.time));
            // an envelope for the ms being handed out is the NEXT one delivered (LIFO, hc_push): exact only behind the step's last delivery
            if (op.time == time && hcLoaded && !last)
              throw WgError(WG_EUNSUPPORTED, "a task registered for the millisecond being delivered, from inside a batched step: "
                                             "resubmit this step without it and deliver such a protocol with wg_next_delivery or cap = 1");
            break;
          default: throw WgError(WG_EINVAL, "wg_step_op.kind");
        }
      }
    }
    if (k != nops) throw WgError(WG_EINVAL, "wg_step_end: an op refers to a delivery the step did not hand out");
    if (nops > 0 && !proto) throw WgError(WG_ESTATE, "load a protocol before sending (message sizes are protocol-defined)");
  }
  // Second pass: apply, in the reference's order (delivery i's pushes, then the re-push of i's envelope :629-632).
  hcStepOpen = false;
  const std::vector<StepPlan> plan = std::move(hcPlan);
  hcPlan.clear();
  int32_t k = 0;
  for (size_t pi = 0; pi < plan.size(); pi++) {
    const StepPlan pl = plan[pi];
    if (pl.outIdx >= 0) {
      for (; k < nops && ops[k].after == pl.outIdx; k++) {
        const wg_step_op& op = ops[k];
        switch (op.kind) {
          case WG_OP_SEND: {
            const int32_t* dd = op.n == 1 ? &op.to : dests + op.to;
            gh.draws++;  // (the caller drew the seed from the rd it holds during the step)
            globalsDirty = true;
            send_seeded(op.msg, op.payload, op.time, op.from, dd, op.n, op.delay, op.seed);
            break;
          }
          case WG_OP_SEND_ARRIVE_AT: send_arrive_at(op.msg, op.payload, op.time, op.from, op.to); break;
          default: register_task(op.msg, op.payload, op.time, op.from);
        }
      }
    }
    if (pl.contSlot >= 0) {
      hcContSlot = pl.contSlot;
      hcContPos = pl.contPos;
      hc_stage_continuation();
    }
  }
}

// ---- batches
Batch::Batch(Engine** es, int n) {
  if (n <= 0 || !es) throw WgError(WG_EINVAL, "empty batch");
  for (int i = 0; i < n; i++) {
    Engine* e = es[i];
    if (!e || !e->proto) throw WgError(WG_ESTATE, "every batch member needs a resident protocol");
    if (e->shardCount > 0) throw WgError(WG_EUNSUPPORTED, "a sharded engine cannot be a batch member");
    e->ensure_device();
    Engine* l = es[0];
    if (e->cfg.device != l->cfg.device || e->dev.horizon != l->dev.horizon || e->dev.nodes.n != l->dev.nodes.n ||
        e->proto->state_size() != l->proto->state_size() || e->proto->has_cond() != l->proto->has_cond() ||
        e->proto->levels() != l->proto->levels() || e->proto->variant() != l->proto->variant() ||
        (e->dev.inbox != nullptr) != (l->dev.inbox != nullptr))  // (the leader's delivery kernel reads every member's inbox lines)
      throw WgError(WG_EINVAL, "batch members must share device, protocol (and its attack scenario), node count, horizon_ms and "
                               "the inbox-line layout (load the protocol before the first call that allocates the engine)");
    for (int j = 0; j < i; j++)
      if (es[j] == e) throw WgError(WG_EINVAL, "an engine appears twice in the batch");
    members.push_back(e);
  }
  Engine& l = *members[0];
  WG_HIP(hipSetDevice(l.cfg.device));
  WG_HIP(hipMalloc((void**)&dTab, sizeof(EngineDev) * (size_t)n));
  WG_HIP(hipMalloc(&dStab, l.proto->state_size() * (size_t)n));
  WG_HIP(hipMalloc((void**)&dCont, sizeof(uint32_t) * (size_t)n));
}
Batch::~Batch() {
  if (dTab) (void)hipFree(dTab);
  if (dStab) (void)hipFree(dStab);
  if (dCont) (void)hipFree(dCont);
}
Group Batch::prepare(const uint8_t* active) {
  Engine& l = *members[0];
  const int n = (int)members.size();
  const size_t sz = l.proto->state_size();
  std::vector<EngineDev> tab(n);
  std::vector<char> stab(sz * n);
  for (int r = 0; r < n; r++) {
    members[r]->upload_down();
    tab[r] = members[r]->dev;
    tab[r].halted = (active && !active[r]) ? 1u : 0u;
    memcpy(stab.data() + sz * r, members[r]->proto->state_host(), sz);
  }
  bool sync = false;
  if (hTab.size() != tab.size() || memcmp(hTab.data(), tab.data(), sizeof(EngineDev) * n) != 0) {
    hTab = tab;
    WG_HIP(hipMemcpyAsync(dTab, hTab.data(), sizeof(EngineDev) * n, hipMemcpyHostToDevice, l.stream));
    sync = true;
  }
  if (hStab != stab) {
    hStab = stab;
    WG_HIP(hipMemcpyAsync(dStab, hStab.data(), hStab.size(), hipMemcpyHostToDevice, l.stream));
    sync = true;
  }
  if (sync) WG_HIP(hipStreamSynchronize(l.stream));
  Group g;
  g.tab = dTab;
  g.stab = dStab;
  g.R = n;
  g.stream = l.stream;
  g.binBits = l.binBits;
  g.histLds = sizeof(uint32_t) * (size_t)l.dev.horizon;
  g.mergeAppend = merge_append_setting();
  g.nodes = l.dev.nodes.n;
  // the periodic tasks registered on ANY member (Group::periodic_may_fire)
  periodicUnion.clear();
  bool known = true;
  for (int r = 0; r < n && known; r++) {
    known = !members[r]->periodicUnknown;
    for (const PeriodicReg& x : members[r]->periodicRegs) {
      bool seen = false;
      for (const PeriodicReg& y : periodicUnion) seen = seen || (x.task == y.task && x.period == y.period && x.phase == y.phase);
      if (!seen) periodicUnion.push_back(x);
    }
    if (periodicUnion.size() > 256) known = false;
  }
  g.periodic = known ? &periodicUnion : nullptr;
  return g;
}
void Batch::run_ms(int32_t ms, const uint8_t* active, uint8_t* did, wg_run_stats* stats) {
  if (ms <= 0) throw WgError(WG_EINVAL, "Should be greater than 0. ms=" + std::to_string(ms));
  bool any = false;
  for (size_t r = 0; r < members.size(); r++) any |= !active || active[r];
  if (!any) {
    if (did) memset(did, 0, members.size());
    if (stats) memset(stats, 0, sizeof(wg_run_stats) * members.size());
    return;
  }
  WG_HIP(hipSetDevice(members[0]->cfg.device));
  Group g = prepare(active);
  Engine::run_group(members.data(), (int)members.size(), active, g, ms, did, stats);
}
void Batch::cont_if(int32_t* out) {
  Engine& l = *members[0];
  WG_HIP(hipSetDevice(l.cfg.device));
  for (Engine* e : members) e->flush_staged(e->time, false);
  Group g = prepare(nullptr);
  const int n = (int)members.size();
  WG_HIP(hipMemsetAsync(dCont, 0, sizeof(uint32_t) * n, l.stream));
  if (!l.proto->launch_cont_if(g, dCont))
    throw WgError(WG_EUNSUPPORTED, "the resident protocol defines no continuation predicate");
  std::vector<uint32_t> v(n);
  WG_HIP(hipMemcpyAsync(v.data(), dCont, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, l.stream));
  WG_HIP(hipStreamSynchronize(l.stream));
  for (int r = 0; r < n; r++) out[r] = (int32_t)v[r];
}


// RunMultipleTimes.run's inner loop (C/RunMultipleTimes.java:50-64) for every member without a host round
// trip per runMs: chunks are enqueued back to back (k_chunk_begin, the ms sequence, the predicate kernel,
// k_chunk_end); the host only looks at the number of members still running every few chunks.
void Batch::run_multiple_times(int32_t chunk, int32_t maxTime, int64_t* delivered, int64_t* simulatedMs) {
  if (chunk <= 0) throw WgError(WG_EINVAL, "Should be greater than 0. ms=" + std::to_string(chunk));
  const int n = (int)members.size();
  Engine& l = *members[0];
  WG_HIP(hipSetDevice(l.cfg.device));
  std::vector<Globals> before(n);
  std::vector<int32_t> t0(n);
  bool hostLoop = false;
  for (int r = 0; r < n; r++) {
    Engine& e = *members[r];
    e.flush_staged(e.time, false);
    if (e.stagedMin != INT32_MAX) hostLoop = true;  // envelopes held on the host beyond the bucket ring
  }
  for (int r = 0; r < n; r++)
    if (members[r]->dev.farBuf) hostLoop = true;  // its far envelopes are collected by the host between milliseconds
  if (hostLoop) throw WgError(WG_EUNSUPPORTED, "host-held envelopes beyond horizon_ms: use wg_batch_run_ms per chunk");
  for (int r = 0; r < n; r++) {
    Engine& e = *members[r];
    before[r] = e.gh;
    t0[r] = e.time;
    e.gh.until = e.time;  // k_chunk_begin reads Network.time from `until`
    e.globalsDirty = true;
    e.sync_globals_to_device();
  }
  Group g = prepare(nullptr);
  EngineDev* tab = dTab;
  uint32_t* dRunning = nullptr;
  WG_HIP(hipMalloc((void**)&dRunning, sizeof(uint32_t)));
  const int CHECK_EVERY = 4;
  uint32_t running = (uint32_t)n;
  // one runMs(chunk) of every member: the same launch sequence every time (the kernels read the clock and the loop
  // state from device memory), which is what makes it a graph
  int32_t tChunk = t0[0];  // Network.time at the head of the next chunk, if the members share a clock (Group::now)
  for (int r = 1; r < n; r++)
    if (t0[r] != t0[0]) tChunk = INT32_MIN;
  auto enqueue_chunk = [&]() {
    hipLaunchKernelGGL(k_chunk_begin, dim3(n), dim3(64), 0, g.stream, tab, chunk);
    Engine::enqueue_ms_sequence(l, g, chunk, tChunk);
    if (tChunk != INT32_MIN) tChunk += chunk;
    WG_HIP(hipMemsetAsync(dCont, 0, sizeof(uint32_t) * n, g.stream));
    if (!l.proto->launch_cont_if(g, dCont))
      throw WgError(WG_EUNSUPPORTED, "the resident protocol defines no continuation predicate");
    WG_HIP(hipMemsetAsync(dRunning, 0, sizeof(uint32_t), g.stream));
    hipLaunchKernelGGL(k_chunk_end, dim3(n), dim3(64), 0, g.stream, tab, dCont, maxTime, dRunning);
  };
  // WG_GRAPH=1: the chunk is captured once into a hipGraph and replayed — one graph launch instead of ~30 kernel
  // launches per simulated ms. The profiler's brackets are device clock stamps there (ProfScope / k_prof_stamp: a HIP
  // event captured into the graph would keep its last replay only).
  const bool wantGraph = getenv("WG_GRAPH") && atoi(getenv("WG_GRAPH")) != 0;  // (read per call)
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  try {
    if (wantGraph) {
      tChunk = INT32_MIN;  // (one captured launch sequence serves every chunk: nothing in it may depend on the ms)
      if (l.profiling) l.prof_stamp(2 * Engine::PC_COUNT);  // (allocates the ring outside the capture; an unpaired tag)
      WG_HIP(hipStreamSynchronize(g.stream));
      WG_HIP(hipStreamBeginCapture(g.stream, hipStreamCaptureModeRelaxed));
      l.profStamping = true;
      try {
        enqueue_chunk();
      } catch (...) {
        l.profStamping = false;
        (void)hipStreamEndCapture(g.stream, &graph);
        throw;
      }
      l.profStamping = false;
      WG_HIP(hipStreamEndCapture(g.stream, &graph));
      WG_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    }
    while (running) {
      for (int k = 0; k < CHECK_EVERY; k++) {
        if (exec)
          WG_HIP(hipGraphLaunch(exec, g.stream));
        else
          enqueue_chunk();
      }
      WG_HIP(hipMemcpyAsync(&running, dRunning, sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
      WG_HIP(hipStreamSynchronize(g.stream));
    }
  } catch (...) {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipFree(dRunning);
    throw;
  }
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipFree(dRunning);
  if (l.profiling) l.prof_collect();
  hTab.clear();  // the device table now carries halted flags the host shadow does not
  std::string firstErr;
  int32_t firstCode = WG_OK;
  for (int r = 0; r < n; r++) {
    Engine& e = *members[r];
    e.sync_globals_to_host();
    e.time = e.gh.until;
    if (delivered) delivered[r] = (int64_t)(e.gh.delivered - before[r].delivered);
    if (simulatedMs) simulatedMs[r] = (int64_t)e.time - t0[r];
    try {
      e.check_device_errors();
    } catch (const WgError& x) {
      e.lastError = x.what();
      if (firstCode == WG_OK) {
        firstCode = x.code;
        firstErr = x.what();
      }
    }
  }
  if (firstCode != WG_OK) throw WgError(firstCode, firstErr);
}

int64_t Engine::queue_size() {  // msgs.size(): number of envelopes (a multi-dest envelope counts once)
  if (!allocated) return (int64_t)staged.size();
  flush_staged(time, false);
  std::vector<uint32_t> c(dev.horizon);
  WG_HIP(hipMemcpy(c.data(), dev.bcnt, sizeof(uint32_t) * dev.horizon, hipMemcpyDeviceToHost));
  int64_t s = (int64_t)staged.size();
  for (uint32_t v : c) s += v;
  return s;
}
int64_t Engine::queue_size_at(int32_t t) {
  if (t < time) return 0;
  int64_t far = 0;
  if (allocated) flush_staged(time, false);
  for (auto& st : staged) far += st.arrival == t;
  if (!allocated || t - time >= dev.horizon) return far;
  uint32_t c;
  WG_HIP(hipMemcpy(&c, dev.bcnt + ((uint32_t)t & (dev.horizon - 1)), 4, hipMemcpyDeviceToHost));
  return c;
}

void Engine::read_i64(int32_t field, int64_t* dst, int32_t n) {
  if (n != (int32_t)hx.size()) throw WgError(WG_EINVAL, "n must equal the node count");
  ensure_device();
  flush_staged(time, false);
  auto rd64 = [&](const long long* src) {
    WG_HIP(hipMemcpy(dst, src, 8 * (size_t)n, hipMemcpyDeviceToHost));
    if (proto) proto->node_counter(*this, field, dst, n);  // (a protocol may keep its share of a Node counter in its own records)
  };
  switch (field) {
    case WG_F_DONE_AT: return rd64(dev.nodes.doneAt);
    case WG_F_MSG_RECEIVED: return rd64(dev.nodes.msgReceived);
    case WG_F_MSG_SENT: return rd64(dev.nodes.msgSent);
    case WG_F_BYTES_SENT: return rd64(dev.nodes.bytesSent);
    case WG_F_BYTES_RECEIVED:  // (a protocol whose messages all have size() 1 keeps the two counters as one)
      return rd64(proto && proto->unit_message_size() ? dev.nodes.msgReceived : dev.nodes.bytesReceived);
    case WG_F_DOWN:
      for (int i = 0; i < n; i++) dst[i] = hdown[i];
      return;
    case WG_F_X:
      for (int i = 0; i < n; i++) dst[i] = hx[i];
      return;
    case WG_F_Y:
      for (int i = 0; i < n; i++) dst[i] = hy[i];
      return;
    case WG_F_EXTRA_LATENCY:
      for (int i = 0; i < n; i++) dst[i] = hextra[i];
      return;
    default:
      if (proto && proto->read_i64(*this, field, dst, n)) return;
      throw WgError(WG_EINVAL, "unknown field for the resident protocol");
  }
}

}  // namespace wg

// ================================================================================================
// Handel resident protocol: host side (allocation, upload of init() state, kernel launches, read-back)
#include "proto_handel.hip.h"

namespace wg {

template void Engine::scan<CondF>(const Group&, const HandelState*);
typedef SnapF<HandelState, H_TASK_DISSEMINATION> HandelSnapF;
template void Engine::scan<HandelSnapF>(const Group&, const HandelState*);

__global__ void k_handel_init(HandelState s, const uint8_t* down, const int32_t* startAt, const int32_t* pairing) {
  int node = s.lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= s.hi) return;
  uint32_t* h = h_hdr(s, node);
  h[HH_START] = (uint32_t)startAt[node];
  h[HH_PAIR] = (uint32_t)pairing[node];
  // HLevel() for level 0 (:413-421): own signature everywhere, outgoingFinished = true
  uint64_t bit = 1ULL << (node & 63);
  *h_row(s, node, HK_TI, 0) |= bit;
  *h_row(s, node, HK_LA, 0) |= bit;
  *h_row(s, node, HK_VI, 0) |= bit;
  *h_lv(s, node, HP_CTI, 0) = 1;
  *h_lv(s, node, HP_CLA, 0) = 1;
  *h_lv(s, node, HP_CVI, 0) = 1;
  *h_lv(s, node, HP_OUTFIN, 0) = 1;
  h[HH_TOTAL] = 1;  // the sum of |totalIncoming| over the levels: the own signature
  h[HH_WINDOW] = (uint32_t)s.p.windowInitial;
  h[HH_ADDED] = (uint32_t)s.p.extraCycle;
  for (int l = 0; l < s.L; l++) *h_lv(s, node, HP_SPARE0, l) = s.atk == 1 ? 0u : 0xFFFFFFFFu;  // suicideBizAfter :406
  // registerConditionalTask(checkSigs, startAt + 1, nodePairingTime, ...) for live nodes (:979-982)
  s.ct[2 * (size_t)node] = (uint32_t)(down[node] ? INT32_MAX : startAt[node] + 1);
}

// Handel.newContIf (P/Handel.java:1044-1053): some live node has doneAt == 0 or addedCycle > 0
__global__ void k_handel_cont_if(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab, uint32_t* out) {
  const EngineDev& d = tab[blockIdx.y];
  const HandelState& s = stab[blockIdx.y];
  // A FEW wavefronts per engine, each walking its share of the nodes 64 at a time and stopping at the first such node: for
  // most of a run that is the first 64 it looks at. (One wavefront per 64 nodes, each reporting through the same word, was
  // 12 k same-address atomics — or stores — per call at 24 copies of 32 768 nodes: 100 - 150 us of every tenth ms.)
  const int stride = (int)(gridDim.x * blockDim.x);
  for (int n0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) & ~63u); n0 < s.hi; n0 += stride) {
    if (__ballot(cont_if_known(out + blockIdx.y))) return;  // (settled by another wavefront; wave-uniform)
    const int node = n0 + (int)WG_LANE;  // (sharded: the predicate over this shard's nodes)
    const bool live = node >= s.lo && node < s.hi && !d.nodes.down[node];
    // doneAt is a dense array; addedCycle sits in the node's 640-byte record (a line per node): it only matters for nodes
    // that are done, i.e. in the last extraCycle periods of a run
    bool c = live && d.nodes.doneAt[node] == 0;
    if (!__ballot(c)) c = live && (int32_t)h_hdr(s, node)[HH_ADDED] > 0;
    if (__ballot(c)) {
      if (WG_LANE == 0) cont_if_set(out + blockIdx.y);
      return;
    }
  }
}

// wg_restore: receptionRanks as init() left them. The only writer is checkSigs' `receptionRanks[from] += nodeCount`
// (k_handel_cond_a2, P/Handel.java:825-828) on values that start in [0, nodeCount), nodeCount a power of two: the low
// bits are the initial rank (a saturated entry would not be: NOTE_RANKS_SATURATED refuses the restore).
__global__ void __launch_bounds__(256) k_handel_ranks_reset(HandelState s) {
  const size_t n4 = ((size_t)(s.hi - s.lo) * (size_t)s.N) >> 2;  // (N >= 4 here: rows are whole 16-byte vectors)
  U4* r = (U4*)(s.ranks + (size_t)s.lo * s.N);
  const uint32_t m = (uint32_t)s.N - 1u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    U4 q = r[i];
    if ((q.x | q.y | q.z | q.w) & ~m) {
      q.x &= m;
      q.y &= m;
      q.z &= m;
      q.w &= m;
      r[i] = q;
    }
  }
}

// HLevel() for level 0 (P/Handel.java:413-421): the node's own signature in totalIncoming / lastAggVerified /
// verifiedIndSignatures; everything else of the five rows is empty after init()
__global__ void k_handel_own_bits(HandelState s) {
  int node = s.lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= s.hi) return;
  uint64_t bit = 1ULL << (node & 63);
  *h_row(s, node, HK_TI, 0) = bit;
  *h_row(s, node, HK_LA, 0) = bit;
  *h_row(s, node, HK_VI, 0) = bit;
  // the verification queues are empty after init(): a run that was cut short (maxTime, an error) leaves entries behind
  for (int l = 0; l < s.L; l++) {
    uint64_t* qr = h_qrec(s, node, l);
    qr[0] = qr[1] = 0;
    qr[H_QVALID] = 0;
    qr[s.qBad] = 0;
  }
}

// read-back: toVerifyAgg.size() of every (owned node, level)
__global__ void __launch_bounds__(256) k_handel_gather_qlen(HandelState s, int32_t* dst) {
  const size_t n = (size_t)(s.hi - s.lo) * s.L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (int32_t)h_qrec(s, s.lo + (int32_t)(i / s.L), (int)(i % s.L))[0];
}
// read-back: kind k of every owned node as an N-bit row in id order (the layout the reference's BitSets have)
__global__ void __launch_bounds__(256) k_handel_gather_row(HandelState s, int k, uint64_t* dst) {
  const size_t n = (size_t)(s.hi - s.lo) * s.W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int32_t node = s.lo + (int32_t)(i / s.W);
    const int x = (int)(i % s.W);
    if (k < HK_COUNT)
      dst[i] = *h_word(s, node, k, x);
    else if (k == HK_COUNT + HD_SEEN)  // toVerifyInd
      dst[i] = h_dword_x(s, node, x)[HD_SEEN] & ~*h_word(s, node, HK_VI, x);
    else
      dst[i] = h_dword_x(s, node, x)[k - HK_COUNT];
  }
}
// read-back of the receptionRanks of a run whose ranks the senders carry: s.ranks holds the re-shuffled initial matrix;
// every bumped sender of every node gets its nodeCount per bump (P/Handel.java:825-828)
__global__ void __launch_bounds__(256) k_handel_ranks_add_bumps(HandelState s) {
  for (int node = blockIdx.x; node < s.N; node += gridDim.x)
    for (int i = threadIdx.x; i < s.bumpCap; i += blockDim.x) {
      const uint32_t e = s.bump[(size_t)node * s.bumpCap + i];
      if (e >> 16) {
        int32_t* r = s.ranks + (size_t)node * s.N + (e & 0xFFFFu);
        *r = h_rank_of(s, (uint32_t)*r, e >> 16);
      }
    }
}

// The N x N matrix a CARRIED init() shuffles its ranks into before the emission lists take them over: 4.3 GB at 32 768
// nodes, and a batch's copies run their init() on several host threads (bench.py, replicas.init_threads) — a batch sized
// to fill the HBM with finished copies must not hold one matrix per thread on top. At most four in flight per process;
// a thread whose allocation fails waits for another's release before it gives up.
struct TmpMatrix {
  static std::mutex& mu() {
    static std::mutex m;
    return m;
  }
  static std::condition_variable& cv() {
    static std::condition_variable c;
    return c;
  }
  static int& inFlight() {
    static int n = 0;
    return n;
  }
  static size_t& inFlightBytes() {
    static size_t n = 0;
    return n;
  }
  // at most two at a time and at most IN_FLIGHT_BYTES between them (one alone may be larger): what a host that sizes a
  // batch of copies to the free HBM has to leave for init() (wittgenstein_amd/replicas.py::handel_init_transient_bytes).
  // (Four / 16 GiB until round 6: two are what lets the 32nd copy of the default bench line fit — with an engine pinned to
  // one XCD, engine_kernels.hip.h wg_place, a batch's step time goes by ceil(copies / 8) — for 3 s more of init().)
  static constexpr int IN_FLIGHT_MAX = 2;
  static constexpr size_t IN_FLIGHT_BYTES = 2ull * 4ull * 32768ull * 32768ull;
  static std::map<int32_t*, size_t>& sizes() {
    static std::map<int32_t*, size_t> m;
    return m;
  }
  static int32_t* acquire(size_t bytes) {
    std::unique_lock<std::mutex> lk(mu());
    for (;;) {
      cv().wait(lk, [bytes] { return inFlight() == 0 || (inFlight() < IN_FLIGHT_MAX && inFlightBytes() + bytes <= IN_FLIGHT_BYTES); });
      int32_t* p = nullptr;
      if (hipMalloc((void**)&p, bytes) == hipSuccess) {
        inFlight()++;
        inFlightBytes() += bytes;
        sizes()[p] = bytes;
        return p;
      }
      (void)hipGetLastError();
      if (inFlight() == 0) throw WgError(WG_ENOMEM, "Handel init(): no room for the nodeCount^2 reception-rank matrix the emission lists are built from");
      const int seen = inFlight();
      cv().wait(lk, [seen] { return inFlight() < seen; });
    }
  }
  static void release(int32_t*& p) {
    if (!p) return;
    (void)hipFree(p);
    {
      std::lock_guard<std::mutex> lk(mu());
      inFlight()--;
      auto it = sizes().find(p);
      if (it != sizes().end()) {
        inFlightBytes() -= it->second;
        sizes().erase(it);
      }
    }
    p = nullptr;
    cv().notify_all();
  }
};

struct HandelHost : ProtoHost {
  HandelState st{};
  Engine& eng;
  uint32_t* dCont = nullptr;
  // (the register allocations of the latency-bound kernels — waves per SIMD: k_handel_wave 4, k_handel_a1c 5,
  // k_handel_update 8, k_handel_dissem 8 — are the measured optima of the sweeps in profiles/INDEX.md; DESIGN.md "Occupancy")
  HandelHost(Engine& e, const wg_handel_params& p, const wg_handel_init_state& init) : eng(e) {
    const auto tCtor = std::chrono::steady_clock::now();
    const int32_t N = p.nodeCount;
    if ((int32_t)e.hx.size() != N) throw WgError(WG_EINVAL, "Handel nodeCount != nodes in the network");
    if (N < 2 || (N & (N - 1))) throw WgError(WG_EINVAL, "We support only power of two nodes in this simulation");
    if (!init.startAt || !init.nodePairingTime || (!init.receptionRanks && init.peers))
      throw WgError(WG_EINVAL, "wg_handel_init_state has NULL members");
    if (p.byzantineSuicide && p.hiddenByzantine) throw WgError(WG_EINVAL, "Only one attack at a time");  // :123-125
    if ((p.byzantineSuicide || p.hiddenByzantine) && e.shardCount > 0)
      throw WgError(WG_EUNSUPPORTED, "Handel byzantineSuicide / hiddenByzantine on a sharded engine");
    int L = 1;
    while ((1 << L) <= N) L++;  // levels 0..log2(N)
    if (L > MAX_LEVELS) throw WgError(WG_EINVAL, "too many levels");
    const int W = N >= 64 ? N / 64 : 1;
    const int Q = e.cfg.queue_cap > 0 ? e.cfg.queue_cap : 32;
    if (Q > 64) throw WgError(WG_EINVAL, "queue_cap must be <= 64");
    // engine payload ring: only fast-path sends (:738-749) snapshot into it — at most one per (node, level)
    // completion; the periodic dissemination snapshots have computed addresses (HandelState::snap).
    // (the engine's payload ring is not used: a fast-path send's snapshot is the constant all-ones block, snapshot_outgoing)
    if (N > (1 << 19)) throw WgError(WG_EINVAL, "Handel resident: at most 2^19 nodes (the signer's id travels in the task word)");
    if (e.allocated && !e.dev.inbox) throw WgError(WG_ESTATE, "load Handel before the first call that allocates the engine");
    e.wantInbox = true;  // a node's events of the ms are read from its inbox line (k_handel_lane / k_handel_wave)
    e.ensure_device();
    if (e.dev.maxOut >= (1u << 28)) throw WgError(WG_EINVAL, "outbox_records must be below 2^28 (an inbox entry carries a task's outbox slot in 28 bits)");
    if (p.disseminationPeriodMs >= e.dev.horizon) throw WgError(WG_ENOMEM, "horizon_ms <= dissemination period");
    st.p = p;
    st.N = N;
    st.L = L;
    st.W = W;
    st.Q = Q;
    st.Qw = e.cfg.queue_cap_wide > 0 ? e.cfg.queue_cap_wide : std::min(Q, 16);
    if (st.Qw > Q) throw WgError(WG_EINVAL, "queue_cap_wide must be <= queue_cap");
    // Per-node rows are held for the nodes [lo, hi) this engine owns — everything when it is not sharded — behind
    // pointers biased by -lo rows, so that the kernels keep indexing them by node id (they touch only owned nodes).
    const int32_t lo = st.lo = e.shardCount > 0 ? e.dev.shardLo : 0;
    const int32_t hi = st.hi = e.shardCount > 0 ? e.dev.shardHi : N;
    const size_t nLoc = (size_t)(hi - lo);
    auto rows = [&](auto* tag, size_t stride, bool zero, int cls = Engine::AC_STATE) {
      typedef std::remove_pointer_t<decltype(tag)> T;
      return e.dalloc<T>(nLoc * stride, zero, cls) - (size_t)lo * stride;
    };
    // (wg_restore re-creates the five bit rows — zero but for the node's own signature, k_handel_own_bits — instead of
    // keeping a copy: AC_SCRATCH)
    st.rows = rows((uint64_t*)nullptr, (size_t)W * HK_COUNT, true, Engine::AC_SCRATCH);
    st.drows = rows((uint64_t*)nullptr, (size_t)W * HD_COUNT, true, Engine::AC_SCRATCH);
    // The reception ranks CARRIED by the senders (proto_handel.hip.h, file header) where init() runs on the device — an
    // unsharded engine of 256 .. 65 536 nodes without an attack (the attacks read receptionRanks of peers that never sent:
    // :545, :868); WG_HANDEL_RANKS=matrix keeps the N x N matrix there too (A/B, tests). Host-built init() hands over the
    // matrix, and so it stays.
    carried = !init.receptionRanks && !init.peers && !p.byzantineSuicide && !p.hiddenByzantine && e.shardCount == 0 && N >= 256 &&
              N <= 65536 && !(getenv("WG_HANDEL_RANKS") && !strcmp(getenv("WG_HANDEL_RANKS"), "matrix"));
    // wg_snapshot / wg_restore: the emission lists are never written; receptionRanks only by `+= nodeCount`
    // (k_handel_cond_a2), which on_restore undoes in place (matrix) or forgets (the bump tables); the verification queues,
    // the dissemination snapshots and the scratch of the conditional-task phase hold nothing before the first event
    int32_t* tmpRanks = nullptr;  // CARRIED: the matrix lives for the length of init() only
    struct FreeTmp {
      int32_t*& p;
      ~FreeTmp() { TmpMatrix::release(p); }
    } guardTmp{tmpRanks};
    const bool peers16 = N <= 65536;
    uint16_t* dPeers16 = nullptr;
    int32_t* dPeers32 = nullptr;
    if (carried) {
      tmpRanks = TmpMatrix::acquire(4 * (size_t)N * N);
      st.ranks = tmpRanks;
      st.peersR = rows((uint32_t*)nullptr, N - 1, false, Engine::AC_CONST);
      // (config 3: a node verifies ~ 100 distinct senders, 148 at most — 512 entries; a network of up to 4096 nodes gets the
      // direct-indexed table of N entries, <= 64 MB, which cannot overflow: small networks with many nodesDown or long horizons
      // verify far more senders per node than the big run does, and the matrix form ran them out of the box — ADVICE.md round 5)
      int cap = e.cfg.rank_bump_cap > 0 ? e.cfg.rank_bump_cap : (N <= 4096 ? N : 512);
      int p2 = 1;
      while (p2 < cap && p2 < N) p2 <<= 1;
      st.bumpCap = std::min(p2, (int)N);
      st.bump = rows((uint32_t*)nullptr, (size_t)st.bumpCap, true, Engine::AC_SCRATCH);
      e.dev.destTagged = 1;  // a fast-path envelope's destination words carry the rank (id | rank << 16)
      e.dev.destTagMsgShift = H_MSG_RANK_SHIFT;
    } else {
      st.ranks = rows((int32_t*)nullptr, N, false, Engine::AC_SCRATCH);
      dPeers16 = peers16 ? rows((uint16_t*)nullptr, N - 1, false, Engine::AC_CONST) : nullptr;
      dPeers32 = peers16 ? nullptr : rows((int32_t*)nullptr, N - 1, false, Engine::AC_CONST);
      st.peersR = nullptr;
      st.bump = nullptr;
      st.bumpCap = 0;
    }
    st.peers16 = dPeers16;
    st.peers32 = dPeers32;
    st.LS = L <= 16 ? 16 : 32;
    st.lsShift = L <= 16 ? 4 : 5;
    st.hdrStride = HH_LV + HP_COUNT * st.LS;  // 160 or 288 words: whole 128-byte lines
    st.hdr = rows((uint32_t*)nullptr, st.hdrStride, true);
    st.ct = rows((uint32_t*)nullptr, 2, true);
    const size_t NL = (size_t)N * L;
    st.qBad = H_QENT + Q;
    st.qStride = (st.qBad + 1 + 7) & ~7;
    // (+ 64 words behind the last record: a wavefront reads `ent[lane]` of a long list with every lane before it masks by the length)
    st.qrec = e.dalloc<uint64_t>(nLoc * (size_t)L * st.qStride + 64, true, Engine::AC_SCRATCH) - (size_t)lo * L * st.qStride;
    unsigned long long off = 0;
    for (int l = 0; l < L; l++) {
      int nw = l == 0 ? 0 : ((1 << (l - 1)) >= 64 ? (1 << (l - 1)) >> 6 : 1);
      const int ql = nw >= 16 ? st.Qw : Q;  // (h_qcap)
      st.qsigOff[l] = off - (unsigned long long)lo * ql * nw;  // (node * ql + slot) * nw is added to it: biased like the rows
      off += (unsigned long long)nLoc * ql * nw;
    }
    st.qsig = e.dalloc<uint64_t>(off, false, Engine::AC_SCRATCH);
    // the cached evaluations of the listed signatures (HandelState::qcache): read only where the record's valid mask says so
    st.QC = (Q + 3) & ~3;
    // (+ 8 words: the group form of k_handel_a1 preloads the words of the slots 0 .. 7 whatever the capacity)
    st.qcache = e.dalloc<uint32_t>(nLoc * (size_t)L * st.QC + 8, false, Engine::AC_SCRATCH) - (size_t)lo * L * st.QC;
    {
      st.snapStride = N >= 128 ? (uint32_t)(N / 128) : 1u;  // words of the top level's block (N/2 ids)
      st.snapNb = (uint32_t)(e.dev.horizon / p.disseminationPeriodMs) + 2;  // a snapshot is read within < horizon ms
      uint64_t words = (uint64_t)st.snapNb * N * st.snapStride;
      if (words >= 0x80000000ull) throw WgError(WG_ENOMEM, "Handel snapshot ring exceeds 2^31 words: lower horizon_ms");
      if ((uint64_t)e.dev.payloadWords >= 0x80000000ull) throw WgError(WG_EINVAL, "payload_words must be < 2^31");
      st.snap = e.dalloc<uint64_t>(words, false, Engine::AC_SCRATCH);
    }
    e.dev.boundMsg = 0;            // onNewSig never sends
    e.dev.boundTask[0] = L - 1;    // dissemination: one send per level >= 1 (+1 periodic re-arm added by expand)
    e.dev.boundTask[1] = L - 1;    // updateVerifiedSignatures: one fast-path send per higher level
    e.dev.boundTask[2] = e.dev.boundTask[3] = 0;
    // the conditional-task phase's scratch: (node, level) items of the edge, candidate levels per node, the draws
    st.itemsLane = e.dalloc<uint32_t>((size_t)nLoc * L, false, Engine::AC_SCRATCH);
    st.itemsWave = e.dalloc<uint32_t>((size_t)nLoc * L, false, Engine::AC_SCRATCH);
    st.itemCount = e.dalloc<uint32_t>(2);
    st.jobs = e.dalloc<CopyJob>(e.dev.maxEvents, false, Engine::AC_SCRATCH);
    st.jobCount = e.dalloc<uint32_t>(1);
    st.jobsSmall = e.dalloc<CopyJob>(e.dev.maxEvents, false, Engine::AC_SCRATCH);
    st.jobSmallCount = e.dalloc<uint32_t>(1);
    st.itemsUpd = e.dalloc<U4>(nLoc, false, Engine::AC_SCRATCH);
    st.updCount = e.dalloc<uint32_t>(1);
    st.itemsTrail = e.dalloc<U4>(nLoc, false, Engine::AC_SCRATCH);
    st.trailCount = e.dalloc<uint32_t>(1);
    st.itemsTrail2 = e.dalloc<U4>(nLoc, false, Engine::AC_SCRATCH);
    st.trail2Count = e.dalloc<uint32_t>(1);
    st.itemsDis = e.dalloc<U4>(nLoc, false, Engine::AC_SCRATCH);
    st.disCount = e.dalloc<uint32_t>(1);
    st.atk = p.byzantineSuicide ? 1 : p.hiddenByzantine ? 2 : 0;
    st.laneNw = getenv("WG_LANE_NW") ? std::max(1, std::min(H_LANE_NW, atoi(getenv("WG_LANE_NW")))) : H_LANE_NW;
    st.blacklist = st.atk == 1 ? e.dalloc<uint64_t>((size_t)N * W, true, Engine::AC_SCRATCH) : nullptr;
    st.candMask = e.dalloc<uint32_t>(N);
    st.cleanMask = e.dalloc<uint32_t>(N);
    st.condList = e.dalloc<uint32_t>(N, true, Engine::AC_SCRATCH);
    st.drawVal = e.dalloc<int32_t>(N, true, Engine::AC_SCRATCH);
    const bool verbose = getenv("WG_INIT_VERBOSE") && atoi(getenv("WG_INIT_VERBOSE"));
    auto tv = tCtor;
    auto lap = [&](const char* what) {
      if (!verbose) return;
      (void)hipStreamSynchronize(e.stream);
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[wittgpu] handel load: %s %.3f s\n", what, std::chrono::duration<double>(now - tv).count());
      tv = now;
    };
    lap("allocations");
    if (verbose)
      fprintf(stderr, "[wittgpu] handel load: reception ranks %s (bump table: %d senders per node)\n",
              carried ? "carried by the senders, no matrix" : "as an N x N matrix", st.bumpCap);
    if (init.receptionRanks) {
      WG_HIP(hipMemcpy(st.ranks + (size_t)lo * N, init.receptionRanks + (size_t)lo * N, 4 * nLoc * N, hipMemcpyHostToDevice));
    } else {
      rngAtRanks = e.gh.rng;
      build_ranks(e, st, true);  // every node's Collections.shuffle on the device (k_handel_init_scan / _perm / _chain)
    }
    lap("reception ranks");
    if (!init.peers) {
      build_peers(e);  // buildEmissionList on the device (k_handel_init_sort / _shuffle)
      lap("emission lists");
      if (carried) {  // the lists hold the ranks now
        WG_HIP(hipStreamSynchronize(e.stream));
        st.ranks = nullptr;
        TmpMatrix::release(tmpRanks);
      }
    } else if (peers16) {  // (narrowed on the host, a slice at a time)
      const size_t total = nLoc * (size_t)(N - 1), step = (size_t)1 << 26;
      std::vector<uint16_t> tmp(std::min(total, step));
      const int32_t* src = init.peers + (size_t)lo * (N - 1);
      for (size_t at = 0; at < total; at += step) {
        const size_t k = std::min(step, total - at);
        for (size_t i = 0; i < k; i++) tmp[i] = (uint16_t)src[at + i];
        WG_HIP(hipMemcpy(dPeers16 + (size_t)lo * (N - 1) + at, tmp.data(), 2 * k, hipMemcpyHostToDevice));
      }
    } else {
      WG_HIP(hipMemcpy(dPeers32 + (size_t)lo * (N - 1), init.peers + (size_t)lo * (N - 1), 4 * nLoc * (N - 1),
                       hipMemcpyHostToDevice));
    }
    st.snapIdx = nullptr;
    st.nSnap = nullptr;
    st.xsnap = nullptr;
    st.xsnapRows = 0;
    st.xcand = e.shardCount > 0 ? e.dalloc<int32_t>((size_t)N / 4 + 1) : nullptr;
    {  // the all-ones block: the payload of a send whose totalOutgoing is complete (dissemination, fast path of a shard)
      std::vector<uint64_t> ones((size_t)W, ~0ULL);
      uint64_t* dOnes = e.dalloc<uint64_t>(W, false, Engine::AC_CONST);
      WG_HIP(hipMemcpy(dOnes, ones.data(), 8 * (size_t)W, hipMemcpyHostToDevice));
      st.ones = dOnes;
    }
    st.xS = st.xMe = 0;
    st.xChunkCap = 0;
    st.xout = st.xin = nullptr;
    st.xoutCount = nullptr;
    st.xcounts = nullptr;
    directed = e.shardCount > 0 && e.has_alltoall();  // (no all-to-all callback: the all-reduce image of rounds 1-4)
    if (directed) {  // the snapshots' sub-rows go to the shard that reads them (HandelState::xout)
      st.xS = e.shardCount;
      st.xMe = e.shardIndex;
      // a node's sub-rows for one destination are nested blocks of its copy: at most twice the copy's width, and a chunk
      // per level for the narrow ones; all of a shard's nodes may disseminate in one ms
      st.xChunkCap = (uint32_t)std::min<uint64_t>(0x7FFFFFFFull / (H_XCHUNK * 2), (uint64_t)nLoc * (st.snapStride / (H_XWORDS / 2) + (uint32_t)L));
      if (st.xS > 1) {
        st.xout = e.dalloc<uint64_t>((size_t)st.xS * st.xChunkCap * H_XCHUNK, false, Engine::AC_SCRATCH);
        st.xin = e.dalloc<uint64_t>((size_t)st.xS * st.xChunkCap * H_XCHUNK, false, Engine::AC_SCRATCH);
      }
      st.xoutCount = e.dalloc<uint32_t>((size_t)st.xS, true, Engine::AC_SCRATCH);
      st.xcounts = e.dalloc<int32_t>((size_t)st.xS * st.xS, true, Engine::AC_SCRATCH);
    } else if (e.shardCount > 0) {
      st.snapIdx = e.dalloc<uint32_t>(e.dev.maxEvents, false);
      st.nSnap = e.dalloc<uint32_t>(1);
      // every node disseminates once per period; a desynchronised start spreads them, a synchronised one puts all
      // of them into the same ms
      st.xsnapRows = (uint32_t)N;
      st.xsnap = e.dalloc<int32_t>((size_t)st.xsnapRows * st.snapStride * 2, false);
    }
    int32_t *dStart = nullptr, *dPair = nullptr;
    WG_HIP(hipMalloc((void**)&dStart, 4 * (size_t)N));
    WG_HIP(hipMalloc((void**)&dPair, 4 * (size_t)N));
    WG_HIP(hipMemcpy(dStart, init.startAt, 4 * (size_t)N, hipMemcpyHostToDevice));
    WG_HIP(hipMemcpy(dPair, init.nodePairingTime, 4 * (size_t)N, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_handel_init, dim3(((int)nLoc + 255) / 256), dim3(256), 0, e.stream, st, e.dev.nodes.down, dStart, dPair);
    WG_HIP(hipStreamSynchronize(e.stream));
    lap("node state");
    (void)hipFree(dStart);
    (void)hipFree(dPair);
  }
  // The reception ranks on the device (P/Handel.java:966-989; see k_handel_init_scan). Leaves rd after the last shuffle.
  // (tests: WG_INIT_BIG=1 the > 65 536-node forms at any size; =2 also histogram bins of four ranks, as 131 072 nodes have)
  static int init_big() { return getenv("WG_INIT_BIG") ? atoi(getenv("WG_INIT_BIG")) : 0; }
  template <int E>
  void launch_chain(Engine& e, const HandelState& st, int threads, size_t lds, int B, uint16_t* net, uint16_t* starts) {
#if !defined(WG_EMU)
    if (lds > 48 * 1024) {
      WG_HIP(hipFuncSetAttribute((const void*)k_handel_init_chain<E, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      WG_HIP(hipFuncSetAttribute((const void*)k_handel_init_chain<E, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      WG_HIP(hipFuncSetAttribute((const void*)k_handel_init_chain_starts<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
#endif
    const int C = st.N / B;
    hipLaunchKernelGGL((k_handel_init_chain<E, false>), dim3(C), dim3(threads), lds, e.stream, st, B, net, (const uint16_t*)starts);
    hipLaunchKernelGGL(k_handel_init_chain_starts<E>, dim3(1), dim3(threads), lds, e.stream, st.N, C, (const uint16_t*)net, starts);
    hipLaunchKernelGGL((k_handel_init_chain<E, true>), dim3(C), dim3(threads), lds, e.stream, st, B, net, (const uint16_t*)starts);
  }
  bool directed = false;    // sharded: dissemination snapshots go to the shard that reads them (all-to-all), not to every shard
  bool carried = false;     // the reception ranks travel with the messages (HandelState::peersR / bump): no matrix
  uint64_t rngAtRanks = 0;  // rd where setReceivingRanks starts (device-built ranks): the read-back of a CARRIED run re-shuffles from it
  // (hs: the state whose `ranks` the rows go to — the engine's, or a read-back's scratch copy; advance: leave rd after the shuffles)
  void build_ranks(Engine& e, const HandelState& hs, bool advance) {
    const HandelState& st = hs;
    const uint64_t rng0 = advance ? e.gh.rng : rngAtRanks;
    const int32_t N = st.N;
    if (e.shardCount > 0 || N > 131072 || N < 256)
      throw WgError(WG_EINVAL, "device-built reception ranks: an unsharded engine of 256 .. 131 072 nodes (pass wg_handel_init_state.receptionRanks)");
    const bool big = N > 65536 || init_big();  // (ids beyond 16 bits: the lists in global memory)
    const unsigned long long total0 = (unsigned long long)N * (N - 1);
    const unsigned long long expectRej = ((unsigned long long)N * N * N) >> 33, expectCand = ((unsigned long long)N * N * N) >> 31;
    const unsigned long long slack = 4 * expectRej + 65536;
    const uint32_t cap = (uint32_t)(2 * expectCand + 4096);
    unsigned long long *dCand = nullptr, *dOffs = nullptr;
    uint32_t* dFlags = nullptr;  // [0] candidates, [1] mismatch
    WG_HIP(hipMalloc((void**)&dCand, 8 * (size_t)cap));
    WG_HIP(hipMalloc((void**)&dOffs, 8 * ((size_t)N + 1)));
    WG_HIP(hipMalloc((void**)&dFlags, 8));
    struct Free {
      void *a, *b, *c;
      ~Free() {
        (void)hipFree(a);
        (void)hipFree(b);
        (void)hipFree(c);
      }
    } guard{dCand, dOffs, dFlags};
    WG_HIP(hipMemsetAsync(dFlags, 0, 8, e.stream));
    hipLaunchKernelGGL(k_handel_init_scan, dim3(2048 / WG_GRID_DIV), dim3(256), 0, e.stream, rng0, total0 + slack, (uint32_t)N, dCand,
                       dFlags, cap);
    uint32_t nCand = 0;
    WG_HIP(hipMemcpyAsync(&nCand, dFlags, 4, hipMemcpyDeviceToHost, e.stream));
    WG_HIP(hipStreamSynchronize(e.stream));
    if (nCand > cap) throw WgError(WG_EHOSTINIT, "reception ranks: more candidate draws than the list holds");
    std::vector<unsigned long long> cand(nCand);
    if (nCand) WG_HIP(hipMemcpy(cand.data(), dCand, 8 * (size_t)nCand, hipMemcpyDeviceToHost));
    std::sort(cand.begin(), cand.end());  // (position in the high bits)
    std::vector<unsigned long long> offs((size_t)N + 1, 0);  // [n + 1] first: rejected draws of node n
    unsigned long long rej = 0;
    for (unsigned long long c : cand) {
      const unsigned long long q = (c >> 20) - rej;  // the draw this stream position belongs to
      if (q >= total0) break;
      const int32_t bound = N - (int32_t)(q % (unsigned long long)(N - 1));
      const int32_t m = bound - 1, u = (int32_t)(0x7FFFFFFFu - (uint32_t)(c & 0xFFFFFULL));
      if ((bound & m) == 0) continue;  // (a power of two: no loop, Random.nextInt)
      if ((int32_t)((uint32_t)u - (uint32_t)(u % bound) + (uint32_t)m) < 0) {
        rej++;
        offs[(size_t)(q / (unsigned long long)(N - 1)) + 1]++;
      }
    }
    if (getenv("WG_INIT_VERBOSE") && atoi(getenv("WG_INIT_VERBOSE")))
      fprintf(stderr, "[wittgpu] reception ranks on the device: %u candidate draws, %llu rejected\n", nCand, rej);
    if (rej > slack) throw WgError(WG_EHOSTINIT, "reception ranks: more rejected draws than the scanned part of rd's stream covers");
    for (int n = 0; n < N; n++) offs[(size_t)n + 1] += offs[n] + (unsigned long long)(N - 1);
    WG_HIP(hipMemcpyAsync(dOffs, offs.data(), 8 * offs.size(), hipMemcpyHostToDevice, e.stream));
    hipLaunchKernelGGL(k_handel_init_perm, dim3((N + 63) / 64), dim3(64), 0, e.stream, st, dOffs, rng0, dFlags + 1);
    const int threads = std::min(N, 1024);
    const size_t lds = 2 * (size_t)N;
    const int B = std::max(8, N / 256);  // nodes per chunk: 256 chunks from 2 048 nodes on
    uint16_t* dLists = nullptr;          // [2][N / B][N]: every chunk's net move, then the list it starts from
    if (big) {                           // (32-bit ids; [2][N / B][N] more: the two buffers of every chunk's list)
      const size_t C = (size_t)(N / B);
      WG_HIP(hipMalloc((void**)&dLists, 4 * 4 * C * N));
      uint32_t *net32 = (uint32_t*)dLists, *starts32 = net32 + C * N, *work = starts32 + C * N;
      hipLaunchKernelGGL((k_handel_init_chain_big<false>), dim3((int)C), dim3(threads), 0, e.stream, st, B, net32, (const uint32_t*)starts32, work);
      hipLaunchKernelGGL(k_handel_init_chain_starts_big, dim3(1), dim3(threads), 0, e.stream, N, (int)C, (const uint32_t*)net32, starts32, work);
      hipLaunchKernelGGL((k_handel_init_chain_big<true>), dim3((int)C), dim3(threads), 0, e.stream, st, B, net32, (const uint32_t*)starts32, work);
    } else {
    WG_HIP(hipMalloc((void**)&dLists, 2 * 2 * (size_t)(N / B) * N));
    uint16_t *net = dLists, *starts = dLists + (size_t)(N / B) * N;
    switch (N / threads) {
      case 1: launch_chain<1>(e, st, threads, lds, B, net, starts); break;
      case 2: launch_chain<2>(e, st, threads, lds, B, net, starts); break;
      case 4: launch_chain<4>(e, st, threads, lds, B, net, starts); break;
      case 8: launch_chain<8>(e, st, threads, lds, B, net, starts); break;
      case 16: launch_chain<16>(e, st, threads, lds, B, net, starts); break;
      case 32: launch_chain<32>(e, st, threads, lds, B, net, starts); break;
      default: launch_chain<64>(e, st, threads, lds, B, net, starts); break;
    }
    }
    uint32_t bad = 0;
    const hipError_t rcBad = hipMemcpyAsync(&bad, dFlags + 1, 4, hipMemcpyDeviceToHost, e.stream);
    const hipError_t rcSync = hipStreamSynchronize(e.stream);
    (void)hipFree(dLists);
    WG_HIP(rcBad);
    WG_HIP(rcSync);
    if (bad) throw WgError(WG_EHOSTINIT, "reception ranks: a node drew another number of times than the candidate walk gave it");
    if (advance) {
      e.gh.rng = lcg_skip(e.gh.rng, total0 + rej);
      e.globalsDirty = true;
    }
  }
  // The emission lists of every live sender on the device (P/Handel.java:991-1013): the sort per (sender, level), the draw
  // counts summed on the host in the reference's order, the equal-rank shuffles from jumped rd states. Leaves the engine's
  // rd advanced by the draws, as init() would.
  void build_peers(Engine& e) {
    const int32_t N = st.N;
    if (e.shardCount > 0 || N > 131072)
      throw WgError(WG_EINVAL, "device-built emission lists: an unsharded engine of at most 131 072 nodes (pass wg_handel_init_state.peers)");
    const size_t NL = (size_t)N * st.L;
    uint32_t* dCnt = nullptr;
    unsigned long long* dOffs = nullptr;
    uint32_t* dRej = nullptr;
    WG_HIP(hipMalloc((void**)&dCnt, 4 * NL));
    WG_HIP(hipMalloc((void**)&dOffs, 8 * NL));
    WG_HIP(hipMalloc((void**)&dRej, 4));
    struct Free {
      void *a, *b, *c;
      ~Free() {
        (void)hipFree(a);
        (void)hipFree(b);
        (void)hipFree(c);
      }
    } guard{dCnt, dOffs, dRej};
    WG_HIP(hipMemsetAsync(dRej, 0, 4, e.stream));
    // levels sorted by counting (k_handel_init_sort): the last one of more than 65 536 nodes — rank (17 bits) and offset (16)
    // no longer pack into 32 bits and its block (256 KB of keys) no longer fits LDS; WG_INIT_BIG=1: the last level always
    const int bigFrom = N > 65536 || init_big() ? st.L - 1 : st.L;
    int shift = 0;
    while ((N >> shift) > 32768) shift++;  // histogram bins: rank >> shift, at most 32 768 of them (128 KB)
    if (init_big() >= 2) shift += 2;
    const size_t ldsWords = bigFrom < st.L ? std::max((size_t)(N >> shift), (size_t)std::max(2, N / 4)) : (size_t)std::max(2, N / 2);
    const size_t lds = sizeof(uint32_t) * ldsWords;
#if !defined(WG_EMU)
    if (lds > 48 * 1024)
      WG_HIP(hipFuncSetAttribute((const void*)k_handel_init_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
    const int grid = std::max(1, std::min(N, 2048 / WG_GRID_DIV));
    unsigned long long* dPairs = nullptr;
    if (bigFrom < st.L) WG_HIP(hipMalloc((void**)&dPairs, 8 * (size_t)grid * (size_t)(N / 2)));
    struct FreePairs {
      void* p;
      ~FreePairs() { (void)hipFree(p); }
    } guardPairs{dPairs};
    hipLaunchKernelGGL(k_handel_init_sort, dim3(grid), dim3(1024), lds, e.stream, st, e.dev.nodes.down, dCnt, bigFrom, shift, dPairs);
    std::vector<uint32_t> cnt(NL);
    WG_HIP(hipMemcpyAsync(cnt.data(), dCnt, 4 * NL, hipMemcpyDeviceToHost, e.stream));
    WG_HIP(hipStreamSynchronize(e.stream));
    std::vector<unsigned long long> offs(NL);
    unsigned long long draws = 0;
    for (size_t i = 0; i < NL; i++) {  // (sender ascending, level ascending: the order init() walks them in)
      offs[i] = draws;
      draws += cnt[i];
    }
    WG_HIP(hipMemcpyAsync(dOffs, offs.data(), 8 * NL, hipMemcpyHostToDevice, e.stream));
    hipLaunchKernelGGL(k_handel_init_shuffle, dim3(grid), dim3(256), 0, e.stream, st, e.dev.nodes.down, dOffs, e.gh.rng, dRej);
    uint32_t rej = 0;
    WG_HIP(hipMemcpyAsync(&rej, dRej, 4, hipMemcpyDeviceToHost, e.stream));
    WG_HIP(hipStreamSynchronize(e.stream));
    if (rej || (getenv("WG_FORCE_INIT_REJECT") && atoi(getenv("WG_FORCE_INIT_REJECT"))))
      throw WgError(WG_EHOSTINIT, "a nextInt(bound) draw of an emission-list shuffle was rejected: the lists need the host's sequential rd");
    e.gh.rng = lcg_skip(e.gh.rng, draws);
    e.globalsDirty = true;
  }
  bool has_cond() const override { return true; }
  int levels() const override { return st.L; }
  void on_restore(Engine& e) override {
    if (carried) {  // the ranks as init() left them = no bumps: the tables and the BUMP bits (with the other delivery-side bits) zeroed
      reset_rows(e);
      WG_HIP(hipMemsetAsync(st.bump + (size_t)st.lo * st.bumpCap, 0, 4 * (size_t)(st.hi - st.lo) * st.bumpCap, e.stream));
      return;
    }
    if (e.gh.notes & NOTE_RANKS_SATURATED)
      throw WgError(WG_EUNSUPPORTED, "wg_restore: a receptionRanks entry saturated at Integer.MAX_VALUE in the last run; "
                                     "the initial ranks cannot be recomputed in place — re-run init()");
    reset_rows(e);
    if (st.N < 4) {  // (rows shorter than a vector: the few words by plain copy of what init() uploaded is not kept; mask on the host)
      std::vector<int32_t> r((size_t)(st.hi - st.lo) * st.N);
      WG_HIP(hipMemcpy(r.data(), st.ranks + (size_t)st.lo * st.N, 4 * r.size(), hipMemcpyDeviceToHost));
      for (auto& v : r) v &= st.N - 1;
      WG_HIP(hipMemcpy(st.ranks + (size_t)st.lo * st.N, r.data(), 4 * r.size(), hipMemcpyHostToDevice));
      return;
    }
    hipLaunchKernelGGL(k_handel_ranks_reset, dim3(2048 / WG_GRID_DIV), dim3(256), 0, e.stream, st);
  }
  void reset_rows(Engine& e) {
    const size_t nLoc = (size_t)(st.hi - st.lo), at = (size_t)st.lo * st.W;
    WG_HIP(hipMemsetAsync(st.rows + at * HK_COUNT, 0, 8 * nLoc * st.W * HK_COUNT, e.stream));
    WG_HIP(hipMemsetAsync(st.drows + at * HD_COUNT, 0, 8 * nLoc * st.W * HD_COUNT, e.stream));
    if (st.atk == 1) WG_HIP(hipMemsetAsync(st.blacklist, 0, 8 * (size_t)st.N * st.W, e.stream));
    hipLaunchKernelGGL(k_handel_own_bits, dim3(((int)nLoc + 255) / 256), dim3(256), 0, e.stream, st);
  }
  bool cont_if(Engine& e, int32_t* out) override {
    if (!dCont) dCont = e.dalloc<uint32_t>(1);
    Group g = e.self();
    WG_HIP(hipMemsetAsync(dCont, 0, 4, e.stream));
    launch_cont_if(g, dCont);
    uint32_t v = 0;
    WG_HIP(hipMemcpyAsync(&v, dCont, 4, hipMemcpyDeviceToHost, e.stream));
    WG_HIP(hipStreamSynchronize(e.stream));
    *out = (int32_t)v;
    return true;
  }
  int host_msg_size(uint32_t msg) const override {
    int l = (int)(msg & 31u);
    return 1 + ((l == 0 ? 1 : (1 << (l - 1))) / 8) + 192;
  }
  int payload_bytes_of_level(int l) const override {
    return l == 0 ? 0 : ((1 << (l - 1)) >= 64 ? (1 << (l - 1)) / 8 : 8);  // 64-bit words of the level's block
  }
  size_t state_size() const override { return sizeof(st); }
  const void* state_host() const override { return &st; }
  int variant() const override { return st.atk; }
  void launch_a1(const Group& g, const HandelState* stab, int R, hipStream_t s) {
    if (st.atk) {  // an attack's run: the instantiation with the attack's paths, every item a wavefront
      hipLaunchKernelGGL((k_handel_a1w<4, true>), dim3(a1_grid(R), R), dim3(256), 0, s, g.tab, stab);
      return;
    }
    // both kinds of items in one launch: the narrow levels' by groups of eight lanes, the wide levels' one wavefront each
    // (as two launches: 44 + 33 us per ordinary ms against ~ 70 us, profiles/r13h_*)
    hipLaunchKernelGGL((k_handel_a1c<5>), dim3(a1_grid(R), R), dim3(256), 0, s, g.tab, stab);
  }
  void launch_cond(Engine& profOwner, const Group& g) override {
    const HandelState* stab = (const HandelState*)g.stab;
    {
      Engine::ProfScope ps(profOwner, Engine::PC_COND_SELECT);
      hipLaunchKernelGGL(k_handel_cond_pre, dim3((st.N + 255) / 256, g.R), dim3(256), 0, g.stream, g.tab, stab);
      launch_a1(g, stab, g.R, g.stream);
    }
    Engine::ProfScope ps(profOwner, Engine::PC_COND_REST);
    Engine::scan<CondF>(g, stab);
    if (st.atk == 2)  // HiddenByzantine.attack on the drawn candidates of the last level
      hipLaunchKernelGGL(k_handel_hidden, dim3(node_grid(g.R), g.R), dim3(256), 0, g.stream, g.tab, stab);
    if (st.atk)
      hipLaunchKernelGGL((k_handel_cond_a2<false, true>), dim3(grid_per_engine(GRID_COND_TAIL, g.R, 1024, st.N, 1024), g.R), dim3(256), 0, g.stream, g.tab, stab);
    else
    {
      // one thread per drawing node — about a quarter of the nodes at pairingTime 4 —, ONE round of them: a grid that makes a
      // thread take a second node doubles the kernel's chain (22 -> 29 us at 32 copies with the batch-wide total alone)
      const int floorBlocks = std::min(GRID_COND_TAIL, ((st.N / 3 + 255) / 256 + 7) / 8 * 8);
      const int gx = std::max(grid_per_engine(GRID_COND_TAIL, g.R, 1024, st.N, 1024), floorBlocks);
      hipLaunchKernelGGL((k_handel_cond_a2<false, false>), dim3(gx, g.R), dim3(256), 0, g.stream, g.tab, stab);
    }
  }
  // ---- node-range sharding (Engine::run_ms_sharded) ----
  bool supports_shards() const override { return true; }
  // the dissemination snapshots written in this ms -> every shard's copy of the snapshot ring
  bool shard_snap_is_scan() const override { return true; }
  bool shard_snap_directed() const override { return directed; }
  uint32_t* shard_snap_enqueue(const Group& g) override {
    Engine::scan<HandelSnapF>(g, (const HandelState*)g.stab);
    return st.nSnap;
  }
  // the owner-directed form: count matrix (a tiny all-reduce: every shard learns what it will receive), the regions by
  // all-to-all, the received chunks into this shard's ring
  void shard_snap_directed_exchange(Engine& e, const Group& g) {
    const int S = st.xS;
    if (S <= 1) return;  // (one shard: no message leaves it)
    hipLaunchKernelGGL(k_handel_xcounts, dim3(1), dim3(64), 0, g.stream, st);
    e.shard_allreduce(st.xcounts, (int64_t)S * S, Engine::XK_COUNTS);
    std::vector<int32_t> cm((size_t)S * S);
    WG_HIP(hipMemcpyAsync(cm.data(), st.xcounts, 4 * cm.size(), hipMemcpyDeviceToHost, g.stream));
    WG_HIP(hipStreamSynchronize(g.stream));
    const int64_t words = 2 * H_XCHUNK;  // int32 words a chunk
    std::vector<int64_t> sc(S), so(S), rc(S), ro(S);
    int64_t inChunks = 0;
    for (int p = 0; p < S; p++) {
      sc[p] = p == st.xMe ? 0 : (int64_t)cm[(size_t)st.xMe * S + p] * words;
      so[p] = (int64_t)p * st.xChunkCap * words;
      rc[p] = p == st.xMe ? 0 : (int64_t)cm[(size_t)p * S + st.xMe] * words;
      ro[p] = inChunks * words;
      inChunks += p == st.xMe ? 0 : cm[(size_t)p * S + st.xMe];
    }
    if (inChunks > (int64_t)S * st.xChunkCap) throw WgError(WG_ENOMEM, "sharded Handel: more snapshot chunks arrive than the exchange buffer holds");
    e.shard_alltoallv(st.xout, sc.data(), so.data(), st.xin, rc.data(), ro.data());
    if (inChunks)
      hipLaunchKernelGGL(k_handel_xunpack, dim3((unsigned)std::min<int64_t>(1024, (inChunks * 16 + 255) / 256)), dim3(256), 0, g.stream, st, (uint32_t)inChunks);
    WG_HIP(hipMemsetAsync(st.xoutCount, 0, 4 * (size_t)S, g.stream));
    WG_HIP(hipMemsetAsync(st.xcounts, 0, 4 * (size_t)S * S, g.stream));
  }
  void shard_snap_exchange(Engine& e, const Group& g, uint32_t nSnap) override {
    if (directed) return shard_snap_directed_exchange(e, g);
    const HandelState* stab = (const HandelState*)g.stab;
    hipLaunchKernelGGL((k_shard_snap<HandelState, H_TASK_DISSEMINATION, true>), dim3(GRID_DELIVER_SMALL, 1), dim3(256), 0, g.stream, g.tab, stab);
    e.shard_allreduce(st.xsnap, (int64_t)nSnap * 2, Engine::XK_SNAPSHOTS);  // (nSnap: 64-bit words of the packed rows)
    hipLaunchKernelGGL((k_shard_snap<HandelState, H_TASK_DISSEMINATION, false>), dim3(GRID_DELIVER_SMALL, 1), dim3(256), 0, g.stream, g.tab, stab);
  }
  // checkSigs' edge (launch_cond above) with the draw order made global: the per-node candidate counts are summed
  // across shards (a shard knows its own nodes'), after which the ordinal of every drawing node, its rd draw and the
  // number of draws are computed identically everywhere; the task registrations go through the exchange image.
  uint32_t shard_cond(Engine& e, const Group& g) override {
    const HandelState* stab = (const HandelState*)g.stab;
    hipLaunchKernelGGL(k_handel_cond_pre, dim3((st.hi - st.lo + 255) / 256, 1), dim3(256), 0, g.stream, g.tab, stab);
    launch_a1(g, stab, 1, g.stream);
    // exchange 5: how many levels of every node have a candidate — a byte per node, its owner's (zeros elsewhere)
    hipLaunchKernelGGL(k_handel_cand_pack, dim3(GRID_SHARD_SMALL, 1), dim3(256), 0, g.stream, g.tab, stab);
    e.shard_allreduce(st.xcand, ((int64_t)st.N + 3) / 4, Engine::XK_CANDIDATES);
    Engine::scan<CondF>(g, stab);
    uint32_t nOut = 0;
    const uint32_t seq = e.publish_counts((const uint32_t*)((const char*)e.dev.g.raw + offsetof(Globals, nOut)), nullptr);
    hipLaunchKernelGGL((k_handel_cond_a2<true, false>), dim3(GRID_COND_TAIL, 1), dim3(256), 0, g.stream, g.tab, stab);
    e.wait_counts(seq, &nOut, nullptr);  // (read while the tail runs)
    return nOut;
  }
  // blocks per engine of the wave-per-item kernels. Their pipelined loops want SEVERAL items per wavefront (the next
  // item's header is fetched during the current one), so the grid is about twice the chip's resident waves over the
  // whole batch (the factor lets the blocks of members whose run has ended — they return at once — leave their share to the
  // others)
  int node_grid(int R) const {
    int b = (2 * 1024 / WG_GRID_DIV) / (R > 0 ? R : 1);
    return b < 16 ? 16 : b;
  }
  // k_handel_a1c wants more blocks than the delivery kernels (its group blocks hold 32 items each and an ordinary ms has
  // ~ 10 k of them per engine): 6 x the delivery kernels' blocks, half of them on the groups (profiles/r13q_sweep_a1_grid.txt)
  int a1_grid(int R) const { return 6 * node_grid(R); }
  // the delivery pass: k_handel_lane (one lane per node: SendSigs deliveries, narrow updateVerifiedSignatures; sorts the
  // other nodes into the next kernel's list), k_handel_copy (the wide payloads it delivered, one wavefront each), then
  // k_handel_wave (one wavefront per listed node / deferred fast path)
  void launch_deliver(const Group& g) override {
    const HandelState* stab = (const HandelState*)g.stab;
    // (workgroups of ONE wavefront: at three wavefronts a SIMD a 256-thread workgroup needs four free slots at once, one per SIMD,
    // and the kernel's wavefronts end at very different times — 576 -> 579 M msgs/s same-box, and the 32-copy cliff halves:
    // profiles/r23d_*. The kernels with more resident wavefronts lose with small workgroups.)
    hipLaunchKernelGGL(k_handel_lane, dim3(4 * grid_per_engine(GRID_LANE_NODES, g.R, 2048, st.N, 512), g.R), dim3(64), 0, g.stream, g.tab, stab);
    hipLaunchKernelGGL(k_handel_update<8>, dim3(node_grid(g.R), g.R), dim3(256), 0, g.stream, g.tab, stab);
    // the deliveries behind a wide update that was its node's first event, one lane per node (after the update)
    if (!st.atk)
      hipLaunchKernelGGL(k_handel_lane2, dim3(grid_per_engine(GRID_COND_TAIL, g.R, 512, st.N, 2048), g.R), dim3(256), 0, g.stream, g.tab, stab, 0);
    const dim3 grid(node_grid(g.R), g.R);
    // In a ms whose phase no member's dissemination task has (19 of 20 with a synchronised start) the lean dissemination kernel
    // would find an empty list: not launched (6 us each at 24 copies). k_handel_wave is told, and stops the run loudly should
    // the list not be empty after all.
    const bool mayDissem = g.periodic_may_fire(H_TASK_DISSEMINATION);
    const int disSkipped = !st.atk && !mayDissem;
    if (!st.atk && mayDissem) {  // nodes whose first event is their dissemination: that event
      hipLaunchKernelGGL(k_handel_dissem<8>, grid, dim3(256), 0, g.stream, g.tab, stab);
      // ... and the plain deliveries behind it, one lane per node (the others' remaining events are visits of k_handel_wave)
      hipLaunchKernelGGL(k_handel_lane2, dim3(grid_per_engine(GRID_COND_TAIL, g.R, 1024, st.N, 1024), g.R), dim3(256), 0, g.stream, g.tab, stab, 1);
    }
    // every wide payload the lane kernels delivered, one wavefront per copy (behind the dissemination: it reads no queue slot)
    hipLaunchKernelGGL(k_handel_copy, dim3(node_grid(g.R), g.R), dim3(256), 0, g.stream, g.tab, stab);
    if (st.atk) {
      hipLaunchKernelGGL((k_handel_wave<4, true>), grid, dim3(256), 0, g.stream, g.tab, stab, 0);
      return;
    }
    hipLaunchKernelGGL((k_handel_wave<4, false>), grid, dim3(256), 0, g.stream, g.tab, stab, disSkipped);
  }
  bool launch_cont_if(const Group& g, uint32_t* dOut) override {
    hipLaunchKernelGGL(k_handel_cont_if, dim3(std::max(1, std::min(8, (st.N + 255) / 256)), g.R), dim3(256), 0, g.stream, g.tab,
                       (const HandelState*)g.stab, dOut);
    return true;
  }
  // read-back (tests, statistics): the header records come back whole and are picked apart on the host
  std::vector<uint32_t> read_hdr() {
    std::vector<uint32_t> h((size_t)st.N * st.hdrStride);  // (sharded: zeros for the nodes of other shards)
    const size_t at = (size_t)st.lo * st.hdrStride;
    WG_HIP(hipMemcpy(h.data() + at, st.hdr + at, 4 * (size_t)(st.hi - st.lo) * st.hdrStride, hipMemcpyDeviceToHost));
    return h;
  }
  bool read_i64(Engine&, int32_t field, int64_t* dst, int32_t n) override {
    int off;
    switch (field) {
      case WG_F_SIGS_CHECKED: off = HH_SIGCHK; break;
      case WG_F_SIG_QUEUE_SIZE: off = HH_SIGQ; break;
      case WG_F_MSG_FILTERED: off = HH_FILT; break;
      case WG_F_CURR_WINDOW_SIZE: off = HH_WINDOW; break;
      case WG_F_ADDED_CYCLE: off = HH_ADDED; break;
      case WG_F_START_AT: off = HH_START; break;
      case WG_F_NODE_PAIRING_TIME: off = HH_PAIR; break;
      default: return false;
    }
    const std::vector<uint32_t> h = read_hdr();
    for (int i = 0; i < n; i++) dst[i] = (int32_t)h[(size_t)i * st.hdrStride + off];
    return true;
  }
  bool read_level_i32(Engine& e, int32_t field, int32_t* dst, int32_t n, int32_t L) override {
    if (field == WG_LF_RECEPTION_RANKS) {  // [node][sender], this shard's rows (zeros elsewhere)
      if (n != st.N || L != st.N) throw WgError(WG_EINVAL, "shape must be [nodeCount][nodeCount]");
      memset(dst, 0, 4 * (size_t)n * n);
      WG_HIP(hipStreamSynchronize(e.stream));
      if (carried) {
        // No matrix is kept: the initial ranks are shuffled again from the rd state init() had there (they are a function of
        // it alone), then every node's bumps are added — nodeCount each, saturating (:825-828). Needs 4 N^2 bytes free.
        int32_t* tmp = nullptr;
        if (hipMalloc((void**)&tmp, 4 * (size_t)n * n) != hipSuccess) {
          (void)hipGetLastError();
          throw WgError(WG_ENOMEM, "read-back of receptionRanks: no room for the 4 * nodeCount^2 bytes the matrix is rebuilt in");
        }
        struct Free {
          void* p;
          ~Free() { (void)hipFree(p); }
        } guard{tmp};
        HandelState hs = st;
        hs.ranks = tmp;
        build_ranks(e, hs, false);
        hipLaunchKernelGGL(k_handel_ranks_add_bumps, dim3(std::max(1, std::min(n, 2048 / WG_GRID_DIV))), dim3(256), 0, e.stream, hs);
        WG_HIP(hipStreamSynchronize(e.stream));
        WG_HIP(hipMemcpy(dst, tmp, 4 * (size_t)n * n, hipMemcpyDeviceToHost));
        return true;
      }
      WG_HIP(hipMemcpy(dst + (size_t)st.lo * n, st.ranks + (size_t)st.lo * n, 4 * (size_t)(st.hi - st.lo) * n, hipMemcpyDeviceToHost));
      return true;
    }
    if (n != st.N || L != st.L) throw WgError(WG_EINVAL, "shape must be [nodeCount][levels]");
    if (field == WG_LF_QUEUE_LEN) {  // toVerifyAgg.size(): the head word of every queue record
      memset(dst, 0, 4 * (size_t)n * L);
      const size_t cnt = (size_t)(st.hi - st.lo) * L;
      int32_t* tmp = nullptr;
      WG_HIP(hipMalloc((void**)&tmp, 4 * cnt));
      hipLaunchKernelGGL(k_handel_gather_qlen, dim3(256 / WG_GRID_DIV > 0 ? 256 / WG_GRID_DIV : 1), dim3(256), 0, e.stream, st, tmp);
      const hipError_t rc = hipMemcpy(dst + (size_t)st.lo * L, tmp, 4 * cnt, hipMemcpyDeviceToHost);
      (void)hipFree(tmp);
      WG_HIP(rc);
      return true;
    }
    const int plane = field == WG_LF_POS_IN_LEVEL ? HP_POS : field == WG_LF_OUTGOING_FINISHED ? HP_OUTFIN
                      : field == WG_LF_SUICIDE_BIZ_AFTER ? HP_SPARE0 : -1;
    if (plane < 0) return false;
    if (plane == HP_SPARE0 && st.atk != 1) {  // HLevel.suicideBizAfter is -1 without byzantineSuicide (:406); an honest run keeps
      for (size_t i = 0; i < (size_t)n * L; i++) dst[i] = -1;  // the level's candidate summary in that plane (h_summary_has_candidate)
      return true;
    }
    const std::vector<uint32_t> h = read_hdr();
    for (int i = 0; i < n; i++)
      for (int l = 0; l < L; l++) dst[(size_t)i * L + l] = (int32_t)h[(size_t)i * st.hdrStride + HH_LV + l * HP_COUNT + plane];
    return true;
  }
  bool read_bits(Engine&, int32_t field, uint64_t* dst, int32_t n, int32_t w) override {
    if (n != st.N || w != st.W) throw WgError(WG_EINVAL, "shape must be [nodeCount][max(1, nodeCount/64)]");
    if (field == WG_B_BLACKLIST) {  // (id order as it is; all zeros without the attack)
      memset(dst, 0, 8 * (size_t)n * w);
      if (st.atk == 1) {
        WG_HIP(hipStreamSynchronize(eng.stream));
        WG_HIP(hipMemcpy(dst, st.blacklist, 8 * (size_t)n * w, hipMemcpyDeviceToHost));
      }
      return true;
    }
    int k;  // (>= HK_COUNT: a delivery-side set — toVerifyInd is computed, SEEN & ~verifiedInd: proto_handel.hip.h, file header)
    switch (field) {
      case WG_B_TOTAL_INCOMING: k = HK_TI; break;
      case WG_B_LAST_AGG_VERIFIED: k = HK_LA; break;
      case WG_B_VERIFIED_IND: k = HK_VI; break;
      case WG_B_TO_VERIFY_IND: k = HK_COUNT + HD_SEEN; break;
      case WG_B_FINISHED_PEERS: k = HK_COUNT + HD_FP; break;
      default: return false;
    }
    memset(dst, 0, 8 * (size_t)n * w);
    const size_t words = (size_t)(st.hi - st.lo) * w;
    uint64_t* tmp = nullptr;  // the device keeps the five kinds level-major (h_row): gathered into id order here
    WG_HIP(hipMalloc((void**)&tmp, 8 * words));
    hipLaunchKernelGGL(k_handel_gather_row, dim3(1024 / WG_GRID_DIV), dim3(256), 0, eng.stream, st, k, tmp);
    const hipError_t rc = hipMemcpy(dst + (size_t)st.lo * w, tmp, 8 * words, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    WG_HIP(rc);
    return true;
  }
  // Node.msgReceived / msgSent / bytesReceived / bytesSent: what the engine's own arrays hold (host-side sends) plus what
  // this protocol's visits counted in the header records
  bool node_counter(Engine& e, int32_t field, int64_t* dst, int32_t n) override {
    int off;
    bool wide;
    switch (field) {
      case WG_F_MSG_RECEIVED: off = HH_NRECV, wide = false; break;
      case WG_F_MSG_SENT: off = HH_NSENT, wide = false; break;
      case WG_F_BYTES_RECEIVED: off = HH_BRECV, wide = true; break;
      case WG_F_BYTES_SENT: off = HH_BSENT, wide = true; break;
      default: return false;
    }
    const std::vector<uint32_t> h = read_hdr();
    for (int i = 0; i < n; i++) {
      const uint32_t* r = &h[(size_t)i * st.hdrStride + off];
      dst[i] += wide ? (int64_t)((uint64_t)r[0] | ((uint64_t)r[1] << 32)) : (int64_t)r[0];
    }
    return true;
  }
};

ProtoHost* make_handel_host(Engine& e, const wg_handel_params& p, const wg_handel_init_state& st) {
  return new HandelHost(e, p, st);
}

}  // namespace wg

// ================================================================================================
// GSFSignature resident protocol: host side
#include "proto_gsf.hip.h"

namespace wg {

template void Engine::scan<GsfCondF>(const Group&, const GsfState*);
typedef SnapF<GsfState, G_TASK_DOCYCLE> GsfSnapF;
template void Engine::scan<GsfSnapF>(const Group&, const GsfState*);

__global__ void k_gsf_init(GsfState s, const uint8_t* down) {
  int node = s.lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= s.hi) return;
  // GSFNode(): verifiedSignatures.set(nodeId) (:179) — for every node, also the ones stopped later;
  // SFLevel() for level 0 (:263-270) and remainingCalls = peers.size() for the others (:282)
  s.V[(size_t)node * s.W + (node >> 6)] |= 1ULL << (node & 63);
  size_t i0 = (size_t)node * s.L;
  s.cV[i0] = 1;
  s.cU[i0] = 1;
  for (int l = 1; l < s.L; l++) s.rem[i0 + l] = down[node] ? 0 : 1 << (l - 1);
  // registerConditionalTask(checkSigs, 1, nodePairingTime, ...) for live nodes (:631-632)
  s.ctMinStart[node] = down[node] ? INT32_MAX : 1;
}

// GSFSignature.newConfIf (P/GSFSignature.java:670-683): some live node holds fewer than `threshold` signatures
__global__ void k_gsf_cont_if(const EngineDev* __restrict__ tab, const GsfState* __restrict__ stab, uint32_t* out) {
  const EngineDev& d = tab[blockIdx.y];
  const GsfState& s = stab[blockIdx.y];
  // (a few wavefronts per engine that stop at the first such node: see k_handel_cont_if)
  const int stride = (int)(gridDim.x * blockDim.x);
  for (int n0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) & ~63u); n0 < s.hi; n0 += stride) {
    if (__ballot(cont_if_known(out + blockIdx.y))) return;
    const int node = n0 + (int)WG_LANE;
    bool c = false;
    if (node >= s.lo && node < s.hi && !d.nodes.down[node]) {  // (sharded: the predicate over this shard's nodes)
      int tot = 0;
      for (int l = 0; l < s.L; l++) tot += s.cV[(size_t)node * s.L + l];
      c = tot < s.p.threshold;
    }
    if (__ballot(c)) {
      if (WG_LANE == 0) cont_if_set(out + blockIdx.y);
      return;
    }
  }
}

struct GsfHost : ProtoHost {
  GsfState st{};
  Engine& eng;
  uint32_t* dCont = nullptr;
  GsfHost(Engine& e, const wg_gsf_params& p, const wg_gsf_init_state& init) : eng(e) {
    const int32_t N = p.nodeCount;
    if ((int32_t)e.hx.size() != N) throw WgError(WG_EINVAL, "GSFSignature nodeCount != nodes in the network");
    if (N < 2 || (N & (N - 1)))
      throw WgError(WG_EUNSUPPORTED, "the resident GSFSignature needs a power-of-two nodeCount");
    if (N > (1 << 24)) throw WgError(WG_EUNSUPPORTED, "nodeCount > 2^24");
    if (!init.nodePairingTime || !init.peers) throw WgError(WG_EINVAL, "wg_gsf_init_state has NULL members");
    if (p.acceleratedCallsCount > 64)
      throw WgError(WG_EUNSUPPORTED, "acceleratedCallsCount > 64 (device multi-destination sends hold <= 64 ids)");
    // a node's events of the ms from its inbox line (k_deliver_inbox) where the engine is not allocated yet
    if (!e.allocated) e.wantInbox = true;
    if (p.periodDurationMs <= 0) throw WgError(WG_EINVAL, "periodDurationMs");
    int L = 1;
    while ((1 << L) <= N) L++;  // levels 0..log2(N) (:182-192)
    if (L > MAX_LEVELS) throw WgError(WG_EINVAL, "too many levels");
    const int W = N >= 64 ? N / 64 : 1;
    int Q = e.cfg.queue_cap > 0 ? e.cfg.queue_cap : 128;
    Q = (Q + 63) / 64 * 64;
    if (Q > G_MAX_Q) throw WgError(WG_EINVAL, "queue_cap must be <= 512 for GSFSignature");
    e.ensure_device();
    if (e.dev.inbox && e.dev.maxOut >= (1u << 28)) throw WgError(WG_EINVAL, "outbox_records must be below 2^28 (an inbox entry carries a task's outbox slot in 28 bits)");
    if (p.periodDurationMs >= e.dev.horizon) throw WgError(WG_ENOMEM, "horizon_ms <= period");
    st.p = p;
    st.N = N;
    st.L = L;
    st.W = W;
    st.Q = Q;
    st.SW = N >= 128 ? N / 128 : 1;
    // the big per-node arrays are held for the owned node range only, behind biased pointers (see HandelHost)
    const int32_t lo = st.lo = e.shardCount > 0 ? e.dev.shardLo : 0;
    const int32_t hi = st.hi = e.shardCount > 0 ? e.dev.shardHi : N;
    const size_t nLoc = (size_t)(hi - lo);
    auto rows = [&](auto* tag, size_t stride, bool zero) {
      typedef std::remove_pointer_t<decltype(tag)> T;
      return e.dalloc<T>(nLoc * stride, zero) - (size_t)lo * stride;
    };
    st.V = rows((uint64_t*)nullptr, W, true);
    st.IS = rows((uint64_t*)nullptr, W, true);
    st.IV = rows((uint64_t*)nullptr, W, true);
    st.peers = rows((int32_t*)nullptr, N - 1, false);
    st.pairing = e.dalloc<int32_t>(N);
    st.sigChecked = e.dalloc<int32_t>(N);
    st.sigQueueSize = e.dalloc<int32_t>(N);
    st.tvLen = e.dalloc<int32_t>(N);
    st.ctMinStart = e.dalloc<int32_t>(N);
    st.ctEpoch = e.dalloc<uint32_t>(N);
    const size_t NL = (size_t)N * L;
    st.pos = e.dalloc<int32_t>(NL);
    st.rem = e.dalloc<int32_t>(NL);
    st.cV = e.dalloc<int32_t>(NL);
    st.cIV = e.dalloc<int32_t>(NL);
    st.cU = e.dalloc<int32_t>(NL);
    st.tvEnt = rows((uint64_t*)nullptr, Q, false);
    st.tvUsed = rows((uint64_t*)nullptr, Q / 64, true);
    st.tvSig = rows((uint64_t*)nullptr, (size_t)Q * st.SW, false);
    {
      uint32_t off = 0;
      for (int l = 0; l < L; l++) {
        st.lvlOff[l] = off;
        off += l == 0 ? 0 : ((1 << (l - 1)) >= 64 ? (1 << (l - 1)) >> 6 : 1);
      }
      st.snapStride = off;
      st.snapNb = (uint32_t)(e.dev.horizon / p.periodDurationMs) + 2;  // a snapshot is read within < horizon ms
      uint64_t words = (uint64_t)st.snapNb * N * st.snapStride;
      if (words >= 0x100000000ull) throw WgError(WG_ENOMEM, "GSF snapshot ring exceeds 2^32 words: lower horizon_ms");
      st.snap = e.dalloc<uint64_t>(words, false);
    }
    e.dev.boundMsg = 0;           // onNewSig never sends
    e.dev.boundTask[0] = L - 1;   // doCycle: one send per level >= 1 (+1 periodic re-arm added by expand)
    e.dev.boundTask[1] = L - 1;   // updateVerifiedSignatures: one accelerated send per higher level
    e.dev.boundTask[2] = e.dev.boundTask[3] = 0;
    st.pend = e.dalloc<uint32_t>((size_t)N * G_PEND);
    st.pendFrom = e.dalloc<int32_t>((size_t)N * G_PEND);
    st.runList = e.dalloc<uint32_t>(N);
    st.runCount = e.dalloc<uint32_t>(1);
    st.runList2 = e.dalloc<uint32_t>(N);
    st.runCount2 = e.dalloc<uint32_t>(1);
    st.candFlag = e.dalloc<uint8_t>(((size_t)N + 3) / 4 * 4);
    st.candPend = e.dalloc<uint8_t>(N);
    st.condList = e.dalloc<uint32_t>(N);
    WG_HIP(hipMemcpy(st.pairing, init.nodePairingTime, 4 * (size_t)N, hipMemcpyHostToDevice));
    WG_HIP(hipMemcpy(st.peers + (size_t)lo * (N - 1), init.peers + (size_t)lo * (N - 1), 4 * nLoc * (N - 1),
                     hipMemcpyHostToDevice));
    st.snapIdx = nullptr;
    st.nSnap = nullptr;
    st.xsnap = nullptr;
    st.xsnapRows = 0;
    if (e.shardCount > 0) {
      st.snapIdx = e.dalloc<uint32_t>(e.dev.maxEvents, false);
      st.nSnap = e.dalloc<uint32_t>(1);
      st.xsnapRows = (uint32_t)N;
      st.xsnap = e.dalloc<int32_t>((size_t)st.xsnapRows * st.snapStride * 2, false);
    }
    hipLaunchKernelGGL(k_gsf_init, dim3(((int)nLoc + 255) / 256), dim3(256), 0, e.stream, st, e.dev.nodes.down);
    WG_HIP(hipStreamSynchronize(e.stream));
  }
  bool has_cond() const override { return true; }
  int levels() const override { return st.L; }
  bool cont_if(Engine& e, int32_t* out) override {
    if (!dCont) dCont = e.dalloc<uint32_t>(1);
    Group g = e.self();
    WG_HIP(hipMemsetAsync(dCont, 0, 4, e.stream));
    launch_cont_if(g, dCont);
    uint32_t v = 0;
    WG_HIP(hipMemcpyAsync(&v, dCont, 4, hipMemcpyDeviceToHost, e.stream));
    WG_HIP(hipStreamSynchronize(e.stream));
    *out = (int32_t)v;
    return true;
  }
  int host_msg_size(uint32_t msg) const override {
    int l = (int)(msg & 31u);
    return 1 + ((l == 0 ? 1 : (1 << (l - 1))) / 8) + 96;
  }
  int payload_bytes_of_level(int l) const override {
    return l == 0 ? 0 : ((1 << (l - 1)) >= 64 ? (1 << (l - 1)) / 8 : 8);
  }
  size_t state_size() const override { return sizeof(st); }
  const void* state_host() const override { return &st; }
  void launch_cond(Engine& profOwner, const Group& g) override {
    const GsfState* stab = (const GsfState*)g.stab;
    {
      Engine::ProfScope ps(profOwner, Engine::PC_COND_SELECT);
      hipLaunchKernelGGL(k_gsf_cond_pre, dim3((st.N + 255) / 256, g.R), dim3(256), 0, g.stream, g.tab, stab);
      // eight lanes per runner for the short lists; what is left over, one wavefront each
      hipLaunchKernelGGL(k_gsf_cond_a1g, dim3(grid_per_engine(GRID_COND_TAIL, g.R, 1024, st.N, 512), g.R), dim3(256), 0, g.stream, g.tab, stab);
      hipLaunchKernelGGL(k_gsf_cond_a1, dim3(grid_per_engine(GRID_COND_TAIL, g.R, 1024, st.N, 512), g.R), dim3(256), 0, g.stream, g.tab, stab, 1);
    }
    Engine::ProfScope ps(profOwner, Engine::PC_COND_REST);
    Engine::scan<GsfCondF>(g, stab);
    hipLaunchKernelGGL(k_gsf_cond_a2<false>, dim3(grid_per_engine(GRID_COND_TAIL, g.R, 1024, st.N, 1024), g.R), dim3(256), 0, g.stream, g.tab, stab);
  }
  // ---- node-range sharding (Engine::run_ms_sharded; the recipe of HandelHost) ----
  bool supports_shards() const override { return true; }
  bool shard_snap_is_scan() const override { return true; }
  uint32_t* shard_snap_enqueue(const Group& g) override {  // the doCycle snapshots (PARTIAL payloads) of this ms
    Engine::scan<GsfSnapF>(g, (const GsfState*)g.stab);
    return st.nSnap;
  }
  void shard_snap_exchange(Engine& e, const Group& g, uint32_t nSnap) override {
    const GsfState* stab = (const GsfState*)g.stab;
    hipLaunchKernelGGL((k_shard_snap<GsfState, G_TASK_DOCYCLE, true>), dim3(GRID_DELIVER_SMALL, 1), dim3(256), 0, g.stream, g.tab, stab);
    e.shard_allreduce(st.xsnap, (int64_t)nSnap * 2, Engine::XK_SNAPSHOTS);  // (nSnap: 64-bit words of the packed rows)
    hipLaunchKernelGGL((k_shard_snap<GsfState, G_TASK_DOCYCLE, false>), dim3(GRID_DELIVER_SMALL, 1), dim3(256), 0, g.stream, g.tab, stab);
  }
  // checkSigs' edge: which nodes register a task is summed across shards, so that the registrations keep their
  // node-id (= push) order; checkSigs draws nothing from rd
  uint32_t shard_cond(Engine& e, const Group& g) override {
    const GsfState* stab = (const GsfState*)g.stab;
    hipLaunchKernelGGL(k_gsf_cond_pre, dim3((st.hi - st.lo + 255) / 256, 1), dim3(256), 0, g.stream, g.tab, stab);
    hipLaunchKernelGGL(k_gsf_cond_a1, dim3(grid_node_waves(1), 1), dim3(256), 0, g.stream, g.tab, stab, 0);
    e.shard_allreduce(st.candFlag, ((int64_t)st.N + 3) / 4, Engine::XK_CANDIDATES);
    Engine::scan<GsfCondF>(g, stab);
    uint32_t nOut = 0;
    const uint32_t seq = e.publish_counts((const uint32_t*)((const char*)e.dev.g.raw + offsetof(Globals, nOut)), nullptr);
    hipLaunchKernelGGL(k_gsf_cond_a2<true>, dim3(GRID_COND_TAIL, 1), dim3(256), 0, g.stream, g.tab, stab);
    e.wait_counts(seq, &nOut, nullptr);  // (read while the tail runs)
    WG_HIP(hipMemsetAsync(st.candFlag, 0, ((size_t)st.N + 3) / 4 * 4, g.stream));  // the other shards' flags
    return nOut;
  }
  void launch_deliver(const Group& g) override {
    // the lean kernels first: a node's doCycle task (k_gsf_docycle, in the ms in which it can fire) and plain SendSigs
    // deliveries (k_gsf_lane) — between them every node without an updateVerifiedSignatures task (gsf_split_ok)
    const bool cycleRan = eng.dev.inbox && g.periodic_may_fire(G_TASK_DOCYCLE);
    // k_deliver_inbox visits the nodes k_gsf_lane lists (what the lean kernels did not take) instead of looking at every active
    // node's inbox count; WG_GSF_REST_LIST=0: as before
    const bool restList = !(getenv("WG_GSF_REST_LIST") && atoi(getenv("WG_GSF_REST_LIST")) == 0);  // (read per call: tests toggle it)
    if (cycleRan && st.L <= 16)  // sixteen lanes per node, four nodes per wavefront
      hipLaunchKernelGGL(k_gsf_docycle16, dim3(grid_node_waves(g.R), g.R), dim3(256), 0, g.stream, g.tab, (const GsfState*)g.stab);
    else if (cycleRan)
      hipLaunchKernelGGL(k_gsf_docycle<8>, dim3(grid_node_waves(g.R), g.R), dim3(256), 0, g.stream, g.tab, (const GsfState*)g.stab);
    if (eng.dev.inbox)
      hipLaunchKernelGGL(k_gsf_lane, dim3(grid_per_engine(GRID_LANE_NODES, g.R, 1024, st.N, 512), g.R), dim3(256), 0, g.stream, g.tab,
                         (const GsfState*)g.stab, cycleRan ? 1 : 0, restList ? 1 : 0);
    if (eng.dev.inbox && restList)  // ... and only the nodes k_gsf_lane listed (EngineDev::activeB)
      // (workgroups of one wavefront: at four wavefronts a SIMD a 256-thread workgroup waits for a free slot on all four SIMDs at
      // once and the visits end at very different times — 575 -> 602 M msgs/s at 256 copies, same box: profiles/r23f_*)
      hipLaunchKernelGGL((k_deliver_inbox<GsfProto, 4, true>), dim3(4 * grid_node_waves(g.R), g.R), dim3(64), 0, g.stream, g.tab,
                         (const GsfState*)g.stab);
    else if (eng.dev.inbox)  // a node's events from its inbox line (one 64-byte read instead of the list walk)
      hipLaunchKernelGGL((k_deliver_inbox<GsfProto, 4>), dim3(grid_node_waves(g.R), g.R), dim3(256), 0, g.stream, g.tab,
                         (const GsfState*)g.stab);
    else
      hipLaunchKernelGGL((k_deliver<GsfProto, 4>), dim3(grid_node_waves(g.R), g.R), dim3(256), 0, g.stream, g.tab,
                         (const GsfState*)g.stab);
  }
  bool launch_cont_if(const Group& g, uint32_t* dOut) override {
    hipLaunchKernelGGL(k_gsf_cont_if, dim3(std::max(1, std::min(4, (st.N + 255) / 256)), g.R), dim3(256), 0, g.stream, g.tab,
                       (const GsfState*)g.stab, dOut);
    return true;
  }
  bool read_i64(Engine&, int32_t field, int64_t* dst, int32_t n) override {
    const int32_t* src = nullptr;
    switch (field) {
      case WG_F_GSF_SIG_CHECKED: src = st.sigChecked; break;
      case WG_F_GSF_SIG_QUEUE_SIZE: src = st.sigQueueSize; break;
      case WG_F_GSF_TO_VERIFY_SIZE: src = st.tvLen; break;
      case WG_F_NODE_PAIRING_TIME: src = st.pairing; break;
      case WG_F_GSF_VERIFIED_CARDINALITY: {
        std::vector<int32_t> h((size_t)n * st.L);
        WG_HIP(hipMemcpy(h.data(), st.cV, 4 * h.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) {
          int64_t t = 0;
          for (int l = 0; l < st.L; l++) t += h[(size_t)i * st.L + l];
          dst[i] = t;
        }
        return true;
      }
      default: return false;
    }
    std::vector<int32_t> h(n);
    WG_HIP(hipMemcpy(h.data(), src, 4 * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) dst[i] = h[i];
    return true;
  }
  bool read_level_i32(Engine&, int32_t field, int32_t* dst, int32_t n, int32_t L) override {
    if (n != st.N || L != st.L) throw WgError(WG_EINVAL, "shape must be [nodeCount][levels]");
    const int32_t* src = field == WG_LF_POS_IN_LEVEL ? st.pos : field == WG_LF_REMAINING_CALLS ? st.rem : nullptr;
    if (!src) return false;
    WG_HIP(hipMemcpy(dst, src, 4 * (size_t)n * L, hipMemcpyDeviceToHost));
    return true;
  }
  bool read_bits(Engine&, int32_t field, uint64_t* dst, int32_t n, int32_t w) override {
    if (n != st.N || w != st.W) throw WgError(WG_EINVAL, "shape must be [nodeCount][max(1, nodeCount/64)]");
    const uint64_t* src = nullptr;
    switch (field) {
      case WG_B_GSF_VERIFIED: src = st.V; break;
      case WG_B_GSF_INDIVIDUAL: src = st.IS; break;
      case WG_B_GSF_INDIV_VERIFIED: src = st.IV; break;
      default: return false;
    }
    memset(dst, 0, 8 * (size_t)n * w);  // (sharded: zeros for the nodes of other shards)
    WG_HIP(hipMemcpy(dst + (size_t)st.lo * w, src + (size_t)st.lo * w, 8 * (size_t)(st.hi - st.lo) * w, hipMemcpyDeviceToHost));
    return true;
  }
};

ProtoHost* make_gsf_host(Engine& e, const wg_gsf_params& p, const wg_gsf_init_state& st) { return new GsfHost(e, p, st); }

}  // namespace wg

// ================================================================================================
// San Fermin resident protocol: host side
#include "proto_sanfermin.hip.h"

namespace wg {

struct SfHost : ProtoHost {
  SfState st{};
  Engine& eng;
  uint32_t* dCont = nullptr;
  SfHost(Engine& e, const wg_sanfermin_params& p) : eng(e) {
    const int32_t N = p.nodeCount;
    if ((int32_t)e.hx.size() != N) throw WgError(WG_EINVAL, "San Fermin nodeCount != nodes in the network");
    if (N < 2 || (N & (N - 1)))
      throw WgError(WG_EUNSUPPORTED, "San Fermin needs a power-of-two nodeCount (toBinaryID, P/SanFerminHelper.java:158-171)");
    if (p.candidateCount < 0 || p.candidateCount > 62)
      throw WgError(WG_EUNSUPPORTED, "candidateCount > 62 (device multi-destination sends hold <= 64 ids)");
    if (p.pairingTime <= 0 || p.replyTimeout <= 0) throw WgError(WG_EINVAL, "pairingTime / replyTimeout");
    int P = 0;
    while ((1 << (P + 1)) <= N) P++;
    if (!e.allocated) e.horizonFloor = std::max(e.horizonFloor, std::max(p.replyTimeout, p.pairingTime) + 8);  // its tasks' delays
    e.ensure_device();
    if (p.replyTimeout >= e.dev.horizon - 1 || p.pairingTime >= e.dev.horizon - 1)
      throw WgError(WG_ENOMEM, "horizon_ms <= replyTimeout: raise wg_config.horizon_ms");
    st.p = p;
    st.N = N;
    st.P = P;
    st.W = N >= 64 ? N / 64 : 1;
    st.cpl = e.dalloc<int32_t>(N);
    st.agg = e.dalloc<int32_t>(N);
    st.sentReq = e.dalloc<int32_t>(N);
    st.recvReq = e.dalloc<int32_t>(N);
    st.thresholdAt = e.dalloc<int32_t>(N);
    st.flags = e.dalloc<uint32_t>(N);
    st.cacheMask = e.dalloc<uint32_t>(N);
    st.cache = e.dalloc<int32_t>((size_t)N * (P + 1));
    st.used = e.dalloc<uint64_t>((size_t)N * st.W);
    st.pending = e.dalloc<uint64_t>((size_t)N * st.W);
    e.dev.boundMsg = 2;  // a reply | a transition task | a request (one multi-destination send) + its timeout task
    for (int k = 0; k < 4; k++) e.dev.boundTask[k] = 2;
    hipLaunchKernelGGL(k_sf_init, dim3((N + 255) / 256), dim3(256), 0, e.stream, st);
    WG_HIP(hipStreamSynchronize(e.stream));
  }
  void launch_deliver(const Group& g) override {
    hipLaunchKernelGGL((k_deliver<SfProto, 4>), dim3(grid_node_waves(g.R), g.R), dim3(256), 0, g.stream, g.tab,
                       (const SfState*)g.stab);
  }
  size_t state_size() const override { return sizeof(st); }
  const void* state_host() const override { return &st; }
  int host_msg_size(uint32_t) const override { return 4 + st.p.signatureSize; }
  int levels() const override { return st.P; }
  // node-range sharding: a visit touches the visited node's rows only, payloads are message words, no conditional tasks;
  // the (shuffled) multi-destination requests go through the replicated envelope creation (k_shard_multi_*)
  bool supports_shards() const override { return true; }
  bool launch_cont_if(const Group& g, uint32_t* dOut) override {
    hipLaunchKernelGGL(k_sf_cont_if, dim3((st.N + 255) / 256, g.R), dim3(256), 0, g.stream, g.tab, (const SfState*)g.stab, dOut);
    return true;
  }
  bool cont_if(Engine& e, int32_t* out) override {
    if (!dCont) dCont = e.dalloc<uint32_t>(1);
    Group g = e.self();
    WG_HIP(hipMemsetAsync(dCont, 0, 4, e.stream));
    launch_cont_if(g, dCont);
    uint32_t v = 0;
    WG_HIP(hipMemcpyAsync(&v, dCont, 4, hipMemcpyDeviceToHost, e.stream));
    WG_HIP(hipStreamSynchronize(e.stream));
    *out = (int32_t)v;
    return true;
  }
  bool read_i64(Engine&, int32_t field, int64_t* dst, int32_t n) override {
    const void* src = nullptr;
    switch (field) {
      case WG_F_SF_AGG_VALUE: src = st.agg; break;
      case WG_F_SF_PREFIX_LENGTH: src = st.cpl; break;
      case WG_F_SF_FLAGS: src = st.flags; break;
      case WG_F_SF_SENT_REQUESTS: src = st.sentReq; break;
      case WG_F_SF_RECEIVED_REQUESTS: src = st.recvReq; break;
      case WG_F_SF_THRESHOLD_AT: src = st.thresholdAt; break;
      default: return false;
    }
    std::vector<int32_t> h(n);
    WG_HIP(hipMemcpy(h.data(), src, 4 * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) dst[i] = h[i];
    return true;
  }
};

ProtoHost* make_sanfermin_host(Engine& e, const wg_sanfermin_params& p) { return new SfHost(e, p); }

}  // namespace wg

// ================================================================================================
// Casper IMD resident protocol: host side
#include "proto_casper.hip.h"

namespace wg {

struct CasperHost : ProtoHost {
  CasperState st{};
  Engine& eng;
  CasperHost(Engine& e, const wg_casper_params& p) : eng(e) {
    const int32_t N = 1 + p.blockProducersCount + p.cycleLength * p.attestersPerRound;
    if ((int32_t)e.hx.size() != N)
      throw WgError(WG_EINVAL, "Casper IMD: the network must hold 1 observer + blockProducersCount + cycleLength * attestersPerRound nodes");
    if (p.cycleLength <= 0 || p.blockProducersCount <= 0 || p.attestersPerRound <= 0 || p.maxSlots <= 0)
      throw WgError(WG_EINVAL, "Casper IMD parameters");
    if (p.blockConstructionTime <= 0 || p.attestationConstructionTime <= 0)
      throw WgError(WG_EUNSUPPORTED, "construction times must be >= 1 ms (a sendAll for the current ms would have to be delivered in it)");
    if (!e.allocated) {
      e.sendAllCapacity = p.attestersPerRound + 2;  // one round of attesters votes in the same ms (+ a block)
      // every in-flight sendAll holds N destinations until its last hop: one slot's worth of them is in flight at most
      const int64_t need = (int64_t)(p.attestersPerRound + 4) * N * 2;
      if (e.cfg.chain_dests == 0) e.cfg.chain_dests = std::max<int64_t>(1 << 20, need);
      if (e.cfg.outbox_records == 0)  // chain runs: every in-flight envelope delivers ~N / latency-spread hops per ms
        e.cfg.outbox_records = std::max<int64_t>(1 << 16, (int64_t)(p.attestersPerRound + 4) * N / 8 + 4 * (int64_t)N);
      if (e.cfg.chain_slots == 0) e.cfg.chain_slots = std::max(4096, 64 * (p.attestersPerRound + 4));
      // a sendAll's first hop arrives construction time + latency ahead: the bucket ring must reach that far; the
      // periodic re-arms (>= 8 s) and the byzantine producer's delayed build go through the far buffer
      e.horizonExtra = std::max(e.horizonExtra, std::max(p.blockConstructionTime, p.attestationConstructionTime) + 1);
      e.farCapacity = std::max(e.farCapacity, 4 * N + 1024);
    }
    e.ensure_device();
    if (e.dev.maxSendAll == 0) throw WgError(WG_ESTATE, "load the Casper protocol before the engine allocates");
    st.p = p;
    st.N = N;
    st.B = p.maxSlots + 8;
    st.A = p.maxSlots * p.attestersPerRound;
    st.Aw = (st.A + 63) / 64;
    st.Bw = (st.B + 63) / 64;
    // Per-node rows are held for the nodes [lo, hi) this engine owns — everything when it is not sharded — behind pointers
    // offset so that kernels keep indexing by node id (as HandelHost does); the block / attestation tables are replicated.
    lo = e.shardCount > 0 ? e.dev.shardLo : 0;
    hi = e.shardCount > 0 ? e.dev.shardHi : N;
    const size_t own = (size_t)(hi - lo);
    st.head = e.dalloc<int32_t>(own) - lo;
    st.recv = e.dalloc<uint64_t>(own * st.Aw) - (size_t)lo * st.Aw;
    st.BS = 4;
    while (st.BS < 3 * st.Bw) st.BS *= 2;
    {
      uint64_t* rec = e.dalloc<uint64_t>(own * st.BS) - (size_t)lo * st.BS;  // per node: blkRecv[Bw] | reeval[Bw] | headsAtt[Bw] | pad
      st.blkRecv = rec;
      st.reeval = rec + st.Bw;
      st.headsAtt = rec + 2 * st.Bw;
    }
    if (e.shardCount > 0) {
      xtabWords = XT_HEAD + 2 * st.Aw + p.attestersPerRound;
      st.xtab = e.dalloc<int32_t>((size_t)xtabWords);
      st.anyTask = e.dalloc<uint32_t>(1);
    }
    st.wf = e.dalloc<int32_t>(4);
    st.bHeight = e.dalloc<int32_t>(st.B);
    st.bParent = e.dalloc<int32_t>(st.B);
    st.bProducer = e.dalloc<int32_t>(st.B);
    st.bTime = e.dalloc<int32_t>(st.B);
    st.nBlocks = e.dalloc<uint32_t>(1);
    st.lastBlockMs = e.dalloc<int32_t>(1);
    st.blockAtt = e.dalloc<uint64_t>((size_t)st.B * st.Aw);
    st.headMask = e.dalloc<uint64_t>((size_t)st.B * st.Aw);
    st.attestsMask = e.dalloc<uint64_t>((size_t)st.B * st.Aw);
    st.attHead = e.dalloc<int32_t>(st.A);
    st.mixed = e.dalloc<uint8_t>(N);
    // a delayed byzantine producer can build in another producer's ms: such ms go through k_casper_seq (CasperState::builds)
    st.seqCapable = p.byzDelay != 0 && e.shardCount == 0 ? 1u : 0u;
    if (p.randomOnTies || st.seqCapable) st.seqBits = e.dalloc<uint64_t>(((size_t)e.dev.maxEvents + 63) / 64 + 64);
    st.tBits = nullptr;
    st.seqPos = nullptr;
    st.xseq = nullptr;
    if (p.randomOnTies && e.shardCount > 0) {  // the ordered visit goes round the shards (k_casper_seq_shard)
      st.tBits = e.dalloc<uint64_t>(((size_t)e.dev.maxEvents + 63) / 64 + 64);
      st.seqPos = e.dalloc<uint32_t>(1);
      st.xseq = e.dalloc<int32_t>(2);
    }
    st.forked = e.dalloc<uint32_t>(1);
    st.builds = e.dalloc<uint32_t>(1);
    // attestation-only nodes are delivered one lane per event (k_casper_attestations): k_casper_seq hands k_deliver an empty
    // set through the mixed flags, and a sharded engine's table exchange hangs on k_casper_classify's anyTask flag
    st.laneEvents = 1u;
    e.dev.laneMsgPlus1 = (uint32_t)C_MSG_ATTESTATION + 1u;  // attestations are not threaded onto inbox lists
    e.dev.boundMsg = 1;  // ByzBlockProducerWF.onBlock: one sendAll or one registerTask
    for (int k = 0; k < 4; k++) e.dev.boundTask[k] = 1;  // one sendAll (+ the periodic re-arm expand adds)
    hipLaunchKernelGGL(k_casper_init, dim3((N + 255) / 256), dim3(256), 0, e.stream, st, lo, hi);
    WG_HIP(hipStreamSynchronize(e.stream));
  }
  int32_t lo = 0, hi = 0, xtabWords = 0;
  const int attGrid = GRID_RESOLVE;  // (per engine)
  // ---- node-range sharding (Engine::run_ms_sharded): a delivery touches the receiver's rows only; sendAll goes through the
  // replicated envelope creation (k_shard_multi_*, k_sendall_*); what the ms's action()s added to the replicated block /
  // attestation tables is exchanged after the delivery pass (CasperState::xtab, k_casper_shard_apply) — in the ms that hold
  // a block or a task at all, which every shard sees from the replicated event list (anyTask)
  bool supports_shards() const override { return true; }
  uint32_t* shard_snap_enqueue(const Group&) override { return st.anyTask; }
  void shard_snap_exchange(Engine& e, const Group& g, uint32_t) override {
    e.shard_allreduce(st.xtab, xtabWords);
    hipLaunchKernelGGL(k_casper_shard_apply, dim3(1, 1), dim3(256), 0, g.stream, g.tab, (const CasperState*)g.stab);
  }
  // randomOnTies on a sharded engine: the delivery pass with its rounds of the ordered visit (k_casper_seq_shard) — one
  // wavefront and one two-word collective per change of owner among the ms's blocks and tasks, one more to find that none is left
  bool shard_deliver(Engine& e, const Group& g) override {
    if (!st.p.randomOnTies) return false;
    const CasperState* stab = (const CasperState*)g.stab;
    hipLaunchKernelGGL(k_casper_classify, dim3(GRID_RESOLVE, g.R), dim3(256), 0, g.stream, g.tab, stab);
    hipLaunchKernelGGL(k_casper_attestations, dim3(attGrid, g.R), dim3(256), 0, g.stream, g.tab, stab);
    hipLaunchKernelGGL(k_casper_mark_shard, dim3(GRID_RESOLVE, g.R), dim3(256), 0, g.stream, g.tab, stab);
    uint32_t cursor = 0, drawBase = 0;
    for (int round = 0;; round++) {
      if (round > (int)e.dev.maxEvents + 1) throw WgError(WG_ESTATE, "the ordered visit of a sharded randomOnTies run does not end");
      WG_HIP(hipMemsetAsync(st.xseq, 0, 2 * sizeof(int32_t), g.stream));
      hipLaunchKernelGGL(k_casper_seq_shard, dim3(1, g.R), dim3(64), 0, g.stream, g.tab, stab, cursor, drawBase);
      e.shard_allreduce(st.xseq, 2);
      int32_t h[2] = {0, 0};
      WG_HIP(hipMemcpyAsync(h, st.xseq, sizeof(h), hipMemcpyDeviceToHost, g.stream));
      WG_HIP(hipStreamSynchronize(g.stream));
      if (h[0] == 0) break;  // no block or task behind the cursor: every shard has finished its share
      cursor = (uint32_t)h[0] - 1u;
      drawBase += (uint32_t)h[1];
    }
    // (k_deliver: every node with a block or a task has been visited and unflagged — its visit_skip admits none; it retires the ms's node list)
    hipLaunchKernelGGL((k_deliver<CasperProto, 4>), dim3(4 * grid_node_waves(g.R), g.R), dim3(64), 0, g.stream, g.tab, stab);  // (one-wavefront workgroups: +5 %, profiles/r23f_*)
    return true;
  }
  void launch_deliver(const Group& g) override {
    const CasperState* stab = (const CasperState*)g.stab;
    if (st.laneEvents) {  // attestation-only nodes: one lane per event (see k_casper_attestations)
      hipLaunchKernelGGL(k_casper_classify, dim3(GRID_RESOLVE, g.R), dim3(256), 0, g.stream, g.tab, stab);
      // (a latency-bound pass of scattered atomics: as many wavefronts in flight as the chip holds)
      hipLaunchKernelGGL(k_casper_attestations, dim3(attGrid, g.R), dim3(256), 0, g.stream, g.tab, stab);
    }
    if (st.seqCapable) hipLaunchKernelGGL(k_casper_builds, dim3(GRID_RESOLVE, g.R), dim3(256), 0, g.stream, g.tab, stab);
    if (st.p.randomOnTies || st.seqCapable) {  // a tie's nextBoolean() needs the global event order: one wavefront — once the chain has forked
      // (CasperState::forked, read on the device: both return at once before that and k_deliver below does the visits;
      // after it k_casper_seq has cleared every mixed flag and k_deliver's visit_skip admits no node)
      hipLaunchKernelGGL(k_casper_mark, dim3(GRID_RESOLVE, g.R), dim3(256), 0, g.stream, g.tab, stab);
      hipLaunchKernelGGL(k_casper_seq, dim3(1, g.R), dim3(64), 0, g.stream, g.tab, stab);
    }
    hipLaunchKernelGGL((k_deliver<CasperProto, 4>), dim3(4 * grid_node_waves(g.R), g.R), dim3(64), 0, g.stream, g.tab, stab);  // (one-wavefront workgroups: +5 %, profiles/r23f_*)
  }
  size_t state_size() const override { return sizeof(st); }
  const void* state_host() const override { return &st; }
  bool unit_message_size() const override { return true; }  // blocks and attestations: Message.size() default 1
  int variant() const override { return (int)st.seqCapable | (st.p.randomOnTies ? 2 : 0); }  // (they select launches)
  bool read_i64(Engine&, int32_t field, int64_t* dst, int32_t n) override {
    if (field < WG_F_CASPER_HEAD_HEIGHT || field > WG_F_CASPER_ATTESTATIONS_HELD) return false;
    if (n != st.N) throw WgError(WG_EINVAL, "n must be the node count");
    for (int i = 0; i < n; i++) dst[i] = 0;  // (sharded: zeros for the nodes of other shards)
    std::vector<int32_t> head(n), bh(st.B), bt(st.B);
    WG_HIP(hipMemcpy(head.data() + lo, st.head + lo, 4 * (size_t)(hi - lo), hipMemcpyDeviceToHost));
    WG_HIP(hipMemcpy(bh.data(), st.bHeight, 4 * (size_t)st.B, hipMemcpyDeviceToHost));
    WG_HIP(hipMemcpy(bt.data(), st.bTime, 4 * (size_t)st.B, hipMemcpyDeviceToHost));
    auto popcounts = [&](const uint64_t* rows, int words, int stride = 0) {  // (stride: words between two nodes' rows)
      if (stride == 0) stride = words;
      if (hi <= lo) return;
      std::vector<uint64_t> h((size_t)(hi - lo - 1) * stride + words);
      WG_HIP(hipMemcpy(h.data(), rows + (size_t)lo * stride, 8 * h.size(), hipMemcpyDeviceToHost));
      for (int i = lo; i < hi; i++) {
        int64_t t = 0;
        for (int w = 0; w < words; w++) t += __builtin_popcountll(h[(size_t)(i - lo) * stride + w]);
        dst[i] = t;
      }
    };
    switch (field) {
      case WG_F_CASPER_HEAD_HEIGHT:
        for (int i = lo; i < hi; i++) dst[i] = bh[head[i]];
        return true;
      case WG_F_CASPER_HEAD_TIME:
        for (int i = lo; i < hi; i++) dst[i] = bt[head[i]];
        return true;
      case WG_F_CASPER_HEAD_ID:
        for (int i = lo; i < hi; i++) dst[i] = head[i];
        return true;
      case WG_F_CASPER_HEADS_ATTESTED: popcounts(st.headsAtt, st.Bw, st.BS); return true;
      case WG_F_CASPER_BLOCKS_RECEIVED: popcounts(st.blkRecv, st.Bw, st.BS); return true;
      default: popcounts(st.recv, st.Aw); return true;
    }
  }
};

ProtoHost* make_casper_host(Engine& e, const wg_casper_params& p) { return new CasperHost(e, p); }

}  // namespace wg

// ================================================================================================
// P2PFlood resident protocol: host side
#include "proto_p2pflood.hip.h"

namespace wg {

struct FloodHost : ProtoHost {
  FloodState st{};
  FloodHost(Engine& e, const wg_p2pflood_params& p, const wg_p2pflood_init_state& init) {
    const int32_t N = p.nodeCount;
    if ((int32_t)e.hx.size() != N) throw WgError(WG_EINVAL, "P2PFlood nodeCount != nodes in the network");
    if (!init.peers || !init.peerCount || !init.senders) throw WgError(WG_EINVAL, "wg_p2pflood_init_state has NULL members");
    if (p.msgCount < 1 || p.msgCount > 64) throw WgError(WG_EUNSUPPORTED, "msgCount must be 1..64 on the device");
    if (init.maxPeers < 1 || init.maxPeers > 64)
      throw WgError(WG_EUNSUPPORTED, "a node with more than 64 peers (device multi-destination sends hold <= 64 ids)");
    if (p.delayBetweenSends < 0 || p.delayBetweenSends >= (1 << 20) || p.delayBeforeResent < 0) throw WgError(WG_EINVAL, "delays");
    if (!e.allocated)  // a hop arrives delayBeforeResent + up to 64 * (delayBetweenSends + 1) + latency ahead
      e.horizonExtra = std::max(e.horizonExtra, p.delayBeforeResent + 1 + init.maxPeers * (p.delayBetweenSends + 1));
    e.ensure_device();
    st.p = p;
    st.N = N;
    st.maxPeers = init.maxPeers;
    st.peers = e.dalloc<int32_t>((size_t)N * st.maxPeers, false);
    st.peerCnt = e.dalloc<int32_t>(N, false);
    st.received = e.dalloc<uint64_t>(N);
    WG_HIP(hipMemcpy(st.peers, init.peers, 4 * (size_t)N * st.maxPeers, hipMemcpyHostToDevice));
    WG_HIP(hipMemcpy(st.peerCnt, init.peerCount, 4 * (size_t)N, hipMemcpyHostToDevice));
    int32_t* dS = nullptr;
    WG_HIP(hipMalloc((void**)&dS, 4 * (size_t)p.msgCount));
    WG_HIP(hipMemcpy(dS, init.senders, 4 * (size_t)p.msgCount, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_flood_init, dim3(1), dim3(64), 0, e.stream, st, e.dev.nodes, dS, p.msgCount);
    WG_HIP(hipStreamSynchronize(e.stream));
    (void)hipFree(dS);
    e.dev.boundMsg = 1;  // one (delayed, shuffled) multi-destination send per first receipt
    for (int k = 0; k < 4; k++) e.dev.boundTask[k] = 0;
  }
  // node-range sharding: a visit touches the visited node's `received` word only; its one (shuffled, delayed) multi-
  // destination send goes through the replicated envelope creation with explicit arrivals (k_shard_multi_*)
  bool supports_shards() const override { return true; }
  void launch_deliver(const Group& g) override {
    hipLaunchKernelGGL((k_deliver<FloodProto, 4>), dim3(grid_node_waves(g.R), g.R), dim3(256), 0, g.stream, g.tab,
                       (const FloodState*)g.stab);
  }
  size_t state_size() const override { return sizeof(st); }
  const void* state_host() const override { return &st; }
  bool read_i64(Engine&, int32_t field, int64_t* dst, int32_t n) override {
    if (field == WG_F_FLOOD_RECEIVED) {
      std::vector<uint64_t> h(n);
      WG_HIP(hipMemcpy(h.data(), st.received, 8 * (size_t)n, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; i++) dst[i] = __builtin_popcountll(h[i]);
      return true;
    }
    if (field == WG_F_FLOOD_PEER_COUNT) {
      std::vector<int32_t> h(n);
      WG_HIP(hipMemcpy(h.data(), st.peerCnt, 4 * (size_t)n, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; i++) dst[i] = h[i];
      return true;
    }
    return false;
  }
};

ProtoHost* make_p2pflood_host(Engine& e, const wg_p2pflood_params& p, const wg_p2pflood_init_state& st) {
  return new FloodHost(e, p, st);
}


// ---- wg_selftest: the wave / block primitives of engine_kernels.hip.h, one at a time, on the caller's inputs -------------
// The reductions and scans have two forms — DPP row operations (gfx950: what the product runs) and a shuffle form (the CPU
// wave emulator of tests/emu) — and are otherwise reached only through whole-protocol runs. This kernel calls ONE of them on
// up to 1024 values so that a test can hold the result against numpy (tests/test_gpu_wave_primitives.py). Their contract is
// the one every call site keeps: all 64 lanes of the wavefront make the call (a lane without an item contributes the
// operation's identity) — the result of a reduction is read from lane 63 with v_readlane, whatever EXEC is.
enum SelfTestOp : int32_t {
  ST_REDUCE_ADD32 = 0, ST_REDUCE_ADD64, ST_REDUCE_MIN_I32, ST_REDUCE_MAX_I32, ST_INCL_SCAN32, ST_INCL_SCAN64, ST_LANE_BCAST,
  ST_LANE_BCAST64, ST_GROUP8_SUM64, ST_GROUP8_OR64, ST_GROUP8_MIN_I32, ST_GROUP8_MAX_I32, ST_BLOCK_EXCL_SCAN32, ST_TILE_RANK,
  ST_SHFL64, ST_BLOCK_SUM64, ST_OPS
};
__global__ void __launch_bounds__(1024) k_selftest(int32_t op, int32_t aux, const uint64_t* __restrict__ in, int32_t n, uint64_t* __restrict__ out) {
  __shared__ uint32_t sh16[16];
  __shared__ uint64_t sh64[16];
  WG_DYN_LDS(uint32_t, hist);  // [256] (ST_TILE_RANK)
  const int i = (int)threadIdx.x;
  const uint64_t v = i < n ? in[i] : 0ULL;
  uint64_t r = 0;
  switch (op) {
    case ST_REDUCE_ADD32: r = wave_reduce_add32((uint32_t)v); break;
    case ST_REDUCE_ADD64: r = wave_reduce_add64(v); break;
    // (lanes beyond n: the identity, as the call sites pass it)
    case ST_REDUCE_MIN_I32: r = (uint64_t)(uint32_t)wave_reduce_min_i32(i < n ? (int32_t)(uint32_t)v : INT32_MAX); break;
    case ST_REDUCE_MAX_I32: r = (uint64_t)(uint32_t)wave_reduce_max_i32(i < n ? (int32_t)(uint32_t)v : INT32_MIN); break;
    case ST_INCL_SCAN32: r = wave_incl_scan32((uint32_t)v); break;
    case ST_INCL_SCAN64: r = wave_incl_scan64(v); break;
    case ST_LANE_BCAST: r = lane_bcast((uint32_t)v, aux & 63); break;
    case ST_LANE_BCAST64: r = lane_bcast64(v, aux & 63); break;
    case ST_GROUP8_SUM64: r = group8_sum64(v); break;
    case ST_GROUP8_OR64: r = group8_or64(v); break;
    case ST_GROUP8_MIN_I32: r = (uint64_t)(uint32_t)group8_min_i32(i < n ? (int32_t)(uint32_t)v : INT32_MAX); break;
    case ST_GROUP8_MAX_I32: r = (uint64_t)(uint32_t)group8_max_i32(i < n ? (int32_t)(uint32_t)v : INT32_MIN); break;
    case ST_BLOCK_EXCL_SCAN32: {
      uint32_t total;
      r = block_excl_scan32_1024((uint32_t)v, sh16, &total);
      if (i == 0) out[blockDim.x] = total;
      break;
    }
    case ST_TILE_RANK: {  // in[i] = bin | valid << 32; aux = bits of a bin; out[i] = stable rank among equal bins, then the bin totals
      const int bins = 1 << aux;
      for (int b = i; b < bins; b += (int)blockDim.x) hist[b] = 0;
      __syncthreads();
      const bool valid = i < n && ((v >> 32) & 1ULL);
      r = tile_rank(hist, (int)((uint32_t)v & (uint32_t)(bins - 1)), valid, aux);
      __syncthreads();
      for (int b = i; b < bins; b += (int)blockDim.x) out[blockDim.x + b] = hist[b];
      break;
    }
    case ST_SHFL64: r = shfl64(v, (i + aux) & 63); break;
    case ST_BLOCK_SUM64: r = block_sum64(v, sh64); break;
    default: break;
  }
  out[i] = r;
}
// (no engine: the call makes its own buffers on the current device; threads = the block the primitive runs in — 64 for
// the wave forms, up to 1024 for the block forms; out holds threads words, + 1 (ST_BLOCK_EXCL_SCAN32: the total) or
// + 2^aux (ST_TILE_RANK: the bins' totals))
void selftest(int32_t op, int32_t aux, const uint64_t* in, int32_t n, int32_t threads, uint64_t* out, int32_t nOut) {
  if (op < 0 || op >= ST_OPS || !in || !out || n < 0 || n > threads || threads < 64 || threads > 1024 || (threads & 63))
    throw WgError(WG_EINVAL, "wg_selftest: op / sizes");
  if (op == ST_TILE_RANK && (aux < 1 || aux > 8)) throw WgError(WG_EINVAL, "wg_selftest: tile_rank takes 1 .. 8 bin bits");
  const int32_t need = threads + (op == ST_BLOCK_EXCL_SCAN32 ? 1 : op == ST_TILE_RANK ? (1 << aux) : 0);
  if (nOut < need) throw WgError(WG_EINVAL, "wg_selftest: out is too short");
  uint64_t *dIn = nullptr, *dOut = nullptr;
  WG_HIP(hipMalloc((void**)&dIn, 8 * (size_t)std::max(1, n)));
  struct Free {
    void*& p;
    ~Free() { (void)hipFree(p); }
  } g1{(void*&)dIn};
  WG_HIP(hipMalloc((void**)&dOut, 8 * (size_t)need));
  Free g2{(void*&)dOut};
  if (n) WG_HIP(hipMemcpy(dIn, in, 8 * (size_t)n, hipMemcpyHostToDevice));
  WG_HIP(hipMemset(dOut, 0, 8 * (size_t)need));
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(threads), sizeof(uint32_t) * 256, 0, op, aux, (const uint64_t*)dIn, n, dOut);
  WG_HIP(hipDeviceSynchronize());
  WG_HIP(hipMemcpy(out, dOut, 8 * (size_t)need, hipMemcpyDeviceToHost));
}

}  // namespace wg
