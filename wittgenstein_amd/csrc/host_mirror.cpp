// Host-side mirror of the reference's Protocol.init() for the resident protocols (see
// include/wittgpu_host.h). Pure host C++ over the public C ABI: nothing here touches engine
// internals, so the same sequence is what a Java `init()` performs through the JNI shim.
#include <algorithm>
#include <chrono>
#include <map>
#include <set>
#include <stdexcept>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/wittgpu_host.h"
#include "jdk_random.h"

using wg::JavaRandom;

static thread_local std::string g_err;
static thread_local double g_initSeconds = 0;
static thread_local int32_t g_initOnDevice = 0;

namespace {

// City builders / city latency models: their inputs are data of the caller (the reference reads cities.csv and the
// wondernetwork ping files, C/geoinfo/GeoAllCities.java, T/CSVLatencyReader.java) registered once per process with
// wgh_register_city_builder / wgh_register_city_latency (include/wittgpu_host.h); the mirrors then restate what
// NodeBuilderWithCity does with them per node (C/NodeBuilder.java:98-147).
struct CityTable {
  std::vector<float> cum;            // CityInfo.cumulativeProbability, in citiesInfo.entrySet() order
  std::vector<int32_t> mercX, mercY;
  int32_t listSize = 0;              // cities.size() of the builder's list
};
struct CityLatency {
  int32_t mode = 0, nCities = 0;
  std::vector<int32_t> tab;
  std::vector<float> ping;
  std::vector<double> jit;
};
std::map<std::string, CityTable>& city_tables() {
  static std::map<std::string, CityTable> m;
  return m;
}
std::map<std::string, CityLatency>& city_latencies() {
  static std::map<std::string, CityLatency> m;
  return m;
}

// RegistryNodeBuilders (C/RegistryNodeBuilders.java:28-81): RANDOM, and AWS / CITIES once their tables are registered.
struct Builder {
  bool speedUniform = false;  // the registry's "GAUSSIAN" entries install UniformSpeed (:59-61)
  double tor = 0.0;
  const CityTable* city = nullptr;
};
bool parse_builder(const char* name, Builder& b) {
  std::string s = name ? name : "";
  if (s.empty()) return true;
  const std::string site = s.substr(0, s.find("_SPEED="));
  if (site == "AWS" || site == "CITIES") {
    auto it = city_tables().find(site);
    if (it == city_tables().end()) {
      g_err = s + ": register the " + site + " city table first (wgh_register_city_builder)";
      return false;
    }
    b.city = &it->second;
  } else if (s.rfind("RANDOM_SPEED=", 0) != 0) {
    g_err = s + " not in the registry";
    return false;
  }
  b.speedUniform = s.find("SPEED=GAUSSIAN") != std::string::npos;
  size_t p = s.find("_TOR=");
  if (p == std::string::npos) {
    g_err = s + " not in the registry";
    return false;
  }
  double tor = atof(s.c_str() + p + 5);
  if (tor > 0.001) b.tor = tor;
  return true;
}

struct NodeSoA {
  std::vector<int32_t> x, y, extra, city;
  bool bad = false;  // a draw fell past the last city's cumulative probability
  std::vector<uint8_t> down;
  std::vector<double> speed;
};

// new Node(rd, nb) — C/Node.java:246-271 with NodeBuilderWithRandomPosition (C/NodeBuilder.java:77-96)
void build_node(JavaRandom& rd, const Builder& b, NodeSoA& out) {
  int32_t r = rd.nextInt();
  if (b.city) {  // NodeBuilderWithCity.getRandomCityInfo / getPos (C/NodeBuilder.java:119-139)
    const CityTable& t = *b.city;
    const int32_t a = r == INT32_MIN ? r : (r < 0 ? -r : r);  // Math.abs(int)
    const int32_t rand = a % t.listSize;
    const float p = (float)rand / (float)t.listSize;
    int32_t c = -1;
    for (size_t i = 0; i < t.cum.size(); i++)
      if (p <= t.cum[i]) {
        c = (int32_t)i;
        break;
      }
    if (c < 0) {  // (the reference dies with a NullPointerException in getPos, C/NodeBuilder.java:121-124)
      out.bad = true;
      c = 0;
    }
    out.city.push_back(c);
    out.x.push_back(t.mercX[c]);
    out.y.push_back(t.mercY[c]);
  } else {
  int64_t rx = (int64_t)(r >> 16);
  if (rx < 0) rx = -rx;
  int64_t ry = (int64_t)(int32_t)((uint32_t)r << 16);
  if (ry < 0) ry = -ry;
  out.x.push_back((int32_t)(rx % 2000 + 1));
  out.y.push_back((int32_t)(ry % 1112 + 1));
  }
  double speed = 1.0;
  if (b.speedUniform) speed = rd.nextBoolean() ? (rd.nextInt(67) + 33) / 100.0 : (rd.nextInt(200) + 100) / 100.0;
  int32_t extra = 0;
  if (b.tor > 0) extra = rd.nextDouble() < b.tor ? 500 : 0;
  out.speed.push_back(speed);
  out.extra.push_back(extra);
  out.down.push_back(0);
}

struct Cleanup {
  wg_engine* e;
  bool keep = false;
  ~Cleanup() {
    if (e && !keep) wg_destroy(e);
  }
};

// Network.setNetworkLatency(RegistryNetworkLatencies.getByName(name)): the table-ised models by name before the nodes
// exist (as the reference's constructors do), a registered city model once the nodes — and their cities — are there
int32_t set_latency_early(wg_engine* e, const char* name) {
  if (name && city_latencies().count(name)) return WG_OK;
  return wg_set_latency_by_name(e, name);
}
int32_t set_latency_late(wg_engine* e, const char* name, const NodeSoA& nodes) {
  if (nodes.bad) {
    g_err = "NullPointerException: a node's draw is past the last city's cumulative probability (C/NodeBuilder.java:121-139)";
    return WG_ESTATE;
  }
  if (!name) return WG_OK;
  auto it = city_latencies().find(name);
  if (it == city_latencies().end()) return WG_OK;
  if (nodes.city.size() != nodes.x.size()) {
    g_err = std::string(name) + " needs a city node builder (IllegalStateException: default city location, C/NetworkLatency.java:175-178)";
    return WG_ESTATE;
  }
  const CityLatency& l = it->second;
  return wg_set_latency_city(e, l.mode, l.nCities, nodes.city.data(), l.tab.empty() ? nullptr : l.tab.data(),
                             l.ping.empty() ? nullptr : l.ping.data(), l.jit.empty() ? nullptr : l.jit.data());
}

#define CK(call)                                   \
  do {                                             \
    int32_t _rc = (call);                          \
    if (_rc != WG_OK) {                            \
      const char* _m = wg_last_error(e);           \
      if (_m && *_m) g_err = _m;                   \
      return _rc;                                  \
    }                                              \
  } while (0)

}  // namespace

extern "C" {

const char* wgh_last_error(void) { return g_err.c_str(); }

int32_t wgh_register_city_builder(const char* site, int32_t n, const float* cum, const int32_t* mercX, const int32_t* mercY,
                                  int32_t listSize) {
  const std::string s = site ? site : "";
  if ((s != "AWS" && s != "CITIES") || n <= 0 || !cum || !mercX || !mercY || listSize <= 0) {
    g_err = "wgh_register_city_builder: site is AWS or CITIES, the tables have n > 0 rows";
    return WG_EINVAL;
  }
  CityTable t;
  t.cum.assign(cum, cum + n);
  t.mercX.assign(mercX, mercX + n);
  t.mercY.assign(mercY, mercY + n);
  t.listSize = listSize;
  city_tables()[s] = t;
  return WG_OK;
}
int32_t wgh_register_city_latency(const char* latencyName, int32_t mode, int32_t nCities, const int32_t* tab, const float* ping,
                                  const double* jitter100) {
  if (!latencyName || nCities <= 0 || mode < 0 || mode > 2) {
    g_err = "wgh_register_city_latency: name / mode / nCities";
    return WG_EINVAL;
  }
  CityLatency l;
  l.mode = mode;
  l.nCities = nCities;
  const size_t cc = (size_t)nCities * nCities;
  if (tab) l.tab.assign(tab, tab + cc);
  if (ping) l.ping.assign(ping, ping + cc);
  if (jitter100) l.jit.assign(jitter100, jitter100 + 100);
  city_latencies()[latencyName] = l;
  return WG_OK;
}
double wgh_last_init_seconds(void) { return g_initSeconds; }
int32_t wgh_last_init_on_device(void) { return g_initOnDevice; }

int32_t wgh_jrandom_ints(int64_t seed, int32_t n, int32_t* out) {
  JavaRandom r(seed);
  for (int i = 0; i < n; i++) out[i] = r.nextInt();
  return WG_OK;
}
int32_t wgh_jrandom_skip_ints(int64_t seed, int32_t n, int32_t* out) {
  uint64_t s0 = wg::lcg_scramble(seed);
  for (int i = 0; i < n; i++) out[i] = (int32_t)(int64_t)(wg::lcg_skip(s0, (uint64_t)i + 1) >> 16);
  return WG_OK;
}
int32_t wgh_jrandom_bounded(int64_t seed, int32_t bound, int32_t n, int32_t* out) {
  if (bound <= 0) return WG_EINVAL;
  JavaRandom r(seed);
  for (int i = 0; i < n; i++) out[i] = r.nextInt(bound);
  return WG_OK;
}

int32_t wgh_pingpong_create(int32_t nodeCt, const char* nodeBuilderName, const char* latencyName, int64_t seed,
                            const wg_config* cfg, wg_engine** out) {
  if (!out) return WG_EINVAL;
  *out = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  Builder b;
  if (!parse_builder(nodeBuilderName, b)) return WG_EINVAL;
  if (nodeCt <= 0) {
    g_err = "nodeCt";
    return WG_EINVAL;
  }
  wg_engine* e = nullptr;
  int32_t rc = wg_create(cfg, &e);
  if (rc != WG_OK) {
    g_err = wg_last_error(nullptr);
    return rc;
  }
  Cleanup guard{e};
  CK(set_latency_early(e, latencyName));  // PingPong ctor :52-57
  JavaRandom rd(seed);
  NodeSoA nodes;
  for (int i = 0; i < nodeCt; i++) build_node(rd, b, nodes);  // init() :82-84
  CK(wg_add_nodes(e, nodeCt, nodes.x.data(), nodes.y.data(), nodes.extra.data(), nodes.down.data(), nullptr,
                  nodes.speed.data()));
  CK(set_latency_late(e, latencyName, nodes));
  CK(wg_rng_set_state(e, rd.s));
  CK(wg_protocol_load(e, WG_PROTO_PINGPONG, nullptr, nullptr));
  // network.sendAll(new Ping(), getNodeById(0))  :86 -> send(m, time + 1, from, allNodes)
  std::vector<int32_t> all(nodeCt);
  for (int i = 0; i < nodeCt; i++) all[i] = i;
  CK(wg_send(e, /*Ping*/ 0u, 0u, 1, 0, all.data(), nodeCt, 0));
  g_initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  guard.keep = true;
  *out = e;
  return WG_OK;
}

static int32_t handel_create(const wg_handel_params* pp, const uint8_t* badNodes, const char* nodeBuilderName,
                             const char* latencyName, int64_t seed, const wg_config* cfg, wg_engine** out, bool devicePeers);
// Handel.init() (P/Handel.java:957-1014). The emission lists (:991-1013) are built on the device where the engine can
// (unsharded, 256 .. 131 072 nodes: wg_handel_init_state.receptionRanks and .peers == NULL) together with the rank shuffles
// (:940-948, 966-989) and on the host otherwise — or after all, when
// the device met a rejected draw (WG_EHOSTINIT); WG_HOST_INIT=1 keeps everything on the host.
int32_t wgh_handel_create(const wg_handel_params* pp, const char* nodeBuilderName, const char* latencyName,
                          int64_t seed, const wg_config* cfg, wg_engine** out) {
  return wgh_handel_create_bad_nodes(pp, nullptr, nodeBuilderName, latencyName, seed, cfg, out);
}
// ... with HandelParameters.badNodes (P/Handel.java:51, 110, 139): the explicit set init() uses INSTEAD of
// Network.chooseBadNodes' draws (:960-964) — one byte per node, non-zero = down (and byzantine under an attack flag)
int32_t wgh_handel_create_bad_nodes(const wg_handel_params* pp, const uint8_t* badNodes, const char* nodeBuilderName,
                                    const char* latencyName, int64_t seed, const wg_config* cfg, wg_engine** out) {
  if (!out || !pp) return WG_EINVAL;
  const bool hostOnly = getenv("WG_HOST_INIT") && atoi(getenv("WG_HOST_INIT")) != 0;
  const bool sharded = cfg && cfg->nshards > 0;
  if (!hostOnly && !sharded && pp->nodeCount >= 256 && pp->nodeCount <= 131072) {
    const int32_t rc = handel_create(pp, badNodes, nodeBuilderName, latencyName, seed, cfg, out, true);
    if (rc != WG_EHOSTINIT) return rc;
  }
  return handel_create(pp, badNodes, nodeBuilderName, latencyName, seed, cfg, out, false);
}
static int32_t handel_create(const wg_handel_params* pp, const uint8_t* badNodes, const char* nodeBuilderName,
                             const char* latencyName, int64_t seed, const wg_config* cfg, wg_engine** out, bool devicePeers) {
  *out = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  wg_handel_params p = *pp;
  if (p.windowInitial == 0 && p.windowMinimum == 0 && p.windowMaximum == 0) {  // new WindowParameters() :155-157
    p.windowInitial = 16;
    p.windowMinimum = 1;
    p.windowMaximum = 128;
  }
  const int32_t N = p.nodeCount;
  // HandelParameters ctor checks (:113-125)
  if (N <= 0 || p.nodesDown >= N || p.nodesDown < 0 || p.threshold > N || p.nodesDown + p.threshold > N) {
    g_err = "nodeCount=" + std::to_string(N) + ", threshold=" + std::to_string(p.threshold);
    return WG_EINVAL;
  }
  if (__builtin_popcount((unsigned)N) != 1) {
    g_err = "We support only power of two nodes in this simulation";
    return WG_EINVAL;
  }
  if (p.disseminationPeriodMs <= 0 || p.pairingTime < 0) {
    g_err = "period/pairingTime";
    return WG_EINVAL;
  }
  Builder b;
  if (!parse_builder(nodeBuilderName, b)) return WG_EINVAL;

  wg_engine* e = nullptr;
  int32_t rc = wg_create(cfg, &e);
  if (rc != WG_OK) {
    g_err = wg_last_error(nullptr);
    return rc;
  }
  Cleanup guard{e};
  CK(set_latency_early(e, latencyName));  // Handel ctor :208-212
  JavaRandom rd(seed);

  // params.badNodes != null ? params.badNodes : Network.chooseBadNodes(rd, nodeCount, nodesDown)  (:960-964; C/Network.java:52-64)
  std::vector<uint8_t> bad(N, 0);
  if (badNodes)
    for (int i = 0; i < N; i++) bad[i] = badNodes[i] != 0;
  for (int setDown = 0; !badNodes && setDown < p.nodesDown;) {
    int32_t d = rd.nextInt(N);
    if (d != 1 && !bad[d]) {
      bad[d] = 1;
      setDown++;
    }
  }
  // node construction loop (:965-974)
  NodeSoA nodes;
  std::vector<int32_t> startAt(N), pairing(N);
  for (int i = 0; i < N; i++) {
    startAt[i] = p.desynchronizedStart == 0 ? 0 : rd.nextInt(p.desynchronizedStart);
    build_node(rd, b, nodes);
    double pt = p.pairingTime * nodes.speed[i];
    pairing[i] = (int32_t)(pt > 1.0 ? pt : 1.0);  // (int) Math.max(1, pairingTime * speedRatio)  :283
    nodes.down[i] = bad[i];
  }
  CK(wg_add_nodes(e, N, nodes.x.data(), nodes.y.data(), nodes.extra.data(), nodes.down.data(), nullptr,
                  nodes.speed.data()));
  CK(set_latency_late(e, latencyName, nodes));
  // registerPeriodicTask(dissemination, startAt + 1, period) for live nodes, in id order (:976-984).
  // The conditional task (checkSigs) is part of the resident protocol's state (minStartTime = startAt + 1).
  for (int i = 0; i < N; i++)
    if (!bad[i]) CK(wg_register_periodic_task(e, /*dissemination*/ 0u, startAt[i] + 1, p.disseminationPeriodMs, i));

  // setReceivingRanks (:940-948): one list, shuffled cumulatively once per node.
  std::vector<int32_t> ranks;
  const uint64_t rdBeforeRanks = rd.s;  // (the device path starts from here: wg_handel_init_state.receptionRanks == NULL)
  if (!devicePeers) {
    ranks.resize((size_t)N * N);
    std::vector<int32_t> expected(N);
    for (int i = 0; i < N; i++) expected[i] = i;
    for (int n = 0; n < N; n++) {
      for (int32_t i = N; i > 1; i--) std::swap(expected[i - 1], expected[rd.nextInt(i)]);  // Collections.shuffle
      int32_t* row = ranks.data() + (size_t)n * N;
      for (int i = 0; i < N; i++) row[expected[i]] = i;
    }
  }
  // Emission lists (:991-1013, buildEmissionList :510-522). For sender s and level l the receivers are
  // the sibling block of size 2^(l-1) in ascending id (expectedNodes :446-455); they are bucketed by
  // receiver.receptionRanks[s], buckets walked in rank order, a bucket with >1 entries shuffled with rd.
  const int L = 32 - __builtin_clz((unsigned)N);  // levels 0..log2(N)
  std::vector<int32_t> peers;
  if (!devicePeers) {
    peers.assign((size_t)N * (N > 1 ? N - 1 : 1), -1);
    // column s of `ranks` is read for every sender: transpose tile-wise first (cache friendly)
    std::vector<int32_t> ranksT((size_t)N * N);
    const int TB = 64;
    for (int i0 = 0; i0 < N; i0 += TB)
      for (int j0 = 0; j0 < N; j0 += TB)
        for (int i = i0; i < std::min(N, i0 + TB); i++)
          for (int j = j0; j < std::min(N, j0 + TB); j++) ranksT[(size_t)j * N + i] = ranks[(size_t)i * N + j];
    std::vector<uint32_t> keyA, keyB;  // (rank << 32 | receiver) would need 64 bits; ranks < N <= 2^28
    std::vector<uint64_t> a, tmp;
    for (int s = 0; s < N; s++) {
      if (bad[s]) continue;
      const int32_t* col = ranksT.data() + (size_t)s * N;  // col[r] = ranks[r][s]
      for (int l = 1; l < L; l++) {
        const int size = 1 << (l - 1);
        const int base = ((s >> (l - 1)) ^ 1) << (l - 1);
        a.resize(size);
        for (int k = 0; k < size; k++) a[k] = ((uint64_t)(uint32_t)col[base + k] << 32) | (uint32_t)(base + k);
        // stable sort by rank: LSD radix on the rank bits (ids ascending initially => stable)
        if (size <= 32) {
          for (int i = 1; i < size; i++) {
            uint64_t v = a[i];
            int k = i;
            while (k > 0 && (a[k - 1] >> 32) > (v >> 32)) {
              a[k] = a[k - 1];
              k--;
            }
            a[k] = v;
          }
        } else {
          tmp.resize(size);
          int bitsNeeded = 32 - __builtin_clz((unsigned)N);
          for (int shift = 0; shift < bitsNeeded; shift += 11) {
            uint32_t cnt[2049];
            memset(cnt, 0, sizeof(cnt));
            for (int k = 0; k < size; k++) cnt[((a[k] >> (32 + shift)) & 2047) + 1]++;
            for (int k = 0; k < 2048; k++) cnt[k + 1] += cnt[k];
            for (int k = 0; k < size; k++) tmp[cnt[(a[k] >> (32 + shift)) & 2047]++] = a[k];
            a.swap(tmp);
          }
        }
        int32_t* dst = peers.data() + (size_t)s * (N - 1) + (size - 1);
        for (int i = 0; i < size;) {
          int j = i;
          while (j < size && (a[j] >> 32) == (a[i] >> 32)) j++;
          int g = j - i;
          if (g > 1)
            for (int32_t k = g; k > 1; k--) std::swap(a[i + k - 1], a[i + rd.nextInt(k)]);
          for (int k = i; k < j; k++) dst[k] = (int32_t)(uint32_t)a[k];
          i = j;
        }
      }
    }
  }
  CK(wg_rng_set_state(e, devicePeers ? rdBeforeRanks : rd.s));
  wg_handel_init_state st;
  st.startAt = startAt.data();
  st.nodePairingTime = pairing.data();
  // (NULL: the engine shuffles the ranks and builds the lists on the device, and advances its rd by their draws)
  st.receptionRanks = devicePeers ? nullptr : ranks.data();
  st.peers = devicePeers ? nullptr : peers.data();
  const bool verbose = getenv("WG_INIT_VERBOSE") && atoi(getenv("WG_INIT_VERBOSE"));
  if (verbose)
    fprintf(stderr, "[wittgpu] handel init(): host part before the protocol load %.3f s\n",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  {
    const int32_t rc = wg_protocol_load(e, WG_PROTO_HANDEL, &p, &st);
    if (rc == WG_EHOSTINIT) {  // (the engine is destroyed by the guard; the caller starts over with host-built lists)
      if (verbose) fprintf(stderr, "[wittgpu] handel init(): the device handed init() back to the host: %s\n", wg_last_error(e));
      return rc;
    }
    if (rc != WG_OK) {
      g_err = wg_last_error(e);
      return rc;
    }
  }
  g_initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (verbose) fprintf(stderr, "[wittgpu] handel init(): %.3f s in all\n", g_initSeconds);
  g_initOnDevice = devicePeers ? 1 : 0;
  guard.keep = true;
  *out = e;
  return WG_OK;
}

int32_t wgh_sanfermin_create(const wg_sanfermin_params* pp, const char* nodeBuilderName, const char* latencyName,
                             int64_t seed, const wg_config* cfg, wg_engine** out) {
  if (!out || !pp) return WG_EINVAL;
  *out = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  const wg_sanfermin_params p = *pp;
  const int32_t N = p.nodeCount;
  if (N < 2 || __builtin_popcount((unsigned)N) != 1) {
    g_err = "San Fermin needs a power-of-two nodeCount (toBinaryID, P/SanFerminHelper.java:158-171)";
    return WG_EUNSUPPORTED;
  }
  Builder b;
  if (!parse_builder(nodeBuilderName, b)) return WG_EINVAL;
  wg_engine* e = nullptr;
  int32_t rc = wg_create(cfg, &e);
  if (rc != WG_OK) {
    g_err = wg_last_error(nullptr);
    return rc;
  }
  Cleanup guard{e};
  CK(set_latency_early(e, latencyName));  // ctor :113-120
  JavaRandom rd(0);                            // the ctor builds the nodes from the fresh Network's rd (:126-131) ...
  NodeSoA nodes;
  for (int i = 0; i < N; i++) build_node(rd, b, nodes);
  CK(wg_add_nodes(e, N, nodes.x.data(), nodes.y.data(), nodes.extra.data(), nodes.down.data(), nullptr,
                  nodes.speed.data()));
  CK(set_latency_late(e, latencyName, nodes));
  CK(wg_rng_set_seed(e, seed));                // ... rd.setSeed(i) comes after it (C/RunMultipleTimes.java:44-48)
  CK(wg_protocol_load(e, WG_PROTO_SANFERMIN, &p, nullptr));
  for (int i = 0; i < N; i++) CK(wg_register_task(e, /*goNextLevel*/ 0u, 0u, 1, i));  // init() :139-141
  g_initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  guard.keep = true;
  *out = e;
  return WG_OK;
}

int32_t wgh_casper_create(const wg_casper_params* pp, const char* nodeBuilderName, const char* latencyName, int64_t seed,
                          const wg_config* cfg, wg_engine** out) {
  if (!out || !pp) return WG_EINVAL;
  *out = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  const wg_casper_params p = *pp;
  if (p.cycleLength <= 0 || p.blockProducersCount <= 0 || p.attestersPerRound <= 0) {
    g_err = "Casper IMD parameters";
    return WG_EINVAL;
  }
  const int32_t attesters = p.cycleLength * p.attestersPerRound, N = 1 + p.blockProducersCount + attesters;
  Builder b;
  if (!parse_builder(nodeBuilderName, b)) return WG_EINVAL;
  wg_engine* e = nullptr;
  int32_t rc = wg_create(cfg, &e);
  if (rc != WG_OK) {
    g_err = wg_last_error(nullptr);
    return rc;
  }
  Cleanup guard{e};
  CK(set_latency_early(e, latencyName));  // ctor :80-87
  NodeSoA nodes;
  JavaRandom rd(0);
  build_node(rd, b, nodes);                    // network.addObserver(new CasperNode(false, genesis) {}) — from new Random(0)
  rd.setSeed(seed);                            // rd.setSeed(i) on the copy  C/RunMultipleTimes.java:44-48
  for (int i = 1; i < N; i++) build_node(rd, b, nodes);  // init(): the byzantine producer, the producers, the attesters
  CK(wg_add_nodes(e, N, nodes.x.data(), nodes.y.data(), nodes.extra.data(), nodes.down.data(), nullptr,
                  nodes.speed.data()));
  CK(set_latency_late(e, latencyName, nodes));
  CK(wg_rng_set_state(e, rd.s));
  CK(wg_protocol_load(e, WG_PROTO_CASPER, &p, nullptr));
  const int32_t SD = 8000;  // SLOT_DURATION
  // registerPeriodicTask calls of init(badNode) :481-509, in its order (= push order of the task envelopes)
  CK(wg_register_periodic_task(e, /*ByzBlockProducerWF*/ 2u, SD + p.byzDelay, SD * p.blockProducersCount, 1));
  for (int i = 1; i < p.blockProducersCount; i++)
    CK(wg_register_periodic_task(e, /*BlockProducer*/ 0u, SD * (i + 1), SD * p.blockProducersCount, 1 + i));
  for (int i = 0; i < attesters; i++)
    CK(wg_register_periodic_task(e, /*Attester*/ 1u, SD * (1 + i % p.cycleLength) + 4000, SD * p.cycleLength,
                                 1 + p.blockProducersCount + i));
  g_initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  guard.keep = true;
  *out = e;
  return WG_OK;
}

int32_t wgh_p2pflood_create(const wg_p2pflood_params* pp, const char* nodeBuilderName, const char* latencyName,
                            int64_t seed, const wg_config* cfg, wg_engine** out) {
  if (!out || !pp) return WG_EINVAL;
  *out = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  const wg_p2pflood_params p = *pp;
  const int32_t N = p.nodeCount;
  if (N <= 0 || p.deadNodeCount < 0 || p.deadNodeCount >= N || p.msgCount < 1) {
    g_err = "P2PFlood parameters";
    return WG_EINVAL;
  }
  if (p.peersCount >= N) {  // P2PNetwork.setPeers :28-35
    g_err = "Wrong configuration: #nodes=" + std::to_string(N) + ", connection target=" + std::to_string(p.peersCount);
    return WG_EINVAL;
  }
  Builder b;
  if (!parse_builder(nodeBuilderName, b)) return WG_EINVAL;
  wg_engine* e = nullptr;
  int32_t rc = wg_create(cfg, &e);
  if (rc != WG_OK) {
    g_err = wg_last_error(nullptr);
    return rc;
  }
  Cleanup guard{e};
  CK(set_latency_early(e, latencyName));  // ctor :88-94
  JavaRandom rd(seed);
  NodeSoA nodes;
  for (int i = 0; i < N; i++) {  // init(): new P2PFloodNode(nb, i < deadNodeCount) :122-124 (its stop() when down :27-31)
    build_node(rd, b, nodes);
    if (i < p.deadNodeCount) nodes.down[i] = 1;
  }
  // P2PNetwork.setPeers with minimum == true (:27-56): nodes in a Collections.shuffle'd order, links drawn until
  // every node has connectionCount peers; createLink :72-93
  std::vector<std::vector<int32_t>> peers(N);
  std::set<int64_t> links;
  auto createLink = [&](int32_t a, int32_t c) {
    if (a == c) return;
    const int64_t link = ((int64_t)std::min(a, c) << 32) + (int64_t)std::max(a, c);
    if (!links.insert(link).second) return;
    peers[a].push_back(c);
    peers[c].push_back(a);
  };
  std::vector<int32_t> an(N);
  for (int i = 0; i < N; i++) an[i] = i;
  for (int32_t k = N; k > 1; k--) std::swap(an[k - 1], an[rd.nextInt(k)]);
  for (int32_t n : an)
    while ((int32_t)peers[n].size() < p.peersCount) createLink(n, rd.nextInt(N));
  int32_t maxPeers = 1;
  for (auto& v : peers) maxPeers = std::max(maxPeers, (int32_t)v.size());
  if (maxPeers > 64) {
    g_err = "a node has more than 64 peers (device multi-destination sends hold <= 64 ids)";
    return WG_EUNSUPPORTED;
  }
  CK(wg_add_nodes(e, N, nodes.x.data(), nodes.y.data(), nodes.extra.data(), nodes.down.data(), nodes.down.data(),
                  nodes.speed.data()));
  CK(set_latency_late(e, latencyName, nodes));
  std::vector<int32_t> flat((size_t)N * maxPeers, -1), cnt(N);
  for (int i = 0; i < N; i++) {
    cnt[i] = (int32_t)peers[i].size();
    for (size_t k = 0; k < peers[i].size(); k++) flat[(size_t)i * maxPeers + k] = peers[i][k];
  }
  // the senders loop :126-139 interleaves three uses of rd: the sender's id, sendPeers' shuffle of its peers
  // (C/P2PNetwork.java:127-132) and the seed Network.send draws (:430). The protocol has to be loaded (with the senders)
  // before the first wg_send, so the loop runs twice: on a copy of rd to learn the senders, then for real.
  std::vector<int32_t> senders;
  {
    JavaRandom probe = rd;
    std::set<int32_t> seen;
    while ((int32_t)senders.size() < p.msgCount) {
      const int32_t nodeId = probe.nextInt(N);
      if (nodes.down[nodeId] || !seen.insert(nodeId).second) continue;
      for (int32_t k = (int32_t)peers[nodeId].size(); k > 1; k--) (void)probe.nextInt(k);
      (void)probe.nextInt();
      senders.push_back(nodeId);
    }
  }
  wg_p2pflood_init_state st{flat.data(), cnt.data(), maxPeers, senders.data()};
  CK(wg_rng_set_state(e, rd.s));
  CK(wg_protocol_load(e, WG_PROTO_P2PFLOOD, &p, &st));
  {
    std::set<int32_t> seen;
    int32_t k = 0;
    while (k < p.msgCount) {
      const int32_t nodeId = rd.nextInt(N);
      if (nodes.down[nodeId] || !seen.insert(nodeId).second) continue;
      std::vector<int32_t> dest(peers[nodeId]);
      for (int32_t q = (int32_t)dest.size(); q > 1; q--) std::swap(dest[q - 1], dest[rd.nextInt(q)]);
      CK(wg_rng_set_state(e, rd.s));  // the engine's rd is the Java rd: wg_send draws the seed from it
      CK(wg_send(e, (uint32_t)k, 0u, 1 + p.delayBeforeResent, nodeId, dest.data(), (int32_t)dest.size(), p.delayBetweenSends));
      (void)rd.nextInt();
      k++;
    }
    CK(wg_rng_set_state(e, rd.s));
  }
  g_initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  guard.keep = true;
  *out = e;
  return WG_OK;
}

int32_t wgh_gsf_create(const wg_gsf_params* pp, const char* nodeBuilderName, const char* latencyName, int64_t seed,
                       const wg_config* cfg, wg_engine** out) {
  if (!out || !pp) return WG_EINVAL;
  *out = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  const wg_gsf_params p = *pp;
  const int32_t N = p.nodeCount;
  // GSFSignatureParameters ctor checks (P/GSFSignature.java:69-74)
  if (N <= 0 || p.nodesDown >= N || p.nodesDown < 0 || p.threshold > N || p.nodesDown + p.threshold > N) {
    g_err = "nodeCount=" + std::to_string(N) + ", threshold=" + std::to_string(p.threshold);
    return WG_EINVAL;
  }
  if (p.periodDurationMs <= 0 || p.pairingTime < 0) {
    g_err = "period/pairingTime";
    return WG_EINVAL;
  }
  if (__builtin_popcount((unsigned)N) != 1 || N < 2) {
    g_err = "the resident GSFSignature needs a power-of-two nodeCount";
    return WG_EUNSUPPORTED;
  }
  Builder b;
  if (!parse_builder(nodeBuilderName, b)) return WG_EINVAL;
  wg_engine* e = nullptr;
  int32_t rc = wg_create(cfg, &e);
  if (rc != WG_OK) {
    g_err = wg_last_error(nullptr);
    return rc;
  }
  Cleanup guard{e};
  CK(set_latency_early(e, latencyName));  // GSFSignature ctor :109-114
  JavaRandom rd(seed);
  // init() :611-635 — node ctors, then the nodesDown loop, then levels + tasks of the live nodes in id order
  NodeSoA nodes;
  std::vector<int32_t> pairing(N);
  for (int i = 0; i < N; i++) {
    build_node(rd, b, nodes);
    double pt = p.pairingTime * nodes.speed[i];
    pairing[i] = (int32_t)(pt > 1.0 ? pt : 1.0);  // (int) Math.max(1, pairingTime * speedRatio)  :170
  }
  for (int setDown = 0; setDown < p.nodesDown;) {
    int32_t d = rd.nextInt(N);
    if (!nodes.down[d] && d != 1) {
      nodes.down[d] = 1;
      setDown++;
    }
  }
  CK(wg_add_nodes(e, N, nodes.x.data(), nodes.y.data(), nodes.extra.data(), nodes.down.data(), nullptr,
                  nodes.speed.data()));
  CK(set_latency_late(e, latencyName, nodes));
  const int L = 32 - __builtin_clz((unsigned)N);  // levels 0..log2(N)
  std::vector<int32_t> peers((size_t)N * (N - 1), -1);
  for (int i = 0; i < N; i++) {
    if (nodes.down[i]) continue;
    // initLevel() :182-192 -> SFLevel(previous, allPreviousNodes) :273-283: peers = randomSubset(waitedSigs):
    // the sibling block in ascending id, Collections.shuffle(res, rd) (:462-476)
    for (int l = 1; l < L; l++) {
      const int size = 1 << (l - 1);
      const int base = ((i >> (l - 1)) ^ 1) << (l - 1);
      int32_t* dst = peers.data() + (size_t)i * (N - 1) + (size - 1);
      for (int k = 0; k < size; k++) dst[k] = base + k;
      for (int32_t k = size; k > 1; k--) std::swap(dst[k - 1], dst[rd.nextInt(k)]);
    }
    // registerPeriodicTask(n::doCycle, 1, periodDurationMs, n) :630; the conditional task (checkSigs, :631-632)
    // is part of the resident protocol's state (minStartTime = 1)
    CK(wg_register_periodic_task(e, /*doCycle*/ 0u, 1, p.periodDurationMs, i));
  }
  CK(wg_rng_set_state(e, rd.s));
  wg_gsf_init_state st;
  st.nodePairingTime = pairing.data();
  st.peers = peers.data();
  CK(wg_protocol_load(e, WG_PROTO_GSF, &p, &st));
  g_initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  guard.keep = true;
  *out = e;
  return WG_OK;
}

}  // extern "C"
