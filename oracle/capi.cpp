// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Flat C API over the oracle so tests/, smoke() and bench.py's cpu_baseline leg can drive it via
// ctypes. Every function returns 0 on success, negative on a restated Java exception
// (-1 IllegalArgumentException, -2 IllegalStateException, -3 other); orc_last_error() has the text.
#include <cstring>
#include <chrono>
#include "casper.hpp"
#include "fuzz.hpp"
#include "gsf.hpp"
#include "handel.hpp"
#include "p2pflood.hpp"
#include "optimistic_p2p.hpp"
#include "dfinity.hpp"
#include "p2phandel.hpp"
#include "paxos.hpp"
#include "slush.hpp"
#include "pingpong.hpp"
#include "sanfermin.hpp"
#include "sanfermin_cappos.hpp"

using namespace orc;

static thread_local std::string g_err;

#define ORC_TRY try {
#define ORC_CATCH                                  \
  }                                                \
  catch (const IllegalArgumentException& e) {      \
    g_err = e.what();                              \
    return -1;                                     \
  }                                                \
  catch (const IllegalStateException& e) {         \
    g_err = e.what();                              \
    return -2;                                     \
  }                                                \
  catch (const std::exception& e) {                \
    g_err = e.what();                              \
    return -3;                                     \
  }                                                \
  return 0;

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- JDK kit
int orc_jrandom_ints(int64_t seed, int n, int32_t* out) {
  JRandom r(seed);
  for (int i = 0; i < n; i++) out[i] = r.nextInt();
  return 0;
}
int orc_jrandom_bounded(int64_t seed, int32_t bound, int n, int32_t* out) {
  ORC_TRY JRandom r(seed);
  for (int i = 0; i < n; i++) out[i] = r.nextInt(bound);
  ORC_CATCH
}
int orc_jrandom_doubles(int64_t seed, int n, double* out) {
  JRandom r(seed);
  for (int i = 0; i < n; i++) out[i] = r.nextDouble();
  return 0;
}
int orc_jrandom_booleans(int64_t seed, int n, uint8_t* out) {
  JRandom r(seed);
  for (int i = 0; i < n; i++) out[i] = r.nextBoolean();
  return 0;
}
int orc_jshuffle_iota(int64_t seed, int n, int rounds, int32_t* out) {
  JRandom r(seed);
  std::vector<int32_t> v(n);
  for (int i = 0; i < n; i++) v[i] = i;
  for (int k = 0; k < rounds; k++) jshuffle(v, r);
  memcpy(out, v.data(), sizeof(int32_t) * n);
  return 0;
}
int32_t orc_pseudo_random(int32_t nodeId, int32_t seed) { return Network::getPseudoRandom(nodeId, seed); }
// NetworkLatencyByDistanceWJitter.getExtendedLatency as a (dist, delta) table entry
int32_t orc_latency_bydistance(int32_t dist, int32_t delta) {
  NetworkLatencyByDistanceWJitter nl;
  double raw = nl.getFixedLatency(dist) + nl.getJitter(delta);
  return (int)(raw / 2);
}
int32_t orc_max_dist() { return Node::MAX_DIST(); }
// NetworkLatency.getLatency(from, to, delta) of a registry model for two ad-hoc nodes
struct ProbeNB : NodeBuilder {
  int px, py;
  int getX(jint) override { return px; }
  int getY(jint) override { return py; }
};
int orc_latency(const char* name, int32_t x1, int32_t y1, int32_t e1, int32_t x2, int32_t y2, int32_t e2, int32_t same,
                int32_t delta, int32_t* out) {
  ORC_TRY auto nl = networkLatencyByName(name ? name : "");
  JRandom rd(0);
  ProbeNB nb;
  nb.px = x1;
  nb.py = y1;
  Node a(rd, nb);
  nb.px = x2;
  nb.py = y2;
  Node b(rd, nb);
  a.extraLatency = e1;
  b.extraLatency = e2;
  *out = nl->getLatency(a, same ? a : b, delta);
  ORC_CATCH
}
// ---- city data, city node builders, city latency models (geo.hpp)
// text: lines "D\t<dir>" (CSVLatencyReader's list, in order), "P\t<dir index>\t<other city>\t<average ms>",
// "C\t<name>\t<Lat>\t<Long>\t<Population>" (cities.csv rows, in order) — tests/golden/city_data.json flattened
int orc_city_data_load(const char* text) {
  ORC_TRY CityData d;
  std::string line;
  auto fields = [](const std::string& l) {
    std::vector<std::string> f;
    size_t a = 0;
    for (;;) {
      size_t b = l.find('\t', a);
      f.push_back(l.substr(a, b == std::string::npos ? std::string::npos : b - a));
      if (b == std::string::npos) break;
      a = b + 1;
    }
    return f;
  };
  for (const char* p = text;; p++) {
    if (*p == '\n' || *p == 0) {
      if (!line.empty()) {
        auto f = fields(line);
        if (f[0] == "D" && f.size() == 2) {
          d.dirs.push_back(f[1]);
          d.ping.emplace_back();
        } else if (f[0] == "P" && f.size() == 4) {
          d.ping.at((size_t)atoi(f[1].c_str()))[f[2]] = f[3];
        } else if (f[0] == "C" && f.size() == 5) {
          d.cityRows.push_back({f[1], f[2], f[3], f[4]});
        } else {
          throw IllegalArgumentException("bad city data line: " + line);
        }
      }
      line.clear();
      if (*p == 0) break;
    } else {
      line.push_back(*p);
    }
  }
  cityData() = d;
  ORC_CATCH
}
// the table a NodeBuilderWithCity works from, in its citiesInfo.entrySet() order: kind 0 = AWS, 1 = CITIES.
// names: '\n'-joined into buf (cap bytes); returns the count in *n, cities.size() of the builder's LIST in *listSize
int orc_city_builder_table(int32_t kind, int32_t cap, char* buf, int32_t* mercX, int32_t* mercY, float* cum, int32_t* n,
                           int32_t* listSize) {
  ORC_TRY std::unique_ptr<NodeBuilder> nb = nodeBuilderByName(kind == 0 ? "AWS_SPEED=CONSTANT_TOR=0.00" : "CITIES_SPEED=CONSTANT_TOR=0.00");
  const CityChooser& c = static_cast<NodeBuilderWithCity*>(nb.get())->chooser;
  std::string names;
  for (size_t i = 0; i < c.name.size(); i++) {
    names += c.name[i];
    names += '\n';
    mercX[i] = c.info[i].mercX;
    mercY[i] = c.info[i].mercY;
    cum[i] = c.info[i].cumulativeProbability;
  }
  if ((int32_t)names.size() + 1 > cap) throw IllegalArgumentException("buffer too small");
  memcpy(buf, names.c_str(), names.size() + 1);
  *n = (int32_t)c.name.size();
  *listSize = c.listSize;
  ORC_CATCH
}
int orc_city_choose(int32_t kind, int32_t rdInt, int32_t* idx) {
  ORC_TRY static std::unique_ptr<NodeBuilder> nbs[2];
  if (!nbs[kind & 1]) nbs[kind & 1] = nodeBuilderByName(kind == 0 ? "AWS_SPEED=CONSTANT_TOR=0.00" : "CITIES_SPEED=CONSTANT_TOR=0.00");
  *idx = static_cast<NodeBuilderWithCity*>(nbs[kind & 1].get())->chooser.choose(rdInt);
  ORC_CATCH
}
// NetworkLatency.getLatency(from, to, delta) of a named model between nodes placed in two named cities
struct OrcLatencyModel {
  std::unique_ptr<NetworkLatency> nl;
};
int orc_latency_model_create(const char* name, void** out) {
  ORC_TRY* out = new OrcLatencyModel{networkLatencyByName(name)};
  ORC_CATCH
}
void orc_latency_model_destroy(void* h) { delete (OrcLatencyModel*)h; }
int orc_latency_model_city(void* h, const char* cityFrom, const char* cityTo, int32_t e1, int32_t e2, int32_t same,
                           int32_t delta, int32_t* out) {
  ORC_TRY JRandom rd(0);
  ProbeNB nb;
  nb.px = nb.py = 1;
  Node a(rd, nb), b(rd, nb);
  a.cityName = cityFrom;
  b.cityName = cityTo;
  a.extraLatency = e1;
  b.extraLatency = e2;
  *out = ((OrcLatencyModel*)h)->nl->getLatency(a, same ? a : b, delta);
  ORC_CATCH
}
int orc_node_xy(int32_t rdInt, int32_t* x, int32_t* y) {
  NodeBuilderWithRandomPosition nb;
  *x = nb.getX(rdInt);
  *y = nb.getY(rdInt);
  return 0;
}

// ---- PingPong
struct OrcPingPong {
  std::unique_ptr<PingPong> p;
};
int orc_pingpong_create(int nodeCt, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY PingPong::PingPongParameters pr;
  pr.nodeCt = nodeCt;
  pr.nodeBuilderName = nb ? nb : "";
  pr.networkLatencyName = nl ? nl : "";
  auto* h = new OrcPingPong();
  h->p = std::make_unique<PingPong>(pr);
  h->p->network().rd.setSeed(seed);
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_pingpong_destroy(void* h) { delete (OrcPingPong*)h; }
int orc_pingpong_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcPingPong*)h)->p->network().runMs(ms);
  ORC_CATCH
}
// fields: 0 pong, 1 msgReceived, 2 msgSent, 3 bytesSent, 4 bytesReceived, 5 x, 6 y, 7 down
int orc_pingpong_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcPingPong*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.pong; break;
      case 1: v = n.msgReceived; break;
      case 2: v = n.msgSent; break;
      case 3: v = n.bytesSent; break;
      case 4: v = n.bytesReceived; break;
      case 5: v = n.x; break;
      case 6: v = n.y; break;
      case 7: v = n.down; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_pingpong_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered) {
  auto& p = *((OrcPingPong*)h)->p;
  *time = p.network().time;
  *queueSize = p.network().msgs.size();
  *rngState = p.network().rd.rawState();
  *delivered = p.network().statDelivered;
  return 0;
}

// ---- Handel
struct OrcHandel {
  std::unique_ptr<Handel> p;
  double initSeconds = 0;
};
// iparams: nodeCount, threshold, pairingTime, levelWaitTime, extraCycle, disseminationPeriodMs,
//          fastPath, nodesDown, desynchronizedStart   (P/Handel.java:97-111 ctor order, ints only)
int orc_handel_create_byz(const int32_t* ip, const char* nb, const char* nl, int64_t seed, int byzSuicide, int hiddenByz,
                          void** out);
int orc_handel_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  return orc_handel_create_byz(ip, nb, nl, seed, 0, 0, out);
}
// ... with the attack scenarios of the parameters (P/Handel.java:108-109): byzantineSuicide, hiddenByzantine
int orc_handel_create_bad(const int32_t* ip, const char* nb, const char* nl, int64_t seed, int byzSuicide, int hiddenByz,
                          const uint8_t* bad, void** out);
int orc_handel_create_byz(const int32_t* ip, const char* nb, const char* nl, int64_t seed, int byzSuicide, int hiddenByz,
                          void** out) {
  return orc_handel_create_bad(ip, nb, nl, seed, byzSuicide, hiddenByz, nullptr, out);
}
// ... and HandelParameters.badNodes (P/Handel.java:51,110,139): bad[nodeCount], non-zero = the bit is set; NULL = null
int orc_handel_create_bad(const int32_t* ip, const char* nb, const char* nl, int64_t seed, int byzSuicide, int hiddenByz,
                          const uint8_t* bad, void** out) {
  ORC_TRY BitSet badSet;
  if (bad)
    for (int i = 0; i < ip[0]; i++)
      if (bad[i]) badSet.set(i);
  Handel::HandelParameters pr(ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], ip[6], ip[7], nb ? nb : "",
                              nl ? nl : "", ip[8], byzSuicide != 0, hiddenByz != 0, bad ? &badSet : nullptr);
  auto* h = new OrcHandel();
  h->p = std::make_unique<Handel>(pr);
  h->p->network().rd.setSeed(seed);
  auto t0 = std::chrono::steady_clock::now();
  h->p->init();
  h->initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *out = h;
  ORC_CATCH
}
void orc_handel_destroy(void* h) { delete (OrcHandel*)h; }
int orc_handel_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcHandel*)h)->p->network().runMs(ms);
  ORC_CATCH
}
int orc_handel_cont_if(void* h) { return ((OrcHandel*)h)->p->contIf(); }
int orc_handel_levels(void* h) { return (int)((OrcHandel*)h)->p->node(0)->levels.size(); }
double orc_handel_init_seconds(void* h) { return ((OrcHandel*)h)->initSeconds; }
// per-node scalar fields: 0 doneAt 1 msgReceived 2 msgSent 3 bytesSent 4 bytesReceived 5 sigsChecked
// 6 sigQueueSize 7 msgFiltered 8 currWindowSize 9 addedCycle 10 down 11 x 12 y 13 startAt
// 14 nodePairingTime 15 extraLatency
int orc_handel_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcHandel*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.doneAt; break;
      case 1: v = n.msgReceived; break;
      case 2: v = n.msgSent; break;
      case 3: v = n.bytesSent; break;
      case 4: v = n.bytesReceived; break;
      case 5: v = n.sigsChecked; break;
      case 6: v = n.sigQueueSize; break;
      case 7: v = n.msgFiltered; break;
      case 8: v = n.currWindowSize; break;
      case 9: v = n.addedCycle; break;
      case 10: v = n.down; break;
      case 11: v = n.x; break;
      case 12: v = n.y; break;
      case 13: v = n.startAt; break;
      case 14: v = n.nodePairingTime; break;
      case 15: v = n.extraLatency; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
// per (node, level) ints, row-major [node][level]: 0 posInLevel 1 outgoingFinished 2 toVerifyAgg.size 3 suicideBizAfter
int orc_handel_read_level(void* h, int field, int32_t* out) {
  ORC_TRY auto& p = *((OrcHandel*)h)->p;
  int L = (int)p.node(0)->levels.size();
  for (size_t i = 0; i < p.nodes.size(); i++)
    for (int l = 0; l < L; l++) {
      auto& n = *p.nodes[i];
      if (n.levels.empty()) {
        out[i * L + l] = 0;
        continue;
      }
      auto& lv = *n.levels[l];
      int v = 0;
      switch (field) {
        case 0: v = lv.posInLevel; break;
        case 1: v = lv.outgoingFinished; break;
        case 2: v = (int)lv.toVerifyAgg.size(); break;
        case 3: v = lv.suicideBizAfter; break;
        default: throw IllegalArgumentException("field");
      }
      out[i * L + l] = v;
    }
  ORC_CATCH
}
// Bitsets in "natural layout": one nodeCount-bit row per node (words = nodeCount/64 rounded up),
// bit j = node id j, union over levels (level blocks are disjoint, P/Handel.java:671-684).
// which: 0 totalIncoming 1 lastAggVerified 2 verifiedIndSignatures 3 toVerifyInd 4 finishedPeers
//        5 totalOutgoing of the LAST level only 6 waitedSigs 7 the node's blacklist (:287)
int orc_handel_read_bits(void* h, int which, uint64_t* out) {
  ORC_TRY auto& p = *((OrcHandel*)h)->p;
  int N = p.params.nodeCount;
  int W = (N + 63) / 64;
  memset(out, 0, sizeof(uint64_t) * (size_t)W * p.nodes.size());
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    uint64_t* row = out + i * W;
    if (which == 7) {
      for (int w = 0; w < W; w++) row[w] = n.blacklist.wordAt(w);
      continue;
    }
    for (size_t l = 0; l < n.levels.size(); l++) {
      auto& lv = *n.levels[l];
      const BitSet* b = nullptr;
      switch (which) {
        case 0: b = &lv.totalIncoming; break;
        case 1: b = &lv.lastAggVerified; break;
        case 2: b = &lv.verifiedIndSignatures; break;
        case 3: b = &lv.toVerifyInd; break;
        case 4: b = &lv.finishedPeers; break;
        case 5: b = (l + 1 == n.levels.size()) ? &lv.totalOutgoing : nullptr; break;
        case 6: b = &lv.waitedSigs; break;
        default: throw IllegalArgumentException("which");
      }
      if (b)
        for (int w = 0; w < W; w++) row[w] |= b->wordAt(w);
    }
  }
  ORC_CATCH
}
int orc_handel_read_ranks(void* h, int node, int32_t* out) {
  auto& n = *((OrcHandel*)h)->p->node(node);
  memcpy(out, n.receptionRanks.data(), sizeof(int32_t) * n.receptionRanks.size());
  return 0;
}
// peers of (node, level) -> ids; returns count via *cnt
int orc_handel_read_peers(void* h, int node, int level, int32_t* out, int32_t* cnt) {
  auto& n = *((OrcHandel*)h)->p->node(node);
  if (n.levels.empty()) {
    *cnt = 0;
    return 0;
  }
  auto& lv = *n.levels[level];
  *cnt = (int)lv.peers.size();
  for (size_t i = 0; i < lv.peers.size(); i++) out[i] = lv.peers[i]->nodeId;
  return 0;
}
int orc_handel_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered,
                    uint64_t* tasks) {
  auto& p = *((OrcHandel*)h)->p;
  *time = p.network().time;
  if (queueSize) *queueSize = p.network().msgs.size();
  *rngState = p.network().rd.rawState();
  *delivered = p.network().statDelivered;
  *tasks = p.network().statTasks;
  return 0;
}
int orc_handel_stats(void* h, uint64_t* deliveredByLevel32, int32_t* queueMax32) {
  auto& p = *((OrcHandel*)h)->p;
  memcpy(deliveredByLevel32, p.statDeliveredByLevel, sizeof(uint64_t) * 32);
  memcpy(queueMax32, p.statQueueMax, sizeof(int32_t) * 32);
  return 0;
}

// ---- GSFSignature
struct OrcGsf {
  std::unique_ptr<GSFSignature> p;
  double initSeconds = 0;
};
// iparams: nodeCount, threshold, pairingTime, timeoutPerLevelMs, periodDurationMs, acceleratedCallsCount,
//          nodesDown   (P/GSFSignature.java:59-68 ctor order, ints only)
int orc_gsf_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY GSFSignature::GSFSignatureParameters pr(ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], ip[6], nb ? nb : "",
                                                  nl ? nl : "");
  auto* h = new OrcGsf();
  h->p = std::make_unique<GSFSignature>(pr);
  h->p->network().rd.setSeed(seed);
  auto t0 = std::chrono::steady_clock::now();
  h->p->init();
  h->initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *out = h;
  ORC_CATCH
}
void orc_gsf_destroy(void* h) { delete (OrcGsf*)h; }
int orc_gsf_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcGsf*)h)->p->network().runMs(ms);
  ORC_CATCH
}
void orc_gsf_set_copy_on_delivery(void* h, int on) { ((OrcGsf*)h)->p->copyOnDelivery = on != 0; }
int orc_gsf_cont_if(void* h) { return ((OrcGsf*)h)->p->contIf(); }
int orc_gsf_levels(void* h) {
  auto& p = *((OrcGsf*)h)->p;
  for (auto& n : p.nodes)
    if (!n->levels.empty()) return (int)n->levels.size();
  return 0;
}
// per-node scalar fields: 0 doneAt 1 msgReceived 2 msgSent 3 bytesSent 4 bytesReceived 5 sigChecked
// 6 sigQueueSize 7 toVerify.size 8 verifiedSignatures.cardinality 10 down 11 x 12 y 14 nodePairingTime
// 15 extraLatency
int orc_gsf_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcGsf*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.doneAt; break;
      case 1: v = n.msgReceived; break;
      case 2: v = n.msgSent; break;
      case 3: v = n.bytesSent; break;
      case 4: v = n.bytesReceived; break;
      case 5: v = n.sigChecked; break;
      case 6: v = n.sigQueueSize; break;
      case 7: v = (int64_t)n.toVerify.size(); break;
      case 8: v = n.verifiedSignatures.cardinality(); break;
      case 10: v = n.down; break;
      case 11: v = n.x; break;
      case 12: v = n.y; break;
      case 14: v = n.nodePairingTime; break;
      case 15: v = n.extraLatency; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
// per (node, level) ints, row-major [node][level]: 0 posInLevel 1 remainingCalls
int orc_gsf_read_level(void* h, int field, int32_t* out) {
  ORC_TRY auto& p = *((OrcGsf*)h)->p;
  int L = orc_gsf_levels(h);
  for (size_t i = 0; i < p.nodes.size(); i++)
    for (int l = 0; l < L; l++) {
      auto& n = *p.nodes[i];
      int v = 0;
      if (!n.levels.empty()) {
        auto& lv = *n.levels[l];
        switch (field) {
          case 0: v = lv.posInLevel; break;
          case 1: v = lv.remainingCalls; break;
          default: throw IllegalArgumentException("field");
        }
      }
      out[i * L + l] = v;
    }
  ORC_CATCH
}
// One nodeCount-bit row per node (words = nodeCount/64 rounded up), bit j = node id j.
// which: 0 GSFNode.verifiedSignatures; union over levels of 1 SFLevel.verifiedSignatures
//        2 individualSignatures 3 indivVerifiedSig 4 waitedSigs
int orc_gsf_read_bits(void* h, int which, uint64_t* out) {
  ORC_TRY auto& p = *((OrcGsf*)h)->p;
  int N = p.params.nodeCount;
  int W = (N + 63) / 64;
  memset(out, 0, sizeof(uint64_t) * (size_t)W * p.nodes.size());
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    uint64_t* row = out + i * W;
    if (which == 0) {
      for (int w = 0; w < W; w++) row[w] = n.verifiedSignatures.wordAt(w);
      continue;
    }
    for (size_t l = 0; l < n.levels.size(); l++) {
      auto& lv = *n.levels[l];
      const BitSet* b = nullptr;
      switch (which) {
        case 1: b = &lv.verifiedSignatures; break;
        case 2: b = &lv.individualSignatures; break;
        case 3: b = &lv.indivVerifiedSig; break;
        case 4: b = &lv.waitedSigs; break;
        default: throw IllegalArgumentException("which");
      }
      for (int w = 0; w < W; w++) row[w] |= b->wordAt(w);
    }
  }
  ORC_CATCH
}
int orc_gsf_read_peers(void* h, int node, int level, int32_t* out, int32_t* cnt) {
  auto& n = *((OrcGsf*)h)->p->node(node);
  if (n.levels.empty()) {
    *cnt = 0;
    return 0;
  }
  auto& lv = *n.levels[level];
  *cnt = (int)lv.peers.size();
  for (size_t i = 0; i < lv.peers.size(); i++) out[i] = lv.peers[i]->nodeId;
  return 0;
}
int orc_gsf_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered, uint64_t* tasks,
                 int32_t* queueMax, double* initSeconds) {
  auto& p = *((OrcGsf*)h)->p;
  *time = p.network().time;
  if (queueSize) *queueSize = p.network().msgs.size();
  *rngState = p.network().rd.rawState();
  *delivered = p.network().statDelivered;
  *tasks = p.network().statTasks;
  *queueMax = p.statQueueMax;
  *initSeconds = ((OrcGsf*)h)->initSeconds;
  return 0;
}
uint64_t orc_gsf_shape_violations(void* h) { return ((OrcGsf*)h)->p->statShapeViolations; }
int orc_gsf_stats(void* h, uint64_t* deliveredByLevel32) {
  memcpy(deliveredByLevel32, ((OrcGsf*)h)->p->statDeliveredByLevel, sizeof(uint64_t) * 32);
  return 0;
}


// ---- Casper IMD (P/CasperIMD.java)
struct OrcCasper {
  std::unique_ptr<CasperIMD> p;
};
// ip: cycleLength, randomOnTies, blockProducersCount, attestersPerRound, blockConstructionTime,
//     attestationConstructionTime, delay of the ByzBlockProducerWF that init(badNode) starts with (:475-479 uses 0).
// As RunMultipleTimes does (C/RunMultipleTimes.java:44-48): new CasperIMD(params) — which already builds the observer
// node from rd (:80-87) — then rd.setSeed(seed), then init().
int orc_casper_create_byz(const int32_t* ip, const char* nb, const char* nl, int64_t seed, int byzKind, void** out);
int orc_casper_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  return orc_casper_create_byz(ip, nb, nl, seed, 0, out);
}
// byzKind: the byzantine producer init(badNode) is given (P/CasperIMD.java:481) — 0 ByzBlockProducerWF (what init() itself
// installs, :475-479), 1 ByzBlockProducer (:511-581), 2 ByzBlockProducerSF (:583-604), 3 ByzBlockProducerNS (:610-633)
int orc_casper_create_byz(const int32_t* ip, const char* nb, const char* nl, int64_t seed, int byzKind, void** out) {
  ORC_TRY CasperIMD::CasperParemeters pr(ip[0], ip[1] != 0, ip[2], ip[3], ip[4], ip[5], nb ? nb : "", nl ? nl : "");
  auto* h = new OrcCasper();
  h->p = std::make_unique<CasperIMD>(pr);
  h->p->network().rd.setSeed(seed);
  CasperIMD::ByzBlockProducer* byz = nullptr;
  switch (byzKind) {
    case 0: byz = h->p->make<CasperIMD::ByzBlockProducerWF>(ip[6]); break;
    case 1: byz = h->p->make<CasperIMD::ByzBlockProducerPlain>(ip[6]); break;
    case 2: byz = h->p->make<CasperIMD::ByzBlockProducerSF>(ip[6]); break;
    case 3: byz = h->p->make<CasperIMD::ByzBlockProducerNS>(ip[6]); break;
    default: delete h; throw IllegalArgumentException("byzKind");
  }
  h->p->init(byz);
  *out = h;
  ORC_CATCH
}
// counters of the byzantine producer (node 1): onDirectFather, onOlderAncestor, incNotTheBestFather, skipped (NS), toSend
int orc_casper_byz_counters(void* h, int32_t* out5) {
  ORC_TRY auto* b = dynamic_cast<CasperIMD::ByzBlockProducer*>(((OrcCasper*)h)->p->network().allNodes[1]);
  if (!b) throw IllegalStateException("node 1 is not a ByzBlockProducer");
  auto* ns = dynamic_cast<CasperIMD::ByzBlockProducerNS*>(b);
  out5[0] = b->onDirectFather;
  out5[1] = b->onOlderAncestor;
  out5[2] = b->incNotTheBestFather;
  out5[3] = ns ? ns->skipped : 0;
  out5[4] = b->toSend;
  ORC_CATCH
}
void orc_casper_destroy(void* h) { delete (OrcCasper*)h; }
int orc_casper_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcCasper*)h)->p->network().runMs(ms);
  ORC_CATCH
}
// BASELINE config 5's "+10 % Byzantine" has no counterpart in the reference (init() installs exactly one
// ByzBlockProducerWF, P/CasperIMD.java:473-476); SURVEY.md §8d defines it as attesters stop()ped after init(),
// identically in the oracle and the engine: Node.stop() (C/Node.java:120-123) on the listed nodes.
int orc_casper_stop(void* h, const int32_t* ids, int n) {
  ORC_TRY auto& net = ((OrcCasper*)h)->p->network();
  for (int i = 0; i < n; i++) {
    if (ids[i] < 0 || ids[i] >= (int)net.allNodes.size()) throw IllegalArgumentException("node id");
    net.allNodes[ids[i]]->stop();
  }
  ORC_CATCH
}
int orc_casper_node_count(void* h) { return (int)((OrcCasper*)h)->p->network().allNodes.size(); }
// fields: 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 head.height, 5 head.proposalTime, 6 head.id,
//         7 attestationsByHead.size(), 8 x, 9 y, 10 blocksReceivedByBlockId.size(), 11 attestations held (all heads)
int orc_casper_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& net = ((OrcCasper*)h)->p->network();
  for (size_t i = 0; i < net.allNodes.size(); i++) {
    auto& n = *static_cast<CasperIMD::CasperNode*>(net.allNodes[i]);
    int64_t v = 0;
    switch (field) {
      case 0: v = n.msgReceived; break;
      case 1: v = n.msgSent; break;
      case 2: v = n.bytesSent; break;
      case 3: v = n.bytesReceived; break;
      case 4: v = n.head->height; break;
      case 5: v = n.head->proposalTime; break;
      case 6: v = n.head->id; break;
      case 7: v = (int64_t)n.attestationsByHead.size(); break;
      case 8: v = n.x; break;
      case 9: v = n.y; break;
      case 10: v = (int64_t)n.blocksReceivedByBlockId.size(); break;
      case 11:
        for (auto& kv : n.attestationsByHead) v += (int64_t)kv.second.size();
        break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_casper_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered, uint64_t* tasks) {
  auto& p = *((OrcCasper*)h)->p;
  *time = p.network().time;
  if (queueSize) *queueSize = p.network().msgs.size();
  *rngState = p.network().rd.rawState();
  *delivered = p.network().statDelivered;
  *tasks = p.network().statTasks;
  return 0;
}


// ---- Fuzz (oracle/fuzz.hpp): scheduler stress protocol, test infrastructure on both sides
struct OrcFuzz {
  std::unique_ptr<Fuzz> p;
};
int orc_fuzz_create(int n, int ttl, const char* nl, int64_t seed, void** out) {
  ORC_TRY auto* h = new OrcFuzz();
  h->p = std::make_unique<Fuzz>(n, ttl, nl ? nl : "");
  h->p->network_.rd.setSeed(seed);
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_fuzz_destroy(void* h) { delete (OrcFuzz*)h; }
int orc_fuzz_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcFuzz*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
// ops between chunks: 0 partition(arg/1000.f), 1 endPartition, 2 stop node arg, 3 start node arg, 4 setMsgDiscardTime(arg)
int orc_fuzz_op(void* h, int op, int arg) {
  ORC_TRY auto& p = *((OrcFuzz*)h)->p;
  switch (op) {
    case 0: p.network_.partition(arg / 1000.f); break;
    case 1: p.network_.endPartition(); break;
    case 2: p.node(arg)->stop(); break;
    case 3: p.node(arg)->start(); break;
    case 4: p.network_.setMsgDiscardTime(arg); break;
    default: throw IllegalArgumentException("op");
  }
  ORC_CATCH
}
// fields: 0 h, 1 c, 2 msgReceived, 3 msgSent, 4 bytesSent, 5 bytesReceived
int orc_fuzz_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcFuzz*)h)->p;
  for (int i = 0; i < p.N; i++) {
    auto& n = *p.nodes[i];
    out[i] = field == 0 ? (int64_t)n.h : field == 1 ? n.c : field == 2 ? n.msgReceived : field == 3 ? n.msgSent
             : field == 4 ? n.bytesSent : n.bytesReceived;
  }
  ORC_CATCH
}
int orc_fuzz_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered, uint64_t* tasks) {
  auto& p = *((OrcFuzz*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  *tasks = p.network_.statTasks;
  return 0;
}


// ---- San Fermin (P/SanFerminSignature.java)
struct OrcSanFermin {
  std::unique_ptr<SanFerminSignature> p;
};
// ip: nodeCount, threshold, pairingTime, signatureSize, replyTimeout, candidateCount (SanFerminSignatureParameters
// ctor order :84-104; shuffledLists is read nowhere in the protocol). The constructor builds the nodes from rd (:126-131),
// so — as RunMultipleTimes does — the seed is set on the fresh Network's rd only after them: here the seed is applied
// BEFORE construction through `seed_before` when nonzero semantics are wanted by the caller (tests use both).
int orc_sanfermin_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY SanFerminSignature::Params pr;
  pr.nodeCount = ip[0];
  pr.threshold = ip[1];
  pr.pairingTime = ip[2];
  pr.signatureSize = ip[3];
  pr.replyTimeout = ip[4];
  pr.candidateCount = ip[5];
  pr.nodeBuilderName = nb ? nb : "";
  pr.networkLatencyName = nl ? nl : "";
  auto* h = new OrcSanFermin();
  h->p = std::make_unique<SanFerminSignature>(pr);
  h->p->network_.rd.setSeed(seed);  // rd.setSeed(i) on the copy, then init()  C/RunMultipleTimes.java:44-48
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_sanfermin_destroy(void* h) { delete (OrcSanFermin*)h; }
int orc_sanfermin_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcSanFermin*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
// fields: 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 aggValue, 5 currentPrefixLength, 6 doneAt,
//         7 thresholdAt, 8 sentRequests, 9 receivedRequests, 10 done, 11 isSwapping, 12 x, 13 y
int orc_sanfermin_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcSanFermin*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.msgReceived; break;
      case 1: v = n.msgSent; break;
      case 2: v = n.bytesSent; break;
      case 3: v = n.bytesReceived; break;
      case 4: v = n.aggValue; break;
      case 5: v = n.currentPrefixLength; break;
      case 6: v = n.doneAt; break;
      case 7: v = n.thresholdAt; break;
      case 8: v = n.sentRequests; break;
      case 9: v = n.receivedRequests; break;
      case 10: v = n.done; break;
      case 11: v = n.isSwapping; break;
      case 12: v = n.x; break;
      case 13: v = n.y; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_sanfermin_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered, uint64_t* tasks,
                       int32_t* finished) {
  auto& p = *((OrcSanFermin*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  *tasks = p.network_.statTasks;
  *finished = (int32_t)p.finishedNodes.size();
  return 0;
}

// ---- San Fermin, Cappos' variant (P/SanFerminCappos.java)
struct OrcCappos {
  std::unique_ptr<SanFerminCappos> p;
};
// ip: nodeCount, threshold, pairingTime, signatureSize, timeout, candidateCount (SanFerminParameters ctor order :87-106)
int orc_cappos_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY SanFerminCappos::Params pr;
  pr.nodeCount = ip[0];
  pr.threshold = ip[1];
  pr.pairingTime = ip[2];
  pr.signatureSize = ip[3];
  pr.timeout = ip[4];
  pr.candidateCount = ip[5];
  pr.nodeBuilderName = nb ? nb : "";
  pr.networkLatencyName = nl ? nl : "";
  auto* h = new OrcCappos();
  h->p = std::make_unique<SanFerminCappos>(pr);
  h->p->network_.rd.setSeed(seed);  // rd.setSeed(i) on the copy, then init() — which builds the nodes here (:121-126)
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_cappos_destroy(void* h) { delete (OrcCappos*)h; }
int orc_cappos_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcCappos*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
// fields: 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 totalNumberOfSigs(-1), 5 currentPrefixLength, 6 doneAt,
//         7 thresholdAt, 8 cached levels, 9 cached values, 10 done, 11 isSwapping, 12 x, 13 y
int orc_cappos_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcCappos*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.msgReceived; break;
      case 1: v = n.msgSent; break;
      case 2: v = n.bytesSent; break;
      case 3: v = n.bytesReceived; break;
      case 4: v = n.totalNumberOfSigs(-1); break;
      case 5: v = n.currentPrefixLength; break;
      case 6: v = n.doneAt; break;
      case 7: v = n.thresholdAt; break;
      case 8: v = (int64_t)n.signatureCache.size(); break;
      case 9:
        for (auto& e : n.signatureCache) v += (int64_t)e.second.size();
        break;
      case 10: v = n.done; break;
      case 11: v = n.isSwapping; break;
      case 12: v = n.x; break;
      case 13: v = n.y; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_cappos_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered, uint64_t* tasks,
                    int32_t* finished) {
  auto& p = *((OrcCappos*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  *tasks = p.network_.statTasks;
  *finished = (int32_t)p.finishedNodes.size();
  return 0;
}

// ---- P2PFlood (P/P2PFlood.java over C/P2PNetwork.java, C/messages/FloodMessage.java)
struct OrcFlood {
  std::unique_ptr<P2PFlood> p;
};
// ip: nodeCount, deadNodeCount, delayBeforeResent, msgCount, msgToReceive, peersCount, delayBetweenSends (:63-86)
int orc_p2pflood_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY P2PFlood::Params pr;
  pr.nodeCount = ip[0];
  pr.deadNodeCount = ip[1];
  pr.delayBeforeResent = ip[2];
  pr.msgCount = ip[3];
  pr.msgToReceive = ip[4];
  pr.peersCount = ip[5];
  pr.delayBetweenSends = ip[6];
  pr.nodeBuilderName = nb ? nb : "";
  pr.networkLatencyName = nl ? nl : "";
  auto* h = new OrcFlood();
  h->p = std::make_unique<P2PFlood>(pr);
  h->p->network_.rd.setSeed(seed);
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_p2pflood_destroy(void* h) { delete (OrcFlood*)h; }
int orc_p2pflood_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcFlood*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
// fields: 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 doneAt, 5 down, 6 getMsgReceived(-1).size(),
//         7 peers.size(), 8 sum of peer ids, 9 x, 10 y
int orc_p2pflood_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcFlood*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.msgReceived; break;
      case 1: v = n.msgSent; break;
      case 2: v = n.bytesSent; break;
      case 3: v = n.bytesReceived; break;
      case 4: v = n.doneAt; break;
      case 5: v = n.down; break;
      case 6: v = (int64_t)n.getMsgReceived(-1).size(); break;
      case 7: v = (int64_t)n.peers.size(); break;
      case 8:
        for (size_t k = 0; k < n.peers.size(); k++) v += (int64_t)(k + 1) * n.peers[k]->nodeId;  // order-sensitive
        break;
      case 9: v = n.x; break;
      case 10: v = n.y; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_p2pflood_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered) {
  auto& p = *((OrcFlood*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  return 0;
}

// ---- OptimisticP2PSignature (P/OptimisticP2PSignature.java over C/P2PNetwork.java)
struct OrcOptP2P {
  std::unique_ptr<OptimisticP2PSignature> p;
};
// ip: nodeCount, threshold, connectionCount, pairingTime (:58-71)
int orc_optp2p_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY OptimisticP2PSignature::Params pr;
  pr.nodeCount = ip[0];
  pr.threshold = ip[1];
  pr.connectionCount = ip[2];
  pr.pairingTime = ip[3];
  pr.nodeBuilderName = nb ? nb : "";
  pr.networkLatencyName = nl ? nl : "";
  auto* h = new OrcOptP2P();
  h->p = std::make_unique<OptimisticP2PSignature>(pr);
  h->p->network_.rd.setSeed(seed);
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_optp2p_destroy(void* h) { delete (OrcOptP2P*)h; }
int orc_optp2p_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcOptP2P*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
// fields: 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 doneAt, 5 done, 6 verifiedSignatures.cardinality(),
//         7 peers.size(), 8 order-sensitive digest of the peer ids, 9 x, 10 y
int orc_optp2p_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcOptP2P*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.msgReceived; break;
      case 1: v = n.msgSent; break;
      case 2: v = n.bytesSent; break;
      case 3: v = n.bytesReceived; break;
      case 4: v = n.doneAt; break;
      case 5: v = n.done; break;
      case 6: v = n.verifiedSignatures.cardinality(); break;
      case 7: v = (int64_t)n.peers.size(); break;
      case 8:
        for (size_t k = 0; k < n.peers.size(); k++) v += (int64_t)(k + 1) * n.peers[k]->nodeId;
        break;
      case 9: v = n.x; break;
      case 10: v = n.y; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_optp2p_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered) {
  auto& p = *((OrcOptP2P*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  return 0;
}

// ---- Slush / Snowflake (P/Slush.java, P/Snowflake.java)
struct OrcSlush {
  std::unique_ptr<Slush> p;
};
// ip: NODES_AV, M, K, B, snowflake (0 / 1); a: alpha (:37-47 / :36-52)
int orc_slush_create(const int32_t* ip, double a, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY Slush::Params pr;
  pr.NODES_AV = ip[0];
  pr.M = ip[1];
  pr.K = ip[2];
  pr.B = ip[3];
  pr.A = a;
  pr.nodeBuilderName = nb ? nb : "";
  pr.networkLatencyName = nl ? nl : "";
  auto* h = new OrcSlush();
  h->p = std::make_unique<Slush>(pr, ip[4] != 0);
  h->p->network_.rd.setSeed(seed);
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_slush_destroy(void* h) { delete (OrcSlush*)h; }
int orc_slush_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcSlush*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
// fields: 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 myColor, 5 myQueryNonce, 6 round, 7 cnt,
//         8 answerIP.size(), 9 x, 10 y
int orc_slush_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcSlush*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.msgReceived; break;
      case 1: v = n.msgSent; break;
      case 2: v = n.bytesSent; break;
      case 3: v = n.bytesReceived; break;
      case 4: v = n.myColor; break;
      case 5: v = n.myQueryNonce; break;
      case 6: v = n.round; break;
      case 7: v = n.cnt; break;
      case 8: v = (int64_t)n.answerIP.size(); break;
      case 9: v = n.x; break;
      case 10: v = n.y; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_slush_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered) {
  auto& p = *((OrcSlush*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  return 0;
}

// ---- Paxos (P/Paxos.java)
struct OrcPaxos {
  std::unique_ptr<Paxos> p;
};
// ip: acceptorCount, proposerCount, timeout (:359-370)
int orc_paxos_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY Paxos::Params pr;
  pr.acceptorCount = ip[0];
  pr.proposerCount = ip[1];
  pr.timeout = ip[2];
  pr.nodeBuilder = nb ? nb : "";
  pr.latency = nl ? nl : "";
  auto* h = new OrcPaxos();
  h->p = std::make_unique<Paxos>(pr);
  h->p->network_.rd.setSeed(seed);
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_paxos_destroy(void* h) { delete (OrcPaxos*)h; }
int orc_paxos_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcPaxos*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
// per node (acceptors first, then proposers; a field of the other kind reads -2; a null Integer reads -1):
// 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 doneAt, 5 x, 6 y,
// acceptors: 7 maxAgreed, 8 acceptedSeq, 9 acceptedVal, 10 agreedTo (node id)
// proposers: 11 valueProposed, 12 valueAccepted, 13 seqIP, 14 seqAccepted, 15 agreeCount, 16 reject1Count, 17 reject2Count,
//            18 timeoutCount, 19 proposalIP, 20 agreeCountIP, 21 acceptCountIP
int orc_paxos_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcPaxos*)h)->p;
  auto opt = [](const std::optional<int>& o) -> int64_t { return o ? *o : -1; };
  for (size_t i = 0; i < p.nodes.size(); i++) {
    Paxos::PaxosNode* n = p.nodes[i].get();
    auto* a = dynamic_cast<Paxos::AcceptorNode*>(n);
    auto* q = dynamic_cast<Paxos::ProposerNode*>(n);
    int64_t v = -2;
    switch (field) {
      case 0: v = n->msgReceived; break;
      case 1: v = n->msgSent; break;
      case 2: v = n->bytesSent; break;
      case 3: v = n->bytesReceived; break;
      case 4: v = n->doneAt; break;
      case 5: v = n->x; break;
      case 6: v = n->y; break;
      case 7: if (a) v = a->maxAgreed; break;
      case 8: if (a) v = opt(a->acceptedSeq); break;
      case 9: if (a) v = opt(a->acceptedVal); break;
      case 10: if (a) v = a->agreedTo ? a->agreedTo->nodeId : -1; break;
      case 11: if (q) v = q->valueProposed; break;
      case 12: if (q) v = opt(q->valueAccepted); break;
      case 13: if (q) v = q->seqIP; break;
      case 14: if (q) v = q->seqAccepted; break;
      case 15: if (q) v = q->agreeCount; break;
      case 16: if (q) v = q->reject1Count; break;
      case 17: if (q) v = q->reject2Count; break;
      case 18: if (q) v = q->timeoutCount; break;
      case 19: if (q) v = q->proposalIP; break;
      case 20: if (q) v = q->agreeCountIP; break;
      case 21: if (q) v = q->acceptCountIP; break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_paxos_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered) {
  auto& p = *((OrcPaxos*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  return 0;
}

// ---- Dfinity (P/Dfinity.java)
struct OrcDfinity {
  std::unique_ptr<Dfinity> p;
};
// ip: blockProducersCount, attestersCount, attestersPerRound, blockConstructionTime, attestationConstructionTime,
// percentageDeadAttester (:34-42). The observer is built by the constructor, BEFORE rd.setSeed(seed) — as `p.copy();
// rd.setSeed(i); init()` does it (C/RunMultipleTimes.java:44-48)
int orc_dfinity_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY Dfinity::Params pr;
  pr.blockProducersCount = ip[0];
  pr.attestersCount = ip[1];
  pr.attestersPerRound = ip[2];
  pr.blockConstructionTime = ip[3];
  pr.attestationConstructionTime = ip[4];
  pr.percentageDeadAttester = ip[5];
  pr.nodeBuilderName = nb ? nb : "";
  pr.networkLatencyName = nl ? nl : "";
  auto* h = new OrcDfinity();
  h->p = std::make_unique<Dfinity>(pr);
  h->p->network_.rd.setSeed(seed);
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_dfinity_destroy(void* h) { delete (OrcDfinity*)h; }
int orc_dfinity_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcDfinity*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
int orc_dfinity_node_count(void* h) { return (int)((OrcDfinity*)h)->p->nodes.size(); }
// per node (observer, attesters, producers, beacon nodes; a field of another kind reads -2):
// 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 x, 5 y, 6 head.height, 7 head.id, 8 head.proposalTime,
// 9 lastRandomBeacon, 10 blocks received, 11 |committeeMajorityBlocks|, 12 sum of committeeMajorityHeight,
// attesters: 13 voteForHeight, 14 proposals.size(), 15 votes.size(); producers: 16 waitForBlockHeight, 17 myRound;
// beacon nodes: 18 height, 19 lastRDSent, 20 rd, 21 exchanged.size()
int orc_dfinity_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcDfinity*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    Dfinity::DfinityNode* n = p.nodes[i].get();
    auto* a = dynamic_cast<Dfinity::AttesterNode*>(n);
    auto* b = dynamic_cast<Dfinity::BlockProducerNode*>(n);
    auto* r = dynamic_cast<Dfinity::RandomBeaconNode*>(n);
    int64_t v = -2;
    switch (field) {
      case 0: v = n->msgReceived; break;
      case 1: v = n->msgSent; break;
      case 2: v = n->bytesSent; break;
      case 3: v = n->bytesReceived; break;
      case 4: v = n->x; break;
      case 5: v = n->y; break;
      case 6: v = n->head->height; break;
      case 7: v = n->head->id; break;
      case 8: v = n->head->proposalTime; break;
      case 9: v = n->lastRandomBeacon; break;
      case 10: v = (int64_t)n->blocksReceivedByBlockId.size(); break;
      case 11: v = (int64_t)n->committeeMajorityBlocks.size(); break;
      case 12:
        v = 0;
        for (int x : n->committeeMajorityHeight) v += x;
        break;
      case 13: if (a) v = a->voteForHeight; break;
      case 14: if (a) v = (int64_t)a->proposals.size(); break;
      case 15: if (a) v = (int64_t)a->votes.size(); break;
      case 16: if (b) v = b->waitForBlockHeight; break;
      case 17: if (b) v = b->myRound; else if (a) v = a->myRound; break;
      case 18: if (r) v = r->height; break;
      case 19: if (r) v = r->lastRDSent; break;
      case 20: if (r) v = r->rd; break;
      case 21: if (r) v = (int64_t)r->exchanged.size(); break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_dfinity_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered) {
  auto& p = *((OrcDfinity*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  return 0;
}

// ---- P2PHandel (P/P2PHandel.java over C/P2PNetwork.java)
struct OrcP2PHandel {
  std::unique_ptr<P2PHandel> p;
};
// ip: signingNodeCount, relayingNodeCount, threshold, connectionCount, pairingTime, sigsSendPeriod, doubleAggregateStrategy,
// sendSigsStrategy (0 all, 1 dif, 2 cmp_all, 3 cmp_diff), sendState (:80-104)
int orc_p2phandel_create(const int32_t* ip, const char* nb, const char* nl, int64_t seed, void** out) {
  ORC_TRY P2PHandel::Params pr;
  pr.signingNodeCount = ip[0];
  pr.relayingNodeCount = ip[1];
  pr.threshold = ip[2];
  pr.connectionCount = ip[3];
  pr.pairingTime = ip[4];
  pr.sigsSendPeriod = ip[5];
  pr.doubleAggregateStrategy = ip[6] != 0;
  pr.sendSigsStrategy = (P2PHandel::SendSigsStrategy)ip[7];
  pr.sendState = ip[8] != 0;
  pr.nodeBuilderName = nb ? nb : "";
  pr.networkLatencyName = nl ? nl : "";
  auto* h = new OrcP2PHandel();
  h->p = std::make_unique<P2PHandel>(pr);
  h->p->network_.rd.setSeed(seed);
  h->p->init();
  *out = h;
  ORC_CATCH
}
void orc_p2phandel_destroy(void* h) { delete (OrcP2PHandel*)h; }
int orc_p2phandel_run_ms(void* h, int ms, int* didSomething) {
  ORC_TRY* didSomething = ((OrcP2PHandel*)h)->p->network_.runMs(ms);
  ORC_CATCH
}
static int64_t bits_digest(const BitSet& b) {  // order-sensitive digest of the set bits
  int64_t v = 0;
  for (int i = b.nextSetBit(0), k = 1; i >= 0; i = b.nextSetBit(i + 1), k++) v += (int64_t)k * (i + 1);
  return v;
}
// fields: 0 msgReceived, 1 msgSent, 2 bytesSent, 3 bytesReceived, 4 doneAt, 5 x, 6 y, 7 |verifiedSignatures|, 8 its digest,
//         9 toVerify.size(), 10 the table length of toVerify, 11 a digest of toVerify in ITERATION order, 12 peers.size(),
//         13 order-sensitive digest of the peer ids, 14 justRelay, 15 sum over the peers of |peersState|
int orc_p2phandel_read(void* h, int field, int64_t* out) {
  ORC_TRY auto& p = *((OrcP2PHandel*)h)->p;
  for (size_t i = 0; i < p.nodes.size(); i++) {
    auto& n = *p.nodes[i];
    int64_t v = 0;
    switch (field) {
      case 0: v = n.msgReceived; break;
      case 1: v = n.msgSent; break;
      case 2: v = n.bytesSent; break;
      case 3: v = n.bytesReceived; break;
      case 4: v = n.doneAt; break;
      case 5: v = n.x; break;
      case 6: v = n.y; break;
      case 7: v = n.verifiedSignatures.cardinality(); break;
      case 8: v = bits_digest(n.verifiedSignatures); break;
      case 9: v = n.toVerify.size(); break;
      case 10: v = n.toVerify.capacity(); break;
      case 11: {
        int k = 1;
        for (const auto& b : n.toVerify.items()) v += (int64_t)(k++) * (bits_digest(*b) % 1000003);
        break;
      }
      case 12: v = (int64_t)n.peers.size(); break;
      case 13:
        for (size_t k = 0; k < n.peers.size(); k++) v += (int64_t)(k + 1) * n.peers[k]->nodeId;
        break;
      case 14: v = n.justRelay; break;
      case 15:
        for (const auto& kv : n.peersState) v += kv.second->cardinality();
        break;
      default: throw IllegalArgumentException("field");
    }
    out[i] = v;
  }
  ORC_CATCH
}
int orc_p2phandel_info(void* h, int32_t* time, int32_t* queueSize, uint64_t* rngState, uint64_t* delivered) {
  auto& p = *((OrcP2PHandel*)h)->p;
  *time = p.network_.time;
  *queueSize = p.network_.msgs.size();
  *rngState = p.network_.rd.rawState();
  *delivered = p.network_.statDelivered;
  return 0;
}
// compressedSize of the set whose bits are the '1' characters of `binary` (PT/P2PHandelTest.testCompressedSize / fromString)
int orc_p2phandel_compressed_size(void* h, const char* binary) {
  BitSet b;
  int i = 0;
  for (const char* c = binary; *c; c++) {
    if (*c == ' ') continue;
    if (*c == '1') b.set(i);
    i++;
  }
  return ((OrcP2PHandel*)h)->p->compressedSize(b);
}
// PT/P2PHandelTest.testCheckSigs (out[0] = toVerify.isEmpty(), out[1] = msgs.size()) and testSigUpdate (out[2] = cardinality) on node 1
int orc_p2phandel_probe(void* h, int32_t* out) {
  ORC_TRY auto& p = *((OrcP2PHandel*)h)->p;
  auto& n1 = *p.nodes.at(1);
  auto sigs = std::make_shared<BitSet>();
  sigs->set(n1.nodeId);
  sigs->set(0);
  n1.toVerify.add(sigs);
  p.network_.msgs.clear();
  n1.checkSigs();
  out[0] = n1.toVerify.isEmpty();
  out[1] = p.network_.msgs.size();
  BitSet s2;
  s2.set(n1.nodeId);
  s2.set(0);
  n1.updateVerifiedSignatures(s2);
  out[2] = n1.verifiedSignatures.cardinality();
  ORC_CATCH
}

}  // extern "C"
