/*
 * wittgpu_host.h — host-side mirror of the reference's Protocol objects for the resident
 * protocols, written in C++ because no JVM exists in the build environment (INTEGRATION.md shows
 * the Java-side equivalent). Each call restates `new P(params); network.rd.setSeed(seed); p.init()`
 * with java.util.Random semantics and hands the result to the C ABI of wittgpu.h:
 *   PingPong.init()  P/PingPong.java:81-87
 *   Handel.init()    P/Handel.java:957-1014  (+ HandelParameters checks :113-125)
 *   GSFSignature.init()  P/GSFSignature.java:611-635  (+ GSFSignatureParameters checks :69-74)
 *   SanFerminSignature ctor + init()  P/SanFerminSignature.java:113-141
 * (P/ = protocols/src/main/java/net/consensys/wittgenstein/protocols/)
 */
#ifndef WITTGPU_HOST_H
#define WITTGPU_HOST_H
#include "wittgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* nodeBuilderName: RegistryNodeBuilders name, RANDOM location only ("RANDOM_SPEED=CONSTANT_TOR=0.00",
 * NULL = that default; C/RegistryNodeBuilders.java:28-81). latencyName: RegistryNetworkLatencies name
 * (NULL = NetworkLatencyByDistanceWJitter). On failure *out is NULL and wgh_last_error() has the text. */
int32_t wgh_pingpong_create(int32_t nodeCt, const char* nodeBuilderName, const char* latencyName, int64_t seed,
                            const wg_config* cfg, wg_engine** out);
int32_t wgh_handel_create(const wg_handel_params* params, const char* nodeBuilderName, const char* latencyName,
                          int64_t seed, const wg_config* cfg, wg_engine** out);
/* ... with HandelParameters.badNodes (P/Handel.java:51,110,139): the explicit set init() uses instead of
 * Network.chooseBadNodes' draws (:960-964) — badNodes[nodeCount], non-zero = that node is down (and byzantine under an attack
 * flag); NULL = wgh_handel_create. params->nodesDown keeps its role in the constructor's checks (:113-118) only. */
int32_t wgh_handel_create_bad_nodes(const wg_handel_params* params, const uint8_t* badNodes, const char* nodeBuilderName,
                                    const char* latencyName, int64_t seed, const wg_config* cfg, wg_engine** out);
int32_t wgh_gsf_create(const wg_gsf_params* params, const char* nodeBuilderName, const char* latencyName,
                       int64_t seed, const wg_config* cfg, wg_engine** out);
/* new SanFerminSignature(params) — which builds the nodes from the fresh Network's rd, new Random(0) (:126-131) —
 * then, as RunMultipleTimes does, rd.setSeed(seed) and init() (P/SanFerminSignature.java:113-141) */
int32_t wgh_sanfermin_create(const wg_sanfermin_params* params, const char* nodeBuilderName, const char* latencyName,
                             int64_t seed, const wg_config* cfg, wg_engine** out);
/* new CasperIMD(params) — which builds the observer from new Random(0) (:80-87) — rd.setSeed(seed), init(new
 * ByzBlockProducerWF(params->byzDelay)) (P/CasperIMD.java:481-509) */
int32_t wgh_casper_create(const wg_casper_params* params, const char* nodeBuilderName, const char* latencyName,
                          int64_t seed, const wg_config* cfg, wg_engine** out);
/* new P2PFlood(params), rd.setSeed(seed), init(): nodes (the first deadNodeCount stopped), P2PNetwork.setPeers, one
 * sendPeers per message from a random live node (P/P2PFlood.java:88-140, C/P2PNetwork.java:27-56,127-132) */
int32_t wgh_p2pflood_create(const wg_p2pflood_params* params, const char* nodeBuilderName, const char* latencyName,
                            int64_t seed, const wg_config* cfg, wg_engine** out);
/* City node builders and city latency models (C/RegistryNodeBuilders.java:44-58, C/NetworkLatency.java:86-233). The
 * reference computes them from its resource files (cities.csv, the wondernetwork ping CSVs); here the caller hands the
 * computed tables over once per process, and the creators above then accept the registry names "AWS_SPEED=..._TOR=..." /
 * "CITIES_SPEED=..._TOR=..." and the registered latency names (e.g. "NetworkLatencyByCityWJitter"):
 *   site "AWS" | "CITIES": NodeBuilderWithCity's citiesInfo in ITS entrySet() order (a java.util.HashMap: the order
 *   decides which city a draw falls into, C/NodeBuilder.java:128-139) — cumulativeProbability, mercX, mercY per city,
 *   and cities.size() of the list the builder was given. A node's city is its row in this table;
 *   latency tables indexed by those rows: see wg_set_latency_city for the three modes. */
int32_t wgh_register_city_builder(const char* site, int32_t n, const float* cumulativeProbability, const int32_t* mercX,
                                  const int32_t* mercY, int32_t listSize);
int32_t wgh_register_city_latency(const char* latencyName, int32_t mode, int32_t nCities, const int32_t* tab,
                                  const float* ping, const double* jitter100);
const char* wgh_last_error(void);
/* seconds spent in the host-side init() of the last wgh_*_create on this thread */
double wgh_last_init_seconds(void);
/* 1 when the last wgh_handel_create on this thread left the emission lists (P/Handel.java:991-1013) to the device
 * (wg_handel_init_state.peers == NULL), 0 when the host built them (sharded, < 256 or > 131 072 nodes, WG_HOST_INIT=1, or the
 * device answered WG_EHOSTINIT) */
int32_t wgh_last_init_on_device(void);

/* java.util.Random known-answer probes of the product's own generator (tests) */
int32_t wgh_jrandom_ints(int64_t seed, int32_t n, int32_t* out);
int32_t wgh_jrandom_skip_ints(int64_t seed, int32_t n, int32_t* out); /* same values via LCG jump-ahead */
int32_t wgh_jrandom_bounded(int64_t seed, int32_t bound, int32_t n, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
