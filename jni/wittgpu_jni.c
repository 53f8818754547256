/*
 * wittgpu_jni.c — the JNI shim between Wittgenstein's Java host and libwittgpu.so: one native method of
 * net.consensys.wittgenstein.core.gpu.WittGpu (java/net/consensys/wittgenstein/core/gpu/WittGpu.java) per export of
 * include/wittgpu.h and include/wittgpu_host.h — nothing else. No engine logic lives here.
 *
 * Build where a JDK exists (none does in this repo's build image: this file is syntax-checked against a minimal
 * declaration of the JNI types it uses, tests/test_jni_sources.py, and every export is exercised through the same C ABI
 * by tests/c/test_abi_full.c and the ctypes tests):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/wittgpu_jni.c \
 *       -Lwittgenstein_amd -lwittgpu -o libwittgpu_jni.so
 *
 * Conventions: an engine / batch handle is a jlong; a non-zero wg_status becomes the Java exception the reference throws
 * at the corresponding site — WG_EINVAL IllegalArgumentException (C/Network.java:320,371,374,386,427,695), WG_ESTATE
 * IllegalStateException (:137,250,333,472,599,609,656,671), WG_EUNSUPPORTED UnsupportedOperationException, WG_ENOMEM /
 * WG_EHIP IllegalStateException with the engine's text; WG_EHOSTINIT is RETURNED (the caller re-runs init() on the host).
 * Arrays are pinned with Get<T>ArrayElements and released with JNI_ABORT when the engine only reads them.
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>
#include "wittgpu.h"
#include "wittgpu_host.h"

#define WG_JNI(ret, name) JNIEXPORT ret JNICALL Java_net_consensys_wittgenstein_core_gpu_WittGpu_##name
#define ENG(h) ((wg_engine*)(intptr_t)(h))
#define BAT(h) ((wg_batch*)(intptr_t)(h))

static void throw_msg(JNIEnv* env, int32_t rc, const char* msg) {
  const char* cls = rc == WG_EINVAL ? "java/lang/IllegalArgumentException"
                  : rc == WG_EUNSUPPORTED ? "java/lang/UnsupportedOperationException"
                                          : "java/lang/IllegalStateException";
  jclass c = (*env)->FindClass(env, cls);
  if (c) (*env)->ThrowNew(env, c, msg && *msg ? msg : "wittgpu");
}
/* status -> exception; returns the status so that `return ck(...)` hands WG_OK / WG_EHOSTINIT through */
static jint ck(JNIEnv* env, wg_engine* e, int32_t rc) {
  if (rc != WG_OK && rc != WG_EHOSTINIT) throw_msg(env, rc, wg_last_error(e));
  return rc;
}
static jint ck_host(JNIEnv* env, int32_t rc) {
  if (rc != WG_OK && rc != WG_EHOSTINIT) throw_msg(env, rc, wgh_last_error());
  return rc;
}
static jint ck_batch(JNIEnv* env, wg_batch* b, int32_t rc) {
  if (rc != WG_OK) throw_msg(env, rc, wg_batch_last_error(b));
  return rc;
}

/* A Java array shorter than what the engine reads or writes through it is an IllegalArgumentException HERE, before anything is
 * pinned — not a native out-of-bounds access into the JVM heap. NULL arrays are the ABI's "not wanted" and pass. */
static int need(JNIEnv* env, jarray a, jlong want, const char* what) {
  if (!a) return 1;
  if (want >= 0 && (jlong)(*env)->GetArrayLength(env, a) >= want) return 1;
  throw_msg(env, WG_EINVAL, what);
  return 0;
}
static jlong batch_members(JNIEnv* env, wg_batch* b) {
  int32_t n = 0;
  if (wg_batch_size(b, &n) != WG_OK) {
    throw_msg(env, WG_EINVAL, "not a batch");
    return -1;
  }
  return n;
}

/* read-only views of Java arrays (NULL array -> NULL pointer) */
#define PIN_I(a) ((a) ? (*env)->GetIntArrayElements(env, (a), NULL) : NULL)
#define UNPIN_I(a, p) do { if (a) (*env)->ReleaseIntArrayElements(env, (a), (p), JNI_ABORT); } while (0)
#define COMMIT_I(a, p) do { if (a) (*env)->ReleaseIntArrayElements(env, (a), (p), 0); } while (0)
#define PIN_J(a) ((a) ? (*env)->GetLongArrayElements(env, (a), NULL) : NULL)
#define UNPIN_J(a, p) do { if (a) (*env)->ReleaseLongArrayElements(env, (a), (p), JNI_ABORT); } while (0)
#define COMMIT_J(a, p) do { if (a) (*env)->ReleaseLongArrayElements(env, (a), (p), 0); } while (0)
#define PIN_B(a) ((a) ? (*env)->GetByteArrayElements(env, (a), NULL) : NULL)
#define UNPIN_B(a, p) do { if (a) (*env)->ReleaseByteArrayElements(env, (a), (p), JNI_ABORT); } while (0)
#define COMMIT_B(a, p) do { if (a) (*env)->ReleaseByteArrayElements(env, (a), (p), 0); } while (0)
#define PIN_D(a) ((a) ? (*env)->GetDoubleArrayElements(env, (a), NULL) : NULL)
#define UNPIN_D(a, p) do { if (a) (*env)->ReleaseDoubleArrayElements(env, (a), (p), JNI_ABORT); } while (0)
#define COMMIT_D(a, p) do { if (a) (*env)->ReleaseDoubleArrayElements(env, (a), (p), 0); } while (0)
#define PIN_F(a) ((a) ? (*env)->GetFloatArrayElements(env, (a), NULL) : NULL)
#define UNPIN_F(a, p) do { if (a) (*env)->ReleaseFloatArrayElements(env, (a), (p), JNI_ABORT); } while (0)
#define LEN(a) ((a) ? (*env)->GetArrayLength(env, (a)) : 0)
#define STR(s) ((s) ? (*env)->GetStringUTFChars(env, (s), NULL) : NULL)
#define UNSTR(s, p) do { if (s) (*env)->ReleaseStringUTFChars(env, (s), (p)); } while (0)

/* wg_config from int[8]: {device, horizon_ms, queue_cap, queue_cap_wide, chain_slots, shard, nshards, rank_bump_cap} then, as longs in
 * long[4], {bucket_pool_records, payload_words, outbox_records, chain_dests}; rcclId: the 128 bytes or null */
static void fill_config(JNIEnv* env, wg_config* c, jintArray ints, jlongArray longs, jbyte* idbuf, jbyteArray rcclId) {
  memset(c, 0, sizeof *c);
  jint iv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  jlong lv[4] = {0, 0, 0, 0};
  if (ints) (*env)->GetIntArrayRegion(env, ints, 0, LEN(ints) < 8 ? LEN(ints) : 8, iv);
  if (longs) (*env)->GetLongArrayRegion(env, longs, 0, LEN(longs) < 4 ? LEN(longs) : 4, lv);
  c->device = iv[0];
  c->horizon_ms = iv[1];
  c->queue_cap = iv[2];
  c->queue_cap_wide = iv[3];
  c->chain_slots = iv[4];
  c->shard = iv[5];
  c->nshards = iv[6];
  c->rank_bump_cap = iv[7];
  c->bucket_pool_records = lv[0];
  c->payload_words = lv[1];
  c->outbox_records = lv[2];
  c->chain_dests = lv[3];
  if (rcclId && LEN(rcclId) == WG_RCCL_UNIQUE_ID_BYTES) {
    (*env)->GetByteArrayRegion(env, rcclId, 0, WG_RCCL_UNIQUE_ID_BYTES, idbuf);
    c->rccl_id = (const uint8_t*)idbuf;
  }
}

/* the binding refuses a library whose structs are not the ones this file was compiled against (include/wittgpu.h) */
JNIEXPORT jint JNICALL JNI_OnLoad(JavaVM* vm, void* reserved) {
  (void)vm;
  (void)reserved;
  static const size_t sizes[9] = {sizeof(wg_config), sizeof(wg_handel_params), sizeof(wg_gsf_params), sizeof(wg_casper_params),
                                  sizeof(wg_sanfermin_params), sizeof(wg_p2pflood_params), sizeof(wg_delivery),
                                  sizeof(wg_step_op), sizeof(wg_run_stats)};
  if (wg_abi_version() != WG_ABI_VERSION) return JNI_ERR;
  for (int k = 0; k < 9; k++)
    if (wg_abi_struct_size(k) != (int32_t)sizes[k]) return JNI_ERR;
  return JNI_VERSION_1_8;
}
WG_JNI(jint, abiVersion)(JNIEnv* env, jclass c) { (void)env; (void)c; return wg_abi_version(); }
WG_JNI(jint, abiStructSize)(JNIEnv* env, jclass c, jint which) { (void)env; (void)c; return wg_abi_struct_size(which); }
/* wg_selftest: one wave / block primitive of the device kernels on the caller's values (tests) */
WG_JNI(jint, selfTest)(JNIEnv* env, jclass c, jint op, jint aux, jlongArray in, jint n, jint threads, jlongArray out) {
  (void)c;
  if (!in || !out || n < 0 || !need(env, in, n, "selfTest: in holds fewer than n values")) {
    if (!in || !out || n < 0) throw_msg(env, WG_EINVAL, "selfTest: in / out / n");
    return WG_EINVAL;
  }
  jlong *pi = PIN_J(in), *po = PIN_J(out);
  int32_t rc = wg_selftest(op, aux, (const uint64_t*)pi, n, threads, (uint64_t*)po, LEN(out));
  UNPIN_J(in, pi); COMMIT_J(out, po);
  if (rc != WG_OK) throw_msg(env, rc, wg_last_error(NULL));
  return rc;
}

/* ---- lifecycle: new Network<>() C/Network.java:14-49 ---- */
WG_JNI(jlong, create)(JNIEnv* env, jclass c, jintArray cfgInts, jlongArray cfgLongs, jbyteArray rcclId) {
  (void)c;
  wg_config cfg;
  jbyte id[WG_RCCL_UNIQUE_ID_BYTES];
  fill_config(env, &cfg, cfgInts, cfgLongs, id, rcclId);
  wg_engine* e = NULL;
  int32_t rc = wg_create(&cfg, &e);
  if (rc != WG_OK) { throw_msg(env, rc, wg_last_error(NULL)); return 0; }
  return (jlong)(intptr_t)e;
}
WG_JNI(void, destroy)(JNIEnv* env, jclass c, jlong h) { (void)env; (void)c; wg_destroy(ENG(h)); }
WG_JNI(jstring, lastError)(JNIEnv* env, jclass c, jlong h) { (void)c; return (*env)->NewStringUTF(env, wg_last_error(ENG(h))); }

/* ---- topology ---- */
WG_JNI(jint, addNodes)(JNIEnv* env, jclass c, jlong h, jintArray x, jintArray y, jintArray extra, jbyteArray down,
                       jbyteArray byzantine, jdoubleArray speed) {  /* Network.addNode C/Network.java:651-659 */
  (void)c;
  jsize n = LEN(x);
  if (!x || !y) { throw_msg(env, WG_EINVAL, "addNodes: x and y are required"); return WG_EINVAL; }
  if (!need(env, y, n, "addNodes: y is shorter than x") || !need(env, extra, n, "addNodes: extraLatency is shorter than x") ||
      !need(env, down, n, "addNodes: down is shorter than x") || !need(env, byzantine, n, "addNodes: byzantine is shorter than x") ||
      !need(env, speed, n, "addNodes: speedRatio is shorter than x"))
    return WG_EINVAL;
  jint *px = PIN_I(x), *py = PIN_I(y), *pe = PIN_I(extra);
  jbyte *pd = PIN_B(down), *pb = PIN_B(byzantine);
  jdouble* ps = PIN_D(speed);
  int32_t rc = wg_add_nodes(ENG(h), n, (const int32_t*)px, (const int32_t*)py, (const int32_t*)pe, (const uint8_t*)pd,
                            (const uint8_t*)pb, (const double*)ps);
  UNPIN_I(x, px); UNPIN_I(y, py); UNPIN_I(extra, pe); UNPIN_B(down, pd); UNPIN_B(byzantine, pb); UNPIN_D(speed, ps);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, nodeCount)(JNIEnv* env, jclass c, jlong h) { (void)env; (void)c; return wg_node_count(ENG(h)); }
WG_JNI(jint, setLatency)(JNIEnv* env, jclass c, jlong h, jint kind, jintArray params) {  /* setNetworkLatency :666-678 */
  (void)c;
  jint* p = PIN_I(params);
  int32_t rc = wg_set_latency(ENG(h), kind, (const int32_t*)p, LEN(params));
  UNPIN_I(params, p);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, setLatencyCity)(JNIEnv* env, jclass c, jlong h, jint mode, jint nCities, jintArray cityOfNode, jintArray tab,
                             jfloatArray ping, jdoubleArray jitter100) {  /* C/NetworkLatency.java:86-233 */
  (void)c;
  jint *pc = PIN_I(cityOfNode), *pt = PIN_I(tab);
  jfloat* pp = PIN_F(ping);
  jdouble* pj = PIN_D(jitter100);
  int32_t rc = wg_set_latency_city(ENG(h), mode, nCities, (const int32_t*)pc, (const int32_t*)pt, (const float*)pp, (const double*)pj);
  UNPIN_I(cityOfNode, pc); UNPIN_I(tab, pt); UNPIN_F(ping, pp); UNPIN_D(jitter100, pj);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, setLatencyByName)(JNIEnv* env, jclass c, jlong h, jstring name) {  /* RegistryNetworkLatencies.getByName */
  (void)c;
  const char* s = STR(name);
  int32_t rc = wg_set_latency_by_name(ENG(h), s);
  UNSTR(name, s);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, latencyProbe)(JNIEnv* env, jclass c, jlong h, jintArray from, jintArray to, jintArray delta, jintArray out) {
  (void)c;
  if (!from || !to || !delta || !out) { throw_msg(env, WG_EINVAL, "latencyProbe: four arrays"); return WG_EINVAL; }
  if (!need(env, to, LEN(from), "latencyProbe: to is shorter than from") || !need(env, delta, LEN(from), "latencyProbe: delta is shorter than from") ||
      !need(env, out, LEN(from), "latencyProbe: out is shorter than from"))
    return WG_EINVAL;
  jint *pf = PIN_I(from), *pt = PIN_I(to), *pd = PIN_I(delta), *po = PIN_I(out);
  int32_t rc = wg_latency_probe(ENG(h), LEN(from), (const int32_t*)pf, (const int32_t*)pt, (const int32_t*)pd, (int32_t*)po);
  UNPIN_I(from, pf); UNPIN_I(to, pt); UNPIN_I(delta, pd); COMMIT_I(out, po);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, setPartitions)(JNIEnv* env, jclass c, jlong h, jintArray xcuts) {  /* partition()/endPartition() :693-707 */
  (void)c;
  jint* p = PIN_I(xcuts);
  int32_t rc = wg_set_partitions(ENG(h), (const int32_t*)p, LEN(xcuts));
  UNPIN_I(xcuts, p);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, setNodeDown)(JNIEnv* env, jclass c, jlong h, jint id, jboolean down) {  /* Node.stop()/start() C/Node.java:120-131 */
  (void)c;
  return ck(env, ENG(h), wg_set_node_down(ENG(h), id, down ? 1 : 0));
}
WG_JNI(jint, setDiscardTime)(JNIEnv* env, jclass c, jlong h, jint ms) {  /* setMsgDiscardTime :103-107 */
  (void)c;
  return ck(env, ENG(h), wg_set_discard_time(ENG(h), ms));
}

/* ---- rd, the shared java.util.Random (C/Network.java:32) ---- */
WG_JNI(jint, rngSetSeed)(JNIEnv* env, jclass c, jlong h, jlong seed) { (void)c; return ck(env, ENG(h), wg_rng_set_seed(ENG(h), seed)); }
WG_JNI(jlong, rngGetState)(JNIEnv* env, jclass c, jlong h) {
  (void)c;
  uint64_t s = 0;
  ck(env, ENG(h), wg_rng_get_state(ENG(h), &s));
  return (jlong)s;
}
WG_JNI(jint, rngSetState)(JNIEnv* env, jclass c, jlong h, jlong s48) { (void)c; return ck(env, ENG(h), wg_rng_set_state(ENG(h), (uint64_t)s48)); }

/* ---- host-side sends / tasks (init() code paths) ---- */
WG_JNI(jint, send)(JNIEnv* env, jclass c, jlong h, jint msg, jint payload, jint sendTime, jint from, jintArray dests,
                   jint delayBetween) {  /* Network.send C/Network.java:369-382,418-447 */
  (void)c;
  jint* p = PIN_I(dests);
  int32_t rc = wg_send(ENG(h), (uint32_t)msg, (uint32_t)payload, sendTime, from, (const int32_t*)p, LEN(dests), delayBetween);
  UNPIN_I(dests, p);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, sendArriveAt)(JNIEnv* env, jclass c, jlong h, jint msg, jint payload, jint arriveAt, jint from, jint to) {
  (void)c;  /* Network.sendArriveAt :384-390 */
  return ck(env, ENG(h), wg_send_arrive_at(ENG(h), (uint32_t)msg, (uint32_t)payload, arriveAt, from, to));
}
WG_JNI(jint, registerTask)(JNIEnv* env, jclass c, jlong h, jint task, jint arg, jint startAt, jint node) {
  (void)c;  /* Network.registerTask :505-509 */
  return ck(env, ENG(h), wg_register_task(ENG(h), (uint32_t)task, (uint32_t)arg, startAt, node));
}
WG_JNI(jint, registerPeriodicTask)(JNIEnv* env, jclass c, jlong h, jint task, jint startAt, jint period, jint node) {
  (void)c;  /* Network.registerPeriodicTask :511-519 */
  return ck(env, ENG(h), wg_register_periodic_task(ENG(h), (uint32_t)task, startAt, period, node));
}

/* ---- resident protocols: wg_protocol_load, one native per protocol (params as int[] in the struct's field order) ---- */
WG_JNI(jint, loadHost)(JNIEnv* env, jclass c, jlong h) { (void)c; return ck(env, ENG(h), wg_protocol_load(ENG(h), WG_PROTO_HOST, NULL, NULL)); }
WG_JNI(jint, loadPingPong)(JNIEnv* env, jclass c, jlong h) { (void)c; return ck(env, ENG(h), wg_protocol_load(ENG(h), WG_PROTO_PINGPONG, NULL, NULL)); }
WG_JNI(jint, loadHandel)(JNIEnv* env, jclass c, jlong h, jintArray params, jintArray startAt, jintArray pairing,
                         jobject ranksBuf, jobject peersBuf) {  /* Handel.init() P/Handel.java:957-1014 */
  (void)c;
  if (LEN(params) != (jsize)(sizeof(wg_handel_params) / 4)) { throw_msg(env, WG_EINVAL, "HandelParameters: 14 ints"); return WG_EINVAL; }
  wg_handel_params p;
  (*env)->GetIntArrayRegion(env, params, 0, (jsize)(sizeof p / 4), (jint*)&p);
  jint *ps = PIN_I(startAt), *pp = PIN_I(pairing);
  wg_handel_init_state st;
  st.startAt = (const int32_t*)ps;
  st.nodePairingTime = (const int32_t*)pp;
  /* both null: the engine runs setReceivingRanks' shuffles (:940-948) and buildEmissionList (:510-522) itself, from the rd
   * state set with rngSetState BEFORE this call, and leaves rd advanced by the draws; WG_EHOSTINIT is returned, not thrown */
  st.receptionRanks = ranksBuf ? (const int32_t*)(*env)->GetDirectBufferAddress(env, ranksBuf) : NULL;
  st.peers = peersBuf ? (const int32_t*)(*env)->GetDirectBufferAddress(env, peersBuf) : NULL;
  int32_t rc = wg_protocol_load(ENG(h), WG_PROTO_HANDEL, &p, &st);
  UNPIN_I(startAt, ps); UNPIN_I(pairing, pp);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, loadGsf)(JNIEnv* env, jclass c, jlong h, jintArray params, jintArray pairing, jobject peersBuf) {
  (void)c;  /* GSFSignature.init() P/GSFSignature.java:611-635 */
  if (LEN(params) != (jsize)(sizeof(wg_gsf_params) / 4)) { throw_msg(env, WG_EINVAL, "GSFSignatureParameters: 7 ints"); return WG_EINVAL; }
  wg_gsf_params p;
  (*env)->GetIntArrayRegion(env, params, 0, (jsize)(sizeof p / 4), (jint*)&p);
  jint* pp = PIN_I(pairing);
  wg_gsf_init_state st;
  st.nodePairingTime = (const int32_t*)pp;
  st.peers = peersBuf ? (const int32_t*)(*env)->GetDirectBufferAddress(env, peersBuf) : NULL;
  int32_t rc = wg_protocol_load(ENG(h), WG_PROTO_GSF, &p, &st);
  UNPIN_I(pairing, pp);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, loadSanFermin)(JNIEnv* env, jclass c, jlong h, jintArray params) {  /* P/SanFerminSignature.java:84-141 */
  (void)c;
  if (LEN(params) != (jsize)(sizeof(wg_sanfermin_params) / 4)) { throw_msg(env, WG_EINVAL, "SanFerminSignatureParameters: 6 ints"); return WG_EINVAL; }
  wg_sanfermin_params p;
  (*env)->GetIntArrayRegion(env, params, 0, (jsize)(sizeof p / 4), (jint*)&p);
  return ck(env, ENG(h), wg_protocol_load(ENG(h), WG_PROTO_SANFERMIN, &p, NULL));
}
WG_JNI(jint, loadCasper)(JNIEnv* env, jclass c, jlong h, jintArray params) {  /* P/CasperIMD.java:52-70,481-509 */
  (void)c;
  if (LEN(params) != (jsize)(sizeof(wg_casper_params) / 4)) { throw_msg(env, WG_EINVAL, "CasperParemeters + byzDelay, maxSlots: 8 ints"); return WG_EINVAL; }
  wg_casper_params p;
  (*env)->GetIntArrayRegion(env, params, 0, (jsize)(sizeof p / 4), (jint*)&p);
  return ck(env, ENG(h), wg_protocol_load(ENG(h), WG_PROTO_CASPER, &p, NULL));
}
WG_JNI(jint, loadP2PFlood)(JNIEnv* env, jclass c, jlong h, jintArray params, jintArray peers, jintArray peerCount, jint maxPeers,
                           jintArray senders) {  /* P/P2PFlood.java:63-140 */
  (void)c;
  if (LEN(params) != (jsize)(sizeof(wg_p2pflood_params) / 4)) { throw_msg(env, WG_EINVAL, "P2PFloodParameters: 7 ints"); return WG_EINVAL; }
  wg_p2pflood_params p;
  (*env)->GetIntArrayRegion(env, params, 0, (jsize)(sizeof p / 4), (jint*)&p);
  jint *pp = PIN_I(peers), *pc = PIN_I(peerCount), *ps = PIN_I(senders);
  wg_p2pflood_init_state st;
  st.peers = (const int32_t*)pp;
  st.peerCount = (const int32_t*)pc;
  st.maxPeers = maxPeers;
  st.senders = (const int32_t*)ps;
  int32_t rc = wg_protocol_load(ENG(h), WG_PROTO_P2PFLOOD, &p, &st);
  UNPIN_I(peers, pp); UNPIN_I(peerCount, pc); UNPIN_I(senders, ps);
  return ck(env, ENG(h), rc);
}

/* ---- run: Network.runMs C/Network.java:318-338 ---- */
static void stats_out(JNIEnv* env, jlongArray dst, const wg_run_stats* st, int n) {
  if (dst) (*env)->SetLongArrayRegion(env, dst, 0, (jsize)(7 * n), (const jlong*)st);
}
WG_JNI(jboolean, runMs)(JNIEnv* env, jclass c, jlong h, jint ms, jlongArray stats7) {
  (void)c;
  uint8_t did = 0;
  wg_run_stats st;
  memset(&st, 0, sizeof st);
  if (ck(env, ENG(h), wg_run_ms(ENG(h), ms, &did, &st)) != WG_OK) return JNI_FALSE;
  stats_out(env, stats7, &st, 1);
  return did ? JNI_TRUE : JNI_FALSE;
}
WG_JNI(jint, time)(JNIEnv* env, jclass c, jlong h) {  /* Network.time :49 */
  (void)c;
  int32_t t = 0;
  ck(env, ENG(h), wg_time(ENG(h), &t));
  return t;
}
WG_JNI(jlong, queueSize)(JNIEnv* env, jclass c, jlong h) {  /* msgs.size() :204-210 */
  (void)c;
  int64_t s = 0;
  ck(env, ENG(h), wg_queue_size(ENG(h), &s));
  return s;
}
WG_JNI(jlong, queueSizeAt)(JNIEnv* env, jclass c, jlong h, jint t) {  /* msgs.sizeAt(t) :212-220 */
  (void)c;
  int64_t s = 0;
  ck(env, ENG(h), wg_queue_size_at(ENG(h), t, &s));
  return s;
}
WG_JNI(jboolean, protocolContIf)(JNIEnv* env, jclass c, jlong h) {  /* Handel.newContIf P/Handel.java:1044-1053, ... */
  (void)c;
  int32_t cont = 0;
  ck(env, ENG(h), wg_protocol_cont_if(ENG(h), &cont));
  return cont ? JNI_TRUE : JNI_FALSE;
}

/* ---- the init() image (C/RunMultipleTimes.java:44-48 without re-running init()) ---- */
WG_JNI(jint, snapshot)(JNIEnv* env, jclass c, jlong h) { (void)c; return ck(env, ENG(h), wg_snapshot(ENG(h))); }
WG_JNI(jint, restore)(JNIEnv* env, jclass c, jlong h) { (void)c; return ck(env, ENG(h), wg_restore(ENG(h))); }
WG_JNI(jlong, snapshotBytes)(JNIEnv* env, jclass c, jlong h) {
  (void)c;
  int64_t b = 0;
  ck(env, ENG(h), wg_snapshot_bytes(ENG(h), &b));
  return b;
}

/* ---- host-callback mode: Network.nextMessage :533-570 + the post-action part of receiveUntil :625-632 ---- */
/* out6 = {kind, time, from, to, msg, payload}; returns false once time > until */
WG_JNI(jboolean, nextDelivery)(JNIEnv* env, jclass c, jlong h, jint until, jint condTime, jintArray out6) {
  (void)c;
  wg_delivery d;
  int32_t got = 0;
  if (ck(env, ENG(h), wg_next_delivery(ENG(h), until, condTime, &d, &got)) != WG_OK || !got) return JNI_FALSE;
  jint v[6] = {d.kind, d.time, d.from, d.to, (jint)d.msg, (jint)d.payload};
  (*env)->SetIntArrayRegion(env, out6, 0, 6, v);
  return JNI_TRUE;
}
WG_JNI(jint, setTime)(JNIEnv* env, jclass c, jlong h, jint t) { (void)c; return ck(env, ENG(h), wg_set_time(ENG(h), t)); }
/* batch6: cap deliveries of 6 ints each (the layout above); returns how many were handed out */
WG_JNI(jint, stepBegin)(JNIEnv* env, jclass c, jlong h, jint until, jint condTime, jintArray batch6) {
  (void)c;
  jint* p = PIN_I(batch6);
  int32_t n = 0;
  int32_t rc = wg_step_begin(ENG(h), until, condTime, (wg_delivery*)p, LEN(batch6) / 6, &n);  /* wg_delivery IS six 32-bit words */
  COMMIT_I(batch6, p);
  return ck(env, ENG(h), rc) == WG_OK ? n : -1;
}
/* ops10: nops records of 10 ints in wg_step_op's field order; dests: the destination lists the ops index into */
WG_JNI(jint, stepEnd)(JNIEnv* env, jclass c, jlong h, jintArray ops10, jint nops, jintArray dests) {
  (void)c;
  if (nops < 0 || (nops > 0 && !ops10) || !need(env, ops10, (jlong)nops * 10, "stepEnd: ops10 holds fewer than nops records"))
    { if (nops < 0 || (nops > 0 && !ops10)) throw_msg(env, WG_EINVAL, "stepEnd: nops"); return WG_EINVAL; }
  jint *po = PIN_I(ops10), *pd = PIN_I(dests);
  const jlong nd = LEN(dests);
  for (jint i = 0; i < nops; i++) {  /* a list send's destinations lie inside `dests` (wg_step_op: to = offset, n = count) */
    const jint kind = po[i * 10 + 1], to = po[i * 10 + 6], n = po[i * 10 + 7];
    if (kind == WG_OP_SEND && n > 1 && (to < 0 || (jlong)to + n > nd)) {
      UNPIN_I(ops10, po); UNPIN_I(dests, pd);
      throw_msg(env, WG_EINVAL, "stepEnd: an op's destination list lies outside dests");
      return WG_EINVAL;
    }
  }
  int32_t rc = wg_step_end(ENG(h), (const wg_step_op*)po, nops, (const int32_t*)pd);  /* wg_step_op IS ten 32-bit words */
  UNPIN_I(ops10, po); UNPIN_I(dests, pd);
  return ck(env, ENG(h), rc);
}

WG_JNI(jint, hostReleased)(JNIEnv* env, jclass c, jlong h, jintArray out) {
  (void)c;
  if (!out) { throw_msg(env, WG_EINVAL, "hostReleased: out is required"); return -1; }
  jint* p = PIN_I(out);
  int32_t n = 0;
  int32_t rc = wg_host_released(ENG(h), (uint32_t*)p, LEN(out), &n);
  COMMIT_I(out, p);
  return ck(env, ENG(h), rc) == WG_OK ? n : -1;
}

/* ---- batches: RunMultipleTimes on the device (C/RunMultipleTimes.java:44-64) ---- */
WG_JNI(jlong, batchCreate)(JNIEnv* env, jclass c, jlongArray handles) {
  (void)c;
  jsize n = LEN(handles);
  jlong* ph = PIN_J(handles);
  wg_engine* eng[1024];
  if (n > 1024) { UNPIN_J(handles, ph); throw_msg(env, WG_EINVAL, "at most 1024 members"); return 0; }
  for (jsize i = 0; i < n; i++) eng[i] = ENG(ph[i]);
  UNPIN_J(handles, ph);
  wg_batch* b = NULL;
  int32_t rc = wg_batch_create(eng, n, &b);
  if (rc != WG_OK) { throw_msg(env, rc, wg_batch_last_error(NULL)); return 0; }
  return (jlong)(intptr_t)b;
}
WG_JNI(void, batchDestroy)(JNIEnv* env, jclass c, jlong b) { (void)env; (void)c; wg_batch_destroy(BAT(b)); }
WG_JNI(jstring, batchLastError)(JNIEnv* env, jclass c, jlong b) { (void)c; return (*env)->NewStringUTF(env, wg_batch_last_error(BAT(b))); }
WG_JNI(jint, batchSize)(JNIEnv* env, jclass c, jlong b) {
  (void)c;
  return (jint)batch_members(env, BAT(b));
}
WG_JNI(jint, batchRunMs)(JNIEnv* env, jclass c, jlong b, jint ms, jbyteArray active, jbyteArray didSomething, jlongArray stats7n) {
  (void)c;
  const jlong n = batch_members(env, BAT(b));
  if (n < 0 || !need(env, active, n, "batchRunMs: active is shorter than the batch") ||
      !need(env, didSomething, n, "batchRunMs: didSomething is shorter than the batch") ||
      !need(env, stats7n, 7 * n, "batchRunMs: stats7n holds fewer than 7 longs per member"))
    return WG_EINVAL;
  jbyte *pa = PIN_B(active), *pd = PIN_B(didSomething);
  jlong* ps = PIN_J(stats7n);
  int32_t rc = wg_batch_run_ms(BAT(b), ms, (const uint8_t*)pa, (uint8_t*)pd, (wg_run_stats*)ps);
  UNPIN_B(active, pa); COMMIT_B(didSomething, pd); COMMIT_J(stats7n, ps);
  return ck_batch(env, BAT(b), rc);
}
WG_JNI(jint, batchContIf)(JNIEnv* env, jclass c, jlong b, jintArray cont) {
  (void)c;
  const jlong n = batch_members(env, BAT(b));
  if (n < 0 || !cont || !need(env, cont, n, "batchContIf: cont is shorter than the batch")) {
    if (n >= 0 && !cont) throw_msg(env, WG_EINVAL, "batchContIf: cont is required");
    return WG_EINVAL;
  }
  jint* p = PIN_I(cont);
  int32_t rc = wg_batch_cont_if(BAT(b), (int32_t*)p);
  COMMIT_I(cont, p);
  return ck_batch(env, BAT(b), rc);
}
WG_JNI(jint, batchRunMultipleTimes)(JNIEnv* env, jclass c, jlong b, jint chunk, jint maxTime, jlongArray delivered, jlongArray simulatedMs) {
  (void)c;
  const jlong n = batch_members(env, BAT(b));
  if (n < 0 || !need(env, delivered, n, "batchRunMultipleTimes: delivered is shorter than the batch") ||
      !need(env, simulatedMs, n, "batchRunMultipleTimes: simulatedMs is shorter than the batch"))
    return WG_EINVAL;
  jlong *pd = PIN_J(delivered), *ps = PIN_J(simulatedMs);
  int32_t rc = wg_batch_run_multiple_times(BAT(b), chunk, maxTime, (int64_t*)pd, (int64_t*)ps);
  COMMIT_J(delivered, pd); COMMIT_J(simulatedMs, ps);
  return ck_batch(env, BAT(b), rc);
}

/* ---- node-range sharding (no reference counterpart: C/Network.java:7-11 is single-threaded) ---- */
WG_JNI(jbyteArray, rcclUniqueId)(JNIEnv* env, jclass c) {
  (void)c;
  uint8_t id[WG_RCCL_UNIQUE_ID_BYTES];
  int32_t rc = wg_rccl_unique_id(id);
  if (rc != WG_OK) { throw_msg(env, rc, wg_last_error(NULL)); return NULL; }
  jbyteArray a = (*env)->NewByteArray(env, WG_RCCL_UNIQUE_ID_BYTES);
  if (a) (*env)->SetByteArrayRegion(env, a, 0, WG_RCCL_UNIQUE_ID_BYTES, (const jbyte*)id);
  return a;
}
WG_JNI(jint, shardConfigureRccl)(JNIEnv* env, jclass c, jlong h, jint shard, jint nshards, jbyteArray id) {
  (void)c;
  jbyte buf[WG_RCCL_UNIQUE_ID_BYTES];
  if (LEN(id) != WG_RCCL_UNIQUE_ID_BYTES) { throw_msg(env, WG_EINVAL, "the RCCL unique id is 128 bytes"); return WG_EINVAL; }
  (*env)->GetByteArrayRegion(env, id, 0, WG_RCCL_UNIQUE_ID_BYTES, buf);
  return ck(env, ENG(h), wg_shard_configure_rccl(ENG(h), shard, nshards, (const uint8_t*)buf));
}
/* the caller-supplied collective (wg_shard_configure's function pointer) is a native-code hook: a Java host uses the
 * engine-owned RCCL communicator above; this entry takes the address of a wg_allreduce_fn and its context from native glue */
WG_JNI(jint, shardConfigure)(JNIEnv* env, jclass c, jlong h, jint shard, jint nshards, jlong fnAddr, jlong ctxAddr) {
  (void)c;
  return ck(env, ENG(h), wg_shard_configure(ENG(h), shard, nshards, (wg_allreduce_fn)(intptr_t)fnAddr, (void*)(intptr_t)ctxAddr));
}
/* the optional all-to-all of a caller-supplied collective (wg_shard_set_alltoallv): a native-code hook like shardConfigure's;
 * an engine that owns its RCCL communicator (shardConfigureRccl) needs none */
WG_JNI(jint, shardSetAlltoallv)(JNIEnv* env, jclass c, jlong h, jlong fnAddr, jlong ctxAddr) {
  (void)c;
  return ck(env, ENG(h), wg_shard_set_alltoallv(ENG(h), (wg_alltoallv_fn)(intptr_t)fnAddr, (void*)(intptr_t)ctxAddr));
}
/* calls8 / words8: the totals of shardInfo by exchange (wg_shard_traffic) */
WG_JNI(jint, shardTraffic)(JNIEnv* env, jclass c, jlong h, jlongArray calls8, jlongArray words8) {
  (void)c;
  if (!need(env, calls8, 8, "calls8") || !need(env, words8, 8, "words8")) return WG_EINVAL;
  int64_t cv[8] = {0}, wv[8] = {0};
  int32_t rc = wg_shard_traffic(ENG(h), cv, wv);
  jlong a[8], b[8];
  for (int k = 0; k < 8; k++) {
    a[k] = cv[k];
    b[k] = wv[k];
  }
  if (calls8) (*env)->SetLongArrayRegion(env, calls8, 0, 8, a);
  if (words8) (*env)->SetLongArrayRegion(env, words8, 0, 8, b);
  return ck(env, ENG(h), rc);
}
/* out4 = {lo, hi, collectives, words} */
WG_JNI(jint, shardInfo)(JNIEnv* env, jclass c, jlong h, jlongArray out4) {
  (void)c;
  int32_t lo = 0, hi = 0;
  int64_t col = 0, words = 0;
  int32_t rc = wg_shard_info(ENG(h), &lo, &hi, &col, &words);
  jlong v[4] = {lo, hi, col, words};
  if (out4) (*env)->SetLongArrayRegion(env, out4, 0, 4, v);
  return ck(env, ENG(h), rc);
}

/* ---- read-back ---- */
WG_JNI(jint, readI64)(JNIEnv* env, jclass c, jlong h, jint field, jlongArray dst) {
  (void)c;
  jlong* p = PIN_J(dst);
  int32_t rc = wg_read_i64(ENG(h), field, (int64_t*)p, LEN(dst));
  COMMIT_J(dst, p);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, readI32)(JNIEnv* env, jclass c, jlong h, jint field, jintArray dst) {
  (void)c;
  jint* p = PIN_I(dst);
  int32_t rc = wg_read_i32(ENG(h), field, (int32_t*)p, LEN(dst));
  COMMIT_I(dst, p);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, readLevelI32)(JNIEnv* env, jclass c, jlong h, jint field, jintArray dst, jint nNodes, jint nLevels) {
  (void)c;
  if (!dst || nNodes < 0 || nLevels < 0 || !need(env, dst, (jlong)nNodes * nLevels, "readLevelI32: dst holds fewer than nNodes * nLevels ints")) {
    if (!dst || nNodes < 0 || nLevels < 0) throw_msg(env, WG_EINVAL, "readLevelI32: dst / shape");
    return WG_EINVAL;
  }
  jint* p = PIN_I(dst);
  int32_t rc = wg_read_level_i32(ENG(h), field, (int32_t*)p, nNodes, nLevels);
  COMMIT_I(dst, p);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, readBits)(JNIEnv* env, jclass c, jlong h, jint field, jlongArray dst, jint nNodes, jint wordsPerNode) {
  (void)c;
  if (!dst || nNodes < 0 || wordsPerNode < 0 || !need(env, dst, (jlong)nNodes * wordsPerNode, "readBits: dst holds fewer than nNodes * wordsPerNode longs")) {
    if (!dst || nNodes < 0 || wordsPerNode < 0) throw_msg(env, WG_EINVAL, "readBits: dst / shape");
    return WG_EINVAL;
  }
  jlong* p = PIN_J(dst);
  int32_t rc = wg_read_bits(ENG(h), field, (uint64_t*)p, nNodes, wordsPerNode);
  COMMIT_J(dst, p);
  return ck(env, ENG(h), rc);
}
WG_JNI(jint, levels)(JNIEnv* env, jclass c, jlong h) {
  (void)c;
  int32_t l = 0;
  ck(env, ENG(h), wg_levels(ENG(h), &l));
  return l;
}
WG_JNI(jlong, deviceBytes)(JNIEnv* env, jclass c, jlong h) {
  (void)c;
  int64_t b = 0;
  ck(env, ENG(h), wg_device_bytes(ENG(h), &b));
  return b;
}
WG_JNI(jint, deliveredByLevel)(JNIEnv* env, jclass c, jlong h, jlongArray dst32) {
  (void)c;
  if (LEN(dst32) < 32) { throw_msg(env, WG_EINVAL, "dst: 32 longs"); return WG_EINVAL; }
  jlong* p = PIN_J(dst32);
  int32_t rc = wg_delivered_by_level(ENG(h), (int64_t*)p);
  COMMIT_J(dst32, p);
  return ck(env, ENG(h), rc);
}

/* ---- measurement (no reference counterpart) ---- */
WG_JNI(jint, profileEnable)(JNIEnv* env, jclass c, jlong h, jint mode) { (void)c; return ck(env, ENG(h), wg_profile_enable(ENG(h), mode)); }
/* names[k], spans[k], totalNs[k] for the first min(cap, n) phases; returns n */
WG_JNI(jint, profileRead)(JNIEnv* env, jclass c, jlong h, jobjectArray names, jlongArray spans, jdoubleArray totalNs) {
  (void)c;
  wg_profile_entry ent[32];
  int32_t n = 0;
  if (ck(env, ENG(h), wg_profile_read(ENG(h), ent, 32, &n)) != WG_OK) return -1;
  jsize cap = LEN(spans);
  for (jsize k = 0; k < n && k < cap && k < 32; k++) {
    jlong s = ent[k].spans;
    jdouble t = ent[k].total_ns;
    (*env)->SetLongArrayRegion(env, spans, k, 1, &s);
    (*env)->SetDoubleArrayRegion(env, totalNs, k, 1, &t);
    if (names) (*env)->SetObjectArrayElement(env, names, k, (*env)->NewStringUTF(env, ent[k].name));
  }
  return n;
}
WG_JNI(jint, profileSetReference)(JNIEnv* env, jclass c, jlong h, jlong ref) { (void)c; return ck(env, ENG(h), wg_profile_set_reference(ENG(h), ENG(ref))); }
WG_JNI(jint, profileReadSpans)(JNIEnv* env, jclass c, jlong h, jint cls, jdoubleArray startNs, jdoubleArray endNs) {
  (void)c;
  jdouble *ps = PIN_D(startNs), *pe = PIN_D(endNs);
  int32_t n = 0;
  int32_t rc = wg_profile_read_spans(ENG(h), cls, (double*)ps, (double*)pe, LEN(startNs), &n);
  COMMIT_D(startNs, ps); COMMIT_D(endNs, pe);
  return ck(env, ENG(h), rc) == WG_OK ? n : -1;
}

/* ---- include/wittgpu_host.h: the C++ mirrors of Protocol.init() (a Java host runs its own init(); these serve hosts
 * that want the engine's restatement — they are what this repo's tests use) ---- */
#define HOST_CREATE(NAME, STRUCT, CALL)                                                                              \
  WG_JNI(jlong, NAME)(JNIEnv* env, jclass c, jintArray params, jstring nb, jstring nl, jlong seed, jintArray cfgInts, \
                      jlongArray cfgLongs, jbyteArray rcclId) {                                                      \
    (void)c;                                                                                                         \
    if (LEN(params) != (jsize)(sizeof(STRUCT) / 4)) { throw_msg(env, WG_EINVAL, #STRUCT ": wrong parameter count"); return 0; } \
    STRUCT p;                                                                                                        \
    (*env)->GetIntArrayRegion(env, params, 0, (jsize)(sizeof p / 4), (jint*)&p);                                     \
    wg_config cfg;                                                                                                   \
    jbyte id[WG_RCCL_UNIQUE_ID_BYTES];                                                                               \
    fill_config(env, &cfg, cfgInts, cfgLongs, id, rcclId);                                                           \
    const char *snb = STR(nb), *snl = STR(nl);                                                                       \
    wg_engine* e = NULL;                                                                                             \
    int32_t rc = CALL(&p, snb, snl, seed, &cfg, &e);                                                                 \
    UNSTR(nb, snb); UNSTR(nl, snl);                                                                                  \
    if (ck_host(env, rc) != WG_OK) return 0;                                                                         \
    return (jlong)(intptr_t)e;                                                                                       \
  }
HOST_CREATE(hostHandelCreate, wg_handel_params, wgh_handel_create)
HOST_CREATE(hostGsfCreate, wg_gsf_params, wgh_gsf_create)
HOST_CREATE(hostSanFerminCreate, wg_sanfermin_params, wgh_sanfermin_create)
HOST_CREATE(hostCasperCreate, wg_casper_params, wgh_casper_create)
HOST_CREATE(hostP2PFloodCreate, wg_p2pflood_params, wgh_p2pflood_create)
/* Handel with HandelParameters.badNodes (P/Handel.java:51,110,139): badNodes[nodeCount], non-zero = the BitSet's bit */
WG_JNI(jlong, hostHandelCreateBadNodes)(JNIEnv* env, jclass c, jintArray params, jbyteArray badNodes, jstring nb, jstring nl, jlong seed,
                                        jintArray cfgInts, jlongArray cfgLongs, jbyteArray rcclId) {
  (void)c;
  if (LEN(params) != (jsize)(sizeof(wg_handel_params) / 4)) { throw_msg(env, WG_EINVAL, "HandelParameters: 14 ints"); return 0; }
  wg_handel_params p;
  (*env)->GetIntArrayRegion(env, params, 0, (jsize)(sizeof p / 4), (jint*)&p);
  if (badNodes && LEN(badNodes) != p.nodeCount) { throw_msg(env, WG_EINVAL, "badNodes: one byte per node"); return 0; }
  wg_config cfg;
  jbyte id[WG_RCCL_UNIQUE_ID_BYTES];
  fill_config(env, &cfg, cfgInts, cfgLongs, id, rcclId);
  const char *snb = STR(nb), *snl = STR(nl);
  jbyte* pb = PIN_B(badNodes);
  wg_engine* e = NULL;
  int32_t rc = wgh_handel_create_bad_nodes(&p, (const uint8_t*)pb, snb, snl, seed, &cfg, &e);
  UNPIN_B(badNodes, pb); UNSTR(nb, snb); UNSTR(nl, snl);
  if (ck_host(env, rc) != WG_OK) return 0;
  return (jlong)(intptr_t)e;
}
WG_JNI(jlong, hostPingPongCreate)(JNIEnv* env, jclass c, jint nodeCt, jstring nb, jstring nl, jlong seed, jintArray cfgInts,
                                  jlongArray cfgLongs, jbyteArray rcclId) {  /* PingPong.init() P/PingPong.java:81-87 */
  (void)c;
  wg_config cfg;
  jbyte id[WG_RCCL_UNIQUE_ID_BYTES];
  fill_config(env, &cfg, cfgInts, cfgLongs, id, rcclId);
  const char *snb = STR(nb), *snl = STR(nl);
  wg_engine* e = NULL;
  int32_t rc = wgh_pingpong_create(nodeCt, snb, snl, seed, &cfg, &e);
  UNSTR(nb, snb); UNSTR(nl, snl);
  if (ck_host(env, rc) != WG_OK) return 0;
  return (jlong)(intptr_t)e;
}
WG_JNI(jint, registerCityBuilder)(JNIEnv* env, jclass c, jstring site, jfloatArray cum, jintArray mercX, jintArray mercY, jint listSize) {
  (void)c;  /* NodeBuilderWithCity's citiesInfo, C/NodeBuilder.java:98-147 */
  const char* s = STR(site);
  jfloat* pc = PIN_F(cum);
  jint *px = PIN_I(mercX), *py = PIN_I(mercY);
  int32_t rc = wgh_register_city_builder(s, LEN(cum), (const float*)pc, (const int32_t*)px, (const int32_t*)py, listSize);
  UNSTR(site, s); UNPIN_F(cum, pc); UNPIN_I(mercX, px); UNPIN_I(mercY, py);
  return ck_host(env, rc);
}
WG_JNI(jint, registerCityLatency)(JNIEnv* env, jclass c, jstring name, jint mode, jint nCities, jintArray tab, jfloatArray ping,
                                  jdoubleArray jitter100) {  /* C/NetworkLatency.java:86-233 */
  (void)c;
  const char* s = STR(name);
  jint* pt = PIN_I(tab);
  jfloat* pp = PIN_F(ping);
  jdouble* pj = PIN_D(jitter100);
  int32_t rc = wgh_register_city_latency(s, mode, nCities, (const int32_t*)pt, (const float*)pp, (const double*)pj);
  UNSTR(name, s); UNPIN_I(tab, pt); UNPIN_F(ping, pp); UNPIN_D(jitter100, pj);
  return ck_host(env, rc);
}
WG_JNI(jstring, hostLastError)(JNIEnv* env, jclass c) { (void)c; return (*env)->NewStringUTF(env, wgh_last_error()); }
WG_JNI(jdouble, hostLastInitSeconds)(JNIEnv* env, jclass c) { (void)env; (void)c; return wgh_last_init_seconds(); }
WG_JNI(jboolean, hostLastInitOnDevice)(JNIEnv* env, jclass c) { (void)env; (void)c; return wgh_last_init_on_device() ? JNI_TRUE : JNI_FALSE; }
/* java.util.Random known answers of the engine's own generator (a host can check them against its JDK at start-up) */
WG_JNI(jint, jrandomInts)(JNIEnv* env, jclass c, jlong seed, jintArray out) {
  (void)c;
  jint* p = PIN_I(out);
  int32_t rc = wgh_jrandom_ints(seed, LEN(out), (int32_t*)p);
  COMMIT_I(out, p);
  return ck_host(env, rc);
}
WG_JNI(jint, jrandomSkipInts)(JNIEnv* env, jclass c, jlong seed, jintArray out) {
  (void)c;
  jint* p = PIN_I(out);
  int32_t rc = wgh_jrandom_skip_ints(seed, LEN(out), (int32_t*)p);
  COMMIT_I(out, p);
  return ck_host(env, rc);
}
WG_JNI(jint, jrandomBounded)(JNIEnv* env, jclass c, jlong seed, jint bound, jintArray out) {
  (void)c;
  jint* p = PIN_I(out);
  int32_t rc = wgh_jrandom_bounded(seed, bound, LEN(out), (int32_t*)p);
  COMMIT_I(out, p);
  return ck_host(env, rc);
}
