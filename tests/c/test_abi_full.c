/* The C ABI of include/wittgpu.h / wittgpu_host.h driven from plain C — no Python, no ctypes: what a JNI shim
 * (jni/wittgpu_jni.c) or any other host calls. TEST INFRASTRUCTURE ONLY; run by tests/test_abi_full_c.py against
 * libwittgpu.so on the MI355X (-m gpu) and against the CPU wave-emulator build of the same sources (-m "not gpu").
 *   1. version / struct-size handshake (wg_abi_version, wg_abi_struct_size);
 *   2. resident Handel, 256 nodes (P/Handel.java; HandelParameters as bench.py's ratios): wgh_handel_create, the
 *      RunMultipleTimes loop `do runMs(10) while contIf` (C/RunMultipleTimes.java:50-64) through wg_run_ms /
 *      wg_protocol_cont_if, read-back through wg_read_i64 / wg_read_i32 / wg_read_bits / wg_read_level_i32; the digest
 *      printed on stdout is compared with the CPU oracle's by the Python harness;
 *   3. the same run a second time from the init() image (wg_snapshot / wg_restore): identical digest;
 *   4. host-callback mode, batched (wg_step_begin / wg_step_end): a four-node ping-pong written in C against the
 *      delivery / op records — node 0 sends "ping" to all, each receiver answers "pong" to the sender (P/PingPong.java:
 *      20-32 in 20 lines of C), seeds drawn by the caller from the rd state it holds while the step is open. */
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/wittgpu.h"
#include "../../include/wittgpu_host.h"

#define CHECK(cond, ...)                                    \
  do {                                                      \
    if (!(cond)) {                                          \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);  \
      fprintf(stderr, __VA_ARGS__);                         \
      fprintf(stderr, "\n");                                \
      exit(1);                                              \
    }                                                       \
  } while (0)
#define OK(e, call) CHECK((call) == WG_OK, "%s -> %s", #call, wg_last_error(e))

/* java.util.Random.next(32) on a 48-bit state (the caller's rd while a step is open) */
static int32_t jnext32(uint64_t* s) {
  *s = (*s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
  return (int32_t)(*s >> 16);
}

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ULL;
  return h;
}

static void handel_digest(wg_engine* e, int N, const char* tag) {
  int64_t* v = (int64_t*)malloc(8 * (size_t)N);
  int32_t* v32 = (int32_t*)malloc(4 * (size_t)N);
  const int fields[] = {WG_F_DONE_AT, WG_F_MSG_RECEIVED, WG_F_MSG_SENT, WG_F_BYTES_SENT, WG_F_BYTES_RECEIVED,
                        WG_F_SIGS_CHECKED, WG_F_SIG_QUEUE_SIZE, WG_F_MSG_FILTERED, WG_F_CURR_WINDOW_SIZE};
  const char* names[] = {"doneAt", "msgReceived", "msgSent", "bytesSent", "bytesReceived", "sigsChecked", "sigQueueSize",
                         "msgFiltered", "currWindowSize"};
  for (int f = 0; f < 9; f++) {
    OK(e, wg_read_i64(e, fields[f], v, N));
    int64_t sum = 0;
    for (int i = 0; i < N; i++) sum += v[i];
    printf("%s.%s.sum=%" PRId64 "\n", tag, names[f], sum);
    printf("%s.%s.fnv=%016" PRIx64 "\n", tag, names[f], fnv(0xCBF29CE484222325ULL, v, 8 * (size_t)N));
  }
  /* wg_read_i32 = the same field narrowed */
  OK(e, wg_read_i32(e, WG_F_SIGS_CHECKED, v32, N));
  OK(e, wg_read_i64(e, WG_F_SIGS_CHECKED, v, N));
  for (int i = 0; i < N; i++) CHECK(v32[i] == (int32_t)v[i], "wg_read_i32 != wg_read_i64 at node %d", i);
  int32_t L = 0;
  OK(e, wg_levels(e, &L));
  const int W = N >= 64 ? N / 64 : 1;
  uint64_t* bits = (uint64_t*)malloc(8 * (size_t)N * W);
  OK(e, wg_read_bits(e, WG_B_TOTAL_INCOMING, bits, N, W));
  printf("%s.totalIncoming.fnv=%016" PRIx64 "\n", tag, fnv(0xCBF29CE484222325ULL, bits, 8 * (size_t)N * W));
  int32_t* lv = (int32_t*)malloc(4 * (size_t)N * L);
  OK(e, wg_read_level_i32(e, WG_LF_POS_IN_LEVEL, lv, N, L));
  printf("%s.posInLevel.fnv=%016" PRIx64 "\n", tag, fnv(0xCBF29CE484222325ULL, lv, 4 * (size_t)N * L));
  int32_t t = 0;
  uint64_t rng = 0;
  int64_t q = 0;
  OK(e, wg_time(e, &t));
  OK(e, wg_rng_get_state(e, &rng));
  OK(e, wg_queue_size(e, &q));
  printf("%s.time=%d\n%s.rng=%012" PRIx64 "\n%s.levels=%d\n", tag, t, tag, rng, tag, L);
  free(v);
  free(v32);
  free(bits);
  free(lv);
}

static int64_t handel_run(wg_engine* e) { /* C/RunMultipleTimes.java:50-64 */
  int64_t delivered = 0;
  int32_t cont = 1;
  for (int k = 0; k < 2000 && cont; k++) {
    uint8_t did = 0;
    wg_run_stats st;
    OK(e, wg_run_ms(e, 10, &did, &st));
    delivered += st.delivered;
    OK(e, wg_protocol_cont_if(e, &cont));
    if (!did) cont = 1; /* (!didSomething || contIf) */
  }
  CHECK(!cont, "Handel did not converge");
  return delivered;
}

int main(void) {
  /* 1. handshake */
  CHECK(wg_abi_version() == WG_ABI_VERSION, "ABI version %d, header %d", wg_abi_version(), WG_ABI_VERSION);
  const size_t sizes[9] = {sizeof(wg_config), sizeof(wg_handel_params), sizeof(wg_gsf_params), sizeof(wg_casper_params),
                           sizeof(wg_sanfermin_params), sizeof(wg_p2pflood_params), sizeof(wg_delivery), sizeof(wg_step_op),
                           sizeof(wg_run_stats)};
  for (int k = 0; k < 9; k++) CHECK(wg_abi_struct_size(k) == (int32_t)sizes[k], "struct %d: library %d, header %zu", k, wg_abi_struct_size(k), sizes[k]);
  CHECK(wg_abi_struct_size(99) == -1, "unknown struct index");

  /* 2. resident Handel, 256 nodes */
  const int N = 256;
  wg_handel_params hp;
  memset(&hp, 0, sizeof hp);
  hp.nodeCount = N;
  hp.nodesDown = N / 10;
  hp.threshold = (int)((N - hp.nodesDown) * 0.99);
  hp.pairingTime = 4;
  hp.levelWaitTime = 50;
  hp.extraCycle = 10;
  hp.disseminationPeriodMs = 20;
  hp.fastPath = 10;
  hp.windowInitial = 16;
  hp.windowMinimum = 1;
  hp.windowMaximum = 128;
  wg_engine* e = NULL;
  CHECK(wgh_handel_create(&hp, NULL, NULL, 0, NULL, &e) == WG_OK, "wgh_handel_create: %s", wgh_last_error());
  CHECK(wg_node_count(e) == N, "node count");
  OK(e, wg_snapshot(e));
  int64_t img = 0;
  OK(e, wg_snapshot_bytes(e, &img));
  CHECK(img > 0, "no init() image");
  const int64_t d1 = handel_run(e);
  printf("handel.delivered=%" PRId64 "\n", d1);
  handel_digest(e, N, "handel");
  int64_t* done = (int64_t*)malloc(8 * (size_t)N);
  int64_t* down = (int64_t*)malloc(8 * (size_t)N);
  OK(e, wg_read_i64(e, WG_F_DONE_AT, done, N));
  OK(e, wg_read_i64(e, WG_F_DOWN, down, N));
  int nd = 0;
  for (int i = 0; i < N; i++) {
    nd += down[i] != 0;
    CHECK(down[i] || done[i] > 0, "live node %d is not done (PT/HandelTest.java:36-49)", i);
  }
  CHECK(nd == hp.nodesDown, "%d nodes down, %d asked", nd, hp.nodesDown);
  /* argument errors map to the reference's exception classes: runMs(0) -> IllegalArgumentException (C/Network.java:320) */
  uint8_t did = 0;
  CHECK(wg_run_ms(e, 0, &did, NULL) == WG_EINVAL, "runMs(0) must be WG_EINVAL");
  CHECK(strlen(wg_last_error(e)) > 0, "no error text");

  /* 3. the same run from the init() image */
  OK(e, wg_restore(e));
  const int64_t d2 = handel_run(e);
  CHECK(d1 == d2, "restored run delivered %" PRId64 ", first run %" PRId64, d2, d1);
  handel_digest(e, N, "restored");
  wg_destroy(e);
  free(done);
  free(down);

  /* 4. host-callback mode, one ms per crossing */
  wg_engine* h = NULL;
  wg_config cfg;
  memset(&cfg, 0, sizeof cfg);
  CHECK(wg_create(&cfg, &h) == WG_OK, "wg_create: %s", wg_last_error(NULL));
  const int32_t xs[4] = {100, 600, 1200, 1900}, ys[4] = {100, 500, 900, 300};
  OK(h, wg_set_latency_by_name(h, NULL)); /* NetworkLatencyByDistanceWJitter */
  OK(h, wg_add_nodes(h, 4, xs, ys, NULL, NULL, NULL, NULL));
  OK(h, wg_rng_set_seed(h, 0));
  OK(h, wg_protocol_load(h, WG_PROTO_HOST, NULL, NULL));
  const int32_t all[4] = {0, 1, 2, 3};
  enum { PING = 1, PONG = 2 };
  OK(h, wg_send(h, PING, 0, 1, 0, all, 4, 0)); /* network.sendAll(new Ping(), node0)  P/PingPong.java:82-87 */
  int pings = 0, pongs = 0, steps = 0;
  int32_t until = 1000;
  for (;;) {
    wg_delivery batch[16];
    int32_t n = 0;
    OK(h, wg_step_begin(h, until, INT32_MAX, batch, 16, &n));
    if (n == 0) break;
    steps++;
    uint64_t rd = 0;
    OK(h, wg_rng_get_state(h, &rd)); /* rd is the caller's while the step is open */
    wg_step_op ops[16];
    int nops = 0;
    for (int i = 0; i < n; i++) {
      CHECK(batch[i].kind == 0, "unexpected delivery kind %d", batch[i].kind);
      if (batch[i].msg == PING) { /* Ping.action: network.send(new Pong(), to, from) */
        pings++;
        wg_step_op* o = &ops[nops++];
        memset(o, 0, sizeof *o);
        o->after = i;
        o->kind = WG_OP_SEND;
        o->msg = PONG;
        o->time = batch[i].time + 1; /* sendTime = time + 1 (:365) */
        o->from = batch[i].to;
        o->to = batch[i].from;
        o->n = 1;
        o->seed = jnext32(&rd); /* Network.send's rd.nextInt() (:377), drawn in action() order */
      } else {
        CHECK(batch[i].msg == PONG && batch[i].to == 0, "a pong for node %d", batch[i].to);
        pongs++;
      }
    }
    OK(h, wg_rng_set_state(h, rd));
    OK(h, wg_step_end(h, ops, nops, NULL));
  }
  OK(h, wg_set_time(h, until));
  int64_t q = -1;
  OK(h, wg_queue_size(h, &q));
  CHECK(pings == 4 && pongs == 4 && q == 0, "pings %d pongs %d queue %" PRId64, pings, pongs, q);
  printf("hostmode.pings=%d\nhostmode.pongs=%d\nhostmode.steps=%d\n", pings, pongs, steps);
  wg_destroy(h);
  printf("OK\n");
  return 0;
}
