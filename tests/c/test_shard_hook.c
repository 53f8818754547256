/* The collective hook of the C ABI (wg_allreduce_fn, include/wittgpu.h "node-range sharding") used from plain C — no
 * Python, no torch: two shards of one PingPong simulation (P/PingPong.java) as two engines on two pthreads, whose
 * per-ms sums meet in a barrier-and-add callback, against the unsharded engine of the same seed. Every per-node
 * counter, the pong counts, rd and the clock must agree. Built against the CPU wave-emulator build of the product's
 * sources (tests/emu: "device" buffers are host memory, so the callback can add them in place); run by
 * tests/test_c_shard_hook.py. TEST INFRASTRUCTURE ONLY. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/wittgpu.h"
#include "../../include/wittgpu_host.h"

#define K 2
static pthread_barrier_t bar;
static struct { int32_t* buf; int64_t count; } slot[K];

static int32_t allreduce(void* ctx, void* buf, int64_t count) {
  const int me = (int)(intptr_t)ctx;
  slot[me].buf = (int32_t*)buf;
  slot[me].count = count;
  const int leader = pthread_barrier_wait(&bar) == PTHREAD_BARRIER_SERIAL_THREAD;
  int rc = 0;
  if (leader) {
    for (int s = 1; s < K; s++)
      if (slot[s].count != slot[0].count) rc = 1;  /* shards must agree on the size of every collective */
    if (!rc)
      for (int64_t i = 0; i < count; i++) {
        int32_t t = 0;
        for (int s = 0; s < K; s++) t += slot[s].buf[i];
        for (int s = 0; s < K; s++) slot[s].buf[i] = t;
      }
  }
  pthread_barrier_wait(&bar);
  return rc;
}

static wg_engine* eng[K];
static int fail[K];
static void* run_shard(void* arg) {
  const int s = (int)(intptr_t)arg;
  for (int step = 0; step < 6; step++) {  /* P/PingPong.java:94-101: runMs(50) steps */
    uint8_t did = 0;
    if (wg_run_ms(eng[s], 50, &did, NULL) != WG_OK) {
      fprintf(stderr, "shard %d: %s\n", s, wg_last_error(eng[s]));
      fail[s] = 1;
      break;
    }
  }
  return NULL;
}

int main(void) {
  const int N = 300;
  pthread_barrier_init(&bar, NULL, K);
  for (int s = 0; s < K; s++) {
    wg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.shard = s;
    cfg.nshards = K;
    cfg.allreduce = allreduce;
    cfg.allreduce_ctx = (void*)(intptr_t)s;
    if (wgh_pingpong_create(N, NULL, NULL, 5, &cfg, &eng[s]) != WG_OK) {
      fprintf(stderr, "create: %s\n", wgh_last_error());
      return 2;
    }
  }
  wg_engine* ref = NULL;
  if (wgh_pingpong_create(N, NULL, NULL, 5, NULL, &ref) != WG_OK) return 2;
  pthread_t th[K];
  for (int s = 0; s < K; s++) pthread_create(&th[s], NULL, run_shard, (void*)(intptr_t)s);
  for (int s = 0; s < K; s++) pthread_join(th[s], NULL);
  for (int s = 0; s < K; s++)
    if (fail[s]) return 3;
  for (int step = 0; step < 6; step++) {
    uint8_t did;
    if (wg_run_ms(ref, 50, &did, NULL) != WG_OK) return 3;
  }
  int bad = 0;
  const int fields[] = {WG_F_PONG, WG_F_MSG_RECEIVED, WG_F_MSG_SENT, WG_F_BYTES_SENT, WG_F_BYTES_RECEIVED};
  int64_t *a = malloc(8 * N), *b = malloc(8 * N), *want = malloc(8 * N);
  for (unsigned f = 0; f < sizeof fields / sizeof fields[0]; f++) {
    wg_read_i64(eng[0], fields[f], a, N);  /* a shard reports its own nodes and zeros for the others */
    wg_read_i64(eng[1], fields[f], b, N);
    wg_read_i64(ref, fields[f], want, N);
    for (int i = 0; i < N; i++)
      if (a[i] + b[i] != want[i]) {
        if (!bad) fprintf(stderr, "field %d node %d: shards %lld + %lld, unsharded %lld\n", fields[f], i, (long long)a[i], (long long)b[i], (long long)want[i]);
        bad++;
      }
  }
  uint64_t r0, r1, rr;
  int32_t t0, tr, lo, hi;
  int64_t calls, words;
  wg_rng_get_state(eng[0], &r0);
  wg_rng_get_state(eng[1], &r1);
  wg_rng_get_state(ref, &rr);
  wg_time(eng[0], &t0);
  wg_time(ref, &tr);
  if (r0 != rr || r1 != rr || t0 != tr) bad++;
  wg_shard_info(eng[1], &lo, &hi, &calls, &words);
  if (lo != N / 2 || hi != N || calls <= 0) bad++;
  printf("%s: %d nodes, 2 shards vs the unsharded engine, %lld collectives / %lld words per shard, pong[0] = %lld\n",
         bad ? "MISMATCH" : "OK", N, (long long)calls, (long long)words, (long long)want[0]);
  for (int s = 0; s < K; s++) wg_destroy(eng[s]);
  wg_destroy(ref);
  return bad ? 1 : 0;
}
