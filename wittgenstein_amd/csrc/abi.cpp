// extern "C" surface of libwittgpu.so (include/wittgpu.h). Thin: argument checks, exception ->
// status mapping, no engine logic.
#include <climits>
#include <cstring>
#include <vector>
#include "engine_host.h"

using namespace wg;

struct wg_engine {
  Engine* e;
};
static thread_local std::string g_createError;

#define WG_TRY(h)                       \
  if (!(h)) return WG_EINVAL;           \
  Engine& E = *(h)->e;                  \
  (void)hipSetDevice(E.cfg.device);     \
  try {
#define WG_END                          \
  }                                     \
  catch (const WgError& x) {            \
    E.lastError = x.what();             \
    return x.code;                      \
  }                                     \
  catch (const std::bad_alloc&) {       \
    E.lastError = "host out of memory"; \
    return WG_ENOMEM;                   \
  }                                     \
  catch (const std::exception& x) {     \
    E.lastError = x.what();             \
    return WG_ESTATE;                   \
  }                                     \
  return WG_OK;

extern "C" {

int32_t wg_create(const wg_config* cfg, wg_engine** out) {
  if (!out) return WG_EINVAL;
  *out = nullptr;
  wg_config c;
  memset(&c, 0, sizeof(c));
  if (cfg) c = *cfg;
  try {
    Engine* e = new Engine(c);
    *out = new wg_engine{e};
  } catch (const WgError& x) {
    g_createError = x.what();
    return x.code;
  } catch (const std::exception& x) {
    g_createError = x.what();
    return WG_EHIP;
  }
  return WG_OK;
}
void wg_destroy(wg_engine* h) {
  if (!h) return;
  delete h->e;
  delete h;
}
const char* wg_last_error(wg_engine* h) { return h ? h->e->lastError.c_str() : g_createError.c_str(); }

int32_t wg_add_nodes(wg_engine* h, int32_t n, const int32_t* x, const int32_t* y, const int32_t* extraLatency,
                     const uint8_t* down, const uint8_t* byzantine, const double* speedRatio) {
  WG_TRY(h) E.add_nodes(n, x, y, extraLatency, down, byzantine, speedRatio);
  WG_END
}
int32_t wg_node_count(wg_engine* h) { return h ? (int32_t)h->e->hx.size() : 0; }
int32_t wg_set_latency(wg_engine* h, int32_t kind, const int32_t* params, int32_t nparams) {
  WG_TRY(h) E.set_latency(kind, params, nparams);
  WG_END
}
int32_t wg_set_latency_city(wg_engine* h, int32_t mode, int32_t n_cities, const int32_t* city_of_node, const int32_t* tab,
                            const float* ping, const double* jitter100) {
  WG_TRY(h) E.set_latency_city(mode, n_cities, city_of_node, tab, ping, jitter100);
  WG_END
}
int32_t wg_set_latency_by_name(wg_engine* h, const char* name) {
  WG_TRY(h) E.set_latency_by_name(name);
  WG_END
}
int32_t wg_latency_probe(wg_engine* h, int32_t n, const int32_t* from, const int32_t* to, const int32_t* delta,
                         int32_t* out) {
  WG_TRY(h) E.latency_probe(n, from, to, delta, out);
  WG_END
}
int32_t wg_set_partitions(wg_engine* h, const int32_t* xcuts, int32_t k) {
  WG_TRY(h) E.set_partitions(xcuts, k);
  WG_END
}
int32_t wg_set_node_down(wg_engine* h, int32_t id, int32_t down) {
  WG_TRY(h) E.set_node_down(id, down != 0);
  WG_END
}
int32_t wg_set_discard_time(wg_engine* h, int32_t ms) {
  WG_TRY(h)
  E.discardTime = ms;      // (C/Network.java:103-107: a plain field, read at send time :481 — may change between runs)
  E.dev.discardTime = ms;  // the device table entry is refreshed before the next launch (Engine::self / Batch::prepare)
  WG_END
}
int32_t wg_rng_set_seed(wg_engine* h, int64_t seed) {
  WG_TRY(h) E.gh.rng = lcg_scramble(seed);
  E.globalsDirty = true;
  WG_END
}
int32_t wg_rng_get_state(wg_engine* h, uint64_t* s48) {
  WG_TRY(h)* s48 = E.gh.rng;
  WG_END
}
int32_t wg_rng_set_state(wg_engine* h, uint64_t s48) {
  WG_TRY(h) E.gh.rng = s48 & LCG_MASK;
  E.globalsDirty = true;
  WG_END
}
int32_t wg_send(wg_engine* h, uint32_t msg, uint32_t payload, int32_t sendTime, int32_t from, const int32_t* dests,
                int32_t n, int32_t delayBetween) {
  WG_TRY(h) E.send(msg, payload, sendTime, from, dests, n, delayBetween);
  WG_END
}
int32_t wg_send_arrive_at(wg_engine* h, uint32_t msg, uint32_t payload, int32_t arriveAt, int32_t from, int32_t to) {
  WG_TRY(h) E.send_arrive_at(msg, payload, arriveAt, from, to);
  WG_END
}
int32_t wg_register_task(wg_engine* h, uint32_t task, uint32_t arg, int32_t startAt, int32_t node) {
  WG_TRY(h) E.register_task(task, arg, startAt, node);
  WG_END
}
int32_t wg_register_periodic_task(wg_engine* h, uint32_t task, int32_t startAt, int32_t period, int32_t node) {
  WG_TRY(h) E.register_periodic_task(task, startAt, period, node);
  WG_END
}
int32_t wg_protocol_load(wg_engine* h, int32_t proto_id, const void* params, const void* init_state) {
  WG_TRY(h) E.load_protocol(proto_id, params, init_state);
  WG_END
}
int32_t wg_run_ms(wg_engine* h, int32_t ms, uint8_t* didSomething, wg_run_stats* stats) {
  WG_TRY(h) E.run_ms(ms, didSomething, stats);
  WG_END
}
int32_t wg_shard_configure(wg_engine* h, int32_t shard, int32_t nshards, wg_allreduce_fn fn, void* ctx) {
  WG_TRY(h) E.configure_shard(shard, nshards, fn, ctx);
  WG_END
}
int32_t wg_shard_set_alltoallv(wg_engine* h, wg_alltoallv_fn fn, void* ctx) {
  WG_TRY(h) E.set_alltoallv(fn, ctx);
  WG_END
}
int32_t wg_shard_configure_rccl(wg_engine* h, int32_t shard, int32_t nshards, const uint8_t* id128) {
  WG_TRY(h) E.configure_shard_rccl(shard, nshards, id128);
  WG_END
}
int32_t wg_rccl_unique_id(uint8_t* id128) {
  if (!id128) return WG_EINVAL;
  try {
    wg::rccl_unique_id(id128);
  } catch (const WgError& x) {
    g_createError = x.what();
    return x.code;
  }
  return WG_OK;
}
int32_t wg_shard_info(wg_engine* h, int32_t* lo, int32_t* hi, int64_t* collectives, int64_t* words) {
  WG_TRY(h)
  if (E.shardCount == 0) throw WgError(WG_ESTATE, "not a sharded engine");
  const int64_t n = (int64_t)E.hx.size();
  if (lo) *lo = (int32_t)(n * E.shardIndex / E.shardCount);
  if (hi) *hi = (int32_t)(n * (E.shardIndex + 1) / E.shardCount);
  if (collectives) *collectives = E.shardCollectives;
  if (words) *words = E.shardWords;
  WG_END
}
int32_t wg_shard_traffic(wg_engine* h, int64_t* calls8, int64_t* words8) {
  WG_TRY(h)
  if (E.shardCount == 0) throw WgError(WG_ESTATE, "not a sharded engine");
  for (int k = 0; k < Engine::XK_KINDS; k++) {
    if (calls8) calls8[k] = E.shardCallsBy[k];
    if (words8) words8[k] = E.shardWordsBy[k];
  }
  WG_END
}
int32_t wg_time(wg_engine* h, int32_t* time) {
  WG_TRY(h)* time = E.time;
  WG_END
}
int32_t wg_queue_size(wg_engine* h, int64_t* size) {
  WG_TRY(h)* size = E.queue_size();
  WG_END
}
int32_t wg_queue_size_at(wg_engine* h, int32_t t, int64_t* size) {
  WG_TRY(h)* size = E.queue_size_at(t);
  WG_END
}
int32_t wg_read_i64(wg_engine* h, int32_t field, int64_t* dst, int32_t n) {
  WG_TRY(h) E.read_i64(field, dst, n);
  WG_END
}
int32_t wg_read_i32(wg_engine* h, int32_t field, int32_t* dst, int32_t n) {  // the int fields (pong, sigsChecked, ...)
  WG_TRY(h)
  if (!dst || n < 0) throw WgError(WG_EINVAL, "dst / n");
  std::vector<int64_t> tmp((size_t)n);
  E.read_i64(field, tmp.data(), n);
  for (int32_t i = 0; i < n; i++) {
    if (tmp[(size_t)i] < INT32_MIN || tmp[(size_t)i] > INT32_MAX) throw WgError(WG_EINVAL, "wg_read_i32: the field does not fit 32 bits (use wg_read_i64)");
    dst[i] = (int32_t)tmp[(size_t)i];
  }
  WG_END
}
int32_t wg_abi_version(void) { return WG_ABI_VERSION; }
int32_t wg_selftest(int32_t op, int32_t aux, const uint64_t* in, int32_t n, int32_t threads, uint64_t* out, int32_t n_out) {
  try {
    selftest(op, aux, in, n, threads, out, n_out);
  } catch (const WgError& x) {
    g_createError = x.what();
    return x.code;
  } catch (const std::exception& x) {
    g_createError = x.what();
    return WG_EHIP;
  }
  return WG_OK;
}
int32_t wg_abi_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int32_t)sizeof(wg_config);
    case 1: return (int32_t)sizeof(wg_handel_params);
    case 2: return (int32_t)sizeof(wg_gsf_params);
    case 3: return (int32_t)sizeof(wg_casper_params);
    case 4: return (int32_t)sizeof(wg_sanfermin_params);
    case 5: return (int32_t)sizeof(wg_p2pflood_params);
    case 6: return (int32_t)sizeof(wg_delivery);
    case 7: return (int32_t)sizeof(wg_step_op);
    case 8: return (int32_t)sizeof(wg_run_stats);
    default: return -1;
  }
}
int32_t wg_read_level_i32(wg_engine* h, int32_t field, int32_t* dst, int32_t n_nodes, int32_t n_levels) {
  WG_TRY(h)
  if (!E.proto || !E.proto->read_level_i32(E, field, dst, n_nodes, n_levels))
    throw WgError(WG_EINVAL, "unknown level field for the resident protocol");
  WG_END
}
int32_t wg_read_bits(wg_engine* h, int32_t field, uint64_t* dst, int32_t n_nodes, int32_t words_per_node) {
  WG_TRY(h)
  if (!E.proto || !E.proto->read_bits(E, field, dst, n_nodes, words_per_node))
    throw WgError(WG_EINVAL, "unknown bitset field for the resident protocol");
  WG_END
}
int32_t wg_levels(wg_engine* h, int32_t* levels) {
  WG_TRY(h)* levels = E.proto ? E.proto->levels() : 0;
  WG_END
}
int32_t wg_device_bytes(wg_engine* h, int64_t* bytes) {
  WG_TRY(h)
  if (!bytes) throw WgError(WG_EINVAL, "bytes");
  int64_t b = 0;
  for (const auto& a : E.allocInfo) b += (int64_t)a.bytes;
  *bytes = b + E.snapshot_bytes();
  WG_END
}
int32_t wg_delivered_by_level(wg_engine* h, int64_t* dst32) {
  WG_TRY(h)
  for (int i = 0; i < 32; i++) dst32[i] = (int64_t)E.gh.deliveredByLevel[i];
  WG_END
}

int32_t wg_protocol_cont_if(wg_engine* h, int32_t* cont) {
  WG_TRY(h)
  if (!cont) throw WgError(WG_EINVAL, "cont");
  E.ensure_device();
  E.flush_staged(E.time, false);
  if (!E.proto || !E.proto->cont_if(E, cont))
    throw WgError(WG_EUNSUPPORTED, "the resident protocol defines no continuation predicate");
  WG_END
}
int32_t wg_snapshot(wg_engine* h) {
  WG_TRY(h) E.snapshot();
  WG_END
}
int32_t wg_restore(wg_engine* h) {
  WG_TRY(h) E.restore();
  WG_END
}
int32_t wg_snapshot_bytes(wg_engine* h, int64_t* bytes) {
  WG_TRY(h)
  if (!bytes) throw WgError(WG_EINVAL, "bytes");
  *bytes = E.snapshot_bytes();
  WG_END
}
int32_t wg_next_delivery(wg_engine* h, int32_t until, int32_t cond_time, wg_delivery* out, int32_t* got) {
  WG_TRY(h)
  if (!out || !got) throw WgError(WG_EINVAL, "out/got");
  *got = E.next_delivery(until, cond_time, out) ? 1 : 0;
  WG_END
}
int32_t wg_step_begin(wg_engine* h, int32_t until, int32_t cond_time, wg_delivery* out, int32_t cap, int32_t* n) {
  WG_TRY(h)
  if (!out || !n) throw WgError(WG_EINVAL, "out/n");
  *n = E.step_begin(until, cond_time, out, cap);
  WG_END
}
int32_t wg_step_end(wg_engine* h, const wg_step_op* ops, int32_t nops, const int32_t* dests) {
  WG_TRY(h) E.step_end(ops, nops, dests);
  WG_END
}
int32_t wg_host_released(wg_engine* h, uint32_t* msgs, int32_t cap, int32_t* n) {
  WG_TRY(h)
  if (!n) throw WgError(WG_EINVAL, "n");
  *n = E.host_released(msgs, cap);
  WG_END
}
int32_t wg_set_time(wg_engine* h, int32_t time) {
  WG_TRY(h) E.host_set_time(time);
  WG_END
}
// ---- batches (RunMultipleTimes on the device)
struct wg_batch {
  Batch* b;
};
static thread_local std::string g_batchError;
#define WGB_TRY try {
#define WGB_END(h)                                       \
  }                                                      \
  catch (const WgError& x) {                             \
    ((h) ? (h)->b->lastError : g_batchError) = x.what(); \
    return x.code;                                       \
  }                                                      \
  catch (const std::exception& x) {                      \
    ((h) ? (h)->b->lastError : g_batchError) = x.what(); \
    return WG_ESTATE;                                    \
  }                                                      \
  return WG_OK;

int32_t wg_batch_create(wg_engine** engines, int32_t n, wg_batch** out) {
  if (!out) return WG_EINVAL;
  *out = nullptr;
  wg_batch* none = nullptr;
  WGB_TRY
  if (!engines || n <= 0) throw WgError(WG_EINVAL, "empty batch");
  std::vector<Engine*> es;
  for (int i = 0; i < n; i++) {
    if (!engines[i]) throw WgError(WG_EINVAL, "NULL engine in batch");
    es.push_back(engines[i]->e);
  }
  *out = new wg_batch{new Batch(es.data(), n)};
  WGB_END(none)
}
void wg_batch_destroy(wg_batch* b) {
  if (!b) return;
  delete b->b;
  delete b;
}
const char* wg_batch_last_error(wg_batch* b) { return b ? b->b->lastError.c_str() : g_batchError.c_str(); }
int32_t wg_batch_size(wg_batch* b, int32_t* n) {
  if (!b || !n) return WG_EINVAL;
  *n = (int32_t)b->b->members.size();
  return WG_OK;
}
int32_t wg_batch_run_ms(wg_batch* b, int32_t ms, const uint8_t* active, uint8_t* didSomething, wg_run_stats* stats) {
  if (!b) return WG_EINVAL;
  WGB_TRY b->b->run_ms(ms, active, didSomething, stats);
  WGB_END(b)
}
int32_t wg_batch_cont_if(wg_batch* b, int32_t* cont) {
  if (!b || !cont) return WG_EINVAL;
  WGB_TRY b->b->cont_if(cont);
  WGB_END(b)
}

int32_t wg_batch_run_multiple_times(wg_batch* b, int32_t chunk, int32_t maxTime, int64_t* delivered, int64_t* simulatedMs) {
  if (!b) return WG_EINVAL;
  WGB_TRY b->b->run_multiple_times(chunk, maxTime, delivered, simulatedMs);
  WGB_END(b)
}

int32_t wg_profile_enable(wg_engine* h, int32_t on) {
  WG_TRY(h)
  if (on < 0 || on > 2) throw WgError(WG_EINVAL, "mode");
  E.profiling = on;
  for (int c = 0; c < Engine::PC_COUNT; c++) {
    E.profNs[c] = 0;
    E.profLaunches[c] = 0;
  }
  WG_END
}
int32_t wg_profile_set_reference(wg_engine* h, wg_engine* ref) {
  WG_TRY(h)
  Engine& R = ref ? *ref->e : E;
  if (!R.profOwnRef) {
    (void)hipSetDevice(R.cfg.device);
    WG_HIP(hipEventCreate(&R.profOwnRef));
    WG_HIP(hipEventRecord(R.profOwnRef, R.stream));
    WG_HIP(hipEventSynchronize(R.profOwnRef));
  }
  E.profRef = R.profOwnRef;
  for (int c = 0; c < Engine::PC_COUNT; c++) E.profTimes[c].clear();
  WG_END
}
int32_t wg_profile_read_spans(wg_engine* h, int32_t cls, double* start_ns, double* end_ns, int32_t cap, int32_t* n) {
  WG_TRY(h)
  if (cls < 0 || cls >= Engine::PC_COUNT || !n) throw WgError(WG_EINVAL, "class / n");
  auto& v = E.profTimes[cls];
  const int32_t k = std::min<int32_t>((int32_t)v.size(), cap < 0 ? 0 : cap);
  for (int32_t i = 0; i < k; i++) {
    if (start_ns) start_ns[i] = v[i].first;
    if (end_ns) end_ns[i] = v[i].second;
  }
  *n = (int32_t)v.size();
  if (k == (int32_t)v.size()) v.clear();  // (read in full: start afresh)
  WG_END
}
int32_t wg_profile_read(wg_engine* h, wg_profile_entry* dst, int32_t cap, int32_t* n) {
  WG_TRY(h)
  static const char* names[Engine::PC_COUNT] = {"expand(k_scan1+k_scan2<ExpandF>)", "group(unused)",
                                                "deliver", "order(k_scan<RecsF>)", "k_resolve",
                                                "append(k_tile_hist+k_col_reserve+k_scatter)", "k_end_phase",
                                                "cond_select", "cond_rest"};
  int k = 0;
  for (int c = 0; c < Engine::PC_COUNT && k < cap; c++, k++) {
    dst[k].name = names[c];
    dst[k].spans = E.profLaunches[c];
    dst[k].total_ns = E.profNs[c];
  }
#ifdef WG_KPROF
  static const char* kn[32] = {"kprof00 visits", "kprof01 a1 cyc lane-part wavefronts", "kprof02 a1 lane-part wavefronts", "kprof03 cyc node_end",
                               "kprof04 n on_message", "kprof05 cyc on_message", "kprof06 n dissemination",
                               "kprof07 cyc dissemination", "kprof08 cyc dissem fin-bits", "kprof09 cyc dissem snapshots",
                               "kprof10 cyc dissem sends", "kprof11 n update", "kprof12 cyc update",
                               "kprof13 n dissem slow levels", "kprof14 a1 lane items", "kprof15 a1 lane items' entries",
                               "kprof16 a1 wave items", "kprof17 a1 lane entries evaluated (not cached)", "kprof18 a1 lane items with an evaluation", "kprof19 a1 cyc list entries",
                               "kprof20 a1 wave items' entries", "kprof21 a1 wave entries evaluated", "kprof22 a1 cyc wave evaluations",
                               "kprof23 a1 cyc curation+store", "kprof24 msgs wave rounds", "kprof25 msgs cyc classify",
                               "kprof26 msgs cyc lane messages", "kprof27 a1 lane items of wide levels (all cached)", "kprof28 a1 lane items",
                               "kprof29 a1 lane items' entries", "kprof30", "kprof31"};
  for (int c = 0; c < 32 && k < cap; c++, k++) {
    dst[k].name = kn[c];
    dst[k].spans = 1;
    dst[k].total_ns = (double)E.kprofSum[c];
  }
#endif
  if (n) *n = k;
  WG_END
}

}  // extern "C"
