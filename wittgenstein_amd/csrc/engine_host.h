// Host-side runtime of the engine: owns device memory and the HIP stream, enqueues the per-ms
// kernel sequence, mirrors the pieces of core.Network state that init() code paths touch from the
// host (time, rd, node flags, staged envelopes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <climits>
#include <vector>
#include "../../include/wittgpu.h"
#include "engine.h"

namespace wg {

struct WgError : std::runtime_error {
  int32_t code;
  WgError(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define WG_HIP(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      throw ::wg::WgError(WG_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));                \
  } while (0)

class Engine;

// What a launch addresses: a device table of R engines (blockIdx.y) + the matching table of protocol
// State structs, on one stream. A stand-alone engine is a group of one; a batch (wg_batch_*) is the
// device-side form of the reference's RunMultipleTimes loop over independent copies.
// a PeriodicTask the HOST registered (wg_register_periodic_task): it fires at startAt, startAt + period, ... — the re-arm
// keeps the phase (C/messages/PeriodicTask.java:39-47), so a protocol host can tell from the simulated ms being enqueued
// whether a kernel that only serves that task can find work at all
struct PeriodicReg {
  uint32_t task;
  int32_t period, phase;
};
struct Group {
  const EngineDev* tab = nullptr;
  const void* stab = nullptr;
  int R = 1;
  int nodes = 0;  // node count of the members (they share it): what sizes the per-engine grids (engine.h grid_per_engine)
  hipStream_t stream = nullptr;
  int binBits = 0;
  size_t histLds = 0;  // dynamic LDS of the multisplit kernels: horizon words
  // the simulated ms the launches being enqueued will see as `now` (INT32_MIN: unknown — members on different clocks, a
  // chunk captured as a graph), and the periodic tasks registered on the members (NULL: unknown): hints, never needed
  int32_t now = INT32_MIN;
  // one append per simulated ms (the drain's outbox kept over the conditional-task phase, Globals::nOutKeep) or two
  // (WG_MERGE_APPEND=0, the A/B and test switch): read once when the group is made, not per enqueued ms
  bool mergeAppend = true;
  const std::vector<PeriodicReg>* periodic = nullptr;
  bool periodic_may_fire(uint32_t task) const {
    if (now == INT32_MIN || !periodic) return true;
    // (a task of which NO registration is known is not "never fires" but "unknown": an envelope of it that reached the queue
    // some other way — a restored image, a registration made on the device — must still find its kernel launched)
    bool known = false;
    for (const PeriodicReg& r : *periodic)
      if (r.task == task) {
        known = true;
        if (now >= r.phase && (now - r.phase) % r.period == 0) return true;
      }
    return !known;
  }
  // ... any of them: the ms in which every node's periodic task sends (Handel's dissemination, GSFSignature's doCycle) has
  // a hundred times the outbox records of the others
  bool any_periodic_may_fire() const {
    if (now == INT32_MIN || !periodic) return true;
    for (const PeriodicReg& r : *periodic)
      if (now >= r.phase && (now - r.phase) % r.period == 0) return true;
    return false;
  }
};

// A resident protocol: device state + the kernels that run its action()/conditional tasks.
struct ProtoHost {
  virtual ~ProtoHost() {}
  virtual bool has_cond() const { return false; }
  // conditional-task phase at the time++ edge -> t (C/Network.java:543-566); emits into the ordered
  // outbox (fin/arr, g->nOut) and g->nDraws. Only called when has_cond().
  // adds the protocol's share of a Node counter (WG_F_MSG_RECEIVED ...) to dst, if it keeps one outside NodeArrays
  virtual bool node_counter(Engine&, int32_t field, int64_t* dst, int32_t n) { return false; }
  virtual void launch_cond(Engine& profOwner, const Group&) {}
  virtual void launch_deliver(const Group&) = 0;
  // a sharded engine's delivery pass when it needs collectives of its own (true = done; false = launch_deliver)
  virtual bool shard_deliver(Engine&, const Group&) { return false; }
  virtual size_t state_size() const = 0;        // sizeof the device State struct ...
  virtual const void* state_host() const = 0;   // ... and its host copy (what a Group's stab holds)
  // continuation predicate for every member of the group: out[R] on the device
  virtual bool launch_cont_if(const Group&, uint32_t* /*dOut*/) { return false; }
  virtual int payload_bytes_of_level(int /*level*/) const { return 0; }
  virtual bool read_i64(Engine&, int32_t /*field*/, int64_t* /*dst*/, int32_t /*n*/) { return false; }
  virtual bool read_level_i32(Engine&, int32_t, int32_t*, int32_t, int32_t) { return false; }
  virtual bool read_bits(Engine&, int32_t, uint64_t*, int32_t, int32_t) { return false; }
  virtual int levels() const { return 0; }
  virtual int variant() const { return 0; }  // (a mode that selects other kernels: batch members must agree on it)
  virtual int host_msg_size(uint32_t /*msg*/) const { return 1; }  // Message.size() of a host-side send
  // every message of the protocol has size() 1: Node.bytesReceived == Node.msgReceived, its lane-per-event delivery kernel
  // counts once and the read-back serves both fields from msgReceived
  virtual bool unit_message_size() const { return false; }
  virtual bool delivered_by_level(Engine&, int64_t* /*dst32*/) { return false; }
  // the protocol's RunMultipleTimes continuation predicate, evaluated on the device
  virtual bool cont_if(Engine&, int32_t* /*out*/) { return false; }
  // node-range sharding (wg_shard_configure): the protocol's kernels touch only the nodes the engine links into
  // its inbox lists, its payloads are self-contained and it has no conditional-task phase
  virtual bool supports_shards() const { return false; }
  // after the delivery kernels of a ms: payloads written by this shard's action()s that other shards will read.
  // enqueue numbers the rows on the device and returns where their count will be (NULL: the protocol has none);
  // exchange ships `nSnap` rows (called only when there are any)
  // payload snapshots of this ms that other shards will read. Two forms: rows numbered by a scan over the events that
  // reported one (shard_snap_is_scan: Handel, GSFSignature — enqueued only in a ms whose order scan counted such events,
  // Globals::nSnapEv), or a count the delivery pass left behind (Casper's table exchange)
  virtual bool shard_snap_is_scan() const { return false; }
  // ... or goes, row by row, to the shards whose nodes will read it (an all-to-all: Engine::shard_alltoallv) — then
  // shard_snap_exchange is called in every ms whose order scan counted a snapshot event and nothing is numbered
  virtual bool shard_snap_directed() const { return false; }
  virtual uint32_t* shard_snap_enqueue(const Group&) { return nullptr; }
  virtual void shard_snap_exchange(Engine&, const Group&, uint32_t /*nSnap*/) {}
  // the conditional-task phase on a sharded engine: leaves the task records of this shard's nodes in the exchange
  // image (EngineDev::xbuf) and returns the (replicated) number of records; only called when has_cond()
  virtual uint32_t shard_cond(Engine&, const Group&) { return 0; }
  // wg_restore: whatever of the protocol's state is cheaper to recompute than to copy (enqueued on the engine's stream
  // after the STATE allocations are back); throws if the state cannot be brought back
  virtual void on_restore(Engine&) {}
};

class Engine {
 public:
  explicit Engine(const wg_config& cfg);
  ~Engine();

  // topology
  void add_nodes(int32_t n, const int32_t* x, const int32_t* y, const int32_t* extra, const uint8_t* down,
                 const uint8_t* byz, const double* speed);
  void set_latency(int32_t kind, const int32_t* params, int32_t nparams);
  void set_latency_by_name(const char* name);
  void set_latency_city(int32_t mode, int32_t nCities, const int32_t* cityOfNode, const int32_t* tab, const float* ping,
                        const double* jit);
  std::vector<uint16_t> cityOf;   // LAT_CITY host tables (mirrored on the device by upload_latency)
  std::vector<int32_t> cityTab;
  std::vector<float> cityPing;
  std::vector<double> cityJit;
  int32_t nCities = 0;
  void *dCity = nullptr, *dCityTab = nullptr, *dCityPing = nullptr, *dCityJit = nullptr;
  void set_partitions(const int32_t* cuts, int32_t k);
  void set_node_down(int32_t id, bool down);
  void latency_probe(int32_t n, const int32_t* from, const int32_t* to, const int32_t* delta, int32_t* out);

  // host-side Network API used by init() code
  void send(uint32_t msg, uint32_t payload, int32_t sendTime, int32_t from, const int32_t* dests, int32_t n,
            int32_t delayBetween);
  void send_arrive_at(uint32_t msg, uint32_t payload, int32_t arriveAt, int32_t from, int32_t to);
  // list sends with >= sendExpandMin destinations are resolved on the device (k_send_expand_*); WG_SEND_EXPAND_MIN
  // (below it the host's latency loop + stable sort is as fast as three launches and a read-back)
  void send_expanded(uint32_t msg, uint32_t payload, int32_t sendTime, int32_t from, const int32_t* dests, int32_t n,
                     int32_t seed);
  int32_t sendExpandMin = getenv("WG_SEND_EXPAND_MIN") ? atoi(getenv("WG_SEND_EXPAND_MIN")) : 4096;
  int32_t *expIn = nullptr, *expLat = nullptr, *expResult = nullptr;
  uint32_t* expHist = nullptr;
  size_t expCap = 0;
  void register_task(uint32_t task, uint32_t arg, int32_t startAt, int32_t node);
  void register_periodic_task(uint32_t task, int32_t startAt, int32_t period, int32_t node);

  void load_protocol(int32_t id, const void* params, const void* initState);
  void run_ms(int32_t ms, uint8_t* didSomething, wg_run_stats* stats);
  // node-range sharding of one simulation over several engines (one process per GPU), include/wittgpu.h
  void configure_shard(int32_t shard, int32_t nshards, wg_allreduce_fn fn, void* ctx);
  void run_ms_sharded(int32_t ms, uint8_t* didSomething, wg_run_stats* stats);
  // kind: which of the per-ms exchanges this is (wg_shard_traffic's rows, include/wittgpu.h)
  enum ShardKind { XK_EVENTS = 0, XK_OUTBOX = 1, XK_ENVELOPES = 2, XK_SNAPSHOTS = 3, XK_CANDIDATES = 4, XK_COUNTS = 5, XK_OTHER = 6, XK_KINDS = 8 };
  void shard_allreduce(void* buf, int64_t count, int kind = XK_OTHER);
  void exchange_outbox(uint32_t nOut);
  int32_t shardIndex = 0, shardCount = 0;  // shardCount == 0: not sharded
  wg_allreduce_fn xfn = nullptr;          // caller-supplied collective (tests: gloo / in-process loopback) ...
  void* xctx = nullptr;
  wg_alltoallv_fn xa2a = nullptr;         // ... and its all-to-all (wg_shard_set_alltoallv; optional)
  void* xa2aCtx = nullptr;
  bool has_alltoall() const;              // an all-to-all transport exists: the callback, or the engine's RCCL communicator
  // shard d gets sc[d] int32 words from word so[d] of sendbuf, rc[r] words from shard r land at word ro[r] of recvbuf
  void shard_alltoallv(const void* sendbuf, const int64_t* sc, const int64_t* so, void* recvbuf, const int64_t* rc, const int64_t* ro, int kind = XK_SNAPSHOTS);
  void set_alltoallv(wg_alltoallv_fn fn, void* ctx);
  void* rcclComm = nullptr;               // ... or the engine's own RCCL communicator (wg_shard_configure_rccl):
                                          // ncclAllReduce enqueued on the engine's stream, no host round trip
  void configure_shard_rccl(int32_t shard, int32_t nshards, const uint8_t* uniqueId128);
  // per-phase counts the host needs to size the next collective, published by the device into pinned host memory
  // (k_publish) and awaited by polling — not a stream synchronisation plus a copy per count
  struct Mailbox {
    uint32_t seq;   // written last by k_publish; the host reads it with acquire semantics (await_counts)
    uint32_t v[7];
  };
  Mailbox* mailbox = nullptr;             // pinned host memory (hipHostMalloc), device-visible: a ring of MAILBOXES
  static constexpr uint32_t MAILBOXES = 8;
  uint32_t mailSeq = 0;
  // publish_counts enqueues the write of two device words into the next mailbox (in stream order, behind their producers);
  // wait_counts polls for it. The caller enqueues, between the two, every kernel that needs the counts only as device
  // values — the host learns a count while the device already works with it, not in an idle gap before the next launch
  uint32_t publish_counts(const uint32_t* a, const uint32_t* b);
  void wait_counts(uint32_t seq, uint32_t* va, uint32_t* vb);
  void await_counts(const uint32_t* a, const uint32_t* b, uint32_t* va, uint32_t* vb) { wait_counts(publish_counts(a, b), va, vb); }
  long long shardCollectives = 0, shardWords = 0;  // all-reduce calls / int32 words summed so far
  long long shardCallsBy[XK_KINDS] = {}, shardWordsBy[XK_KINDS] = {};  // ... by exchange
  int64_t queue_size();
  int64_t queue_size_at(int32_t t);
  struct StagedChainKeep {
    bool live = false;
    Chain c;
    std::vector<int32_t> words;
  };
  // host-callback mode (WG_PROTO_HOST): Network.nextMessage + the post-action part of receiveUntil
  bool next_delivery(int32_t until, int32_t condTime, wg_delivery* out);
  void host_set_time(int32_t t);
  // ... and its batched form (wg_step_begin / wg_step_end): the deliveries of a ms in one call, the caller's pushes in one call
  struct StepPlan {
    int32_t outIdx;             // index in the caller's batch, -1: consumed undelivered (:606), only its re-push is owed
    int32_t contSlot, contPos;  // multi-destination envelope to re-push after it (-1: none)
  };
  std::vector<StepPlan> hcPlan;
  bool hcStepOpen = false;
  // host-callback mode: message / task handles whose envelope has ENDED since the caller last asked (wg_host_released) — its
  // one destination was handed out (or consumed, C/Network.java:606), the last hop of a multi-destination envelope was, or no
  // destination was reachable when it was sent: the caller may forget the object behind the handle (the reference drops
  // its Envelope there; a binding that kept every Message of a run alive ran out of memory at ~1e7 messages)
  std::vector<uint32_t> hcReleased;
  int32_t host_released(uint32_t* msgs, int32_t cap);
  int32_t step_begin(int32_t until, int32_t condTime, wg_delivery* out, int32_t cap);
  void step_end(const wg_step_op* ops, int32_t nops, const int32_t* dests);
  void hc_load_ms(int32_t until);
  struct HostEv {
    Rec rec;
    EvAux aux;
  };
  std::vector<HostEv> hcEvents;    // the ms being handed out, in event order
  size_t hcCursor = 0;
  bool hcLoaded = false;
  int32_t hcContSlot = -1, hcContPos = 0;  // chain re-push owed after the delivery in progress (:629-632)
  std::vector<StagedChainKeep> hostChains; // host copies of the multi-destination envelopes, by slot
  void hc_push(int32_t arrival, const Rec& rec);
  void send_seeded(uint32_t msg, uint32_t payload, int32_t sendTime, int32_t from, const int32_t* dests, int32_t n,
                   int32_t delayBetween, int32_t seed);
  void hc_stage_continuation();
  void hc_finish_ms();
  void read_i64(int32_t field, int64_t* dst, int32_t n);

  // used by protocol hosts
  void ensure_device();          // allocate device state once the node count is known
  void sync_globals_to_host();
  void sync_globals_to_device();
  static void append_phase(const Group& g, bool needHist);  // multisplit of the ordered outbox into the buckets
  static void end_phase(const Group& g, bool drained, bool keep = false);
  static void append_end_phase(const Group& g, bool drained);  // append_phase + end_phase in two launches (k_col_reserve_end)
  template <class F>
  static void scan(const Group& g, const typename F::Aux* atab);
  void expand(const Group& g);  // scan<ExpandF> + k_expand_runs
  int32_t max_latency() const;
  uint32_t sendall_bins() const;
  Group self();                  // this engine as a group of one (refreshes its device table entry)
  // Network.runMs for every active member of a group in lock-step (one launch sequence for all)
  static void run_group(Engine** es, int R, const uint8_t* active, const Group& g, int32_t ms, uint8_t* did,
                        wg_run_stats* stats);
  // the kernel sequence of runMs(ms) for a group whose members' globals are set up (begin_run / k_chunk_begin)
  // tStart: the simulated ms of the sequence's first drain (INT32_MIN: unknown)
  static void enqueue_ms_sequence(Engine& lead, const Group& g, int32_t ms, int32_t tStart = INT32_MIN);
  std::vector<PeriodicReg> periodicRegs;  // distinct (task, period, startAt mod period) of the host's registrations
  bool periodicUnknown = false;           // more distinct ones than worth tracking: the hint is off
  void begin_run(int32_t ms, int32_t* endAt);
  EngineDev* dTab = nullptr;     // device copy of `dev` (table of one)
  void* dStab = nullptr;         // device copy of the protocol State struct
  EngineDev tabShadow{};         // what dTab / dStab currently hold (skip the upload when unchanged)
  std::vector<char> stabShadow;
  void flush_staged(int32_t t, bool inRun);
  int32_t host_latency(int32_t from, int32_t to, int32_t seed) const;
  int32_t part_of(int32_t x) const;
  void check_device_errors();

  wg_config cfg;
  std::string lastError;
  hipStream_t stream = nullptr;
  EngineDev dev{};               // device pointers (passed by value to kernels)
  Globals gh{};                  // host shadow of the device globals (valid between runs)
  unsigned long long kprofSum[32] = {0};  // -DWG_KPROF builds: the in-kernel marks, summed over wavefronts and runs
  bool globalsDirty = true;
  bool allocated = false;
  int32_t time = 0;              // Network.time
  int32_t discardTime = INT32_MAX;
  int32_t binBits = 0;
  int32_t farCapacity = 0;       // > 0: envelopes registered beyond the bucket ring are parked for the host (FarRec)
  int32_t horizonExtra = 0;      // the longest sendTime - time a resident protocol's sends use (added to the latency bound)
  void collect_far();            // FarRec -> staged (host-held) envelopes
  int32_t* skipBuf = nullptr;    // k_next_busy's result, one word per member of the group this engine leads
  int skipCap = 0;
  bool wantInbox = false;        // the resident protocol reads a node's events from its inbox line (EngineDev::inbox)
  int32_t sendAllCapacity = 0;   // Network.sendAll calls per simulated ms a resident protocol's action()s may make
  int32_t horizonFloor = 0;      // a resident protocol's longest task delay (default horizon_ms only)
  uint32_t maxTiles = 0;
  // host copies of node fields (send-time decisions of host-side sends)
  std::vector<int32_t> hx, hy, hextra;
  std::vector<uint8_t> hdown, hbyz;
  bool downDirty = false;        // hdown changed since its last upload (set_node_down)
  void upload_down();
  bool geoStale = false;         // NodeArrays::geo is behind x / y / extraLatency / down / the cuts
  void upload_geo();
  std::vector<double> hspeed;
  std::vector<int32_t> cuts;
  // latency model (host tables mirrored on device)
  int32_t latKind = LAT_IC3;     // C/Network.java:43 default
  int32_t latParam = 0;
  std::vector<uint8_t> lutDist;
  std::vector<int32_t> tabDelta, tabDist;
  void* dLut = nullptr;
  void* dTabDelta = nullptr;
  void* dTabDist = nullptr;
  // envelopes pushed from the host and not yet on the device, in push order
  struct Staged {
    int32_t arrival;
    Rec rec;
  };
  std::vector<Staged> staged;
  int32_t stagedMin = INT32_MAX;  // earliest arrival among host-held envelopes
  struct StagedChain {
    uint32_t slot;
    Chain c;
    std::vector<int32_t> words;  // dest ids (+ explicit arrivals)
  };
  std::vector<StagedChain> stagedChains;
  struct PendingSent {  // Node.msgSent/bytesSent of host-side sends, applied at flush
    int32_t node;
    long long msgs;
    uint32_t msg;
  };
  std::vector<PendingSent> pendingSent;
  char* sentBuf = nullptr;  // flush_staged's staging buffer for Node.msgSent / bytesSent of host-side sends (grow-only)
  size_t sentBufBytes = 0;
  ProtoHost* proto = nullptr;
  std::vector<void*> allocs;     // everything to hipFree
  // What wg_snapshot / wg_restore do with an allocation (parallel to `allocs`): STATE is copied, SCRATCH holds
  // nothing that outlives a simulated ms (or nothing before the first event), CONST is never written by a kernel.
  enum AllocClass : int { AC_STATE = 0, AC_SCRATCH = 1, AC_CONST = 2 };
  struct AllocInfo {
    size_t bytes;
    int cls;
  };
  std::vector<AllocInfo> allocInfo;
  // The image of the engine right after init() (wg_snapshot): device copies of every STATE allocation in one arena +
  // the host-side pieces of Network state; wg_restore puts the engine back there — the cheap form of
  // RunMultipleTimes' `p.copy(); rd.setSeed(i); init()` for a seed that was initialised once (C/RunMultipleTimes.java:44-48).
  struct Snapshot {
    char* arena = nullptr;
    size_t bytes = 0;
    std::vector<size_t> offs;    // per allocation: offset in the arena, (size_t)-1 = not copied
    // the bucket pool is kept by its pages IN USE (after init(): the periodic tasks' records — a few pages of 134 MB); the
    // envelope table and its destination ring not at all while no envelope was ever made (the table is re-zeroed instead)
    std::vector<std::pair<uint32_t, size_t>> poolPages;  // page, offset in the arena
    bool chainsZero = false;
    Globals gh;
    int32_t time = 0, discardTime = 0, stagedMin = 0;
    std::vector<uint8_t> hdown;
    std::vector<int32_t> cuts;
    std::vector<Staged> staged;  // envelopes still held on the host (beyond the bucket ring)
    size_t nAllocs = 0;
  };
  Snapshot* snap = nullptr;
  void snapshot();
  void restore();
  int64_t snapshot_bytes() const { return snap ? (int64_t)snap->bytes : 0; }

  // optional per-kernel-class timing with HIP events recorded on the engine's stream (bench.py's
  // roofline leg; off by default because every bracket costs two event records)
  enum ProfClass { PC_EXPAND = 0, PC_GROUP, PC_DELIVER, PC_ORDER, PC_RESOLVE, PC_APPEND, PC_END, PC_COND_SELECT,
                   PC_COND_REST, PC_COUNT };
  int profiling = 0;                       // 0 off, 1 every phase, 2 the delivery kernel only
  struct ProfSpan {
    int cls;
    hipEvent_t a, b;
  };
  std::vector<ProfSpan> profSpans;         // recorded in the current run
  std::vector<hipEvent_t> profFree;        // recycled events
  double profNs[PC_COUNT] = {0};
  long long profLaunches[PC_COUNT] = {0};
  hipEvent_t prof_event();
  void prof_collect();                     // after the stream is idle
  // spans with their place in time (wg_profile_set_reference / wg_profile_read_spans): start / end of every bracketed span
  // in ns since a reference event — this engine's own or ANOTHER engine's, so that spans recorded on different streams
  // (concurrent batches) can be laid on one time axis and their union measured
  hipEvent_t profOwnRef = nullptr;         // recorded on this engine's stream by wg_profile_enable
  hipEvent_t profRef = nullptr;            // the reference in use (may belong to another engine)
  std::vector<std::pair<double, double>> profTimes[PC_COUNT];
  // Under stream capture (the device loop replayed as a hipGraph) a HIP event would be re-recorded by every replay and keep
  // the last one only: there the bracket is a pair of one-lane kernels that append (tag, s_memrealtime) records to a device
  // ring (k_prof_stamp) — the same span, on the same stream, read back by prof_collect().
  bool profStamping = false;               // set while the chunk is being captured
  uint64_t* dStamps = nullptr;             // [PROF_STAMP_CAP][2]: tag (class * 2 + end), clock
  uint32_t* dStampCnt = nullptr;
  static constexpr uint32_t PROF_STAMP_CAP = 1u << 18;
  void prof_stamp(int tag);
  struct ProfScope {
    Engine& e;
    size_t idx = (size_t)-1;
    int stamped = -1;
    ProfScope(Engine& en, int cls) : e(en) {
      if (!(e.profiling == 1 || (e.profiling == 2 && cls == PC_DELIVER))) return;
      if (e.profStamping) {
        e.prof_stamp(cls * 2);
        stamped = cls;
        return;
      }
      ProfSpan s{cls, e.prof_event(), e.prof_event()};
      (void)hipEventRecord(s.a, e.stream);
      idx = e.profSpans.size();
      e.profSpans.push_back(s);
    }
    ~ProfScope() {
      if (stamped >= 0) e.prof_stamp(stamped * 2 + 1);
      if (idx != (size_t)-1) (void)hipEventRecord(e.profSpans[idx].b, e.stream);
    }
  };

  template <class T>
  T* dalloc(size_t count, bool zero = true, int cls = AC_STATE) {
    void* p = nullptr;
    WG_HIP(hipMalloc(&p, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    if (zero) WG_HIP(hipMemsetAsync(p, 0, count * sizeof(T), stream));
    if (count * sizeof(T) >= (8u << 20) && getenv("WG_INIT_VERBOSE") && atoi(getenv("WG_INIT_VERBOSE")) >= 2)
      fprintf(stderr, "[wittgpu] alloc #%zu: %.1f MB (class %d, %zu x %zu B)\n", allocs.size(), count * sizeof(T) / 1e6, cls, count, sizeof(T));
    allocs.push_back(p);
    allocInfo.push_back({count * sizeof(T), cls});
    return (T*)p;
  }
  void upload_latency();
  void rebuild_partitions();
};

// A batch of engines advanced in lock-step by one launch sequence (gridDim.y = members): the device
// form of RunMultipleTimes' loop over independent copies (C/RunMultipleTimes.java:44-64).
class Batch {
 public:
  Batch(Engine** es, int n);
  ~Batch();
  void run_ms(int32_t ms, const uint8_t* active, uint8_t* did, wg_run_stats* stats);
  void cont_if(int32_t* out);
  // RunMultipleTimes.run's inner loop for every member, on the device (include/wittgpu.h)
  void run_multiple_times(int32_t chunk, int32_t maxTime, int64_t* delivered, int64_t* simulatedMs);
  std::vector<Engine*> members;
  std::string lastError;

 private:
  Group prepare(const uint8_t* active);
  EngineDev* dTab = nullptr;
  void* dStab = nullptr;
  uint32_t* dCont = nullptr;
  std::vector<EngineDev> hTab;
  std::vector<char> hStab;
  std::vector<PeriodicReg> periodicUnion;  // Group::periodic of the batch: the members' registrations
};

void selftest(int32_t op, int32_t aux, const uint64_t* in, int32_t n, int32_t threads, uint64_t* out, int32_t nOut);  // wg_selftest
void rccl_unique_id(uint8_t* id128);  // ncclGetUniqueId of the dynamically loaded librccl

ProtoHost* make_pingpong_host(Engine& e);
ProtoHost* make_host_proto(Engine& e);
ProtoHost* make_handel_host(Engine& e, const wg_handel_params& p, const wg_handel_init_state& st);
ProtoHost* make_gsf_host(Engine& e, const wg_gsf_params& p, const wg_gsf_init_state& st);
ProtoHost* make_sanfermin_host(Engine& e, const wg_sanfermin_params& p);
ProtoHost* make_casper_host(Engine& e, const wg_casper_params& p);
ProtoHost* make_p2pflood_host(Engine& e, const wg_p2pflood_params& p, const wg_p2pflood_init_state& st);

}  // namespace wg
