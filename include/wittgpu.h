/*
 * wittgpu.h — C ABI of libwittgpu.so, the MI355X-native engine behind Wittgenstein's
 * core.Network scheduler (runMs / send / message queue / NetworkLatency sampling).
 *
 * The reference has no FFI boundary for this path: protocols are written against the Java class
 * core.Network (core/src/main/java/net/consensys/wittgenstein/core/Network.java, "C/Network.java"
 * below). Each entry point cites the reference method(s) it replaces. A JNI shim binding these is
 * shown in INTEGRATION.md. Conventions:
 *   - plain pointers and sizes; caller owns every buffer; the engine copies before returning;
 *   - single caller thread per engine (C/Network.java:7-11); not re-entrant;
 *   - every call returns a wg_status; WG_EINVAL <=> the reference's IllegalArgumentException sites
 *     (C/Network.java:320,371,374,386,427,695), WG_ESTATE <=> its IllegalStateException sites
 *     (:137,250,333,472,599,609,656,671); wg_last_error() has the message text.
 */
#ifndef WITTGPU_H
#define WITTGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wg_engine wg_engine;

/* ABI version of this header. The parameter structs have GROWN across versions (wg_handel_params gained byzantineSuicide /
 * hiddenByzantine in version 3, wg_config gained queue_cap_wide in version 2 and rank_bump_cap, alltoallv, alltoallv_ctx in version 5): a caller compiled against an older header
 * would pass shorter structs than the library reads. A binding checks once, at load time, that wg_abi_version() equals the
 * WG_ABI_VERSION it was compiled with and that wg_abi_struct_size(k) equals its own sizeof — wittgenstein_amd/_lib.py and
 * jni/wittgpu_jni.c (JNI_OnLoad) do; a mismatch is a load error, not a silent over-read.
 * which: 0 wg_config, 1 wg_handel_params, 2 wg_gsf_params, 3 wg_casper_params, 4 wg_sanfermin_params,
 * 5 wg_p2pflood_params, 6 wg_delivery, 7 wg_step_op, 8 wg_run_stats; -1 for an unknown index. */
#define WG_ABI_VERSION 5
int32_t wg_abi_version(void);
int32_t wg_abi_struct_size(int32_t which);
/* One wave / block primitive of the device kernels (DPP reductions and scans, the 8-lane group forms, the ballot multisplit
 * rank) run on the caller's values by a test-only kernel, so that a test can hold it against a host computation — the
 * primitives have no reference counterpart; they are what the per-ms kernels are built from. op: 0 wave sum (32 bit), 1 wave
 * sum (64), 2 wave min (int32), 3 wave max, 4 inclusive scan (32), 5 inclusive scan (64), 6 broadcast of lane aux (32), 7 (64),
 * 8 sum / 9 or (64 bit) / 10 min / 11 max (int32) inside groups of eight lanes, 12 block exclusive scan (32 bit; out[threads]
 * = the total), 13 stable rank among equal bins (in[i] = bin | valid << 32, aux = bits of a bin; out[threads + b] = bin b's
 * count), 14 shuffle from lane (i + aux) % 64, 15 block sum (64). in[n], n <= threads (the rest: zeros / the identity);
 * threads: 64 for the wave forms, a multiple of 64 up to 1024 for the block forms. Needs no engine; wg_last_error(NULL). */
int32_t wg_selftest(int32_t op, int32_t aux, const uint64_t* in, int32_t n, int32_t threads, uint64_t* out, int32_t n_out);

typedef enum {
  WG_OK = 0,
  WG_EINVAL = -1,   /* IllegalArgumentException */
  WG_ESTATE = -2,   /* IllegalStateException */
  WG_ENOMEM = -3,   /* a device pool/ring overflowed (raise the wg_config capacity named in the message) */
  WG_EHIP = -4,     /* HIP runtime error / no device */
  WG_EUNSUPPORTED = -5, /* a reference feature the resident protocol does not cover (message says which) */
  WG_EHOSTINIT = -6     /* init() state the device was asked to build (wg_handel_init_state.peers == NULL) needs the host's
                           sequential rd after all — a rejected nextInt(bound) draw inside a shuffle, p ~ 1e-9 per draw:
                           build it on the host and load again */
} wg_status;

/* in-place SUM of `count` int32 words at buf (device memory) across the shards of one simulation; see
 * "node-range sharding" below */
typedef int32_t (*wg_allreduce_fn)(void* ctx, void* buf, int64_t count);

/* Engine capacities. 0 = pick a default from node count / protocol. */
typedef struct {
  int32_t device;               /* HIP device ordinal */
  int32_t horizon_ms;           /* power of two > max(arrival - time); ring of per-ms buckets (C/Network.java:116-132) */
  int64_t bucket_pool_records;  /* total in-flight envelope records (paged) */
  int64_t payload_words;        /* 64-bit words of in-flight message payload (bitsets) */
  int64_t outbox_records;       /* max records emitted within one simulated ms */
  int64_t chain_dests;          /* total destination ids held by in-flight multi-destination envelopes */
  int32_t chain_slots;          /* in-flight multi-destination envelopes (C/Envelope.java:57) */
  int32_t queue_cap;            /* Handel: per (node, level) toVerifyAgg capacity (P/Handel.java:385), <= 64;
                                   GSFSignature: per node toVerify capacity (P/GSFSignature.java:167), <= 512 */
  /* node-range sharding (same effect as wg_shard_configure right after wg_create); nshards == 0: not sharded */
  int32_t shard, nshards;
  wg_allreduce_fn allreduce;
  void* allreduce_ctx;
  /* ... or, with allreduce == NULL, the 128-byte RCCL unique id of wg_rccl_unique_id: the engine creates and owns the
   * communicator (same effect as wg_shard_configure_rccl right after wg_create) */
  const uint8_t* rccl_id;
  int32_t queue_cap_wide;       /* Handel: toVerifyAgg capacity of the levels whose block is >= 16 words (>= 1024 ids), whose
                                   queues stay short and whose slots are large; 0 = min(queue_cap, 16). Overflow is loud
                                   (WG_ENOMEM), as for queue_cap */
  int32_t rank_bump_cap;        /* Handel with the reception ranks carried by the senders (init() on the device, unsharded, no
                                   attack, <= 65 536 nodes): senders per node whose rank checkSigs may bump (P/Handel.java:825-828)
                                   — one per distinct sender a node ever verifies; rounded up to a power of two, at most
                                   nodeCount. 0 = min(nodeCount, 512): a node of config 3 verifies about 100 senders. Overflow is loud (WG_ENOMEM) */
  /* the sharded engine's all-to-all (wg_shard_set_alltoallv right after wg_create; with `allreduce`; NULL: none). Declared
   * below; same calling convention as the typedef there */
  int32_t (*alltoallv)(void* ctx, const void* sendbuf, const int64_t* send_counts, const int64_t* send_offsets, void* recvbuf,
                       const int64_t* recv_counts, const int64_t* recv_offsets);
  void* alltoallv_ctx;
} wg_config;

/* ---- lifecycle -------------------------------------------------------------------------- */
int32_t wg_create(const wg_config* cfg, wg_engine** out);       /* new Network<>()  C/Network.java:14-49 */
void wg_destroy(wg_engine* e);
const char* wg_last_error(wg_engine* e);                        /* e may be NULL: error of a failed wg_create */

/* ---- topology --------------------------------------------------------------------------- */
/* Network.addNode for n nodes with dense ids continuing from the current count (C/Network.java:651-659);
 * fields are Node's (C/Node.java:22-79). down/byzantine/speedRatio/extraLatency may be NULL (0 / 0 / 1.0 / 0). */
int32_t wg_add_nodes(wg_engine* e, int32_t n, const int32_t* x, const int32_t* y, const int32_t* extraLatency,
                     const uint8_t* down, const uint8_t* byzantine, const double* speedRatio);
int32_t wg_node_count(wg_engine* e);

typedef enum {
  WG_LAT_BY_DISTANCE_WJITTER = 0, /* NetworkLatency.NetworkLatencyByDistanceWJitter  C/NetworkLatency.java:49-73 */
  WG_LAT_FIXED = 1,               /* NetworkFixedLatency(params[0])    :235-249 */
  WG_LAT_UNIFORM = 2,             /* NetworkUniformLatency(params[0])  :255-269 */
  WG_LAT_NONE = 3,                /* NetworkNoLatency                  :271-275 */
  WG_LAT_MEASURED = 4,            /* MeasuredNetworkLatency, params = longDistrib[100]  :277-313 */
  WG_LAT_IC3 = 5,                 /* IC3NetworkLatency                 :399-417 */
  WG_LAT_ETHSCAN = 6              /* EthScanNetworkLatency             :366-384 */
} wg_latency_kind;
/* Network.setNetworkLatency (C/Network.java:666-678): WG_ESTATE if messages are in flight. */
int32_t wg_set_latency(wg_engine* e, int32_t kind, const int32_t* params, int32_t nparams);
/* The city-based models (C/NetworkLatency.java:86-233). Their inputs are the caller's: Node.cityName as an index per
 * node into the caller's own city list (city_of_node[node_count]; the nodes must have been added) and, for that list,
 *   WG_CITY_AWS              tab[C*C] = latencies[min][max] / 2 both ways (:113-133), jitter[100] = gpd.inverseF(delta / 100.0)
 *                            of NetworkLatencyByDistanceWJitter (:53-65); same region -> 1
 *   WG_CITY_BY_CITY          tab[C*C] = max(1, Math.round(0.5f * ping)) with CSVLatencyReader's matrix (T/CSVLatencyReader.java:
 *                            303-312: 30 inside a city, the to->from fallback of :187-197); ping, jitter unused (NULL)
 *   WG_CITY_BY_CITY_WJITTER  ping[C*C] (float, both ways), jitter[100] (double) of GeneralizedParetoDistribution(1.4, -0.3, 0.35)
 * — the FP of these models that depends on (from, to, delta) jointly (:228-231) is evaluated on the device in IEEE double,
 * as the JVM does. Node.extraLatency, `from == to -> 1` and max(1, ...) are NetworkLatency.getLatency's (:27-34), as for
 * every model. */
typedef enum { WG_CITY_AWS = 0, WG_CITY_BY_CITY = 1, WG_CITY_BY_CITY_WJITTER = 2 } wg_city_latency_mode;
int32_t wg_set_latency_city(wg_engine* e, int32_t mode, int32_t n_cities, const int32_t* city_of_node,
                            const int32_t* tab, const float* ping, const double* jitter100);
/* RegistryNetworkLatencies.getByName (C/RegistryNetworkLatencies.java:42-58); NULL = ByDistanceWJitter */
int32_t wg_set_latency_by_name(wg_engine* e, const char* name);
/* getLatency(from, to, delta) of the installed model, evaluated by the device kernel (tests, estimateLatency) */
int32_t wg_latency_probe(wg_engine* e, int32_t n, const int32_t* from, const int32_t* to, const int32_t* delta,
                         int32_t* out);
int32_t wg_set_partitions(wg_engine* e, const int32_t* xcuts, int32_t k); /* partition()/endPartition()  :693-707 */
int32_t wg_set_node_down(wg_engine* e, int32_t id, int32_t down);        /* Node.stop()/start()  C/Node.java:120-131 */
int32_t wg_set_discard_time(wg_engine* e, int32_t ms);                   /* setMsgDiscardTime  C/Network.java:103-107 */

/* ---- RNG: the single shared java.util.Random `rd` (C/Network.java:32) --------------------- */
int32_t wg_rng_set_seed(wg_engine* e, int64_t seed);            /* rd.setSeed  C/RunMultipleTimes.java:47 */
int32_t wg_rng_get_state(wg_engine* e, uint64_t* s48);
int32_t wg_rng_set_state(wg_engine* e, uint64_t s48);

/* ---- host-side sends / tasks (init() code paths; action() of resident protocols runs on device) */
/* Network.send(m, sendTime, from, dests, delayBetween) C/Network.java:369-382,418-447 (n==1: single-dest
 * overload). msg = protocol message word, payload = protocol payload handle. Draws one rd.nextInt() — also for
 * n == 0, as the list overload does (:430; the 3-argument overload that returns early on an empty list, :353-356,
 * is the caller's business). */
int32_t wg_send(wg_engine* e, uint32_t msg, uint32_t payload, int32_t sendTime, int32_t from, const int32_t* dests,
                int32_t n, int32_t delayBetween);
/* Network.sendArriveAt (C/Network.java:384-390): no latency, no rd draw; WG_EINVAL if arriveAt <= time */
int32_t wg_send_arrive_at(wg_engine* e, uint32_t msg, uint32_t payload, int32_t arriveAt, int32_t from, int32_t to);
/* Network.registerTask / registerPeriodicTask (C/Network.java:505-519): task = protocol task word */
int32_t wg_register_task(wg_engine* e, uint32_t task, uint32_t arg, int32_t startAt, int32_t node);
int32_t wg_register_periodic_task(wg_engine* e, uint32_t task, int32_t startAt, int32_t period, int32_t node);

/* ---- resident protocols ----------------------------------------------------------------- */
typedef enum {
  WG_PROTO_HOST = 0, /* no resident protocol: Message.action() stays with the caller (wg_next_delivery below) */
  WG_PROTO_PINGPONG = 1, WG_PROTO_HANDEL = 2, WG_PROTO_GSF = 3, WG_PROTO_SANFERMIN = 4, WG_PROTO_CASPER = 5,
  WG_PROTO_P2PFLOOD = 6
} wg_proto_id;

/* P2PFlood parameters: P2PFloodParameters ctor order (P/P2PFlood.java:63-86); msgCount <= 64 on the device.
 * Init state produced by P2PFlood.init() (:121-140) on the host: the peer lists P2PNetwork.setPeers built (row i =
 * node i's peers in list order, peerCount[i] of them, rows maxPeers <= 64 wide) and the sender of each message
 * (sendPeers' addToReceived and `doneAt = 1` are applied on the device; the sends themselves are wg_send calls with
 * delaysBetweenMessage). */
typedef struct {
  int32_t nodeCount, deadNodeCount, delayBeforeResent, msgCount, msgToReceive, peersCount, delayBetweenSends;
} wg_p2pflood_params;
typedef struct {
  const int32_t* peers;      /* [nodeCount][maxPeers] */
  const int32_t* peerCount;  /* [nodeCount] */
  int32_t maxPeers;
  const int32_t* senders;    /* [msgCount] */
} wg_p2pflood_init_state;

/* Casper IMD parameters: CasperParemeters ctor order (sic, P/CasperIMD.java:52-70), then the delay of the
 * ByzBlockProducerWF that init(badNode) starts with (:475-479 uses 0) and a capacity: the run may reach slot maxSlots
 * (a slot is 8000 ms; attestation and block tables are sized by it). Nodes: 0 the observer, 1 the byzantine producer,
 * 2..blockProducersCount the other producers, then cycleLength * attestersPerRound attesters (:481-509).
 * randomOnTies (:250-253; the reference's default is true) is honoured: the tie's rd.nextBoolean() decides a head inside
 * action(), so once the chain has forked the events that can call best() are delivered by one wavefront in global event
 * order (exact; the parallel path until then; on a sharded engine the ordered visit goes round the shards, see wg_shard_configure). init() =
 * wg_register_periodic_task per node in the reference's order: task words 2 (byzantine producer), 0 (producer), 1 (attester). */
typedef struct {
  int32_t cycleLength, randomOnTies, blockProducersCount, attestersPerRound, blockConstructionTime,
      attestationConstructionTime;
  int32_t byzDelay, maxSlots;
} wg_casper_params;

/* San Fermin parameters: SanFerminSignatureParameters ctor order (P/SanFerminSignature.java:84-104; shuffledLists is
 * read nowhere in the protocol). nodeCount must be a power of two (toBinaryID, P/SanFerminHelper.java:158-171).
 * No init state: the nodes' swap state starts from constants (:202-219); init() registers goNextLevel at t = 1 for
 * every node (:139-141) = wg_register_task(e, 0, 0, 1, node) in id order. */
typedef struct {
  int32_t nodeCount, threshold, pairingTime, signatureSize, replyTimeout, candidateCount;
} wg_sanfermin_params;

/* Handel parameters: HandelParameters ctor order (P/Handel.java:97-142) + WindowParameters (:147-174) */
typedef struct {
  int32_t nodeCount, threshold, pairingTime, levelWaitTime, extraCycle, disseminationPeriodMs, fastPath, nodesDown;
  int32_t desynchronizedStart;
  int32_t windowInitial, windowMinimum, windowMaximum; /* 16, 1, 128 */
  /* HandelParameters.byzantineSuicide / hiddenByzantine (P/Handel.java:64-71, 108-109): the down nodes are byzantine and
   * attack — byzantineSuicide :406, 538-559, 577-584, 688-694; hiddenByzantine :813-817, 840-917. Both are resident on the
   * device (unsharded engines; every node visit then takes the wave-per-node kernels); at most one of them (WG_EINVAL). */
  int32_t byzantineSuicide, hiddenByzantine;
} wg_handel_params;

/* Per-node state produced by Handel.init() (P/Handel.java:957-1014), uploaded once:
 *   startAt[n], nodePairingTime[n]   HNode fields (:280-283)
 *   receptionRanks[n*n]              row i = HNode i's receptionRanks (:285, :940-948)
 *                                    NULL (with peers NULL): the engine runs setReceivingRanks' Collections.shuffle of the one
 *                                    shared list once per node (:940-948, 966-989) on the device, from rd where init() has it
 *                                    at that point; same limits and WG_EHOSTINIT as for peers.
 *   peers[n*(n-1)]                   row i = concatenation over levels 1..L-1 of HLevel.peers (emission
 *                                    order, :510-522); level l occupies [2^(l-1)-1, 2^l-1). Ignored for down nodes.
 *                                    NULL: the engine builds the lists on the device from receptionRanks (buildEmissionList
 *                                    :510-522 for every live sender: sort by the receivers' rank of the sender, shuffle equal
 *                                    ranks with rd) — rd must be where init() has it at that point (wg_rng_set_state before
 *                                    the load) and comes back advanced by the shuffles' draws; unsharded engines of up to
 *                                    131 072 nodes; WG_EHOSTINIT if a draw was rejected (then load again with host-built lists). */
typedef struct {
  const int32_t* startAt;
  const int32_t* nodePairingTime;
  const int32_t* receptionRanks;
  const int32_t* peers;
} wg_handel_init_state;

/* GSFSignature parameters: GSFSignatureParameters ctor order (P/GSFSignature.java:59-84) */
typedef struct {
  int32_t nodeCount, threshold, pairingTime, timeoutPerLevelMs, periodDurationMs, acceleratedCallsCount, nodesDown;
} wg_gsf_params;

/* Per-node state produced by GSFSignature.init() (P/GSFSignature.java:611-635), uploaded once:
 *   nodePairingTime[n]   GSFNode.nodePairingTime (:170)
 *   peers[n*(n-1)]       row i = concatenation over levels 1..L-1 of SFLevel.peers (randomSubset's shuffle of the
 *                        level's waitedSigs, :278, :462-476); level l occupies [2^(l-1)-1, 2^l-1). Ignored for
 *                        down nodes (they never get levels, :627-634). */
typedef struct {
  const int32_t* nodePairingTime;
  const int32_t* peers;
} wg_gsf_init_state;

int32_t wg_protocol_load(wg_engine* e, int32_t proto_id, const void* params, const void* init_state);

/* ---- run -------------------------------------------------------------------------------- */
typedef struct {
  int64_t delivered;     /* sum of msgReceived increments (C/Network.java:607-613) during this call */
  int64_t tasks;         /* Task envelopes executed */
  int64_t events;        /* envelopes polled (delivered + tasks + skipped because down/partitioned) */
  int64_t draws;         /* rd draws consumed */
  int64_t simulated_ms;
  int64_t wall_ns;       /* host wall time of the call, device idle at both ends */
  int64_t payload_bytes; /* message payload bytes read at delivery (roofline accounting) */
} wg_run_stats;
/* Network.runMs (C/Network.java:318-338). stats may be NULL. */
int32_t wg_run_ms(wg_engine* e, int32_t ms, uint8_t* didSomething, wg_run_stats* stats);
int32_t wg_time(wg_engine* e, int32_t* time);                   /* Network.time  :49 */
int32_t wg_queue_size(wg_engine* e, int64_t* size);             /* msgs.size()   :204-210 */
int32_t wg_queue_size_at(wg_engine* e, int32_t t, int64_t* size); /* msgs.sizeAt(t) :212-220 */

/* The resident protocol's continuation predicate of the RunMultipleTimes loop (C/RunMultipleTimes.java:50-64),
 * evaluated on the device: Handel.newContIf (P/Handel.java:1044-1053), GSFSignature.newConfIf
 * (P/GSFSignature.java:670-683). *cont = 1 while the run must go on. */
int32_t wg_protocol_cont_if(wg_engine* e, int32_t* cont);

/* ---- init() image: RunMultipleTimes without re-running init() ---------------------------------------- */
/* C/RunMultipleTimes.java:44-48 builds every run from scratch: `p.copy(); rd.setSeed(i); init()`. init() is sequential
 * host work (Handel: nodeCount cumulative Collections.shuffle calls, P/Handel.java:940-948) that depends only on the
 * parameters and the seed, so for a seed that is run more than once the engine can keep what init() produced:
 *   wg_snapshot  after init() (nodes, protocol state, host-side sends / task registrations) and before the first
 *                event is polled (WG_ESTATE otherwise): keeps an image of the engine in device memory —
 *                Network.time, rd, node flags, the queued envelopes, the resident protocol's state;
 *   wg_restore   puts the engine back to that image (device-to-device; what the protocol can recompute in place is
 *                recomputed: Handel's receptionRanks, whose only mutation is `+= nodeCount`, P/Handel.java:825-828).
 * Every observable after a restore + run equals a fresh engine's after init() + run (tests/test_snapshot_*.py).
 * wg_snapshot_bytes: size of the image (0 without one). Not for host-callback mode (WG_EUNSUPPORTED). */
int32_t wg_snapshot(wg_engine* e);
int32_t wg_restore(wg_engine* e);
int32_t wg_snapshot_bytes(wg_engine* e, int64_t* bytes);

/* ---- host-callback mode: any protocol, action() stays in the caller ------------------------------------ */
/* For protocols without a resident device form (the reference's San Fermin, Paxos, Slush, Dfinity, P2P* ...): the
 * message queue, its LIFO / chain ordering (C/Network.java:116-299, C/Envelope.java:57-301), NetworkLatency
 * sampling and the shared rd live in the engine; the caller keeps its Node / Message objects and runs action().
 * After wg_protocol_load(e, WG_PROTO_HOST, NULL, NULL):
 *   wg_send / wg_register_task take caller-chosen 32-bit handles as msg / payload (task / arg);
 *   wg_next_delivery is Network.nextMessage (C/Network.java:533-570) plus the post-action part of receiveUntil
 *   (:625-632): it returns the next envelope to deliver in exactly the reference's order — kind 0 message, 1 task —
 *   for the caller to apply (stats :607-613 and action() :616-626 are the caller's), or a time edge (kind 2) when
 *   `time++` reached cond_time, the earliest ConditionalTask.minStartTime the caller holds (INT32_MAX: none), so
 *   that it can run the conditional-task scan of :543-566 at that edge; *got = 0 once time > until.
 *   Deliveries to a down node or across a partition are consumed silently (:606). Envelopes pushed by the caller
 *   while a delivery is being applied land exactly where Java's msgs.addMsg would put them.
 *   wg_set_time is the tail of Network.runMs (`time = endAt`, :336). */
typedef struct {
  int32_t kind;     /* 0 message, 1 task, 2 time edge */
  int32_t time;     /* Network.time at the delivery (= arrival) / the new time of an edge */
  int32_t from, to; /* node ids (task: from == to) */
  uint32_t msg;     /* the handle given to wg_send / wg_register_task */
  uint32_t payload; /* second handle word (wg_send payload / wg_register_task arg) */
} wg_delivery;
int32_t wg_next_delivery(wg_engine* e, int32_t until, int32_t cond_time, wg_delivery* out, int32_t* got);
int32_t wg_set_time(wg_engine* e, int32_t time);
/* The same, a millisecond at a time (SURVEY.md 8(b) wg_step_begin / wg_step_end): one call hands out every deliverable
 * envelope left in the current ms (at most cap; *n = 0 once time > until; a time edge is a batch of one kind-2 entry), the
 * caller applies them in order, and ONE call takes back what their action()s pushed — the shape of
 * External.receive(EnvelopeInfo) -> List<SendMessage> (C/Network.java:616-623): two FFI crossings per simulated ms instead
 * of one per delivered message. Every op names the delivery that issued it (`after`, index into the batch; ops ordered by
 * it), so the engine files the pushes exactly where Java's msgs.addMsg would have: delivery i's, then the re-push of i's
 * multi-destination envelope (:629-632). The shared rd is the CALLER's while a step is open (wg_rng_get_state at
 * wg_step_begin, wg_rng_set_state before wg_step_end): Network.send's seed draw (:377, :430) happens inside the
 * caller's action(), in order with the action()'s own draws, and travels in the op. Node up / down and partition changes
 * made by an action() apply from the next step on; a task registered for the ms being delivered is accepted only from
 * the step's last delivery (WG_EUNSUPPORTED otherwise: use cap = 1 or wg_next_delivery for such a protocol). */
typedef enum { WG_OP_SEND = 0, WG_OP_SEND_ARRIVE_AT = 1, WG_OP_TASK = 2 } wg_step_op_kind;
typedef struct {
  int32_t after;     /* the delivery (index in wg_step_begin's batch) whose action() issued it */
  int32_t kind;      /* wg_step_op_kind */
  uint32_t msg;      /* message / task handle */
  uint32_t payload;  /* second handle word (task: arg) */
  int32_t time;      /* SEND: sendTime, SEND_ARRIVE_AT: arriveAt, TASK: startAt */
  int32_t from;      /* sender (TASK: the node) */
  int32_t to;        /* n == 1: the destination; n > 1: offset of the n destinations in `dests` */
  int32_t n;         /* SEND: number of destinations (an empty list is not an op: its draw is the caller's) */
  int32_t delay;     /* SEND: delaysBetweenMessage */
  int32_t seed;      /* SEND: the rd.nextInt() Network.send drew */
} wg_step_op;
int32_t wg_step_begin(wg_engine* e, int32_t until, int32_t cond_time, wg_delivery* out, int32_t cap, int32_t* n);
int32_t wg_step_end(wg_engine* e, const wg_step_op* ops, int32_t nops, const int32_t* dests);
/* Handles whose envelope has ENDED since the last call, oldest first (at most cap; *n of them): the message / task handle of
 * every envelope made by wg_send, wg_send_arrive_at, wg_register_task or a step op is reported exactly once — after its one
 * destination was handed out or consumed (C/Network.java:606), after the last hop of a multi-destination envelope was
 * (:629-632 finds hasNextReader() false), or at once when no destination was reachable at send time (:469-487). It is where
 * the reference lets go of its Envelope: a binding that maps handles to Message objects releases them here (a handle sent
 * k times is reported k times). A handle reported together with its last delivery is released AFTER that delivery is
 * applied. Host-callback mode only. */
int32_t wg_host_released(wg_engine* e, uint32_t* msgs, int32_t cap, int32_t* n);

/* ---- batches: RunMultipleTimes on the device ------------------------------------------------ */
/* The reference's way to run many simulations is C/RunMultipleTimes.java:44-64: for each of runCount
 * copies, rd.setSeed(i); init(); runMs(10) while the continuation predicate holds. A wg_batch advances n
 * such independent engines in lock-step with ONE launch sequence per simulated ms (gridDim.y = member),
 * which is how a 288 GB MI355X is kept busy by a latency-bound event loop. Members must share device,
 * resident protocol, node count and horizon_ms; each keeps its own state, rd stream and clock, and every
 * per-engine call (wg_read_*, wg_time, ...) stays valid. The batch does not own its members. */
typedef struct wg_batch wg_batch;
int32_t wg_batch_create(wg_engine** engines, int32_t n, wg_batch** out);
void wg_batch_destroy(wg_batch* b);
const char* wg_batch_last_error(wg_batch* b);       /* b may be NULL: error of a failed wg_batch_create */
int32_t wg_batch_size(wg_batch* b, int32_t* n);     /* members of the batch: the length of every per-member array below */
/* Network.runMs(ms) on every member with active[i] != 0 (NULL = all); the others are not advanced
 * (a copy whose predicate turned false stops, C/RunMultipleTimes.java:56-61). didSomething / stats: [n] or NULL. */
int32_t wg_batch_run_ms(wg_batch* b, int32_t ms, const uint8_t* active, uint8_t* didSomething, wg_run_stats* stats);
int32_t wg_batch_cont_if(wg_batch* b, int32_t* cont); /* wg_protocol_cont_if for every member, one launch; cont[n] */
/* RunMultipleTimes.run's inner loop (C/RunMultipleTimes.java:50-64) for every member, evaluated on the device:
 *   do { didSomething = runMs(chunk); } while ((maxTime == 0 || time < maxTime) && (!didSomething || contIf(p)));
 * Chunks are enqueued back to back with no host round trip per runMs; a member whose loop ended is no longer
 * advanced. delivered[n] / simulatedMs[n] (NULL allowed): msgReceived increments and Network.time advance. */
int32_t wg_batch_run_multiple_times(wg_batch* b, int32_t chunk, int32_t maxTime, int64_t* delivered,
                                    int64_t* simulatedMs);

/* ---- node-range sharding: ONE simulation over several engines (one process per GPU) --------- */
/* (SURVEY.md 8(b) sketches this group as `wg_config.devices[]` / `.shards` with "a single host thread driving all devices".
 * What is built is ONE PROCESS PER GPU, each with its own wg_engine and wg_config.{shard, nshards, rccl_id}: it is how
 * RCCL / torch.distributed address a box's GPUs, it keeps "single caller thread per engine" true, and a Java host starts
 * one JVM per GPU the way it would start one per seed range. shards.LoopbackGroup is the in-process form for tests.) */
/* The reference is single-threaded (C/Network.java:7-11); this group has no counterpart there. Every shard is a
 * full wg_engine built by the SAME sequence of calls (nodes, latency, seed, protocol, host-side sends/tasks) in its
 * own process; shard s of S owns the nodes [N*s/S, N*(s+1)/S). The scheduler state (per-ms buckets, multi-destination
 * envelopes, rd, clock) is replicated and evolves identically on every shard; Message.action() runs, and node /
 * protocol state lives, only on the owner of the destination node. Bit-exactness with the unsharded engine (global
 * LIFO order, one shared rd stream) is kept by two sums across shards per simulated ms:
 *   1. per-event (records emitted, rd draws) after delivery          -> every shard derives the global push order
 *                                                                        and the rd index of every send;
 *   2. the resolved outbox (record, arrival) after latency sampling  -> every shard appends the same records to
 *                                                                        the same buckets.
 * The engine calls `allreduce` (in-place SUM of count int32 words at buf, which is DEVICE memory of this engine's
 * device) with its stream idle; the function returns once the summed values are visible to any stream. Bind it to
 * RCCL (torch.distributed backend "nccl": wittgenstein_amd/shards.py) — over xGMI the payload is a few bytes per event,
 * so the latency of the collective, not its bandwidth, is what a simulated ms pays.
 * Call before the engine allocates (before wg_protocol_load / the first wg_send). Resident protocols: PingPong, Handel, GSFSignature, San Fermin,
 * P2PFlood, Casper IMD (with randomOnTies the ms's blocks and tasks are visited in event order round the shards: one two-word
 * collective per change of owner among them — exact, slow).
 * wg_read_i64 on a shard returns its own nodes' values and zeros for the others (sum across shards for the whole
 * network); wg_run_stats counts are whole-network on every shard. WG_EUNSUPPORTED for a protocol that does not
 * shard yet, batches, and host-callback mode.
 * Limits of a sharded engine (loud, WG_EUNSUPPORTED "sharded engine: one event ..."): an event's result travels as ONE
 * packed int32 word (records 10 bits, draws 11 bits), so a single action() may emit at most 1023 records and make at
 * most 2047 rd draws — far beyond what the resident protocols do (Handel: one record per level). */
int32_t wg_shard_configure(wg_engine* e, int32_t shard, int32_t nshards, wg_allreduce_fn allreduce, void* ctx);
/* The second collective of a sharded engine (round 5): an all-to-all of int32 words — shard s sends send_counts[d] words
 * starting at word send_offsets[d] of sendbuf to every shard d and receives recv_counts[r] words at word recv_offsets[r] of
 * recvbuf from every shard r (arrays of nshards entries; both sides know their counts; buffers are DEVICE memory of this
 * engine's device; the engine's stream is idle during the call). It carries what only ONE shard needs: the payload a
 * message of a node of one shard brings to a node of another — Handel's dissemination snapshots (SendSigs.sigs,
 * P/Handel.java:254: up to nodeCount / 16 bytes a message) go to the shard that owns the receiver, not to every shard.
 * Optional: without it those rows travel inside an all-reduce image, to everybody (the form of rounds 1-4). Call after
 * wg_shard_configure and before wg_protocol_load. An engine that owns its RCCL communicator (wg_shard_configure_rccl)
 * does the same with grouped ncclSend / ncclRecv on its own stream and needs no callback. */
typedef int32_t (*wg_alltoallv_fn)(void* ctx, const void* sendbuf, const int64_t* send_counts, const int64_t* send_offsets,
                                   void* recvbuf, const int64_t* recv_counts, const int64_t* recv_offsets);
int32_t wg_shard_set_alltoallv(wg_engine* e, wg_alltoallv_fn fn, void* ctx);
/* The same with the collective OWNED BY THE ENGINE: an RCCL communicator over the box's GPUs (xGMI), created from a
 * unique id that shard 0's process obtains with wg_rccl_unique_id and hands to the other processes by whatever channel
 * the host application has (the Java host: its own launcher; wittgenstein_amd/shards.py: one torch.distributed
 * broadcast at start-up). The per-ms sums are then ncclAllReduce calls enqueued on the engine's own HIP stream — no
 * callback into the caller, no host synchronisation per collective. librccl.so.1 of the ROCm installation libwittgpu.so
 * was built against is loaded on first use (WG_EHIP if absent). Call before the engine allocates. */
#define WG_RCCL_UNIQUE_ID_BYTES 128
int32_t wg_rccl_unique_id(uint8_t* id128);
int32_t wg_shard_configure_rccl(wg_engine* e, int32_t shard, int32_t nshards, const uint8_t* id128);
/* [lo, hi) of this shard (valid once the nodes are added); collectives / int32 words exchanged so far */
int32_t wg_shard_info(wg_engine* e, int32_t* lo, int32_t* hi, int64_t* collectives, int64_t* words);
/* the same two totals split by exchange (eight entries each; no reference counterpart, statistics only): [0] the events'
 * packed result words, [1] the resolved outbox (record + arrival), [2] multi-destination envelopes, [3] payload snapshots —
 * words RECEIVED where they travel owner-directed (wg_shard_set_alltoallv / an engine-owned communicator), the all-reduce
 * image's words otherwise —, [4] the checkSigs edge's candidate counts, [5] the count matrix of the owner-directed
 * exchange, [6] everything else (Casper's tables, its ordered visit's hand-overs), [7] 0 */
int32_t wg_shard_traffic(wg_engine* e, int64_t* calls8, int64_t* words8);

/* ---- read-back -------------------------------------------------------------------------- */
typedef enum {
  /* Node counters (C/Node.java:69-79) */
  WG_F_DONE_AT = 0, WG_F_MSG_RECEIVED = 1, WG_F_MSG_SENT = 2, WG_F_BYTES_SENT = 3, WG_F_BYTES_RECEIVED = 4,
  WG_F_DOWN = 5, WG_F_X = 6, WG_F_Y = 7, WG_F_EXTRA_LATENCY = 8,
  /* PingPong (P/PingPong.java:61) */
  WG_F_PONG = 16,
  /* Handel HNode (P/Handel.java:280-299) */
  WG_F_SIGS_CHECKED = 32, WG_F_SIG_QUEUE_SIZE = 33, WG_F_MSG_FILTERED = 34, WG_F_CURR_WINDOW_SIZE = 35,
  WG_F_ADDED_CYCLE = 36, WG_F_START_AT = 37, WG_F_NODE_PAIRING_TIME = 38,
  /* GSFSignature GSFNode (P/GSFSignature.java:166-175): sigChecked, sigQueueSize, toVerify.size(),
   * verifiedSignatures.cardinality() */
  WG_F_GSF_SIG_CHECKED = 48, WG_F_GSF_SIG_QUEUE_SIZE = 49, WG_F_GSF_TO_VERIFY_SIZE = 50,
  WG_F_GSF_VERIFIED_CARDINALITY = 51,
  /* San Fermin SanFerminNode (P/SanFerminSignature.java:149-221): aggValue, currentPrefixLength, done |
   * thresholdDone << 1 | isSwapping << 2, sentRequests, receivedRequests, thresholdAt */
  WG_F_SF_AGG_VALUE = 64, WG_F_SF_PREFIX_LENGTH = 65, WG_F_SF_FLAGS = 66, WG_F_SF_SENT_REQUESTS = 67,
  WG_F_SF_RECEIVED_REQUESTS = 68, WG_F_SF_THRESHOLD_AT = 69,
  /* Casper IMD CasperNode (P/CasperIMD.java:177-368): head.height, head.proposalTime, head.id (creation order, genesis 0),
   * attestationsByHead.size(), blocksReceivedByBlockId.size(), attestations held (all heads) */
  WG_F_CASPER_HEAD_HEIGHT = 80, WG_F_CASPER_HEAD_TIME = 81, WG_F_CASPER_HEAD_ID = 82, WG_F_CASPER_HEADS_ATTESTED = 83,
  WG_F_CASPER_BLOCKS_RECEIVED = 84, WG_F_CASPER_ATTESTATIONS_HELD = 85,
  /* P2PFlood P2PNode (C/P2PNode.java): getMsgReceived(-1).size(), peers.size() */
  WG_F_FLOOD_RECEIVED = 96, WG_F_FLOOD_PEER_COUNT = 97
} wg_field;
int32_t wg_read_i64(wg_engine* e, int32_t field, int64_t* dst, int32_t n);
/* the same fields narrowed to 32 bits, as SURVEY.md 8(b) spells the int-valued read-back (pong, sigsChecked, msgFiltered, ...):
 * WG_EINVAL if a value does not fit (the long counters of C/Node.java:75-79 can exceed it: use wg_read_i64 for those) */
int32_t wg_read_i32(wg_engine* e, int32_t field, int32_t* dst, int32_t n);
typedef enum { /* per (node, level), row-major [node][level] */
  WG_LF_POS_IN_LEVEL = 0, WG_LF_OUTGOING_FINISHED = 1, WG_LF_QUEUE_LEN = 2,
  WG_LF_REMAINING_CALLS = 3, /* GSF SFLevel.remainingCalls (P/GSFSignature.java:257) */
  /* Handel HNode.receptionRanks (P/Handel.java:285, 825-828) as they are now: per (node, sender), n_levels = nodeCount */
  WG_LF_RECEPTION_RANKS = 4,
  WG_LF_SUICIDE_BIZ_AFTER = 5 /* Handel HLevel.suicideBizAfter (P/Handel.java:406) */
} wg_level_field;
int32_t wg_read_level_i32(wg_engine* e, int32_t field, int32_t* dst, int32_t n_nodes, int32_t n_levels);
typedef enum { /* Handel HLevel bitsets (P/Handel.java:373-394) as one nodeCount-bit row per node, bit j = node j */
  WG_B_TOTAL_INCOMING = 0, WG_B_LAST_AGG_VERIFIED = 1, WG_B_VERIFIED_IND = 2, WG_B_TO_VERIFY_IND = 3,
  WG_B_FINISHED_PEERS = 4,
  WG_B_BLACKLIST = 5, /* HNode.blacklist (P/Handel.java:289; all zeros unless byzantineSuicide) */
  /* GSFSignature: GSFNode.verifiedSignatures (= the union of SFLevel.verifiedSignatures, :171,:244) and the
   * unions over levels of SFLevel.individualSignatures / indivVerifiedSig (:245-246) */
  WG_B_GSF_VERIFIED = 8, WG_B_GSF_INDIVIDUAL = 9, WG_B_GSF_INDIV_VERIFIED = 10
} wg_bits_field;
int32_t wg_read_bits(wg_engine* e, int32_t field, uint64_t* dst, int32_t n_nodes, int32_t words_per_node);
int32_t wg_levels(wg_engine* e, int32_t* levels);
/* bytes of device memory the engine holds (every allocation of the engine and its resident protocol; a sharded
 * engine: this shard's) — capacity planning for configurations that need several GPUs */
int32_t wg_device_bytes(wg_engine* e, int64_t* bytes);
/* per-level count of SendSigs delivered so far (roofline accounting); dst[32] */
int32_t wg_delivered_by_level(wg_engine* e, int64_t* dst32);

/* ---- measurement ------------------------------------------------------------------------- */
/* Per-phase device time measured with HIP events recorded on the engine's own stream around each
 * kernel (group) of the per-ms pipeline. No reference counterpart (the reference's only timing is the
 * wall-clock print of C/ProgressPerTime.java:68,96,111). Off by default: each span costs two event records. */
typedef struct {
  const char* name;   /* static string: phase and the kernel(s) it brackets */
  int64_t spans;      /* bracketed launches (groups) since the engine was created */
  double total_ns;    /* summed device time of those spans */
} wg_profile_entry;
/* mode 0 = off, 1 = every phase, 2 = the delivery kernel only (cheap enough to leave on in a timed run).
 * Resets the accumulated spans. In a batch the spans are recorded on the first member. */
int32_t wg_profile_enable(wg_engine* e, int32_t mode);
int32_t wg_profile_read(wg_engine* e, wg_profile_entry* dst, int32_t cap, int32_t* n);
/* Spans on a common time axis. wg_profile_set_reference(e, ref) makes e keep, for every span it brackets from now on, start and
 * end in ns since a reference event of engine `ref` (NULL: e's own; recorded once, on ref's stream). Engines of concurrently
 * running batches given the same reference can have their spans merged — e.g. the union of the intervals in which a delivery
 * kernel of ANY batch runs. wg_profile_read_spans copies the spans of phase `cls` (the index of its wg_profile_read entry: 2 =
 * deliver) recorded since the last full read; *n = how many there are (call with cap 0 to ask). */
int32_t wg_profile_set_reference(wg_engine* e, wg_engine* ref);
int32_t wg_profile_read_spans(wg_engine* e, int32_t cls, double* start_ns, double* end_ns, int32_t cap, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* WITTGPU_H */
