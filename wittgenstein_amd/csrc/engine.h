// Shared host/device definitions of the time-stepped engine that replaces core.Network's
// runMs()/receiveUntil()/nextMessage()/send() loop (C/Network.java:318-338,533-637,369-487).
//
// Data layout in HBM (all struct-of-arrays, one simulated node per index):
//   nodes      x,y (int16), extraLatency, down, partition id, 4 x int64 counters, doneAt
//   buckets    ring of `horizon` per-ms buckets (the reference's MsgsSlot array, :116-132); a bucket is
//              a list of 16-byte envelope records kept in PUSH ORDER inside 1024-record pages drawn
//              from one pool. LIFO drain (:145-161) = reverse iteration.
//   chains     MultipleDestEnvelope (C/Envelope.java:57-155): {from, seed, sendTime, ndest, destOff}
//              + a ring of destination ids sorted by arrival.
//   payload    ring of 64-bit words holding immutable message payloads (Handel bitsets).
//   per-ms scratch: events (expanded bucket), per-node inbox, unordered outbox, ordered outbox.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "jdk_random.h"

namespace wg {

// ---- device pointers in the GLOBAL address space -----------------------------------------------------------------
// Every pointer the kernels follow is loaded from a table in memory (EngineDev, the protocol's State), and a pointer
// loaded from memory is a generic ("flat") pointer to the compiler: it emits flat_load / flat_store, and while a FLAT
// operation is pending the gfx9-family wait-count logic can only wait for ALL outstanding memory operations
// (s_waitcnt vmcnt(0) lgkmcnt(0)) — the first use of any loaded value then also waits for every load issued after it,
// which serialises exactly the loads the node-visit kernels issue ahead of time (the next visit's header, the next
// level's list). GP<T> stores the plain pointer (host code allocates, copies and biases it as before) and hands device
// code a pointer qualified with address space 1: global_load / global_store / global_atomic, s_waitcnt vmcnt(N).
// Device code keeps the qualifier on everything it derives (`const uint64_t WG_G* row = s.TI + ...`), loads and stores
// whole records with gld / gst, and casts to a generic pointer (F) only at the call of an atomic builtin.
#if defined(__HIP_DEVICE_COMPILE__)
#define WG_G __attribute__((address_space(1)))
#else
#define WG_G
#endif
template <class T>
struct GP {
  T* raw;
  GP() = default;
  WG_HD GP(T* p) : raw(p) {}
#if defined(__HIP_DEVICE_COMPILE__)
  // (host functions are parsed in the device pass too: they take the template; the built-in [] and + of device code find
  // only the address-space-1 conversion)
  template <class U, class = std::enable_if_t<std::is_convertible<T*, U*>::value>>
  __host__ operator U*() const { return raw; }
  __device__ operator WG_G T*() const { return (WG_G T*)raw; }
  __host__ T* operator->() const { return raw; }
  __device__ WG_G T* operator->() const { return (WG_G T*)raw; }
#else
  WG_HD operator T*() const { return raw; }
  WG_HD T* operator->() const { return raw; }
#endif
};

constexpr int PAGE_SHIFT = 10;
constexpr int PAGE_RECS = 1 << PAGE_SHIFT;
constexpr int MAX_CUTS = 8;
constexpr int MAX_LEVELS = 24;

// Launch widths of the grid-stride kernels (blocks; every one of them loops, so results never depend
// on these). 2048 blocks x 4 waves = 8 waves per SIMD of a 256-CU MI355X. WG_GRID_DIV exists for
// tests/emu, which builds these kernels for a CPU wave emulator and wants tiny grids.
#ifndef WG_GRID_DIV
#define WG_GRID_DIV 1
#endif
constexpr int GRID_NODE_WAVES = 2048 / WG_GRID_DIV;    // one wavefront per node visit (deliver, cond_a1), per engine
// ... of a batch of R engines: ~4 rounds of the chip's resident waves in total are enough, and every launched
// wavefront that finds no work still costs its launch (131 072 waves per launch at R = 16 otherwise)
inline int grid_node_waves(int R) {
  constexpr int total = 4 * GRID_NODE_WAVES, loDiv = 8;
  int b = total / (R > 0 ? R : 1);
  // (the floor: 16 blocks an engine. It was GRID_NODE_WAVES / 8 = 256 until round 6 — at 512 copies of GSFSignature that
  // launched 131 072 blocks per kernel for a few hundred node visits per engine, and the launch of the empty ones was 10 % of
  // the run: 712 -> 784 M msgs/s, profiles/r24n_*)
  const int lo = GRID_NODE_WAVES / loDiv > 16 ? 16 : (GRID_NODE_WAVES / loDiv > 0 ? GRID_NODE_WAVES / loDiv : 1);
  if (b < lo) b = lo;
  if (b > GRID_NODE_WAVES) b = GRID_NODE_WAVES;
  return b;
}
// The same for the kernels of the ordering / append chain (scans, k_resolve, the multisplit) and the lane-per-item kernels:
// their grid.x is blocks per ENGINE and grid.y the batch. Two needs set it, and the larger wins:
//   * the CHIP wants `total` blocks over the whole batch (about twice its resident blocks: the call sites' values are the
//     measured optima of the round-3/4 sweeps) — total / R per engine, what a small batch is sized by;
//   * an ENGINE needs blocks for ITS work whatever the batch — a ms's events, records, runners, all proportional to its node
//     count: n / perBlock (`perBlock` = nodes of the network per block of this kernel).
// Until round 6 the rule was total / R alone, rounded up to a multiple of the 8 XCDs: at 512 copies of a 4096-node GSFSignature
// that is 8 blocks an engine where 2 - 4 have work, and the launch of the empty ones was 10 % of the run (a block that finds
// nothing still costs ~ 7 ns; 726 -> 805 M msgs/s, profiles/r24n_* .. r24s_*); n / perBlock alone starves a 64-copy batch
// (-8 %, r24r). The rounding is kept only below 8 engines: from 8 on a batch deals its blocks to the XCDs by ENGINE
// (engine_kernels.hip.h wg_place), whatever the blocks per engine are.
inline int grid_per_engine(int base, int R, int total, int n, int perBlock) {
  if (R <= 1) return base;
  int b = (total / WG_GRID_DIV + R - 1) / R;
  const int own = (n + perBlock - 1) / perBlock / WG_GRID_DIV;
  if (b < own) b = own;
  if (b < 2) b = 2;
  if (R < 8) b = (b + 7) / 8 * 8;  // (the plain block mapping: a multiple of the 8 XCDs)
  return b > base ? base : b;
}
constexpr int GRID_DELIVER_SMALL = 512 / WG_GRID_DIV;
constexpr int GRID_LANE_NODES = 128 / WG_GRID_DIV;   // one lane per node visit (k_handel_lane)
constexpr int GRID_RESOLVE = 512 / WG_GRID_DIV;
constexpr int GRID_TILES = 256 / WG_GRID_DIV;
constexpr int GRID_EXPAND_RUNS = 1024 / WG_GRID_DIV;  // x 4 wavefronts: one per long chain run
constexpr int GRID_COND_TAIL = 128 / WG_GRID_DIV;
constexpr int GRID_SHARD_SMALL = 64 / WG_GRID_DIV > 0 ? 64 / WG_GRID_DIV : 1;  // the per-ms glue kernels of a sharded engine: one engine's ~ 10 k items

enum RecKind : uint32_t { K_MSG = 0, K_TASK = 1, K_PERIODIC = 2, K_CHAIN = 3 };

// 16-byte envelope record (SingleDestEnvelope C/Envelope.java:230-234 is {from,to,arrival,message};
// arrival is implied by the bucket).
struct Rec {
  uint32_t w0;  // kind << 28 | from
  uint32_t w1;  // MSG: to | TASK/PERIODIC: node | CHAIN: chain slot
  uint32_t w2;  // MSG: message word | TASK/PERIODIC: task word | CHAIN: curPos
  uint32_t w3;  // MSG: payload ref | TASK: arg | PERIODIC: period | CHAIN: 0
};
WG_HD inline uint32_t rec_kind(const Rec& r) { return r.w0 >> 28; }
WG_HD inline int32_t rec_from(const Rec& r) { return (int32_t)(r.w0 & 0x0FFFFFFFu); }
WG_HD inline Rec make_rec(uint32_t kind, int32_t from, uint32_t w1, uint32_t w2, uint32_t w3) {
  Rec r;
  r.w0 = (kind << 28) | (uint32_t)from;
  r.w1 = w1;
  r.w2 = w2;
  r.w3 = w3;
  return r;
}

struct Chain {  // 32 bytes
  int32_t from;
  int32_t seed;
  int32_t sendTime;
  int32_t ndest;
  uint32_t destOff;   // into the dest ring; explicit arrivals (WithDelay envelopes) follow at destOff+ndest
  uint32_t msg;
  uint32_t payload;
  uint32_t flags;     // bit0: busy, bit1: explicit arrivals (MultipleDestWithDelayEnvelope :157-228), bit2 (CHAIN_LAT): destination words are id | latency << 16
};

enum OutKind : uint32_t { O_SEND = 0, O_MULTI = 1, O_TASK = 2, O_PERIODIC = 3, O_CHAINCONT = 4, O_SENDALL = 5 };

// 32-byte outbox record written by action() code. It lives at outTmp[evOutBase[event] + sub]: every
// event owns a private slice of the outbox sized by the protocol's emission bound for that kind of
// event (assigned by the expand scan), so emitting needs no allocation and `sub` IS the push order
// inside the event.
struct Out {
  uint32_t kindfrom;  // OutKind << 28 | from
  int32_t to;         // SEND: dest | MULTI: ndest | TASK/PERIODIC: node | CHAINCONT: chain slot
  uint32_t a;         // SEND/MULTI: message word | TASK/PERIODIC: task word | CHAINCONT: pos
  uint32_t b;         // SEND/MULTI: payload | TASK: arg | PERIODIC: period
  int32_t t;          // SEND/MULTI: sendTime | TASK/PERIODIC: arrival
  uint32_t destOff;   // MULTI: offset of the (unsorted) dest list in the dest ring
  uint32_t drawsub;   // ordinal of this record's first rd draw among the event's draws
  uint32_t pad;       // flags: OUT_SHUFFLE
};
constexpr uint32_t OUT_SHUFFLE = 1u;  // MULTI: Collections.shuffle(dests, rd) precedes the send (n - 1 draws, then the seed)
constexpr uint32_t OUT_DELAYED = 2u;  // MULTI: send(m, sendTime, from, dests, delaysBetweenMessage) with the delay in pad >> 8:
                                      // a MultipleDestWithDelayEnvelope (C/Envelope.java:157-228), explicit arrivals

// per-event side data written by expand (16 bytes, one store)
struct EvAux {
  int32_t chain;      // chain slot of a chain hop, else -1
  int32_t cpos;       // position inside the chain | last-of-run << 31
  uint32_t outBase;   // first outbox slot of the event
  uint32_t outCap;    // slots owned
};
// what a wave-per-node visit of a list-based inbox starts from: the node, its newest event and that event's record
struct alignas(16) U4 {  // one 16-byte vector load / store per lane
  uint32_t x, y, z, w;
};
constexpr uint32_t VD_DOWN = 1u;  // VisitDesc::flags: the node is down | partition id << 8
struct alignas(16) VisitDesc {
  int32_t node, e0, next0;
  uint32_t flags;
  Rec rec0;
  EvAux aux0;
};
// A node's inbox as ONE 64-byte line (EngineDev::inbox, protocols that ask for it: Engine::wantInbox): the first
// INBOX_SLOTS events of the ms that go to the node, written by expand in whatever order the lanes arrive (the consumer
// sorts the <= 4 by event index) — a node visit reads its events with one coalesced load instead of chasing
// head[to] -> evNext[e] -> ev[e] through three dependent scattered reads per event (the reference applies a ms's
// envelopes to `to` one after the other, C/Network.java:603-626: this is the per-destination grouping of that loop).
// Events beyond the line's four go onto the node's overflow list (head / evNext, as for protocols without lines).
constexpr int INBOX_SLOTS = 4;
constexpr uint32_t INBOX_CHAIN = 1u << 31;  // InboxEntry::w0: the event is a hop of a multi-destination envelope (its EvAux matters)
struct alignas(16) InboxEntry {
  uint32_t e;    // event index in the global order of the ms
  uint32_t w0;   // Rec::w0 (kind << 28 | from) | INBOX_CHAIN
  uint32_t w2;   // Rec::w2: message word / task word
  uint32_t w3;   // Rec::w3: payload ref / task argument / period
};
// per-event result written by deliver (8 bytes)
struct EvRes {
  uint32_t nrec;      // records emitted | EV_DELIVERED / EV_TASK_RUN flags | level << 24
  uint32_t ndraw;     // rd draws consumed
};
constexpr uint32_t EV_NREC_MASK = 0xFFFFu;
constexpr uint32_t EV_DELIVERED = 1u << 16;  // counted in msgReceived (C/Network.java:607-613)
constexpr uint32_t EV_TASK_RUN = 1u << 17;
// a periodic task that wrote a payload snapshot other shards will read (sharded engines: Handel's dissemination, GSF's
// doCycle): code k in bits 18..21 = a row of 2^(k-1) 64-bit words (0: none — e.g. a Handel node whose open levels are all
// complete sends the constant all-ones block)
constexpr int EV_SNAP_SHIFT = 18;
constexpr uint32_t EV_SNAP_MASK = 0xFu << EV_SNAP_SHIFT;
__host__ __device__ inline uint32_t ev_snap_code(uint32_t words) {  // smallest k with 2^(k-1) >= words (words >= 1)
  uint32_t k = 1;
  while ((1u << (k - 1)) < words) k++;
  return k;
}
// exchange 1 of a sharded engine ships ONE int32 per event (the owner's; zeros elsewhere, summed): records 10 bits | delivered
// | task run | level 5 | snapshot code 4 | draws 11 — an event beyond those ranges stops the run loudly (ERR_OUTBOX)
__host__ __device__ inline bool evres_packable(const EvRes& r) { return (r.nrec & EV_NREC_MASK) < 1024u && r.ndraw < 2048u; }
__host__ __device__ inline uint32_t evres_pack(const EvRes& r) {
  return (r.nrec & 1023u) | ((r.nrec & EV_DELIVERED) ? 1u << 10 : 0u) | ((r.nrec & EV_TASK_RUN) ? 1u << 11 : 0u) |
         (((r.nrec >> 24) & 31u) << 12) | (((r.nrec & EV_SNAP_MASK) >> EV_SNAP_SHIFT) << 17) | (r.ndraw << 21);
}
__host__ __device__ inline EvRes evres_unpack(uint32_t w) {
  EvRes r;
  r.nrec = (w & 1023u) | ((w >> 10) & 1u ? EV_DELIVERED : 0u) | ((w >> 11) & 1u ? EV_TASK_RUN : 0u) | (((w >> 12) & 31u) << 24) |
           (((w >> 17) & 15u) << EV_SNAP_SHIFT);
  r.ndraw = w >> 21;
  return r;
}

enum LatKind : int32_t { LAT_BYDIST = 0, LAT_FIXED = 1, LAT_UNIFORM = 2, LAT_NONE = 3, LAT_MEASURED = 4, LAT_IC3 = 5, LAT_ETHSCAN = 6, LAT_CITY = 7 };

enum ErrBits : uint32_t {
  ERR_BUCKET_POOL = 1u << 0,   // out of bucket pages
  ERR_BUCKET_PAGES = 1u << 1,  // one bucket exceeded max pages per bucket
  ERR_OUTBOX = 1u << 2,
  ERR_HORIZON = 1u << 3,       // arrival - time >= horizon
  ERR_SAME_MS = 1u << 4,       // device action pushed into the bucket being drained
  ERR_CHAIN_SLOTS = 1u << 5,
  ERR_CHAIN_DESTS = 1u << 6,
  ERR_PAYLOAD = 1u << 7,
  ERR_QUEUE_CAP = 1u << 8,     // Handel toVerifyAgg capacity
  ERR_MULTI_TOO_BIG = 1u << 9, // device multi-dest send with > 64 destinations
  ERR_PENDING = 1u << 10,      // Handel pending-verification table full
  ERR_PROTOCOL = 1u << 11,     // a reference IllegalStateException site inside action()
  ERR_EVENTS = 1u << 12,       // events in one ms exceed scratch capacity
  ERR_ARRIVAL_PAST = 1u << 13,
  ERR_SHARD_MULTI = 1u << 14,  // sharded engine: an action() emitted a multi-destination envelope
  ERR_SAME_MS_BLOCKS = 1u << 15,  // Casper resident: two blocks created in one simulated ms (block-id order across wavefronts)
  ERR_RANK_BUMPS = 1u << 16,      // Handel, ranks carried by the senders: a node's table of bumped senders is full
  ERR_SHARD_EVENT = 1u << 17      // sharded engine: an event's (records, draws) do not fit the packed exchange word (evres_packable)
};

// Device-resident engine globals (one instance).
struct Globals {
  uint64_t rng;            // rd's 48-bit state (C/Network.java:32)
  uint32_t epoch;          // nextMessage() call counter
  uint32_t err;
  int32_t now;             // the ms being processed (Network.time during the drain; the time++ target
                           // during a conditional-task phase). Advanced by k_end_phase after a drain.
  int32_t until;           // receiveUntil's bound for the run in progress
  // cumulative statistics
  unsigned long long delivered, tasks, events, draws, payloadBytes;
  unsigned long long deliveredByLevel[32];
  uint32_t anyEvent;       // receiveUntil's didSomething
  // per-ms scratch counters
  uint32_t nEvents;        // events in the bucket being drained (after chain-run expansion)
  uint32_t nActive;        // nodes with >= 1 event
  uint32_t nActiveB;       // items in activeB
  uint32_t outSlots;       // outbox slots handed out to the events of this ms
  uint32_t nOut;           // ordered outbox length
  uint32_t nDraws;         // draws in this phase
  uint32_t rejectSeen;     // a nextInt(bound) rejection happened in this phase
  // allocators
  uint32_t freeTop;        // free bucket pages
  uint32_t chainHead;      // monotone
  unsigned long long destHead;     // monotone (ring index = destHead % chainDests)
  unsigned long long payloadHead;  // monotone, in 64-bit words
  // in-kernel cycle counters of investigation builds (-DWG_KPROF, tools/kprof.sh); untouched otherwise
  unsigned long long kprof[32];
  GP<unsigned long long> kprofBuf;  // [KPROF_WAVES][32] per-wavefront rows the marks add to (summed into kprof[] by the host)
  // sharded engines only (0 otherwise): multi-destination envelopes created in this phase / their destinations
  // (replicated), and this shard's private scratch-ring head for the unsorted destination lists of its action()s
  uint32_t nMulti, nMultiDests;
  unsigned long long localDestHead;
  uint32_t nSendAll;       // Network.sendAll calls made by action()s in this phase (k_sendall_*), reset by k_end_phase
  uint32_t nFar;           // records parked in EngineDev::farBuf since the host last collected them
  uint32_t nRuns;          // long chain runs of this ms left to k_expand_runs (EngineDev::runs), reset by k_end_phase
  uint32_t notes;          // sticky, non-fatal remarks of the resident protocol (NOTE_*)
  uint32_t nScatter;       // nOut as k_col_reserve_end saw it: what k_scatter appends after that kernel has reset nOut
  uint32_t nOutKeep;       // the drain's ordered outbox, kept UNFILED over the conditional-task phase of the edge: that phase's
                           // records follow it in fin / arr (from this index on) and ONE append files both, in push order
  uint32_t nSnapEv;        // sharded engines: events of this ms that wrote a payload snapshot (EV_SNAP_*), counted by the order scan
};
constexpr uint32_t KPROF_WAVES = 16384;
constexpr uint32_t NOTE_RANKS_SATURATED = 1u;  // Handel: a receptionRanks entry hit Integer.MAX_VALUE (P/Handel.java:826-828)

struct LatencyModel {
  int32_t kind;
  int32_t param;            // FIXED: latency, UNIFORM: maxLatency
  GP<const uint8_t> lutDist;   // BYDIST: [1145][100]
  GP<const int32_t> tabDelta;  // MEASURED/ETHSCAN/UNIFORM: [100]
  GP<const int32_t> tabDist;   // IC3: [1145]
  // LAT_CITY — the city-based models (C/NetworkLatency.java:86-233); param = wg_city_latency_mode, C = nCities:
  GP<const uint16_t> city;     // [n] the node's city (Node.cityName as an index into the caller's city list)
  GP<const int32_t> cityTab;   // [C][C] AWS: ping / 2; BY_CITY: max(1, round(0.5f * ping)), the whole answer
  GP<const float> cityPing;    // [C][C] BY_CITY_WJITTER: the measured round trip (float, as CSVLatencyReader parsed it)
  GP<const double> cityJit;    // [100]  AWS / BY_CITY_WJITTER: gpd.inverseF(delta / 100.0)
  int32_t nCities;
};

// what a send needs of its two ends — position, extra latency, stopped, partition — in ONE 16-byte record per node (a
// copy of the arrays below, rebuilt by the host whenever one of them changes): five look-ups per destination were what
// k_resolve's 8 M records cost in the ms in which every node sends
struct NodeGeo {
  int16_t x, y;
  int32_t extraLatency;
  uint8_t down, part;
  uint8_t pad[6];
};
struct NodeArrays {
  int32_t n;
  GP<NodeGeo> geo;
  GP<int16_t> x;
  GP<int16_t> y;
  GP<int32_t> extraLatency;
  GP<uint8_t> down;
  GP<uint8_t> part;            // partitionId (C/Network.java:639-649), recomputed on partition()
  GP<long long> msgReceived;
  GP<long long> msgSent;
  GP<long long> bytesSent;
  GP<long long> bytesReceived;
  GP<long long> doneAt;
};

struct SendAllDesc;
struct FarRec;
struct RunDesc;
// Everything a kernel needs, passed by value.
struct EngineDev {
  uint32_t hostMode;        // wg_next_delivery mode: events go to the host, no device inbox lists are built
  uint32_t halted;          // batch member that is not advanced by the current run (RunMultipleTimes: its
                            // continuation predicate turned false); every kernel returns at once for it
  GP<Globals> g;
  NodeArrays nodes;
  LatencyModel lat;
  int32_t discardTime;
  uint32_t nparts;          // number of partition cuts (C/Network.java:639-649); 0 = the lookup is skipped
  // buckets
  int32_t horizon;          // D (power of two)
  int32_t maxPagesPerBucket;
  uint32_t nPages;
  GP<Rec> pool;
  GP<uint32_t> freeStack;
  GP<uint32_t> pagetab;        // [D][maxPagesPerBucket]
  GP<uint32_t> bcnt;           // [D]
  // chains
  GP<Chain> chains;
  uint32_t chainSlots;
  GP<int32_t> dests;
  unsigned long long chainDests;
  // payload ring
  GP<uint64_t> payload;
  unsigned long long payloadWords;
  // ring-safety bookkeeping: allocator heads at the end of each of the last `horizon` ms. Everything
  // allocated at ms s is dead by s + horizon (arrival - time < horizon is enforced), so a ring is
  // safe iff head(t) - head(t - horizon) <= capacity.
  GP<unsigned long long> destHeadAt;     // [D]
  GP<unsigned long long> payloadHeadAt;  // [D]
  // per-ms scratch
  uint32_t maxEvents;
  GP<Rec> ev;                  // expanded events (MSG/TASK/PERIODIC form) in global event order
  GP<EvAux> evAux;
  GP<EvRes> evRes;
  GP<uint32_t> evRecOff;       // exclusive scans of nrec / ndraw in event order
  GP<uint32_t> evDrawOff;
  GP<int32_t> evNext;          // per-node inbox as a linked list through the events
  GP<int32_t> head;            // [n] newest event of the node this ms, -1 = none
  GP<InboxEntry> inbox;        // [n][INBOX_SLOTS] the node's first events of this ms (NULL: lists only)
  GP<uint32_t> icnt;           // [n] events of the node this ms (inbox lines only); reset by the delivery pass
  GP<uint32_t> active;         // nodes with >= 1 event (unordered)
  GP<VisitDesc> activeB;       // work list of a protocol's second delivery kernel (Handel: the items k_handel_lane leaves to k_handel_wave, 16 bytes each)
  uint32_t maxOut;
  GP<Out> outTmp;              // per-event slices (see Out)
  GP<uint32_t> recEv;          // event of each ordered outbox position
  GP<Rec> fin;                 // ordered outbox
  GP<int32_t> arr;             // arrival per ordered record, -1 = dropped at send time
  // protocol emission bounds used by expand to size the per-event outbox slices
  uint32_t boundMsg;        // max records a delivered message's action() can emit
  uint32_t boundTask[4];    // ... a task's action(), by task word (words >= 3 use [3])
  // multisplit scratch
  GP<uint32_t> tileHist;       // [maxTiles][D]
  GP<uint32_t> binBase;        // [D] position of this phase's first record inside each bucket
  // scan scratch
  GP<unsigned long long> scanPartials;
  // node-range sharding of ONE simulation over several engines (wg_shard_configure): the scheduler state above
  // is replicated on every shard and evolves identically; node / protocol state is touched only for the nodes
  // of [shardLo, shardHi). Not sharded: sharded = 0, range = everything.
  uint32_t sharded;
  int32_t shardLo, shardHi;
  GP<int32_t> xbuf;            // [maxOut][5] exchange image of the ordered outbox: Rec words + (arrival + 1); preceded by
                            // XB_HEAD header words (xbuf[-XB_HEAD] = multi-destination envelopes among the records),
                            // which travel in the same all-reduce
  GP<int32_t> xmulti;          // [maxMulti][XM_WORDS] exchange image of the multi-destination envelopes of a phase
  GP<int32_t> xev;             // [maxEvents] exchange image of the events' results: one packed word each (evres_pack)
  uint32_t maxMulti;
  GP<uint32_t> multiK;         // [maxOut] ordinal of a fresh multi-destination record / offset of its destinations
  GP<uint32_t> multiOff;
  // where action() code parks the (unsorted) destination list of a multi-destination send until `resolve`:
  // the envelope ring itself, or — sharded, where that ring is replicated state — a private scratch ring
  GP<int32_t> sdests;
  GP<int32_t> arvTmp;  // [sdestCap] arrivals of the list being sorted at the same entries of sdests (resolve_multi)
  unsigned long long sdestCap;
  // Network.sendAll issued by an action() (O_SENDALL): every node is a destination, so the envelope is resolved by
  // k_sendall_* after `resolve` (one descriptor per call, latency scratch and tile histograms per descriptor);
  // maxSendAll == 0: the resident protocol never calls it
  GP<SendAllDesc> saDesc;
  GP<int32_t> saLat;           // [maxSendAll][n]
  GP<uint32_t> saHist;         // [maxSendAll][tiles(n)][D]
  uint32_t maxSendAll;
  uint32_t saBins;          // latency bins of saHist (<= horizon): max latency of the model + 1, rounded up to 64
  GP<FarRec> farBuf;    // NULL: arrivals beyond the ring are an error
  uint32_t farCap;
  // chain runs (consecutive hops of one multi-destination envelope arriving in the same ms) of >= runMin hops are
  // not unrolled by the lane that scans their bucket record but listed here and unrolled one wavefront per run by
  // k_expand_runs (a sendAll to N nodes has runs of ~N/300 hops: Casper). runMin == 0: every run is unrolled in place
  GP<RunDesc> runs;
  uint32_t maxRuns;
  uint32_t runMin;
  // message word + 1 of the K_MSG events the resident protocol delivers one lane per event, without inbox lists
  // (ExpandF::lane_only); 0: every event is threaded onto its node's list
  uint32_t laneMsgPlus1;
  // A resident protocol whose node ids fit 16 bits may carry a per-destination TAG in the upper half of the destination
  // words of its multi-destination envelopes (Handel: the receiver's initial reception rank of the sender, which the
  // SENDER knows from its emission list — P/Handel.java:991-1013 — so that the receiver need not look it up in an N x N
  // matrix at delivery). destTagged != 0: a destination word is id | tag << 16 wherever one is read (dest_id), and the
  // message word a hop is delivered with is the envelope's | tag << destTagMsgShift (dest_msg). The words travel through
  // the sort by arrival, the envelope ring and a shard's exchange image as they are.
  uint32_t destTagged;
  uint32_t destTagMsgShift;
  // batch launches deal their blocks so that an engine stays on ONE XCD (engine_kernels.hip.h: wg_place); read from the
  // FIRST engine of the table. 0: the plain (blockIdx.x, blockIdx.y) mapping (WG_XCD_PLACE=0, the A/B switch)
  uint32_t xcdPlace;
};
WG_HD inline int32_t dest_id(const EngineDev& d, int32_t w) { return d.destTagged ? (int32_t)((uint32_t)w & 0xFFFFu) : w; }
WG_HD inline uint32_t dest_msg(const EngineDev& d, uint32_t msg, int32_t w) {
  return d.destTagged ? msg | (((uint32_t)w >> 16) << d.destTagMsgShift) : msg;
}
struct RunDesc {  // 32 bytes
  uint32_t chain, pos;      // envelope slot, first hop of the run
  uint32_t e, ob;           // its first event index / first outbox slot (the expand scan's exclusive prefix)
  uint32_t len, pad0, pad1, pad2;
};
// An envelope an action() registered for a time beyond the bucket ring (a task seconds ahead: Casper's 8 s slots): parked
// here, collected by the host every `horizon` ms of simulated time and staged like a host-side registration — it is
// injected before any device push can reach its bucket, so the bucket's push order is what the reference's would be
// provided arrival - time >= 2 * horizon (else ERR_HORIZON, as without the buffer).
struct FarRec {  // 32 bytes
  int32_t ms;              // Network.time when it was pushed
  uint32_t p;              // its position in that ms's ordered outbox
  Rec rec;
  int32_t arrival;
  uint32_t pad;
};
struct SendAllDesc {  // 32 bytes
  uint32_t p;               // position in the ordered outbox
  int32_t from, seed, sendTime;
  uint32_t slot, msg, payload, pad;
  unsigned long long destOff;
};
constexpr int XB_HEAD = 4;                    // header words in front of EngineDev::xbuf (16-byte alignment kept)
constexpr int XM_WORDS = 6 + 128;              // seed, sendTime, msg, payload, ndest, explicit arrivals?, dest[64], arrival[64]
constexpr uint32_t MULTI_FRESH = 0xFFFFFFFFu;  // Rec::w3 of a K_CHAIN record whose envelope is yet to be created
constexpr uint32_t SENDALL_FRESH = 0xFFFFFFFEu;  // ... of a Network.sendAll an action() made on a sharded engine (w1 = node count)
WG_HD inline bool shard_owns(const EngineDev& d, int32_t node) { return node >= d.shardLo && node < d.shardHi; }

WG_HD inline int32_t isqrt_floor(int32_t v) {  // (int) Math.sqrt(v), C/Node.java:281
  int32_t r = (int32_t)
#if defined(__HIP_DEVICE_COMPILE__)
      __fsqrt_rn((float)v);
#else
      __builtin_sqrtf((float)v);
#endif
  while (r * r > v) r--;
  while ((r + 1) * (r + 1) <= v) r++;
  return r;
}

WG_HD inline int32_t node_dist(int32_t x1, int32_t y1, int32_t x2, int32_t y2) {  // C/Node.java:278-282
  int32_t ax = x1 > x2 ? x1 - x2 : x2 - x1;
  int32_t ay = y1 > y2 ? y1 - y2 : y2 - y1;
  int32_t dx = ax < 2000 - ax ? ax : 2000 - ax;
  int32_t dy = ay < 1112 - ay ? ay : 1112 - ay;
  return isqrt_floor(dx * dx + dy * dy);
}

// NetworkLatency.getLatency (C/NetworkLatency.java:27-34) over the table-ised models.
WG_HD inline int32_t latency_of(const LatencyModel& m, int32_t from, int32_t to, int32_t x1, int32_t y1, int32_t e1,
                                int32_t x2, int32_t y2, int32_t e2, int32_t delta) {
  if (from == to) return 1;
  int32_t base = e1 + e2;
  int32_t ext;
  switch (m.kind) {
    case LAT_BYDIST: ext = m.lutDist[node_dist(x1, y1, x2, y2) * 100 + delta]; break;
    case LAT_FIXED: ext = m.param; break;
    case LAT_NONE: ext = 1; break;
    case LAT_UNIFORM:
    case LAT_MEASURED: ext = m.tabDelta[delta]; break;
    case LAT_IC3: ext = m.tabDist[node_dist(x1, y1, x2, y2)]; break;
    case LAT_CITY: {
      const int32_t c1 = m.city[from], c2 = m.city[to];
      if (m.param == 0) {  // AwsRegionNetworkLatency.getLatency :135-145: same datacenter 1, else ping / 2 + (int) jitter
        ext = c1 == c2 ? 1 : m.cityTab[c1 * m.nCities + c2] + (int32_t)m.cityJit[delta];
        if (ext < 1) ext = 1;
      } else if (m.param == 1) {  // NetworkLatencyByCity :168-185 (the table holds max(1, Math.round(0.5f * ping)))
        ext = m.cityTab[c1 * m.nCities + c2];
      } else {  // NetworkLatencyByCityWJitter :212-232: max(1, (int) Math.round(0.5 * (jitter + (same city ? 10 : ping))))
        const double raw = m.cityJit[delta] + (c1 == c2 ? 10.0 : (double)m.cityPing[c1 * m.nCities + c2]);
        const double h = 0.5 * raw + 0.5;  // Math.round(double) = floor(x + 1/2)
        long long r = (long long)h;
        if ((double)r > h) r--;
        ext = r < 1 ? 1 : (int32_t)r;
      }
      break;
    }
    default: {  // LAT_ETHSCAN delegates to an inner MeasuredNetworkLatency.getLatency (:366-384)
      int32_t inner = base + m.tabDelta[delta];
      ext = inner > 1 ? inner : 1;
    }
  }
  base += ext;
  return base > 1 ? base : 1;
}

}  // namespace wg
