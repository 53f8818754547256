// Handel (P/Handel.java) as a resident device protocol. One wavefront per simulated node: scalars
// are wave-uniform, lane k owns the 64-bit words w with (w & 63) == k of every bitset row of the
// node, so bitset algebra (or/and/cardinality/intersects) is a coalesced pass + a wave reduction.
//
// State layout (struct-of-arrays, HBM):
//   bit rows  TI,LA,VI,TV,FP : [N][W] uint64, W = N/64. Bit j = node id j. HLevel l's bitsets
//             (totalIncoming, lastAggVerified, verifiedIndSignatures, toVerifyInd, finishedPeers
//             :373-394) only ever hold ids of the level's aligned sibling block of 2^(l-1) ids
//             (allSigsAtLevel :671-684), and the blocks of different levels are disjoint, so one row
//             per kind holds all levels. totalOutgoing of level l is always the union of
//             totalIncoming of levels < l (:728-731) = the node's OWN aligned block in the TI row.
//   ranks     [N][N] int32  receptionRanks (:285)        peers [N][N-1] int32 emission lists (:510-522)
//   queues    toVerifyAgg (:385): per (node, level) up to Q slots {from, rank, sig[2^(l-1) bits]} in a
//             private slab + an order list; a slot stays allocated while a registered
//             updateVerifiedSignatures task still references it (:833-836).
// Honest-node paths only: byzantineSuicide / hiddenByzantine (:538-559, :840-917) are not resident.
#pragma once
#include "engine_kernels.hip.h"

namespace wg {

constexpr int H_PEND = 4;          // outstanding updateVerifiedSignatures tasks per node
constexpr uint32_t H_TASK_DISSEMINATION = 0;
constexpr uint32_t H_TASK_UPDATE = 1;

struct HandelState {
  wg_handel_params p;
  int32_t N, L, W, Q;
  GP<uint64_t> TI, LA, VI, TV, FP;   // [N][W]
  GP<int32_t> ranks;                      // [N][N]
  // emission lists [N][N-1] (:510-522), never written after init(): 16-bit ids when N <= 65536 (half the bytes of the
  // second-largest array of a copy — more resident copies per GPU), 32-bit otherwise; read through h_peer()
  GP<const uint16_t> peers16;
  GP<const int32_t> peers32;
  // toVerifyAgg slots per (node, level): Q for the levels whose block is below 16 words, Qw (wg_config.queue_cap_wide)
  // for the wide ones — their queues stay short (a few entries: one sender per period and level) while a slot is up to
  // 2 KB, so a flat capacity spent gigabytes on slots that are never used
  int32_t Qw;
  // Node header: every scalar of a node and its per-level scalars in ONE record of hdrStride 32-bit words
  // (array of structs). A node visit is one wavefront touching one node, so the record is read and written
  // back as a few consecutive cache lines of one page, instead of twenty 64-byte lines in twenty arrays:
  //   [HH_ADDED .. HH_CTEPOCH]  addedCycle, sigQueueSize, msgFiltered, startAt, nodePairingTime, currWindowSize,
  //                              sigsChecked, ConditionalTask.minStartTime, the epoch it last left nextMessage()'s copy
  //   [HH_DONE_LO, HH_DONE_HI]   Node.doneAt, mirrored from NodeArrays::doneAt (written through when it changes)
  //   [HH_PEND +4] [HH_PENDFROM +4]  outstanding updateVerifiedSignatures tasks: valid<<31 | level<<8 | slot ; from
  //   [HH_CAND +12]              checkSigs' candidates of this edge, in level order: 16 bits each, level << 8 | slot (written by
  //                              k_handel_cond_a1 with the rest of the record, read by k_handel_cond_a2 in the same phase)
  //   [HH_QMASK]                 bit l: level l's verification queue is not empty (what k_handel_cond_pre looks at)
  //   [HH_LV + l*8 + plane]      level-major: the eight scalars of HLevel l side by side (32 bytes, two levels a 64-byte
  //                              line) — posInLevel, |totalIncoming|, |lastAggVerified|, |verifiedInd|, queue length,
  //                              outgoingFinished, queue slots in use (low / high word); LS = 16 or 32 >= L levels.
  //                              An event works on ONE level: the record goes back as the few 16-byte pieces that
  //                              changed (store_levels), i.e. the scalars' line and the level's, not all ten lines
  GP<uint32_t> hdr;
  int32_t LS, lsShift, hdrStride;
  // ConditionalTask.minStartTime and the epoch in which the task last left nextMessage()'s copy, two words a node, dense:
  // k_handel_cond_pre looks at every node every ms — 8 bytes of a coalesced stream instead of a 64-byte line of the record
  // per node; only the nodes whose task is due touch their record (HH_CTMIN / HH_CTEPOCH of the record are unused)
  GP<uint32_t> ct;
  GP<uint64_t> qent;                      // [N][L][64] list entries in list order: rank << 32 | slot
  GP<int32_t> qfrom;                      // [N][L][Q]
  GP<uint64_t> qsig;                      // per level l: [N][Q][nw(l)] at qsigOff[l]
  unsigned long long qsigOff[MAX_LEVELS];
  // dissemination snapshots (SendSigs.sigs = totalOutgoing.clone(), :254): a node disseminates exactly once
  // per aligned window of `period` ms, so its snapshot lives at a computed address — no allocation:
  //   snap[((t / period) % snapNb) * N + node][0 .. snapStride)   the own block of the highest open level;
  //   the lower levels' blocks are sub-ranges of it (see dissemination)
  GP<uint64_t> snap;
  uint32_t snapNb, snapStride;
  // conditional-task phase scratch
  GP<uint32_t> runList;                   // [N] nodes whose checkSigs runs at this edge (unordered)
  GP<uint32_t> runCount;                  // [1]
  GP<uint8_t> candCnt;                    // [N] number of levels with a candidate
  GP<uint32_t> condOrd;                   // [N] ordinal among drawing nodes
  GP<uint32_t> condList;                  // drawing nodes in id order
  GP<int32_t> drawVal;                    // [N]
  // node-range sharding (wg_shard_configure): this engine holds the per-node rows above only for the nodes
  // [lo, hi) (the row pointers are biased so that they are still indexed by node id); 0 / N when not sharded
  int32_t lo, hi;
  GP<const uint64_t> ones;                // [W] all-ones words: the payload of a sharded fast-path send (see snapshot_outgoing)
  GP<uint32_t> snapIdx;                   // [maxEvents] row of a dissemination event in the snapshot exchange image
  GP<uint32_t> nSnap;                     // [1]
  GP<int32_t> xsnap;                      // [xsnapRows][snapStride * 2] exchange image of this ms's dissemination snapshots
  uint32_t xsnapRows;
};

enum HandelHdr : int { HH_ADDED = 0, HH_SIGQ = 1, HH_FILT = 2, HH_START = 3, HH_PAIR = 4, HH_WINDOW = 5, HH_SIGCHK = 6,
                       HH_CTMIN = 7, HH_CTEPOCH = 8, HH_DONE_LO = 9, HH_DONE_HI = 10, HH_QMASK = 11, HH_PEND = 12,
                       HH_PENDFROM = 16, HH_CAND = 20, HH_LV = 32 };
enum HandelPlane : int { HP_POS = 0, HP_CTI, HP_CLA, HP_CVI, HP_QLEN, HP_OUTFIN, HP_QUSED_LO, HP_QUSED_HI, HP_COUNT };
__device__ __forceinline__ uint32_t WG_G* h_hdr(const HandelState& s, int32_t node) { return s.hdr + (size_t)node * s.hdrStride; }
__device__ __forceinline__ uint32_t WG_G* h_lv(const HandelState& s, int32_t node, int plane, int l) {
  return s.hdr + (size_t)node * s.hdrStride + HH_LV + l * HP_COUNT + plane;
}

// geometry of one level inside a row
struct Lv {
  int32_t size;   // ids in the block = expectedSigs()
  int32_t bw;     // first 64-bit word
  int32_t nw;     // number of words
  uint64_t mask;  // bits of the block inside the word when nw == 1 and size < 64, else ~0
};
__device__ __forceinline__ Lv block_view(int32_t firstId, int32_t size) {
  Lv v;
  v.size = size;
  v.bw = firstId >> 6;
  if (size >= 64) {
    v.nw = size >> 6;
    v.mask = ~0ULL;
  } else {
    v.nw = 1;
    v.mask = ((1ULL << size) - 1ULL) << (firstId & 63);
  }
  return v;
}
// the sibling block: ids whose signatures level l waits for (waitedSigs :424-432)
__device__ __forceinline__ Lv sib_view(int32_t node, int l) {
  if (l == 0) return block_view(node, 1);
  return block_view(((node >> (l - 1)) ^ 1) << (l - 1), 1 << (l - 1));
}
// the node's own block: ids that totalOutgoing of level l can hold
__device__ __forceinline__ Lv own_view(int32_t node, int l) { return block_view((node >> (l - 1)) << (l - 1), 1 << (l - 1)); }

__device__ __forceinline__ int32_t h_peer(const HandelState& s, size_t idx) {
  return s.peers16 ? (int32_t)s.peers16[idx] : s.peers32[idx];
}
__device__ __forceinline__ int h_nw(int l) { return l == 0 ? 1 : ((1 << (l - 1)) >= 64 ? (1 << (l - 1)) >> 6 : 1); }
__device__ __forceinline__ int h_qcap(const HandelState& s, int l) { return h_nw(l) >= 16 ? s.Qw : s.Q; }
__device__ __forceinline__ int h_msg_size(int l) { return 1 + ((l == 0 ? 1 : (1 << (l - 1))) / 8) + 96 * 2; }  // :256-260

// word j of a view is owned by lane (bw + j) & 63
#define H_FOR_WORDS(v, j)                                                                      \
  for (int j = (int)((WG_LANE - (v).bw) & 63); j < (v).nw; j += 64)

__device__ __forceinline__ uint64_t ld_coherent(const uint64_t WG_G* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool row_get(const uint64_t WG_G* row, int32_t id) { return (ld_coherent(row + (id >> 6)) >> (id & 63)) & 1ULL; }
__device__ __forceinline__ void row_set(uint64_t WG_G* row, int32_t id, bool v) {
  int w = id >> 6;
  if ((int)WG_LANE == (w & 63)) {
    uint64_t x = row[w];
    row[w] = v ? (x | (1ULL << (id & 63))) : (x & ~(1ULL << (id & 63)));
  }
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) { return wave_reduce_add64(v); }

// per-wave LDS mirror of the (node, level) scalars
struct LevelScalars {  // LDS image of a node header: the planes HP_POS..HP_QUSED_HI (32 words each), then the scalars
  int32_t pos[32];
  int32_t cTI[32];
  int32_t cLA[32];
  int32_t cVI[32];
  int32_t qlen[32];
  int32_t outFin[32];
  uint32_t quLo[32];
  uint32_t quHi[32];
  uint32_t sc[HH_LV];
  U4 orig[(HH_LV + 8 * 32) / 4];  // the record as it was loaded, 16-byte pieces: store_levels writes back what differs
};
__device__ __forceinline__ unsigned long long ls_qused(const LevelScalars* ls, int l) {
  return (unsigned long long)ls->quLo[l] | ((unsigned long long)ls->quHi[l] << 32);
}
__device__ __forceinline__ void ls_set_qused(LevelScalars* ls, int l, unsigned long long v) {
  ls->quLo[l] = (uint32_t)v;
  ls->quHi[l] = (uint32_t)(v >> 32);
}

constexpr uint32_t H_REF_RING = 0x80000000u;  // payload ref flag: engine payload ring (fast-path sends)
constexpr uint32_t H_REF_ONES = 0xFFFFFFFFu;  // payload ref: the all-ones block (fast-path sends of a sharded engine)
__device__ __forceinline__ const uint64_t WG_G* h_payload(const EngineDev& d, const HandelState& s, uint32_t payload) {
  if (payload & H_REF_RING) return payload == H_REF_ONES ? s.ones : d.payload + (payload & ~H_REF_RING);
  return s.snap + payload;
}

struct HandelProto {
  typedef HandelState State;
  typedef LevelScalars WaveShared;

  // node-scoped registers (wave-uniform) live in this struct for the duration of a node's events
  struct NodeRegs {
    long long doneAt, doneAt0;
    int32_t addedCycle, sigQueueSize, msgFiltered, startAt;
    LevelScalars* ls;  // (the outstanding updateVerifiedSignatures tasks stay in the LDS image: sc[HH_PEND ..], sc[HH_PENDFROM ..])
  };

  __device__ static int msg_size(const State&, uint32_t msg) { return h_msg_size((int)(msg & 31u)); }
  __device__ static int msg_level(uint32_t msg) { return (int)(msg & 31u); }

  __device__ static void node_begin(Ctx& c, const State& s, NodeRegs& r, LevelScalars* ls) {
    load_levels(s, c.node, ls);  // the whole header, one memory instruction
    regs_from_image(r, ls);
  }
  // the header fetched ahead of the visit (k_deliver's pipelined loop): 16 bytes a lane, two rounds when L > 16
  struct Pre {
    U4 q0;
  };
  __device__ static Pre prefetch(const State& s, int32_t node) {
    const U4 WG_G* g = (const U4 WG_G*)h_hdr(s, node);
    const int n4 = s.hdrStride >> 2;
    Pre p;
    p.q0 = g[(int)WG_LANE < n4 ? (int)WG_LANE : 0];
    return p;
  }
  // piece i of the record (16 bytes) <-> the LDS image: scalars as they are; a level's two pieces are the planes
  // 0..3 / 4..7 of that level (the image keeps one array per plane: ls->cTI[l], ...)
  __device__ static void scatter_levels(const State&, LevelScalars* ls, int i, const U4 q) {
    const int w = i << 2;
    ls->orig[i] = q;
    if (w < HH_LV) {
      uint32_t* dst = ls->sc + w;
      dst[0] = q.x;
      dst[1] = q.y;
      dst[2] = q.z;
      dst[3] = q.w;
    } else {
      const int r = w - HH_LV;
      uint32_t* dst = (uint32_t*)ls + (((r >> 2) & 1) << 7) + (r >> 3);  // plane (0 or 4) * 32 + level
      dst[0] = q.x;
      dst[32] = q.y;
      dst[64] = q.z;
      dst[96] = q.w;
    }
  }
  __device__ static U4 gather_levels(const LevelScalars* ls, int i) {
    const int w = i << 2;
    U4 q;
    if (w < HH_LV) {
      const uint32_t* src = ls->sc + w;
      q.x = src[0];
      q.y = src[1];
      q.z = src[2];
      q.w = src[3];
    } else {
      const int r = w - HH_LV;
      const uint32_t* src = (const uint32_t*)ls + (((r >> 2) & 1) << 7) + (r >> 3);
      q.x = src[0];
      q.y = src[32];
      q.z = src[64];
      q.w = src[96];
    }
    return q;
  }
  __device__ static void node_begin_pre(Ctx& c, const State& s, NodeRegs& r, LevelScalars* ls, const Pre& p) {
    const int n4 = s.hdrStride >> 2;
    __builtin_amdgcn_wave_barrier();  // the previous visit's store_levels has read the image
    if ((int)WG_LANE < n4) scatter_levels(s, ls, (int)WG_LANE, p.q0);
    if ((int)WG_LANE + 64 < n4)  // (L > 16: the record's tail is fetched here, not ahead)
      scatter_levels(s, ls, (int)WG_LANE + 64, ((const U4*)h_hdr(s, c.node))[(int)WG_LANE + 64]);
    __builtin_amdgcn_wave_barrier();
    regs_from_image(r, ls);
  }
  __device__ static void regs_from_image(NodeRegs& r, LevelScalars* ls) {
    // (every lane reads the same LDS words: readfirstlane tells the compiler the values are wave-uniform — SGPRs)
    const uint32_t* h = ls->sc;
    r.doneAt = r.doneAt0 = (long long)((unsigned long long)WG_READFIRST(h[HH_DONE_LO]) | ((unsigned long long)WG_READFIRST(h[HH_DONE_HI]) << 32));
    r.addedCycle = (int32_t)WG_READFIRST(h[HH_ADDED]);
    r.sigQueueSize = (int32_t)WG_READFIRST(h[HH_SIGQ]);
    r.msgFiltered = (int32_t)WG_READFIRST(h[HH_FILT]);
    r.startAt = (int32_t)WG_READFIRST(h[HH_START]);
    r.ls = ls;
  }
  __device__ static void node_end(Ctx& c, const State& s, NodeRegs& r) {
    const int32_t node = c.node;
    __builtin_amdgcn_wave_barrier();
    if (WG_LANE == 0) {
      uint32_t* h = r.ls->sc;
      h[HH_DONE_LO] = (uint32_t)(unsigned long long)r.doneAt;
      h[HH_DONE_HI] = (uint32_t)((unsigned long long)r.doneAt >> 32);
      h[HH_ADDED] = (uint32_t)r.addedCycle;
      h[HH_SIGQ] = (uint32_t)r.sigQueueSize;
      h[HH_FILT] = (uint32_t)r.msgFiltered;
      if (r.doneAt != r.doneAt0) c.d.nodes.doneAt[node] = r.doneAt;  // Node.doneAt proper (read-back, contIf)
    }
    store_levels(s, node, r.ls);
  }
  __device__ static void on_message(Ctx& c, const State& s, NodeRegs& r, int32_t from, uint32_t msg, uint32_t payload) {
    KPROF_DECL;
    on_new_sig(c, s, r, from, msg, payload);
    KPROF_COUNT(c.d.g, 4);
    KPROF_MARK(c.d.g, 5);
  }
  __device__ static void on_task(Ctx& c, const State& s, NodeRegs& r, uint32_t word, uint32_t arg) {
    KPROF_DECL;
    if (word == H_TASK_DISSEMINATION) {
      dissemination(c, s, r);
      KPROF_COUNT(c.d.g, 6);
      KPROF_MARK(c.d.g, 7);
    } else {
      update_verified(c, s, r, arg);
      KPROF_COUNT(c.d.g, 11);
      KPROF_MARK(c.d.g, 12);
    }
  }

  // header <-> LDS image, 16 bytes a lane: 640 bytes (L <= 16) are one memory instruction
  __device__ static void load_levels(const State& s, int32_t node, LevelScalars* ls) {
    const U4 WG_G* g = (const U4 WG_G*)h_hdr(s, node);
    __builtin_amdgcn_wave_barrier();
    for (int i = WG_LANE; i < (s.hdrStride >> 2); i += 64) scatter_levels(s, ls, i, g[i]);
    __builtin_amdgcn_wave_barrier();
  }
  // ... and back: only the 16-byte pieces that differ from what was loaded (an event changes the scalars' line and
  // its level's: the other lines of the record stay clean in L2 and are never written back to HBM)
  __device__ static void store_levels(const State& s, int32_t node, const LevelScalars* ls) {
    __builtin_amdgcn_wave_barrier();
    U4 WG_G* g = (U4 WG_G*)h_hdr(s, node);
    for (int i = WG_LANE; i < (s.hdrStride >> 2); i += 64) {
      const U4 q = gather_levels(ls, i), o = ls->orig[i];
      if (q.x != o.x || q.y != o.y || q.z != o.z || q.w != o.w) g[i] = q;
    }
  }

  // ---- lane-per-node form of onNewSig for k_deliver_msgs: the same statements as on_new_sig below, one
  // lane per receiving node; payloads wider than one word are handed back as a copy job --------------------
  struct LaneNode {
    long long doneAt;
    int32_t startAt, sigQueueSize, msgFiltered;
    int32_t sigQueueSize0, msgFiltered0;
    uint32_t qmask, qmask0;
  };
  __device__ static void lane_begin(const EngineDev& d, const State& s, int32_t node, LaneNode& r) {
    const uint32_t WG_G* h = h_hdr(s, node);
    r.doneAt = (long long)((unsigned long long)h[HH_DONE_LO] | ((unsigned long long)h[HH_DONE_HI] << 32));
    r.startAt = (int32_t)h[HH_START];
    r.sigQueueSize = r.sigQueueSize0 = (int32_t)h[HH_SIGQ];
    r.msgFiltered = r.msgFiltered0 = (int32_t)h[HH_FILT];
    r.qmask = r.qmask0 = h[HH_QMASK];
  }
  __device__ static void lane_end(const EngineDev&, const State& s, int32_t node, const LaneNode& r) {
    if (r.sigQueueSize != r.sigQueueSize0) h_hdr(s, node)[HH_SIGQ] = (uint32_t)r.sigQueueSize;
    if (r.msgFiltered != r.msgFiltered0) h_hdr(s, node)[HH_FILT] = (uint32_t)r.msgFiltered;
    if (r.qmask != r.qmask0) h_hdr(s, node)[HH_QMASK] = r.qmask;
  }
  __device__ static void lane_message(const EngineDev& d, const State& s, int32_t t, int32_t node, LaneNode& r,
                                      int32_t from, uint32_t msg, uint32_t payload, CopyJob& job) {
    const int l = (int)(msg & 31u);
    const bool levelFinished = (msg >> 5) & 1u;
    if (r.doneAt > 0) {  // :758-761
      r.msgFiltered++;
      return;
    }
    if (t < r.startAt) return;
    const int w = from >> 6;
    const uint64_t bit = 1ULL << (from & 63);
    uint64_t WG_G* fpp = s.FP + (size_t)node * s.W + w;
    const uint64_t WG_G* vip = s.VI + (size_t)node * s.W + w;
    uint64_t WG_G* tvp = s.TV + (size_t)node * s.W + w;
    const size_t nl = (size_t)node * s.L + l;
    // every load of the event before the first use
    const uint64_t viv = *vip;
    const uint64_t fpv = levelFinished ? *fpp : 0ULL;
    const uint64_t tvv = *tvp;
    const int32_t rank = s.ranks[(size_t)node * s.N + from];  // read at receive time (:769)
    uint32_t WG_G* qlo = h_lv(s, node, HP_QUSED_LO, l);
    uint32_t WG_G* qhi = h_lv(s, node, HP_QUSED_HI, l);
    uint32_t WG_G* qln = h_lv(s, node, HP_QLEN, l);
    const unsigned long long used = (unsigned long long)*qlo | ((unsigned long long)*qhi << 32);
    const int len = (int)*qln;
    const uint64_t WG_G* src = h_payload(d, s, payload);
    const int nw = h_nw(l);
    const uint64_t pw0 = nw == 1 ? src[0] & sib_view(node, l).mask : 0ULL;
    if (levelFinished) *fpp = fpv | bit;         // finishedPeers.set(from)
    if (!(viv & bit)) *tvp = tvv | bit;          // toVerifyInd.set(from) unless verified
    r.sigQueueSize++;
    const int qc = h_qcap(s, l);
    const unsigned long long capMask = qc >= 64 ? ~0ULL : ((1ULL << qc) - 1ULL);
    const unsigned long long freeM = ~used & capMask;
    if (freeM == 0 || len >= 64) {
      set_err(d.g, ERR_QUEUE_CAP);
      return;
    }
    const int slot = __ffsll(freeM) - 1;
    uint64_t WG_G* dst = sig_ptr(s, node, l, slot);
    s.qfrom[nl * s.Q + slot] = from;
    s.qent[nl * 64 + len] = ((uint64_t)(uint32_t)rank << 32) | (uint32_t)slot;
    if (slot < 32)
      *qlo = (uint32_t)used | (1u << slot);
    else
      *qhi = (uint32_t)(used >> 32) | (1u << (slot - 32));
    *qln = (uint32_t)(len + 1);
    r.qmask |= 1u << l;
    if (nw == 1) {
      dst[0] = pw0;
    } else {
      job.src = src;
      job.dst = dst;
      job.nw = nw;
    }
  }

  // ---- queue helpers ---------------------------------------------------------------------------
  __device__ static uint64_t WG_G* sig_ptr(const State& s, int32_t node, int l, int slot) {
    return s.qsig + s.qsigOff[l] + ((size_t)node * h_qcap(s, l) + slot) * (size_t)h_nw(l);
  }
  __device__ static bool slot_pending(const NodeRegs& r, int l, int slot) {
    bool p = false;
#pragma unroll
    for (int k = 0; k < H_PEND; k++) p |= (r.ls->sc[HH_PEND + k] == (0x80000000u | ((uint32_t)l << 8) | (uint32_t)slot));
    return p;
  }
  // ---- getRemainingPeers (:486-508), wave-parallel but sequentially equivalent ------------------
  // Scans the emission list from posInLevel, 64 peers per step. Accepted peers (not finished) are
  // appended to the dest ring at destOff (want > 1) or returned (want == 1). Returns the count.
  __device__ static int remaining_peers(Ctx& c, const State& s, LevelScalars* ls, int l, int want, uint32_t destOff,
                                        int32_t* single) {
    const int32_t node = c.node;
    const int size = 1 << (l - 1);
    const size_t peers0 = (size_t)node * (s.N - 1) + (size - 1);
    const uint64_t WG_G* fp = s.FP + (size_t)node * s.W;
    int pos = ls->pos[l];
    const int start = pos;
    int got = 0;
    bool fin = false;
    while (want > 0 && !fin) {
      int len = min(64, size - pos);
      int k = WG_LANE;
      bool in = k < len;
      int32_t p = in ? h_peer(s, peers0 + pos + k) : 0;
      bool ok = in && !row_get(fp, p);
      uint64_t okm = __ballot(ok);
      // a rejected peer whose successor position is `start` finishes the level (:499-503)
      int nextPos = pos + k + 1;
      if (nextPos >= size) nextPos = 0;
      uint64_t finm = __ballot(in && !ok && nextPos == start);
      // stop index: the want-th accepted lane, or the finishing lane, whichever comes first
      int stopAcc = 64, stopFin = finm ? __ffsll((unsigned long long)finm) - 1 : 64;
      if (__popcll(okm) >= want) {
        uint64_t m = okm;
        for (int i = 1; i < want; i++) m &= m - 1;
        stopAcc = __ffsll((unsigned long long)m) - 1;
      }
      int stop = min(stopAcc, stopFin);
      int consumed = stop < 64 ? stop + 1 : len;
      uint64_t take = okm & (consumed >= 64 ? ~0ULL : ((1ULL << consumed) - 1ULL));
      int ntake = __popcll(take);
      if (ok && k < consumed) {
        int idx = got + __popcll(take & lanes_lt());
        if (want == 1 && single && ntake >= 1 && idx == 0) *single = p;  // written by exactly one lane
        if (destOff != 0xFFFFFFFFu) c.dest_put(destOff, idx, p);
      }
      got += ntake;
      want -= ntake;
      pos += consumed;
      if (pos >= size) pos = 0;
      if (stopFin < 64 && stopFin <= stopAcc && stopFin < consumed) fin = true;
    }
    if (WG_LANE == 0) {
      ls->pos[l] = pos;
      if (fin) ls->outFin[l] = 1;
    }
    __builtin_amdgcn_wave_barrier();
    return got;
  }

  // snapshot of totalOutgoing of level l (the node's own block of the TI row) for a fast-path send
  // (irregular: engine payload ring); the periodic dissemination has its own computed slot.
  __device__ static uint32_t snapshot_outgoing(Ctx& c, const State& s, int l) {
    // A fast-path send happens when totalOutgoing of level l is complete (cur == 2^(l-1), :738-741): its snapshot
    // is the node's whole own block, i.e. all ones under the receiver's level mask. A sharded engine, whose payload
    // ring is private to the shard, sends that constant instead of a copy.
    if (c.d.sharded) return H_REF_ONES;
    Lv v = own_view(c.node, l);
    const uint64_t WG_G* ti = s.TI + (size_t)c.node * s.W;
    uint32_t ref = c.alloc_payload(v.nw);
    H_FOR_WORDS(v, j) c.d.payload[ref + j] = ti[v.bw + j] & v.mask;
    return ref | H_REF_RING;
  }

  // ---- Message.action: SendSigs -> onNewSig (:757-790) -------------------------------------------
  __device__ static void on_new_sig(Ctx& c, const State& s, NodeRegs& r, int32_t from, uint32_t msg, uint32_t payload) {
    const int l = (int)(msg & 31u);
    const bool levelFinished = (msg >> 5) & 1u;
    const int32_t node = c.node;
    if (r.doneAt > 0) {
      r.msgFiltered++;
      return;
    }
    if (c.t < r.startAt) return;
    LevelScalars* ls = r.ls;
    // Everything this event reads from HBM depends only on (node, from, payload): issue it all before the
    // first use so the event costs ONE memory round trip (the path is latency-bound).
    const int w = from >> 6;
    const uint64_t bit = 1ULL << (from & 63);
    uint64_t WG_G* fpp = s.FP + (size_t)node * s.W + w;
    uint64_t WG_G* vip = s.VI + (size_t)node * s.W + w;
    uint64_t WG_G* tvp = s.TV + (size_t)node * s.W + w;
    const uint64_t fpv = ld_coherent(fpp), viv = ld_coherent(vip), tvv = ld_coherent(tvp);
    const int32_t rank = s.ranks[(size_t)node * s.N + from];  // read at receive time (:769)
    const Lv v = sib_view(node, l);
    const uint64_t WG_G* src = h_payload(c.d, s, payload);
    const int j0 = (int)((WG_LANE - v.bw) & 63);
    const bool has0 = j0 < v.nw;
    uint64_t pw0 = 0;
    if (has0) pw0 = src[j0] & v.mask;
    const bool owner = (int)WG_LANE == (w & 63);
    if (levelFinished && owner) *fpp = fpv | bit;                 // finishedPeers.set(from)
    if (!(viv & bit) && owner) *tvp = tvv | bit;                  // toVerifyInd.set(from) unless verified
    r.sigQueueSize++;
    // toVerifyAgg.add(new SigToVerify(from, level, receptionRanks[from], cs, badSig))
    unsigned long long used = ls_qused(ls, l);
    const int qc = h_qcap(s, l);
    unsigned long long capMask = qc >= 64 ? ~0ULL : ((1ULL << qc) - 1ULL);
    unsigned long long freeM = ~used & capMask;
    int len = ls->qlen[l];
    if (freeM == 0 || len >= 64) {
      if (WG_LANE == 0) set_err(c.d.g, ERR_QUEUE_CAP);
      return;
    }
    int slot = __ffsll(freeM) - 1;
    uint64_t WG_G* dst = sig_ptr(s, node, l, slot);
    if (has0) dst[j0] = pw0;
    for (int j = j0 + 64; j < v.nw; j += 64) dst[j] = src[j];
    __builtin_amdgcn_wave_barrier();  // every lane has read qused/qlen before lane 0 replaces them
    if (WG_LANE == 0) {
      s.qfrom[((size_t)node * s.L + l) * s.Q + slot] = from;
      s.qent[((size_t)node * s.L + l) * 64 + len] = ((uint64_t)(uint32_t)rank << 32) | (uint32_t)slot;
      ls_set_qused(ls, l, used | (1ULL << slot));
      ls->qlen[l] = len + 1;
      ls->sc[HH_QMASK] |= 1u << l;
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- PeriodicTask: dissemination (:331-343) -> HLevel.doCycle (:474-484) ------------------------
  // The reference walks the levels one after the other; done literally that is one chain of dependent
  // HBM round trips per level (peer id -> its finishedPeers bit -> the snapshot's words). Here lane l
  // resolves level l's isOpen() and first candidate peer, so all levels' lookups are two round trips in
  // total, and the snapshots of all open levels are one flat copy. A level whose first candidate is a
  // finished peer (the scan of getRemainingPeers :486-508 has to go on) takes the sequential path.
  __device__ static void dissemination(Ctx& c, const State& s, NodeRegs& r) {
    if (r.doneAt > 0) {
      if (r.addedCycle > 0)
        r.addedCycle--;
      else
        return;
    }
    LevelScalars* ls = r.ls;
    const int lane = WG_LANE;
    const int32_t node = c.node;
    // |totalOutgoing| of level l = sum of |totalIncoming| below l
    const int cti = lane < s.L ? ls->cTI[lane] : 0;
    const int incl = (int)wave_incl_scan32((uint32_t)cti);
    const int below = incl - cti;
    bool open = false, fin = false;
    int32_t cand = 0, cand2 = 0;
    int myPos = 0;
    bool two = false, fin2 = true;
    const int mySize = (lane >= 1 && lane < s.L) ? 1 << (lane - 1) : 0;
    if (mySize) {
      open = !ls->outFin[lane] && (c.t >= (lane - 1) * s.p.levelWaitTime || below == mySize);  // isOpen :458-472
      if (open) {
        myPos = ls->pos[lane];
        const size_t at = (size_t)node * (s.N - 1) + (mySize - 1) + myPos;
        cand = h_peer(s, at);
        // ... and the peer after it, in the same round trip: when the first candidate is a finished peer the scan of
        // getRemainingPeers usually ends at the next one. Only where neither a wrap of posInLevel nor the end-of-scan test
        // (`posInLevel == start`, :499-503) can fall between the two: a level of more than two peers, not at its last position
        two = mySize > 2 && myPos + 1 < mySize;
        if (two) cand2 = h_peer(s, at + 1);
      }
    }
    KPROF_DECL;
    const uint64_t openM = __ballot(open);
    if (!openM) return;
    if (open) {
      const uint64_t WG_G* fpRow = s.FP + (size_t)node * s.W;
      const uint64_t w1 = ld_coherent(fpRow + (cand >> 6));
      const uint64_t w2 = two ? ld_coherent(fpRow + (cand2 >> 6)) : ~0ULL;
      fin = (w1 >> (cand & 63)) & 1ULL;
      fin2 = (w2 >> (cand2 & 63)) & 1ULL;
      if (fin && two && !fin2) {  // the first candidate is rejected (no effect but posInLevel++), the second one is taken
        cand = cand2;
        myPos++;
        fin = false;
      }
    }
    const bool lf = cti == mySize;  // incomingComplete :524-526
    // snapshots (SendSigs.sigs = totalOutgoing.clone() :254): totalOutgoing of level l is the node's own aligned
    // block of 2^(l-1) ids in the TI row, and those blocks are nested — so ONE copy of the highest open level's
    // block serves every level: level l's message points at its sub-range (the receiver masks single-word
    // blocks with its level mask). Loads first, stores after.
    const uint32_t win = ((uint32_t)c.t / (uint32_t)s.p.disseminationPeriodMs) % s.snapNb;
    const uint32_t refBase = (win * (uint32_t)s.N + (uint32_t)node) * s.snapStride;
    const Lv tv = own_view(node, 63 - __clzll((unsigned long long)openM));
    {
      const uint64_t WG_G* ti = s.TI + (size_t)node * s.W + tv.bw;
      for (int j0 = 0; j0 < tv.nw; j0 += 256) {
        uint64_t v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int j = j0 + q * 64 + lane;
          v[q] = j < tv.nw ? ti[j] & tv.mask : 0ULL;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int j = j0 + q * 64 + lane;
          if (j < tv.nw) s.snap[refBase + j] = v[q];
        }
      }
    }
    KPROF_MARK(c.d.g, 9);   // snapshots
    const uint64_t okM = __ballot(open && !fin);
    KPROF_MARK(c.d.g, 8);   // candidate peers' finished bits
    if (open && !fin) ls->pos[lane] = myPos + 1 >= mySize ? 0 : myPos + 1;  // getRemainingPeers(1) took the candidate
    __builtin_amdgcn_wave_barrier();
    // levels whose first candidate is a finished peer: the sequential scan of getRemainingPeers, before any send
    // (the scan does not depend on the sends; the records keep level order through their rank)
    uint64_t sendM = okM;
    for (uint64_t m = openM & ~okM; m; m &= m - 1) {
      const int l = __ffsll((unsigned long long)m) - 1;
      int32_t dest = -1;
      const int got = remaining_peers(c, s, ls, l, 1, 0xFFFFFFFFu, &dest);
      KPROF_COUNT(c.d.g, 13);
      if (got <= 0) continue;
      dest = (int32_t)lane_bcast((uint32_t)dest, __ffsll((unsigned long long)__ballot(dest >= 0)) - 1);
      if (lane == l) cand = dest;
      sendM |= 1ULL << l;
    }
    if (sendM) {  // one rd.nextInt() per send, in level order (:374-382): lane l writes level l's record
      long long bytes = 0;
      for (uint64_t m = sendM; m; m &= m - 1) bytes += h_msg_size(__ffsll((unsigned long long)m) - 1);
      c.send_many((sendM >> lane) & 1ULL, __popcll(sendM & lanes_lt()), __popcll(sendM), cand,
                  (uint32_t)lane | (lf ? 32u : 0u), refBase + (uint32_t)(own_view(node, lane >= 1 ? lane : 1).bw - tv.bw), bytes);
    }
    KPROF_MARK(c.d.g, 10);  // sends
  }

  // ---- Task: updateVerifiedSignatures (:690-754) --------------------------------------------------
  __device__ static void update_verified(Ctx& c, const State& s, NodeRegs& r, uint32_t arg) {
    const int32_t node = c.node;
    const int lane = WG_LANE;
    // (the task's record is read from the node's LDS image, indexed at run time there: a register array indexed at run
    // time would put the whole NodeRegs in scratch memory — every access to any of its fields a memory round trip)
    LevelScalars* ls = r.ls;
    const int pk = (int)(arg & (H_PEND - 1));
    const uint32_t pe = WG_READFIRST(ls->sc[HH_PEND + pk]);
    const int32_t from = (int32_t)WG_READFIRST(ls->sc[HH_PENDFROM + pk]);
    if (!(pe & 0x80000000u)) {
      if (lane == 0) set_err(c.d.g, ERR_PROTOCOL);
      return;
    }
    const int lv = (int)((pe >> 8) & 0xFF), slot = (int)(pe & 0xFF);
    __builtin_amdgcn_wave_barrier();  // every lane has read the record before lane 0 clears it
    if (lane == 0) ls->sc[HH_PEND + pk] = 0;
    const Lv v = sib_view(node, lv);
    uint64_t WG_G* ti = s.TI + (size_t)node * s.W;
    uint64_t WG_G* la = s.LA + (size_t)node * s.W;
    uint64_t WG_G* vi = s.VI + (size_t)node * s.W;
    const uint64_t WG_G* sig = sig_ptr(s, node, lv, slot);
    // ---- every load of the event, issued before the first use (one memory round trip)
    const int wF = from >> 6, jF = wF - v.bw;  // `from` lies in the level's block
    const uint64_t bit = 1ULL << (from & 63);
    uint64_t WG_G* tvp = s.TV + (size_t)node * s.W + wF;
    const uint64_t tvv = ld_coherent(tvp);
    uint64_t WG_G* ent = s.qent + ((size_t)node * s.L + lv) * 64;
    const int len = ls->qlen[lv];
    const uint64_t myEnt = lane < len ? ent[lane] : ~0ULL;
    const int j0 = (int)((lane - v.bw) & 63);
    const bool has0 = j0 < v.nw;
    uint64_t sg0 = 0, vi0 = 0, la0 = 0, ti0 = 0;
    if (has0) {
      sg0 = sig[j0];
      vi0 = vi[v.bw + j0];
      la0 = la[v.bw + j0];
      ti0 = ti[v.bw + j0];
    }
    // the VI / TI words holding `from` are among the block words just loaded (the lane owning row word wF);
    // only beyond the first 64 words of a wide level do they cost memory instructions of their own
    uint64_t viF, tiF;
    if (jF < 64) {
      viF = lane_bcast64(vi0, wF & 63);
      tiF = lane_bcast64(ti0, wF & 63);
    } else {
      viF = ld_coherent(vi + wF);
      tiF = ld_coherent(ti + wF);
    }
    const bool owner = lane == (wF & 63);
    if (owner) *tvp = tvv & ~bit;  // toVerifyInd.set(from, false)
    // toVerifyAgg.remove(vs): identity remove, sigQueueSize untouched
    {
      const uint64_t hit = __ballot(lane < len && (int)(myEnt & 0xFF) == slot);
      if (hit) {
        const int at = __ffsll((unsigned long long)hit) - 1;
        const uint64_t next = shfl64(myEnt, (lane + 1) & 63);
        if (lane >= at && lane < len - 1) ent[lane] = next;
        if (lane == 0) {
          ls->qlen[lv] = len - 1;
          if (len == 1) ls->sc[HH_QMASK] &= ~(1u << lv);
        }
      }
    }
    const bool hadVI = (viF & bit) != 0, hadTI = (tiF & bit) != 0;
    // verifiedIndSignatures.set(from); totalIncoming.set(from) if new — applied to the register copies of
    // the words and written back by the owning lane
    if (owner) {
      if (!hadVI) vi[wF] = viF | bit;
    }
    int cVI = ls->cVI[lv] + (hadVI ? 0 : 1);
    int cTI = ls->cTI[lv];
    int cLA = ls->cLA[lv];
    bool improved = false;
    if (!hadTI) {
      cTI++;
      improved = true;
    }
    const uint64_t ti0m = ti0;  // the word as it is in memory
    if (has0 && j0 == jF) {
      vi0 |= bit;
      if (!hadTI) ti0 |= bit;
    }
    // all = sig | verifiedInd ; intersects(lastAgg, sig)
    uint64_t acc = 0;
    if (has0) acc = (uint64_t)__popcll(sg0 | (vi0 & v.mask)) | ((uint64_t)((sg0 & la0 & v.mask) != 0) << 32);
    for (int j = j0 + 64; j < v.nw; j += 64) {
      uint64_t sg = sig[j], viw = vi[v.bw + j] & v.mask, law = la[v.bw + j] & v.mask;
      if (j == jF) viw |= bit;
      acc += (uint64_t)__popcll(sg | viw) | ((uint64_t)((sg & law) != 0) << 32);
    }
    acc = wave_sum64(acc);
    const int u2 = (int)(acc & 0xFFFFFFFFu);
    const bool inter = (acc >> 32) != 0;
    if (u2 > cVI) {
      improved = true;
      uint64_t cnt = 0;
      // (only the words that change are written: a verified aggregate usually adds a few bits to a block whose other
      // words — up to 2 KB of them — would otherwise be dirtied in L2 and written back to HBM for nothing)
      if (has0) {
        uint64_t nla = (inter ? 0ULL : (la0 & v.mask)) | sg0;
        uint64_t nti = nla | (vi0 & v.mask);
        if (nla != (la0 & v.mask)) la[v.bw + j0] = (la0 & ~v.mask) | nla;
        if (nti != (ti0m & v.mask)) ti[v.bw + j0] = (ti0m & ~v.mask) | nti;
        cnt = (uint64_t)__popcll(nla) | ((uint64_t)__popcll(nti) << 32);
      }
      for (int j = j0 + 64; j < v.nw; j += 64) {
        uint64_t sg = sig[j];
        uint64_t law = la[v.bw + j], viw = vi[v.bw + j], tiw = ti[v.bw + j];
        if (j == jF) viw |= bit;  // (this lane stored it above; same-lane order makes the reload see it anyway)
        uint64_t nla = (inter ? 0ULL : (law & v.mask)) | sg;
        uint64_t nti = nla | (viw & v.mask);
        if (nla != (law & v.mask)) la[v.bw + j] = (law & ~v.mask) | nla;
        if (nti != (tiw & v.mask)) ti[v.bw + j] = (tiw & ~v.mask) | nti;
        cnt += (uint64_t)__popcll(nla) | ((uint64_t)__popcll(nti) << 32);
      }
      cnt = wave_sum64(cnt);
      cLA = (int)(cnt & 0xFFFFFFFFu);
      cTI = (int)(cnt >> 32);
    } else if (!hadTI && owner) {
      ti[wF] = tiF | bit;
    }
    if (lane == 0) {
      ls->cVI[lv] = cVI;
      ls->cTI[lv] = cTI;
      ls->cLA[lv] = cLA;
      // The entry was just unlisted (an entry is listed at most once), so its slot dies with this task
      // unless another registered task still references it (checkSigs can pick the same entry twice).
      if (!slot_pending(r, lv, slot)) ls_set_qused(ls, lv, ls_qused(ls, lv) & ~(1ULL << slot));
    }
    __builtin_amdgcn_wave_barrier();
    if (!improved) return;
    const bool justCompleted = cTI == v.size;  // incomingComplete()
    // totalOutgoing(l) = the sum of |totalIncoming| below l (:728-731): lane l takes level l, one prefix scan over the
    // lanes instead of a loop of dependent LDS reads; the fast path (:738-749) is rare, its levels come from a ballot
    const int myCTI = lane < s.L ? ls->cTI[lane] : 0;
    const int incl = (int)wave_incl_scan32((uint32_t)myCTI);
    const int cur = (int)lane_bcast((uint32_t)incl, 63);
    const bool fp = justCompleted && s.p.fastPath > 0 && lane > lv && lane < s.L && !ls->outFin[lane] &&
                    incl - myCTI == (1 << (lane - 1));
    for (uint64_t fpM = __ballot(fp); fpM; fpM &= fpM - 1) {
      const int l = __ffsll((unsigned long long)fpM) - 1;
      __threadfence_block();  // the snapshot below reads the totalIncoming words stored above
      uint32_t destOff = c.dest_reserve(s.p.fastPath);
      int n = remaining_peers(c, s, ls, l, s.p.fastPath, destOff, nullptr);
      if (n > 0) {
        uint32_t ref = snapshot_outgoing(c, s, l);
        bool lf = ls->cTI[l] == (1 << (l - 1));
        __threadfence_block();
        c.send_list(destOff, n, (uint32_t)l | (lf ? 32u : 0u), ref, h_msg_size(l));
      }
    }
    if (r.doneAt == 0 && cur >= s.p.threshold) r.doneAt = c.t;
  }
};

// ---- conditional-task phase (C/Network.java:543-566 driving HNode.checkSigs :796-837) -------------
// PRE: which conditional tasks run at this edge — one lane per node, coalesced reads of the four words
// that decide it; the runners go to a compact list (one atomic per wavefront) so the expensive part
// below is launched over runners only, not over all N nodes.
__global__ void __launch_bounds__(256) k_handel_cond_pre(const EngineDev* __restrict__ tab,
                                                         const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[blockIdx.y];
  const int32_t t = d.g->now, until = d.g->until;
  const uint32_t epoch = d.g->epoch;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t n0 = (uint32_t)s.lo + blockIdx.x * blockDim.x; n0 < (uint32_t)s.hi; n0 += stride) {
    const uint32_t node = n0 + threadIdx.x;
    // nextMessage(): drop from the private copy if minStartTime > until or the node is down; evaluate
    // at most once per call (epoch); evaluate only when minStartTime <= time.
    bool run = false;
    if (node < (uint32_t)s.hi) {
      uint32_t WG_G* ct = s.ct + 2 * (size_t)node;
      uint32_t WG_G* h = h_hdr(s, (int32_t)node);
      const uint32_t ctMin = ct[0], ctEpoch = ct[1];
      if (!d.nodes.down[node] && ctEpoch != epoch) {
        const int32_t ms = (int32_t)ctMin;
        if (ms <= until && ms <= t) {
          ct[1] = epoch;
          run = h[HH_SIGQ] != 0;  // startIf = hasSigToVerify (:345-347)
        }
      }
      if (run) {
        ct[0] = (uint32_t)(t + (int32_t)h[HH_PAIR]);  // minStartTime = time + duration (:557-560)
        // sigQueueSize drifts above the real queue lengths: checkSigs then runs over empty
        // lists, finds no candidate, draws nothing and changes nothing (:800-806) — such a node needs no visit
        if (h[HH_QMASK] == 0) run = false;
      }
      if (!run) s.candCnt[node] = 0;
    }
    const uint64_t m = __ballot(run);
    if (m) {
      uint32_t base = 0;
      const int leader = __ffsll((unsigned long long)m) - 1;
      if ((int)WG_LANE == leader) base = atomicAdd(F(s.runCount + 0), (uint32_t)__popcll(m));
      base = lane_bcast(base, leader);
      if (run) s.runList[base + __popcll(m & lanes_lt())] = node;
    }
  }
}

// A1: bestToVerify for every level (:570-634) of every runner: curates the lists, records the candidates.
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k_handel_cond_a1(const EngineDev* __restrict__ tab,
                                                             const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[blockIdx.y];
  __shared__ LevelScalars shLevels[4];
  const int lane = WG_LANE;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t nRun = *s.runCount;
  LevelScalars* ls = &shLevels[threadIdx.x >> 6];
  // Software-pipelined over the runners of this wavefront: while runner q is worked on, the header of runner
  // q + nWaves and the id of runner q + 2 nWaves are in flight (a runner's checkSigs touches its own node only).
  if (wave >= nRun) return;
  int32_t nodeCur = (int32_t)s.runList[wave];
  HandelProto::Pre hdrCur = HandelProto::prefetch(s, nodeCur);
  int32_t nodeNext = wave + nWaves < nRun ? (int32_t)s.runList[wave + nWaves] : 0;
  // list entries of the lowest level with a queue, from a record still in registers: lane l's piece is words 4l..4l+3
  auto first_entries = [&](const HandelProto::Pre& h, int32_t nd) -> uint64_t {
    const uint32_t qm = WG_READLANE(h.q0.w, 2) & ~1u;  // HH_QMASK = word 11
    if (!qm) return ~0ULL;
    const int l0 = __ffs(qm) - 1;
    const int len0 = (int)WG_READLANE(h.q0.x, (HH_LV + l0 * HP_COUNT + HP_QLEN) >> 2);
    return lane < len0 ? s.qent[((size_t)nd * s.L + l0) * 64 + lane] : ~0ULL;
  };
  uint64_t entFirst = first_entries(hdrCur, nodeCur);
  for (uint32_t q = wave; q < nRun; q += nWaves) {
    KPROF_DECL;
    KPROF_COUNT(d.g, 16);
    const int32_t node = nodeCur;
    const bool haveNext = q + nWaves < nRun;
    HandelProto::Pre hdrNext = hdrCur;
    int32_t nodeNext2 = 0;
    if (haveNext) {
      hdrNext = HandelProto::prefetch(s, nodeNext);
      if (q + 2 * nWaves < nRun) nodeNext2 = (int32_t)s.runList[q + 2 * nWaves];
    }
    {
      const int n4 = s.hdrStride >> 2;
      __builtin_amdgcn_wave_barrier();  // the previous runner's store_levels has read the image
      if (lane < n4) HandelProto::scatter_levels(s, ls, lane, hdrCur.q0);
      if (lane + 64 < n4) HandelProto::scatter_levels(s, ls, lane + 64, ((const U4*)h_hdr(s, node))[lane + 64]);
      __builtin_amdgcn_wave_barrier();
    }
    nodeCur = nodeNext;
    nodeNext = nodeNext2;
    hdrCur = hdrNext;
    KPROF_MARK(d.g, 17);  // header image (prefetched a runner ahead)
    const uint64_t WG_G* ti = s.TI + (size_t)node * s.W;
    const uint64_t WG_G* la = s.LA + (size_t)node * s.W;
    const uint64_t WG_G* vi = s.VI + (size_t)node * s.W;
    const int window = (int)WG_READFIRST(ls->sc[HH_WINDOW]);
    int sigQueueSize = (int)WG_READFIRST(ls->sc[HH_SIGQ]);
    uint32_t pend[H_PEND];
#pragma unroll
    for (int k = 0; k < H_PEND; k++) pend[k] = WG_READFIRST(ls->sc[HH_PEND + k]);
    int ncand = 0;
    // levels with a non-empty queue; the next level's list entries are fetched while this one is worked on
    uint32_t lvMask = WG_READFIRST(ls->sc[HH_QMASK]) & ~1u;  // (bit l <=> qlen[l] > 0, kept by every writer of qlen)
    // (the first level's list entries were requested at the end of the previous runner — before ITS stores: loads and
    // stores retire through one in-order counter on this ISA, so a load issued after stores waits for their acknowledgement)
    uint64_t entNext = entFirst;
    while (lvMask) {
      const int l = __ffs(lvMask) - 1;
      lvMask &= lvMask - 1;
      const int len = ls->qlen[l];
      const Lv v = sib_view(node, l);
      uint64_t WG_G* ent = s.qent + ((size_t)node * s.L + l) * 64;
      const uint64_t myEnt = entNext;
      entNext = ~0ULL;
      if (lvMask) {
        const int ln = __ffs(lvMask) - 1;
        if (lane < ls->qlen[ln]) entNext = s.qent[((size_t)node * s.L + ln) * 64 + lane];
      }
      const int mySlot = lane < len ? (int)(myEnt & 0xFF) : 0;
      const int myRank = lane < len ? (int)(uint32_t)(myEnt >> 32) : INT32_MAX;
      KPROF_COUNT(d.g, 18);
      KPROF_MARK(d.g, 19);  // the level's list entries
      const int windowIndex = wave_reduce_min_i32(myRank);  // Collections.min(rank)
      const int curSize = ls->cTI[l], cLA = ls->cLA[l];
      int bestInside = -1, bestScore = 0, bestOutside = -1, bestOutsideRank = 0;
      uint64_t keep = 0;
      if (v.nw == 1) {
        // the level's block fits one 64-bit word (levels <= 7): one lane per queue entry, no reductions.
        // Same selection as the sequential walk below: best inside = FIRST entry with the strictly
        // greatest positive score, best outside = FIRST entry with the smallest rank.
        const bool mineIn = lane < len;
        const uint64_t sg = mineIn ? *HandelProto::sig_ptr(s, node, l, mySlot) : 0ULL;
        const uint64_t tiw = ti[v.bw] & v.mask, viw = vi[v.bw] & v.mask, law = la[v.bw] & v.mask;
        const int u1 = __popcll(sg | tiw | viw), u2 = __popcll(sg | viw), cs = __popcll(sg);
        const bool iTI = (sg & tiw) != 0, iLA = (sg & law) != 0;
        const int sII = iTI ? u2 : u1;  // sizeIfIncluded :532-540
        const bool kept1 = mineIn && sII > curSize;
        const bool inside = kept1 && myRank <= windowIndex + window;
        const bool outside = kept1 && !inside;
        int score = 0;  // score(l, sig) :655-668
        if (inside) score = cLA >= v.size ? 0 : (!iLA ? cLA + cs : max(0, u2 - cLA));
        keep = __ballot(kept1);
        const int maxScore = wave_reduce_max_i32(score), minRank = wave_reduce_min_i32(outside ? myRank : INT32_MAX);
        if (maxScore > 0) {
          const uint64_t mm = __ballot(inside && score == maxScore);
          bestInside = (int)lane_bcast((uint32_t)mySlot, __ffsll((unsigned long long)mm) - 1);
        }
        const uint64_t om = __ballot(outside && myRank == minRank);
        if (om) bestOutside = (int)lane_bcast((uint32_t)mySlot, __ffsll((unsigned long long)om) - 1);
        KPROF_MARK(d.g, 20);  // a single-word level
      } else {
      // blocks of up to 64 words (levels <= 13): the level's three row words of this lane are loaded once, not once
      // per queue entry
      const int jh = (int)((lane - v.bw) & 63);
      const bool oneRound = v.nw <= 64;
      uint64_t tih = 0, vih = 0, lah = 0;
      if (oneRound && jh < v.nw) {
        tih = ti[v.bw + jh];
        vih = vi[v.bw + jh];
        lah = la[v.bw + jh];
      }
      // (the entries' signatures are independent of each other: entry i + 1's word is in flight while entry i is reduced)
      uint64_t sgAhead = 0;
      {
        const int slot0 = (int)lane_bcast((uint32_t)mySlot, 0);
        if (oneRound && jh < v.nw && len > 0) sgAhead = HandelProto::sig_ptr(s, node, l, slot0)[jh];
      }
      for (int i = 0; i < len; i++) {
        const int slot = (int)lane_bcast((uint32_t)mySlot, i);
        const int rank = (int)lane_bcast((uint32_t)myRank, i);
        const uint64_t WG_G* sig = HandelProto::sig_ptr(s, node, l, slot);
        uint64_t a = 0, b = 0;
        if (oneRound) {
          const uint64_t sgCur = sgAhead;
          const int slotN = (int)lane_bcast((uint32_t)mySlot, i + 1 < len ? i + 1 : i);
          if (i + 1 < len && jh < v.nw) sgAhead = HandelProto::sig_ptr(s, node, l, slotN)[jh];
          if (jh < v.nw) {
            const uint64_t sg = sgCur;
            a = (uint64_t)__popcll(sg | tih | vih) | ((uint64_t)__popcll(sg | vih) << 21) | ((uint64_t)__popcll(sg) << 42);
            b = (uint64_t)((sg & tih) != 0) | ((uint64_t)((sg & lah) != 0) << 21);
          }
        } else
        H_FOR_WORDS(v, j) {
          uint64_t sg = sig[j], tiw = ti[v.bw + j] & v.mask, viw = vi[v.bw + j] & v.mask, law = la[v.bw + j] & v.mask;
          a += (uint64_t)__popcll(sg | tiw | viw) | ((uint64_t)__popcll(sg | viw) << 21) | ((uint64_t)__popcll(sg) << 42);
          b += (uint64_t)((sg & tiw) != 0) | ((uint64_t)((sg & law) != 0) << 21);
        }
        a = wave_sum64(a);
        b = wave_sum64(b);
        const int u1 = (int)(a & 0x1FFFFF), u2 = (int)((a >> 21) & 0x1FFFFF), cs = (int)((a >> 42) & 0x1FFFFF);
        const bool iTI = (b & 0x1FFFFF) != 0, iLA = ((b >> 21) & 0x1FFFFF) != 0;
        const int sII = iTI ? u2 : u1;  // sizeIfIncluded :532-540
        if (sII > curSize) {
          keep |= 1ULL << i;
          if (rank <= windowIndex + window) {
            int score;  // score(l, sig) :655-668
            if (cLA >= v.size)
              score = 0;
            else if (!iLA)
              score = cLA + cs;
            else
              score = max(0, u2 - cLA);
            if (score > bestScore) {
              bestScore = score;
              bestInside = slot;
            }
          } else if (bestOutside < 0 || rank < bestOutsideRank) {
            bestOutside = slot;
            bestOutsideRank = rank;
          }
        }
      }
      KPROF_ADD(d.g, 21, len);
      KPROF_MARK(d.g, 22);  // a multi-word level (kprof21: its entries)
      }
      const int kept = __popcll(keep);
      if (kept != len) {  // replaceToVerifyAgg :636-646
        int newPos = __popcll(keep & lanes_lt());
        bool mineKept = lane < len && ((keep >> lane) & 1ULL);
        if (mineKept) ent[newPos] = myEnt;
        // slots of dropped entries are released unless a registered task still holds them
        bool mineDropped = lane < len && !mineKept;
        bool held = false;
#pragma unroll
        for (int k = 0; k < H_PEND; k++) held |= pend[k] == (0x80000000u | ((uint32_t)l << 8) | (uint32_t)mySlot);
        uint64_t rel = __ballot(mineDropped && !held);
        unsigned long long relMask = 0;
        for (uint64_t m = rel; m; m &= m - 1) relMask |= 1ULL << lane_bcast((uint32_t)mySlot, __ffsll((unsigned long long)m) - 1);
        if (lane == 0) {
          ls_set_qused(ls, l, ls_qused(ls, l) & ~relMask);
          ls->qlen[l] = kept;
          if (kept == 0) ls->sc[HH_QMASK] &= ~(1u << l);
        }
        sigQueueSize += kept - len;
        __builtin_amdgcn_wave_barrier();
      }
      const int cand = bestInside >= 0 ? bestInside : bestOutside;
      if (cand >= 0) {
        if (lane == 0) {  // (into the record's image: stored with it at the end of the runner, no memory traffic here)
          uint32_t* cw = ls->sc + HH_CAND + (ncand >> 1);
          const uint32_t e16 = ((uint32_t)l << 8) | (uint32_t)cand;
          *cw = (ncand & 1) ? ((*cw & 0xFFFFu) | (e16 << 16)) : ((*cw & 0xFFFF0000u) | e16);
        }
        ncand++;
      }
    }
    __builtin_amdgcn_wave_barrier();
    entFirst = q + nWaves < nRun ? first_entries(hdrCur, nodeCur) : ~0ULL;  // (hdrCur / nodeCur: the next runner's by now)
    if (lane == 0) {
      ls->sc[HH_SIGQ] = (uint32_t)sigQueueSize;
      s.candCnt[node] = (uint8_t)ncand;
    }
    HandelProto::store_levels(s, node, ls);
    __builtin_amdgcn_wave_barrier();
    KPROF_MARK(d.g, 23);  // list curation, candidates, header store
  }
}

// scan over nodes: ordinal of each node that draws (checkSigs draws iff some level has a candidate),
// and the draw itself — chooseBestFromLevels: rd.nextInt(byLevels.size()) (:788-790) — by jump-ahead
// assuming no earlier nextInt(bound) rejection; a rejection anywhere is flagged and re-walked in A2.
struct CondF {
  typedef HandelState Aux;
  const EngineDev& d;
  const HandelState& s;
  __device__ CondF(const EngineDev& d_, const Aux* a) : d(d_), s(*a) {}
  __device__ uint32_t count() const { return (uint32_t)s.N; }
  __device__ uint64_t value(uint32_t i) const { return s.candCnt[i] > 0; }
  __device__ void tally(uint32_t, uint32_t) const {}
  __device__ void total(uint64_t tot) const {
    d.g->nOut = (uint32_t)tot;  // one registerTask per drawing node
    d.g->nDraws = (uint32_t)tot;
  }
  __device__ void write(uint32_t i, uint64_t excl, bool valid) const {
    if (!valid || s.candCnt[i] == 0) return;
    s.condList[(uint32_t)excl] = i;
    uint64_t st = lcg_skip(d.g->rng, excl);
    int consumed;
    s.drawVal[i] = lcg_next_int_bounded(st, (int32_t)s.candCnt[i], &consumed);
    if (consumed != 1) d.g->rejectSeen = 1;
  }
};

// A2: the rest of checkSigs (:816-836) for the drawn candidate, one lane per drawing node.
// SH (sharded engine): a drawing node is handled by its owner, the task record goes to the exchange image.
template <bool SH>
__global__ void __launch_bounds__(256) k_handel_cond_a2(const EngineDev* __restrict__ tab,
                                                        const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[blockIdx.y];
  const uint32_t n = d.g->nOut;
  const int32_t t = d.g->now;
  const bool rejected = d.g->rejectSeen != 0;
  const uint32_t D = (uint32_t)d.horizon;
  const uint32_t stride = gridDim.x * blockDim.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) *s.runCount = 0;  // for the next edge's k_handel_cond_pre
  for (uint32_t j0 = blockIdx.x * blockDim.x; j0 < n; j0 += stride) {
    const uint32_t j = j0 + threadIdx.x;
    uint32_t histKey = 0xFFFFFFFFu;
    if (j < n) {
      const int32_t node = (int32_t)s.condList[j];
      int k = s.drawVal[node];
      const bool owned = !SH || shard_owns(d, node);
      if (rejected && (owned || j + 1 == n)) {  // a nextInt(bound) rejection shifted the stream: walk it serially up to this draw (rare)
        uint64_t st = d.g->rng;
        uint32_t total = 0;
        for (uint32_t q = 0; q <= j; q++) {
          int consumed;
          k = lcg_next_int_bounded(st, (int32_t)s.candCnt[s.condList[q]], &consumed);
          total += (uint32_t)consumed;
        }
        if (j + 1 == n) d.g->nDraws = total;
      }
      if (!owned) {
        for (int q = 0; q < 5; q++) d.xbuf[(size_t)j * 5 + q] = 0;
        continue;
      }
      uint32_t WG_G* h = h_hdr(s, node);
      const uint32_t e16 = (h[HH_CAND + (k >> 1)] >> ((k & 1) * 16)) & 0xFFFFu;
      const int l = (int)(e16 >> 8);
      const int slot = (int)(e16 & 0xFFu);
      const int32_t from = s.qfrom[((size_t)node * s.L + l) * s.Q + slot];
      // currWindowSize = min(window.newSize(cur, correct = true), l.size)  (:821-822, ScoringExp :192-200)
      int w = (int)h[HH_WINDOW] * 2;
      if (w > s.p.windowMaximum) w = s.p.windowMaximum;
      if (w < s.p.windowMinimum) w = s.p.windowMinimum;
      h[HH_WINDOW] = (uint32_t)min(w, 1 << (l - 1));
      // receptionRanks[best.from] += nodeCount, saturating (:825-828)
      int32_t WG_G* rk = s.ranks + (size_t)node * s.N + from;
      int32_t nr = (int32_t)((uint32_t)*rk + (uint32_t)s.N);
      *rk = nr < 0 ? INT32_MAX : nr;
      if (nr < 0) atomicOr(&d.g->notes, NOTE_RANKS_SATURATED);
      h[HH_SIGCHK]++;
      int pe = -1;
      for (int q = 0; q < H_PEND; q++)
        if (!(h[HH_PEND + q] & 0x80000000u)) {
          pe = q;
          break;
        }
      if (pe < 0) {
        set_err(d.g, ERR_PENDING);
        pe = 0;
      }
      h[HH_PEND + pe] = 0x80000000u | ((uint32_t)l << 8) | (uint32_t)slot;
      h[HH_PENDFROM + pe] = (uint32_t)from;
      // registerTask(updateVerifiedSignatures(best), time + nodePairingTime, this)
      const int32_t arrival = t + (int32_t)h[HH_PAIR];
      const Rec fin = make_rec(K_TASK, node, (uint32_t)node, H_TASK_UPDATE, (uint32_t)pe);
      const bool ok = arrival - t < d.horizon - 1;  // see Engine::run_ms on host-held envelopes
      if (!ok) set_err(d.g, ERR_HORIZON);
      if (SH) {
        int32_t WG_G* x = d.xbuf + (size_t)j * 5;
        x[0] = (int32_t)fin.w0;
        x[1] = (int32_t)fin.w1;
        x[2] = (int32_t)fin.w2;
        x[3] = (int32_t)fin.w3;
        x[4] = ok ? arrival + 1 : 0;
        continue;
      }
      d.fin[j] = fin;
      d.arr[j] = ok ? arrival : -1;
      if (ok) histKey = (j / TILE) * D + ((uint32_t)arrival & (D - 1));
    }
    if (SH) continue;  // (k_shard_unpack builds the tile histograms from the summed image)
    // per-tile arrival histogram of the multisplit; pairing times are nearly uniform, so aggregate equal
    // keys inside the wavefront before touching memory
    uint64_t todo = __ballot(histKey != 0xFFFFFFFFu);
    while (todo) {
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const uint32_t key = lane_bcast(histKey, leader);
      const uint64_t m = __ballot(histKey == key) & todo;
      if ((int)WG_LANE == leader) atomicAdd(&d.tileHist[key], (uint32_t)__popcll(m));
      todo &= ~m;
    }
  }
}

// ---- sharded engine: the periodic-task snapshots of this ms (Handel SendSigs.sigs, P/Handel.java:254; GSFSignature
// toSend.clone(), P/GSFSignature.java:146) reach the other shards -------------------------------------------------
// A snapshot is read at delivery by the receiver's shard at the address the message carries, so every shard keeps
// the whole snapshot ring and the rows written in this ms are summed across shards (zeros from non-owners). The
// rows are those of the periodic-task events (task word TASK) of the (replicated) event list, numbered in event
// order by SnapF. S = HandelState / GsfState: snap, snapNb, snapStride, N, snapIdx, nSnap, xsnap, xsnapRows.
__device__ __forceinline__ int32_t snap_period(const HandelState& s) { return s.p.disseminationPeriodMs; }
template <uint32_t TASK>
__device__ __forceinline__ bool is_snapshot_event(const EngineDev& d, uint32_t e) {
  const Rec r = d.ev[e];
  return rec_kind(r) == K_PERIODIC && r.w2 == TASK;
}
template <class S, uint32_t TASK>
struct SnapF {
  typedef S Aux;
  const EngineDev& d;
  const S& s;
  __device__ SnapF(const EngineDev& d_, const Aux* a) : d(d_), s(*a) {}
  __device__ uint32_t count() const { return d.g->nEvents; }
  __device__ uint64_t value(uint32_t e) const { return is_snapshot_event<TASK>(d, e); }
  __device__ void tally(uint32_t, uint32_t) const {}
  __device__ void total(uint64_t tot) const {
    if ((uint32_t)tot > s.xsnapRows) set_err(d.g, ERR_PAYLOAD);
    *s.nSnap = min((uint32_t)tot, s.xsnapRows);
  }
  __device__ void write(uint32_t e, uint64_t excl, bool valid) const {
    if (valid) s.snapIdx[e] = (uint32_t)excl;
  }
};
// one wavefront per snapshot event: pack = owner's ring row -> image (zeros elsewhere); unpack = summed
// image -> ring row on the shards that do not own the node
template <class S, uint32_t TASK, bool PACK>
__global__ void __launch_bounds__(256) k_shard_snap(const EngineDev* __restrict__ tab, const S* __restrict__ stab) {
  WG_ENGINE(tab);
  const S& s = stab[blockIdx.y];
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t nEv = d.g->nEvents;
  const uint32_t win = ((uint32_t)d.g->now / (uint32_t)snap_period(s)) % s.snapNb;
  for (uint32_t e = wave; e < nEv; e += nWaves) {
    if (!is_snapshot_event<TASK>(d, e) || s.snapIdx[e] >= s.xsnapRows) continue;
    const int32_t node = (int32_t)d.ev[e].w1;
    const bool owned = shard_owns(d, node);
    uint64_t WG_G* row = s.snap + (size_t)(win * (uint32_t)s.N + (uint32_t)node) * s.snapStride;
    uint64_t WG_G* img = (uint64_t WG_G*)(int32_t WG_G*)s.xsnap + (size_t)s.snapIdx[e] * s.snapStride;
    for (uint32_t j = WG_LANE; j < s.snapStride; j += 64) {
      if (PACK)
        img[j] = owned ? row[j] : 0ULL;
      else if (!owned)
        row[j] = img[j];
    }
  }
}

}  // namespace wg
