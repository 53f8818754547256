// GSFSignature (P/GSFSignature.java) — gossiping San Fermin — as a resident device protocol. One
// wavefront per simulated node, lanes = 64-bit words of the node's bitset rows (as proto_handel.hip.h).
//
// What makes it resident in a small state (each point is checked against the oracle by tests/):
//  * SFLevel l (l >= 1) waits for the node's aligned sibling block of 2^(l-1) ids (allSigsAtLevel minus
//    the previous levels, :274-283, :362-375); the blocks of different levels are disjoint, so ONE row
//    per kind holds every level: V (SFLevel.verifiedSignatures; its union over levels is exactly
//    GSFNode.verifiedSignatures :171 — both are updated by the same two statements :423-427), IS
//    (individualSignatures), IV (indivVerifiedSig). Bit `node` of V is level 0.
//  * A SendSigs payload (`toSend` of doCycle :217-224, `bestToSend` of the accelerated calls :433-443) is
//    getLastFinishedLevel() | verified(levels below l). With k = first incomplete level of the sender
//    that is: k > l  the sender's aligned block of 2^(k-1) ids  ("OVER", strictly larger than the level)
//             k == l the receiver's whole level block            ("FULL")
//             k < l  the sender's verified bits of levels < l    ("PARTIAL": the only kind that needs
//                    payload words; snapshot at send, private copy at delivery).
//    updateVerifiedSignatures' in-place `sigs.or(...)` (:390, :419) mutates the message object that a
//    multi-destination send shares between receivers; such messages are always FULL or OVER (accelerated
//    calls are only made for completed levels), OR-ing subsets of the level block into them changes
//    nothing, and every PARTIAL message has one receiver — so the aliasing is unobservable and a private
//    copy per receiver is exact (tests/test_oracle_protocols.py::test_gsf_aliasing_is_unobservable).
//  * toVerify (:167) is one list per node in arrival order: tvEnt[node][Q] entries
//    {from, level, kind, slot | j}; PARTIAL entries own one of the node's Q payload slots.
// N must be a power of two (the reference's own non-power-of-two test is disabled, PT/GSFSignatureTest.java:60-70).
#pragma once
#include "proto_handel.hip.h"

namespace wg {

constexpr int G_PEND = 4;  // outstanding updateVerifiedSignatures tasks per node
constexpr int G_MAX_Q = 512;
constexpr uint32_t G_TASK_DOCYCLE = 0;
constexpr uint32_t G_TASK_UPDATE = 1;
enum GsfKind : uint32_t { GK_PARTIAL = 0, GK_FULL = 1, GK_OVER = 2, GK_INDIV = 3 };

struct GsfState {
  wg_gsf_params p;
  int32_t N, L, W, Q, SW;             // SW = payload slot size in 64-bit words (largest level block)
  GP<uint64_t> V, IS, IV;              // [N][W]
  GP<int32_t> peers;                     // [N][N-1] SFLevel.peers, level l at [2^(l-1)-1, 2^l-1)
  GP<int32_t> pairing, sigChecked, sigQueueSize, tvLen;  // [N]
  GP<int32_t> ctMinStart;                // ConditionalTask.minStartTime
  GP<uint32_t> ctEpoch;                  // epoch in which the task left nextMessage()'s private copy
  GP<int32_t> pos, rem, cV, cIV, cU; // [N][L] posInLevel, remainingCalls, |V|, |IV|, |V u IV| inside the level block
  GP<uint64_t> tvEnt;                    // [N][Q] toVerify, list order
  GP<uint64_t> tvUsed;                   // [N][Q/64] payload slots in use
  GP<uint64_t> tvSig;                    // [N][Q][SW]
  // doCycle snapshots (SendSigs.sigs = toSend.clone() :146): one doCycle per node and period window, so the
  // snapshot of level l lives at snap[((t / period) % snapNb) * N + node][lvlOff[l] ...] (no allocation)
  GP<uint64_t> snap;
  uint32_t snapNb, snapStride;
  uint32_t lvlOff[MAX_LEVELS];
  GP<uint32_t> pend;                     // [N][G_PEND] valid<<31 | kind<<29 | level<<24 | slot or j
  GP<int32_t> pendFrom;                  // [N][G_PEND]
  // conditional-task phase scratch
  GP<uint32_t> runList;                  // nodes whose checkSigs runs at this edge (unordered)
  GP<uint32_t> runCount;
  GP<uint32_t> runList2;                 // ... those of them k_gsf_cond_a1g leaves to the wavefront-per-runner kernel
  GP<uint32_t> runCount2;
  GP<uint8_t> candFlag;                  // [N] checkSigs found a best -> registers a task
  GP<uint8_t> candPend;                  // [N] its pend index
  GP<uint32_t> condList;                 // registering nodes in id order
  // node-range sharding (wg_shard_configure): the rows V / IS / IV, peers and the toVerify storage are held for
  // the nodes [lo, hi) only (pointers biased by -lo rows); 0 / N when not sharded. Snapshot exchange as Handel's.
  int32_t lo, hi;
  GP<uint32_t> snapIdx;
  GP<uint32_t> nSnap;
  GP<int32_t> xsnap;
  uint32_t xsnapRows;
};
__device__ __forceinline__ int32_t snap_period(const GsfState& s) { return s.p.periodDurationMs; }

__device__ __forceinline__ uint64_t g_ent(int32_t from, int l, uint32_t kind, uint32_t aux) {
  return (uint64_t)(uint32_t)from | ((uint64_t)l << 24) | ((uint64_t)kind << 29) | ((uint64_t)aux << 32);
}
__device__ __forceinline__ int32_t g_ent_from(uint64_t e) { return (int32_t)(e & 0xFFFFFFu); }
__device__ __forceinline__ int g_ent_level(uint64_t e) { return (int)((e >> 24) & 31u); }
__device__ __forceinline__ uint32_t g_ent_kind(uint64_t e) { return (uint32_t)((e >> 29) & 3u); }
__device__ __forceinline__ uint32_t g_ent_aux(uint64_t e) { return (uint32_t)(e >> 32) & 0xFFFFu; }
__device__ __forceinline__ int g_msg_size(int l) { return 1 + ((l == 0 ? 1 : (1 << (l - 1))) / 8) + 96; }  // :150

// evaluateSig's tail (:519-534) given the totals its three branches compute
__device__ __forceinline__ int g_score(int l, int size, int newTotal, int added, bool single, bool interIV) {
  if (added <= 0) return (single && !interIV) ? 1 : 0;
  if (newTotal == size) return 1000000 - l * 10;
  return 100000 - l * 100 + added;
}

// per-wave LDS mirror of the (node, level) scalars and the slot bitmap
struct GLevels {
  int32_t pos[MAX_LEVELS];
  int32_t rem[MAX_LEVELS];
  int32_t cV[MAX_LEVELS];
  int32_t cIV[MAX_LEVELS];
  int32_t cU[MAX_LEVELS];
  unsigned long long used[G_MAX_Q / 64];
};

struct GsfProto {
  typedef GsfState State;
  typedef GLevels WaveShared;
  struct NodeRegs {
    long long doneAt;
    int32_t sigQueueSize, tvLen;
    uint32_t pend[G_PEND];
    int32_t pendFrom[G_PEND];
    GLevels* ls;
    // what the visit's events changed of the node's arrays (1: the per-level scalars, 2: the slot bitmap, 4: pend, 8: doneAt):
    // node_end writes back only that — a SendSigs delivery touches none of the five per-level arrays
    uint32_t dirty;
    long long doneAt0;
  };

  __device__ static int msg_size(const State&, uint32_t msg) { return g_msg_size((int)(msg & 31u)); }
  __device__ static int msg_level(uint32_t msg) { return (int)(msg & 31u); }

  __device__ static void load_levels(const State& s, int32_t node, GLevels* ls) {
    for (int l = WG_LANE; l < s.L; l += 64) {
      size_t i = (size_t)node * s.L + l;
      ls->pos[l] = s.pos[i];
      ls->rem[l] = s.rem[i];
      ls->cV[l] = s.cV[i];
      ls->cIV[l] = s.cIV[i];
      ls->cU[l] = s.cU[i];
    }
    for (int q = WG_LANE; q < s.Q / 64; q += 64) ls->used[q] = s.tvUsed[(size_t)node * (s.Q / 64) + q];
    __builtin_amdgcn_wave_barrier();
  }
  __device__ static void store_levels(const State& s, int32_t node, const GLevels* ls, uint32_t dirty = 3u) {
    __builtin_amdgcn_wave_barrier();
    if (dirty & 1u)
      for (int l = WG_LANE; l < s.L; l += 64) {
        size_t i = (size_t)node * s.L + l;
        s.pos[i] = ls->pos[l];
        s.rem[i] = ls->rem[l];
        s.cV[i] = ls->cV[l];
        s.cIV[i] = ls->cIV[l];
        s.cU[i] = ls->cU[l];
      }
    if (dirty & 2u)
      for (int q = WG_LANE; q < s.Q / 64; q += 64) s.tvUsed[(size_t)node * (s.Q / 64) + q] = ls->used[q];
  }
  __device__ static void node_begin(Ctx& c, const State& s, NodeRegs& r, GLevels* ls) {
    const int32_t node = c.node;
    r.doneAt = r.doneAt0 = c.d.nodes.doneAt[node];
    r.dirty = 0;
    r.sigQueueSize = s.sigQueueSize[node];
    r.tvLen = s.tvLen[node];
#pragma unroll
    for (int k = 0; k < G_PEND; k++) {
      r.pend[k] = s.pend[(size_t)node * G_PEND + k];
      r.pendFrom[k] = s.pendFrom[(size_t)node * G_PEND + k];
    }
    r.ls = ls;
    load_levels(s, node, ls);
  }
  __device__ static void node_end(Ctx& c, const State& s, NodeRegs& r) {
    const int32_t node = c.node;
    store_levels(s, node, r.ls, r.dirty);
    if (WG_LANE == 0) {
      if (r.doneAt != r.doneAt0) c.d.nodes.doneAt[node] = r.doneAt;
      s.sigQueueSize[node] = r.sigQueueSize;
      s.tvLen[node] = r.tvLen;
      if (r.dirty & 4u) {
#pragma unroll
        for (int k = 0; k < G_PEND; k++) s.pend[(size_t)node * G_PEND + k] = r.pend[k];
      }
    }
  }
  __device__ static void on_message(Ctx& c, const State& s, NodeRegs& r, int32_t from, uint32_t msg, uint32_t payload) {
    on_new_sig(c, s, r, from, msg, payload);
  }
  __device__ static void on_task(Ctx& c, const State& s, NodeRegs& r, uint32_t word, uint32_t arg) {
    if (word == G_TASK_DOCYCLE)
      do_cycle(c, s, r);
    else
      update_verified(c, s, r, arg);
  }

  __device__ static uint64_t WG_G* sig_ptr(const State& s, int32_t node, int slot) {
    return s.tvSig + ((size_t)node * s.Q + slot) * (size_t)s.SW;
  }
  // first incomplete level (levels below it form getLastFinishedLevel :194-211); L if every level is complete
  __device__ static int first_incomplete(const State& s, const GLevels* ls) {  // (every lane of the wavefront calls it: lane l looks at level l)
    const int lane = WG_LANE;
    const bool inc = lane >= 1 && lane < s.L && ls->cV[lane] != (1 << (lane - 1));
    const uint64_t m = __ballot(inc);
    return m ? __ffsll((unsigned long long)m) - 1 : s.L;
  }
  // message word of a SendSigs built at level l while the sender's first incomplete level is k (header comment)
  __device__ static uint32_t msg_word(int l, int k, bool levelFinished) {
    uint32_t kind = k > l ? GK_OVER : (k == l ? GK_FULL : GK_PARTIAL);
    return (uint32_t)l | (levelFinished ? 32u : 0u) | (kind << 6) | ((uint32_t)(k - 1) << 8);
  }

  // ---- Message.action: SendSigs -> onNewSig (:538-556) ----------------------------------------------
  __device__ static void on_new_sig(Ctx& c, const State& s, NodeRegs& r, int32_t from, uint32_t msg, uint32_t payload) {
    KPROF_COUNT(c.d.g, 9);
    const int l = (int)(msg & 31u);
    const uint32_t kind = (msg >> 6) & 3u;
    uint32_t aux = (msg >> 8) & 31u;
    const int32_t node = c.node;
    GLevels* ls = r.ls;
    uint64_t WG_G* isRow = s.IS + (size_t)node * s.W;
    const bool hadIS = row_get(isRow, from);
    const int len = r.tvLen;
    const int need = hadIS ? 1 : 2;
    if (len + need > s.Q) {
      if (WG_LANE == 0) set_err(c.d.g, ERR_QUEUE_CAP);
      return;
    }
    if (kind == GK_PARTIAL) {
      int slot = -1;
      for (int q = 0; q < s.Q / 64 && slot < 0; q++) {
        const unsigned long long u = ls->used[q];
        if (~u) slot = q * 64 + __ffsll(~u) - 1;
      }
      if (slot < 0) {
        if (WG_LANE == 0) set_err(c.d.g, ERR_QUEUE_CAP);
        return;
      }
      const Lv v = sib_view(node, l);
      const uint64_t WG_G* src = s.snap + payload;
      uint64_t WG_G* dst = sig_ptr(s, node, slot);
      H_FOR_WORDS(v, j) dst[j] = src[j];
      __builtin_amdgcn_wave_barrier();  // every lane has read the slot bitmap before lane 0 changes it
      if (WG_LANE == 0) ls->used[slot >> 6] |= 1ULL << (slot & 63);
      r.dirty |= 2u;
      aux = (uint32_t)slot;
    }
    if (WG_LANE == 0) {
      uint64_t WG_G* ent = s.tvEnt + (size_t)node * s.Q;
      ent[len] = g_ent(from, l, kind, aux);                         // toVerify.add(ssigs)
      if (!hadIS) ent[len + 1] = g_ent(from, l, GK_INDIV, 0);       // the individual signature (:547-553)
    }
    if (!hadIS) row_set(isRow, from, true);
    r.tvLen = len + need;
    r.sigQueueSize = r.tvLen;
    __builtin_amdgcn_wave_barrier();
  }

  // ---- PeriodicTask: doCycle (:213-225) -> SFLevel.doCycle (:317-327) -------------------------------
  __device__ static void do_cycle(Ctx& c, const State& s, NodeRegs& r) {
    // The reference walks the levels one after the other (:317-327 inside :213-225); done literally that is one chain of
    // dependent round trips per level (posInLevel -> the peer's id -> the record, and the snapshot's words for a PARTIAL
    // payload) — twelve of them at 4096 nodes, in the ms in which every node has this task. Here lane l takes level l: the
    // peers are one round of loads, the snapshots of all PARTIAL levels one flat copy (the levels' words lie back to back in
    // the node's snapshot row, lvlOff), the records one send_many in level order.
    GLevels* ls = r.ls;
    r.dirty |= 1u;  // (posInLevel / remainingCalls)
    const int32_t node = c.node;
    const int lane = WG_LANE;
    const int k = first_incomplete(s, ls);
    const bool lv = lane >= 1 && lane < s.L;
    int rem = 0, pos = 0, cvl = 0;
    if (lv) {
      rem = ls->rem[lane];
      pos = ls->pos[lane];
      cvl = ls->cV[lane];
    }
    const int size = lv ? 1 << (lane - 1) : 0;
    const bool act = lv && rem != 0 && (c.t >= lane * s.p.timeoutPerLevelMs || k >= lane);  // hasStarted :294-315
    int32_t dest = 0;
    if (act) dest = s.peers[(size_t)node * (s.N - 1) + (size - 1) + pos];  // getRemainingPeers(1)
    const uint64_t actM = __ballot(act);
    if (!actM) return;
    const uint64_t partM = __ballot(act && k < lane);  // PARTIAL: snapshot the verified bits of the levels below l (the node's own block)
    const uint32_t win = ((uint32_t)c.t / (uint32_t)s.p.periodDurationMs) % s.snapNb;
    const uint32_t refBase = (win * (uint32_t)s.N + (uint32_t)node) * s.snapStride;
    if (partM) {
      const uint64_t WG_G* vr = s.V + (size_t)node * s.W;
      for (uint32_t q = (uint32_t)lane; q < s.snapStride; q += 64) {
        int l = 1;  // the level whose words hold flat word q (lvlOff is increasing from level 1 on)
        for (int i = 2; i < s.L; i++)
          if (s.lvlOff[i] <= q) l = i;
        if ((partM >> l) & 1ULL) {
          const Lv v = own_view(node, l);
          s.snap[refBase + q] = vr[v.bw + (int)(q - s.lvlOff[l])] & v.mask;
        }
      }
      c.evFlags = ev_snap_code(s.snapStride) << EV_SNAP_SHIFT;  // (a sharded engine ships the node's row of this window)
    }
    long long bytes = 0;
    for (uint64_t m = actM; m; m &= m - 1) bytes += g_msg_size(__ffsll((unsigned long long)m) - 1);
    c.send_many(act, __popcll(actM & lanes_lt()), __popcll(actM), dest, msg_word(lane, k, cvl == size),
                act && k < lane ? refBase + s.lvlOff[lane] : 0u, bytes);
    if (act) {
      ls->pos[lane] = pos + 1 >= size ? 0 : pos + 1;
      ls->rem[lane] = rem - 1;
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- Task: updateVerifiedSignatures (:387-460) ------------------------------------------------------
  __device__ static void update_verified(Ctx& c, const State& s, NodeRegs& r, uint32_t arg) {
    const int lane = WG_LANE;
    KPROF_DECL;
    KPROF_COUNT(c.d.g, 11);
    r.dirty |= 1u | 2u | 4u;  // (the level's counts, a PARTIAL message's slot, the pending table)
    const int32_t node = c.node;
    // (selects, not r.pend[arg & 3]: a register array indexed at run time puts the whole NodeRegs — and the LDS pointer in
    // it — in scratch memory; every access to a node scalar is then a memory round trip and every LDS access a FLAT one)
    const int pk = (int)(arg & (G_PEND - 1));
    uint32_t pe = r.pend[0];
    int32_t from = r.pendFrom[0];
#pragma unroll
    for (int k = 1; k < G_PEND; k++) {
      pe = pk == k ? r.pend[k] : pe;
      from = pk == k ? r.pendFrom[k] : from;
    }
    if (!(pe & 0x80000000u)) {
      if (lane == 0) set_err(c.d.g, ERR_PROTOCOL);
      return;
    }
#pragma unroll
    for (int k = 0; k < G_PEND; k++) r.pend[k] = pk == k ? 0u : r.pend[k];
    const uint32_t kind = (pe >> 29) & 3u;
    const int l = (int)((pe >> 24) & 31u);
    const uint32_t aux = pe & 0xFFFFu;
    GLevels* ls = r.ls;
    const Lv v = sib_view(node, l);
    const int size = v.size;
    uint64_t WG_G* vr = s.V + (size_t)node * s.W;
    uint64_t WG_G* ivr = s.IV + (size_t)node * s.W;
    const uint64_t WG_G* sig = kind == GK_PARTIAL ? sig_ptr(s, node, (int)aux) : nullptr;
    const int wF = from >> 6, jF = wF - v.bw;
    const uint64_t bitF = 1ULL << (from & 63);
    // word j of the signature set, as held by the lane that owns row word v.bw + j
    auto sword = [&](int j) -> uint64_t {
      if (kind == GK_PARTIAL) return sig[j] & v.mask;
      if (kind == GK_INDIV) return j == jF ? bitF : 0ULL;
      return v.mask;  // FULL (and OVER after `sigs = waitedSigs.clone()` :412)
    };
    int cVl = ls->cV[l], cIVl = ls->cIV[l], cUl = ls->cU[l];
    bool reset = false, improved = false;
    if (kind != GK_OVER && v.nw <= 64) {
      // ---- the usual case, one word of each row per lane (blocks of <= 4096 ids): the signature's, indivVerifiedSig's and
      // verifiedSignatures' words of the level in ONE round of loads — taken one after the other as the statements of
      // :388-427 come (the cardinality, the sender's two bits, the merge test, the replacement) they were four dependent
      // round trips of a visit that is nothing but a chain of those
      const bool in = lane < v.nw;
      const uint64_t sgRaw = (kind == GK_PARTIAL && in) ? sig[lane] : 0ULL;
      uint64_t ivw = in ? ivr[v.bw + lane] : 0ULL;
      const uint64_t vw = in ? vr[v.bw + lane] : 0ULL;
      const uint64_t sg = !in ? 0ULL : kind == GK_PARTIAL ? (sgRaw & v.mask) : kind == GK_INDIV ? (lane == jF ? bitF : 0ULL) : v.mask;
      const int cs = (int)wave_sum64((uint64_t)__popcll(sg));
      if (cs == 1) {  // sigs.cardinality() == 1 -> indivVerifiedSig.set(from) (:388-390)
        const uint64_t ivF = lane_bcast64(ivw, jF), vF = lane_bcast64(vw, jF);
        if (!(ivF & bitF)) {
          if (lane == jF) {
            ivw |= bitF;
            ivr[wF] = ivw;
          }
          cIVl++;
          if (!(vF & bitF)) cUl++;
        }
      }
      // sigs |= indivVerifiedSig; merge with the level's verified set when disjoint (:390, :415-420)
      const uint64_t sgI = sg | (ivw & v.mask), vwm = vw & v.mask;
      const uint64_t acc = wave_sum64((uint64_t)__popcll(sgI) | ((uint64_t)__popcll(sgI | vwm) << 24));
      const bool inter = __ballot((sgI & vwm) != 0) != 0;
      const bool merge = cVl > 0 && !inter;
      const int cFinal = merge ? (int)(acc >> 24) : (int)(acc & 0xFFFFFFu);
      improved = cFinal > cVl;
      if (improved) {  // verifiedSignatures.andNot(waitedSigs); verifiedSignatures.or(sigs) (:423-427)
        if (in) vr[v.bw + lane] = (vw & ~v.mask) | sgI | (merge ? vwm : 0ULL);
        cVl = cFinal;
        cUl = cFinal;  // the new set includes indivVerifiedSig
      }
    } else {
      if (kind == GK_OVER) {
        // the sender sent its next levels too (:397-413): every level i <= j whose block the set includes is
        // completed outright
        const int jTop = min((int)aux, s.L - 1);
        uint32_t incomplete = 0;
        for (int i = 1; i <= jTop; i++)
          if (ls->cV[i] != (1 << (i - 1))) incomplete |= 1u << i;
        __builtin_amdgcn_wave_barrier();  // every lane has read the counts before lane 0 replaces them
        if (incomplete) {
          reset = true;  // resetRemaining stays true from the first completed level on (:404-408)
          for (int i = __ffs(incomplete) - 1; i <= jTop; i++) {
            const int sz = 1 << (i - 1);
            if ((incomplete >> i) & 1u) {
              const Lv vi = sib_view(node, i);
              H_FOR_WORDS(vi, j) vr[vi.bw + j] |= vi.mask;
              if (lane == 0) {
                ls->cV[i] = sz;
                ls->cU[i] = sz;
              }
            }
            if (lane == 0) ls->rem[i] = sz;
          }
        }
        __builtin_amdgcn_wave_barrier();
        cVl = ls->cV[l];
        cUl = ls->cU[l];
      } else {
        // sigs.cardinality() == 1 -> indivVerifiedSig.set(from) (:388-390)
        uint64_t a = 0;
        H_FOR_WORDS(v, j) a += (uint64_t)__popcll(sword(j));
        const int cs = (int)wave_sum64(a);
        if (cs == 1) {
          const uint64_t ivF = ld_coherent(ivr + wF), vF = ld_coherent(vr + wF);
          if (!(ivF & bitF)) {
            if (lane == (wF & 63)) ivr[wF] = ivF | bitF;
            cIVl++;
            if (!(vF & bitF)) cUl++;
          }
        }
      }
      // sigs |= indivVerifiedSig; merge with the level's verified set when disjoint (:390, :415-420)
      uint64_t acc = 0, flg = 0;
      H_FOR_WORDS(v, j) {
        const uint64_t sg = sword(j) | (ivr[v.bw + j] & v.mask), vw = vr[v.bw + j] & v.mask;
        acc += (uint64_t)__popcll(sg) | ((uint64_t)__popcll(sg | vw) << 24);
        flg |= (uint64_t)((sg & vw) != 0);
      }
      acc = wave_sum64(acc);
      const bool inter = __ballot(flg != 0) != 0;
      const bool merge = cVl > 0 && !inter;
      const int cFinal = merge ? (int)(acc >> 24) : (int)(acc & 0xFFFFFFu);
      improved = cFinal > cVl || reset;
      if (improved) {
        // verifiedSignatures.andNot(waitedSigs); verifiedSignatures.or(sigs) — level and node rows are one (:423-427)
        H_FOR_WORDS(v, j) {
          const uint64_t old = vr[v.bw + j];
          const uint64_t nv = sword(j) | (ivr[v.bw + j] & v.mask) | (merge ? (old & v.mask) : 0ULL);
          vr[v.bw + j] = (old & ~v.mask) | nv;
        }
        cVl = cFinal;
        cUl = cFinal;  // the new set includes indivVerifiedSig
      }
    }
    KPROF_MARK(c.d.g, 4);  // update: the level's rows (loads, counts, stores)
    if (lane == 0) {
      ls->cV[l] = cVl;
      ls->cIV[l] = cIVl;
      ls->cU[l] = cUl;
      if (kind == GK_PARTIAL) ls->used[aux >> 6] &= ~(1ULL << (aux & 63));  // the message object dies with the task
    }
    if (improved && lane >= l && lane < s.L) ls->rem[lane] = 1 << (lane - 1);  // :421-425 (L <= 64 lanes)
    __builtin_amdgcn_wave_barrier();
    KPROF_MARK(c.d.g, 5);  // update: the level's scalars
    if (!improved) return;
    KPROF_COUNT(c.d.g, 12);
    if (s.p.acceleratedCallsCount > 0) {  // :429-444
      const int k = first_incomplete(s, ls);
      int cur = l;
      while (cur <= k - 1 && cur < s.L - 1) {
        cur++;
        const int sz = 1 << (cur - 1);
        const int rem = ls->rem[cur], p0 = ls->pos[cur];
        const int n = min(s.p.acceleratedCallsCount, rem);  // getRemainingPeers(acceleratedCallsCount) :329-353
        if (n > 0) {
          __threadfence_block();
          const uint32_t destOff = c.dest_reserve(n);
          if (lane < n) c.dest_put(destOff, lane, s.peers[(size_t)node * (s.N - 1) + (sz - 1) + (p0 + lane) % sz]);
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) {
            ls->pos[cur] = (p0 + n) % sz;
            ls->rem[cur] = rem - n;
          }
          __threadfence_block();
          c.send_list(destOff, n, msg_word(cur, k, ls->cV[cur] == sz), 0, g_msg_size(cur));
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    KPROF_MARK(c.d.g, 6);  // update: the accelerated calls
    if (r.doneAt == 0) {
      const int tot = (int)wave_sum64(lane < s.L ? (uint64_t)ls->cV[lane] : 0ULL);
      if (tot >= s.p.threshold) r.doneAt = c.t;
    }
    KPROF_MARK(c.d.g, 7);  // update: doneAt
  }
};

// ---- conditional-task phase (C/Network.java:543-566 driving GSFNode.checkSigs :558-584) --------------
__global__ void __launch_bounds__(256) k_gsf_cond_pre(const EngineDev* __restrict__ tab, const GsfState* __restrict__ stab) {
  WG_ENGINE(tab);
  const GsfState& s = stab[wgBy];
  const int32_t t = d.g->now, until = d.g->until;
  const uint32_t epoch = d.g->epoch;
  const uint32_t stride = wgGx * blockDim.x;
  for (uint32_t n0 = (uint32_t)s.lo + wgBx * blockDim.x; n0 < (uint32_t)s.hi; n0 += stride) {
    const uint32_t node = n0 + threadIdx.x;
    bool run = false;
    if (node < (uint32_t)s.hi) {
      if (!d.nodes.down[node] && s.ctEpoch[node] != epoch) {
        const int32_t ms = s.ctMinStart[node];
        if (ms <= until && ms <= t) {
          s.ctEpoch[node] = epoch;
          run = s.tvLen[node] > 0;  // startIf = !toVerify.isEmpty() (:632)
        }
      }
      if (run) s.ctMinStart[node] = t + s.pairing[node];  // minStartTime = time + duration
      s.candFlag[node] = 0;
    }
    const uint64_t m = __ballot(run);
    if (m) {
      uint32_t base = 0;
      const int leader = __ffsll((unsigned long long)m) - 1;
      if ((int)WG_LANE == leader) base = atomicAdd(F(s.runCount + 0), (uint32_t)__popcll(m));
      base = __shfl(base, leader, 64);
      if (run) s.runList[base + __popcll(m & lanes_lt())] = node;
    }
  }
}

// checkSigs (:558-584) of the runners whose list holds at most eight entries — 3.9 on average (profiles/r16d) — none of
// them a PARTIAL signature of a multi-word level: EIGHT LANES per runner, lane g = entry g; the level's three counts come
// from memory per entry (no LDS image), maximum and compaction inside the group. One wavefront per runner (k_gsf_cond_a1)
// ran its 64 lanes for four entries and its duration was one runner's chain of round trips; the runners this kernel cannot
// take go to that kernel through runList2. Same statements, same order of the entries, same results.
__global__ void __launch_bounds__(256) k_gsf_cond_a1g(const EngineDev* __restrict__ tab, const GsfState* __restrict__ stab) {
  WG_ENGINE(tab);
  const GsfState& s = stab[wgBy];
  const int lane = WG_LANE, g8 = lane & 7, gsh = lane & ~7;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nRun = *s.runCount;
  for (uint32_t qb = wave * 8; qb < nRun; qb += nWaves * 8) {
    const uint32_t q = qb + (uint32_t)(lane >> 3);
    const bool have = q < nRun;
    const int32_t node = have ? (int32_t)s.runList[q] : 0;
    const int len = have ? s.tvLen[node] : 0;
    bool ok = have && len <= 8;
    const bool mine = ok && g8 < len;
    uint64_t WG_G* ent = s.tvEnt + (size_t)node * s.Q;
    const uint64_t e = mine ? ent[g8] : 0ULL;
    const int32_t from = g_ent_from(e);
    const int l = mine ? g_ent_level(e) : 1;
    const uint32_t kind = g_ent_kind(e), aux = g_ent_aux(e);
    const int size = 1 << (l - 1);
    const size_t li = (size_t)node * s.L + l;
    const int cVl = mine ? s.cV[li] : 0, cIVl = mine ? s.cIV[li] : 0, cUl = mine ? s.cU[li] : 0;
    const Lv v = sib_view(node, l);
    const uint64_t WG_G* vr = s.V + (size_t)node * s.W;
    const uint64_t WG_G* ivr = s.IV + (size_t)node * s.W;
    const bool open = mine && cVl < size;  // :490-492
    // a PARTIAL entry of a multi-word level: the whole runner is the wavefront kernel's
    const uint32_t wideG = (uint32_t)(__ballot(open && kind == GK_PARTIAL && v.nw > 1) >> gsh) & 0xFFu;
    if (wideG) ok = false;
    {  // the runners left over: one atomic per wavefront
      const bool left = have && !ok && g8 == 0;
      const uint64_t lm = __ballot(left);
      if (lm) {
        uint32_t base = 0;
        const int leader = __ffsll((unsigned long long)lm) - 1;
        if (lane == leader) base = atomicAdd(F(s.runCount2 + 0), (uint32_t)__popcll(lm));
        base = __shfl(base, leader, 64);
        if (left) s.runList2[base + __popcll(lm & lanes_lt())] = (uint32_t)node;
      }
    }
    int ns = 0;
    if (ok && open) {
      if (kind == GK_FULL) {
        ns = 1000000 - l * 10;  // completes the level
      } else if (kind == GK_OVER) {
        ns = 100000 - l * 100 + ((1 << aux) - cVl);  // 2^j ids, a replace or a first set
      } else if (kind == GK_INDIV) {
        const uint64_t bit = 1ULL << (from & 63);
        const bool vHas = (vr[from >> 6] & bit) != 0, ivHas = (ivr[from >> 6] & bit) != 0;
        int newTotal, added;
        if (cVl == 0) {
          newTotal = 1;
          added = 1;
        } else if (vHas) {
          newTotal = cIVl + (ivHas ? 0 : 1);
          added = newTotal - cVl;
        } else {
          newTotal = cUl + (ivHas ? 0 : 1);
          added = newTotal - cVl;
        }
        ns = g_score(l, size, newTotal, added, true, ivHas);
      } else {  // PARTIAL, one word
        const uint64_t sg = *GsfProto::sig_ptr(s, node, (int)aux) & v.mask;
        const uint64_t vw = vr[v.bw] & v.mask, iw = ivr[v.bw] & v.mask;
        const int cs = __popcll(sg);
        int newTotal;
        if (cVl == 0)
          newTotal = cs;
        else if (sg & vw)
          newTotal = __popcll(sg | iw);
        else
          newTotal = __popcll(sg | iw | vw);
        ns = g_score(l, size, newTotal, cVl == 0 ? cs : newTotal - cVl, cs == 1, (sg & iw) != 0);
      }
    }
    // `ns > score` from zero on: the first entry reaching the maximum wins (:565-570)
    const int mx = group8_max_i32(ns);
    const uint32_t bm = (uint32_t)(__ballot(ok && mine && mx > 0 && ns == mx) >> gsh) & 0xFFu;
    const int bestIdx = bm ? __ffs(bm) - 1 : -1;
    const uint64_t bestEnt = shfl64(e, gsh + (bestIdx < 0 ? 0 : bestIdx));
    // rewrite the list: zeros are removed (:571-573), the best leaves it (:577)
    const bool keep = ok && mine && ns != 0 && g8 != bestIdx;
    const uint32_t km = (uint32_t)(__ballot(keep) >> gsh) & 0xFFu;
    if (ok && mine && ns == 0 && kind == GK_PARTIAL)  // its payload slot is free again
      atomicAnd((unsigned long long*)F(s.tvUsed + (size_t)node * (s.Q / 64) + (aux >> 6)), ~(1ULL << (aux & 63)));
    if (keep) ent[__popc(km & ((1u << g8) - 1u))] = e;  // (every lane of the group read its entry above)
    const int newLen = __popc(km);
    if (ok && g8 == 0) {
      s.tvLen[node] = newLen;
      if (bestIdx >= 0) {
        // sigChecked++; sigQueueSize = toVerify.size(); registerTask(updateVerifiedSignatures(best), ...) :576-583.
        // The task's closure (tBest) is the pend entry; k_gsf_cond_a2 emits the task record in node order.
        const U4 pd = gld((const U4 WG_G*)(s.pend + (size_t)node * G_PEND));
        int pe = !(pd.x & 0x80000000u) ? 0 : !(pd.y & 0x80000000u) ? 1 : !(pd.z & 0x80000000u) ? 2 : !(pd.w & 0x80000000u) ? 3 : -1;
        if (pe < 0) {
          set_err(d.g, ERR_PENDING);
          pe = 0;
        }
        s.pend[(size_t)node * G_PEND + pe] = 0x80000000u | (g_ent_kind(bestEnt) << 29) | ((uint32_t)g_ent_level(bestEnt) << 24) | g_ent_aux(bestEnt);
        s.pendFrom[(size_t)node * G_PEND + pe] = g_ent_from(bestEnt);
        s.candPend[node] = (uint8_t)pe;
        s.sigChecked[node]++;
        s.sigQueueSize[node] = newLen;
        s.candFlag[node] = 1;
      }
    }
  }
}

// checkSigs (:558-584) of every runner: score every toVerify entry (evaluateSig :482-535), drop the zeros,
// take the FIRST entry with the greatest score out of the list and park it in the pend table.
__global__ void __launch_bounds__(256) k_gsf_cond_a1(const EngineDev* __restrict__ tab, const GsfState* __restrict__ stab, int second) {
  WG_ENGINE(tab);
  const GsfState& s = stab[wgBy];
  __shared__ GLevels shLevels[4];
  __shared__ int32_t shNs[4][G_MAX_Q];
  const int lane = WG_LANE, w = threadIdx.x >> 6;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  // (second: the runners k_gsf_cond_a1g left — a list of more than eight entries, a PARTIAL entry of a multi-word level)
  const uint32_t nRun = second ? *s.runCount2 : *s.runCount;
  const uint32_t WG_G* runList = second ? (const uint32_t WG_G*)s.runList2 : (const uint32_t WG_G*)s.runList;
  GLevels* ls = &shLevels[w];
  for (uint32_t q = wave; q < nRun; q += nWaves) {
    const int32_t node = (int32_t)runList[q];
    KPROF_DECL;
    KPROF_COUNT(d.g, 20);
    GsfProto::load_levels(s, node, ls);
    KPROF_MARK(d.g, 16);  // checkSigs: the level scalars
    const uint64_t WG_G* vr = s.V + (size_t)node * s.W;
    const uint64_t WG_G* ivr = s.IV + (size_t)node * s.W;
    uint64_t WG_G* ent = s.tvEnt + (size_t)node * s.Q;
    const int len = s.tvLen[node];
    int bestScore = 0, bestIdx = -1;
    uint64_t bestEnt = 0;
    for (int base = 0; base < len; base += 64) {
      const int i = base + lane;
      const bool mine = i < len;
      const uint64_t e = mine ? ent[i] : 0ULL;
      const int32_t from = g_ent_from(e);
      const int l = mine ? g_ent_level(e) : 1;
      const uint32_t kind = g_ent_kind(e), aux = g_ent_aux(e);
      const int size = 1 << (l - 1);
      const int cVl = ls->cV[l];
      const Lv v = sib_view(node, l);
      int ns = 0;
      bool wide = false;
      if (mine && cVl < size) {  // :490-492
        if (kind == GK_FULL) {
          ns = 1000000 - l * 10;  // completes the level
        } else if (kind == GK_OVER) {
          ns = 100000 - l * 100 + ((1 << aux) - cVl);  // 2^j ids, a replace or a first set
        } else if (kind == GK_INDIV) {
          const uint64_t bit = 1ULL << (from & 63);
          const bool vHas = (vr[from >> 6] & bit) != 0, ivHas = (ivr[from >> 6] & bit) != 0;
          int newTotal, added;
          if (cVl == 0) {
            newTotal = 1;
            added = 1;
          } else if (vHas) {
            newTotal = ls->cIV[l] + (ivHas ? 0 : 1);
            added = newTotal - cVl;
          } else {
            newTotal = ls->cU[l] + (ivHas ? 0 : 1);
            added = newTotal - cVl;
          }
          ns = g_score(l, size, newTotal, added, true, ivHas);
        } else if (v.nw == 1) {
          const uint64_t sg = *GsfProto::sig_ptr(s, node, (int)aux) & v.mask;
          const uint64_t vw = vr[v.bw] & v.mask, iw = ivr[v.bw] & v.mask;
          const int cs = __popcll(sg);
          int newTotal;
          if (cVl == 0)
            newTotal = cs;
          else if (sg & vw)
            newTotal = __popcll(sg | iw);
          else
            newTotal = __popcll(sg | iw | vw);
          ns = g_score(l, size, newTotal, cVl == 0 ? cs : newTotal - cVl, cs == 1, (sg & iw) != 0);
        } else {
          wide = true;
        }
      }
      // PARTIAL entries of multi-word levels: the whole wavefront streams the block, one entry at a time
      KPROF_ADD(d.g, 22, __popcll(__ballot(wide)));
      for (uint64_t m = __ballot(wide); m; m &= m - 1) {
        const int src = __ffsll((unsigned long long)m) - 1;
        const int el = __shfl(l, src, 64);
        const int eslot = (int)__shfl(aux, src, 64);
        const Lv ev = sib_view(node, el);
        const uint64_t WG_G* sig = GsfProto::sig_ptr(s, node, eslot);
        uint64_t a = 0, f = 0;
        H_FOR_WORDS(ev, j) {
          const uint64_t sg = sig[j], vw = vr[ev.bw + j], iw = ivr[ev.bw + j];
          a += (uint64_t)__popcll(sg) | ((uint64_t)__popcll(sg | iw) << 21) | ((uint64_t)__popcll(sg | iw | vw) << 42);
          f |= (uint64_t)((sg & vw) != 0) | ((uint64_t)((sg & iw) != 0) << 1);
        }
        a = wave_sum64(a);
        const bool iV = __ballot((f & 1) != 0) != 0, iIV = __ballot((f & 2) != 0) != 0;
        if (lane == src) {
          const int cs = (int)(a & 0x1FFFFF), cSI = (int)((a >> 21) & 0x1FFFFF), cM = (int)((a >> 42) & 0x1FFFFF);
          const int newTotal = cVl == 0 ? cs : (iV ? cSI : cM);
          ns = g_score(l, size, newTotal, cVl == 0 ? cs : newTotal - cVl, cs == 1, iIV);
        }
      }
      if (mine) shNs[w][i] = ns;
      int mx = ns;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
      if (mx > bestScore) {  // `ns > score`: the first entry reaching a new maximum wins
        bestScore = mx;
        const int src = __ffsll((unsigned long long)__ballot(mine && ns == mx)) - 1;
        bestIdx = base + src;
        bestEnt = shfl64(e, src);
      }
    }
    __builtin_amdgcn_wave_barrier();
    KPROF_ADD(d.g, 21, len);
    KPROF_MARK(d.g, 17);  // checkSigs: the entries' scores
    // rewrite the list: zeros are removed (:571-573), the best leaves it (:577)
    int newLen = 0;
    for (int base = 0; base < len; base += 64) {
      const int i = base + lane;
      const bool mine = i < len;
      const uint64_t e = mine ? ent[i] : 0ULL;
      const int ns = mine ? shNs[w][i] : 0;
      const bool keep = mine && ns != 0 && i != bestIdx;
      const uint64_t km = __ballot(keep);
      const bool freeSlot = mine && ns == 0 && g_ent_kind(e) == GK_PARTIAL;
      for (uint64_t m = __ballot(freeSlot); m; m &= m - 1) {
        const uint32_t sl = __shfl(g_ent_aux(e), __ffsll((unsigned long long)m) - 1, 64);
        if (lane == 0) ls->used[sl >> 6] &= ~(1ULL << (sl & 63));
      }
      if (keep) ent[newLen + __popcll(km & lanes_lt())] = e;
      newLen += __popcll(km);
    }
    __builtin_amdgcn_wave_barrier();
    KPROF_MARK(d.g, 18);  // checkSigs: the list rewritten
    for (int k = lane; k < s.Q / 64; k += 64) s.tvUsed[(size_t)node * (s.Q / 64) + k] = ls->used[k];
    if (lane == 0) {
      s.tvLen[node] = newLen;
      if (bestIdx >= 0) {
        // sigChecked++; sigQueueSize = toVerify.size(); registerTask(updateVerifiedSignatures(best), ...) :576-583.
        // The task's closure (tBest) is the pend entry; k_gsf_cond_a2 emits the task record in node order.
        int pe = -1;
        for (int k = 0; k < G_PEND; k++)
          if (!(s.pend[(size_t)node * G_PEND + k] & 0x80000000u)) {
            pe = k;
            break;
          }
        if (pe < 0) {
          set_err(d.g, ERR_PENDING);
          pe = 0;
        }
        s.pend[(size_t)node * G_PEND + pe] = 0x80000000u | (g_ent_kind(bestEnt) << 29) |
                                             ((uint32_t)g_ent_level(bestEnt) << 24) | g_ent_aux(bestEnt);
        s.pendFrom[(size_t)node * G_PEND + pe] = g_ent_from(bestEnt);
        s.candPend[node] = (uint8_t)pe;
        s.sigChecked[node]++;
        s.sigQueueSize[node] = newLen;
        s.candFlag[node] = 1;
      }
    }
    __builtin_amdgcn_wave_barrier();
    KPROF_MARK(d.g, 19);  // checkSigs: the chosen entry parked (pend table, counters)
  }
}

// Which nodes the two lean kernels of the delivery pass take between them: a node all of whose events of the ms (as many as
// its inbox line holds) are SendSigs deliveries and, at most once, its doCycle task. onNewSig (:538-556) touches the toVerify
// list, individualSignatures and the payload store; doCycle (:213-225) reads posInLevel / remainingCalls / the verified
// counts and writes the first two — disjoint state, and an event's records and draws take their place in the global order
// from the event's index, not from when it ran: the task goes to k_gsf_docycle, the deliveries to k_gsf_lane, in either
// order. (updateVerifiedSignatures changes what both read: such a node is visited in event order by k_deliver_inbox.)
// Both kernels evaluate this one predicate on the same line, so they agree. `cycleRan`: k_gsf_docycle is launched in this ms.
__device__ __forceinline__ bool gsf_split_ok(const EngineDev& d, int32_t node, uint32_t cnt, const InboxEntry (&in)[INBOX_SLOTS],
                                             bool cycleRan, int& cycleAt) {
  cycleAt = -1;
  if (cnt == 0 || cnt > (uint32_t)INBOX_SLOTS || d.nparts || d.boundMsg || d.nodes.down[node]) return false;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < INBOX_SLOTS; k++) {
    if ((uint32_t)k >= cnt) continue;
    const uint32_t w0 = in[k].w0, msg = in[k].w2, kind = (w0 >> 28) & 3u;
    if (kind == K_MSG) {
      if (((msg >> 6) & 3u) == GK_PARTIAL && h_nw((int)(msg & 31u)) > 32) ok = false;  // (a payload wider than a lane copies)
    } else if (kind == K_PERIODIC && msg == G_TASK_DOCYCLE && cycleRan && cycleAt < 0) {
      cycleAt = k;
    } else {
      ok = false;
    }
  }
  return ok;
}

// The delivery pass, doCycle tier: one WAVEFRONT per node whose only event of the ms is its doCycle task (:213-225) — with a
// synchronised start that is nearly every node, once per period, in a ms that costs several ordinary ones. The task reads
// posInLevel / remainingCalls / |verifiedSignatures| of the levels and nothing else of the node's state: a kernel that holds
// neither onNewSig nor updateVerifiedSignatures loads three arrays instead of node_begin's thirteen lines and fits more
// wavefronts per SIMD. Everything else (a node that also received a message in that ms, down nodes, a partitioned network)
// stays with k_deliver_inbox; a node delivered here has its inbox count zeroed.
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k_gsf_docycle(const EngineDev* __restrict__ tab, const GsfState* __restrict__ stab) {  // (k_gsf_lane must follow)
  WG_ENGINE(tab);
  const GsfState& s = stab[wgBy];
  __shared__ GLevels shL[4];
  const int lane = WG_LANE, w = threadIdx.x >> 6;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nActive = d.g->nActive;
  const int32_t t = d.g->now;
  if (d.nparts) return;
  GLevels* ls = &shL[w];
  for (uint32_t a = wave; a < nActive; a += nWaves) {
    const int32_t node = (int32_t)d.active[a];
    const uint32_t cnt = d.icnt[node];
    if (cnt == 0 || cnt > (uint32_t)INBOX_SLOTS) continue;
    InboxEntry line[INBOX_SLOTS];
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) line[k] = gld(d.inbox + ((size_t)node * INBOX_SLOTS + k));
    int cycleAt;
    if (!gsf_split_ok(d, node, cnt, line, true, cycleAt) || cycleAt < 0) continue;
    InboxEntry in = line[0];
#pragma unroll
    for (int k = 1; k < INBOX_SLOTS; k++)
      if (k == cycleAt) in = line[k];
    // (the inbox count stays: k_gsf_lane, which runs behind this kernel, delivers the node's messages if it has any and zeroes it)
    for (int l = lane; l < s.L; l += 64) {
      const size_t i = (size_t)node * s.L + l;
      ls->pos[l] = s.pos[i];
      ls->rem[l] = s.rem[i];
      ls->cV[l] = s.cV[i];
    }
    __builtin_amdgcn_wave_barrier();
    // (a task's `from` is the node itself; its inbox entry carries the event's first outbox slot, deliver_visit_inbox)
    Ctx c{d, t, node, in.e, 0, 0, in.w0 & 0x0FFFFFFFu, d.boundTask[G_TASK_DOCYCLE] + 1u, 0, 0, 0};
    GsfProto::NodeRegs r;
    r.ls = ls;
    r.dirty = 0;
    r.doneAt = r.doneAt0 = 0;
    GsfProto::do_cycle(c, s, r);
    // PeriodicTask.action re-arm (C/messages/PeriodicTask.java:39-47), as deliver_event
    c.put(O_PERIODIC, node, in.w2, in.w3, t + (int32_t)in.w3, 0, false);
    if (lane == 0) {
      EvRes res;
      res.nrec = c.sub | EV_TASK_RUN | c.evFlags;
      res.ndraw = c.draws;
      gst(d.evRes + in.e, res);
      if (c.msgSent) {
        atomicAdd((unsigned long long*)&d.nodes.msgSent[node], (unsigned long long)c.msgSent);
        atomicAdd((unsigned long long*)&d.nodes.bytesSent[node], (unsigned long long)c.bytesSent);
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int l = lane; l < s.L; l += 64) {
      const size_t i = (size_t)node * s.L + l;
      s.pos[i] = ls->pos[l];
      s.rem[i] = ls->rem[l];
    }
    __builtin_amdgcn_wave_barrier();  // (the LDS image is the next node's from here on)
  }
}

// ... the same with SIXTEEN LANES per node, four nodes per wavefront (networks of up to 32 768 nodes: a level per lane).
// The task is a level per lane (twelve of them at 4096 nodes): one wavefront per node left three quarters of its lanes idle,
// and the kernel's duration is nodes / resident wavefronts x the task's chain of round trips. No LDS image: lane l reads
// level l's three scalars itself; first incomplete level, ranks of the sends and counts are 16-bit pieces of the ballots.
__global__ void __launch_bounds__(256) k_gsf_docycle16(const EngineDev* __restrict__ tab, const GsfState* __restrict__ stab) {
  WG_ENGINE(tab);
  const GsfState& s = stab[wgBy];
  const int lane = WG_LANE, l16 = lane & 15, gsh = lane & ~15;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nActive = d.g->nActive;
  const int32_t t = d.g->now;
  if (d.nparts) return;
  for (uint32_t a0 = wave * 4; a0 < nActive; a0 += nWaves * 4) {
    const uint32_t a = a0 + (uint32_t)(lane >> 4);
    const bool have = a < nActive;
    const int32_t node = have ? (int32_t)d.active[a] : 0;
    const uint32_t cnt = have ? d.icnt[node] : 0u;
    InboxEntry line[INBOX_SLOTS];
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) {
      line[k].e = 0;
      line[k].w0 = line[k].w2 = line[k].w3 = 0;
      if (have && cnt != 0 && cnt <= (uint32_t)INBOX_SLOTS) line[k] = gld(d.inbox + ((size_t)node * INBOX_SLOTS + k));
    }
    int cycleAt = -1;
    const bool ok = have && gsf_split_ok(d, node, cnt, line, true, cycleAt) && cycleAt >= 0;
    InboxEntry in = line[0];
#pragma unroll
    for (int k = 1; k < INBOX_SLOTS; k++)
      if (k == cycleAt) in = line[k];
    // (the inbox count stays: k_gsf_lane, which runs behind this kernel, delivers the node's messages if it has any and zeroes it)
    const bool lv = ok && l16 >= 1 && l16 < s.L;
    const size_t li = (size_t)node * s.L + l16;
    int pos = 0, rem = 0, cvl = 0;
    if (lv) {
      pos = s.pos[li];
      rem = s.rem[li];
      cvl = s.cV[li];
    }
    const int size = lv ? 1 << (l16 - 1) : 0;
    const uint32_t incM = (uint32_t)(__ballot(lv && cvl != size) >> gsh) & 0xFFFFu;
    const int k = incM ? __ffs(incM) - 1 : s.L;  // first incomplete level (levels below it form getLastFinishedLevel :194-211)
    const bool act = lv && rem != 0 && (t >= l16 * s.p.timeoutPerLevelMs || k >= l16);  // hasStarted :294-315
    int32_t dest = 0;
    if (act) dest = s.peers[(size_t)node * (s.N - 1) + (size - 1) + pos];  // getRemainingPeers(1)
    const uint32_t actM = (uint32_t)(__ballot(act) >> gsh) & 0xFFFFu;
    const uint32_t partM = (uint32_t)(__ballot(act && k < l16) >> gsh) & 0xFFFFu;  // PARTIAL: the verified bits of the levels below l
    const uint32_t win = ((uint32_t)t / (uint32_t)s.p.periodDurationMs) % s.snapNb;
    const uint32_t refBase = (win * (uint32_t)s.N + (uint32_t)node) * s.snapStride;
    if (partM) {
      const uint64_t WG_G* vr = s.V + (size_t)node * s.W;
      for (uint32_t q = (uint32_t)l16; q < s.snapStride; q += 16) {
        int l = 1;  // the level whose words hold flat word q (lvlOff is increasing from level 1 on)
        for (int i = 2; i < s.L; i++)
          if (s.lvlOff[i] <= q) l = i;
        if ((partM >> l) & 1u) {
          const Lv v = own_view(node, l);
          s.snap[refBase + q] = vr[v.bw + (int)(q - s.lvlOff[l])] & v.mask;
        }
      }
    }
    // the records, in level order: the sends (as Ctx::send_many writes them), then the task's re-arm (PeriodicTask.action,
    // C/messages/PeriodicTask.java:39-47, as deliver_event's Ctx::put)
    const uint32_t outBase = in.w0 & 0x0FFFFFFFu, outCap = d.boundTask[G_TASK_DOCYCLE] + 1u;
    const int n = __popc(actM), rank = __popc(actM & ((1u << l16) - 1u));
    if (act || (ok && l16 == 0)) {
      const uint32_t idx = act ? (uint32_t)rank : (uint32_t)n;
      if (idx < outCap && outBase + idx < d.maxOut) {
        Out o;
        if (act) {
          o.kindfrom = (O_SEND << 28) | (uint32_t)node;
          o.to = dest;
          o.a = GsfProto::msg_word(l16, k, cvl == size);
          o.b = k < l16 ? refBase + s.lvlOff[l16] : 0u;
          o.t = t + 1;
          o.drawsub = (uint32_t)rank;
        } else {
          o.kindfrom = (O_PERIODIC << 28) | (uint32_t)node;
          o.to = node;
          o.a = in.w2;
          o.b = in.w3;
          o.t = t + (int32_t)in.w3;
          o.drawsub = (uint32_t)n;
        }
        o.destOff = 0;
        o.pad = 0;
        d.outTmp[outBase + idx] = o;
      } else {
        set_err(d.g, ERR_OUTBOX);
      }
    }
    if (act) {
      s.pos[li] = pos + 1 >= size ? 0 : pos + 1;
      s.rem[li] = rem - 1;
    }
    if (ok && l16 == 0) {
      EvRes res;
      res.nrec = (uint32_t)(n + 1) | EV_TASK_RUN | (partM ? ev_snap_code(s.snapStride) << EV_SNAP_SHIFT : 0u);
      res.ndraw = (uint32_t)n;
      gst(d.evRes + in.e, res);
      if (n) {
        long long bytes = 0;
        for (uint32_t m = actM; m; m &= m - 1) bytes += g_msg_size(__ffs(m) - 1);
        atomicAdd((unsigned long long*)&d.nodes.msgSent[node], (unsigned long long)n);
        atomicAdd((unsigned long long*)&d.nodes.bytesSent[node], (unsigned long long)bytes);
      }
    }
  }
}

// The delivery pass, lane tier: one LANE per node whose events of the ms are all plain SendSigs deliveries — onNewSig
// (:538-556) is an append to the node's toVerify list, one bit of individualSignatures and, for a PARTIAL payload, a slot of
// the node's payload store: four or five lines of the node, where a wavefront's visit loads the node's whole level state
// (node_begin: thirteen lines) to touch none of it. Left to k_deliver_inbox: nodes with a task (updateVerifiedSignatures,
// doCycle), with more events than the
// inbox line holds, down nodes, a partitioned network, payloads wider than a lane copies (> 32 words). A node delivered
// here has its inbox count zeroed: the wavefront kernel passes it over.
// `listB`: the nodes this kernel and the ones before it leave to k_deliver_inbox are listed in EngineDev::activeB (one atomic per
// wavefront) — that kernel then visits those and nothing else (k_deliver_inbox<.., LISTB>).
__global__ void __launch_bounds__(256) k_gsf_lane(const EngineDev* __restrict__ tab, const GsfState* __restrict__ stab, int cycleRan, int listB) {
  WG_ENGINE(tab);
  const GsfState& s = stab[wgBy];
  const uint32_t nActive = d.g->nActive;
  const int lane = WG_LANE;
  uint32_t WG_G* rest = (uint32_t WG_G*)(VisitDesc WG_G*)d.activeB;
  const bool lean = !(d.nparts || d.boundMsg);
  if (!lean && !listB) return;
  for (uint32_t a = wgBx * blockDim.x + threadIdx.x; a < nActive; a += wgGx * blockDim.x) {
    const int32_t node = (int32_t)d.active[a];
    const uint32_t cnt = d.icnt[node];
    InboxEntry in[INBOX_SLOTS];
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) in[k].e = in[k].w0 = in[k].w2 = in[k].w3 = 0;
    int cycleAt = -1;
    bool mine = false;
    if (lean && cnt != 0 && cnt <= (uint32_t)INBOX_SLOTS) {
#pragma unroll
      for (int k = 0; k < INBOX_SLOTS; k++) in[k] = gld(d.inbox + ((size_t)node * INBOX_SLOTS + k));
      mine = gsf_split_ok(d, node, cnt, in, cycleRan != 0, cycleAt);
    }
    if (listB) {
      const uint64_t m = __ballot(cnt != 0 && !mine);
      if (m) {
        uint32_t bb = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane == leader) bb = atomicAdd(F(&d.g->nActiveB), (uint32_t)__popcll(m));
        bb = lane_bcast(bb, leader);
        if (cnt != 0 && !mine) rest[bb + __popcll(m & lanes_lt())] = (uint32_t)node;
      }
    }
    if (!mine) continue;
    d.icnt[node] = 0;  // the line is consumed by this visit
    // event order = ascending event index (the line holds them in arrival order of the atomics, not in event order)
    uint32_t rank[INBOX_SLOTS];
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) {
      rank[k] = 0;
#pragma unroll
      for (int j = 0; j < INBOX_SLOTS; j++)
        if ((uint32_t)j < cnt && in[j].e < in[k].e) rank[k]++;
    }
    // everything the events' addresses alone decide, in ONE round of loads (the kernel is one wave-round of lanes: its
    // duration is a lane's chain of dependent round trips): the list's length, per event the sender's word of
    // individualSignatures and — for a hop of a multi-destination envelope — its EvAux; the envelope's length follows
    uint64_t WG_G* isRow = s.IS + (size_t)node * s.W;
    uint64_t WG_G* ent = s.tvEnt + (size_t)node * s.Q;
    int len = s.tvLen[node];
    uint64_t iswK[INBOX_SLOTS];
    EvAux auxK[INBOX_SLOTS];
    int32_t ndK[INBOX_SLOTS];
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) {
      const bool have = (uint32_t)k < cnt && ((in[k].w0 >> 28) & 3u) == K_MSG;  // (a task's w0 holds an outbox slot, not a sender)
      iswK[k] = have ? isRow[(in[k].w0 & 0x0FFFFFFFu) >> 6] : 0ULL;
      auxK[k].chain = -1;
      auxK[k].cpos = 0;
      auxK[k].outBase = auxK[k].outCap = 0;
      if (have && (in[k].w0 & INBOX_CHAIN)) auxK[k] = gld(d.evAux + in[k].e);
    }
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++)
      ndK[k] = (auxK[k].chain >= 0 && auxK[k].cpos < 0) ? d.chains[auxK[k].chain].ndest : 0;
    long long bytes = 0;
    uint32_t nMsg = 0;
    for (uint32_t r = 0; r < cnt; r++) {
      InboxEntry ev = in[0];
      uint64_t isw = iswK[0];
      EvAux aux0 = auxK[0];
      int32_t nd = ndK[0];
#pragma unroll
      for (int k = 1; k < INBOX_SLOTS; k++)
        if ((uint32_t)k < cnt && rank[k] == r) {
          ev = in[k];
          isw = iswK[k];
          aux0 = auxK[k];
          nd = ndK[k];
        }
      if (((ev.w0 >> 28) & 3u) != K_MSG) continue;  // (the node's doCycle task: k_gsf_docycle has run it)
      nMsg++;
      const int32_t from = (int32_t)(ev.w0 & 0x0FFFFFFFu);
      const uint32_t msg = ev.w2, payload = ev.w3;
      const int l = (int)(msg & 31u);
      const uint32_t kind = (msg >> 6) & 3u;
      uint32_t aux = (msg >> 8) & 31u;
      bytes += g_msg_size(l);
      EvRes res;
      res.nrec = EV_DELIVERED | ((uint32_t)l << 24);
      res.ndraw = 0;
      if (aux0.chain >= 0 && aux0.cpos < 0) {  // last hop of a multi-destination envelope's run (the accelerated calls :445-449):
        const int32_t next = (aux0.cpos & 0x7FFFFFFF) + 1;  // markRead(); if (hasNextReader()) msgs.addMsg(m)  C/Network.java:629-632
        if (next < nd) {
          Out o;
          o.kindfrom = (O_CHAINCONT << 28) | (uint32_t)node;
          o.to = aux0.chain;
          o.a = (uint32_t)next;
          o.b = 0;
          o.t = 0;
          o.destOff = 0;
          o.drawsub = 0;
          o.pad = 0;
          if (aux0.outCap && aux0.outBase < d.maxOut) {  // (as Ctx::put: an over-subscribed ms must not write past the outbox)
            d.outTmp[aux0.outBase] = o;
            res.nrec |= 1u;
          } else {
            set_err(d.g, ERR_OUTBOX);
          }
        } else {
          d.chains[aux0.chain].flags = 0;  // envelope fully delivered
        }
      }
      gst(d.evRes + ev.e, res);
      // onNewSig :538-556
      const bool hadIS = (isw >> (from & 63)) & 1ULL;
      const int need = hadIS ? 1 : 2;
      if (len + need > s.Q) {
        set_err(d.g, ERR_QUEUE_CAP);
        continue;
      }
      if (kind == GK_PARTIAL) {
        int slot = -1;
        uint64_t WG_G* used = s.tvUsed + (size_t)node * (s.Q / 64);
        for (int q = 0; q < s.Q / 64 && slot < 0; q++) {
          const unsigned long long u = used[q];
          if (~u) {
            slot = q * 64 + __ffsll(~u) - 1;
            used[q] = u | (1ULL << (slot & 63));
          }
        }
        if (slot < 0) {
          set_err(d.g, ERR_QUEUE_CAP);
          continue;
        }
        const int nw = h_nw(l);
        const uint64_t WG_G* src = s.snap + payload;
        uint64_t WG_G* dst = GsfProto::sig_ptr(s, node, slot);
        for (int j0 = 0; j0 < nw; j0 += 8) {  // (eight words in flight)
          uint64_t v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = j0 + u < nw ? src[j0 + u] : 0ULL;
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (j0 + u < nw) dst[j0 + u] = v[u];
        }
        aux = (uint32_t)slot;
      }
      ent[len] = g_ent(from, l, kind, aux);                    // toVerify.add(ssigs)
      if (!hadIS) {
        ent[len + 1] = g_ent(from, l, GK_INDIV, 0);            // the individual signature (:547-553)
        isRow[from >> 6] = isw | (1ULL << (from & 63));
#pragma unroll
        for (int k = 0; k < INBOX_SLOTS; k++)  // (a later event of this visit whose sender shares the word: it was loaded before this store)
          if (((in[k].w0 >> 28) & 3u) == K_MSG && (int32_t)((in[k].w0 & 0x0FFFFFFFu) >> 6) == (from >> 6)) iswK[k] |= 1ULL << (from & 63);
      }
      len += need;
    }
    if (nMsg) {
      s.tvLen[node] = len;
      s.sigQueueSize[node] = len;
      atomicAdd((unsigned long long*)&d.nodes.msgReceived[node], (unsigned long long)nMsg);
      atomicAdd((unsigned long long*)&d.nodes.bytesReceived[node], (unsigned long long)bytes);
    }
  }
}

// scan over nodes: ordinal of every node whose checkSigs registers a task (conditional tasks run in
// registration = node id order, so that is the push order). GSF's checkSigs draws nothing from rd.
struct GsfCondF {
  typedef GsfState Aux;
  const EngineDev& d;
  const GsfState& s;
  __device__ GsfCondF(const EngineDev& d_, const Aux* a) : d(d_), s(*a) {}
  __device__ uint32_t count() const { return (uint32_t)s.N; }
  __device__ uint64_t value(uint32_t i) const { return s.candFlag[i] != 0; }
  __device__ void tally(uint32_t, uint32_t) const {}
  __device__ void total(uint64_t tot) const {
    if (d.g->nOutKeep + (uint32_t)tot > d.maxOut) {  // (the drain's outbox is still in fin / arr: the edge's records follow it)
      set_err(d.g, ERR_OUTBOX);
      tot = 0;
    }
    d.g->nOut = (uint32_t)tot;
    d.g->nDraws = 0;
  }
  __device__ void write(uint32_t i, uint64_t excl, bool valid) const {
    if (valid && s.candFlag[i]) s.condList[(uint32_t)excl] = i;
  }
};

// SH (sharded engine): a registering node is handled by its owner, the task record goes to the exchange image.
template <bool SH>
__global__ void __launch_bounds__(256) k_gsf_cond_a2(const EngineDev* __restrict__ tab, const GsfState* __restrict__ stab) {
  WG_ENGINE(tab);
  const GsfState& s = stab[wgBy];
  const uint32_t n = d.g->nOut;
  const int32_t t = d.g->now;
  const uint32_t D = (uint32_t)d.horizon;
  const uint32_t stride = wgGx * blockDim.x;
  if (wgBx == 0 && threadIdx.x == 0) {  // for the next edge's k_gsf_cond_pre / k_gsf_cond_a1g
    *s.runCount = 0;
    *s.runCount2 = 0;
  }
  for (uint32_t j0 = wgBx * blockDim.x; j0 < n; j0 += stride) {
    const uint32_t j = j0 + threadIdx.x;
    uint32_t histKey = 0xFFFFFFFFu;
    if (j < n) {
      const int32_t node = (int32_t)s.condList[j];
      if (SH && !shard_owns(d, node)) {
        for (int q = 0; q < 5; q++) d.xbuf[(size_t)j * 5 + q] = 0;
        continue;
      }
      const int32_t arrival = t + s.pairing[node];
      const Rec fin = make_rec(K_TASK, node, (uint32_t)node, G_TASK_UPDATE, (uint32_t)s.candPend[node]);
      const bool ok = arrival - t < d.horizon - 1;
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
      const uint32_t jp = d.g->nOutKeep + j;  // (behind the drain's records, if the drain left them to this phase's append)
      d.fin[jp] = fin;
      d.arr[jp] = ok ? arrival : -1;
      if (ok) histKey = (jp / TILE) * D + ((uint32_t)arrival & (D - 1));
    }
    if (SH) continue;  // (k_shard_unpack builds the tile histograms from the summed image)
    uint64_t todo = __ballot(histKey != 0xFFFFFFFFu);
    while (todo) {
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const uint32_t key = __shfl(histKey, leader, 64);
      const uint64_t m = __ballot(histKey == key) & todo;
      if ((int)WG_LANE == leader) atomicAdd(&d.tileHist[key], (uint32_t)__popcll(m));
      todo &= ~m;
    }
  }
}

}  // namespace wg
