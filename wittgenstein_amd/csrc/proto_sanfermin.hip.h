// San Fermin signature aggregation (P/SanFerminSignature.java, P/SanFerminHelper.java) as a resident device
// protocol. One wavefront per simulated node (k_deliver): scalars are wave-uniform, lanes share the scans of the
// node's candidate bitsets.
//
// With 2^P nodes (the reference's toBinaryID padding throws for other counts, P/SanFerminHelper.java:158-171) the
// helper's interval arithmetic (:38-92) is block arithmetic: at level l (currentPrefixLength after its decrement,
// P-1 .. 0) the own set is the node's aligned block of S = N >> (l+1) ids and the candidate set is the sibling block;
// allNodes is in id order, so list indices are id offsets.
//   used     [N][N bits]  SanFerminHelper.usedNodes: level l's BitSet at bit offset N - 2S (the levels' sizes sum to N-1)
//   pending  [N][N/2 bits] SanFerminNode.pendingNodes (ids sent to at the current level) as offsets into the candidate block
//   cache    [N][P+1]      signatureCache values, present bits in cacheMask
// pickNextNodes ends with Collections.shuffle(newList, rd) (:144): the permutation only orders the destinations of the
// send that follows, so it is deferred to `resolve` (Ctx::send_list_shuffled, shuffle_dests in engine_kernels.hip.h).
// futurSigs is never written by the reference (only read, :398-408) and is not kept.
#pragma once
#include "engine_kernels.hip.h"

namespace wg {

constexpr uint32_t SF_TASK_START = 0;     // init(): registerTask(n::goNextLevel, 1, n)  :139-141
constexpr uint32_t SF_TASK_TIMEOUT = 1;   // sendToNodes' timeout, arg = the level it was armed at  :346-362
constexpr uint32_t SF_TASK_VERIFIED = 2;  // transition's task, arg = the value to aggregate  :431-449
enum SfFlags : uint32_t { SF_DONE = 1, SF_THRESHOLD_DONE = 2, SF_SWAPPING = 4 };

struct SfState {
  wg_sanfermin_params p;
  int32_t N, P, W;        // W = 64-bit words of an N-bit row
  GP<int32_t> cpl, agg, sentReq, recvReq, thresholdAt;
  GP<uint32_t> flags, cacheMask;
  GP<int32_t> cache;         // [N][P + 1]
  GP<uint64_t> used;         // [N][W]
  GP<uint64_t> pending;      // [N][W]  (only the first S bits of the current level are meaningful)
};

struct SfProto {
  typedef SfState State;
  struct WaveShared {
    int32_t list[64];
  };
  struct NodeRegs {
    int32_t cpl, agg, sentReq, recvReq, thresholdAt;
    uint32_t flags, cacheMask;
    long long doneAt;
    WaveShared* sh;
  };
  // message word: bit 0 reply, bit 1 status NO, bits 2..7 level; payload = aggValue
  __device__ static uint32_t word(bool reply, bool no, int level) { return (reply ? 1u : 0u) | (no ? 2u : 0u) | ((uint32_t)level << 2); }
  __device__ static int msg_size(const State& s, uint32_t) { return 4 + s.p.signatureSize; }  // :547-550, :569-572
  __device__ static int msg_level(uint32_t msg) { return (int)((msg >> 2) & 31u); }

  __device__ static void node_begin(Ctx& c, const State& s, NodeRegs& r, WaveShared* sh) {
    const int32_t n = c.node;
    r.cpl = s.cpl[n];
    r.agg = s.agg[n];
    r.sentReq = s.sentReq[n];
    r.recvReq = s.recvReq[n];
    r.thresholdAt = s.thresholdAt[n];
    r.flags = s.flags[n];
    r.cacheMask = s.cacheMask[n];
    r.doneAt = c.d.nodes.doneAt[n];
    r.sh = sh;
  }
  __device__ static void node_end(Ctx& c, const State& s, NodeRegs& r) {
    __builtin_amdgcn_wave_barrier();
    if (WG_LANE == 0) {
      const int32_t n = c.node;
      s.cpl[n] = r.cpl;
      s.agg[n] = r.agg;
      s.sentReq[n] = r.sentReq;
      s.recvReq[n] = r.recvReq;
      s.thresholdAt[n] = r.thresholdAt;
      s.flags[n] = r.flags;
      s.cacheMask[n] = r.cacheMask;
      c.d.nodes.doneAt[n] = r.doneAt;
    }
  }

  // ---- SanFerminHelper on aligned blocks -------------------------------------------------------------
  __device__ static int32_t block_size(const State& s, int level) { return s.N >> (level + 1); }
  __device__ static int32_t own_base(const State& s, int32_t node, int level) {
    const int32_t S = block_size(s, level);
    return S > 0 ? (node / S) * S : node;
  }
  __device__ static bool is_candidate(const State& s, int32_t self, int32_t other, int level) {  // :94-96
    if (level < 0 || level >= s.P) return false;
    const int32_t S = block_size(s, level), cb = own_base(s, self, level) ^ S;
    return other >= cb && other < cb + S;
  }
  __device__ static bool row_bit(const uint64_t WG_G* row, int64_t bit) {
    return (__hip_atomic_load(row + (bit >> 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (bit & 63)) & 1ULL;
  }
  __device__ static void row_set_bit(uint64_t WG_G* row, int64_t bit) {  // (called by one lane)
    row[bit >> 6] |= 1ULL << (bit & 63);
  }
  // first index i in [0, len) with bit (off + i) of `row` clear, or len; wave-parallel over the words
  __device__ static int32_t first_clear(const uint64_t WG_G* row, int64_t off, int32_t len) {
    const int64_t lo = off, hi = off + len;
    int32_t best = len;
    for (int64_t w0 = (lo >> 6); w0 <= ((hi - 1) >> 6) && best == len; w0 += 64) {
      const int64_t w = w0 + WG_LANE;
      int32_t mine = len;
      if (w <= ((hi - 1) >> 6)) {
        uint64_t v = ~__hip_atomic_load(row + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == (lo >> 6)) v &= ~0ULL << (lo & 63);
        if (w == ((hi - 1) >> 6) && ((hi & 63) != 0)) v &= (1ULL << (hi & 63)) - 1ULL;
        if (v) mine = (int32_t)((w << 6) + (__ffsll((unsigned long long)v) - 1) - lo);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mine = min(mine, __shfl_xor(mine, o, 64));
      best = mine;
    }
    return best;
  }
  // pickNextNodes(level, howMany) before its shuffle (:112-143): the list in r.sh->list, returns its length
  __device__ static int pick_next_nodes(Ctx& c, const State& s, NodeRegs& r, int level, int howMany) {
    const int32_t node = c.node;
    const int32_t S = block_size(s, level), ob = own_base(s, node, level), cb = ob ^ S, idx = node - ob;
    uint64_t WG_G* used = s.used + (size_t)node * s.W;
    const int64_t off = (int64_t)s.N - 2 * (int64_t)S;
    int n = 0;
    const bool first = !row_bit(used, off + idx);
    __builtin_amdgcn_wave_barrier();
    if (first) {  // "add the correct one first if not already": candidateSet.get(idx), removed from the list
      if (WG_LANE == 0) {
        r.sh->list[n] = cb + idx;
        row_set_bit(used, off + idx);
      }
      n++;
      __threadfence_block();
    }
    const int32_t len = first ? S - 1 : S;  // the list the stream below runs over
    for (int k = 0; k < howMany && n < 64; k++) {
      const int32_t i = first_clear(used, off, len);
      if (i >= len) break;
      if (WG_LANE == 0) {
        r.sh->list[n] = first ? (i < idx ? cb + i : cb + i + 1) : cb + i;
        row_set_bit(used, off + i);
      }
      n++;
      __threadfence_block();
    }
    __builtin_amdgcn_wave_barrier();
    return n;
  }

  // ---- SanFerminNode ---------------------------------------------------------------------------------
  __device__ static void send_swap_reply(Ctx& c, const State& s, int32_t to, bool no, int level, int32_t value) {  // :416-423
    c.send(to, word(true, no, level), (uint32_t)value, 4 + s.p.signatureSize);
  }
  __device__ static void transition(Ctx& c, const State& s, NodeRegs& r, int32_t toAggregate) {  // :429-450
    r.flags |= SF_SWAPPING;
    c.register_task(c.t + s.p.pairingTime, SF_TASK_VERIFIED, (uint32_t)toAggregate);
  }
  __device__ static void send_to_nodes(Ctx& c, const State& s, NodeRegs& r, int n) {  // :322-363
    if (n == 0) return;
    const int32_t node = c.node;
    const int32_t S = block_size(s, r.cpl), cb = own_base(s, node, r.cpl) ^ S;
    uint64_t WG_G* pend = s.pending + (size_t)node * s.W;
    const uint32_t destOff = c.dest_reserve(n);
    if ((int)WG_LANE < n) c.dest_put(destOff, (int)WG_LANE, r.sh->list[WG_LANE]);
    if (WG_LANE == 0)
      for (int k = 0; k < n; k++) row_set_bit(pend, r.sh->list[k] - cb);  // pendingNodes.addAll
    r.sentReq += n;
    __threadfence_block();
    c.send_list_shuffled(destOff, n, word(false, false, r.cpl), (uint32_t)r.agg, 4 + s.p.signatureSize);
    c.register_task(c.t + s.p.replyTimeout, SF_TASK_TIMEOUT, (uint32_t)r.cpl);
  }
  __device__ static void go_next_level(Ctx& c, const State& s, NodeRegs& r) {  // :373-414
    if (r.flags & SF_DONE) return;
    if (r.agg >= s.p.threshold && !(r.flags & SF_THRESHOLD_DONE)) {
      r.flags |= SF_THRESHOLD_DONE;
      r.thresholdAt = c.t + s.p.pairingTime * 2;
    }
    if (r.cpl == 0) {
      r.doneAt = c.t + s.p.pairingTime * 2;
      r.flags |= SF_DONE;
      return;
    }
    r.cpl--;
    if (WG_LANE == 0) s.cache[(size_t)c.node * (s.P + 1) + r.cpl] = r.agg;  // signatureCache.put(level, aggValue)
    r.cacheMask |= 1u << r.cpl;
    r.flags &= ~SF_SWAPPING;
    {  // pendingNodes = new HashSet<>()
      uint64_t WG_G* pend = s.pending + (size_t)c.node * s.W;
      const int32_t words = (block_size(s, r.cpl) + 63) >> 6;
      for (int w = WG_LANE; w < words; w += 64) pend[w] = 0;
      __threadfence_block();
    }
    send_to_nodes(c, s, r, pick_next_nodes(c, s, r, r.cpl, s.p.candidateCount));
  }

  __device__ static void on_message(Ctx& c, const State& s, NodeRegs& r, int32_t from, uint32_t msg, uint32_t payload) {
    const int level = (int)((msg >> 2) & 63u);
    const int32_t value = (int32_t)payload;
    if (!(msg & 1u)) {  // SwapRequest -> onSwapRequest :224-264
      r.recvReq++;
      if ((r.flags & SF_DONE) || level != r.cpl) {
        if (level <= s.P && ((r.cacheMask >> level) & 1u)) {
          // (an earlier event of this visit may have written it: read past the L1, as proto_handel's ld_coherent)
          const int32_t cached = __hip_atomic_load(&s.cache[(size_t)c.node * (s.P + 1) + level], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
          send_swap_reply(c, s, from, false, level, cached);  // optimistic reply with the cached signature
        } else {
          send_swap_reply(c, s, from, true, r.cpl, 0);
          if (is_candidate(s, c.node, from, level)) {  // a value we might want later
            if (WG_LANE == 0) s.cache[(size_t)c.node * (s.P + 1) + level] = value;
            r.cacheMask |= 1u << level;
          }
        }
        return;
      }
      if (r.flags & SF_SWAPPING) {
        send_swap_reply(c, s, from, false, level, r.agg);
        return;
      }
      if (is_candidate(s, c.node, from, r.cpl)) transition(c, s, r, value);
      return;
    }
    // SwapReply -> onSwapReply :266-316
    if (level != r.cpl || (r.flags & SF_DONE)) return;
    if (r.flags & SF_SWAPPING) return;
    const int32_t S = block_size(s, r.cpl), cb = own_base(s, c.node, r.cpl) ^ S;
    const bool inBlock = from >= cb && from < cb + S;
    const bool pending = inBlock && row_bit(s.pending + (size_t)c.node * s.W, from - cb);
    if (!(msg & 2u)) {  // OK
      if (!pending) {
        if (inBlock) transition(c, s, r, value);  // unexpected but valid
        return;
      }
      transition(c, s, r, value);
    } else if (pending) {  // NO from a node we asked: try the next ones
      send_to_nodes(c, s, r, pick_next_nodes(c, s, r, r.cpl, s.p.candidateCount));
    }
  }
  __device__ static void on_task(Ctx& c, const State& s, NodeRegs& r, uint32_t wordT, uint32_t arg) {
    if (wordT == SF_TASK_START) {
      go_next_level(c, s, r);
    } else if (wordT == SF_TASK_TIMEOUT) {
      if (!(r.flags & SF_DONE) && r.cpl == (int32_t)arg)
        send_to_nodes(c, s, r, pick_next_nodes(c, s, r, r.cpl, s.p.candidateCount));
    } else {
      r.agg += (int32_t)arg;
      go_next_level(c, s, r);
    }
  }
};

// RunMultipleTimes.contUntilDone-style predicate: some live node has not finished
__global__ void k_sf_cont_if(const EngineDev* __restrict__ tab, const SfState* __restrict__ stab, uint32_t* out) {
  const EngineDev& d = tab[blockIdx.y];
  const SfState& s = stab[blockIdx.y];
  int node = blockIdx.x * blockDim.x + threadIdx.x;
  bool c = node < s.N && shard_owns(d, node) && !d.nodes.down[node] && !(s.flags[node] & SF_DONE);  // (sharded: own nodes)
  if (__ballot(c) && WG_LANE == 0) cont_if_set(out + blockIdx.y);
}
__global__ void k_sf_init(SfState s) {
  int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= s.N) return;
  s.cpl[node] = s.P;  // currentPrefixLength = powerOfTwo :216
  s.agg[node] = 1;    // aggValue = 1 :213
}

}  // namespace wg
