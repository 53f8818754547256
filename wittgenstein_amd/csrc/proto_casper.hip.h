// Casper IMD (P/CasperIMD.java over C/Block.java, C/BlockChainNode.java, C/BlockChainNetwork.java) as a resident device
// protocol. One wavefront per simulated node; lanes split the words of the attestation bitsets.
//
// The reference keeps per node a Map<Long, Set<Attestation>> (attestationsByHead) and per block a
// Map<Integer, Set<Attestation>>, and its fork choice (best / countAttestations :186-266) and block building (:389-434)
// are set algebra over them. Here every Attestation has an index a = height * attestersPerRound + (attester ordinal /
// cycleLength) — an attester votes once per cycle, in the slot (1 + ordinal % cycleLength) of it (:501-507), so the index
// is unique and the attestations of one height are a contiguous index range — and every set is a bitset over that index:
//   recv[node]        the attestations the node has received (the union of its attestationsByHead sets)
//   headMask[b]       attestations whose head is block b          (attestationsByHead.get(b) = recv & headMask[b])
//   attestsMask[b]    attestations a with b in a.hs               (Attestation.attests :113-115)
//   blockAtt[b]       the attestations block b carries            (its attestationsByHeight, all heights)
// so countAttestations / buildBlock are OR / AND / popcount passes over words. Blocks are rows of a table, ids in creation
// order as the reference's Block.blockId (at most one block is created per simulated ms, checked).
// randomOnTies (:250-253, the CasperParemeters() default): the tie's rd.nextBoolean() decides the node's head INSIDE action(),
// so its value cannot be deferred to `resolve` as a send's seed is — and its place in the rd sequence is the number of draws
// of every earlier event of the ms, network-wide. Such a configuration delivers the events of its "mixed" nodes (those
// with a block or a task in the ms — the only events that call best()) by ONE wavefront in global event order
// (k_casper_seq), which knows that number as it goes; attestation-only nodes keep the lane-per-event kernel (an
// attestation draws nothing and commutes). Exact, and as slow as one wavefront is — so only once it can matter: best() reaches
// its tie-break only between two branches, and until some block has a second child (CasperState::forked, set by build_block;
// never, without a byzantine delay) no event can draw in best(): the parallel k_deliver is exact and stays in charge. Both
// paths are enqueued every ms and the flag, on the device, decides which one finds work (k_casper_seq leaves no mixed node
// for k_deliver's visit_skip to admit).
// Not resident: the byzantine producers other than the ByzBlockProducerWF that init() installs (:475-479).
#pragma once
#include "engine_kernels.hip.h"

namespace wg {

constexpr int32_t C_SLOT = 8000;  // SLOT_DURATION :19
constexpr uint32_t C_TASK_PRODUCER = 0, C_TASK_ATTESTER = 1, C_TASK_WF = 2, C_TASK_WF_BUILD = 3;
constexpr uint32_t C_MSG_BLOCK = 0, C_MSG_ATTESTATION = 1;

struct CasperState {
  wg_casper_params p;
  int32_t N, B, A, Aw, Bw;
  int32_t BS;           // words of a node's block record: blkRecv | reeval | headsAtt side by side (3 Bw rounded up to a power of two), so an
                        // attestation's look-ups of the three land in ONE cache line of its receiver instead of three
  GP<int32_t> head;        // [N] block index
  GP<uint64_t> recv;       // [N][Aw]
  GP<uint64_t> blkRecv;    // [N][Bw] blocksReceivedByBlockId
  GP<uint64_t> reeval;     // [N][Bw] blocksToReevaluate
  GP<uint64_t> headsAtt;   // [N][Bw] keys of attestationsByHead
  GP<int32_t> wf;          // ByzBlockProducerWF (node 1): toSend, late, onTime
  GP<int32_t> bHeight, bParent, bProducer, bTime;  // [B]
  GP<uint32_t> nBlocks;    // [1]
  GP<int32_t> lastBlockMs; // [1]
  GP<uint64_t> blockAtt, headMask, attestsMask;     // [B][Aw]
  GP<int32_t> attHead;     // [A]
  GP<uint8_t> mixed;       // [N] this ms the node has an event that is not an attestation (block, task): ordered visit
  uint32_t laneEvents;  // 1: attestation-only nodes are delivered one lane per event (k_casper_attestations)
  // Node-range sharding (Engine::run_ms_sharded): per-node rows are touched by the node's owner only; the block and
  // attestation tables are replicated and filled by exchange — what the action()s of this ms added to them (at most one
  // block, the votes of one height) goes into `xtab`, is summed across shards and applied by k_casper_shard_apply on the
  // shards that did not create it. A table entry written in ms t is first read in a later ms (a block or an attestation
  // is known to a node only once delivered), so an exchange at the end of the ms's delivery pass is early enough.
  //   xtab: [0] blocks created, [1] height, [2] parent, [3] producer, [4] time, [XT_HEAD + 2 w + {0,1}] halves of
  //         blockAtt word w, [XT_HEAD + 2 Aw + j] head + 1 of the vote with attestation index height * attestersPerRound + j
  GP<int32_t> xtab;
  GP<uint32_t> anyTask;    // [1] this ms holds an event that is not an attestation (replicated: the exchange is due)
  GP<uint64_t> seqBits;    // randomOnTies: [maxEvents / 64] bit e = event e belongs to a mixed node (k_casper_mark -> k_casper_seq)
  // two blocks in one ms (valid in the reference: a delayed byzantine build landing on another producer's slot): block ids
  // are creation order, which two wavefronts of one launch do not have — in a run whose byzantine producer has a delay
  // (`seqCapable`) k_casper_builds counts the events of the ms that can build a block (a producer's task; a block arriving at
  // the byzantine producer, whose onBlock builds at once when it is late), and from two on the ms goes through k_casper_seq
  GP<uint32_t> builds;     // [1] (reset by k_casper_seq)
  uint32_t seqCapable;
  // randomOnTies on a sharded engine (k_casper_mark_shard / k_casper_seq_shard, CasperHost::shard_deliver): the events
  // that are not attestations are the same on every shard (`tBits`); `seqBits` holds this shard's share of the ordered
  // visit (its own nodes' such events and the attestations of its nodes that have one); `seqPos` = how far that share is
  // done; `xseq` = {next cursor + 1, draws made} of a round, summed across shards (only one shard writes it)
  GP<uint64_t> tBits;
  GP<uint32_t> seqPos;
  GP<int32_t> xseq;
  GP<uint32_t> forked;     // randomOnTies: [1] some block has two children. best() can reach its tie-break only between two
                           // branches, i.e. never before that: until then the parallel k_deliver is exact (no draw to order),
                           // from the next ms on k_casper_seq takes the mixed nodes
};
constexpr int XT_HEAD = 8;

struct CasperProto {
  typedef CasperState State;
  struct WaveShared {
    int unused;
  };
  struct NodeRegs {
    int32_t head;
    uint32_t drawBase;  // randomOnTies: rd draws of the ms's earlier events, network-wide (k_casper_seq)
  };
  __device__ static int msg_size(const State&, uint32_t) { return 1; }  // Message.size() default
  __device__ static int msg_level(uint32_t) { return 0; }
  __device__ static void node_begin(Ctx& c, const State& s, NodeRegs& r, WaveShared*) {
    r.head = s.head[c.node];
    r.drawBase = 0xFFFFFFFFu;  // (k_casper_seq sets the real one)
  }
  __device__ static void node_end(Ctx& c, const State& s, NodeRegs& r) {
    if (WG_LANE == 0) s.head[c.node] = r.head;
  }

  // k_deliver leaves a node to k_casper_attestations unless the node has a block or a task this ms (and resets the flag)
  __device__ static bool visit_skip(const EngineDev&, const State& s, int32_t node) {
    if (!s.laneEvents) return false;
    const bool mixed = __hip_atomic_load(s.mixed + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __builtin_amdgcn_wave_barrier();
    if (mixed && WG_LANE == 0) s.mixed[node] = 0;
    return !mixed;
  }
  __device__ static uint64_t ldc(const uint64_t WG_G* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ static int32_t ldi(const int32_t WG_G* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ static bool bit(const uint64_t WG_G* row, int32_t i) { return (ldc(row + (i >> 6)) >> (i & 63)) & 1ULL; }
  __device__ static void set_bit(uint64_t WG_G* row, int32_t i) {  // one lane
    row[i >> 6] |= 1ULL << (i & 63);
  }
  // bits [lo, hi) of word w
  __device__ static uint64_t range_mask(int w, int64_t lo, int64_t hi) {
    const int64_t b0 = (int64_t)w << 6, b1 = b0 + 64;
    if (hi <= b0 || lo >= b1 || hi <= lo) return 0ULL;
    uint64_t m = ~0ULL;
    if (lo > b0) m &= ~0ULL << (lo - b0);
    if (hi < b1) m &= (1ULL << (hi - b0)) - 1ULL;
    return m;
  }
  __device__ static bool has_direct_link(const State& s, int32_t a, int32_t b) {  // C/Block.java:86-99
    if (a == b) return true;
    const int32_t ha = ldi(s.bHeight + a), hb = ldi(s.bHeight + b);
    if (ha == hb) return false;
    int32_t older = ha > hb ? a : b, young = ha > hb ? b : a;
    const int32_t hy = ldi(s.bHeight + young);
    while (ldi(s.bHeight + older) > hy) older = ldi(s.bParent + older);
    return older == young;
  }
  // countAttestations(start, h) :241-266
  __device__ static int count_attestations(Ctx& c, const State& s, int32_t start, int32_t h) {
    const uint64_t WG_G* rv = s.recv + (size_t)c.node * s.Aw;
    const int32_t hh = ldi(s.bHeight + h);
    const int64_t pr = s.p.attestersPerRound;
    int cnt = 0;
    for (int w = WG_LANE; w < s.Aw; w += 64) {
      uint64_t x = 0;
      for (int32_t cur = start; cur != h; cur = ldi(s.bParent + cur)) {
        const int32_t ch = ldi(s.bHeight + cur);
        x |= ldc(s.blockAtt + (size_t)cur * s.Aw + w) & range_mask(w, ((int64_t)hh + 1) * pr, (int64_t)ch * pr);
        x |= ldc(rv + w) & ldc(s.headMask + (size_t)cur * s.Aw + w);
      }
      cnt += __popcll(x & ldc(s.attestsMask + (size_t)h * s.Aw + w));
    }
    return wave_sum(cnt);
  }
  __device__ static int32_t best(Ctx& c, const State& s, const NodeRegs& r, int32_t o1, int32_t o2) {  // :186-236
    if (o1 == o2) return o1;
    const int32_t h1 = ldi(s.bHeight + o1), h2 = ldi(s.bHeight + o2);
    if (h1 == h2) {
      if (WG_LANE == 0) set_err(c.d.g, ERR_PROTOCOL);  // "Someone sent two blocks for the same height": IllegalStateException
      return o1;
    }
    if (has_direct_link(s, o1, o2)) return h1 < h2 ? o2 : o1;
    int32_t b1 = o1, b2 = o2;
    while (ldi(s.bParent + b1) != ldi(s.bParent + b2)) {
      if (ldi(s.bHeight + ldi(s.bParent + b1)) > ldi(s.bHeight + ldi(s.bParent + b2)))
        b1 = ldi(s.bParent + b1);
      else
        b2 = ldi(s.bParent + b2);
    }
    const int32_t h = ldi(s.bParent + b1);
    const int v1 = count_attestations(c, s, o1, h), v2 = count_attestations(c, s, o2, h);
    if (v1 > v2) return o1;
    if (v1 < v2) return o2;
    if (s.p.randomOnTies) {  // network.rd.nextBoolean() ? o1 : o2  (:250-253) — only ever reached from k_casper_seq
      // (the parallel k_deliver carries no draw count: it only runs before the chain forks, where no tie can arise — except
      // for the forking producer's own later events in the very ms of its fork: loud, not guessed)
      if (r.drawBase == 0xFFFFFFFFu && WG_LANE == 0) set_err(c.d.g, ERR_PROTOCOL);
      // java.util.Random.nextBoolean() = next(1) != 0: the top bit of the 48-bit state after one more step
      const uint64_t st = lcg_skip(c.d.g->rng, (uint64_t)(r.drawBase + c.draws) + 1);
      c.draws++;
      return ((st >> 47) & 1ULL) ? o1 : o2;
    }
    return b1 >= b2 ? o1 : o2;
  }
  __device__ static void reevaluate_head(Ctx& c, const State& s, NodeRegs& r) {  // :349-356, ascending block id
    uint64_t WG_G* re = s.reeval + (size_t)c.node * s.BS;
    for (int w = 0; w < s.Bw; w++) {
      uint64_t m = ldc(re + w);
      while (m) {
        const int32_t b = (w << 6) + (__ffsll((unsigned long long)m) - 1);
        m &= m - 1;
        r.head = best(c, s, r, r.head, b);
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int w = WG_LANE; w < s.Bw; w += 64) re[w] = 0;
    __threadfence_block();
  }
  // buildBlock(base, height) :389-434 -> the new block's index
  __device__ static int32_t build_block(Ctx& c, const State& s, int32_t base, int32_t height) {
    int32_t idx = 0;
    if (WG_LANE == 0) {
      idx = (int32_t)atomicAdd(F(s.nBlocks + 0), 1u);
      // two blocks in one ms: valid in the reference (e.g. a delayed byzantine build landing on another producer's slot),
      // not resident — block ids are creation order and two wavefronts of one launch have none: its own error
      // (not under k_casper_seq: one wavefront in event order creates them in the reference's order)
      if (atomicExch(F(s.lastBlockMs + 0), c.t) == c.t && !*s.forked && *s.builds < 2u) set_err(c.d.g, ERR_SAME_MS_BLOCKS);
      if (idx >= s.B || height <= 0 || c.t < ldi(s.bTime + base) || ldi(s.bHeight + base) >= height)
        set_err(c.d.g, idx >= s.B ? ERR_PAYLOAD : ERR_PROTOCOL);  // table full / Block's ctor checks :36-47
      if (idx >= s.B) idx = s.B - 1;
      s.bHeight[idx] = height;
      s.bParent[idx] = base;
      s.bProducer[idx] = c.node;
      s.bTime[idx] = c.t;
    }
    idx = __shfl(idx, 0, 64);
    if (s.p.randomOnTies) {  // a second child of `base`: from now on two branches exist (CasperState::forked)
      bool sib = false;
      for (int32_t b = 1 + WG_LANE; b < idx; b += 64) sib |= ldi(s.bParent + b) == base;
      if (__ballot(sib) && WG_LANE == 0) *s.forked = 1u;
    }
    const int cl = s.p.cycleLength;
    const uint64_t WG_G* rv = s.recv + (size_t)c.node * s.Aw;
    for (int w = WG_LANE; w < s.Aw; w += 64) {
      uint64_t all = 0, res = 0;
      for (int32_t cur = base; cur != 0 && ldi(s.bHeight + cur) >= height - cl; cur = ldi(s.bParent + cur))
        all |= ldc(s.blockAtt + (size_t)cur * s.Aw + w);  // phase 1: what the parents' blocks already carry
      for (int32_t cur = base; cur >= 0 && ldi(s.bHeight + cur) >= height - cl; cur = ldi(s.bParent + cur))
        res |= ldc(rv + w) & ldc(s.headMask + (size_t)cur * s.Aw + w);  // phase 2: attestationsByHead of the branch
      res &= range_mask(w, 0, (int64_t)height * s.p.attestersPerRound) & ~all;  // a.height < height, not yet included
      s.blockAtt[(size_t)idx * s.Aw + w] = res;
      if (c.d.sharded) {
        s.xtab[XT_HEAD + 2 * w] = (int32_t)(uint32_t)res;
        s.xtab[XT_HEAD + 2 * w + 1] = (int32_t)(uint32_t)(res >> 32);
      }
    }
    if (c.d.sharded && WG_LANE == 0) {
      atomicAdd(F(s.xtab + 0), 1);
      s.xtab[1] = height;
      s.xtab[2] = base;
      s.xtab[3] = c.node;
      s.xtab[4] = c.t;
    }
    __threadfence_block();
    return idx;
  }
  __device__ static void send_block(Ctx& c, const State& s, int32_t b) {  // sendAll(new SendBlock<>(head), time + blockConstructionTime, this)
    c.send_all(C_MSG_BLOCK, (uint32_t)b, c.t + s.p.blockConstructionTime, 1);
  }
  __device__ static void create_and_send_block(Ctx& c, const State& s, NodeRegs& r, int32_t height) {  // :436-442
    r.head = build_block(c, s, r.head, height);
    send_block(c, s, r.head);
  }
  __device__ static void vote(Ctx& c, const State& s, NodeRegs& r, int32_t height) {  // :453-466 + Attestation() :107-121
    reevaluate_head(c, s, r);
    const int32_t ordinal = c.node - (s.p.blockProducersCount + 1);
    const int64_t a = (int64_t)height * s.p.attestersPerRound + ordinal / s.p.cycleLength;
    if (a >= s.A) {
      if (WG_LANE == 0) set_err(c.d.g, ERR_PAYLOAD);  // wg_casper_params.maxSlots
      return;
    }
    if (WG_LANE == 0) {
      s.attHead[a] = r.head;
      if (c.d.sharded) s.xtab[XT_HEAD + 2 * s.Aw + ordinal / s.p.cycleLength] = r.head + 1;
      atomicOr((unsigned long long*)(s.headMask + (size_t)r.head * s.Aw + (a >> 6)), 1ULL << (a & 63));
      const int32_t hh = ldi(s.bHeight + r.head);
      for (int32_t cur = ldi(s.bParent + r.head); cur >= 0 && ldi(s.bHeight + cur) >= hh - s.p.cycleLength; cur = ldi(s.bParent + cur))
        atomicOr((unsigned long long*)(s.attestsMask + (size_t)cur * s.Aw + (a >> 6)), 1ULL << (a & 63));
    }
    c.send_all(C_MSG_ATTESTATION, (uint32_t)a, c.t + s.p.attestationConstructionTime, 1);
  }
  // BlockChainNode.onBlock :29-47 under CasperNode.onBlock :276-292 (delta >= 0 always: the formula adds the slot time)
  __device__ static bool on_block(Ctx& c, const State& s, NodeRegs& r, int32_t b) {
    uint64_t WG_G* re = s.reeval + (size_t)c.node * s.BS;
    uint64_t WG_G* br = s.blkRecv + (size_t)c.node * s.BS;
    const bool known = bit(br, b);
    __builtin_amdgcn_wave_barrier();
    if (WG_LANE == 0) {
      set_bit(re, r.head);
      set_bit(re, b);
      if (!known) set_bit(br, b);
    }
    __threadfence_block();
    if (known) return false;
    r.head = best(c, s, r, r.head, b);
    return true;
  }
  __device__ static void on_message(Ctx& c, const State& s, NodeRegs& r, int32_t, uint32_t msg, uint32_t payload) {
    if (msg == C_MSG_ATTESTATION) {  // onAttestation :294-337
      const int32_t a = (int32_t)payload;
      const int32_t h = ldi(s.attHead + a);
      const bool haveBlock = bit(s.blkRecv + (size_t)c.node * s.BS, h);
      __builtin_amdgcn_wave_barrier();
      if (WG_LANE == 0) {
        set_bit(s.recv + (size_t)c.node * s.Aw, a);
        set_bit(s.headsAtt + (size_t)c.node * s.BS, h);
        if (haveBlock) set_bit(s.reeval + (size_t)c.node * s.BS, h);
      }
      __threadfence_block();
      return;
    }
    const int32_t b = (int32_t)payload;
    if (!on_block(c, s, r, b)) return;
    if (c.node != 1) return;
    // ByzBlockProducerWF.onBlock :651-683
    const int32_t toSend = ldi(s.wf + 0);
    const int32_t bh = ldi(s.bHeight + b);
    __builtin_amdgcn_wave_barrier();  // every lane has read toSend before lane 0 replaces it
    if (bh != toSend - 1) return;
    const int32_t perfectDate = C_SLOT * toSend + s.p.byzDelay;
    if (WG_LANE == 0) s.wf[0] = toSend + s.p.blockProducersCount;
    if (c.t >= perfectDate) {
      r.head = build_block(c, s, b, toSend);
      send_block(c, s, r.head);
      if (WG_LANE == 0) s.wf[1]++;
    } else {
      c.register_task(perfectDate, C_TASK_WF_BUILD, (uint32_t)b);
      if (WG_LANE == 0) s.wf[2]++;
    }
    __threadfence_block();
  }
  __device__ static void on_task(Ctx& c, const State& s, NodeRegs& r, uint32_t word, uint32_t arg) {
    if (word == C_TASK_PRODUCER) {  // BlockProducer.periodicTask :381-386
      reevaluate_head(c, s, r);
      create_and_send_block(c, s, r, c.t / C_SLOT);
    } else if (word == C_TASK_ATTESTER) {
      vote(c, s, r, c.t / C_SLOT);
    } else if (word == C_TASK_WF) {  // ByzBlockProducerWF.periodicTask :640-649
      const int32_t toSend = ldi(s.wf + 0);
      __builtin_amdgcn_wave_barrier();
      if (r.head == 0 && toSend == 1) {
        reevaluate_head(c, s, r);  // reevaluateH :529-543
        while (ldi(s.bHeight + r.head) >= toSend) r.head = ldi(s.bParent + r.head);
        const int32_t h = (c.t - s.p.byzDelay) / C_SLOT;
        if (h != toSend) {
          if (WG_LANE == 0) set_err(c.d.g, ERR_PROTOCOL);
          return;
        }
        create_and_send_block(c, s, r, h);
        if (WG_LANE == 0) s.wf[0] = toSend + s.p.blockProducersCount;
        __threadfence_block();
      }
    } else {  // the Runnable of ByzBlockProducerWF.onBlock :657-669
      const int32_t b = (int32_t)arg;
      r.head = build_block(c, s, b, ldi(s.bHeight + b) + 1);
      send_block(c, s, r.head);
    }
  }
};

// An attestation delivery (onAttestation :294-337) is three bit-sets and two counters, and it commutes with the node's
// other attestation deliveries — and with a block delivery of the same ms (whichever comes first, the head's bit ends up
// in blocksToReevaluate: onBlock adds the block itself). Only relative to the node's own tasks does the order show (what
// a vote or a block built in that ms includes). So: nodes with a block or a task this ms are flagged and visited in event
// order by k_deliver; for all the others one lane per EVENT applies the delivery with atomics — at 4096 attesters per
// slot a node receives a dozen attestations per ms, which one wavefront per node would apply one after the other.
__device__ __forceinline__ bool casper_is_attestation(const Rec& r) { return rec_kind(r) == K_MSG && r.w2 == C_MSG_ATTESTATION; }
__global__ void __launch_bounds__(256) k_casper_classify(const EngineDev* __restrict__ tab, const CasperState* __restrict__ stab) {
  WG_ENGINE(tab);
  const CasperState& s = stab[wgBy];
  const uint32_t n = d.g->nEvents;
  for (uint32_t e = wgBx * blockDim.x + threadIdx.x; e < n; e += wgGx * blockDim.x) {
    const Rec r = d.ev[e];
    if (casper_is_attestation(r)) continue;
    if (d.sharded) {  // (the event list is replicated: every shard sees that the table exchange is due; flags for its own nodes)
      *s.anyTask = 1;
      if (!shard_owns(d, (int32_t)r.w1)) continue;
    }
    s.mixed[r.w1] = 1;
  }
}
__global__ void __launch_bounds__(256) k_casper_attestations(const EngineDev* __restrict__ tab, const CasperState* __restrict__ stab) {
  WG_ENGINE(tab);
  const CasperState& s = stab[wgBy];
  const uint32_t n = d.g->nEvents;
  for (uint32_t e = wgBx * blockDim.x + threadIdx.x; e < n; e += wgGx * blockDim.x) {
    const Rec r = d.ev[e];
    if (!casper_is_attestation(r)) continue;
    const int32_t to = (int32_t)r.w1, from = rec_from(r);
    if (d.sharded && !shard_owns(d, to)) {  // another shard's node: zeros (summed across shards before `order`)
      d.evRes[e] = EvRes{0u, 0u};
      continue;
    }
    if (s.mixed[to]) {  // k_deliver applies this node's events in order: the attestation joins the node's inbox list now
      // (expand did not thread it, ExpandF::lane_only; the node is on the active list through its other event)
      const int32_t prev = atomicExch(F(&d.head[to]), (int32_t)e);
      d.evNext[e] = prev;
      if (prev < 0) set_err(d.g, ERR_PROTOCOL);
      continue;
    }
    const EvAux aux = gld(d.evAux + e);
    EvRes res;
    res.nrec = 0;
    res.ndraw = 0;
    if (!d.nodes.down[to] && (d.nparts == 0 || d.nodes.part[from] == d.nodes.part[to])) {  // C/Network.java:606
      const int32_t a = (int32_t)r.w3, h = s.attHead[a];
      // (every Casper message has size() 1, C/messages/Message.java:19: bytesReceived == msgReceived for every node — kept
      // once; the host reads msgReceived for both, CasperHost::unit_message_size)
      atomicAdd((unsigned long long*)&d.nodes.msgReceived[to], 1ULL);
      atomicOr((unsigned long long*)(s.recv + (size_t)to * s.Aw + (a >> 6)), 1ULL << (a & 63));
      // (the votes of a slot mostly share their head: after a node's first one the bit is set, and a plain load leaves
      // the line clean — the atomics' dirty lines are what this kernel pays for, profiles/r02p_casper_pmc_WRITE_SIZE.md)
      if (!((CasperProto::ldc(s.headsAtt + (size_t)to * s.BS + (h >> 6)) >> (h & 63)) & 1ULL))
        atomicOr((unsigned long long*)(s.headsAtt + (size_t)to * s.BS + (h >> 6)), 1ULL << (h & 63));
      if (((s.blkRecv[(size_t)to * s.BS + (h >> 6)] >> (h & 63)) & 1ULL) &&
          !((CasperProto::ldc(s.reeval + (size_t)to * s.BS + (h >> 6)) >> (h & 63)) & 1ULL))  // (as headsAtt: set once, then clean)
        atomicOr((unsigned long long*)(s.reeval + (size_t)to * s.BS + (h >> 6)), 1ULL << (h & 63));
      res.nrec = EV_DELIVERED;
    }
    if (aux.chain >= 0 && aux.cpos < 0) {  // last hop of the run: markRead(); if (hasNextReader()) msgs.addMsg(m)  :629-632
      const int32_t next = (aux.cpos & 0x7FFFFFFF) + 1;
      if (next < d.chains[aux.chain].ndest) {
        Out o;
        o.kindfrom = (O_CHAINCONT << 28) | (uint32_t)to;
        o.to = aux.chain;
        o.a = (uint32_t)next;
        o.b = 0;
        o.t = 0;
        o.destOff = 0;
        o.drawsub = 0;
        o.pad = 0;
        if (aux.outCap && aux.outBase < d.maxOut) {  // (as Ctx::put: an over-subscribed ms must not write past the outbox)
          d.outTmp[aux.outBase] = o;
          res.nrec |= 1u;
        } else {
          set_err(d.g, ERR_OUTBOX);
        }
      } else {
        d.chains[aux.chain].flags = 0;
      }
    }
    d.evRes[e] = res;
  }
}

// events of this ms that can create a block (CasperState::builds): a task of a block producer (nodes 1 .. blockProducersCount;
// BlockProducer.periodicTask :381-386, ByzBlockProducerWF.periodicTask :640-649 and its delayed build :667-676), and a
// block arriving at the byzantine producer (node 1), whose onBlock builds on the spot when it is late (:671-673)
__global__ void __launch_bounds__(256) k_casper_builds(const EngineDev* __restrict__ tab, const CasperState* __restrict__ stab) {
  WG_ENGINE(tab);
  const CasperState& s = stab[wgBy];
  const uint32_t n = d.g->nEvents;
  for (uint32_t e0 = (wgBx * blockDim.x + threadIdx.x) & ~63u; e0 < n; e0 += wgGx * blockDim.x) {
    const uint32_t e = e0 + WG_LANE;
    bool b = false;
    if (e < n) {
      const Rec r = d.ev[e];
      const int32_t to = (int32_t)r.w1;
      const uint32_t k = rec_kind(r);
      b = to >= 1 && to <= s.p.blockProducersCount &&
          ((k == K_TASK || k == K_PERIODIC) || (k == K_MSG && to == 1 && r.w2 == C_MSG_BLOCK));
    }
    const uint64_t m = __ballot(b);
    if (m && WG_LANE == 0) atomicAdd(F(s.builds + 0), (uint32_t)__popcll(m));
  }
}

// one event of an ordered visit (k_casper_seq, k_casper_seq_shard): delivered as k_deliver delivers it, with the number of rd
// draws of the ms's earlier events; returns the draws it made
__device__ __forceinline__ uint32_t casper_seq_event(const EngineDev& d, const CasperState& s, CasperProto::WaveShared* shP, uint32_t e,
                                                     int32_t t, uint32_t drawBase) {
  const int lane = WG_LANE;
  const Rec rec = d.ev[e];
  const EvAux aux = d.evAux[e];
  const int32_t node = (int32_t)rec.w1;
  Ctx c{d, t, node, 0, 0, 0, 0, 0, 0, 0};
  CasperProto::NodeRegs r;
  CasperProto::node_begin(c, s, r, shP);
  r.drawBase = drawBase;
  long long nRecv = 0, bRecv = 0;
  deliver_event<CasperProto>(d, s, c, r, e, rec, aux, d.nodes.down[node] != 0, d.nparts ? d.nodes.part[node] : (uint8_t)0, true, nRecv,
                             bRecv);
  __builtin_amdgcn_wave_barrier();
  CasperProto::node_end(c, s, r);
  if (lane == 0) {
    if (nRecv) atomicAdd((unsigned long long*)&d.nodes.msgReceived[node], (unsigned long long)nRecv);  // (unit_message_size)
    if (c.msgSent) {
      atomicAdd((unsigned long long*)&d.nodes.msgSent[node], (unsigned long long)c.msgSent);
      atomicAdd((unsigned long long*)&d.nodes.bytesSent[node], (unsigned long long)c.bytesSent);
    }
    d.head[node] = -1;  // (its inbox list, threaded by expand / k_casper_attestations, is not used here)
    s.mixed[node] = 0;
  }
  return c.draws;
}

// randomOnTies (see the header): which events belong to a mixed node, one bit per event (64 consecutive events per
// wavefront: one ballot, one store), then ONE wavefront delivers exactly those in global event order — receiveUntil's
// own order (C/Network.java:594-635) — carrying the number of rd draws made so far in the ms, which is the index of a
// tie's nextBoolean() in the rd sequence. Results, counters and records as k_deliver leaves them.
__global__ void __launch_bounds__(256) k_casper_mark(const EngineDev* __restrict__ tab, const CasperState* __restrict__ stab) {
  WG_ENGINE(tab);
  const CasperState& s = stab[wgBy];
  if (!*s.forked && *s.builds < 2u) return;  // one chain so far: no tie-break can be reached, k_deliver takes the mixed nodes in parallel
  const uint32_t n = d.g->nEvents;
  for (uint32_t e0 = (wgBx * blockDim.x + threadIdx.x) & ~63u; e0 < n; e0 += wgGx * blockDim.x) {
    const uint32_t e = e0 + WG_LANE;
    bool mix = false;
    if (e < n) {
      const Rec r = d.ev[e];
      mix = !s.laneEvents || !casper_is_attestation(r) || s.mixed[r.w1] != 0;
    }
    const uint64_t m = __ballot(mix);
    if (WG_LANE == 0) s.seqBits[e0 >> 6] = m;
  }
}
__global__ void __launch_bounds__(64) k_casper_seq(const EngineDev* __restrict__ tab, const CasperState* __restrict__ stab) {
  WG_ENGINE(tab);
  const CasperState& s = stab[wgBy];
  if (!*s.forked && *s.builds < 2u) {
    if (WG_LANE == 0) *s.builds = 0;
    return;
  }
  __shared__ CasperProto::WaveShared shP;
  const uint32_t n = d.g->nEvents;
  const int32_t t = d.g->now;
  const int lane = WG_LANE;
  uint32_t drawBase = 0;
  for (uint32_t w0 = 0; w0 * 64 < n; w0 += 64) {  // 64 words = 4096 events per round
    const uint32_t wi = w0 + (uint32_t)lane;
    const uint64_t mine = (uint64_t)wi * 64 < n ? s.seqBits[wi] : 0ULL;
    uint64_t any = __ballot(mine != 0);
    while (any) {
      const int k = __ffsll((unsigned long long)any) - 1;
      any &= any - 1;
      uint64_t bits = lane_bcast64(mine, k);
      while (bits) {
        const uint32_t e = (w0 + (uint32_t)k) * 64 + (uint32_t)(__ffsll((unsigned long long)bits) - 1);
        bits &= bits - 1;
        drawBase += casper_seq_event(d, s, &shP, e, t, drawBase);
        __threadfence_block();  // the node's next event (another round of this loop) reads what this one wrote
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (lane == 0) *s.builds = 0;
}

// randomOnTies on a SHARDED engine. A tie's nextBoolean() takes its index in the rd sequence from the draws of every earlier
// event of the ms, and those events are other shards' too: the ordered visit goes round the shards. The events that are not
// attestations (blocks, tasks: the only ones that draw) are known to every shard from the replicated event list; a ROUND
// belongs to the owner of the first of them at or behind the cursor: it delivers its share of the visit — its nodes' such
// events and the attestations of its nodes that have one, in event order — up to the next such event of another shard,
// and reports {that event's index + 1, the draws it made}; the sum across shards (nobody else wrote) moves every shard's
// cursor and draw count. A round that finds no such event is the last one: every shard finishes its share (attestations
// draw nothing). Exact, and as slow as one wavefront and one small collective per change of owner are.
__global__ void __launch_bounds__(256) k_casper_mark_shard(const EngineDev* __restrict__ tab, const CasperState* __restrict__ stab) {
  WG_ENGINE(tab);
  const CasperState& s = stab[wgBy];
  const uint32_t n = d.g->nEvents;
  if (wgBx == 0 && threadIdx.x == 0) *s.seqPos = 0;
  for (uint32_t e0 = (wgBx * blockDim.x + threadIdx.x) & ~63u; e0 < n; e0 += wgGx * blockDim.x) {
    const uint32_t e = e0 + WG_LANE;
    bool tt = false, own = false;
    if (e < n) {
      const Rec r = d.ev[e];
      tt = !casper_is_attestation(r);
      own = shard_owns(d, (int32_t)r.w1) && (tt || s.mixed[r.w1] != 0);
    }
    const uint64_t mt = __ballot(tt), mo = __ballot(own);
    if (WG_LANE == 0) {
      s.tBits[e0 >> 6] = mt;
      s.seqBits[e0 >> 6] = mo;
    }
  }
}
// first set bit of `bits` at or behind `from` (< n), or n — the whole wavefront, 64 words a step
__device__ __forceinline__ uint32_t casper_next_bit(const uint64_t WG_G* bits, uint32_t from, uint32_t n) {
  for (uint32_t w0 = from >> 6; (uint64_t)w0 * 64 < n; w0 += 64) {
    const uint32_t wi = w0 + (uint32_t)WG_LANE;
    uint64_t v = (uint64_t)wi * 64 < n ? bits[wi] : 0ULL;
    if (wi == (from >> 6)) v &= ~0ULL << (from & 63);
    const uint64_t any = __ballot(v != 0);
    if (any) {
      const int k = __ffsll((unsigned long long)any) - 1;
      const uint64_t word = lane_bcast64(v, k);
      const uint32_t e = (w0 + (uint32_t)k) * 64 + (uint32_t)(__ffsll((unsigned long long)word) - 1);
      return e < n ? e : n;
    }
  }
  return n;
}
__global__ void __launch_bounds__(64) k_casper_seq_shard(const EngineDev* __restrict__ tab, const CasperState* __restrict__ stab,
                                                         uint32_t cursor, uint32_t drawBase0) {
  WG_ENGINE(tab);
  const CasperState& s = stab[wgBy];
  __shared__ CasperProto::WaveShared shP;
  const uint32_t n = d.g->nEvents;
  const int32_t t = d.g->now;
  const int lane = WG_LANE;
  const uint32_t e0 = casper_next_bit(s.tBits, cursor, n);
  uint32_t stop = n;  // this shard's share is delivered up to here
  if (e0 < n) {
    if (!shard_owns(d, (int32_t)d.ev[e0].w1)) return;  // another shard's round
    for (uint32_t e = e0;;) {  // ... up to the next such event that is another shard's
      e = casper_next_bit(s.tBits, e + 1, n);
      if (e >= n) break;
      if (!shard_owns(d, (int32_t)d.ev[e].w1)) {
        stop = e;
        break;
      }
    }
  }
  uint32_t drawBase = drawBase0;
  for (uint32_t e = casper_next_bit(s.seqBits, *s.seqPos, stop); e < stop; e = casper_next_bit(s.seqBits, e + 1, stop)) {
    drawBase += casper_seq_event(d, s, &shP, e, t, drawBase);
    __threadfence_block();  // the node's next event (another round of this loop) reads what this one wrote
    __builtin_amdgcn_wave_barrier();
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    *s.seqPos = stop;
    if (e0 < n) {
      s.xseq[0] = (int32_t)stop + 1;
      s.xseq[1] = (int32_t)(drawBase - drawBase0);
    }
  }
}

// sharded engine: the summed table image of this ms (CasperState::xtab) -> the block and the votes the other shards'
// nodes created, on every shard; the image is left zeroed for the next ms
__global__ void __launch_bounds__(256) k_casper_shard_apply(const EngineDev* __restrict__ tab, const CasperState* __restrict__ stab) {
  WG_ENGINE(tab);
  const CasperState& s = stab[wgBy];
  const int32_t t = d.g->now;
  const int32_t made = s.xtab[0], height = s.xtab[1], parent = s.xtab[2], producer = s.xtab[3], when = s.xtab[4];
  const bool mine = made == 1 && shard_owns(d, producer);
  const int32_t idx = made == 1 ? (int32_t)*s.nBlocks - (mine ? 1 : 0) : 0;  // (the creator's shard has counted it already)
  __syncthreads();
  if (made > 1 && threadIdx.x == 0) set_err(d.g, ERR_SAME_MS_BLOCKS);  // two shards built a block in this ms
  if (made == 1 && !mine) {
    if (idx >= s.B) {
      if (threadIdx.x == 0) set_err(d.g, ERR_PAYLOAD);
    } else {
      for (int w = threadIdx.x; w < s.Aw; w += blockDim.x)
        s.blockAtt[(size_t)idx * s.Aw + w] = (uint64_t)(uint32_t)s.xtab[XT_HEAD + 2 * w] | ((uint64_t)(uint32_t)s.xtab[XT_HEAD + 2 * w + 1] << 32);
      if (threadIdx.x == 0) {
        s.bHeight[idx] = height;
        s.bParent[idx] = parent;
        s.bProducer[idx] = producer;
        s.bTime[idx] = when;
        *s.nBlocks = (uint32_t)idx + 1u;
        *s.lastBlockMs = t;
      }
    }
  }
  const int32_t pr = s.p.attestersPerRound, slot = t / C_SLOT;
  for (int j = threadIdx.x; j < pr; j += blockDim.x) {
    const int32_t v = s.xtab[XT_HEAD + 2 * s.Aw + j];
    if (!v) continue;
    const int32_t head = v - 1;
    const int64_t a = (int64_t)slot * pr + j;
    // (Attestation() :107-121 as in CasperProto::vote; idempotent on the voter's own shard)
    s.attHead[a] = head;
    atomicOr((unsigned long long*)(s.headMask + (size_t)head * s.Aw + (a >> 6)), 1ULL << (a & 63));
    const int32_t hh = s.bHeight[head];
    for (int32_t cur = s.bParent[head]; cur >= 0 && s.bHeight[cur] >= hh - s.p.cycleLength; cur = s.bParent[cur])
      atomicOr((unsigned long long*)(s.attestsMask + (size_t)cur * s.Aw + (a >> 6)), 1ULL << (a & 63));
  }
  __syncthreads();
  for (int k = threadIdx.x; k < XT_HEAD + 2 * s.Aw + pr; k += blockDim.x) s.xtab[k] = 0;
  if (threadIdx.x == 0) *s.anyTask = 0;
}

__global__ void k_casper_init(CasperState s, int32_t lo, int32_t hi) {  // [lo, hi): the nodes whose rows this engine holds
  int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= s.N) return;
  if (node >= lo && node < hi)
    s.blkRecv[(size_t)node * s.BS] = 1ULL;  // blocksReceivedByBlockId.put(genesis.id, genesis)  C/BlockChainNode.java:21-26
  if (node == 0) {
    s.bParent[0] = -1;  // genesis: Block(0)  C/Block.java:22-30
    *s.nBlocks = 1;
    *s.lastBlockMs = -1;
    s.wf[0] = 1;  // toSend = 1 :512
  }
}

}  // namespace wg
