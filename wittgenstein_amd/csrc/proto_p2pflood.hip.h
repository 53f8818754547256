// P2PFlood (P/P2PFlood.java) over core.P2PNetwork / core.messages.FloodMessage (C/P2PNetwork.java, C/messages/
// FloodMessage.java) as a resident device protocol. The peer graph is built by the host (setPeers draws from the shared
// rd, :27-56) and uploaded; FloodMessage.action (:47-55) runs here: on first receipt, onFlood, then the peers other than
// the sender, shuffled with rd, as one MultipleDestWithDelayEnvelope — Ctx::send_list_delayed_shuffled: the shuffle's
// draws and the explicit arrivals are resolved in `resolve` (shuffle_dests, resolve_multi with OUT_DELAYED).
// The msgCount messages are distinct objects with the same msgId (-1): a node's received set is a bit per message.
#pragma once
#include "engine_kernels.hip.h"

namespace wg {

struct FloodState {
  wg_p2pflood_params p;
  int32_t N, maxPeers;
  GP<int32_t> peers;      // [N][maxPeers] P2PNode.peers in list order
  GP<int32_t> peerCnt;    // [N]
  GP<uint64_t> received;  // [N] getMsgReceived(-1) as a bit per message
};

struct FloodProto {
  typedef FloodState State;
  struct WaveShared {
    int unused;
  };
  struct NodeRegs {};
  __device__ static int msg_size(const State&, uint32_t) { return 1; }  // new FloodMessage<>(1, ...)  P/P2PFlood.java:131
  __device__ static int msg_level(uint32_t) { return 0; }
  __device__ static void node_begin(Ctx&, const State&, NodeRegs&, WaveShared*) {}
  __device__ static void node_end(Ctx&, const State&, NodeRegs&) {}
  __device__ static void on_message(Ctx& c, const State& s, NodeRegs&, int32_t from, uint32_t msg, uint32_t) {
    const int32_t node = c.node;
    const uint64_t have = __hip_atomic_load(s.received + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_wave_barrier();
    if ((have >> msg) & 1ULL) return;  // addToReceived(to) == false
    const uint64_t now = have | (1ULL << msg);
    if (WG_LANE == 0) {
      s.received[node] = now;
      if (__popcll(now) == s.p.msgCount) c.d.nodes.doneAt[node] = c.t;  // P2PFloodNode.onFlood :34-38
    }
    // dest = to.peers.stream().filter(n -> n != from): list order kept
    const int32_t cnt = s.peerCnt[node];
    const int32_t mine = (int)WG_LANE < cnt ? s.peers[(size_t)node * s.maxPeers + WG_LANE] : -1;
    const uint64_t keep = __ballot((int)WG_LANE < cnt && mine != from);
    const int n = __popcll(keep);
    const uint32_t destOff = c.dest_reserve(2 * (n > 0 ? n : 1));
    if ((keep >> WG_LANE) & 1ULL) c.dest_put(destOff, __popcll(keep & lanes_lt()), mine);
    __threadfence_block();
    c.send_list_delayed_shuffled(destOff, n, msg, 0, c.t + 1 + s.p.delayBeforeResent, s.p.delayBetweenSends, 1);
  }
  __device__ static void on_task(Ctx&, const State&, NodeRegs&, uint32_t, uint32_t) {}
};

__global__ void k_flood_init(FloodState s, NodeArrays nd, const int32_t* senders, int nSenders) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nSenders) return;
  // sendPeers: msg.addToReceived(from); init(): if (msgCount == 1) from.doneAt = 1  (P/P2PFlood.java:126-138)
  atomicOr((unsigned long long*)(s.received + senders[k]), 1ULL << k);
  if (s.p.msgCount == 1) nd.doneAt[senders[k]] = 1;
}

}  // namespace wg
