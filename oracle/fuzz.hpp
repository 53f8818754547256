// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// NOT a reference protocol: a deterministic stress protocol for the scheduler itself (core.Network's send family,
// multi-destination envelopes with and without delays, tasks, periodic and conditional tasks, rd use inside
// action(), message sizes), written once here against the oracle's Network and once in tests/fuzz_protocol.py
// against wittgenstein_amd.hostnet (the engine's host-callback mode). Every action() derives what it does from a
// per-node hash, so any deviation of the engine's delivery order, latency or rd stream diverges the hashes at once.
#pragma once
#include "network.hpp"

namespace orc {

class Fuzz {
 public:
  static uint32_t mix(uint32_t h, uint32_t x) { return h ^ (x + 0x9e3779b9u + (h << 6) + (h >> 2)); }

  struct FuzzNode : Node {
    Fuzz& p;
    uint32_t h;
    int c = 0;
    explicit FuzzNode(Fuzz& pp) : Node(pp.network_.rd, *pp.nb), p(pp), h((uint32_t)nodeId * 2654435761u) {}
  };
  struct Msg : Message {
    Fuzz& p;
    uint32_t v;
    int ttl;
    Msg(Fuzz& pp, uint32_t vv, int t) : p(pp), v(vv), ttl(t) {}
    int size() const override { return 1 + (int)(v % 5u); }
    void action(Network&, Node* from, Node* to) override { p.onMsg(static_cast<FuzzNode*>(from), static_cast<FuzzNode*>(to), *this); }
  };

  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<FuzzNode>> nodes;
  int N, ttl0;

  Fuzz(int n, int ttl, const std::string& nl) : N(n), ttl0(ttl) {
    nb = nodeBuilderByName("");
    network_.setNetworkLatency(networkLatencyByName(nl));
  }
  FuzzNode* node(uint32_t i) { return nodes[i % (uint32_t)N].get(); }
  std::shared_ptr<Message> msg(uint32_t v, int ttl) { return std::make_shared<Msg>(*this, v, ttl); }

  void init() {
    for (int i = 0; i < N; i++) {
      nodes.push_back(std::make_unique<FuzzNode>(*this));
      network_.addNode(nodes.back().get());
    }
    network_.sendAll(msg(1, ttl0), node(0));
    network_.send(msg(2, ttl0), 3, node(1), {node(2), node(3), node(2), node(5)}, 4);
    FuzzNode* pn = node(1);
    network_.registerPeriodicTask([this, pn] { pn->h = mix(pn->h, (uint32_t)network_.time); network_.send(msg(pn->h, 2), pn, node(pn->h >> 5)); },
                                  7, 13, pn, [pn] { return pn->c < 400; });
    FuzzNode* cn = node(2);
    network_.registerConditionalTask([this, cn] { cn->h = mix(cn->h, 0xC0DEu); network_.send(msg(cn->h, 1), cn, node(cn->h >> 7)); },
                                     5, 9, cn, [cn] { return cn->c % 3 == 0; }, [cn] { return cn->c < 300; });
  }

  void onMsg(FuzzNode* from, FuzzNode* to, const Msg& m) {
    Network& net = network_;
    to->h = mix(mix(mix(to->h, m.v), (uint32_t)from->nodeId), (uint32_t)net.time);
    to->c++;
    if (m.ttl <= 0) return;
    const uint32_t r = to->h;
    const int t = m.ttl - 1;
    switch (r % 8u) {
      case 0:
        if (((r >> 20) % 16u) == 0) net.sendAll(msg(r, t > 2 ? 2 : t), to);  // a mid-run sendAll: N destinations, no delays
        break;
      case 1: break;
      case 2: net.send(msg(r, t), to, node(r >> 3)); break;
      case 3: {
        std::vector<Node*> d;
        for (uint32_t j = 0; j < 2u + ((r >> 3) % 5u); j++) d.push_back(node((uint32_t)to->nodeId + 1u + ((r >> (6 + j)) % (uint32_t)(N - 1))));
        net.send(msg(r, t), to, d);
        break;
      }
      case 4: {
        std::vector<Node*> d;
        for (uint32_t j = 0; j < 2u + ((r >> 3) % 4u); j++) d.push_back(node(r >> (5 + 2 * j)));
        net.send(msg(r, t), net.time + 1 + (int)((r >> 12) % 3u), to, d, 1 + (int)((r >> 8) % 7u));
        break;
      }
      case 5: {
        FuzzNode* n = to;
        const uint32_t rv = r;
        net.registerTask([this, n, rv, t] {
          n->h = mix(n->h, 77u);
          if (rv & 16u) network_.send(msg(n->h, t), n, node(n->h >> 4));
        }, net.time + 1 + (int)((r >> 3) % 20u), to);
        break;
      }
      case 6: {
        const int x = net.rd.nextInt(10);
        net.send(msg(r ^ (uint32_t)x, t), to, node((uint32_t)to->nodeId + (uint32_t)x));
        break;
      }
      default: net.sendArriveAt(msg(r, t), net.time + 1 + (int)((r >> 3) % 5u), to, node(r >> 9));
    }
  }
};

}  // namespace orc
