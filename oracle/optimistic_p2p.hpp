// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.OptimisticP2PSignature (P/OptimisticP2PSignature.java:25-193) over core.P2PNetwork
// (p2pflood.hpp: C/P2PNetwork.java with minimum == false, :83). Pinned against PT/OptimisticP2PSignatureTest.java:14-50
// (testSimple: every node done with more than half of the signatures; testCopy: two copies agree) in
// tests/test_oracle_protocols.py; seed-dependent trajectories are unpinned (no JVM in the image).
#pragma once
#include "p2pflood.hpp"

namespace orc {

class OptimisticP2PSignature {  // P/OptimisticP2PSignature.java
 public:
  struct Params {  // OptimisticP2PSignatureParameters :33-72, ctor order
    int nodeCount = 100, threshold = 99, connectionCount = 20, pairingTime = 1;
    std::string nodeBuilderName, networkLatencyName;
  };
  struct P2PSigNode;
  struct SendSig : Message, std::enable_shared_from_this<SendSig> {  // :86-103
    const int sig;
    explicit SendSig(int who) : sig(who) {}
    int size() const override { return 4 + 48; }  // NodeId + sig
    void action(Network&, Node* from, Node* to) override;
  };
  struct P2PSigNode : P2PNode {  // :105-156
    OptimisticP2PSignature& p;
    BitSet verifiedSignatures;
    bool done = false;
    explicit P2PSigNode(OptimisticP2PSignature& pp) : P2PNode(pp.network_.rd, *pp.nb, false), p(pp) {}
    void onSig(P2PSigNode* from, std::shared_ptr<SendSig> ss) {  // :114-133
      if (done || verifiedSignatures.get(ss->sig)) return;
      verifiedSignatures.set(ss->sig);
      std::vector<Node*> dests;
      for (P2PNode* n : peers)
        if (n != from) dests.push_back(n);
      p.network_.send(ss, p.network_.time + 1, this, dests);
      if (verifiedSignatures.cardinality() >= p.params.threshold) {
        done = true;
        doneAt = p.network_.time + p.params.pairingTime * 2;
      }
    }
  };
  Params params;
  P2PNetwork network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<P2PSigNode>> nodes;
  explicit OptimisticP2PSignature(const Params& pr) : params(pr), network_(pr.connectionCount, false) {  // :74-80
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
  }
  void init() {  // :158-167
    for (int i = 0; i < params.nodeCount; i++) {
      nodes.push_back(std::make_unique<P2PSigNode>(*this));
      P2PSigNode* n = nodes.back().get();
      network_.addNode(n);
      network_.registerTask([n] { n->onSig(n, std::make_shared<SendSig>(n->nodeId)); }, 1, n);
    }
    network_.setPeers();
  }
};
inline void OptimisticP2PSignature::SendSig::action(Network&, Node* from, Node* to) {
  static_cast<P2PSigNode*>(to)->onSig(static_cast<P2PSigNode*>(from), shared_from_this());
}

}  // namespace orc
