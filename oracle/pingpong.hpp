// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.PingPong (P/PingPong.java:7-102).
#pragma once
#include "network.hpp"

namespace orc {

class PingPong {
 public:
  struct PingPongParameters {  // :34-50
    int nodeCt = 1000;
    std::string nodeBuilderName, networkLatencyName;
  };
  struct PingPongNode : Node {  // :60-74
    PingPong& p;
    int pong = 0;
    explicit PingPongNode(PingPong& pp) : Node(pp.network_.rd, *pp.nb), p(pp) {}
    void onPing(PingPongNode* from);
    void onPong() { pong++; }
  };
  struct Ping : Message {  // :20-25
    void action(Network&, Node* from, Node* to) override {
      static_cast<PingPongNode*>(to)->onPing(static_cast<PingPongNode*>(from));
    }
  };
  struct Pong : Message {  // :27-32
    void action(Network&, Node*, Node* to) override { static_cast<PingPongNode*>(to)->onPong(); }
  };

  PingPongParameters params;
  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<PingPongNode>> nodes;

  explicit PingPong(const PingPongParameters& pr) : params(pr) {  // :52-57
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
  }
  void init() {  // :81-87
    for (int i = 0; i < params.nodeCt; i++) {
      nodes.push_back(std::make_unique<PingPongNode>(*this));
      network_.addNode(nodes.back().get());
    }
    network_.sendAll(std::make_shared<Ping>(), network_.getNodeById(0));
  }
  Network& network() { return network_; }
};

inline void PingPong::PingPongNode::onPing(PingPongNode* from) {  // :67-69
  p.network_.send(std::make_shared<Pong>(), this, from);
}

}  // namespace orc
