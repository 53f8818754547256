// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of core.P2PNetwork (C/P2PNetwork.java:9-133), core.P2PNode (C/P2PNode.java:11-28),
// core.messages.FloodMessage (C/messages/FloodMessage.java:11-61) and protocols.P2PFlood (P/P2PFlood.java:20-152).
// Pinned against PT/P2PFloodTest.java:12-31 (testSimpleRun) in oracle/test_casper.cpp; testLongRun needs the AWS
// latency / node builder (city tables, not restated).
#pragma once
#include <map>
#include <set>
#include "network.hpp"

namespace orc {

struct FloodMessage;
struct P2PNode : Node {  // C/P2PNode.java
  std::vector<P2PNode*> peers;
  std::map<jlong, std::set<const FloodMessage*>> received;
  P2PNode(JRandom& rd, NodeBuilder& nb, bool byz) : Node(rd, nb, byz) {}
  std::set<const FloodMessage*>& getMsgReceived(jlong id) { return received[id]; }
  virtual void onFlood(P2PNode*, const FloodMessage&) {}
};

class P2PNetwork : public Network {  // C/P2PNetwork.java
 public:
  const int connectionCount;
  const bool minimum;
  std::set<jlong> existingLinks;
  P2PNetwork(int cc, bool min) : connectionCount(cc), minimum(min) {}
  P2PNode* p2p(int id) { return static_cast<P2PNode*>(allNodes.at(id)); }
  void setPeers() {  // :27-56
    const int n = (int)allNodes.size();
    if (connectionCount >= n) throw IllegalArgumentException("Wrong configuration: #nodes=" + std::to_string(n));
    if (!minimum) {
      const size_t toCreate = ((size_t)n * connectionCount) / 2;
      while (toCreate != existingLinks.size()) {
        int pp1 = rd.nextInt(n);
        int pp2 = rd.nextInt(n);
        createLink(pp1, pp2);
      }
    }
    std::vector<Node*> an(allNodes);
    jshuffle(an, rd);
    const int want = minimum ? connectionCount : std::min(3, connectionCount);
    for (Node* nn : an) {
      P2PNode* p = static_cast<P2PNode*>(nn);
      while ((int)p->peers.size() < want) createLink(p->nodeId, rd.nextInt(n));
    }
  }
  void createLink(int pp1, int pp2) {  // :72-93
    if (pp1 == pp2) return;
    const jlong link = ((jlong)std::min(pp1, pp2) << 32) + (jlong)std::max(pp1, pp2);
    if (!existingLinks.insert(link).second) return;
    p2p(pp1)->peers.push_back(p2p(pp2));
    p2p(pp2)->peers.push_back(p2p(pp1));
  }
  void sendPeers(std::shared_ptr<FloodMessage> msg, P2PNode* from);  // :127-132
};

struct FloodMessage : Message, std::enable_shared_from_this<FloodMessage> {  // C/messages/FloodMessage.java
  const int size_, localDelay, delayBetweenPeers;
  FloodMessage(int s, int ld, int dbp) : size_(s), localDelay(ld), delayBetweenPeers(dbp) {}
  virtual jlong msgId() const { return -1; }
  bool addToReceived(P2PNode* to) const { return to->getMsgReceived(msgId()).insert(this).second; }
  void action(Network& network, Node* from, Node* toN) override {  // :47-55
    P2PNode* to = static_cast<P2PNode*>(toN);
    if (!addToReceived(to)) return;
    to->onFlood(static_cast<P2PNode*>(from), *this);
    std::vector<Node*> dest;
    for (P2PNode* n : to->peers)
      if (n != from) dest.push_back(n);
    jshuffle(dest, network.rd);
    network.send(shared_from_this(), network.time + 1 + localDelay, to, dest, delayBetweenPeers);
  }
  int size() const override { return size_; }
};

inline void P2PNetwork::sendPeers(std::shared_ptr<FloodMessage> msg, P2PNode* from) {
  msg->addToReceived(from);
  std::vector<Node*> dest(from->peers.begin(), from->peers.end());
  jshuffle(dest, rd);
  send(msg, time + 1 + msg->localDelay, from, dest, msg->delayBetweenPeers);
}

class P2PFlood {  // P/P2PFlood.java
 public:
  struct Params {  // P2PFloodParameters :41-86, ctor order
    int nodeCount = 100, deadNodeCount = 10, delayBeforeResent = 50, msgCount = 1, msgToReceive = 1, peersCount = 10,
        delayBetweenSends = 30;
    std::string nodeBuilderName, networkLatencyName;
  };
  struct P2PFloodNode : P2PNode {  // :25-39
    P2PFlood& p;
    P2PFloodNode(P2PFlood& pp, bool down) : P2PNode(pp.network_.rd, *pp.nb, down), p(pp) {
      if (down) stop();
    }
    void onFlood(P2PNode*, const FloodMessage& m) override {
      if ((int)getMsgReceived(m.msgId()).size() == p.params.msgCount) doneAt = p.network_.time;
    }
  };
  Params params;
  P2PNetwork network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<P2PFloodNode>> nodes;
  explicit P2PFlood(const Params& pr) : params(pr), network_(pr.peersCount, true) {  // :88-94
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
  }
  void init() {  // :121-140
    for (int i = 0; i < params.nodeCount; i++) {
      nodes.push_back(std::make_unique<P2PFloodNode>(*this, i < params.deadNodeCount));
      network_.addNode(nodes.back().get());
    }
    network_.setPeers();
    std::set<int> senders;
    while ((int)senders.size() < params.msgCount) {
      const int nodeId = network_.rd.nextInt(params.nodeCount);
      P2PFloodNode* from = nodes[nodeId].get();
      if (!from->isDown() && senders.insert(nodeId).second) {
        network_.sendPeers(std::make_shared<FloodMessage>(1, params.delayBeforeResent, params.delayBetweenSends), from);
        if (params.msgCount == 1) from->doneAt = 1;
      }
    }
  }
};

}  // namespace orc
