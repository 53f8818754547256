// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.SanFerminSignature (P/SanFerminSignature.java:24-619) and SanFerminHelper
// (P/SanFerminHelper.java:12-173). Pinned against PT/SanFerminTest.java (oracle/test_casper.cpp).
// allNodes is the id-ordered node list (nodes are created and added in id order, :126-131), so a subList(min, max)
// of it is the id range [min, max): contains(node) is a range test and indexOf(node) is id - min. Power-of-two node
// counts only — for other counts toBinaryID's padding throws in the reference too (P/SanFerminHelper.java:158-171).
#pragma once
#include <map>
#include <set>
#include "network.hpp"

namespace orc {

inline int moreMathLog2(int n) {  // C/utils/MoreMath.java:5-10
  if (n <= 0) throw IllegalArgumentException("n=" + std::to_string(n));
  int r = 0;
  while ((1 << (r + 1)) <= n && r < 30) r++;
  return r;
}

struct SanFerminHelper {  // P/SanFerminHelper.java
  int n;                  // the node's id (= its index in allNodes)
  int size;               // allNodes.size()
  std::string binaryId;
  std::map<int, std::vector<bool>> usedNodes;
  JRandom* rd;
  SanFerminHelper(int node, int allNodes, JRandom* r) : n(node), size(allNodes), rd(r) { binaryId = toBinaryID(node, allNodes); }
  static std::string toBinaryID(int nodeId, int setSize) {  // :168-171
    const int log2 = moreMathLog2(setSize);
    std::string s;
    for (int v = nodeId; v > 0; v >>= 1) s.insert(s.begin(), (char)('0' + (v & 1)));
    if (s.empty()) s = "0";
    if ((int)s.size() > log2) throw IllegalStateException("StringIndexOutOfBounds: node id wider than log2(setSize)");
    return std::string(log2 - s.size(), '0') + s;
  }
  void range(int level, bool candidate, int& min, int& max) const {  // getOwnSet :38-56 / getCandidateSet :62-92
    min = 0;
    max = size;
    for (int currLevel = 0; currLevel <= level && min <= max; currLevel++) {
      const int m = (max + min) / 2;
      const bool swap = candidate && currLevel == level;  // "when we are at the right level, swap the order"
      if (binaryId.at(currLevel) == '0') {
        if (swap) min = m; else max = m;
      } else {
        if (swap) max = m; else min = m;
      }
      if (max == min) break;
      if (max - 1 == 0 || min == size) break;
    }
  }
  bool isCandidate(int node, int level) const {  // :94-96
    int lo, hi;
    range(level, true, lo, hi);
    return node >= lo && node < hi;
  }
  std::vector<int> pickNextNodes(int level, int howMany) {  // :112-146
    int cmin, cmax, omin, omax;
    range(level, true, cmin, cmax);
    range(level, false, omin, omax);
    std::vector<int> candidateSet;
    for (int i = cmin; i < cmax; i++) candidateSet.push_back(i);
    const int idx = (n >= omin && n < omax) ? n - omin : -1;
    if (idx == -1 || omax - omin < idx) throw IllegalStateException("pickNextNodes");
    std::vector<int> newList;
    std::vector<bool>& set = usedNodes[level];
    auto get = [&](int i) { return i < (int)set.size() && set[i]; };
    auto put = [&](int i) {
      if (i >= (int)set.size()) set.resize(i + 1, false);
      set[i] = true;
    };
    if (!get(idx)) {
      newList.push_back(candidateSet.at(idx));
      candidateSet.erase(candidateSet.begin() + idx);
      put(idx);
    }
    int taken = 0;
    for (int i = 0; i < (int)candidateSet.size() && taken < howMany; i++)
      if (!get(i)) {
        put(i);
        newList.push_back(candidateSet[i]);
        taken++;
      }
    jshuffle(newList, *rd);
    return newList;
  }
};

class SanFerminSignature {
 public:
  struct Params {  // SanFerminSignatureParameters :39-111
    int nodeCount = 1024, powerOfTwo = 10, threshold = 1024, pairingTime = 2, signatureSize = 48, replyTimeout = 300,
        candidateCount = 1;
    bool shuffledLists = false;
    std::string nodeBuilderName, networkLatencyName;
  };
  enum Status { OK, NO };
  struct SanFerminNode;
  struct SwapRequest : Message {  // :553-574
    SanFerminSignature& p;
    int level, aggValue;
    SwapRequest(SanFerminSignature& pp, int l, int a) : p(pp), level(l), aggValue(a) {}
    void action(Network&, Node* from, Node* to) override;
    int size() const override { return 4 + p.params.signatureSize; }
  };
  struct SwapReply : Message {  // :525-551
    SanFerminSignature& p;
    Status status;
    int level, aggValue;
    SwapReply(SanFerminSignature& pp, Status s, int l, int a) : p(pp), status(s), level(l), aggValue(a) {}
    void action(Network&, Node* from, Node* to) override;
    int size() const override { return 4 + p.params.signatureSize; }
  };

  struct SanFerminNode : Node {  // :149-517
    SanFerminSignature& p;
    int currentPrefixLength;
    std::unique_ptr<SanFerminHelper> candidateTree;
    std::map<int, int> signatureCache, futurSigs;
    std::set<int> pendingNodes;
    bool havePending = false;  // pendingNodes is null until the first goNextLevel (:396)
    bool isSwapping = false;
    int aggValue = 1;
    jlong thresholdAt = 0;
    bool thresholdDone = false, done = false;
    int sentRequests = 0, receivedRequests = 0;
    explicit SanFerminNode(SanFerminSignature& pp) : Node(pp.network_.rd, *pp.nb), p(pp), currentPrefixLength(pp.params.powerOfTwo) {}

    void onSwapRequest(SanFerminNode* node, const SwapRequest& request) {  // :224-264
      receivedRequests++;
      if (done || request.level != currentPrefixLength) {
        auto it = signatureCache.find(request.level);
        if (it != signatureCache.end()) {
          sendSwapReply(node, OK, request.level, it->second);
        } else {
          sendSwapReply(node, NO, currentPrefixLength, 0);
          if (candidateTree->isCandidate(node->nodeId, request.level)) signatureCache[request.level] = request.aggValue;
        }
        return;
      }
      if (isSwapping) {
        sendSwapReply(node, OK, request.level, aggValue);
        return;
      }
      if (candidateTree->isCandidate(node->nodeId, currentPrefixLength)) transition(request.aggValue);
    }
    void onSwapReply(SanFerminNode* from, const SwapReply& reply) {  // :266-316
      if (reply.level != currentPrefixLength || done) return;
      if (isSwapping) return;
      if (!havePending) throw IllegalStateException("NullPointerException: pendingNodes");
      if (reply.status == OK) {
        if (!pendingNodes.count(from->nodeId)) {
          if (candidateTree->isCandidate(from->nodeId, currentPrefixLength)) transition(reply.aggValue);
          return;
        }
        transition(reply.aggValue);
      } else if (pendingNodes.count(from->nodeId)) {
        sendToNodes(candidateTree->pickNextNodes(currentPrefixLength, p.params.candidateCount));
      }
    }
    void sendToNodes(const std::vector<int>& candidates) {  // :322-363
      if (candidates.empty()) return;
      if (!havePending) throw IllegalStateException("NullPointerException: pendingNodes");
      for (int c : candidates) pendingNodes.insert(c);
      sentRequests += (int)candidates.size();
      std::vector<Node*> dests;
      for (int c : candidates) dests.push_back(p.nodes[c].get());
      p.network_.send(std::make_shared<SwapRequest>(p, currentPrefixLength, aggValue), this, dests);
      const int currLevel = currentPrefixLength;
      p.network_.registerTask(
          [this, currLevel] {
            if (!done && currentPrefixLength == currLevel)
              sendToNodes(candidateTree->pickNextNodes(currentPrefixLength, p.params.candidateCount));
          },
          p.network_.time + p.params.replyTimeout, this);
    }
    void goNextLevel() {  // :373-414
      if (done) return;
      const bool enoughSigs = aggValue >= p.params.threshold;
      const bool noMoreSwap = currentPrefixLength == 0;
      if (enoughSigs && !thresholdDone) {
        thresholdDone = true;
        thresholdAt = p.network_.time + p.params.pairingTime * 2;
      }
      if (noMoreSwap && !done) {
        doneAt = p.network_.time + p.params.pairingTime * 2;
        p.finishedNodes.push_back(this);
        done = true;
        return;
      }
      currentPrefixLength--;
      signatureCache[currentPrefixLength] = aggValue;
      isSwapping = false;
      pendingNodes.clear();
      havePending = true;
      auto it = futurSigs.find(currentPrefixLength);
      if (it != futurSigs.end()) {
        aggValue += it->second;
        goNextLevel();
        return;
      }
      sendToNodes(candidateTree->pickNextNodes(currentPrefixLength, p.params.candidateCount));
    }
    void sendSwapReply(SanFerminNode* n, Status s, int level, int value) {  // :416-423
      p.network_.send(std::make_shared<SwapReply>(p, s, level, value), this, std::vector<Node*>{n});
    }
    void transition(int toAggregate) {  // :429-450
      isSwapping = true;
      p.network_.registerTask(
          [this, toAggregate] {
            aggValue += toAggregate;
            goNextLevel();
          },
          p.network_.time + p.params.pairingTime, this);
    }
  };

  Params params;
  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<SanFerminNode>> nodes;  // allNodes
  std::vector<SanFerminNode*> finishedNodes;

  explicit SanFerminSignature(const Params& pr) : params(pr) {  // :113-133
    params.powerOfTwo = moreMathLog2(params.nodeCount);
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
    for (int i = 0; i < params.nodeCount; i++) {
      nodes.push_back(std::make_unique<SanFerminNode>(*this));
      network_.addNode(nodes.back().get());
    }
    for (auto& n : nodes) n->candidateTree = std::make_unique<SanFerminHelper>(n->nodeId, params.nodeCount, &network_.rd);
  }
  void init() {  // :139-141
    for (auto& n : nodes) {
      SanFerminNode* nn = n.get();
      network_.registerTask([nn] { nn->goNextLevel(); }, 1, nn);
    }
  }
};

inline void SanFerminSignature::SwapRequest::action(Network&, Node* from, Node* to) {
  static_cast<SanFerminNode*>(to)->onSwapRequest(static_cast<SanFerminNode*>(from), *this);
}
inline void SanFerminSignature::SwapReply::action(Network&, Node* from, Node* to) {
  static_cast<SanFerminNode*>(to)->onSwapReply(static_cast<SanFerminNode*>(from), *this);
}

}  // namespace orc
