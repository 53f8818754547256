// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.SanFerminCappos (P/SanFerminCappos.java:24-523) — San Fermin with one Swap message
// (wantReply) instead of request / reply, a list of cached signatures per level, and candidate sets re-tried on a
// timeout — over the same SanFerminHelper (oracle/sanfermin.hpp). The reference holds no test of this class
// (PT/ has SanFerminTest only): it is pinned by the invariants of tests/test_oracle_protocols.py (every node finishes
// with the whole set aggregated) and serves as the checker of examples/hostmode/sanfermin_cappos.py on the engine.
// Nodes are created in init() (:121-126), so rd.setSeed() before init() decides their positions too.
#pragma once
#include <map>
#include "sanfermin.hpp"

namespace orc {

class SanFerminCappos {
 public:
  struct Params {  // SanFerminParameters :44-108
    int nodeCount = 32768 / 16, pairingTime = 2, signatureSize = 48, candidateCount = 50, threshold = 32768 / 32,
        timeout = 150;
    std::string nodeBuilderName, networkLatencyName;
  };
  struct SanFerminNode;
  struct Swap : Message {  // :437-460
    SanFerminCappos& p;
    bool wantReply;
    int level, aggValue;
    Swap(SanFerminCappos& pp, int l, int a, bool reply) : p(pp), wantReply(reply), level(l), aggValue(a) {}
    void action(Network&, Node* from, Node* to) override;
    int size() const override { return 4 + p.params.signatureSize; }
  };

  struct SanFerminNode : Node {  // :146-435
    SanFerminCappos& p;
    std::unique_ptr<SanFerminHelper> helper;
    int currentPrefixLength;
    std::map<int, std::vector<int>> signatureCache;  // HashMap<Integer, List<Integer>>: only sums / maxima are read
    bool isSwapping = false;
    int aggValue = 1;
    jlong thresholdAt = 0;
    bool thresholdDone = false, done = false;
    explicit SanFerminNode(SanFerminCappos& pp)
        : Node(pp.network_.rd, *pp.nb), p(pp), currentPrefixLength(moreMathLog2(pp.params.nodeCount)) {
      (void)SanFerminHelper::toBinaryID(nodeId, p.params.nodeCount);  // binaryId (:176): throws for ids wider than log2(n)
    }
    void onSwap(SanFerminNode* from, const Swap& swap) {  // :200-237
      const bool wantReply = swap.wantReply;
      if (done || swap.level != currentPrefixLength) {
        const bool isValueCached = signatureCache.count(swap.level) != 0;
        if (wantReply && isValueCached) {
          sendSwap({from}, swap.level, getBestCachedSig(swap.level), false);
        } else if (helper->isCandidate(from->nodeId, swap.level)) {
          putCachedSig(swap.level, swap.aggValue);
        }
        return;
      }
      if (wantReply) sendSwap({from}, swap.level, totalNumberOfSigs(swap.level), false);
      const bool goodLevel = swap.level == currentPrefixLength;
      const bool isCandidate = helper->isCandidate(from->nodeId, currentPrefixLength);
      if (isCandidate && goodLevel && !isSwapping) transition(swap.level, swap.aggValue);
    }
    void tryNextNodes(const std::vector<int>& candidates) {  // :239-279
      if (candidates.empty()) return;
      for (int c : candidates)
        if (!helper->isCandidate(c, currentPrefixLength)) throw IllegalStateException("tryNextNodes: not a candidate");
      std::vector<SanFerminNode*> dests;
      for (int c : candidates) dests.push_back(p.nodes[c].get());
      sendSwap(dests, currentPrefixLength, totalNumberOfSigs(currentPrefixLength + 1), true);
      const int currLevel = currentPrefixLength;
      p.network_.registerTask(
          [this, currLevel] {
            if (!done && currentPrefixLength == currLevel)
              tryNextNodes(helper->pickNextNodes(currentPrefixLength, p.params.candidateCount));
          },
          p.network_.time + p.params.timeout, this);
    }
    void goNextLevel() {  // :281-321
      if (done) return;
      const bool enoughSigs = totalNumberOfSigs(currentPrefixLength) >= p.params.threshold;
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
      isSwapping = false;
      if (signatureCache.count(currentPrefixLength)) {  // a value for the new level came early: move on directly
        goNextLevel();
        return;
      }
      tryNextNodes(helper->pickNextNodes(currentPrefixLength, p.params.candidateCount));
    }
    void sendSwap(const std::vector<SanFerminNode*>& nodes, int level, int value, bool wantReply) {  // :323-326
      std::vector<Node*> dests(nodes.begin(), nodes.end());
      p.network_.send(std::make_shared<Swap>(p, level, value, wantReply), this, dests);
    }
    int totalNumberOfSigs(int level) const {  // :328-335
      int sum = 0;
      for (auto& e : signatureCache)
        if (e.first >= level) sum += *std::max_element(e.second.begin(), e.second.end());
      return sum + 1;  // +1 for own sig
    }
    void transition(int level, int toAggregate) {  // :337-347
      isSwapping = true;
      p.network_.registerTask(
          [this, level, toAggregate] {
            putCachedSig(level, toAggregate);
            goNextLevel();
          },
          p.network_.time + p.params.pairingTime, this);
    }
    int getBestCachedSig(int level) const {  // :349-353
      auto it = signatureCache.find(level);
      if (it == signatureCache.end() || it->second.empty()) throw IllegalStateException("NoSuchElementException");
      return *std::max_element(it->second.begin(), it->second.end());
    }
    void putCachedSig(int level, int value) {  // :355-366
      signatureCache[level].push_back(value);
      const bool enoughSigs = totalNumberOfSigs(currentPrefixLength) >= p.params.threshold;
      if (enoughSigs && !thresholdDone) {
        thresholdDone = true;
        thresholdAt = p.network_.time + p.params.pairingTime * 2;
      }
    }
  };

  Params params;
  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<SanFerminNode>> nodes;  // allNodes
  std::vector<SanFerminNode*> finishedNodes;

  explicit SanFerminCappos(const Params& pr) : params(pr) {  // :110-116
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
  }
  void init() {  // :119-134
    for (int i = 0; i < params.nodeCount; i++) {
      nodes.push_back(std::make_unique<SanFerminNode>(*this));
      network_.addNode(nodes.back().get());
    }
    for (auto& n : nodes) n->helper = std::make_unique<SanFerminHelper>(n->nodeId, params.nodeCount, &network_.rd);
    finishedNodes.clear();
    for (auto& n : nodes) {
      SanFerminNode* nn = n.get();
      network_.registerTask([nn] { nn->goNextLevel(); }, 1, nn);
    }
  }
};

inline void SanFerminCappos::Swap::action(Network&, Node* from, Node* to) {
  static_cast<SanFerminNode*>(to)->onSwap(static_cast<SanFerminNode*>(from), *this);
}

}  // namespace orc
