// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.GSFSignature (P/GSFSignature.java:22-687): gossiping San Fermin.
// Payload aliasing is restated as it is in Java: a SendSigs' BitSet is shared by reference between
// the envelope, every receiver's toVerify list and the registered updateVerifiedSignatures task, and
// updateVerifiedSignatures mutates it in place (:390, :419) — SURVEY App. D/E.
#pragma once
#include <unordered_map>
#include "network.hpp"

namespace orc {

class GSFSignature {
 public:
  struct GSFSignatureParameters {  // :26-107
    int nodeCount, threshold, pairingTime, timeoutPerLevelMs, periodDurationMs, acceleratedCallsCount, nodesDown;
    std::string nodeBuilderName, networkLatencyName;
    GSFSignatureParameters(int nodeCount_, int threshold_, int pairingTime_, int timeoutPerLevelMs_,
                           int periodDurationMs_, int acceleratedCallsCount_, int nodesDown_, std::string nb,
                           std::string nl)
        : nodeCount(nodeCount_), threshold(threshold_), pairingTime(pairingTime_),
          timeoutPerLevelMs(timeoutPerLevelMs_), periodDurationMs(periodDurationMs_),
          acceleratedCallsCount(acceleratedCallsCount_), nodesDown(nodesDown_), nodeBuilderName(std::move(nb)),
          networkLatencyName(std::move(nl)) {
      if (nodesDown >= nodeCount || nodesDown < 0 || threshold > nodeCount || (nodesDown + threshold > nodeCount))
        throw IllegalArgumentException("nodeCount=" + std::to_string(nodeCount) + ", threshold=" + std::to_string(threshold));
    }
  };

  class GSFNode;
  struct SFLevel;

  struct SendSigs : Message, std::enable_shared_from_this<SendSigs> {  // :137-164
    BitSet sigs;  // mutable, shared by reference (see header)
    GSFNode* from;
    int level;
    bool levelFinished;
    int size_;
    int received;
    SendSigs(GSFNode* from_, const BitSet& s, const SFLevel& l);
    int size() const override { return size_; }
    void action(Network&, Node* from, Node* to) override;
  };
  typedef std::shared_ptr<SendSigs> SendSigsP;

  struct SFLevel {  // :236-356
    GSFNode& n;
    int level;
    std::vector<GSFNode*> peers;
    BitSet waitedSigs, verifiedSignatures, individualSignatures, indivVerifiedSig;
    int posInLevel = 0;
    int remainingCalls;

    explicit SFLevel(GSFNode& node) : n(node), level(0) {  // :263-270
      waitedSigs.set(n.nodeId);
      verifiedSignatures.set(n.nodeId);
      remainingCalls = 0;
    }
    SFLevel(GSFNode& node, const SFLevel& previousLevel, const BitSet& allPreviousNodes) : n(node) {  // :273-283
      level = previousLevel.level + 1;
      waitedSigs = n.allSigsAtLevel(level);
      waitedSigs.andNot(allPreviousNodes);
      peers = n.randomSubset(waitedSigs, INT32_MAX);
      remainingCalls = (int)peers.size();
    }
    int expectedSigs() const { return waitedSigs.cardinality(); }  // :289-291
    bool hasStarted(const BitSet& toSend) const {  // :294-315
      if (n.g.network_.time >= level * n.g.params.timeoutPerLevelMs) return true;
      if (toSend.cardinality() >= expectedSigs()) return true;
      return false;
    }
    void doCycle(const BitSet& toSend) {  // :317-327
      if (remainingCalls == 0 || !hasStarted(toSend)) return;
      std::vector<GSFNode*> dest = getRemainingPeers(1);
      if (!dest.empty()) {
        auto ss = std::make_shared<SendSigs>(&n, toSend, *this);
        n.g.network_.send(ss, &n, dest[0]);
      }
    }
    std::vector<GSFNode*> getRemainingPeers(int peersCt) {  // :329-353 (the `received` test is dead: `|| true`)
      std::vector<GSFNode*> res;
      while (peersCt > 0 && remainingCalls > 0) {
        remainingCalls--;
        GSFNode* p = peers[posInLevel++];
        if (posInLevel >= (int)peers.size()) posInLevel = 0;
        res.push_back(p);
        peersCt--;
      }
      return res;
    }
    bool hasReceivedAll() const {  // :355-359 (result unused by hasStarted; kept for completeness)
      BitSet wanted = waitedSigs;
      wanted.and_(verifiedSignatures);
      return wanted.cardinality() >= .8 * expectedSigs();
    }
  };

  class GSFNode : public Node {  // :166-604
   public:
    GSFSignature& g;
    std::vector<SendSigsP> toVerify;
    std::vector<std::unique_ptr<SFLevel>> levels;
    BitSet verifiedSignatures;
    int nodePairingTime;
    bool done = false;
    int sigChecked = 0;
    int sigQueueSize = 0;

    explicit GSFNode(GSFSignature& gg) : Node(gg.network_.rd, *gg.nb), g(gg) {  // :177-180
      nodePairingTime = (int)std::max(1.0, g.params.pairingTime * speedRatio);
      verifiedSignatures.set(nodeId);
    }
    void initLevel() {  // :182-192
      int rounded = roundPow2(g.params.nodeCount);
      BitSet allPreviousNodes;
      levels.push_back(std::make_unique<SFLevel>(*this));
      for (int l = 1; (1 << l) <= rounded; l++) {  // Math.pow(2, l) <= roundedPow2NodeCount
        allPreviousNodes.or_(levels.back()->waitedSigs);
        levels.push_back(std::make_unique<SFLevel>(*this, *levels.back(), allPreviousNodes));
      }
    }
    BitSet getLastFinishedLevel() const {  // :194-211
      BitSet res;
      const SFLevel* sfl = levels[0].get();
      bool fin = false;
      while (!fin) {
        if (sfl->waitedSigs.equals(sfl->verifiedSignatures)) {
          res.or_(sfl->waitedSigs);
          if (sfl->level < (int)levels.size() - 1)
            sfl = levels[sfl->level + 1].get();
          else
            fin = true;
        } else {
          fin = true;
        }
      }
      return res;
    }
    void doCycle() {  // :213-225
      BitSet toSend = getLastFinishedLevel();
      for (auto& sfl : levels) {
        sfl->doCycle(toSend);
        toSend.or_(sfl->verifiedSignatures);
      }
    }
    BitSet allSigsAtLevel(int round) const {  // :362-375
      if (round < 1) throw IllegalArgumentException("round=" + std::to_string(round));
      BitSet res;
      int cMask = (1 << round) - 1;
      int start = (cMask | nodeId) ^ cMask;
      int end = nodeId | cMask;
      end = std::min(end, g.params.nodeCount - 1);
      res.setRange(start, end + 1);
      res.clear(nodeId);
      return res;
    }
    static bool include(const BitSet& large, const BitSet& small) {  // :377-381
      BitSet a = large;
      a.and_(small);
      return a.equals(small);
    }
    // :387-460. `msg` is the SendSigs whose BitSet Java passes by reference (tBest.sigs).
    void updateVerifiedSignatures(GSFNode* from, int level, const SendSigsP& msg) {
      SFLevel* sfl = levels[level].get();
      BitSet* sigs = &msg->sigs;
      BitSet local;
      if (sigs->cardinality() == 1) sfl->indivVerifiedSig.set(from->nodeId);
      sigs->or_(sfl->indivVerifiedSig);  // in place: visible to every other holder of the message
      bool resetRemaining = false;
      if (sigs->cardinality() > sfl->expectedSigs()) {
        for (int i = 1; i < (int)levels.size() && include(*sigs, levels[i]->waitedSigs); i++) {
          SFLevel& l = *levels[i];
          if (!l.verifiedSignatures.equals(l.waitedSigs)) {
            l.verifiedSignatures.or_(l.waitedSigs);
            verifiedSignatures.or_(l.waitedSigs);
            resetRemaining = true;
          }
          if (resetRemaining) l.remainingCalls = (int)l.peers.size();
        }
        local = sfl->waitedSigs;  // sigs = (BitSet) sfl.waitedSigs.clone(): rebinds the local only
        sigs = &local;
      }
      if (sfl->verifiedSignatures.cardinality() > 0 && !sigs->intersects(sfl->verifiedSignatures))
        sigs->or_(sfl->verifiedSignatures);  // in place on whichever object `sigs` now names
      if (sigs->cardinality() > sfl->verifiedSignatures.cardinality() || resetRemaining) {
        for (int i = sfl->level; i < (int)levels.size(); i++) levels[i]->remainingCalls = (int)levels[i]->peers.size();
        sfl->verifiedSignatures.andNot(sfl->waitedSigs);
        sfl->verifiedSignatures.or_(*sigs);
        verifiedSignatures.andNot(sfl->waitedSigs);
        verifiedSignatures.or_(*sigs);
        if (g.params.acceleratedCallsCount > 0) {
          BitSet bestToSend = getLastFinishedLevel();
          while (include(bestToSend, sfl->waitedSigs) && sfl->level < (int)levels.size() - 1) {
            sfl = levels[sfl->level + 1].get();
            auto sendSigs = std::make_shared<SendSigs>(this, bestToSend, *sfl);
            std::vector<GSFNode*> peers = sfl->getRemainingPeers(g.params.acceleratedCallsCount);
            if (!peers.empty()) {
              std::vector<Node*> dests(peers.begin(), peers.end());
              g.network_.send(sendSigs, this, dests);
            }
          }
        }
        if (doneAt == 0 && verifiedSignatures.cardinality() >= g.params.threshold) doneAt = g.network_.time;
      }
    }
    std::vector<GSFNode*> randomSubset(const BitSet& nodes, int nodeCt) {  // :462-476
      std::vector<GSFNode*> res;
      for (int cur = nodes.nextSetBit(0); cur >= 0; cur = nodes.nextSetBit(cur + 1))
        res.push_back(static_cast<GSFNode*>(g.network_.getNodeById(cur)));
      jshuffle(res, g.network_.rd);
      if ((int)res.size() > nodeCt) res.resize(nodeCt);
      return res;
    }
    int evaluateSig(const SFLevel& l, const BitSet& sig) const {  // :482-535
      int newTotal, addedSigs;
      if (l.verifiedSignatures.cardinality() >= l.expectedSigs()) return 0;
      BitSet withIndiv = l.indivVerifiedSig;
      withIndiv.or_(sig);
      if (l.verifiedSignatures.cardinality() == 0) {
        newTotal = sig.cardinality();
        addedSigs = newTotal;
      } else if (sig.intersects(l.verifiedSignatures)) {
        newTotal = withIndiv.cardinality();
        addedSigs = newTotal - l.verifiedSignatures.cardinality();
      } else {
        withIndiv.or_(l.verifiedSignatures);
        newTotal = withIndiv.cardinality();
        addedSigs = newTotal - l.verifiedSignatures.cardinality();
      }
      if (addedSigs <= 0) {
        if (sig.cardinality() == 1 && !sig.intersects(l.indivVerifiedSig)) return 1;
        return 0;
      }
      if (newTotal == l.expectedSigs()) return 1000000 - l.level * 10;
      return 100000 - l.level * 100 + addedSigs;
    }
    void onNewSig(GSFNode* from, const SendSigsP& ssigs) {  // :538-556
      SFLevel& l = *levels[ssigs->level];
      // (l.received.put(from, 1) when levelFinished: the map is never read, :337-345 `|| true`)
      // copyOnDelivery (not in the reference): the receiver keeps a private clone instead of the shared
      // object; tests use it to show that the in-place mutation of :390/:419 is unobservable
      toVerify.push_back(g.copyOnDelivery ? std::make_shared<SendSigs>(*ssigs) : ssigs);
      if (!l.individualSignatures.get(from->nodeId)) {
        BitSet indiv;
        indiv.set(from->nodeId);
        toVerify.push_back(std::make_shared<SendSigs>(from, indiv, l));
        l.individualSignatures.set(from->nodeId);
      }
      sigQueueSize = (int)toVerify.size();
      g.statQueueMax = std::max(g.statQueueMax, sigQueueSize);
    }
    void checkSigs() {  // :558-584
      SendSigsP best;
      int score = 0;
      for (auto it = toVerify.begin(); it != toVerify.end();) {
        const SendSigsP& cur = *it;
        int ns = evaluateSig(*levels[cur->level], cur->sigs);
        if (ns > score) {
          score = ns;
          best = cur;
          ++it;
        } else if (ns == 0) {
          it = toVerify.erase(it);
        } else {
          ++it;
        }
      }
      if (best) {
        auto it = std::find(toVerify.begin(), toVerify.end(), best);  // ArrayList.remove(Object): first identical
        toVerify.erase(it);
        sigChecked++;
        sigQueueSize = (int)toVerify.size();
        GSFNode* me = this;
        g.network_.registerTask([me, best] { me->updateVerifiedSignatures(best->from, best->level, best); },
                                g.network_.time + nodePairingTime, this);
      }
    }
  };

  GSFSignatureParameters params;
  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<GSFNode>> nodes;
  // instrumentation (not in the reference)
  uint64_t statDeliveredByLevel[32] = {0};
  int statQueueMax = 0;
  bool copyOnDelivery = false;
  uint64_t statShapeViolations = 0;

  explicit GSFSignature(const GSFSignatureParameters& p) : params(p) {  // :109-114
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
  }
  GSFNode* node(int i) { return nodes[i].get(); }
  Network& network() { return network_; }

  void init() {  // :611-635
    for (int i = 0; i < params.nodeCount; i++) {
      nodes.push_back(std::make_unique<GSFNode>(*this));
      network_.addNode(nodes.back().get());
    }
    for (int setDown = 0; setDown < params.nodesDown;) {
      int down = network_.rd.nextInt(params.nodeCount);
      Node* n = network_.getNodeById(down);
      if (!n->isDown() && down != 1) {
        n->stop();
        setDown++;
      }
    }
    for (auto& np : nodes) {
      GSFNode* n = np.get();
      if (!n->isDown()) {
        n->initLevel();
        network_.registerPeriodicTask([n] { n->doCycle(); }, 1, params.periodDurationMs, n);
        network_.registerConditionalTask([n] { n->checkSigs(); }, 1, n->nodePairingTime, n,
                                         [n] { return !n->toVerify.empty(); }, [n] { return !n->done; });
      }
    }
  }
  bool contIf() {  // newConfIf :670-683
    for (auto& n : nodes)
      if (!n->isDown() && n->verifiedSignatures.cardinality() < params.threshold) return true;
    return false;
  }
};

inline GSFSignature::SendSigs::SendSigs(GSFNode* from_, const BitSet& s, const SFLevel& l)
    : sigs(s), from(from_), level(l.level) {  // :145-153
  size_ = 1 + l.expectedSigs() / 8 + 96;
  levelFinished = l.verifiedSignatures.equals(l.waitedSigs);
  received = l.verifiedSignatures.cardinality();
}
inline void GSFSignature::SendSigs::action(Network&, Node* from_, Node* to) {  // :160-163
  GSFNode* t = static_cast<GSFNode*>(to);
  t->g.statDeliveredByLevel[level]++;
  // instrumentation (not in the reference): the device protocol stores a payload either as the bits of
  // the receiver's level block or as "the aligned block of 2^j ids around the receiver" (DESIGN.md);
  // count every delivered message that has neither shape
  if (!t->levels.empty()) {
    const BitSet& w = t->levels[level]->waitedSigs;
    BitSet in = sigs;
    in.and_(w);
    bool ok = in.equals(sigs);
    for (int j = level; !ok && (1 << j) <= roundPow2(t->g.params.nodeCount); j++) {
      BitSet blk;
      int lo = (t->nodeId >> j) << j;
      blk.setRange(lo, std::min(lo + (1 << j), t->g.params.nodeCount));
      ok = blk.equals(sigs);
    }
    if (!ok) t->g.statShapeViolations++;
  }
  t->onNewSig(static_cast<GSFNode*>(from_), shared_from_this());
}

}  // namespace orc
