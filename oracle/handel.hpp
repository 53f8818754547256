// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.Handel (P/Handel.java:18-1054), including the two attack scenarios of its parameters:
// byzantineSuicide (:64-69, 406, 538-559, 577-584, 688-694) and hiddenByzantine (:70-71, 303, 813-817, 840-917).
// The reference holds no test of either (PT/HandelTest runs honest parameters only): they are pinned by construction
// and by the invariants tests/test_oracle_protocols.py checks (blacklists only hold down nodes, the run still converges).
#pragma once
#include "network.hpp"

namespace orc {

class Handel {
 public:
  struct WindowParameters {  // :147-174 with ScoringExp(2,4) :176-206
    int initial = 16, minimum = 1, maximum = 128;
    int newSize(int cur, bool correct) const {
      int updated = correct ? (int)std::ceil((double)cur * 2.0) : (int)std::floor((double)cur / 4.0);
      if (updated > maximum) return maximum;
      if (updated < minimum) return minimum;
      return updated;
    }
  };
  struct HandelParameters {  // :22-142
    int nodeCount, threshold, pairingTime, levelWaitTime, extraCycle, disseminationPeriodMs, fastPath, nodesDown;
    std::string nodeBuilderName, networkLatencyName;
    int desynchronizedStart;
    bool byzantineSuicide, hiddenByzantine;
    const BitSet* badNodes;
    WindowParameters window;
    HandelParameters(int nodeCount_, int threshold_, int pairingTime_, int levelWaitTime_, int extraCycle_,
                     int disseminationPeriodMs_, int fastPath_, int nodesDown_, std::string nb, std::string nl,
                     int desynchronizedStart_, bool byzSuicide, bool hiddenByz, const BitSet* bad)
        : nodeCount(nodeCount_), threshold(threshold_), pairingTime(pairingTime_), levelWaitTime(levelWaitTime_),
          extraCycle(extraCycle_), disseminationPeriodMs(disseminationPeriodMs_), fastPath(fastPath_),
          nodesDown(nodesDown_), nodeBuilderName(std::move(nb)), networkLatencyName(std::move(nl)),
          desynchronizedStart(desynchronizedStart_), byzantineSuicide(byzSuicide), hiddenByzantine(hiddenByz),
          badNodes(bad) {
      if (nodesDown >= nodeCount || nodesDown < 0 || threshold > nodeCount || (nodesDown + threshold > nodeCount))
        throw IllegalArgumentException("nodeCount=" + std::to_string(nodeCount));
      if (__builtin_popcount((unsigned)nodeCount) != 1)
        throw IllegalArgumentException("We support only power of two nodes in this simulation");
      if (byzantineSuicide && hiddenByzantine) throw IllegalArgumentException("Only one attack at a time");
    }
  };

  struct SigToVerify {  // :919-938
    int from, level, rank;
    BitSet sig;
    bool badSig;
  };
  typedef std::shared_ptr<SigToVerify> SigP;

  class HNode;
  struct HLevel;

  struct SendSigs : Message {  // :239-276
    int level;
    BitSet sigs;
    bool levelFinished;
    int size_;
    bool badSig = false;
    SendSigs(const BitSet& s, const HLevel& l);
    int size() const override { return size_; }
    void action(Network&, Node* from, Node* to) override;
  };

  struct HLevel {  // :371-643
    HNode& n;
    int level, size;
    std::vector<HNode*> peers;
    BitSet waitedSigs, lastAggVerified, totalIncoming, verifiedIndSignatures, toVerifyInd, finishedPeers,
        totalOutgoing;
    std::vector<SigP> toVerifyAgg;
    bool outgoingFinished = false;
    int posInLevel = 0;
    int suicideBizAfter;  // :406 a cache for the first suicidal byz node in our list (if any)

    explicit HLevel(HNode& node) : n(node), level(0), size(1) {  // :413-421
      suicideBizAfter = n.h.params.byzantineSuicide ? 0 : -1;
      outgoingFinished = true;
      lastAggVerified.set(n.nodeId);
      verifiedIndSignatures.set(n.nodeId);
      totalIncoming.set(n.nodeId);
    }
    HLevel(HNode& node, const HLevel& previous, const BitSet& allPreviousNodes) : n(node) {  // :424-435
      suicideBizAfter = n.h.params.byzantineSuicide ? 0 : -1;
      level = previous.level + 1;
      waitedSigs.or_(n.allSigsAtLevel(level));
      waitedSigs.andNot(allPreviousNodes);
      totalOutgoing.set(n.nodeId);
      size = waitedSigs.cardinality();
    }
    int expectedSigs() const { return size; }
    std::vector<HNode*> expectedNodes() const {  // :446-455
      std::vector<HNode*> e;
      for (int cur = waitedSigs.nextSetBit(0); cur >= 0; cur = waitedSigs.nextSetBit(cur + 1))
        e.push_back(static_cast<HNode*>(n.h.network_.getNodeById(cur)));
      return e;
    }
    bool isOpen() const {  // :458-472
      if (outgoingFinished) return false;
      if (n.h.network_.time >= (level - 1) * n.h.params.levelWaitTime) return true;
      if (outgoingComplete()) return true;
      return false;
    }
    void doCycle() {  // :474-484
      if (!isOpen()) return;
      std::vector<HNode*> dest = getRemainingPeers(1);
      if (!dest.empty()) {
        auto ss = std::make_shared<SendSigs>(totalOutgoing, *this);
        n.h.network_.send(ss, &n, dest[0]);
      }
    }
    std::vector<HNode*> getRemainingPeers(int peersCt) {  // :486-508
      std::vector<HNode*> res;
      int start = posInLevel;
      while (peersCt > 0 && !outgoingFinished) {
        HNode* p = peers[posInLevel++];
        if (posInLevel >= (int)peers.size()) posInLevel = 0;
        if (!finishedPeers.get(p->nodeId) && !n.blacklist.get(p->nodeId)) {
          res.push_back(p);
          peersCt--;
        } else if (posInLevel == start) {
          outgoingFinished = true;
        }
      }
      return res;
    }
    bool incomingComplete() const { return waitedSigs.equals(totalIncoming); }      // :524-526
    bool outgoingComplete() const { return totalOutgoing.cardinality() == size; }   // :528-530
    int sizeIfIncluded(const SigToVerify& sig) const {                              // :532-540
      BitSet c = sig.sig;
      if (!c.intersects(totalIncoming)) c.or_(totalIncoming);
      c.or_(verifiedIndSignatures);
      return c.cardinality();
    }
    SigP createSuicideByzantineSig(int maxRank) {  // :538-559
      bool reset = false;
      for (int i = suicideBizAfter; i < (int)peers.size(); i++) {
        HNode* p = peers[i];
        if (p->isDown() && !n.blacklist.get(p->nodeId)) {
          if (!reset) {
            suicideBizAfter = i;
            reset = true;
          }
          if (n.receptionRanks[p->nodeId] < maxRank) {
            auto b = std::make_shared<SigToVerify>();
            b->from = p->nodeId;
            b->level = level;
            b->rank = n.receptionRanks[p->nodeId];
            b->sig = waitedSigs;
            b->badSig = true;
            return b;
          }
        }
      }
      if (!reset) suicideBizAfter = -1;  // no byzantine nodes left in this level
      return nullptr;
    }
    SigP bestToVerify() {  // :570-634
      if (toVerifyAgg.empty()) return nullptr;
      if (n.currWindowSize < 1) throw IllegalStateException("currWindowSize");
      int windowIndex = toVerifyAgg[0]->rank;  // Collections.min(..., comparingInt(rank)).rank
      for (auto& s : toVerifyAgg) windowIndex = std::min(windowIndex, s->rank);
      if (suicideBizAfter >= 0) {  // :577-584
        SigP bSig = createSuicideByzantineSig(windowIndex + n.currWindowSize);
        if (bSig) {
          toVerifyAgg.push_back(bSig);
          n.sigQueueSize++;
          return bSig;
        }
      }
      int curSignatureSize = totalIncoming.cardinality();
      SigP bestOutside, bestInside;
      int bestScoreInside = 0;
      int removed = 0;
      std::vector<SigP> curated;
      for (auto& stv : toVerifyAgg) {
        int s = sizeIfIncluded(*stv);
        if (!n.blacklist.get(stv->from) && s > curSignatureSize) {
          curated.push_back(stv);
          if (stv->rank <= windowIndex + n.currWindowSize) {
            int sc = n.score(*this, stv->sig);
            if (sc > bestScoreInside) {
              bestScoreInside = sc;
              bestInside = stv;
            }
          } else if (!bestOutside || stv->rank < bestOutside->rank) {
            bestOutside = stv;
          }
        } else {
          removed++;
        }
      }
      if (removed > 0) {  // replaceToVerifyAgg :636-646
        int oldSize = (int)toVerifyAgg.size();
        toVerifyAgg = curated;
        n.sigQueueSize -= oldSize;
        n.sigQueueSize += (int)toVerifyAgg.size();
        if (n.sigQueueSize < 0) throw IllegalStateException("sigQueueSize");
      }
      if (bestInside) return bestInside;
      return bestOutside;  // may be null
    }
  };

  class HNode : public Node {  // :278-838
   public:
    Handel& h;
    int startAt;
    std::vector<std::unique_ptr<HLevel>> levels;
    int nodePairingTime;
    std::vector<int> receptionRanks;
    BitSet blacklist;
    struct HiddenByzantine {  // :840-917
      bool noByzantinePeers = false;
      SigP last;
      HNode* firstByzantine(HNode& t, HLevel& l) {  // :844-858
        HNode* best = nullptr;
        int bestRank = INT32_MAX;
        for (HNode* p : l.peers) {
          if (p->isDown() && t.receptionRanks[p->nodeId] < bestRank && !l.totalIncoming.get(p->nodeId)) {
            bestRank = t.receptionRanks[p->nodeId];
            best = p;
            if (bestRank == 0) return p;
          }
        }
        return best;  // can be null if this node has no byzantine peer
      }
      SigP attack(HNode& target, const SigP& currentBest) {  // :861-916
        if (noByzantinePeers) return currentBest;
        if (last == currentBest) {  // a previous attack finally worked
          last = nullptr;
          return currentBest;
        }
        HLevel& l = *target.levels[currentBest->level];
        if (last) {
          if (std::find(l.toVerifyAgg.begin(), l.toVerifyAgg.end(), last) != l.toVerifyAgg.end()) return currentBest;
          if (!l.totalIncoming.get(last->from)) throw IllegalStateException("byz signature pruned!");
          last = nullptr;
        }
        HNode* fb = firstByzantine(target, l);
        if (!fb) {
          noByzantinePeers = true;
          return currentBest;
        }
        if (target.receptionRanks[fb->nodeId] >= currentBest->rank) return currentBest;  // we can't improve it, we're too far
        auto bad = std::make_shared<SigToVerify>();
        bad->from = fb->nodeId;
        bad->level = l.level;
        bad->rank = target.receptionRanks[fb->nodeId];
        bad->sig.set(fb->nodeId);
        bad->badSig = false;
        l.toVerifyAgg.push_back(bad);
        target.sigQueueSize++;
        SigP newBest = l.bestToVerify();
        if (newBest != bad) last = bad;
        return newBest;
      }
    };
    std::unique_ptr<HiddenByzantine> hiddenByzantine;  // :293, :303
    int currWindowSize;
    int addedCycle;
    bool done = false;
    int sigsChecked = 0, sigQueueSize = 0, msgFiltered = 0;

    HNode(Handel& hh, int startAt_, NodeBuilder& nb, bool byz)
        : Node(hh.network_.rd, nb, byz), h(hh), startAt(startAt_) {
      nodePairingTime = (int)std::max(1.0, h.params.pairingTime * speedRatio);  // :283
      receptionRanks.assign(h.params.nodeCount, 0);
      currWindowSize = h.params.window.initial;
      addedCycle = h.params.extraCycle;
      if (h.params.hiddenByzantine && !byz) hiddenByzantine = std::make_unique<HiddenByzantine>();
    }
    void initLevel() {  // :319-329
      int rounded = roundPow2(h.params.nodeCount);
      BitSet allPreviousNodes;
      levels.push_back(std::make_unique<HLevel>(*this));
      for (int l = 1; (1 << l) <= rounded; l++) {
        allPreviousNodes.or_(levels.back()->waitedSigs);
        levels.push_back(std::make_unique<HLevel>(*this, *levels.back(), allPreviousNodes));
      }
    }
    void dissemination() {  // :331-343
      if (doneAt > 0) {
        if (addedCycle > 0)
          addedCycle--;
        else
          return;
      }
      for (auto& sfl : levels) sfl->doCycle();
    }
    bool hasSigToVerify() const { return sigQueueSize != 0; }
    int totalSigSize() const {  // :349-352
      const HLevel& last = *levels.back();
      return last.totalOutgoing.cardinality() + last.totalIncoming.cardinality();
    }
    int score(const HLevel& l, const BitSet& sig) const {  // :655-668
      if (l.lastAggVerified.cardinality() >= l.expectedSigs()) return 0;
      if (!l.lastAggVerified.intersects(sig)) return l.lastAggVerified.cardinality() + sig.cardinality();
      BitSet withIndiv = l.verifiedIndSignatures;
      withIndiv.or_(sig);
      return std::max(0, withIndiv.cardinality() - l.lastAggVerified.cardinality());
    }
    BitSet allSigsAtLevel(int round) const {  // :671-684
      if (round < 1) throw IllegalArgumentException("round");
      BitSet res;
      int cMask = (1 << round) - 1;
      int start = (cMask | nodeId) ^ cMask;
      int end = nodeId | cMask;
      end = std::min(end, h.params.nodeCount - 1);
      res.setRange(start, end + 1);
      res.clear(nodeId);
      return res;
    }
    void updateVerifiedSignatures(const SigP& vs) {  // :690-754
      if (vs->badSig) {  // :688-694
        blacklist.set(vs->from);
        if (!h.params.byzantineSuicide) throw IllegalStateException("We should not have invalid signatures in this scenario");
        return;
      }
      HLevel& vsl = *levels[vs->level];
      if (!bitsetInclude(vsl.waitedSigs, vs->sig)) throw IllegalStateException("bad signature received");
      vsl.toVerifyInd.clear(vs->from);
      auto it = std::find(vsl.toVerifyAgg.begin(), vsl.toVerifyAgg.end(), vs);
      if (it != vsl.toVerifyAgg.end()) vsl.toVerifyAgg.erase(it);
      vsl.verifiedIndSignatures.set(vs->from);
      bool improved = false;
      if (!vsl.totalIncoming.get(vs->from)) {
        vsl.totalIncoming.set(vs->from);
        improved = true;
      }
      BitSet all = vs->sig;
      all.or_(vsl.verifiedIndSignatures);
      if (all.cardinality() > vsl.verifiedIndSignatures.cardinality()) {
        improved = true;
        if (vsl.lastAggVerified.intersects(vs->sig)) vsl.lastAggVerified.clear();
        vsl.lastAggVerified.or_(vs->sig);
        vsl.totalIncoming.clear();
        vsl.totalIncoming.or_(vsl.lastAggVerified);
        vsl.totalIncoming.or_(vsl.verifiedIndSignatures);
      }
      if (!improved) return;
      bool justCompleted = vsl.incomingComplete();
      BitSet cur;
      for (auto& lp : levels) {
        HLevel& l = *lp;
        if (l.level > vsl.level) {
          l.totalOutgoing.clear();
          l.totalOutgoing.or_(cur);
          if (justCompleted && h.params.fastPath > 0 && !l.outgoingFinished && l.outgoingComplete()) {
            std::vector<HNode*> peers = l.getRemainingPeers(h.params.fastPath);
            auto sendSigs = std::make_shared<SendSigs>(l.totalOutgoing, l);
            std::vector<Node*> dests(peers.begin(), peers.end());
            h.network_.send(sendSigs, this, dests);
          }
        }
        cur.or_(l.totalIncoming);
      }
      if (doneAt == 0 && cur.cardinality() >= h.params.threshold) doneAt = h.network_.time;
    }
    void onNewSig(HNode* from, const SendSigs& ssigs) {  // :757-790
      if (doneAt > 0) {
        msgFiltered++;
        return;
      }
      if (h.network_.time < startAt || blacklist.get(from->nodeId)) return;
      HLevel& l = *levels[ssigs.level];
      if (!bitsetInclude(l.waitedSigs, ssigs.sigs)) throw IllegalStateException("bad signatures received");
      BitSet cs = ssigs.sigs;
      cs.and_(l.waitedSigs);
      if (!cs.equals(ssigs.sigs) || ssigs.sigs.isEmpty()) throw IllegalStateException("bad message");
      if (ssigs.levelFinished) l.finishedPeers.set(from->nodeId);
      if (!l.verifiedIndSignatures.get(from->nodeId)) l.toVerifyInd.set(from->nodeId);
      sigQueueSize++;
      auto stv = std::make_shared<SigToVerify>();
      stv->from = from->nodeId;
      stv->level = l.level;
      stv->rank = receptionRanks[from->nodeId];
      stv->sig = cs;
      stv->badSig = ssigs.badSig;
      l.toVerifyAgg.push_back(stv);
      h.statQueueMax[l.level] = std::max(h.statQueueMax[l.level], (int)l.toVerifyAgg.size());
    }
    void checkSigs() {  // :796-837
      std::vector<SigP> byLevels;
      for (auto& l : levels) {
        SigP ss = l->bestToVerify();
        if (!ss) continue;
        byLevels.push_back(ss);
      }
      if (byLevels.empty()) return;
      SigP best = byLevels[h.network_.rd.nextInt((jint)byLevels.size())];  // :788-790
      // trying to add a nearly useless signature in the list so that signatures with a lower rank are not retained (:813-817)
      if (hiddenByzantine && best->level == (int)levels.size() - 1) best = hiddenByzantine->attack(*this, best);
      if (!best) throw IllegalStateException("checkSigs: attack() left no signature to check");  // (Java would NPE at :819)
      HLevel& l = *levels[best->level];
      int newSize = h.params.window.newSize(currWindowSize, !best->badSig);
      currWindowSize = std::min(newSize, l.size);
      // receptionRanks[best.from] += nodeCount; if (< 0) = MAX_VALUE  (int overflow intended)
      jint r = (jint)((uint32_t)receptionRanks[best->from] + (uint32_t)h.params.nodeCount);
      if (r < 0) r = INT32_MAX;
      receptionRanks[best->from] = r;
      sigsChecked++;
      h.network_.registerTask([this, best] { updateVerifiedSignatures(best); }, h.network_.time + nodePairingTime,
                              this);
    }
  };

  HandelParameters params;
  Network network_;
  std::vector<std::unique_ptr<HNode>> nodes;
  // instrumentation (not in the reference)
  uint64_t statDeliveredByLevel[32] = {0};
  int statQueueMax[32] = {0};

  explicit Handel(const HandelParameters& p) : params(p) {  // :208-212
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
  }
  HNode* node(int i) { return nodes[i].get(); }
  Network& network() { return network_; }

  void setReceivingRanks() {  // :940-948
    std::vector<HNode*> expected;
    for (auto& n : nodes) expected.push_back(n.get());
    for (auto& n : nodes) {
      jshuffle(expected, network_.rd);
      for (int i = 0; i < (int)expected.size(); i++) n->receptionRanks[expected[i]->nodeId] = i;
    }
  }

  void init() {  // :957-1014
    std::unique_ptr<NodeBuilder> nb = nodeBuilderByName(params.nodeBuilderName);
    BitSet badNodes =
        params.badNodes != nullptr ? *params.badNodes : Network::chooseBadNodes(network_.rd, params.nodeCount, params.nodesDown);
    for (int i = 0; i < params.nodeCount; i++) {
      int startAt = params.desynchronizedStart == 0 ? 0 : network_.rd.nextInt(params.desynchronizedStart);
      bool byz = (params.byzantineSuicide | params.hiddenByzantine) && badNodes.get(i);
      nodes.push_back(std::make_unique<HNode>(*this, startAt, *nb, byz));
      HNode* n = nodes.back().get();
      if (badNodes.get(i)) n->stop();
      network_.addNode(n);
    }
    for (auto& np : nodes) {
      HNode* n = np.get();
      n->initLevel();
      if (!n->isDown()) {
        network_.registerPeriodicTask([n] { n->dissemination(); }, n->startAt + 1, params.disseminationPeriodMs, n);
        network_.registerConditionalTask([n] { n->checkSigs(); }, n->startAt + 1, n->nodePairingTime, n,
                                         [n] { return n->hasSigToVerify(); }, [n] { return !n->done; });
      }
    }
    setReceivingRanks();
    // Emission lists (:991-1013 + buildEmissionList :510-522). The reference buckets receivers into
    // a List[nodeCount] indexed by reception rank and walks it; walking rank-ascending over non-empty
    // buckets (insertion order inside a bucket) is the same as a stable sort by rank.
    for (auto& sp : nodes) {
      HNode* sender = sp.get();
      if (sender->isDown()) continue;
      for (auto& lp : sender->levels) {
        HLevel& l = *lp;
        std::vector<std::pair<int, HNode*>> byRank;
        for (HNode* receiver : l.expectedNodes()) byRank.emplace_back(receiver->receptionRanks[sender->nodeId], receiver);
        std::stable_sort(byRank.begin(), byRank.end(),
                         [](const std::pair<int, HNode*>& a, const std::pair<int, HNode*>& b) { return a.first < b.first; });
        if (!l.peers.empty()) throw IllegalStateException("peers");
        for (size_t i = 0; i < byRank.size();) {
          size_t j = i;
          std::vector<HNode*> ranks;
          while (j < byRank.size() && byRank[j].first == byRank[i].first) ranks.push_back(byRank[j++].second);
          if (ranks.size() > 1) jshuffle(ranks, network_.rd);
          l.peers.insert(l.peers.end(), ranks.begin(), ranks.end());
          i = j;
        }
      }
    }
  }

  bool contIf() {  // newContIf :1044-1053
    for (auto& n : nodes)
      if (!n->isDown() && (n->doneAt == 0 || n->addedCycle > 0)) return true;
    return false;
  }
};

inline Handel::SendSigs::SendSigs(const BitSet& s, const HLevel& l) : level(l.level), sigs(s) {  // :253-265
  size_ = 1 + l.expectedSigs() / 8 + 96 * 2;
  levelFinished = l.incomingComplete();
  if (sigs.isEmpty() || sigs.cardinality() > l.size) throw IllegalStateException("bad level: " + std::to_string(l.level));
}
inline void Handel::SendSigs::action(Network&, Node* from, Node* to) {  // :272-275
  HNode* t = static_cast<HNode*>(to);
  t->h.statDeliveredByLevel[level]++;
  t->onNewSig(static_cast<HNode*>(from), *this);
}

}  // namespace orc
