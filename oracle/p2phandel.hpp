// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.P2PHandel (P/P2PHandel.java:20-520) over core.P2PNetwork (p2pflood.hpp: C/P2PNetwork.java with
// minimum == false, :113): every node sends, every sigsSendPeriod, what the neediest of its peers lacks (SendSigs), keeps what
// it receives in a HashSet<BitSet> (`toVerify`) and verifies it on a conditional task — the best one (checkSigs1) or the union
// of all (checkSigs2, `doubleAggregateStrategy`) — after 2 x pairingTime.
// What makes the trajectory depend on the JDK: checkSigs2 ORs the set's other elements INTO the first one its iterator
// returns (:455-464), and that BitSet is the `sigs` object of a SendSigs message — one object for every receiver of a
// multi-destination send (sendFinalSigToPeers :305-317) and for every set it was added to. Which element comes first is
// java.util.HashSet's iteration order over BitSet.hashCode() of elements that may have been mutated since they were added:
// JHashSet (jdk.hpp) restates it (hash at insertion, bucket order, resizes; a treeified bucket is refused loudly).
// A bucket of nine elements in a table of 64 or more becomes a red-black tree in the JDK (HashMap.treeifyBin), iterated in an
// order that follows the tree's shape and, for equal hashes, System.identityHashCode: not restatable, refused loudly. It
// happens where the single-best strategy (checkSigs1) lets the sets grow — BitSet hashes of small sets cluster —, e.g. with
// PT/P2PHandelTest.testRepeatability's own parameters; checkSigs2 empties the set at every check and never gets there.
// Pinned against PT/P2PHandelTest.java: testCompressedSize's fourteen values, testSetup, testCheckSigs, testSigUpdate,
// the two runs to completion and testRepeatability, in tests/test_oracle_protocols.py. The HashSet order itself is pinned by
// nothing the reference holds (no JVM in the image): "parity unpinned" for it.
#pragma once
#include "p2pflood.hpp"

namespace orc {

class P2PHandel {
 public:
  enum SendSigsStrategy { all = 0, dif = 1, cmp_all = 2, cmp_diff = 3 };  // :25-30
  struct Params {  // P2PHandelParameters :36-109, ctor order
    int signingNodeCount = 100, relayingNodeCount = 20, threshold = 99, connectionCount = 40, pairingTime = 100, sigsSendPeriod = 1000;
    bool doubleAggregateStrategy = true;
    SendSigsStrategy sendSigsStrategy = dif;
    bool sendState = false;
    std::string nodeBuilderName, networkLatencyName;
  };
  typedef std::shared_ptr<BitSet> Bits;
  static Bits cloneBits(const BitSet& b) { return std::make_shared<BitSet>(b); }
  struct P2PHandelNode;
  struct State : Message {  // :119-140
    const Bits desc;
    P2PHandelNode* const who;
    explicit State(P2PHandelNode* w);
    int size() const override { return std::max(1, desc->length() / 8); }
    void action(Network&, Node*, Node* to) override;
  };
  struct SendSigs : Message {  // :231-253
    const Bits sigs;  // ONE object for every receiver: what onNewSig adds to toVerify (and checkSigs2 may OR into)
    const int size_;
    explicit SendSigs(const BitSet& s) : sigs(cloneBits(s)), size_(std::max(1, s.cardinality())) {}
    SendSigs(const BitSet& s, int sigCount) : sigs(cloneBits(s)), size_(std::max(1, sigCount)) {}
    int size() const override { return size_; }
    void action(Network&, Node* from, Node* to) override;
  };
  struct P2PHandelNode : P2PNode {  // :255-481
    P2PHandel& p;
    BitSet verifiedSignatures;
    JHashSet<BitSet> toVerify;
    std::map<int, Bits> peersState;  // (a HashMap there: get / put only)
    const bool justRelay;
    P2PHandelNode(P2PHandel& pp, bool relay) : P2PNode(pp.network_.rd, *pp.nb, false), p(pp), justRelay(relay) {
      if (!justRelay) verifiedSignatures.set(nodeId);
    }
    void start() override {  // :270-275
      P2PNode::start();
      for (P2PNode* q : peers) peersState[q->nodeId] = std::make_shared<BitSet>();
    }
    P2PHandelNode* peer(size_t i) const { return static_cast<P2PHandelNode*>(peers[i]); }
    void onPeerState(const State& st) { peersState.at(st.who->nodeId)->or_(*st.desc); }  // :281-283
    void updateVerifiedSignatures(const BitSet& sigs) {  // :290-303
      const int oldCard = verifiedSignatures.cardinality();
      verifiedSignatures.or_(sigs);
      const int newCard = verifiedSignatures.cardinality();
      if (newCard > oldCard) {
        if (doneAt == 0 && verifiedSignatures.cardinality() >= p.params.threshold) {
          doneAt = p.network_.time;
          sendFinalSigToPeers();
        } else if (doneAt == 0 && p.params.sendState) {
          sendStateToPeers();
        }
      }
    }
    void sendFinalSigToPeers() {  // :305-317
      std::vector<Node*> dest;
      for (P2PNode* q : peers) {
        BitSet& ps = *peersState.at(q->nodeId);
        if (ps.cardinality() < p.params.threshold) {
          dest.push_back(q);
          ps.or_(verifiedSignatures);
        }
      }
      p.network_.send(std::make_shared<SendSigs>(verifiedSignatures, 1), this, dest);
    }
    void sendStateToPeers() {  // :319-322
      std::vector<Node*> dest(peers.begin(), peers.end());
      p.network_.send(std::make_shared<State>(this), this, dest);
    }
    void onNewSig(Node* from, const Bits& sigs) {  // :325-328
      peersState.at(from->nodeId)->or_(*sigs);
      toVerify.add(sigs);
    }
    BitSet diff(P2PHandelNode* q) const {  // :356-360
      BitSet needed = verifiedSignatures;
      needed.andNot(*peersState.at(q->nodeId));
      return needed;
    }
    P2PHandelNode* bestDest() const {  // :367-378
      P2PHandelNode* dest = nullptr;
      int destSize = 0;
      for (size_t i = 0; i < peers.size(); i++) {
        const int size = diff(peer(i)).cardinality();
        if (size > destSize) {
          dest = peer(i);
          destSize = size;
        }
      }
      return dest;
    }
    std::shared_ptr<SendSigs> createSendSigs(const BitSet& toSend) const {  // :389-404
      switch (p.params.sendSigsStrategy) {
        case dif: return std::make_shared<SendSigs>(toSend);
        case cmp_all: return std::make_shared<SendSigs>(verifiedSignatures, p.compressedSize(verifiedSignatures));
        case cmp_diff: return std::make_shared<SendSigs>(verifiedSignatures, std::min(p.compressedSize(verifiedSignatures), p.compressedSize(toSend)));
        default: return std::make_shared<SendSigs>(verifiedSignatures);
      }
    }
    void sendSigs() {  // :336-354
      if (doneAt > 0) return;
      P2PHandelNode* dest = bestDest();
      if (!dest) return;  // nobody needs anything from us right now
      const BitSet toSend = diff(dest);
      peersState.at(dest->nodeId)->or_(verifiedSignatures);
      p.network_.send(createSendSigs(toSend), this, dest);
    }
    void registerUpdate(const Bits& tBest) {
      p.network_.registerTask([this, tBest] { updateVerifiedSignatures(*tBest); }, p.network_.time + p.params.pairingTime * 2, this);
    }
    void checkSigs() {  // :406-412
      if (p.params.doubleAggregateStrategy)
        checkSigs2();
      else
        checkSigs1();
    }
    void checkSigs1() {  // :419-449: the best signature (the first of the strictly greatest gain, in iteration order)
      Bits best;
      int bestV = 0;
      for (const Bits& o1 : toVerify.items()) {
        BitSet oo1 = *o1;
        oo1.andNot(verifiedSignatures);
        const int v1 = oo1.cardinality();
        if (v1 == 0) {
          toVerify.removeNode(o1);  // it.remove()
        } else if (v1 > bestV) {
          bestV = v1;
          best = o1;
        }
      }
      if (best) {
        toVerify.remove(best);
        registerUpdate(best);
      }
    }
    void checkSigs2() {  // :455-480: everything at once — ORed INTO the first element the iterator returns
      Bits agg;
      for (const Bits& o1 : toVerify.items()) {
        if (!agg)
          agg = o1;
        else
          agg->or_(*o1);
      }
      toVerify.clear();
      if (agg) {
        BitSet oo1 = *agg;
        oo1.andNot(verifiedSignatures);
        if (oo1.cardinality() > 0) registerUpdate(agg);
      }
    }
  };

  Params params;
  P2PNetwork network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<P2PHandelNode>> nodes;
  explicit P2PHandel(const Params& pr) : params(pr), network_(pr.connectionCount, false) {  // :111-117
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
  }
  // ---- compressedSize (:160-202) / mergeRanges (:204-229): how many signatures a set of them costs once the aligned ranges
  // of complete pairs are merged (pinned by PT/P2PHandelTest.testCompressedSize)
  int compressedSize(const BitSet& sigs) const {
    if (sigs.length() == params.signingNodeCount) return 1;
    int firstOneAt = -1, sigCt = 0, pos = -1;
    bool compressing = false, wasCompressing = false;
    while (++pos <= sigs.length() + 1) {
      if (!sigs.get(pos)) {
        compressing = false;
        sigCt -= mergeRanges(firstOneAt, pos);
        firstOneAt = -1;
      } else if (compressing) {
        if ((pos + 1) % 2 == 0) {
          compressing = false;
          wasCompressing = true;
        }
      } else {
        sigCt++;
        if (pos % 2 == 0) {
          compressing = true;
          if (!wasCompressing)
            firstOneAt = pos;
          else
            wasCompressing = false;
        }
      }
    }
    return sigCt;
  }
  static int log2i(int n) {  // C/utils/MoreMath.java:5-10
    if (n <= 0) throw IllegalArgumentException("log2");
    return 31 - __builtin_clz((unsigned)n);
  }
  int mergeRanges(int firstOneAt, int pos) const {
    if (firstOneAt < 0) return 0;
    if (firstOneAt % (2 * 2) != 0) firstOneAt += (2 * 2) - (firstOneAt % (2 * 2));
    const int rangeCt = (pos - firstOneAt) / 2;
    if (rangeCt < 2) return 0;
    int max = log2i(rangeCt);
    while (max > 0) {
      const int sizeInBlocks = 1 << max;  // (int) Math.pow(2, max)
      const int size = sizeInBlocks * 2;
      if (firstOneAt % size == 0) return (sizeInBlocks - 1) + mergeRanges(firstOneAt + size, pos);
      max--;
    }
    return 0;
  }
  void init() {  // :483-510
    const int total = params.signingNodeCount + params.relayingNodeCount;
    std::set<int> justRelay;  // (a HashSet there: add / contains / size only)
    while ((int)justRelay.size() < params.relayingNodeCount) justRelay.insert(network_.rd.nextInt(total));
    for (int i = 0; i < total; i++) {
      nodes.push_back(std::make_unique<P2PHandelNode>(*this, justRelay.count(i) != 0));
      P2PHandelNode* n = nodes.back().get();
      network_.addNode(n);
      if (params.sendState) network_.registerTask([n] { n->sendStateToPeers(); }, 1, n);
      network_.registerPeriodicTask([n] { n->sendSigs(); }, 1, params.sigsSendPeriod, n);
      network_.registerConditionalTask([n] { n->checkSigs(); }, 1, params.pairingTime, n, [n] { return !n->toVerify.isEmpty(); },
                                       [n] { return n->doneAt == 0; });
    }
    network_.setPeers();
  }
};
inline P2PHandel::State::State(P2PHandelNode* w) : desc(cloneBits(w->verifiedSignatures)), who(w) {}
inline void P2PHandel::State::action(Network&, Node*, Node* to) { static_cast<P2PHandelNode*>(to)->onPeerState(*this); }
inline void P2PHandel::SendSigs::action(Network&, Node* from, Node* to) { static_cast<P2PHandelNode*>(to)->onNewSig(from, sigs); }

}  // namespace orc
