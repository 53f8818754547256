// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.CasperIMD (P/CasperIMD.java:14-707) with core.Block / BlockChainNode /
// BlockChainNetwork (C/Block.java:4-117, C/BlockChainNode.java:6-75, C/BlockChainNetwork.java:9-38).
//
// Pinned against every value PT/CasperIMDTest.java and PT/CasperByzantineTest.java hold (oracle/test_casper.cpp).
// Two places where the Java result depends on something the JDK leaves unspecified, and what the oracle does:
//   * `for (CasperBlock b : blocksToReevaluate)` (:352-356) iterates a HashSet of objects without hashCode(), i.e.
//     in identity-hash order, which differs from JVM run to JVM run. best() is a maximum under the fork-choice
//     preference and with randomOnTies == false the outcome is the same for every order in which the reference's
//     own tests exercise it; the oracle iterates in ascending block id.
//   * Block.blockId (:10) is a JVM-wide static counter: ids depend on what else ran in the JVM, only their order
//     is meaningful (:best tie-break :261). The oracle counts per protocol instance.
//   * ByzBlockProducer / SF / NS (:545-633) take `blocksReceivedByHeight.get(h).iterator().next()` from a HashSet<Block> —
//     again identity-hash order when a node holds several blocks of one height; the oracle takes the block with the
//     smallest id (its sets are ordered by block id). No test of the reference uses the three classes (init() installs
//     a ByzBlockProducerWF, :475-479), so they are pinned by nothing but this restatement of their lines; the host-mode
//     mirror (examples/hostmode/casper.py) and the engine are checked against it.
#pragma once
#include <map>
#include <set>
#include "network.hpp"

namespace orc {

class CasperIMD {
 public:
  struct CasperParemeters {  // :18-70 (sic)
    static constexpr int SLOT_DURATION = 8000;
    int cycleLength = 4;
    bool randomOnTies = true;
    int blockProducersCount = 2;
    int attestersPerRound = 20;
    int attestersCount = 80;
    int blockConstructionTime = 1000;
    int attestationConstructionTime = 1;
    std::string nodeBuilderName, networkLatencyName;
    CasperParemeters() {}
    CasperParemeters(int cl, bool rot, int bpc, int apr, int bct, int act, const std::string& nb, const std::string& nl)
        : cycleLength(cl), randomOnTies(rot), blockProducersCount(bpc), attestersPerRound(apr), attestersCount(apr * cl),
          blockConstructionTime(bct), attestationConstructionTime(act), nodeBuilderName(nb), networkLatencyName(nl) {}
  };

  struct CasperNode;
  struct Attestation;
  struct ByAid {
    bool operator()(const Attestation* a, const Attestation* b) const;
  };
  typedef std::set<const Attestation*, ByAid> AttSet;

  struct CasperBlock {  // C/Block.java:4-117 + P/CasperIMD.java:129-175
    int height = 0;
    int proposalTime = 0;
    jlong lastTxId = 0;
    jlong id = 0;
    CasperBlock* parent = nullptr;
    CasperNode* producer = nullptr;
    bool valid = true;
    std::map<int, AttSet> attestationsByHeight;
    CasperBlock() {}  // genesis: Block(0) :22-30
    CasperBlock(jlong newId, CasperNode* prod, int h, CasperBlock* father, std::map<int, AttSet> abh, int time)
        : height(h), proposalTime(time), lastTxId(time), id(newId), parent(father), producer(prod),
          attestationsByHeight(std::move(abh)) {  // :36-54
      if (h <= 0) throw IllegalArgumentException("Only the genesis block has a special height");
      if (father != nullptr && time < father->proposalTime) throw IllegalArgumentException("bad time");
      if (father != nullptr && father->height >= h) throw IllegalArgumentException("Bad parent");
    }
    bool hasDirectLink(const CasperBlock* b) const {  // C/Block.java:86-99
      if (b == this) return true;
      if (b->height == height) return false;
      const CasperBlock* older = height > b->height ? this : b;
      const CasperBlock* young = height < b->height ? this : b;
      while (older->height > young->height) older = older->parent;
      return older == young;
    }
    jlong txCount() const { return id == 0 ? 0 : lastTxId - parent->lastTxId; }  // :57-67
  };

  struct Attestation : Message {  // :98-127
    CasperIMD& p;
    int aid;  // creation ordinal (oracle bookkeeping: a total order for the sets)
    CasperNode* attester;
    int height;
    std::set<jlong> hs;
    CasperBlock* head;
    Attestation(CasperIMD& pp, CasperNode* at, int h);
    void action(Network&, Node*, Node* to) override;
    bool attests(const CasperBlock* cb) const { return hs.count(cb->id) != 0; }
  };

  struct SendBlock : Message {  // C/BlockChainNetwork.java:22-38
    CasperBlock* toSend;
    explicit SendBlock(CasperBlock* b) : toSend(b) {}
    void action(Network&, Node*, Node* to) override;
  };

  struct ByBlockId {
    bool operator()(const CasperBlock* a, const CasperBlock* b) const { return a->id < b->id; }
  };

  struct CasperNode : Node {  // C/BlockChainNode.java:6-75 + P/CasperIMD.java:177-368
    CasperIMD& p;
    CasperBlock* genesis;
    std::map<jlong, CasperBlock*> blocksReceivedByBlockId;
    std::map<jlong, std::set<CasperBlock*, ByBlockId>> blocksReceivedByFatherId;
    std::map<int, std::set<CasperBlock*, ByBlockId>> blocksReceivedByHeight;
    CasperBlock* head;
    std::map<jlong, AttSet> attestationsByHead;
    std::set<CasperBlock*, ByBlockId> blocksToReevaluate;

    CasperNode(CasperIMD& pp, bool byz) : Node(pp.network_.rd, *pp.nb, byz), p(pp), genesis(&pp.genesis), head(&pp.genesis) {
      blocksReceivedByBlockId[genesis->id] = genesis;
    }
    virtual std::function<void()> periodicTask() { return nullptr; }

    bool baseOnBlock(CasperBlock* b) {  // BlockChainNode.onBlock C/BlockChainNode.java:29-47
      if (!b->valid) return false;
      if (!blocksReceivedByBlockId.emplace(b->id, b).second) return false;
      blocksReceivedByFatherId[b->parent->id].insert(b);
      blocksReceivedByHeight[b->height].insert(b);
      head = best(head, b);
      return true;
    }

    CasperBlock* best(CasperBlock* o1, CasperBlock* o2) {  // :186-236
      if (o1 == o2) return o1;
      if (o1->height == o2->height) throw IllegalStateException("two blocks for the same height");
      if (o1->hasDirectLink(o2)) return o1->height < o2->height ? o2 : o1;
      CasperBlock* b1 = o1;
      CasperBlock* b2 = o2;
      while (b1->parent != b2->parent) {
        if (b1->parent->height > b2->parent->height)
          b1 = b1->parent;
        else
          b2 = b2->parent;
      }
      CasperBlock* h = b1->parent;
      int b1Votes = countAttestations(o1, h);
      int b2Votes = countAttestations(o2, h);
      if (b1Votes > b2Votes) return o1;
      if (b1Votes < b2Votes) return o2;
      if (p.params.randomOnTies) return p.network_.rd.nextBoolean() ? o1 : o2;
      return b1->id >= b2->id ? o1 : o2;
    }

    int countAttestations(CasperBlock* start, CasperBlock* h) {  // :241-266
      AttSet a1;
      for (CasperBlock* cur = start; cur != h; cur = cur->parent) {
        for (int i = cur->height - 1; i > h->height; i--) {
          auto it = cur->attestationsByHeight.find(i);
          if (it == cur->attestationsByHeight.end()) continue;
          for (const Attestation* a : it->second)
            if (a->attests(h)) a1.insert(a);
        }
        auto it = attestationsByHead.find(cur->id);
        if (it != attestationsByHead.end())
          for (const Attestation* a : it->second)
            if (a->attests(h)) a1.insert(a);
      }
      return (int)a1.size();
    }

    virtual bool onBlock(CasperBlock* b) {  // :276-292
      const int delta = p.network_.time - genesis->proposalTime + b->height * CasperParemeters::SLOT_DURATION;
      if (delta >= 0) {
        blocksToReevaluate.insert(head);
        blocksToReevaluate.insert(b);
        return baseOnBlock(b);
      }
      p.network_.registerTask([this, b] { onBlock(b); }, delta * -1, this);
      return false;
    }

    void onAttestation(const Attestation* a) {  // :294-337
      attestationsByHead[a->head->id].insert(a);
      if (blocksReceivedByBlockId.count(a->head->id)) blocksToReevaluate.insert(a->head);
    }

    void reevaluateHead() {  // :349-356 (iteration order: see the header)
      for (CasperBlock* b : blocksToReevaluate) head = best(head, b);
      blocksToReevaluate.clear();
    }
  };

  struct BlockProducer : CasperNode {  // :370-443
    BlockProducer(CasperIMD& pp, bool byz = false) : CasperNode(pp, byz) {}
    std::function<void()> periodicTask() override {
      return [this] {
        reevaluateHead();
        createAndSendBlock(p.network_.time / CasperParemeters::SLOT_DURATION);
      };
    }
    CasperBlock* buildBlock(CasperBlock* base, int height) {  // :389-434
      const int cl = p.params.cycleLength;
      std::map<int, AttSet> res;
      for (int i = height - 1; i >= 0 && i >= height - cl; i--) res[i];
      AttSet allFromBlocks;
      for (CasperBlock* cur = base; cur != genesis && cur->height >= height - cl; cur = cur->parent)
        for (auto& kv : cur->attestationsByHeight) allFromBlocks.insert(kv.second.begin(), kv.second.end());
      for (CasperBlock* cur = base; cur != nullptr && cur->height >= height - cl; cur = cur->parent) {
        auto it = attestationsByHead.find(cur->id);
        if (it == attestationsByHead.end()) continue;
        for (const Attestation* a : it->second)
          if (a->height < height && !allFromBlocks.count(a)) res[a->height].insert(a);
      }
      return p.newBlock(this, height, base, std::move(res), p.network_.time);
    }
    void createAndSendBlock(int height) {  // :436-442
      head = buildBlock(head, height);
      p.network_.sendAll(std::make_shared<SendBlock>(head), p.network_.time + p.params.blockConstructionTime, this);
    }
  };

  struct Attester : CasperNode {  // :445-473
    explicit Attester(CasperIMD& pp) : CasperNode(pp, false) {}
    std::function<void()> periodicTask() override {
      return [this] { vote(p.network_.time / CasperParemeters::SLOT_DURATION); };
    }
    void vote(int height) {
      reevaluateHead();
      auto v = p.newAttestation(this, height);
      p.network_.sendAll(v, p.network_.time + p.params.attestationConstructionTime, this);
    }
  };

  struct ByzBlockProducer : BlockProducer {  // :511-581
    int toSend = 1, h = 0;
    const int delay;
    int onDirectFather = 0, onOlderAncestor = 0, incNotTheBestFather = 0;
    ByzBlockProducer(CasperIMD& pp, int d) : BlockProducer(pp, true), delay(d) {}
    void reevaluateH(int time) {  // :529-543
      reevaluateHead();
      while (head->height >= toSend) head = head->parent;
      const int slotTime = time - delay;
      h = slotTime / CasperParemeters::SLOT_DURATION;
      if (h != toSend) throw IllegalStateException("h=" + std::to_string(h) + ", toSend=" + std::to_string(toSend));
    }
  };

  // the plain delayed producer (ByzBlockProducer itself, :545-563), "skip father" (:583-604) and "no skip" (:610-633)
  struct ByzBlockProducerPlain : ByzBlockProducer {
    ByzBlockProducerPlain(CasperIMD& pp, int d) : ByzBlockProducer(pp, d) {}
    std::function<void()> periodicTask() override {  // :545-563
      return [this] {
        reevaluateH(p.network_.time);
        if (head->height == h - 1) {
          onDirectFather++;
        } else {
          onOlderAncestor++;
          // Block possibleFather = blocksReceivedByHeight.get(h - 1).iterator().next(): a NullPointerException when no
          // block of that height was received, as in the reference
          auto it = blocksReceivedByHeight.find(h - 1);
          if (it == blocksReceivedByHeight.end() || it->second.empty()) throw IllegalStateException("NullPointerException (:555)");
          CasperBlock* possibleFather = *it->second.begin();
          if (possibleFather != nullptr && possibleFather->parent->height != h - 1) incNotTheBestFather++;
        }
        createAndSendBlock(toSend);
        toSend += p.params.blockProducersCount;
      };
    }
  };
  struct ByzBlockProducerSF : ByzBlockProducer {  // :583-604 — skip its father's block
    ByzBlockProducerSF(CasperIMD& pp, int d) : ByzBlockProducer(pp, d) {}
    std::function<void()> periodicTask() override {
      return [this] {
        reevaluateH(p.network_.time);
        if (head->id != 0 && head->height == h - 1) {
          head = head->parent;
          onDirectFather++;
        } else {
          onOlderAncestor++;
        }
        createAndSendBlock(toSend);
        toSend += p.params.blockProducersCount;
      };
    }
  };
  struct ByzBlockProducerNS : ByzBlockProducer {  // :610-633 — skip the father if the father skipped the grand father
    int skipped = 0;
    ByzBlockProducerNS(CasperIMD& pp, int d) : ByzBlockProducer(pp, d) {}
    std::function<void()> periodicTask() override {
      return [this] {
        reevaluateH(p.network_.time);
        if (head->id != 0 && head->height == h - 1 && head->parent->height == h - 3) {
          auto it = blocksReceivedByHeight.find(h - 2);
          if (it == blocksReceivedByHeight.end() || it->second.empty()) throw IllegalStateException("NullPointerException (:619)");
          CasperBlock* b = *it->second.begin();
          if (b != nullptr) {
            head = b;
            skipped++;
          }
        }
        createAndSendBlock(toSend);
        toSend += p.params.blockProducersCount;
      };
    }
  };

  struct ByzBlockProducerWF : ByzBlockProducer {  // :635-692
    int late = 0, onTime = 0;
    ByzBlockProducerWF(CasperIMD& pp, int d) : ByzBlockProducer(pp, d) {}
    std::function<void()> periodicTask() override {
      return [this] {
        if (head == genesis && toSend == 1) {
          reevaluateH(p.network_.time);
          createAndSendBlock(h);
          toSend += p.params.blockProducersCount;
        }
      };
    }
    bool onBlock(CasperBlock* b) override {
      if (!CasperNode::onBlock(b)) return false;
      if (b->height == toSend - 1) {
        const int perfectDate = CasperParemeters::SLOT_DURATION * toSend + delay;
        const int th = toSend;
        auto r = [this, b, th] {
          head = buildBlock(b, th);
          p.network_.sendAll(std::make_shared<SendBlock>(head), p.network_.time + p.params.blockConstructionTime, this);
        };
        toSend += p.params.blockProducersCount;
        if (p.network_.time >= perfectDate) {
          r();
          late++;
        } else {
          p.network_.registerTask(r, perfectDate, this);
          onTime++;
        }
      }
      return true;
    }
  };

  CasperParemeters params;
  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  CasperBlock genesis;
  CasperNode* observer = nullptr;                 // BlockChainNetwork.observer C/BlockChainNetwork.java:14-19
  std::vector<Attester*> attesters;               // :94
  std::vector<BlockProducer*> bps;                // :95
  std::vector<std::unique_ptr<CasperNode>> nodes; // owner
  std::vector<std::unique_ptr<CasperBlock>> blocks;
  std::vector<std::shared_ptr<Attestation>> attestations;
  jlong nextBlockId = 1;                          // C/Block.java:10
  int nextAid = 0;

  explicit CasperIMD(const CasperParemeters& pr) : params(pr) {  // :80-87
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
    nodes.push_back(std::make_unique<CasperNode>(*this, false));
    observer = nodes.back().get();
    network_.addNode(observer);
  }
  Network& network() { return network_; }

  CasperBlock* newBlock(CasperNode* prod, int h, CasperBlock* father, std::map<int, AttSet> abh, int time) {
    blocks.push_back(std::make_unique<CasperBlock>(nextBlockId, prod, h, father, std::move(abh), time));
    nextBlockId++;  // (the Java ctor increments after its argument checks pass, C/Block.java:49)
    return blocks.back().get();
  }
  std::shared_ptr<Attestation> newAttestation(CasperNode* at, int h) {
    attestations.push_back(std::make_shared<Attestation>(*this, at, h));
    return attestations.back();
  }
  template <class T, class... A>
  T* make(A&&... a) {  // nodes created outside init() (the unit tests' bp1, at1, ...)
    nodes.push_back(std::make_unique<T>(*this, std::forward<A>(a)...));
    return static_cast<T*>(nodes.back().get());
  }

  void init() { init(make<ByzBlockProducerWF>(0)); }  // :475-479
  void init(ByzBlockProducer* byzantineNode) {        // :481-509
    const int SD = CasperParemeters::SLOT_DURATION;
    bps.push_back(byzantineNode);
    network_.addNode(byzantineNode);
    network_.registerPeriodicTask(byzantineNode->periodicTask(), SD + byzantineNode->delay, SD * params.blockProducersCount,
                                  byzantineNode);
    for (int i = 1; i < params.blockProducersCount; i++) {
      BlockProducer* n = make<BlockProducer>();
      bps.push_back(n);
      network_.addNode(n);
      network_.registerPeriodicTask(n->periodicTask(), SD * (i + 1), SD * params.blockProducersCount, n);
    }
    for (int i = 0; i < params.attestersCount; i++) {
      Attester* n = make<Attester>();
      attesters.push_back(n);
      network_.addNode(n);
      network_.registerPeriodicTask(n->periodicTask(), SD * (1 + i % params.cycleLength) + 4000, SD * params.cycleLength, n);
    }
  }
};

inline bool CasperIMD::ByAid::operator()(const Attestation* a, const Attestation* b) const { return a->aid < b->aid; }

inline CasperIMD::Attestation::Attestation(CasperIMD& pp, CasperNode* at, int h)
    : p(pp), aid(pp.nextAid++), attester(at), height(h), head(at->head) {  // :107-121
  for (CasperBlock* cur = at->head->parent; cur != nullptr && cur->height >= at->head->height - pp.params.cycleLength;
       cur = cur->parent)
    hs.insert(cur->id);
}
inline void CasperIMD::Attestation::action(Network&, Node*, Node* to) { static_cast<CasperNode*>(to)->onAttestation(this); }
inline void CasperIMD::SendBlock::action(Network&, Node*, Node* to) { static_cast<CasperNode*>(to)->onBlock(toSend); }

}  // namespace orc
