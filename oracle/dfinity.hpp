// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.Dfinity (P/Dfinity.java:12-480) with the block-chain classes it stands on (C/Block.java:4-117,
// C/BlockChainNode.java:6-75, C/BlockChainNetwork.java:10-43): an observer, attesters, block producers and random-beacon nodes on
// one Network. A random-beacon committee exchanges "signatures" (RandomBeaconExchange) and publishes the beacon of a height
// (RandomBeaconResult, a sendAll); the producers whose round the beacon selects propose a block to the attesters in a shuffled
// order (one multi-destination send at an explicit sendTime, C/Network.java:418-447), the attesters of the selected round vote
// among themselves and the one that sees a majority sends the block to everybody (SendBlock, sendAll); a block starts the next
// beacon round.
// Pinned against PT/DfinityTest.java:10-26 (testRun: 10 producers, 10 attesters all in one round, no latency: the observer's head
// is at height 3 after run(11)) in tests/test_oracle_protocols.py; seed-dependent trajectories are unpinned (no JVM in the image).
// Where the reference leaves things open: the block id is a JVM-wide static counter there (C/Block.java:11) and a per-protocol one
// here (ids only have to be unique and increasing inside a run); DfinityParameters' three node lists (:30-32) belong to the
// parameter object there, so copies of one protocol share and grow them — here every protocol instance has its own.
#pragma once
#include <map>
#include <set>
#include "network.hpp"

namespace orc {

class Dfinity {
 public:
  struct Params {  // DfinityParameters :14-71, ctor order
    int blockProducersCount = 10, attestersCount = 10, attestersPerRound = 10, blockConstructionTime = 1,
        attestationConstructionTime = 1, percentageDeadAttester = 0;
    std::string nodeBuilderName, networkLatencyName;
    static constexpr int roundTime = 3000;           // :15-16
    static constexpr int blockProducersPerRound = 5;  // :18
    int blockProducersRound() const { return blockProducersCount / blockProducersPerRound; }
    int attestersRound() const { return attestersCount / attestersPerRound; }
    int randomBeaconCount() const { return attestersPerRound; }  // :56-57
    int majority() const { return attestersPerRound / 2 + 1; }
  };
  struct DfinityNode;
  struct DfinityBlock {  // C/Block.java + P/Dfinity.java:92-105
    int height = 0, proposalTime = 0;
    jlong lastTxId = 0, id = 0;
    const DfinityBlock* parent = nullptr;
    const DfinityNode* producer = nullptr;
    bool valid = true;
    bool hasDirectLink(const DfinityBlock* b) const {  // C/Block.java:86-99
      if (b == this) return true;
      if (b->height == height) return false;
      const DfinityBlock* older = height > b->height ? this : b;
      const DfinityBlock* young = height < b->height ? this : b;
      while (older->height > young->height) older = older->parent;
      return older == young;
    }
  };
  static int compareBlocks(const DfinityBlock* o1, const DfinityBlock* o2) {  // DfinityBlockComparator :107-130
    if (o1 == o2) return 0;
    if (!o2->valid) return 1;
    if (!o1->valid) return -1;
    if (o1->hasDirectLink(o2)) return o1->height < o2->height ? -1 : 1;
    if (o1->height != o2->height) return o1->height < o2->height ? -1 : 1;
    return 0;  // Long.compare(o1.producer.nodeId, o1.producer.nodeId): o1 against ITSELF in the reference (:128)
  }
  struct DfinityNode : Node {  // C/BlockChainNode.java + :188-213
    Dfinity& p;
    const DfinityBlock* head;
    std::map<jlong, const DfinityBlock*> blocksReceivedByBlockId;
    std::set<jlong> committeeMajorityBlocks;
    std::set<int> committeeMajorityHeight;
    int lastRandomBeacon = 0;
    explicit DfinityNode(Dfinity& pp) : Node(pp.network_.rd, *pp.nb, false), p(pp), head(&pp.genesis) {
      blocksReceivedByBlockId[pp.genesis.id] = &pp.genesis;
    }
    virtual ~DfinityNode() {}
    const DfinityBlock* best(const DfinityBlock* o1, const DfinityBlock* o2) const {  // :194-196
      return compareBlocks(o1, o2) >= 0 ? o1 : o2;
    }
    bool baseOnBlock(const DfinityBlock* b) {  // BlockChainNode.onBlock C/BlockChainNode.java:29-47
      if (!b->valid) return false;
      if (!blocksReceivedByBlockId.emplace(b->id, b).second) return false;
      head = best(head, b);
      return true;
    }
    virtual bool onBlock(const DfinityBlock* b) { return baseOnBlock(b); }
    virtual void onVote(Node*, const DfinityBlock*) {}  // :202
    void onRandomBeacon(int height, jlong rd) {          // :205-210
      if (lastRandomBeacon < height) {
        lastRandomBeacon = height;
        onRandomBeaconOnce(height, rd);
      }
    }
    virtual void onRandomBeaconOnce(int, jlong) {}  // :212
  };
  // ---- messages (:132-186, C/BlockChainNetwork.java:22-38)
  struct BlockProposal : Message {
    const DfinityBlock* block;
    explicit BlockProposal(const DfinityBlock* b) : block(b) {}
    void action(Network&, Node*, Node* to) override;
  };
  struct Vote : Message {
    const DfinityBlock* voteFor;
    explicit Vote(const DfinityBlock* b) : voteFor(b) {}
    void action(Network&, Node* from, Node* to) override { static_cast<DfinityNode*>(to)->onVote(from, voteFor); }
  };
  struct RandomBeaconExchange : Message {
    const int height;
    explicit RandomBeaconExchange(int h) : height(h) {}
    void action(Network&, Node* from, Node* to) override;
  };
  struct RandomBeaconResult : Message {
    const int height;
    const jlong rd;
    RandomBeaconResult(int h, jlong r) : height(h), rd(r) {}
    void action(Network&, Node*, Node* to) override { static_cast<DfinityNode*>(to)->onRandomBeacon(height, rd); }
  };
  struct SendBlock : Message {
    const DfinityBlock* toSend;
    explicit SendBlock(const DfinityBlock* b) : toSend(b) {}
    void action(Network&, Node*, Node* to) override { static_cast<DfinityNode*>(to)->onBlock(toSend); }
  };

  struct BlockProducerNode : DfinityNode {  // :215-263
    const int myRound;
    int waitForBlockHeight = -1;
    BlockProducerNode(Dfinity& pp, int r) : DfinityNode(pp), myRound(r) {}
    void createProposal(int height) {  // :225-240
      if (head->height != height - 1) throw IllegalArgumentException("createProposal: the head is not the parent");
      const DfinityBlock* nb = p.newBlock(this, height, head, true, p.network_.time);
      std::vector<Node*> attestersS(p.attesters.begin(), p.attesters.end());
      jshuffle(attestersS, p.network_.rd);
      p.network_.send(std::make_shared<BlockProposal>(nb), p.network_.time + p.params.blockConstructionTime, this, attestersS);
      waitForBlockHeight = -1;
    }
    bool onBlock(const DfinityBlock* b) override {  // :243-253
      if (!baseOnBlock(b)) return false;
      if (head->height == waitForBlockHeight) createProposal(waitForBlockHeight + 1);
      return true;
    }
    void onRandomBeaconOnce(int h, jlong rd) override {  // :256-262
      if (rd % p.params.blockProducersRound() == myRound && head->height == h - 1) createProposal(h);
    }
  };
  struct AttesterNode : DfinityNode {  // :265-351
    std::map<jlong, std::set<int>> votes;
    std::vector<const DfinityBlock*> proposals;
    const int myRound;
    int voteForHeight = -1;
    AttesterNode(Dfinity& pp, int r) : DfinityNode(pp), myRound(r) {}
    void voteTo(const DfinityBlock* b) {  // the three identical send sequences :309-313, :341-345
      std::vector<Node*> attestersS(p.attesters.begin(), p.attesters.end());
      jshuffle(attestersS, p.network_.rd);
      p.network_.send(std::make_shared<Vote>(b), p.network_.time + p.params.attestationConstructionTime, this, attestersS);
    }
    void onVote(Node* voter, const DfinityBlock* voteFor) override {  // :277-284
      std::set<int>& voters = votes[voteFor->id];
      if (voteForHeight == voteFor->height)
        if (voters.insert(voter->nodeId).second && (int)voters.size() >= p.params.majority()) sendBlock(voteFor);
    }
    void sendBlock(const DfinityBlock* voteFor) {  // :286-292
      committeeMajorityBlocks.insert(voteFor->id);
      committeeMajorityHeight.insert(voteFor->height);
      voteForHeight = -1;
      p.network_.sendAll(std::make_shared<SendBlock>(voteFor), this);
    }
    void onProposal(const DfinityBlock* b) {  // :298-318
      if (voteForHeight == b->height) {
        std::set<int>& voters = votes[b->id];
        if (voters.insert(nodeId).second) {
          if ((int)voters.size() >= p.params.majority())
            sendBlock(b);
          else
            voteTo(b);
        }
      } else if (b->height > head->height) {
        proposals.push_back(b);
      }
    }
    bool onBlock(const DfinityBlock* b) override {  // :321-332
      if (!baseOnBlock(b)) return false;
      committeeMajorityBlocks.insert(b->id);
      committeeMajorityHeight.insert(b->height);
      if (voteForHeight == b->height) voteForHeight = -1;
      return true;
    }
    void onRandomBeaconOnce(int h, jlong rd) override {  // :335-350
      if (rd % p.params.attestersRound() == myRound && !committeeMajorityHeight.count(h)) {
        voteForHeight = h;
        std::set<const DfinityBlock*> sent;  // (a HashSet there: contains / add only)
        for (const DfinityBlock* b : proposals)
          if (b->height == h && sent.insert(b).second) voteTo(b);
        proposals.clear();
      }
    }
  };
  struct RandomBeaconNode : DfinityNode {  // :353-424
    jlong rd = 0;
    int height = 1, lastRDSent = 0;
    std::map<int, std::set<int>> exchanged;
    explicit RandomBeaconNode(Dfinity& pp) : DfinityNode(pp) {}
    void onRandomBeaconExchange(RandomBeaconNode* from, int h) {  // :367-374
      if (h >= height && h > lastRDSent) {
        std::set<int>& voters = exchanged[h];
        if (voters.insert(from->nodeId).second && h == height && (int)voters.size() >= p.params.majority()) sendRB();
      }
    }
    void sendRB() {  // :376-381
      rd = height;
      lastRDSent = height;
      p.network_.sendAll(std::make_shared<RandomBeaconResult>(height, rd), p.network_.time + p.params.attestationConstructionTime, this);
    }
    bool onBlock(const DfinityBlock* b) override {  // :387-410
      if (!baseOnBlock(b)) return true;
      if (head->height == height) {
        height++;
        std::set<int>& voters = exchanged[height];
        if (voters.insert(nodeId).second && (int)voters.size() >= p.params.majority()) {
          sendRB();
        } else {
          int wt = head->parent->proposalTime + Params::roundTime * 2;
          if (wt <= p.network_.time) wt = p.network_.time + p.params.attestationConstructionTime;
          std::vector<Node*> rdsSends(p.rds.begin(), p.rds.end());
          jshuffle(rdsSends, p.network_.rd);
          p.network_.send(std::make_shared<RandomBeaconExchange>(height), wt, this, rdsSends);
        }
      }
      return false;
    }
    void onRandomBeaconOnce(int h, jlong r) override {  // :417-423
      if (h > height) {
        lastRDSent = height;
        height = h;
        rd = r;
      }
    }
  };

  Params params;
  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  DfinityBlock genesis;  // DfinityBlock.createGenesis() :101-103
  jlong nextBlockId = 1;
  std::vector<std::unique_ptr<DfinityBlock>> blocks;
  std::vector<std::unique_ptr<DfinityNode>> nodes;
  DfinityNode* observer = nullptr;
  std::vector<AttesterNode*> attesters;
  std::vector<BlockProducerNode*> bps;
  std::vector<RandomBeaconNode*> rds;
  explicit Dfinity(const Params& pr) : params(pr) {  // :86-90 (the observer is built — and draws its position — HERE)
    nb = nodeBuilderByName(params.nodeBuilderName);
    // (the reference's constructor leaves the Network's default latency in place; PT/DfinityTest sets the field itself :16)
    if (!params.networkLatencyName.empty()) network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
    nodes.push_back(std::make_unique<DfinityNode>(*this));
    observer = nodes.back().get();
    network_.addNode(observer);  // BlockChainNetwork.addObserver C/BlockChainNetwork.java:15-18
  }
  const DfinityBlock* newBlock(const DfinityNode* producer, int height, const DfinityBlock* parent, bool valid, int time) {
    // Block(producer, height, parent, valid, time) C/Block.java:38-55
    if (height <= 0) throw IllegalArgumentException("Only the genesis block has a special height");
    if (parent && time < parent->proposalTime) throw IllegalArgumentException("bad time");
    if (parent && parent->height >= height) throw IllegalArgumentException("Bad parent");
    auto b = std::make_unique<DfinityBlock>();
    b->producer = producer;
    b->height = height;
    b->id = nextBlockId++;
    b->parent = parent;
    b->valid = valid;
    b->lastTxId = time;
    b->proposalTime = time;
    blocks.push_back(std::move(b));
    return blocks.back().get();
  }
  void init() {  // :426-450
    for (int i = 0; i < params.attestersCount; i++) {
      auto n = std::make_unique<AttesterNode>(*this, i % params.attestersRound());
      attesters.push_back(n.get());
      network_.addNode(n.get());
      nodes.push_back(std::move(n));
    }
    for (int i = 0; i < params.blockProducersCount; i++) {
      auto n = std::make_unique<BlockProducerNode>(*this, i % params.blockProducersRound());
      bps.push_back(n.get());
      network_.addNode(n.get());
      nodes.push_back(std::move(n));
    }
    for (int i = 0; i < params.randomBeaconCount(); i++) {
      auto n = std::make_unique<RandomBeaconNode>(*this);
      rds.push_back(n.get());
      network_.addNode(n.get());
      nodes.push_back(std::move(n));
    }
    jshuffle(bps, network_.rd);
    for (RandomBeaconNode* n : rds) n->sendRB();
  }
};
inline void Dfinity::BlockProposal::action(Network&, Node*, Node* to) { static_cast<AttesterNode*>(to)->onProposal(block); }
inline void Dfinity::RandomBeaconExchange::action(Network&, Node* from, Node* to) {
  static_cast<RandomBeaconNode*>(to)->onRandomBeaconExchange(static_cast<RandomBeaconNode*>(from), height);
}

}  // namespace orc
