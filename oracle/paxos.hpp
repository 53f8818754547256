// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.Paxos (P/Paxos.java:22-525): acceptors and proposers on the plain Network; a proposer sends
// Propose(seq) to the acceptors in a shuffled order (Collections.shuffle with network.rd, one multi-destination send at an
// explicit sendTime, C/Network.java:418-447) and arms a timeout task; acceptors answer Agree / Reject, a majority of Agrees
// makes the proposer send Commit, acceptors answer Accept / RejectOnCommit; a majority of rejections or the timeout starts the
// next proposal with a higher sequence number of the proposer's residue class (:313-338).
// Pinned against PT/PaxosTest.java:9-37 (testSimple: 4 nodes, majority 2, seqIP > 0 after run(10); testCopy: two copies agree)
// and the final check of Paxos.play() (:473-486: every proposer that accepted a value accepted the same one) in
// tests/test_oracle_protocols.py; seed-dependent trajectories are unpinned (no JVM in the image).
#pragma once
#include <optional>
#include "network.hpp"

namespace orc {

class Paxos {
 public:
  static constexpr int MAX_VAL = 1000;  // :24
  struct Params {                       // PaxosParameters :352-371, ctor order
    int acceptorCount = 3, proposerCount = 3, timeout = 1000;
    std::string nodeBuilder, latency;
  };
  struct PaxosNode : Node {  // :147-151
    Paxos& p;
    explicit PaxosNode(Paxos& pp) : Node(pp.network_.rd, *pp.nb), p(pp) {}
    virtual ~PaxosNode() {}
  };
  struct ProposerNode;
  struct AcceptorNode : PaxosNode {  // :153-207
    int maxAgreed = -1;
    std::optional<int> acceptedSeq, acceptedVal;
    ProposerNode* agreedTo = nullptr;
    explicit AcceptorNode(Paxos& pp) : PaxosNode(pp) {}
    void onPropose(PaxosNode* from, int seq);
    void onCommit(PaxosNode* from, int seq, int val);
  };
  struct ProposerNode : PaxosNode {  // :209-339
    const int rank;
    int valueProposed;
    std::optional<int> valueAccepted, acceptedSeqIP, acceptedValIP;
    int seqIP = 0, agreeCountIP = 0, reject1CountIP = 0, acceptCountIP = 0, reject2CountIP = 0;
    bool proposalIP = false;
    int seqAccepted = 0, agreeCount = 0, reject1Count = 0, reject2Count = 0, timeoutCount = 0;
    ProposerNode(Paxos& pp, int rk) : PaxosNode(pp), rank(rk) { valueProposed = pp.network_.rd.nextInt(MAX_VAL); }  // :232-236
    void onReject(int seq, int serverCurSeq) {  // :238-248
      if (seq != seqIP) return;
      reject1CountIP++;
      if (reject1CountIP == p.majority) {
        proposalIP = false;
        seqAccepted = std::max(seqAccepted, serverCurSeq);
        reject1Count++;
        startNextProposal();
      }
    }
    void onAgree(int seq, std::optional<int> acceptedSeq, std::optional<int> acceptedVal);
    void onAccept(int seq) {  // :270-285
      if (seq != seqIP || acceptCountIP >= p.majority) return;
      acceptCountIP++;
      if (acceptCountIP >= p.majority) {
        proposalIP = false;
        if (!acceptedValIP) throw IllegalStateException("onAccept without a value");
        if (valueAccepted) throw IllegalStateException("Already accepted a value");
        valueAccepted = acceptedValIP;
        doneAt = p.network_.time;
      }
    }
    void onRejectOnCommit(int seq, int serverCurSeq) {  // :287-297
      if (seq != seqIP) return;
      reject2CountIP++;
      if (reject2CountIP == p.majority) {
        proposalIP = false;
        seqAccepted = std::max(seqAccepted, serverCurSeq);
        reject2Count++;
        startNextProposal();
      }
    }
    void sendToAcceptors(std::shared_ptr<Message> m, int sentTime) {  // :299-303
      std::vector<Node*> dest(p.acceptors.begin(), p.acceptors.end());
      jshuffle(dest, p.network_.rd);
      p.network_.send(std::move(m), sentTime, this, dest);
    }
    void onTimeout(int seq) {  // :305-311
      if (seq == seqIP && proposalIP) {
        proposalIP = false;
        timeoutCount++;
        startNextProposal();
      }
    }
    void startNextProposal();
  };
  // ---- messages (:43-145)
  struct Propose : Message {
    const int seq;
    explicit Propose(int s) : seq(s) {}
    void action(Network&, Node* from, Node* to) override {
      static_cast<AcceptorNode*>(to)->onPropose(static_cast<PaxosNode*>(from), seq);
    }
  };
  struct Reject : Message {
    const int seqRejected, seqAccepted;
    Reject(int r, int a) : seqRejected(r), seqAccepted(a) {}
    void action(Network&, Node*, Node* to) override { static_cast<ProposerNode*>(to)->onReject(seqRejected, seqAccepted); }
  };
  struct Agree : Message {
    const int yourSeq;
    const std::optional<int> acceptedSeq, acceptedVal;
    Agree(int y, std::optional<int> s, std::optional<int> v) : yourSeq(y), acceptedSeq(s), acceptedVal(v) {}
    void action(Network&, Node*, Node* to) override { static_cast<ProposerNode*>(to)->onAgree(yourSeq, acceptedSeq, acceptedVal); }
  };
  struct Commit : Message {
    const int seq, val;
    Commit(int s, int v) : seq(s), val(v) {}
    void action(Network&, Node* from, Node* to) override {
      static_cast<AcceptorNode*>(to)->onCommit(static_cast<PaxosNode*>(from), seq, val);
    }
  };
  struct Accept : Message {
    const int yourSeq;
    explicit Accept(int y) : yourSeq(y) {}
    void action(Network&, Node*, Node* to) override { static_cast<ProposerNode*>(to)->onAccept(yourSeq); }
  };
  struct RejectOnCommit : Message {
    const int seqRejected, seqAccepted;
    RejectOnCommit(int r, int a) : seqRejected(r), seqAccepted(a) {}
    void action(Network&, Node*, Node* to) override {
      static_cast<ProposerNode*>(to)->onRejectOnCommit(seqRejected, seqAccepted);
    }
  };

  Params params;
  const int majority;
  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<PaxosNode>> nodes;
  std::vector<AcceptorNode*> acceptors;
  std::vector<ProposerNode*> proposers;
  explicit Paxos(const Params& pr) : params(pr), majority(pr.acceptorCount / 2 + 1) {  // :32-37
    nb = nodeBuilderByName(params.nodeBuilder);
    network_.setNetworkLatency(networkLatencyByName(params.latency));
  }
  void init() {  // :374-387
    for (int i = 0; i < params.acceptorCount; i++) {
      auto an = std::make_unique<AcceptorNode>(*this);
      network_.addNode(an.get());
      acceptors.push_back(an.get());
      nodes.push_back(std::move(an));
    }
    for (int i = 0; i < params.proposerCount; i++) {
      auto pn = std::make_unique<ProposerNode>(*this, i);
      ProposerNode* q = pn.get();
      network_.addNode(q);
      proposers.push_back(q);
      nodes.push_back(std::move(pn));
      q->startNextProposal();  // (before the next proposer is built: its draws come between the constructors')
    }
  }
};

inline void Paxos::AcceptorNode::onPropose(PaxosNode* from, int seq) {  // :163-177
  if (seq < maxAgreed) {
    p.network_.send(std::make_shared<Reject>(seq, maxAgreed), this, from);
  } else if (seq == maxAgreed) {
    throw IllegalStateException("a proposal with the sequence number already agreed to");
  } else {
    auto a = std::make_shared<Agree>(seq, acceptedSeq, acceptedVal);
    maxAgreed = seq;
    agreedTo = static_cast<ProposerNode*>(from);
    p.network_.send(a, this, from);
  }
}
inline void Paxos::AcceptorNode::onCommit(PaxosNode* from, int seq, int val) {  // :179-190
  if (seq != maxAgreed || (acceptedVal && *acceptedVal != val)) {
    p.network_.send(std::make_shared<RejectOnCommit>(seq, maxAgreed), this, from);
  } else {
    acceptedVal = val;
    acceptedSeq = acceptedSeq ? std::max(*acceptedSeq, seq) : seq;
    p.network_.send(std::make_shared<Accept>(seq), this, from);
  }
}
inline void Paxos::ProposerNode::onAgree(int seq, std::optional<int> acceptedSeq, std::optional<int> acceptedVal) {  // :250-268
  if (seq != seqIP || agreeCountIP >= p.majority) return;
  agreeCountIP++;
  if (acceptedSeq && (!acceptedSeqIP || *acceptedSeqIP < *acceptedSeq)) {
    acceptedSeqIP = acceptedSeq;
    acceptedValIP = acceptedVal;
  }
  if (agreeCountIP >= p.majority) {
    agreeCount++;
    if (!acceptedValIP) acceptedValIP = valueProposed;
    sendToAcceptors(std::make_shared<Commit>(seqIP, *acceptedValIP), p.network_.time + 1);
  }
}
inline void Paxos::ProposerNode::startNextProposal() {  // :313-338
  if (proposalIP) throw IllegalStateException("a proposal is in progress");
  acceptedSeqIP.reset();
  acceptedValIP.reset();
  proposalIP = true;
  agreeCountIP = reject1CountIP = acceptCountIP = reject2CountIP = 0;
  const int gap = seqAccepted % p.params.proposerCount;
  const int newSeqIP = seqAccepted + p.params.proposerCount - gap + rank;
  seqIP = newSeqIP > seqIP ? newSeqIP : seqIP + p.params.proposerCount;
  const int seq = seqIP;
  const int sentTime = p.network_.time + 1;
  sendToAcceptors(std::make_shared<Propose>(seq), sentTime);
  p.network_.registerTask([this, seq] { onTimeout(seq); }, sentTime + p.params.timeout, this);
}

}  // namespace orc
