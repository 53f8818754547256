// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header).
// Restatement of protocols.Slush (P/Slush.java:11-296) and protocols.Snowflake (P/Snowflake.java:11-312), the two
// sampling protocols of the Avalanche family in the reference: a coloured node queries K random remotes (one
// multi-destination send, C/Network.java:353-362,418-447), every remote answers with its colour (adopting the query's when it
// has none and starting to query itself), and once the K answers of a query are in the node flips when more than alpha * K
// of them are of the other colour. Slush stops after M rounds per node (:172-175); Snowflake counts consecutive
// confirmations of its own colour and stops past beta of them (P/Snowflake.java:175-188).
// Pinned against PT/SlushTest.java:14-45 and PT/SnowflakeTest.java:14-47 (testSimple: every node ends on node 0's colour after
// run(10); testCopy: two copies agree) in tests/test_oracle_protocols.py; seed-dependent trajectories are unpinned (no JVM
// in the image).
#pragma once
#include <algorithm>
#include <map>
#include "network.hpp"

namespace orc {

class Slush {  // P/Slush.java; snowflake == true: P/Snowflake.java (the two files differ in onAnswer and one counter)
 public:
  struct Params {  // SlushParameters :17-52 / SnowflakeParameters :18-58, ctor order (B: Snowflake only)
    int NODES_AV = 100, M = 4, K = 7;
    double A = 4;
    int B = 7;
    std::string nodeBuilderName, networkLatencyName;
    double AK() const { return K * A; }
  };
  struct SlushNode;
  struct Query : Message {  // :86-99
    const int id, color;
    Query(int i, int c) : id(i), color(c) {}
    void action(Network&, Node* from, Node* to) override;
  };
  struct AnswerQuery : Message {  // :101-114
    const std::shared_ptr<Query> originalQuery;
    const int color;
    AnswerQuery(std::shared_ptr<Query> q, int c) : originalQuery(std::move(q)), color(c) {}
    void action(Network&, Node* from, Node* to) override;
  };
  struct Answer {  // :205-221
    int round = 0;
    int colorsFound[3] = {0, 0, 0};
    int answerCount() const { return colorsFound[0] + colorsFound[1] + colorsFound[2]; }
  };
  struct SlushNode : Node {  // :116-203
    Slush& p;
    int myColor = 0, myQueryNonce = 0;
    int round = 0;  // Slush :119
    int cnt = 0;    // Snowflake :128
    std::map<int, Answer> answerIP;  // (a HashMap in the reference: get / put / remove only, never iterated)
    explicit SlushNode(Slush& pp) : Node(pp.network_.rd, *pp.nb), p(pp) {}
    std::vector<Node*> randomRemotes() {  // :126-137
      std::vector<Node*> res;
      while ((int)res.size() != p.params.K) {
        int r = p.network_.rd.nextInt(p.params.NODES_AV);
        Node* cand = p.network_.getNodeById(r);
        if (r != nodeId && std::find(res.begin(), res.end(), cand) == res.end()) res.push_back(cand);
      }
      return res;
    }
    int otherColor() const { return myColor == 1 ? 2 : 1; }  // :139-141
    void onQuery(const std::shared_ptr<Query>& qa, SlushNode* from) {  // :148-154
      if (myColor == 0) {
        myColor = qa->color;
        sendQuery(1);
      }
      p.network_.send(std::make_shared<AnswerQuery>(qa, myColor), this, from);
    }
    void onAnswer(int queryId, int color) {  // Slush :161-176, Snowflake :170-189
      auto it = answerIP.find(queryId);
      if (it == answerIP.end()) throw IllegalStateException("NullPointerException: answerIP.get(queryId)");
      Answer& asw = it->second;
      asw.colorsFound[color]++;
      if (asw.answerCount() != p.params.K) return;
      const Answer done = asw;
      answerIP.erase(it);
      if (!p.snowflake) {
        if (done.colorsFound[otherColor()] > p.params.AK()) myColor = otherColor();
        if (round < p.params.M) {
          round++;
          sendQuery(done.round + 1);
        }
      } else {
        if (done.colorsFound[otherColor()] > p.params.AK()) {
          myColor = otherColor();
          cnt = 0;
        } else if (done.colorsFound[myColor] > p.params.AK()) {
          cnt++;
        }
        if (cnt <= p.params.B) sendQuery(done.round + 1);
      }
    }
    void sendQuery(int countInM) {  // :178-182 (Snowflake :190-194)
      auto q = std::make_shared<Query>(++myQueryNonce, myColor);
      Answer a;
      a.round = countInM;
      answerIP[q->id] = a;
      p.network_.send(q, this, randomRemotes());
    }
  };

  Params params;
  const bool snowflake;
  Network network_;
  std::unique_ptr<NodeBuilder> nb;
  std::vector<std::unique_ptr<SlushNode>> nodes;
  Slush(const Params& pr, bool snow) : params(pr), snowflake(snow) {  // :54-60
    nb = nodeBuilderByName(params.nodeBuilderName);
    network_.setNetworkLatency(networkLatencyByName(params.networkLatencyName));
  }
  void init() {  // :63-75
    for (int i = 0; i < params.NODES_AV; i++) {
      nodes.push_back(std::make_unique<SlushNode>(*this));
      network_.addNode(nodes.back().get());
    }
    SlushNode* uncolored1 = nodes.at(0).get();
    SlushNode* uncolored2 = nodes.at(1).get();
    uncolored1->myColor = 1;
    uncolored1->sendQuery(1);
    uncolored2->myColor = 2;
    uncolored2->sendQuery(1);
  }
};
inline void Slush::Query::action(Network&, Node* from, Node* to) {
  // (the envelope holds the message: the answer keeps the query it answers, as AnswerQuery.originalQuery does)
  auto self = std::make_shared<Query>(id, color);
  static_cast<SlushNode*>(to)->onQuery(self, static_cast<SlushNode*>(from));
}
inline void Slush::AnswerQuery::action(Network&, Node*, Node* to) {
  static_cast<SlushNode*>(to)->onAnswer(originalQuery->id, color);
}

}  // namespace orc
