// ORACLE — TEST INFRASTRUCTURE ONLY. Pins the oracle against every value the reference's own
// engine unit tests fix:
//   CT/NetworkTest.java:34-506, CT/EnvelopeStorageTest.java:34-129, CT/NetworkLatencyTest.java:56-79,
//   PT/PingPongTest.java:8-37, PT/HandelTest.java:14-49
// (CT/ = core/src/test/java/net/consensys/wittgenstein/core/, PT/ = protocols/src/test/...).
// Prints one "ok <name>" / "FAIL <name>" line per restated test; exit code = number of failures.
#include <cstdio>
#include "handel.hpp"
#include "pingpong.hpp"

using namespace orc;

static int g_fail = 0;
static const char* g_cur = "";
#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      printf("FAIL %s: %s (line %d)\n", g_cur, #c, __LINE__);           \
      g_fail++;                                                         \
      return;                                                           \
    }                                                                   \
  } while (0)
#define CHECK_EQ(a, b)                                                                        \
  do {                                                                                        \
    long long _a = (long long)(a), _b = (long long)(b);                                       \
    if (_a != _b) {                                                                           \
      printf("FAIL %s: %s == %lld, expected %lld (line %d)\n", g_cur, #b, _b, _a, __LINE__);  \
      g_fail++;                                                                               \
      return;                                                                                 \
    }                                                                                         \
  } while (0)

struct Fn : Message {
  std::function<void(Network&, Node*, Node*)> f;
  explicit Fn(std::function<void(Network&, Node*, Node*)> ff) : f(std::move(ff)) {}
  void action(Network& n, Node* a, Node* b) override { f(n, a, b); }
};
static std::shared_ptr<Message> dummy() {
  return std::make_shared<Fn>([](Network&, Node*, Node*) {});
}

// fixture of CT/NetworkTest.java:12-33 (NoLatency, 4 nodes at (1,1))
struct Fx {
  Network network;
  NodeBuilder nb;
  Node n0{network.rd, nb}, n1{network.rd, nb}, n2{network.rd, nb}, n3{network.rd, nb};
  std::shared_ptr<Message> m = dummy();
  Fx() {
    network.setNetworkLatency(std::make_unique<NetworkNoLatency>());
    network.addNode(&n0);
    network.addNode(&n1);
    network.addNode(&n2);
    network.addNode(&n3);
  }
};

static void testSimpleMessage() {  // CT/NetworkTest.java:34-55
  Fx f;
  int a1 = -1, a2 = -1;
  auto act = std::make_shared<Fn>([&](Network&, Node* from, Node* to) {
    a1 = from->nodeId;
    a2 = to->nodeId;
  });
  f.network.send(act, 1, &f.n1, &f.n2);
  CHECK_EQ(1, f.network.msgs.size());
  CHECK_EQ(-1, a1);
  f.network.run(5);
  CHECK_EQ(1, a1);
  CHECK_EQ(2, a2);
}
static void testRegisterTask() {  // :57-68
  Fx f;
  bool ab = false;
  f.network.registerTask([&] { ab = true; }, 100, &f.n0);
  f.network.runMs(99);
  CHECK(!ab);
  f.network.runMs(1);
  CHECK(ab);
  CHECK_EQ(0, f.network.msgs.size());
}
static void testAllFavorsOfSend() {  // :70-97
  Fx f;
  int a1 = 0, a2 = 0;
  auto act = std::make_shared<Fn>([&](Network&, Node* from, Node* to) {
    a1 += from->nodeId;
    a2 += to->nodeId;
  });
  std::vector<Node*> dests{&f.n2, &f.n3};
  f.network.send(act, &f.n1, &f.n2);
  f.network.send(act, 1, &f.n1, &f.n2);
  f.network.send(act, 1, &f.n1, dests);
  f.network.send(act, &f.n1, dests);
  CHECK_EQ(4, f.network.msgs.size());
  f.network.run(1);
  CHECK_EQ(0, f.network.msgs.size());
  CHECK_EQ(6, a1);
  CHECK_EQ(14, a2);
}
static void testMultipleMessage() {  // :99-117
  Fx f;
  int ab = 0;
  auto act = std::make_shared<Fn>([&](Network&, Node*, Node*) { ab++; });
  f.network.send(act, 1, &f.n0, std::vector<Node*>{&f.n1, &f.n2, &f.n3});
  f.network.runMs(2);
  CHECK_EQ(3, ab);
  CHECK_EQ(0, f.network.msgs.size());
}
static void testMultipleMessageWithDelays() {  // :119-144
  Fx f;
  int ab = 0;
  auto act = std::make_shared<Fn>([&](Network&, Node*, Node*) { ab++; });
  f.network.send(act, 1, &f.n0, std::vector<Node*>{&f.n1, &f.n2, &f.n3}, 10);
  f.network.runMs(2);
  CHECK_EQ(1, ab);
  f.network.runMs(11);
  CHECK_EQ(2, ab);
  f.network.runMs(11);
  CHECK_EQ(3, ab);
  CHECK_EQ(0, f.network.msgs.size());
}
static void testMultipleMessageWithDelaysAcrossSlots() {  // :146-163
  Fx f;
  int ab = 0;
  auto act = std::make_shared<Fn>([&](Network&, Node*, Node*) { ab++; });
  f.network.send(act, 59000, &f.n0, std::vector<Node*>{&f.n1, &f.n2, &f.n3}, 55000);
  f.network.runMs(200000);
  CHECK_EQ(0, f.network.msgs.size());
  CHECK_EQ(3, ab);
}
static void testMultipleMessageWithDelaysEndOfSlot() {  // :165-185
  Fx f;
  int ab = 0;
  auto act = std::make_shared<Fn>([&](Network&, Node*, Node*) { ab++; });
  f.network.send(act, 58998, &f.n0, std::vector<Node*>{&f.n1, &f.n2, &f.n3}, 1000);
  CHECK_EQ(1, f.network.msgs.size());
  f.network.runMs(59000);
  CHECK_EQ(1, f.network.msgs.size());
  f.network.runMs(3000);
  CHECK_EQ(0, f.network.msgs.size());
  CHECK_EQ(3, ab);
}
static void testMsgArrival() {  // :187-207
  Fx f;
  auto mas = f.network.createMessageArrivals(*f.m, 1, &f.n0, {&f.n1, &f.n2, &f.n3}, 1, 10);
  CHECK_EQ(3, mas.size());
  CHECK_EQ(2, mas[0].arrival);
  CHECK_EQ(13, mas[1].arrival);
  CHECK_EQ(24, mas[2].arrival);
  MultipleDestWithDelayEnvelope e(f.m, f.n0, mas, 1);
  CHECK_EQ(2, e.nextArrivalTime(f.network));
  e.markRead();
  CHECK_EQ(13, e.nextArrivalTime(f.network));
  e.markRead();
  CHECK_EQ(24, e.nextArrivalTime(f.network));
  CHECK(e.hasNextReader());
  e.markRead();
  CHECK(!e.hasNextReader());
}
static void testMsgArrivalWithRandomNoDelay() {  // :209-239
  Network network;
  NodeBuilderWithRandomPosition nb;
  Node n0(network.rd, nb), n1(network.rd, nb), n2(network.rd, nb), n3(network.rd, nb);
  network.setNetworkLatency(std::make_unique<NetworkLatencyByDistanceWJitter>());
  for (Node* n : {&n0, &n1, &n2, &n3}) network.addNode(n);
  auto m = dummy();
  auto mas = network.createMessageArrivals(*m, 1, &n0, {&n1, &n2, &n3}, 2, 0);
  CHECK_EQ(3, mas.size());
  MultipleDestEnvelope e(m, n0, mas, 1, 2);
  CHECK_EQ(2, e.randomSeed);
  for (int i = 0; i < 3; i++) {
    CHECK_EQ(mas[i].arrival, e.nextArrivalTime(network));
    if (i == 2) CHECK(e.hasNextReader());
    e.markRead();
  }
  CHECK(!e.hasNextReader());
}
static void testMsgArrivalWithRandom() {  // :241-271
  Network network;
  NodeBuilderWithRandomPosition nb;
  Node n0(network.rd, nb), n1(network.rd, nb), n2(network.rd, nb), n3(network.rd, nb);
  network.setNetworkLatency(std::make_unique<NetworkLatencyByDistanceWJitter>());
  for (Node* n : {&n0, &n1, &n2, &n3}) network.addNode(n);
  auto m = dummy();
  auto mas = network.createMessageArrivals(*m, 1, &n0, {&n1, &n2, &n3}, 1, 20);
  CHECK_EQ(3, mas.size());
  MultipleDestWithDelayEnvelope e(m, n0, mas, 1);
  for (int i = 0; i < 3; i++) {
    CHECK_EQ(mas[i].arrival, e.nextArrivalTime(network));
    e.markRead();
  }
  CHECK(!e.hasNextReader());
}
static void testStats() {  // :273-298
  Fx f;
  f.network.send(f.m, &f.n0, std::vector<Node*>{&f.n1, &f.n2, &f.n3});
  f.network.send(f.m, &f.n0, &f.n1);
  f.network.runMs(2);
  CHECK_EQ(0, f.n0.msgReceived);
  CHECK_EQ(0, f.n0.bytesReceived);
  CHECK_EQ(4, f.n0.msgSent);
  CHECK_EQ(4, f.n0.bytesSent);
  CHECK_EQ(2, f.n1.msgReceived);
  CHECK_EQ(2, f.n1.bytesReceived);
  CHECK_EQ(0, f.n1.msgSent);
  CHECK_EQ(1, f.n2.msgReceived);
  CHECK_EQ(1, f.n2.bytesReceived);
  CHECK_EQ(1, f.n3.msgReceived);
  CHECK_EQ(1, f.n3.bytesReceived);
  CHECK_EQ(0, f.n3.bytesSent);
}
static void testSortedArrivals() {  // :300-327
  Fx f;
  f.network.send(f.m, 1, &f.n0, std::vector<Node*>{&f.n1, &f.n2, &f.n3});
  Envelope* m = f.network.msgs.peekFirst();
  CHECK(m != nullptr);
  int seen = 0;
  int l = m->nextArrivalTime(f.network);
  for (int i = 0; i < 3; i++) {
    CHECK(m->nextArrivalTime(f.network) >= l);
    l = m->nextArrivalTime(f.network);
    int d = m->getNextDestId();
    CHECK(d >= 1 && d <= 3 && !(seen & (1 << d)));
    seen |= 1 << d;
    m->markRead();
    CHECK(m->hasNextReader() == (i < 2));
  }
}
static void testDelays() {  // :329-346 (EthScan)
  Fx f;
  f.network.setNetworkLatency(std::make_unique<EthScanNetworkLatency>());
  f.network.send(f.m, 1, &f.n0, std::vector<Node*>{&f.n1, &f.n2, &f.n3});
  Envelope* e = f.network.msgs.pollFirst();
  CHECK(e != nullptr);
  auto* mm = dynamic_cast<MultipleDestEnvelope*>(e);
  CHECK(mm != nullptr);
  auto mas = f.network.createMessageArrivals(*f.m, 1, &f.n0, {&f.n1, &f.n2, &f.n3}, mm->randomSeed, 0);
  for (auto& ma : mas) {
    CHECK_EQ(ma.arrival, e->nextArrivalTime(f.network));
    e->markRead();
  }
  delete e;
}
static void testPartition() {  // :348-422
  Network outer;  // the reference draws node randomness from the *field* network's rd
  Network net;
  struct NB : NodeBuilder {
    int ai = 0;
    int getX(jint) override { return ai += Node::MAX_X / 10; }
  } nb;
  Node n0(outer.rd, nb), n1(outer.rd, nb), n2(outer.rd, nb), n3(outer.rd, nb);
  for (Node* n : {&n0, &n1, &n2, &n3}) net.addNode(n);
  int ab = 0;
  auto act = std::make_shared<Fn>([&](Network&, Node*, Node*) { ab++; });
  net.partition(0.25f);
  int bound = (int)(0.25f * Node::MAX_X);
  CHECK(std::find(net.partitionsInX.begin(), net.partitionsInX.end(), bound) != net.partitionsInX.end());
  CHECK_EQ(0, net.partitionId(n0));
  CHECK_EQ(0, net.partitionId(n1));
  CHECK_EQ(1, net.partitionId(n2));
  CHECK_EQ(1, net.partitionId(n3));
  net.send(act, &n0, &n1);
  CHECK(net.msgs.peekFirst() != nullptr);
  net.msgs.clearAndCleanup();
  net.send(act, &n1, &n2);
  CHECK(net.msgs.peekFirst() == nullptr);
  net.send(act, &n2, &n3);
  CHECK(net.msgs.peekFirst() != nullptr);
  net.msgs.clearAndCleanup();
  net.partition(0.35f);
  CHECK_EQ(0, net.partitionId(n0));
  CHECK_EQ(0, net.partitionId(n1));
  CHECK_EQ(1, net.partitionId(n2));
  CHECK_EQ(2, net.partitionId(n3));
  net.send(act, &n0, &n1);
  CHECK(net.msgs.peekFirst() != nullptr);
  net.msgs.clearAndCleanup();
  net.send(act, &n1, &n2);
  CHECK(net.msgs.peekFirst() == nullptr);
  net.send(act, &n2, &n3);
  CHECK(net.msgs.peekFirst() == nullptr);
  net.send(act, &n3, &n0);
  CHECK(net.msgs.peekFirst() == nullptr);
  bool threw = false;
  try {
    net.partition(0.35f);
  } catch (const IllegalArgumentException&) {
    threw = true;
  }
  CHECK(threw);
}
static void testLongRunning() {  // :424-434
  Fx f;
  auto act = dummy();
  while (f.network.time < 100000000) {
    f.network.runMs(10000);
    f.network.send(act, &f.n0, &f.n1);
  }
  CHECK(f.network.time >= 100000000);
}
static void testTask() {  // :436-450
  Fx f;
  int ai = 0;
  f.network.registerTask([&] { ai++; }, 1000, &f.n0);
  f.network.runMs(500);
  CHECK_EQ(0, ai);
  f.network.runMs(500);
  CHECK_EQ(1, ai);
  f.network.runMs(100);
  CHECK_EQ(1, ai);
  f.network.runMs(5000);
  CHECK_EQ(1, ai);
}
static void testTaskOnStoppedNode() {  // :452-460
  Fx f;
  int ai = 0;
  f.network.registerTask([&] { ai++; }, 1000, &f.n0);
  f.n0.stop();
  f.network.runMs(5000);
  CHECK_EQ(0, ai);
}
static void testPeriodicTask() {  // :462-479
  Fx f;
  int ai = 0;
  f.network.registerPeriodicTask([&] { ai++; }, 1000, 100, &f.n0);
  f.network.runMs(500);
  CHECK_EQ(0, ai);
  f.network.runMs(500);
  CHECK_EQ(1, ai);
  f.network.runMs(100);
  CHECK_EQ(2, ai);
  f.network.runMs(50);
  CHECK_EQ(2, ai);
  f.n0.stop();
  f.network.runMs(1000);
  CHECK_EQ(2, ai);
}
static void testConditionalTask() {  // :481-506
  Fx f;
  bool ab = false;
  int ai = 0;
  f.network.registerConditionalTask([&] { ai++; }, 1000, 100, &f.n0, [&] { return ab; }, [] { return true; });
  f.network.runMs(500);
  CHECK_EQ(0, ai);
  f.network.runMs(500);
  CHECK_EQ(0, ai);
  ab = true;
  f.network.runMs(1);
  CHECK_EQ(1, ai);
  f.network.runMs(99);
  CHECK_EQ(1, ai);
  f.network.runMs(1);
  CHECK_EQ(2, ai);
  f.n0.stop();
  f.network.runMs(1000);
  CHECK_EQ(2, ai);
}
static void testRunMsArg() {  // C/Network.java:319-321
  Fx f;
  bool threw = false;
  try {
    f.network.runMs(0);
  } catch (const IllegalArgumentException&) {
    threw = true;
  }
  CHECK(threw);
}

// ---- CT/EnvelopeStorageTest.java
struct Fs {
  Network network;
  NodeBuilder nb;
  JRandom rd{0};
  Node n0{rd, nb}, n1{rd, nb}, n2{rd, nb}, n3{rd, nb};
  Fs() {
    for (Node* n : {&n0, &n1, &n2, &n3}) network.addNode(n);
  }
};
static void storageWorkflow() {  // :34-60
  Fs f;
  auto* m1 = new SingleDestEnvelope(dummy(), f.n0, f.n1, 1, 1);
  auto* m2 = new SingleDestEnvelope(dummy(), f.n0, f.n1, 1, 1);
  f.network.msgs.addMsg(m1);
  f.network.msgs.addMsg(m2);
  CHECK(f.network.msgs.peek(2) == nullptr);
  CHECK(f.network.msgs.peek(1) == m2);
  CHECK(f.network.msgs.poll(1) == m2);
  CHECK(f.network.msgs.poll(1) == m1);
  CHECK(f.network.msgs.peek(1) == nullptr);
  delete m1;
  delete m2;
  auto* m3 = new SingleDestEnvelope(dummy(), f.n0, f.n1, 1, Network::duration + 1);
  f.network.msgs.addMsg(m3);
  CHECK_EQ(2, f.network.msgs.msgsBySlot.size());
  f.network.time = Network::duration + 1;
  auto* m4 = new SingleDestEnvelope(dummy(), f.n0, f.n1, 1, Network::duration + 1);
  f.network.msgs.addMsg(m4);
  CHECK_EQ(1, f.network.msgs.msgsBySlot.size());
  f.network.msgs.clearAndCleanup();
  f.network.run(1);
}
static void storageAction() {  // :62-83
  Fs f;
  bool ab = false;
  auto act = std::make_shared<Fn>([&](Network&, Node*, Node*) { ab = true; });
  f.network.msgs.addMsg(new SingleDestEnvelope(act, f.n0, f.n1, 1, 7 * 1000 + 1));
  f.network.run(7);
  CHECK(!ab);
  f.network.run(1);
  CHECK(ab);
  ab = false;
  f.network.msgs.addMsg(new SingleDestEnvelope(act, f.n0, f.n1, 1, 8 * 1000));
  f.network.run(1);
  CHECK(ab);
}
static void storageMsgArrival() {  // :85-103
  Fs f;
  long ab = 0;
  auto act = std::make_shared<Fn>([&](Network& n, Node*, Node*) { ab = n.time; });
  f.network.msgs.addMsg(new SingleDestEnvelope(act, f.n0, f.n1, 1, 5));
  f.network.run(1);
  CHECK_EQ(5, ab);
  CHECK_EQ(0, f.network.msgs.size());
}
static void storageEdgeCases() {  // :105-129
  {
    Fs f;
    CHECK(f.network.msgs.peek(0) == nullptr);
    CHECK(f.network.msgs.peek(10 * 60 * 1000 + 1) == nullptr);
    f.network.msgs.addMsg(new SingleDestEnvelope(dummy(), f.n0, f.n1, 1, 10 * 60 * 1000 + 1));
    CHECK(f.network.msgs.peek(10 * 60 * 1000 + 1) != nullptr);
  }
  {
    Fs f;
    CHECK(f.network.msgs.peek(Network::duration) == nullptr);
    f.network.msgs.addMsg(new SingleDestEnvelope(dummy(), f.n0, f.n1, 1, Network::duration));
    CHECK(f.network.msgs.peek(Network::duration) != nullptr);
    CHECK_EQ(2, f.network.msgs.msgsBySlot.size());
  }
}

// ---- CT/NetworkLatencyTest.java:56-79
static void ic3Latency() {
  IC3NetworkLatency nl;
  NodeBuilder nb0, nb00;
  JRandom r0(0), r00(0), r1(0);
  Node a0(r0, nb0), a00(r00, nb00);
  CHECK_EQ(IC3NetworkLatency::S10 / 2, nl.getLatency(a0, a00, 0));
  struct NB : NodeBuilder {
    int getX(jint) override { return Node::MAX_X / 2; }
    int getY(jint) override { return Node::MAX_Y / 2; }
  } nb;
  Node a1(r1, nb);
  CHECK_EQ(IC3NetworkLatency::SW / 2, nl.getLatency(a0, a1, 0));
  CHECK_EQ(IC3NetworkLatency::SW / 2, nl.getLatency(a1, a0, 0));
  CHECK_EQ(46, IC3NetworkLatency::S10 / 2);
  CHECK_EQ(175, IC3NetworkLatency::SW / 2);
}

// ---- PT/PingPongTest.java
static void pingPongSimple() {  // :8-19
  PingPong p{PingPong::PingPongParameters()};
  p.init();
  p.network().run(10);
  CHECK_EQ(1000, p.network().allNodes.size());
  for (auto& n : p.nodes) {
    CHECK(!n->isDown());
    CHECK(n->pong == 0 || n->pong == 1000);
  }
  CHECK_EQ(1000, p.nodes[0]->pong);
}
static void pingPongCopy() {  // :22-37
  PingPong p1{PingPong::PingPongParameters()}, p2{PingPong::PingPongParameters()};
  p1.init();
  p1.network().runMs(200);
  p2.init();
  p2.network().runMs(200);
  for (size_t i = 0; i < p1.nodes.size(); i++) CHECK_EQ(p1.nodes[i]->pong, p2.nodes[i]->pong);
}

// ---- PT/HandelTest.java
static Handel::HandelParameters handelTestParams() {
  return Handel::HandelParameters(64, 60, 6, 10, 5, 5, 10, 2, "RANDOM_SPEED=CONSTANT_TOR=0.00",
                                  "NetworkLatencyByDistanceWJitter", 100, false, false, nullptr);
}
static void handelCopy() {  // :14-34
  Handel p1(handelTestParams()), p2(handelTestParams());
  p1.init();
  p2.init();
  while (p1.network().time < 2000) {
    p1.network().runMs(1);
    p2.network().runMs(1);
    CHECK_EQ(p1.network().msgs.size(), p2.network().msgs.size());
    for (int i = 0; i < 64; i++) {
      CHECK_EQ(p1.node(i)->doneAt, p2.node(i)->doneAt);
      CHECK_EQ(p1.node(i)->totalSigSize(), p2.node(i)->totalSigSize());
    }
  }
}
static bool contUntilDone(Handel& p) {  // C/RunMultipleTimes.java:88-97
  for (auto& n : p.nodes)
    if (!n->isDown() && n->doneAt == 0) return true;
  return false;
}
static void handelRun() {  // :36-49
  Handel p1(handelTestParams());
  p1.init();
  while (contUntilDone(p1) && p1.network().time < 20000) p1.network().runMs(1000);
  CHECK(!contUntilDone(p1));
}

#define RUN(t)            \
  do {                    \
    g_cur = #t;           \
    int before = g_fail;  \
    try {                 \
      t();                \
    } catch (const std::exception& e) { \
      printf("FAIL %s: exception %s\n", #t, e.what()); \
      g_fail++;           \
    }                     \
    if (g_fail == before) printf("ok %s\n", #t); \
  } while (0)

int main() {
  RUN(testSimpleMessage);
  RUN(testRegisterTask);
  RUN(testAllFavorsOfSend);
  RUN(testMultipleMessage);
  RUN(testMultipleMessageWithDelays);
  RUN(testMultipleMessageWithDelaysAcrossSlots);
  RUN(testMultipleMessageWithDelaysEndOfSlot);
  RUN(testMsgArrival);
  RUN(testMsgArrivalWithRandomNoDelay);
  RUN(testMsgArrivalWithRandom);
  RUN(testStats);
  RUN(testSortedArrivals);
  RUN(testDelays);
  RUN(testPartition);
  RUN(testLongRunning);
  RUN(testTask);
  RUN(testTaskOnStoppedNode);
  RUN(testPeriodicTask);
  RUN(testConditionalTask);
  RUN(testRunMsArg);
  RUN(storageWorkflow);
  RUN(storageAction);
  RUN(storageMsgArrival);
  RUN(storageEdgeCases);
  RUN(ic3Latency);
  RUN(pingPongSimple);
  RUN(pingPongCopy);
  RUN(handelCopy);
  RUN(handelRun);
  printf("%d failure(s)\n", g_fail);
  return g_fail;
}
