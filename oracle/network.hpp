// ORACLE — TEST INFRASTRUCTURE ONLY (see jdk.hpp header). CPU restatement, event for event, of
// the reference scheduler:
//   core.Network          C/Network.java:14-708
//   core.Envelope         C/Envelope.java:11-302
//   core.NetworkLatency   C/NetworkLatency.java:12-34,49-73,235-313,366-417
//   core.Node             C/Node.java:13-291
//   core.NodeBuilder      C/NodeBuilder.java:19-96
//   core.messages.*       C/messages/{Message,Task,PeriodicTask,ConditionalTask}.java
//   GeneralizedParetoDistribution  C/utils/GeneralizedParetoDistribution.java:26-46
// (C/ = core/src/main/java/net/consensys/wittgenstein/core/). Single-threaded like the reference
// (C/Network.java:7-11). PARITY UNPINNED vs a JVM run (none available); pinned against every
// value the reference's own unit tests fix (oracle/test_network.cpp).
#pragma once
#include <cmath>
#include <functional>
#include <list>
#include <map>
#include <memory>
#include "jdk.hpp"
#include "geo.hpp"

namespace orc {

class Node;
class Network;

// ---------------------------------------------------------------- GeneralizedParetoDistribution
struct GeneralizedParetoDistribution {  // C/utils/GeneralizedParetoDistribution.java
  double shape, location, scale;
  GeneralizedParetoDistribution(double sh, double lo, double sc) : shape(sh), location(lo), scale(sc) {
    if (scale <= 0.0) throw IllegalArgumentException("scale");
  }
  double inverseF(double y) const {  // :26-46
    const double ONE = 0.999999, ZERO = 0.000001;
    if (y < 0.0 || y > 1.0) throw IllegalArgumentException("y");
    if (y < ZERO) return location;
    if (y > ONE && shape >= 0) return INFINITY;
    if (y > ONE && shape < 0) return location - scale / shape;
    if (std::fabs(shape) < ZERO) return location - scale * std::log1p(-y);
    return location + scale / shape * (-1 + std::pow(1 - y, -shape));
  }
};

// ---------------------------------------------------------------- NodeBuilder
struct NodeBuilder {  // C/NodeBuilder.java:19-75
  int nodeIds = 0;
  virtual std::string getCityName(jint) { return "world"; }  // Node.DEFAULT_CITY (C/Node.java:20, C/NodeBuilder.java:64-66)
  // aspects (C/Node.java:145-243): only the two the RANDOM registry entries can add
  bool speedUniform = false;  // SpeedRatioAspect(UniformSpeed)  (C/RegistryNodeBuilders.java:59-61)
  double torRatio = 0.0;      // ExtraLatencyAspect(tor) when tor > 0.001 (:62-64)
  virtual ~NodeBuilder() {}
  int allocateNodeId() { return nodeIds++; }
  virtual int getX(jint) { return 1; }
  virtual int getY(jint) { return 1; }
};

struct NodeBuilderWithRandomPosition : NodeBuilder {  // C/NodeBuilder.java:77-96
  int getX(jint rdInt) override {
    jlong r = (jlong)(rdInt >> 16);  // int shift, then widened
    if (r < 0) r = -r;
    return (int)(r % 2000 + 1);
  }
  int getY(jint rdInt) override {
    jlong r = (jlong)(jint)((uint32_t)rdInt << 16);  // int shift (wraps), then widened
    if (r < 0) r = -r;
    return (int)(r % 1112 + 1);
  }
};

struct NodeBuilderWithCity : NodeBuilder {  // C/NodeBuilder.java:98-147 (oracle/geo.hpp)
  CityChooser chooser;
  NodeBuilderWithCity(const std::vector<std::string>& cities, const JHashMap<CityInfo>& geo) : chooser(cities, geo) {}
  int city(jint rdInt) const {
    const int c = chooser.choose(rdInt);
    if (c < 0) throw IllegalStateException("NullPointerException: no city for this draw (C/NodeBuilder.java:121-124)");
    return c;
  }
  std::string getCityName(jint rdInt) override { return chooser.name[city(rdInt)]; }
  int getX(jint rdInt) override { return chooser.info[city(rdInt)].mercX; }
  int getY(jint rdInt) override { return chooser.info[city(rdInt)].mercY; }
};

inline std::vector<std::string> awsCitiesSorted() {  // AwsRegionNetworkLatency.cities() (C/NetworkLatency.java:104-109)
  std::vector<std::string> c = awsRegions();
  std::sort(c.begin(), c.end());
  return c;
}

// RegistryNodeBuilders.getByName (C/RegistryNodeBuilders.java:28-81).
// Names look like "RANDOM_SPEED=CONSTANT_TOR=0.00" (also AWS_..., CITIES_...). Null/empty -> RANDOM, constant, tor 0.
inline std::unique_ptr<NodeBuilder> nodeBuilderByName(const std::string& name) {
  std::unique_ptr<NodeBuilder> nb;
  if (name.rfind("AWS_SPEED=", 0) == 0)
    nb = std::make_unique<NodeBuilderWithCity>(awsCitiesSorted(), geoAwsPosition());
  else if (name.rfind("CITIES_SPEED=", 0) == 0)
    nb = std::make_unique<NodeBuilderWithCity>(CSVLatencyReader().cities(), geoAllCitiesPosition());
  else
    nb = std::make_unique<NodeBuilderWithRandomPosition>();
  if (name.empty()) return nb;
  if (name.rfind("RANDOM_SPEED=", 0) != 0 && name.rfind("AWS_SPEED=", 0) != 0 && name.rfind("CITIES_SPEED=", 0) != 0)
    throw IllegalArgumentException(name + " not in the registry");
  nb->speedUniform = name.find("SPEED=GAUSSIAN") != std::string::npos;
  size_t p = name.find("_TOR=");
  if (p == std::string::npos) throw IllegalArgumentException(name);
  double tor = atof(name.c_str() + p + 5);
  if (tor > 0.001) nb->torRatio = tor;
  return nb;
}

// ---------------------------------------------------------------- Node
class Node {  // C/Node.java
 public:
  static constexpr int MAX_X = 2000, MAX_Y = 1112;
  static int MAX_DIST() {
    return (int)std::sqrt((MAX_X / 2.0) * (MAX_X / 2.0) + (MAX_Y / 2.0) * (MAX_Y / 2.0));
  }
  int nodeId;
  int x, y;
  std::string cityName;
  int extraLatency = 0;
  bool byzantine;
  double speedRatio = 1.0;
  bool down = false;
  jlong doneAt = 0;
  jlong msgReceived = 0, msgSent = 0, bytesSent = 0, bytesReceived = 0;

  Node(JRandom& rd, NodeBuilder& nb, bool byz = false) : byzantine(byz) {  // :246-271
    nodeId = nb.allocateNodeId();
    jint rdNode = rd.nextInt();
    cityName = nb.getCityName(rdNode);
    x = nb.getX(rdNode);
    y = nb.getY(rdNode);
    if (x <= 0 || x > MAX_X) throw IllegalArgumentException("bad x");
    if (y <= 0 || y > MAX_Y) throw IllegalArgumentException("bad y");
    if (nb.speedUniform)  // UniformSpeed.getSpeedRatio :233-238
      speedRatio = rd.nextBoolean() ? (rd.nextInt(67) + 33) / 100.0 : (rd.nextInt(200) + 100) / 100.0;
    if (nb.torRatio > 0) extraLatency = rd.nextDouble() < nb.torRatio ? 500 : 0;  // :151-161
    if (speedRatio <= 0) throw IllegalArgumentException("speedRatio");
  }
  virtual ~Node() {}
  virtual void start() { down = false; }
  virtual void stop() { down = true; }
  bool isDown() const { return down; }
  int dist(const Node& n) const {  // :278-282
    int dx = std::min(std::abs(x - n.x), MAX_X - std::abs(x - n.x));
    int dy = std::min(std::abs(y - n.y), MAX_Y - std::abs(y - n.y));
    return (int)std::sqrt((double)(dx * dx + dy * dy));
  }
};

// ---------------------------------------------------------------- NetworkLatency
struct NetworkLatency {  // C/NetworkLatency.java:12-34
  virtual ~NetworkLatency() {}
  virtual int getExtendedLatency(const Node& from, const Node& to, int delta) const = 0;
  static void checkDelta(int delta) {
    if (delta < 0 || delta > 99) throw IllegalArgumentException("delta=" + std::to_string(delta));
  }
  int getLatency(const Node& from, const Node& to, int delta) const {
    if (&from == &to) return 1;
    int base = from.extraLatency + to.extraLatency;
    base += getExtendedLatency(from, to, delta);
    return std::max(1, base);
  }
};

struct NetworkLatencyByDistanceWJitter : NetworkLatency {  // :49-73
  GeneralizedParetoDistribution gpd{1.4, -0.3, 0.35};
  double distToMile(int dist) const {
    const double earthPerimeter = 24860;
    const double pointValue = (earthPerimeter / 2) / Node::MAX_DIST();
    return pointValue * dist;
  }
  double getJitter(int delta) const { return gpd.inverseF(delta / 100.0); }
  double getFixedLatency(int dist) const { return distToMile(dist) * 0.022 + 4.862; }
  int getExtendedLatency(const Node& from, const Node& to, int delta) const override {
    checkDelta(delta);
    double raw = getFixedLatency(from.dist(to)) + getJitter(delta);
    return (int)(raw / 2);
  }
};

struct NetworkFixedLatency : NetworkLatency {  // :235-249
  int fixedLatency;
  explicit NetworkFixedLatency(int f) : fixedLatency(std::max(1, f)) {}
  int getExtendedLatency(const Node&, const Node&, int) const override { return fixedLatency; }
};

struct NetworkUniformLatency : NetworkLatency {  // :255-269
  int maxLatency;
  explicit NetworkUniformLatency(int m) : maxLatency(std::max(1, m)) {}
  int getExtendedLatency(const Node&, const Node&, int delta) const override {
    return (int)((delta / 99.0) * maxLatency);
  }
};

struct NetworkNoLatency : NetworkLatency {  // :271-275
  int getExtendedLatency(const Node&, const Node&, int) const override { return 1; }
};

struct MeasuredNetworkLatency : NetworkLatency {  // :277-313
  int longDistrib[100];
  MeasuredNetworkLatency(const std::vector<int>& proportions, const std::vector<int>& values) {
    int li = 0, cur = 0, sum = 0;
    for (size_t i = 0; i < proportions.size(); i++) {
      if (proportions[i] == 0) {
        cur = values[i];
        continue;
      }
      sum += proportions[i];
      int step = (values[i] - cur) / proportions[i];
      for (int ii = 0; ii < proportions[i]; ii++) {
        cur += step;
        if (li >= 100) throw IllegalArgumentException("li");
        longDistrib[li++] = cur;
      }
    }
    if (sum != 100) throw IllegalArgumentException("sum");
    if (li != 100) throw IllegalArgumentException("li");
  }
  int getExtendedLatency(const Node&, const Node&, int delta) const override {
    checkDelta(delta);
    return longDistrib[delta];
  }
};

struct EthScanNetworkLatency : NetworkLatency {  // :366-384
  MeasuredNetworkLatency networkLatency{
      {16, 18, 17, 12, 8, 5, 4, 3, 3, 1, 1, 2, 1, 1, 8},
      {250, 500, 1000, 1250, 1500, 1750, 2000, 2250, 2500, 2750, 4500, 6000, 8500, 9750, 10000}};
  int getExtendedLatency(const Node& from, const Node& to, int delta) const override {
    return networkLatency.getLatency(from, to, delta);
  }
};

struct IC3NetworkLatency : NetworkLatency {  // :399-417
  static constexpr int S10 = 92, SW = 350;
  int getExtendedLatency(const Node& from, const Node& to, int) const override {
    double dist = from.dist(to);
    double surface = dist * dist * M_PI;
    double totalSurface = Node::MAX_X * Node::MAX_Y;
    int position = (int)((surface * 100) / totalSurface);
    if (position <= 10) return S10 / 2;
    if (position <= 33) return 125 / 2;
    if (position <= 50) return 152 / 2;
    if (position <= 67) return 200 / 2;
    if (position <= 90) return 276 / 2;
    return SW / 2;
  }
};

struct AwsRegionNetworkLatency : NetworkLatency {  // C/NetworkLatency.java:86-157
  NetworkLatencyByDistanceWJitter var;
  static int region(const std::string& city) {
    const auto& r = awsRegions();
    for (size_t i = 0; i < r.size(); i++)
      if (r[i] == city) return (int)i;
    return -1;
  }
  static int ping(int a, int b) {  // :113-133 (upper triangle; that's ping time)
    static const int L[10][11] = {{0, 81, 216, 126, 165, 138, 97, 64, 164, 131, 141}, {0, 0, 182, 181, 232, 195, 167, 13, 88, 80, 75},
                                  {0, 0, 0, 152, 62, 223, 123, 194, 111, 122, 113},  {0, 0, 0, 0, 97, 133, 35, 184, 259, 254, 264},
                                  {0, 0, 0, 0, 0, 169, 69, 218, 162, 174, 171},      {0, 0, 0, 0, 0, 0, 105, 210, 282, 269, 271},
                                  {0, 0, 0, 0, 0, 0, 0, 156, 235, 222, 234},         {0, 0, 0, 0, 0, 0, 0, 0, 101, 78, 87},
                                  {0, 0, 0, 0, 0, 0, 0, 0, 0, 24, 13},               {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 12}};
    return L[a][b];
  }
  int getExtendedLatency(const Node& from, const Node& to, int delta) const override {
    const int reg1 = region(from.cityName), reg2 = region(to.cityName);
    if (reg1 < 0 || reg2 < 0) throw IllegalArgumentException("not in our aws cities list");
    if (reg1 == reg2) return 1;
    const int minReg = std::min(reg1, reg2), maxReg = std::max(reg1, reg2);
    return std::max(1, ping(minReg, maxReg) / 2 + (int)var.getJitter(delta));
  }
};

struct NetworkLatencyByCity : NetworkLatency {  // C/NetworkLatency.java:159-198
  CSVLatencyReader reader;
  float cityLatency(const std::string& cityFrom, const std::string& cityTo) const {  // getLatency(String, String) :187-197
    const JHashMap<float>* from = reader.latencyMatrix.get(cityFrom);
    if (!from) throw IllegalArgumentException("Can't find latencies for " + cityFrom);
    const float* res = from->get(cityTo);
    if (!res) res = reader.latencyMatrix.get(cityTo)->get(cityFrom);
    return *res;
  }
  int getExtendedLatency(const Node& from, const Node& to, int) const override {
    if (from.nodeId == to.nodeId) return 1;
    if (from.cityName == "world" || to.cityName == "world")
      throw IllegalStateException("Can't use NetworkLatencyByCity model with default city location");
    return std::max(1, jround_f(0.5f * cityLatency(from.cityName, to.cityName)));
  }
};

struct NetworkLatencyByCityWJitter : NetworkLatencyByCity {  // C/NetworkLatency.java:200-233
  GeneralizedParetoDistribution gpd{1.4, -0.3, 0.35};
  int getExtendedLatency(const Node& from, const Node& to, int delta) const override {
    if (from.nodeId == to.nodeId) return 1;
    if (from.cityName == "world" || to.cityName == "world")
      throw IllegalStateException("Can't use NetworkLatencyByCity model with default city location");
    double raw = gpd.inverseF(delta / 100.0);
    if (from.cityName == to.cityName)
      raw += 10;
    else
      raw += cityLatency(from.cityName, to.cityName);
    return std::max(1, (int)jround(0.5 * raw));
  }
};

// RegistryNetworkLatencies.getByName (C/RegistryNetworkLatencies.java:26-58)
inline std::unique_ptr<NetworkLatency> networkLatencyByName(const std::string& name) {
  if (name == "AwsRegionNetworkLatency") return std::make_unique<AwsRegionNetworkLatency>();
  if (name == "NetworkLatencyByCity") return std::make_unique<NetworkLatencyByCity>();
  if (name == "NetworkLatencyByCityWJitter") return std::make_unique<NetworkLatencyByCityWJitter>();
  if (name.empty() || name == "NetworkLatencyByDistanceWJitter")
    return std::make_unique<NetworkLatencyByDistanceWJitter>();
  if (name.rfind("NetworkFixedLatency(", 0) == 0)
    return std::make_unique<NetworkFixedLatency>(atoi(name.c_str() + 20));
  if (name.rfind("NetworkUniformLatency(", 0) == 0)
    return std::make_unique<NetworkUniformLatency>(atoi(name.c_str() + 22));
  if (name == "NetworkNoLatency") return std::make_unique<NetworkNoLatency>();
  if (name == "IC3NetworkLatency") return std::make_unique<IC3NetworkLatency>();
  if (name == "EthScanNetworkLatency") return std::make_unique<EthScanNetworkLatency>();
  throw IllegalArgumentException("unknown latency " + name);
}

// ---------------------------------------------------------------- Messages
struct Message {  // C/messages/Message.java:15-29
  virtual ~Message() {}
  virtual void action(Network& network, Node* from, Node* to) = 0;
  virtual int size() const { return 1; }
  virtual bool isTask() const { return false; }
};

struct Task : Message {  // C/messages/Task.java:8-31
  std::function<void()> r;
  explicit Task(std::function<void()> rr) : r(std::move(rr)) {}
  int size() const override { return 0; }
  bool isTask() const override { return true; }
  void action(Network&, Node*, Node*) override { r(); }
};

struct ConditionalTask {  // C/messages/ConditionalTask.java:6-36
  std::function<bool()> startIf, repeatIf;
  std::function<void()> r;
  int duration;
  int minStartTime;
  Node* from;
};

// ---------------------------------------------------------------- Envelope
struct MessageArrival {  // C/Network.java:392-412
  Node* dest;
  int arrival;
};

struct Envelope {  // C/Envelope.java:11-43
  int sendTime;
  std::shared_ptr<Message> message;
  int fromNodeId;
  Envelope* nextSameTime = nullptr;
  Envelope(int st, std::shared_ptr<Message> m, int from) : sendTime(st), message(std::move(m)), fromNodeId(from) {}
  virtual ~Envelope() {}
  virtual int getNextDestId() const = 0;
  virtual int nextArrivalTime(const Network& network) const = 0;
  virtual void markRead() = 0;
  virtual bool hasNextReader() const = 0;
};

struct SingleDestEnvelope : Envelope {  // :230-301
  int toNodeId, arrivalTime;
  SingleDestEnvelope(std::shared_ptr<Message> m, const Node& from, const Node& to, int sendTime, int arrival)
      : Envelope(sendTime, std::move(m), from.nodeId), toNodeId(to.nodeId), arrivalTime(arrival) {}
  int getNextDestId() const override { return toNodeId; }
  int nextArrivalTime(const Network&) const override { return arrivalTime; }
  void markRead() override {}
  bool hasNextReader() const override { return false; }
};

struct MultipleDestEnvelope : Envelope {  // :57-155
  jint randomSeed;
  std::vector<int> destIds;
  int curPos = 0;
  MultipleDestEnvelope(std::shared_ptr<Message> m, const Node& from, const std::vector<MessageArrival>& dests,
                       int sendTime, jint seed)
      : Envelope(sendTime, std::move(m), from.nodeId), randomSeed(seed) {
    for (auto& d : dests) destIds.push_back(d.dest->nodeId);
  }
  int getNextDestId() const override { return destIds[curPos]; }
  int arrivalTime(const Network& network, int destId) const;  // :107-113
  int nextArrivalTime(const Network& network) const override { return arrivalTime(network, getNextDestId()); }
  void markRead() override { curPos++; }
  bool hasNextReader() const override { return curPos < (int)destIds.size(); }
};

struct MultipleDestWithDelayEnvelope : Envelope {  // :157-228
  std::vector<int> destIds, arrivals;
  int curPos = 0;
  MultipleDestWithDelayEnvelope(std::shared_ptr<Message> m, const Node& from,
                                const std::vector<MessageArrival>& dests, int sendTime)
      : Envelope(sendTime, std::move(m), from.nodeId) {
    for (auto& d : dests) {
      destIds.push_back(d.dest->nodeId);
      arrivals.push_back(d.arrival);
    }
  }
  int getNextDestId() const override { return destIds[curPos]; }
  int nextArrivalTime(const Network&) const override { return arrivals[curPos]; }
  void markRead() override { curPos++; }
  bool hasNextReader() const override { return curPos < (int)destIds.size(); }
};

// ---------------------------------------------------------------- Network
class Network {
 public:
  static constexpr int duration = 60 * 1000;  // C/Network.java:15

  // MessageStorage (:116-299). Semantics: arrival-ms -> LIFO intrusive list (push at head :145-147,
  // pop head :155-161), paged in 60 000-ms slots. The reference's MsgsSlot.endTime is computed from
  // the un-rounded ctor argument (:122-123, a latent bug that can throw after idle gaps); the oracle
  // uses the rounded start ("A").
  struct MsgsSlot {
    int startTime, endTime;
    std::vector<Envelope*> msgsByMs;
    explicit MsgsSlot(int st) : startTime(st - (st % duration)), msgsByMs(duration, nullptr) {
      endTime = startTime + duration;
    }
    int getPos(int aTime) const {
      if (aTime < startTime || aTime >= startTime + duration) throw IllegalArgumentException("aTime");
      return aTime % duration;
    }
  };
  class MessageStorage {
    Network& net;

   public:
    std::vector<std::unique_ptr<MsgsSlot>> msgsBySlot;
    explicit MessageStorage(Network& n) : net(n) {}
    ~MessageStorage() { clear(); }
    void cleanup() {  // :222-229
      while (!msgsBySlot.empty() && net.time >= msgsBySlot[0]->endTime) {
        freeSlot(*msgsBySlot[0]);
        msgsBySlot.erase(msgsBySlot.begin());
      }
      if (msgsBySlot.empty()) msgsBySlot.push_back(std::make_unique<MsgsSlot>(net.time));
    }
    void ensureSize(int aTime) {  // :231-235
      while (msgsBySlot.back()->endTime <= aTime)
        msgsBySlot.push_back(std::make_unique<MsgsSlot>(msgsBySlot.back()->endTime));
    }
    MsgsSlot& findSlot(int aTime) {  // :237-245
      cleanup();
      ensureSize(aTime);
      int pos = (aTime - msgsBySlot[0]->startTime) / duration;
      if (pos >= (int)msgsBySlot.size()) throw IllegalStateException("pos");
      return *msgsBySlot[pos];
    }
    void addMsg(Envelope* m) {  // :247-255 + :134-148
      int na = m->nextArrivalTime(net);
      if (na < net.time) {
        delete m;
        throw IllegalStateException("Arriving in the past: arrival=" + std::to_string(na));
      }
      MsgsSlot& slot = findSlot(na);
      int pos = slot.getPos(na);
      m->nextSameTime = slot.msgsByMs[pos];
      slot.msgsByMs[pos] = m;
    }
    Envelope* peek(int t) {
      MsgsSlot& s = findSlot(t);
      return s.msgsByMs[s.getPos(t)];
    }
    Envelope* poll(int t) {  // :155-161
      MsgsSlot& s = findSlot(t);
      int pos = s.getPos(t);
      Envelope* m = s.msgsByMs[pos];
      if (m != nullptr) s.msgsByMs[pos] = m->nextSameTime;
      return m;
    }
    int size() const {  // :204-210
      int sz = 0;
      for (auto& ms : msgsBySlot)
        for (Envelope* m : ms->msgsByMs)
          for (; m != nullptr; m = m->nextSameTime) sz++;
      return sz;
    }
    int sizeAt(int t) {  // :212-220
      int sz = 0;
      for (Envelope* cur = peek(t); cur != nullptr; cur = cur->nextSameTime) sz++;
      return sz;
    }
    Envelope* peekFirst() {  // :268-274
      for (auto& ms : msgsBySlot)
        for (Envelope* m : ms->msgsByMs)
          if (m != nullptr) return m;
      return nullptr;
    }
    Envelope* pollFirst() {  // :293-297
      Envelope* m = peekFirst();
      return m == nullptr ? nullptr : poll(m->nextArrivalTime(net));
    }
    void clear() {  // :262-265
      for (auto& s : msgsBySlot) freeSlot(*s);
      msgsBySlot.clear();
    }
    void clearAndCleanup() {
      clear();
      cleanup();
    }

   private:
    static void freeSlot(MsgsSlot& s) {
      for (Envelope*& h : s.msgsByMs) {
        while (h != nullptr) {
          Envelope* n = h->nextSameTime;
          delete h;
          h = n;
        }
      }
    }
  };

  MessageStorage msgs{*this};
  std::list<ConditionalTask*> conditionalTasks;       // :23
  std::vector<std::unique_ptr<ConditionalTask>> ctOwner;
  std::vector<Node*> allNodes;                        // :29 (not owning)
  JRandom rd{0};                                      // :32
  std::vector<int> partitionsInX;                     // :34
  int msgDiscardTime = INT32_MAX;                     // :40
  std::unique_ptr<NetworkLatency> networkLatency = std::make_unique<IC3NetworkLatency>();  // :43
  int time = 0;                                       // :49
  // instrumentation (not in the reference): events delivered, per kind
  uint64_t statDelivered = 0, statTasks = 0;

  // :52-64
  static BitSet chooseBadNodes(JRandom& rd, int nodeCount, int nodesDown) {
    BitSet bad;
    for (int setDown = 0; setDown < nodesDown;) {
      int down = rd.nextInt(nodeCount);
      if (down != 1 && !bad.get(down)) {
        bad.set(down);
        setDown++;
      }
    }
    return bad;
  }
  Node* getNodeById(int id) const { return allNodes.at(id); }
  Network& setMsgDiscardTime(int l) {
    msgDiscardTime = l;
    return *this;
  }

  void run(int seconds) { runMs(seconds * 1000); }  // :306-308
  bool runMs(int ms) {                               // :318-338
    if (ms <= 0) throw IllegalArgumentException("Should be greater than 0. ms=" + std::to_string(ms));
    if (time == 0)
      for (Node* n : allNodes)
        if (!n->isDown()) n->start();
    int endAt = (int)((uint32_t)time + (uint32_t)ms);
    if (endAt <= 0) throw IllegalStateException("Maximum time reached!");
    bool didSomething = receiveUntil(endAt);
    time = endAt;
    return didSomething;
  }

  // ---- send family (:340-447)
  void sendAll(std::shared_ptr<Message> m, int sendTime, Node* from) { send(m, sendTime, from, allNodes); }
  void sendAll(std::shared_ptr<Message> m, Node* from) { send(m, time + 1, from, allNodes); }
  void send(std::shared_ptr<Message> m, Node* from, const std::vector<Node*>& dests) {  // :353-362
    if (dests.empty()) return;
    if (dests.size() == 1)
      send(m, time + 1, from, dests[0]);
    else
      send(m, time + 1, from, dests);
  }
  void send(std::shared_ptr<Message> m, Node* from, Node* to) { send(m, time + 1, from, to); }
  void send(std::shared_ptr<Message> mc, int sendTime, Node* from, Node* to) {  // :369-382
    checkInNetwork(from, "From");
    checkInNetwork(to, "To");
    MessageArrival ms;
    if (createMessageArrival(*mc, from, to, sendTime, rd.nextInt(), ms))
      msgs.addMsg(new SingleDestEnvelope(mc, *from, *to, sendTime, ms.arrival));
  }
  void sendArriveAt(std::shared_ptr<Message> mc, int arriveAt, Node* from, Node* to) {  // :384-390
    if (arriveAt <= time) throw IllegalArgumentException("wrong arrival time: arriveAt=" + std::to_string(arriveAt));
    msgs.addMsg(new SingleDestEnvelope(mc, *from, *to, time, arriveAt));
  }
  void send(std::shared_ptr<Message> m, int sendTime, Node* from, const std::vector<Node*>& dests,
            int delaysBetweenMessage = 0) {  // :418-447
    checkInNetwork(from, "From");
    jint randomSeed = rd.nextInt();
    std::vector<MessageArrival> da = createMessageArrivals(*m, sendTime, from, dests, randomSeed, delaysBetweenMessage);
    if (!da.empty()) {
      Envelope* msg;
      if (da.size() == 1)
        msg = new SingleDestEnvelope(m, *from, *da[0].dest, sendTime, da[0].arrival);
      else if (delaysBetweenMessage == 0)
        msg = new MultipleDestEnvelope(m, *from, da, sendTime, randomSeed);
      else
        msg = new MultipleDestWithDelayEnvelope(m, *from, da, sendTime);
      msgs.addMsg(msg);
    }
  }
  std::vector<MessageArrival> createMessageArrivals(const Message& m, int sendTime, Node* from,
                                                    const std::vector<Node*>& dests, jint randomSeed,
                                                    int delaysBetweenMessage) {  // :449-467
    std::vector<MessageArrival> da;
    da.reserve(dests.size());
    for (Node* n : dests) {
      MessageArrival ma;
      bool ok = createMessageArrival(m, from, n, sendTime, randomSeed, ma);
      sendTime += delaysBetweenMessage + (delaysBetweenMessage > 0 ? 1 : 0);
      if (ok) da.push_back(ma);
    }
    std::stable_sort(da.begin(), da.end(),
                     [](const MessageArrival& a, const MessageArrival& b) { return a.arrival < b.arrival; });
    return da;
  }
  static int getPseudoRandom(jint nodeId, jint randomSeed) {  // :493-503
    uint32_t a = (uint32_t)nodeId;
    a ^= (a << 13);
    a ^= (a >> 17);
    a ^= (a << 5);
    jint x = (jint)(a ^ (uint32_t)randomSeed);
    jint r = x % 100;  // truncated, sign of dividend
    return r < 0 ? -r : r;
  }

  // ---- tasks (:505-531)
  void registerTask(std::function<void()> task, int startAt, Node* from) {
    msgs.addMsg(new SingleDestEnvelope(std::make_shared<Task>(std::move(task)), *from, *from, time, startAt));
  }
  void registerPeriodicTask(std::function<void()> task, int startAt, int period, Node* from,
                            std::function<bool()> cond = [] { return true; });
  void registerConditionalTask(std::function<void()> task, int startAt, int duration_, Node* from,
                               std::function<bool()> startIf, std::function<bool()> repeatIf) {
    auto ct = std::make_unique<ConditionalTask>();
    ct->startIf = std::move(startIf);
    ct->repeatIf = std::move(repeatIf);
    ct->r = std::move(task);
    ct->minStartTime = startAt;
    ct->from = from;
    ct->duration = duration_;
    conditionalTasks.push_back(ct.get());
    ctOwner.push_back(std::move(ct));
  }

  // ---- partitions (:639-649, :693-707)
  int partitionId(const Node& to) const {
    int pId = 0;
    for (int x : partitionsInX) {
      if (x > to.x) return pId;
      pId++;
    }
    return pId;
  }
  void partition(float part) {
    if (part <= 0 || part >= 1) throw IllegalArgumentException("part needs to be a percentage between 0 & 100 excluded");
    int xPoint = (int)(Node::MAX_X * part);
    if (std::find(partitionsInX.begin(), partitionsInX.end(), xPoint) != partitionsInX.end())
      throw IllegalArgumentException("this partition exists already");
    partitionsInX.push_back(xPoint);
    std::sort(partitionsInX.begin(), partitionsInX.end());
  }
  void endPartition() { partitionsInX.clear(); }

  void addNode(Node* node) {  // :651-659
    while ((int)allNodes.size() <= node->nodeId) allNodes.push_back(nullptr);
    if (allNodes[node->nodeId] != nullptr) throw IllegalStateException("There is already a node with this id");
    allNodes[node->nodeId] = node;
  }
  std::vector<Node*> liveNodes() const {
    std::vector<Node*> r;
    for (Node* n : allNodes)
      if (!n->isDown()) r.push_back(n);
    return r;
  }
  Network& setNetworkLatency(std::unique_ptr<NetworkLatency> nl) {  // :670-678
    if (msgs.size() != 0)
      throw IllegalStateException("You can't change the latency while the system as on going messages");
    networkLatency = std::move(nl);
    return *this;
  }

 private:
  void checkInNetwork(Node* n, const char* what) {
    if (n->nodeId >= (int)allNodes.size() || allNodes[n->nodeId] != n)
      throw IllegalArgumentException(std::string("The node is not in the network. ") + what);
  }
  bool createMessageArrival(const Message& m, Node* from, Node* to, int sendTime, jint randomSeed,
                            MessageArrival& out) {  // :469-487
    if (sendTime <= time) throw IllegalStateException("sendTime=" + std::to_string(sendTime) + ", time=" + std::to_string(time));
    from->msgSent++;
    from->bytesSent += m.size();
    if (partitionId(*from) == partitionId(*to) && !from->isDown() && !to->isDown()) {
      int nt = networkLatency->getLatency(*from, *to, getPseudoRandom(to->nodeId, randomSeed));
      if (nt < msgDiscardTime) {
        out.dest = to;
        out.arrival = sendTime + nt;
        return true;
      }
    }
    return false;
  }

  Envelope* nextMessage(int until) {  // :533-570
    std::vector<ConditionalTask*> cts;
    bool haveCts = false;
    while (time <= until) {
      Envelope* m = msgs.poll(time);
      if (m != nullptr) return m;
      time++;
      if (!haveCts) {
        cts.assign(conditionalTasks.begin(), conditionalTasks.end());
        haveCts = true;
      }
      size_t keep = 0;
      for (size_t i = 0; i < cts.size(); i++) {
        ConditionalTask* ct = cts[i];
        if (ct->minStartTime > until || ct->from->isDown()) continue;  // it.remove()
        if (ct->minStartTime <= time) {
          // it.remove()
          if (ct->startIf()) {
            ct->r();
            ct->minStartTime = time + ct->duration;
            if (!ct->repeatIf()) conditionalTasks.remove(ct);
          }
          continue;
        }
        cts[keep++] = ct;
      }
      cts.resize(keep);
    }
    return nullptr;
  }

  bool receiveUntil(int until) {  // :587-637
    int previousTime = time;
    Envelope* next = nextMessage(until);
    if (next == nullptr) return false;
    while (next != nullptr) {
      Envelope* m = next;
      int na = m->nextArrivalTime(*this);
      if (na != previousTime && time > na) throw IllegalStateException("time:" + std::to_string(time));
      Node* from = allNodes[m->fromNodeId];
      Node* to = allNodes[m->getNextDestId()];
      if (!to->isDown() && partitionId(*from) == partitionId(*to)) {
        if (!m->message->isTask()) {
          if (m->message->size() == 0) throw IllegalStateException("Message size should be greater than zero");
          to->msgReceived++;
          to->bytesReceived += m->message->size();
          statDelivered++;
        } else {
          statTasks++;
        }
        std::shared_ptr<Message> keepAlive = m->message;
        keepAlive->action(*this, from, to);
      }
      m->markRead();
      if (m->hasNextReader())
        msgs.addMsg(m);
      else
        delete m;
      previousTime = time;
      next = nextMessage(until);
    }
    return true;
  }
};

inline int MultipleDestEnvelope::arrivalTime(const Network& network, int destId) const {
  int rd = Network::getPseudoRandom(destId, randomSeed);
  Node* f = network.getNodeById(fromNodeId);
  Node* t = network.getNodeById(destId);
  return sendTime + network.networkLatency->getLatency(*f, *t, rd);
}

struct PeriodicTask : Task {  // C/messages/PeriodicTask.java:10-47
  int period;
  Node* sender;
  std::function<bool()> continuationCondition;
  std::weak_ptr<PeriodicTask> self;
  PeriodicTask(std::function<void()> r, Node* from, int p, std::function<bool()> c)
      : Task(std::move(r)), period(p), sender(from), continuationCondition(std::move(c)) {}
  void action(Network& network, Node*, Node*) override {
    r();
    if (continuationCondition()) network.sendArriveAt(self.lock(), network.time + period, sender, sender);
  }
};

inline void Network::registerPeriodicTask(std::function<void()> task, int startAt, int period, Node* from,
                                          std::function<bool()> cond) {
  auto sw = std::make_shared<PeriodicTask>(std::move(task), from, period, std::move(cond));
  sw->self = sw;
  msgs.addMsg(new SingleDestEnvelope(sw, *from, *from, time, startAt));
}

}  // namespace orc
