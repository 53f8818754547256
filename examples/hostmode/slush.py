"""protocols.Slush (P/Slush.java) and protocols.Snowflake (P/Snowflake.java) — the sampling protocols of the Avalanche family —
written against the reference's own protocol API and run on the engine in host-callback mode (wittgenstein_amd.hostnet): the
queries (one multi-destination send to K random remotes, C/Network.java:353-362,418-447), the answers, their latency sampling
and ordering and the shared `rd` that picks the remotes live in libwittgpu.so on the MI355X; the per-node colour / round state
stays host objects as in the reference. Host-side Python stand-in for the Java classes (no JVM in the build image,
INTEGRATION.md); class, field and method names follow the Java source."""
from wittgenstein_amd.hostnet import HostNetwork, Message, Node

COLOR_NB = 2


class SlushParameters:  # P/Slush.java:17-52
    def __init__(self, NODES_AV=100, M=4, K=7, A=4, nodeBuilderName=None, networkLatencyName=None):
        self.NODES_AV, self.M, self.K, self.A = NODES_AV, M, K, A
        self.AK = K * A
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName


class SnowflakeParameters(SlushParameters):  # P/Snowflake.java:18-58
    def __init__(self, nodeAv=100, M=4, K=7, A=4, B=7, nodeBuilderName=None, networkLatencyName=None):
        super().__init__(nodeAv, M, K, A, nodeBuilderName, networkLatencyName)
        self.B = B
        self.AK = A * K


class Query(Message):  # :86-99
    def __init__(self, id, color):
        self.id, self.color = id, color

    def action(self, network, frm, to):
        to.onQuery(self, frm)


class AnswerQuery(Message):  # :101-114
    def __init__(self, originalQuery, color):
        self.originalQuery, self.color = originalQuery, color

    def action(self, network, frm, to):
        to.onAnswer(self.originalQuery.id, self.color)


class Answer:  # :215-231
    def __init__(self, round):
        self.round = round
        self.colorsFound = [0] * (COLOR_NB + 1)

    def answerCount(self):
        return sum(self.colorsFound)


class SlushNode(Node):  # :116-213
    def __init__(self, p):
        super().__init__(p.network)
        self.p = p
        self.myColor = 0
        self.myQueryNonce = 0
        self.round = 0
        self.answerIP = {}

    def randomRemotes(self):  # :126-137
        net, K, res = self.p.network, self.p.params.K, []
        while len(res) != K:
            r = net.rd.nextInt(self.p.params.NODES_AV)
            if r != self.nodeId and net.getNodeById(r) not in res:
                res.append(net.getNodeById(r))
        return res

    def otherColor(self):  # :139-141
        return 2 if self.myColor == 1 else 1

    def onQuery(self, qa, frm):  # :148-154
        if self.myColor == 0:
            self.myColor = qa.color
            self.sendQuery(1)
        self.p.network.send(AnswerQuery(qa, self.myColor), self, frm)

    def onAnswer(self, queryId, color):  # :161-176
        asw = self.answerIP[queryId]
        asw.colorsFound[color] += 1
        if asw.answerCount() == self.p.params.K:
            del self.answerIP[queryId]
            if asw.colorsFound[self.otherColor()] > self.p.params.AK:
                self.myColor = self.otherColor()
            if self.round < self.p.params.M:
                self.round += 1
                self.sendQuery(asw.round + 1)

    def sendQuery(self, countInM):  # :178-182
        self.myQueryNonce += 1
        q = Query(self.myQueryNonce, self.myColor)
        self.answerIP[q.id] = Answer(countInM)
        self.p.network.send(q, self, self.randomRemotes())


class SnowflakeNode(SlushNode):  # P/Snowflake.java:116-226
    def __init__(self, p):
        super().__init__(p)
        self.cnt = 0

    def onAnswer(self, queryId, color):  # :173-192
        asw = self.answerIP[queryId]
        asw.colorsFound[color] += 1
        if asw.answerCount() == self.p.params.K:
            del self.answerIP[queryId]
            if asw.colorsFound[self.otherColor()] > self.p.params.AK:
                self.myColor = self.otherColor()
                self.cnt = 0
            elif asw.colorsFound[self.myColor] > self.p.params.AK:
                self.cnt += 1
            if self.cnt <= self.p.params.B:
                self.sendQuery(asw.round + 1)


class Slush:
    NODE = SlushNode

    def __init__(self, params=None, config=None):
        self.params = params or SlushParameters()
        self._config = config
        self.network = HostNetwork(self.params.networkLatencyName, config)  # :54-60

    def copy(self):
        return type(self)(self.params, self._config)

    def init(self):  # :62-74
        net = self.network
        for _ in range(self.params.NODES_AV):
            net.addNode(self.NODE(self))
        uncolored1, uncolored2 = net.getNodeById(0), net.getNodeById(1)
        uncolored1.myColor = 1
        uncolored1.sendQuery(1)
        uncolored2.myColor = 2
        uncolored2.sendQuery(1)

    def getDominantColor(self):  # :283-290
        colors = [0, 0, 0]
        for n in self.network.allNodes:
            colors[n.myColor] += 1
        return colors


class Snowflake(Slush):  # P/Snowflake.java
    NODE = SnowflakeNode

    def __init__(self, params=None, config=None):
        super().__init__(params or SnowflakeParameters(), config)
