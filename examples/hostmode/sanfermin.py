"""San Fermin signature aggregation (P/SanFerminSignature.java, P/SanFerminHelper.java) written against the
reference's own protocol API and run on the engine in host-callback mode (wittgenstein_amd.hostnet): the swap
requests / replies, their latency sampling and ordering, the timeout and pairing tasks and the shared `rd` (which
SanFerminHelper.pickNextNodes shuffles with, :144) live in libwittgpu.so on the MI355X; the per-node swap state stays
host objects as in the reference. Host-side Python stand-in for the Java classes (no JVM in the build image,
INTEGRATION.md); class, field and method names follow the Java source. As in the reference, node counts must be powers
of two (toBinaryID's padding throws otherwise, P/SanFerminHelper.java:158-171)."""
from wittgenstein_amd.core import IllegalArgumentException, IllegalStateException
from wittgenstein_amd.hostnet import HostNetwork, Message, Node

OK, NO = 0, 1  # enum Status :520-523


def log2(n):  # C/utils/MoreMath.java:5-10
    if n <= 0:
        raise IllegalArgumentException("n=%d" % n)
    return n.bit_length() - 1


def shuffle(lst, rd):  # java.util.Collections.shuffle(list, rnd)
    for i in range(len(lst), 1, -1):
        j = rd.nextInt(i)
        lst[i - 1], lst[j] = lst[j], lst[i - 1]


class SanFerminHelper:  # P/SanFerminHelper.java
    def __init__(self, n, allNodes, rd):
        self.n, self.allNodes, self.rd = n, allNodes, rd
        self.binaryId = self.toBinaryID(n, len(allNodes))
        self.usedNodes = {}
        self.currentLevel = log2(len(allNodes))

    @staticmethod
    def toBinaryID(node, setSize):  # :168-171
        s = format(node.nodeId, "b")
        width = log2(setSize)
        if len(s) > width:
            raise IllegalStateException("StringIndexOutOfBounds: node id wider than log2(setSize)")
        return "0" * (width - len(s)) + s

    def _range(self, level, candidate):  # getOwnSet :38-56 / getCandidateSet :62-92
        lo, hi = 0, len(self.allNodes)
        currLevel = 0
        while currLevel <= level and lo <= hi:
            m = (hi + lo) // 2
            swap = candidate and currLevel == level  # "when we are at the right level, swap the order"
            if self.binaryId[currLevel] == "0":
                if swap:
                    lo = m
                else:
                    hi = m
            else:
                if swap:
                    hi = m
                else:
                    lo = m
            if hi == lo:
                break
            if hi - 1 == 0 or lo == len(self.allNodes):
                break
            currLevel += 1
        return lo, hi

    def getOwnSet(self, level):
        lo, hi = self._range(level, False)
        return self.allNodes[lo:hi]

    def getCandidateSet(self, level):
        lo, hi = self._range(level, True)
        return self.allNodes[lo:hi]

    def isCandidate(self, node, level):  # :94-96
        return node in self.getCandidateSet(level)

    def pickNextNodes(self, level, howMany):  # :112-146
        candidateSet = list(self.getCandidateSet(level))
        ownSet = self.getOwnSet(level)
        idx = ownSet.index(self.n) if self.n in ownSet else -1
        if idx == -1 or len(ownSet) < idx:
            raise IllegalStateException("pickNextNodes")
        newList = []
        used = self.usedNodes.setdefault(level, set())
        if idx not in used:
            newList.append(candidateSet[idx])
            del candidateSet[idx]
            used.add(idx)
        taken = 0
        for i in range(len(candidateSet)):
            if taken >= howMany:
                break
            if i not in used:
                used.add(i)
                newList.append(candidateSet[i])
                taken += 1
        shuffle(newList, self.rd)
        return newList


class SanFerminSignatureParameters:  # :39-111
    def __init__(self, nodeCount=32768 // 32, threshold=32768 // 32, pairingTime=2, signatureSize=48, replyTimeout=300,
                 candidateCount=1, shuffledLists=False, nodeBuilderName=None, networkLatencyName=None):
        if nodeBuilderName not in (None, "", "RANDOM_SPEED=CONSTANT_TOR=0.00"):
            raise IllegalArgumentException("hostnet.Node builds RANDOM / constant-speed nodes only")
        self.nodeCount, self.powerOfTwo, self.threshold = nodeCount, log2(nodeCount), threshold
        self.pairingTime, self.signatureSize, self.replyTimeout = pairingTime, signatureSize, replyTimeout
        self.candidateCount, self.shuffledLists = candidateCount, shuffledLists
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName


class SwapRequest(Message):  # :553-574
    def __init__(self, p, level, aggValue):
        self.p, self.level, self.aggValue = p, level, aggValue

    def action(self, network, frm, to):
        to.onSwapRequest(frm, self)

    def size(self):
        return 4 + self.p.params.signatureSize


class SwapReply(Message):  # :525-551
    def __init__(self, p, status, level, aggValue):
        self.p, self.status, self.level, self.aggValue = p, status, level, aggValue

    def action(self, network, frm, to):
        to.onSwapReply(frm, self)

    def size(self):
        return 4 + self.p.params.signatureSize


class SanFerminNode(Node):  # :149-517
    def __init__(self, p):
        super().__init__(p.network)
        self.p, self.network = p, p.network
        self.binaryId = SanFerminHelper.toBinaryID(self, p.params.nodeCount)
        self.done = self.thresholdDone = self.isSwapping = False
        self.sentRequests = self.receivedRequests = 0
        self.aggValue = 1
        self.thresholdAt = 0
        self.currentPrefixLength = p.params.powerOfTwo
        self.signatureCache, self.futurSigs = {}, {}
        self.pendingNodes = None
        self.candidateTree = None

    def onSwapRequest(self, node, request):  # :224-264
        self.receivedRequests += 1
        if self.done or request.level != self.currentPrefixLength:
            if request.level in self.signatureCache:
                self.sendSwapReply(node, OK, request.level, self.signatureCache[request.level])
            else:
                self.sendSwapReply(node, NO, self.currentPrefixLength, 0)
                if self.candidateTree.isCandidate(node, request.level):
                    self.signatureCache[request.level] = request.aggValue
            return
        if self.isSwapping:
            self.sendSwapReply(node, OK, request.level, self.aggValue)
            return
        if self.candidateTree.isCandidate(node, self.currentPrefixLength):
            self.transition(request.aggValue)

    def onSwapReply(self, frm, reply):  # :266-316
        if reply.level != self.currentPrefixLength or self.done:
            return
        if self.isSwapping:
            return
        if reply.status == OK:
            if frm.nodeId not in self.pendingNodes:
                if self.candidateTree.isCandidate(frm, self.currentPrefixLength):
                    self.transition(reply.aggValue)
                return
            self.transition(reply.aggValue)
        elif frm.nodeId in self.pendingNodes:
            self.sendToNodes(self.candidateTree.pickNextNodes(self.currentPrefixLength, self.p.params.candidateCount))

    def sendToNodes(self, candidates):  # :322-363
        if not candidates:
            return
        self.pendingNodes.update(n.nodeId for n in candidates)
        self.sentRequests += len(candidates)
        self.network.send(SwapRequest(self.p, self.currentPrefixLength, self.aggValue), self, candidates)
        currLevel = self.currentPrefixLength

        def timeout():
            if not self.done and self.currentPrefixLength == currLevel:
                self.sendToNodes(self.candidateTree.pickNextNodes(self.currentPrefixLength,
                                                                  self.p.params.candidateCount))
        self.network.registerTask(timeout, self.network.time + self.p.params.replyTimeout, self)

    def goNextLevel(self):  # :373-414
        if self.done:
            return
        params = self.p.params
        if self.aggValue >= params.threshold and not self.thresholdDone:
            self.thresholdDone = True
            self.thresholdAt = self.network.time + params.pairingTime * 2
        if self.currentPrefixLength == 0 and not self.done:
            self.doneAt = self.network.time + params.pairingTime * 2
            self.p.finishedNodes.append(self)
            self.done = True
            return
        self.currentPrefixLength -= 1
        self.signatureCache[self.currentPrefixLength] = self.aggValue
        self.isSwapping = False
        self.pendingNodes = set()
        if self.currentPrefixLength in self.futurSigs:
            self.aggValue += self.futurSigs[self.currentPrefixLength]
            self.goNextLevel()
            return
        self.sendToNodes(self.candidateTree.pickNextNodes(self.currentPrefixLength, params.candidateCount))

    def sendSwapReply(self, n, s, level, value):  # :416-423
        self.network.send(SwapReply(self.p, s, level, value), self, [n])

    def transition(self, toAggregate):  # :429-450
        self.isSwapping = True

        def verified():
            self.aggValue += toAggregate
            self.goNextLevel()
        self.network.registerTask(verified, self.network.time + self.p.params.pairingTime, self)


class SanFerminSignature:  # :24-146
    def __init__(self, params=None, config=None):
        self.params = params or SanFerminSignatureParameters()
        self.network = HostNetwork(self.params.networkLatencyName, config)
        self.allNodes, self.finishedNodes = [], []
        for _ in range(self.params.nodeCount):
            n = SanFerminNode(self)
            self.allNodes.append(n)
            self.network.addNode(n)
        for n in self.allNodes:
            n.candidateTree = SanFerminHelper(n, self.allNodes, self.network.rd)

    def copy(self):
        return SanFerminSignature(self.params)

    def init(self):  # :139-141
        for n in self.allNodes:
            self.network.registerTask(n.goNextLevel, 1, n)
