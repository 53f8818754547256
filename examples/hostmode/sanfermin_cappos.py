"""San Fermin, Cappos' variant (P/SanFerminCappos.java) written against the reference's own protocol API and run on the
engine in host-callback mode (wittgenstein_amd.hostnet) — SURVEY.md §8(f)-1 names it with SanFerminSignature as what that
mode must carry. One Swap message (wantReply) instead of request / reply, a list of cached signatures per level,
candidate sets re-tried on a timeout; SanFerminHelper is the one of examples/hostmode/sanfermin.py. Host-side Python
stand-in for the Java classes (no JVM in the build image); class, field and method names follow the Java
source. Checked against oracle/sanfermin_cappos.hpp after every chunk: tests/test_zx_gpu_sanfermin_cappos.py."""
from wittgenstein_amd.core import IllegalArgumentException, IllegalStateException
from wittgenstein_amd.hostnet import HostNetwork, Message, Node

from .sanfermin import SanFerminHelper, log2


class SanFerminParameters:  # :44-108
    def __init__(self, nodeCount=32768 // 16, threshold=32768 // 32, pairingTime=2, signatureSize=48, timeout=150,
                 candidateCount=50, nodeBuilderName=None, networkLatencyName=None):
        if nodeBuilderName not in (None, "", "RANDOM_SPEED=CONSTANT_TOR=0.00"):
            raise IllegalArgumentException("hostnet.Node builds RANDOM / constant-speed nodes only")
        self.nodeCount, self.threshold, self.pairingTime, self.signatureSize = nodeCount, threshold, pairingTime, signatureSize
        self.timeout, self.candidateCount = timeout, candidateCount
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName
        self.finishedNodes = None


class Swap(Message):  # :437-460
    def __init__(self, p, level, aggValue, reply):
        self.p, self.level, self.wantReply, self.aggValue = p, level, reply, aggValue

    def action(self, network, frm, to):
        to.onSwap(frm, self)

    def size(self):
        return 4 + self.p.params.signatureSize


class SanFerminNode(Node):  # :146-435
    def __init__(self, p):
        super().__init__(p.network)
        self.p, self.network = p, p.network
        self.binaryId = SanFerminHelper.toBinaryID(self, p.params.nodeCount)
        self.helper = None
        self.done = self.thresholdDone = self.isSwapping = False
        self.aggValue = 1
        self.thresholdAt = 0
        self.currentPrefixLength = log2(p.params.nodeCount)
        self.signatureCache = {}

    def onSwap(self, frm, swap):  # :200-237
        wantReply = swap.wantReply
        if self.done or swap.level != self.currentPrefixLength:
            isValueCached = swap.level in self.signatureCache
            if wantReply and isValueCached:
                self.sendSwap([frm], swap.level, self.getBestCachedSig(swap.level), False)
            elif self.helper.isCandidate(frm, swap.level):
                self.putCachedSig(swap.level, swap.aggValue)
            return
        if wantReply:
            self.sendSwap([frm], swap.level, self.totalNumberOfSigs(swap.level), False)
        goodLevel = swap.level == self.currentPrefixLength
        isCandidate = self.helper.isCandidate(frm, self.currentPrefixLength)
        if isCandidate and goodLevel and not self.isSwapping:
            self.transition(swap.level, swap.aggValue)

    def tryNextNodes(self, candidates):  # :239-279
        if not candidates:
            return
        for n in candidates:
            if not self.helper.isCandidate(n, self.currentPrefixLength):
                raise IllegalStateException()
        self.sendSwap(candidates, self.currentPrefixLength, self.totalNumberOfSigs(self.currentPrefixLength + 1), True)
        currLevel = self.currentPrefixLength

        def timeout():
            if not self.done and self.currentPrefixLength == currLevel:
                self.tryNextNodes(self.helper.pickNextNodes(self.currentPrefixLength, self.p.params.candidateCount))
        self.network.registerTask(timeout, self.network.time + self.p.params.timeout, self)

    def goNextLevel(self):  # :281-321
        if self.done:
            return
        params = self.p.params
        enoughSigs = self.totalNumberOfSigs(self.currentPrefixLength) >= params.threshold
        noMoreSwap = self.currentPrefixLength == 0
        if enoughSigs and not self.thresholdDone:
            self.thresholdDone = True
            self.thresholdAt = self.network.time + params.pairingTime * 2
        if noMoreSwap and not self.done:
            self.doneAt = self.network.time + params.pairingTime * 2
            params.finishedNodes.append(self)
            self.done = True
            return
        self.currentPrefixLength -= 1
        self.isSwapping = False
        if self.currentPrefixLength in self.signatureCache:
            self.goNextLevel()
            return
        self.tryNextNodes(self.helper.pickNextNodes(self.currentPrefixLength, params.candidateCount))

    def sendSwap(self, nodes, level, value, wantReply):  # :323-326
        self.network.send(Swap(self.p, level, value, wantReply), self, list(nodes))

    def totalNumberOfSigs(self, level):  # :328-335
        return sum(max(v) for k, v in self.signatureCache.items() if k >= level) + 1

    def transition(self, level, toAggregate):  # :337-347
        self.isSwapping = True

        def verified():
            self.putCachedSig(level, toAggregate)
            self.goNextLevel()
        self.network.registerTask(verified, self.network.time + self.p.params.pairingTime, self)

    def getBestCachedSig(self, level):  # :349-353
        cached = self.signatureCache.get(level, [])
        if not cached:
            raise IllegalStateException("NoSuchElementException")
        return max(cached)

    def putCachedSig(self, level, value):  # :355-366
        self.signatureCache.setdefault(level, []).append(value)
        if self.totalNumberOfSigs(self.currentPrefixLength) >= self.p.params.threshold and not self.thresholdDone:
            self.thresholdDone = True
            self.thresholdAt = self.network.time + self.p.params.pairingTime * 2


class SanFerminCappos:  # :24-141
    def __init__(self, params=None, config=None, batched=None):
        self.params = params or SanFerminParameters()
        self.network = HostNetwork(self.params.networkLatencyName, config, batched=batched)
        self.allNodes = []

    def copy(self):
        return SanFerminCappos(self.params)

    def init(self):  # :119-134: the nodes are built here, from rd
        self.allNodes = []
        for _ in range(self.params.nodeCount):
            n = SanFerminNode(self)
            self.allNodes.append(n)
            self.network.addNode(n)
        for n in self.allNodes:
            n.helper = SanFerminHelper(n, self.allNodes, self.network.rd)
        self.params.finishedNodes = []
        for n in self.allNodes:
            self.network.registerTask(n.goNextLevel, 1, n)
