"""Handel (P/Handel.java) written against the reference's own protocol API and run on the engine in host-callback mode
(wittgenstein_amd.hostnet) — including the two attack scenarios of its parameters, `byzantineSuicide` (:64-69, 406,
538-559, 577-584, 688-694) and `hiddenByzantine` (:70-71, 303, 813-817, 840-917), which the resident device form runs
too (wittgenstein_amd/csrc/proto_handel.hip.h). Queue, LIFO / chain ordering, latency sampling and the shared `rd` live in libwittgpu.so on the
MI355X; HNode / HLevel / SigToVerify / SendSigs stay host objects as in the reference. Host-side Python stand-in for the
Java classes (no JVM in the build image); class, field and method names follow the Java source. BitSets
are Python ints (bit j = node id j). Checked against oracle/handel.hpp after every chunk: tests/test_gpu_handel_hostmode.py."""
import math

from wittgenstein_amd.core import IllegalArgumentException, IllegalStateException
from wittgenstein_amd.hostnet import HostNetwork, Message, Node

INT_MAX = 2**31 - 1


def shuffle(lst, rd):  # java.util.Collections.shuffle(list, rnd)
    for i in range(len(lst), 1, -1):
        j = rd.nextInt(i)
        lst[i - 1], lst[j] = lst[j], lst[i - 1]


def card(b):
    return bin(b).count("1")


def bits(b):  # BitSet.stream(): set bits in ascending order
    i = 0
    while b:
        if b & 1:
            yield i
        b >>= 1
        i += 1


class WindowParameters:  # :147-206 with ScoringExp(2, 4)
    def __init__(self, initial=16, minimum=1, maximum=128):
        self.initial, self.minimum, self.maximum = initial, minimum, maximum

    def newSize(self, cur, correct):
        updated = int(math.ceil(cur * 2.0)) if correct else int(math.floor(cur / 4.0))
        return max(self.minimum, min(self.maximum, updated))


class HandelParameters:  # :22-142, constructor argument order preserved
    def __init__(self, nodeCount, threshold, pairingTime, levelWaitTime, extraCycle, disseminationPeriodMs, fastPath,
                 nodesDown, nodeBuilderName=None, networkLatencyName=None, desynchronizedStart=0,
                 byzantineSuicide=False, hiddenByzantine=False, badNodes=None):
        if nodesDown >= nodeCount or nodesDown < 0 or threshold > nodeCount or nodesDown + threshold > nodeCount:
            raise IllegalArgumentException("nodeCount=%d, threshold=%d" % (nodeCount, threshold))
        if bin(nodeCount).count("1") != 1:
            raise IllegalArgumentException("We support only power of two nodes in this simulation")
        if byzantineSuicide and hiddenByzantine:
            raise IllegalArgumentException("Only one attack at a time")
        self.nodeCount, self.threshold, self.pairingTime, self.levelWaitTime = nodeCount, threshold, pairingTime, levelWaitTime
        self.extraCycle, self.disseminationPeriodMs, self.fastPath, self.nodesDown = extraCycle, disseminationPeriodMs, fastPath, nodesDown
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName
        self.desynchronizedStart, self.byzantineSuicide, self.hiddenByzantine = desynchronizedStart, byzantineSuicide, hiddenByzantine
        self.badNodes = badNodes
        self.window = WindowParameters()


class SigToVerify:  # :919-938 (identity semantics: no equals())
    __slots__ = ("frm", "level", "rank", "sig", "badSig")

    def __init__(self, frm, level, rank, sig, badSig):
        self.frm, self.level, self.rank, self.sig, self.badSig = frm, level, rank, sig, badSig


class SendSigs(Message):  # :239-276
    def __init__(self, sigs, l):
        self.sigs = sigs  # (BitSet) sigs.clone(): ints are immutable
        self.level = l.level
        self._size = 1 + l.expectedSigs() // 8 + 96 * 2
        self.levelFinished = l.incomingComplete()
        self.badSig = False
        if sigs == 0 or card(sigs) > l.size:
            raise IllegalStateException("bad level: %d" % l.level)

    def size(self):
        return self._size

    def action(self, network, frm, to):
        to.onNewSig(frm, self)


class HLevel:  # :371-643
    def __init__(self, n, previous=None, allPreviousNodes=0):
        self.n = n
        p = n.h.params
        self.suicideBizAfter = 0 if p.byzantineSuicide else -1  # :406
        self.toVerifyAgg = []
        self.outgoingFinished = False
        self.posInLevel = 0
        self.peers = []
        self.toVerifyInd = self.finishedPeers = 0
        if previous is None:  # :413-421
            self.level, self.size = 0, 1
            self.outgoingFinished = True
            self.waitedSigs = 0
            self.lastAggVerified = self.verifiedIndSignatures = self.totalIncoming = 1 << n.nodeId
            self.totalOutgoing = 0
        else:  # :424-435
            self.level = previous.level + 1
            self.waitedSigs = n.allSigsAtLevel(self.level) & ~allPreviousNodes
            self.totalOutgoing = 1 << n.nodeId
            self.size = card(self.waitedSigs)
            self.lastAggVerified = self.verifiedIndSignatures = self.totalIncoming = 0

    def expectedSigs(self):
        return self.size

    def expectedNodes(self):  # :446-455
        return [self.n.h.network.getNodeById(i) for i in bits(self.waitedSigs)]

    def isOpen(self):  # :458-472
        if self.outgoingFinished:
            return False
        if self.n.h.network.time >= (self.level - 1) * self.n.h.params.levelWaitTime:
            return True
        return self.outgoingComplete()

    def doCycle(self):  # :474-484
        if not self.isOpen():
            return
        dest = self.getRemainingPeers(1)
        if dest:
            self.n.h.network.send(SendSigs(self.totalOutgoing, self), self.n, dest[0])

    def getRemainingPeers(self, peersCt):  # :486-508
        res = []
        start = self.posInLevel
        while peersCt > 0 and not self.outgoingFinished:
            p = self.peers[self.posInLevel]
            self.posInLevel += 1
            if self.posInLevel >= len(self.peers):
                self.posInLevel = 0
            if not (self.finishedPeers >> p.nodeId) & 1 and not (self.n.blacklist >> p.nodeId) & 1:
                res.append(p)
                peersCt -= 1
            elif self.posInLevel == start:
                self.outgoingFinished = True
        return res

    def buildEmissionList(self, emissions):  # :510-522
        if self.peers:
            raise IllegalStateException()
        for ranks in emissions:
            if ranks:
                if len(ranks) > 1:
                    shuffle(ranks, self.n.h.network.rd)
                self.peers.extend(ranks)

    def incomingComplete(self):
        return self.waitedSigs == self.totalIncoming

    def outgoingComplete(self):
        return card(self.totalOutgoing) == self.size

    def sizeIfIncluded(self, sig):  # :532-540
        c = sig.sig
        if not c & self.totalIncoming:
            c |= self.totalIncoming
        return card(c | self.verifiedIndSignatures)

    def createSuicideByzantineSig(self, maxRank):  # :538-559
        reset = False
        n = self.n
        for i in range(self.suicideBizAfter, len(self.peers)):
            p = self.peers[i]
            if p.isDown() and not (n.blacklist >> p.nodeId) & 1:
                if not reset:
                    self.suicideBizAfter = i
                    reset = True
                if n.receptionRanks[p.nodeId] < maxRank:
                    return SigToVerify(p.nodeId, self.level, n.receptionRanks[p.nodeId], self.waitedSigs, True)
        if not reset:
            self.suicideBizAfter = -1  # no byzantine nodes left in this level
        return None

    def bestToVerify(self):  # :566-634
        if not self.toVerifyAgg:
            return None
        n = self.n
        if n.currWindowSize < 1:
            raise IllegalStateException()
        windowIndex = min(s.rank for s in self.toVerifyAgg)
        if self.suicideBizAfter >= 0:
            bSig = self.createSuicideByzantineSig(windowIndex + n.currWindowSize)
            if bSig is not None:
                self.toVerifyAgg.append(bSig)
                n.sigQueueSize += 1
                return bSig
        curSignatureSize = card(self.totalIncoming)
        bestOutside = bestInside = None
        bestScoreInside = 0
        removed = 0
        curated = []
        for stv in self.toVerifyAgg:
            s = self.sizeIfIncluded(stv)
            if not (n.blacklist >> stv.frm) & 1 and s > curSignatureSize:
                curated.append(stv)
                if stv.rank <= windowIndex + n.currWindowSize:
                    score = n.score(self, stv.sig)
                    if score > bestScoreInside:
                        bestScoreInside, bestInside = score, stv
                elif bestOutside is None or stv.rank < bestOutside.rank:
                    bestOutside = stv
            else:
                removed += 1
        if removed > 0:  # replaceToVerifyAgg :636-646
            oldSize = len(self.toVerifyAgg)
            self.toVerifyAgg = curated
            n.sigQueueSize += len(curated) - oldSize
            if n.sigQueueSize < 0:
                raise IllegalStateException("sigQueueSize=%d" % n.sigQueueSize)
        return bestInside if bestInside is not None else bestOutside


class HiddenByzantine:  # :840-917
    def __init__(self):
        self.noByzantinePeers = False
        self.last = None

    @staticmethod
    def firstByzantine(t, l):  # :844-858
        best, bestRank = None, INT_MAX
        for p in l.peers:
            if p.isDown() and t.receptionRanks[p.nodeId] < bestRank and not (l.totalIncoming >> p.nodeId) & 1:
                bestRank, best = t.receptionRanks[p.nodeId], p
                if bestRank == 0:
                    return p
        return best

    def attack(self, target, currentBest):  # :861-916
        if self.noByzantinePeers:
            return currentBest
        if self.last is currentBest:  # a previous attack finally worked
            self.last = None
            return currentBest
        l = target.levels[currentBest.level]
        if self.last is not None:
            if any(x is self.last for x in l.toVerifyAgg):
                return currentBest
            if not (l.totalIncoming >> self.last.frm) & 1:
                raise IllegalStateException("byz signature pruned!")
            self.last = None
        fb = self.firstByzantine(target, l)
        if fb is None:
            self.noByzantinePeers = True
            return currentBest
        if target.receptionRanks[fb.nodeId] >= currentBest.rank:
            return currentBest
        bad = SigToVerify(fb.nodeId, l.level, target.receptionRanks[fb.nodeId], 1 << fb.nodeId, False)
        l.toVerifyAgg.append(bad)
        target.sigQueueSize += 1
        newBest = l.bestToVerify()
        if newBest is not bad:
            self.last = bad
        return newBest


class HNode(Node):  # :278-838
    def __init__(self, h, startAt, byzantine):
        super().__init__(h.network)
        self.h = h
        self.byzantine = byzantine
        self.startAt = startAt
        self.levels = []
        self.nodePairingTime = int(max(1, h.params.pairingTime * 1.0))  # speedRatio 1: the constant-speed builders
        self.receptionRanks = [0] * h.params.nodeCount
        self.blacklist = 0
        self.currWindowSize = h.params.window.initial
        self.addedCycle = h.params.extraCycle
        self.done = False
        self.sigsChecked = self.sigQueueSize = self.msgFiltered = 0
        self.hiddenByzantine = HiddenByzantine() if h.params.hiddenByzantine and not byzantine else None

    def initLevel(self):  # :319-329
        rounded = 1 << (self.h.params.nodeCount - 1).bit_length()
        allPreviousNodes = 0
        last = HLevel(self)
        self.levels.append(last)
        l = 1
        while (1 << l) <= rounded:
            allPreviousNodes |= last.waitedSigs
            last = HLevel(self, last, allPreviousNodes)
            self.levels.append(last)
            l += 1

    def dissemination(self):  # :331-343
        if self.doneAt > 0:
            if self.addedCycle > 0:
                self.addedCycle -= 1
            else:
                return
        for sfl in self.levels:
            sfl.doCycle()

    def hasSigToVerify(self):
        return self.sigQueueSize != 0

    def score(self, l, sig):  # :655-668
        cla = card(l.lastAggVerified)
        if cla >= l.expectedSigs():
            return 0
        if not l.lastAggVerified & sig:
            return cla + card(sig)
        return max(0, card(l.verifiedIndSignatures | sig) - cla)

    def allSigsAtLevel(self, rnd):  # :671-684
        if rnd < 1:
            raise IllegalArgumentException("round=%d" % rnd)
        cMask = (1 << rnd) - 1
        start = (cMask | self.nodeId) ^ cMask
        end = min(self.nodeId | cMask, self.h.params.nodeCount - 1)
        res = ((1 << (end + 1)) - 1) & ~((1 << start) - 1)
        return res & ~(1 << self.nodeId)

    def updateVerifiedSignatures(self, vs):  # :686-754
        p = self.h.params
        if vs.badSig:
            self.blacklist |= 1 << vs.frm
            if not p.byzantineSuicide:
                raise IllegalStateException("We should not have invalid signatures in this scenario")
            return
        vsl = self.levels[vs.level]
        if vs.sig & ~vsl.waitedSigs:
            raise IllegalStateException("bad signature received")
        vsl.toVerifyInd &= ~(1 << vs.frm)
        for i, x in enumerate(vsl.toVerifyAgg):  # toVerifyAgg.remove(vs): identity
            if x is vs:
                del vsl.toVerifyAgg[i]
                break
        vsl.verifiedIndSignatures |= 1 << vs.frm
        improved = False
        if not (vsl.totalIncoming >> vs.frm) & 1:
            vsl.totalIncoming |= 1 << vs.frm
            improved = True
        if card(vs.sig | vsl.verifiedIndSignatures) > card(vsl.verifiedIndSignatures):
            improved = True
            if vsl.lastAggVerified & vs.sig:
                vsl.lastAggVerified = 0
            vsl.lastAggVerified |= vs.sig
            vsl.totalIncoming = vsl.lastAggVerified | vsl.verifiedIndSignatures
        if not improved:
            return
        justCompleted = vsl.incomingComplete()
        cur = 0
        for l in self.levels:
            if l.level > vsl.level:
                l.totalOutgoing = cur
                if justCompleted and p.fastPath > 0 and not l.outgoingFinished and l.outgoingComplete():
                    peers = l.getRemainingPeers(p.fastPath)
                    self.h.network.send(SendSigs(l.totalOutgoing, l), self, peers)
            cur |= l.totalIncoming
        if self.doneAt == 0 and card(cur) >= p.threshold:
            self.doneAt = self.h.network.time

    def onNewSig(self, frm, ssigs):  # :757-790
        if self.doneAt > 0:
            self.msgFiltered += 1
            return
        if self.h.network.time < self.startAt or (self.blacklist >> frm.nodeId) & 1:
            return
        l = self.levels[ssigs.level]
        if ssigs.sigs & ~l.waitedSigs:
            raise IllegalStateException("bad signatures received")
        cs = ssigs.sigs & l.waitedSigs
        if cs != ssigs.sigs or ssigs.sigs == 0:
            raise IllegalStateException("bad message")
        if ssigs.levelFinished:
            l.finishedPeers |= 1 << frm.nodeId
        if not (l.verifiedIndSignatures >> frm.nodeId) & 1:
            l.toVerifyInd |= 1 << frm.nodeId
        self.sigQueueSize += 1
        l.toVerifyAgg.append(SigToVerify(frm.nodeId, l.level, self.receptionRanks[frm.nodeId], cs, ssigs.badSig))

    def checkSigs(self):  # :796-837
        byLevels = []
        for l in self.levels:
            ss = l.bestToVerify()
            if ss is not None:
                byLevels.append(ss)
        if not byLevels:
            return
        best = byLevels[self.h.network.rd.nextInt(len(byLevels))]  # chooseBestFromLevels :788-790
        if self.hiddenByzantine is not None and best.level == len(self.levels) - 1:
            best = self.hiddenByzantine.attack(self, best)
        l = self.levels[best.level]
        p = self.h.params
        self.currWindowSize = min(p.window.newSize(self.currWindowSize, not best.badSig), l.size)
        r = (self.receptionRanks[best.frm] + p.nodeCount) & 0xFFFFFFFF  # int overflow, then the clamp of :826-828
        self.receptionRanks[best.frm] = INT_MAX if r & 0x80000000 else r
        self.sigsChecked += 1
        self.h.network.registerTask(lambda: self.updateVerifiedSignatures(best), self.h.network.time + self.nodePairingTime, self)


class Handel:  # :18-1054
    def __init__(self, params, config=None, batched=None):
        self.params = params
        self.network = HostNetwork(params.networkLatencyName, config, batched=batched)

    def chooseBadNodes(self):  # C/Network.java:52-64
        rd, p = self.network.rd, self.params
        bad = 0
        setDown = 0
        while setDown < p.nodesDown:
            down = rd.nextInt(p.nodeCount)
            if down != 1 and not (bad >> down) & 1:
                bad |= 1 << down
                setDown += 1
        return bad

    def setReceivingRanks(self):  # :940-948
        expected = list(self.network.allNodes)
        for n in self.network.allNodes:
            shuffle(expected, self.network.rd)
            for i, e in enumerate(expected):
                n.receptionRanks[e.nodeId] = i

    def init(self):  # :957-1014
        p, net = self.params, self.network
        badNodes = p.badNodes if p.badNodes is not None else self.chooseBadNodes()
        for i in range(p.nodeCount):
            startAt = 0 if p.desynchronizedStart == 0 else net.rd.nextInt(p.desynchronizedStart)
            byz = (p.byzantineSuicide or p.hiddenByzantine) and bool((badNodes >> i) & 1)
            n = HNode(self, startAt, byz)
            if (badNodes >> i) & 1:
                n.stop()
            net.addNode(n)
        for n in net.allNodes:
            n.initLevel()
            if not n.isDown():
                net.registerPeriodicTask(n.dissemination, n.startAt + 1, p.disseminationPeriodMs, n)
                net.registerConditionalTask(n.checkSigs, n.startAt + 1, n.nodePairingTime, n, n.hasSigToVerify,
                                            lambda n=n: not n.done)
        self.setReceivingRanks()
        for sender in net.allNodes:
            if sender.isDown():
                continue
            for l in sender.levels:
                emissionList = [None] * p.nodeCount
                for receiver in l.expectedNodes():
                    recRank = receiver.receptionRanks[sender.nodeId]
                    if emissionList[recRank] is None:
                        emissionList[recRank] = []
                    emissionList[recRank].append(receiver)
                l.buildEmissionList(emissionList)

    def contIf(self):  # newContIf :1044-1053
        return any(not n.isDown() and (n.doneAt == 0 or n.addedCycle > 0) for n in self.network.allNodes)
