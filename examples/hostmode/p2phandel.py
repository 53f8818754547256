"""protocols.P2PHandel (P/P2PHandel.java) — signature aggregation over a peer-to-peer network: every node sends, every
sigsSendPeriod, what the neediest of its peers lacks, keeps what it receives in a HashSet<BitSet> (`toVerify`) and verifies, on a
conditional task, the best element (checkSigs1) or the union of all of them (checkSigs2) — on the engine in host-callback mode over
the reference's P2P layer (p2p.P2PNetwork). The third P2PNetwork protocol that runs this way.

What makes its trajectory depend on the JDK, and how this file keeps it: checkSigs2 ORs the set's other elements INTO the first
one its iterator returns (:455-464), and that BitSet is the `sigs` object of a SendSigs message — one object for every receiver of
a multi-destination send and for every set it was added to. Host-callback mode hands every receiver the SAME Python message
object, so the sharing is the reference's; the iteration order is java.util.HashSet's (JavaHashSet below: hash at insertion,
bucket order, resizes), as oracle/jdk.hpp restates it. A bucket that would become a red-black tree (nine elements in a table of 64
or more — the single-best strategy lets the sets grow) is refused loudly: its order follows the tree's shape and
System.identityHashCode. Host-side Python stand-in for the Java classes (no JVM in the build image); names follow the Java source."""
from wittgenstein_amd.core import IllegalArgumentException, IllegalStateException
from wittgenstein_amd.hostnet import Message
from .p2p import P2PNetwork, P2PNode

_M64 = (1 << 64) - 1


class BitSet:
    """java.util.BitSet as far as P2PHandel uses it: a mutable object (identity matters: messages share it)"""
    __slots__ = ("v",)

    def __init__(self, v=0):
        self.v = v

    def get(self, i):
        return (self.v >> i) & 1 == 1

    def set(self, i):
        self.v |= 1 << i

    def or_(self, o):
        self.v |= o.v

    def andNot(self, o):
        self.v &= ~o.v

    def clone(self):
        return BitSet(self.v)

    def cardinality(self):
        return bin(self.v).count("1")

    def length(self):
        return self.v.bit_length()

    def equals(self, o):
        return self.v == o.v

    def hashCode(self):  # h = 1234; for (i = wordsInUse; --i >= 0;) h ^= words[i] * (i + 1); return (int) ((h >> 32) ^ h)
        h, v, i = 1234, self.v, 0
        while v:
            h ^= ((v & _M64) * (i + 1)) & _M64
            v >>= 64
            i += 1
        return ((h >> 32) ^ h) & 0xFFFFFFFF  # (the int's bits; callers only mix and mask them)


class JavaHashSet:
    """java.util.HashSet (a HashMap, JDK 8+) where the iteration order matters — oracle/jdk.hpp JHashSet, line for line"""

    def __init__(self):
        self.tab, self.size = None, 0

    @staticmethod
    def _spread(h):
        return (h ^ (h >> 16)) & 0xFFFFFFFF

    def _resize(self):
        if self.tab is None:
            self.tab = [[] for _ in range(16)]
            return
        old = len(self.tab)
        nt = [[] for _ in range(2 * old)]
        for j, b in enumerate(self.tab):
            for e in b:
                nt[j + old if e[0] & old else j].append(e)
        self.tab = nt

    def isEmpty(self):
        return self.size == 0

    def add(self, k):  # HashMap.putVal
        if self.tab is None:
            self._resize()
        h = self._spread(k.hashCode())
        b = self.tab[h & (len(self.tab) - 1)]
        for eh, ek in b:
            if eh == h and (ek is k or k.equals(ek)):
                return False
        before = len(b)
        b.append((h, k))
        if before >= 8:  # treeifyBin
            if len(self.tab) >= 64:
                raise IllegalStateException("JavaHashSet: a bucket became a tree (not restated)")
            self._resize()
        self.size += 1
        if self.size > len(self.tab) * 3 // 4:
            self._resize()
        return True

    def remove(self, k):  # by the element's hash NOW
        if self.tab is None:
            return False
        h = self._spread(k.hashCode())
        b = self.tab[h & (len(self.tab) - 1)]
        for i, (eh, ek) in enumerate(b):
            if eh == h and (ek is k or k.equals(ek)):
                del b[i]
                self.size -= 1
                return True
        return False

    def removeNode(self, k):  # Iterator.remove() of the element just returned
        for b in self.tab or ():
            for i, (_, ek) in enumerate(b):
                if ek is k:
                    del b[i]
                    self.size -= 1
                    return

    def clear(self):  # (the capacity stays)
        for b in self.tab or ():
            del b[:]
        self.size = 0

    def items(self):
        return [ek for b in (self.tab or ()) for _, ek in b]

    def capacity(self):
        return len(self.tab) if self.tab else 0


class P2PHandelParameters:  # :36-109
    def __init__(self, signingNodeCount=100, relayingNodeCount=20, threshold=99, connectionCount=40, pairingTime=100,
                 sigsSendPeriod=1000, doubleAggregateStrategy=True, sendSigsStrategy="dif", sendState=False, nodeBuilderName=None,
                 networkLatencyName=None):
        if sendSigsStrategy not in ("all", "dif", "cmp_all", "cmp_diff"):  # enum SendSigsStrategy :25-30
            raise IllegalArgumentException("sendSigsStrategy")
        self.signingNodeCount, self.relayingNodeCount, self.threshold = signingNodeCount, relayingNodeCount, threshold
        self.connectionCount, self.pairingTime, self.sigsSendPeriod = connectionCount, pairingTime, sigsSendPeriod
        self.doubleAggregateStrategy, self.sendSigsStrategy, self.sendState = doubleAggregateStrategy, sendSigsStrategy, sendState
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName


class State(Message):  # :119-140
    def __init__(self, who):
        self.desc = who.verifiedSignatures.clone()
        self.who = who

    def size(self):
        return max(1, self.desc.length() // 8)

    def action(self, network, frm, to):
        to.onPeerState(self)


class SendSigs(Message):  # :231-253
    def __init__(self, sigs, sigCount=None):
        self.sigs = sigs.clone()
        self._size = max(1, sigs.cardinality() if sigCount is None else sigCount)

    def size(self):
        return self._size

    def action(self, network, frm, to):
        to.onNewSig(frm, self.sigs)


class P2PHandelNode(P2PNode):  # :255-481
    def __init__(self, p, justRelay):
        super().__init__(p.network)
        self.p = p
        self.verifiedSignatures = BitSet()
        self.toVerify = JavaHashSet()
        self.peersState = {}
        self.justRelay = justRelay
        if not justRelay:
            self.verifiedSignatures.set(self.nodeId)

    def start(self):  # :270-275
        super().start()
        for q in self.peers:
            self.peersState[q.nodeId] = BitSet()

    def onPeerState(self, state):  # :281-283
        self.peersState[state.who.nodeId].or_(state.desc)

    def updateVerifiedSignatures(self, sigs):  # :290-303
        oldCard = self.verifiedSignatures.cardinality()
        self.verifiedSignatures.or_(sigs)
        newCard = self.verifiedSignatures.cardinality()
        if newCard > oldCard:
            if self.doneAt == 0 and self.verifiedSignatures.cardinality() >= self.p.params.threshold:
                self.doneAt = self.p.network.time
                self.sendFinalSigToPeers()
            elif self.doneAt == 0 and self.p.params.sendState:
                self.sendStateToPeers()

    def sendFinalSigToPeers(self):  # :305-317
        dest = []
        for q in self.peers:
            ps = self.peersState[q.nodeId]
            if ps.cardinality() < self.p.params.threshold:
                dest.append(q)
                ps.or_(self.verifiedSignatures)
        self.p.network.send(SendSigs(self.verifiedSignatures, 1), self, dest)

    def sendStateToPeers(self):  # :319-322
        self.p.network.send(State(self), self, list(self.peers))

    def onNewSig(self, frm, sigs):  # :325-328
        self.peersState[frm.nodeId].or_(sigs)
        self.toVerify.add(sigs)

    def sendSigs(self):  # :336-354
        if self.doneAt > 0:
            return
        dest = self.bestDest()
        if dest is None:
            return
        toSend = self.diff(dest)
        self.peersState[dest.nodeId].or_(self.verifiedSignatures)
        self.p.network.send(self.createSendSigs(toSend), self, dest)

    def diff(self, peer):  # :356-360
        needed = self.verifiedSignatures.clone()
        needed.andNot(self.peersState[peer.nodeId])
        return needed

    def bestDest(self):  # :367-378
        dest, destSize = None, 0
        for q in self.peers:
            size = self.diff(q).cardinality()
            if size > destSize:
                dest, destSize = q, size
        return dest

    def createSendSigs(self, toSend):  # :389-404
        st, vs, p = self.p.params.sendSigsStrategy, self.verifiedSignatures, self.p
        if st == "dif":
            return SendSigs(toSend)
        if st == "cmp_all":
            return SendSigs(vs, p.compressedSize(vs))
        if st == "cmp_diff":
            return SendSigs(vs, min(p.compressedSize(vs), p.compressedSize(toSend)))
        return SendSigs(vs)

    def _registerUpdate(self, tBest):
        net = self.p.network
        net.registerTask(lambda: self.updateVerifiedSignatures(tBest), net.time + self.p.params.pairingTime * 2, self)

    def checkSigs(self):  # :406-412
        if self.p.params.doubleAggregateStrategy:
            self.checkSigs2()
        else:
            self.checkSigs1()

    def checkSigs1(self):  # :419-449
        best, bestV = None, 0
        for o1 in self.toVerify.items():
            oo1 = o1.clone()
            oo1.andNot(self.verifiedSignatures)
            v1 = oo1.cardinality()
            if v1 == 0:
                self.toVerify.removeNode(o1)
            elif v1 > bestV:
                bestV, best = v1, o1
        if best is not None:
            self.toVerify.remove(best)
            self._registerUpdate(best)

    def checkSigs2(self):  # :455-480
        agg = None
        for o1 in self.toVerify.items():
            if agg is None:
                agg = o1
            else:
                agg.or_(o1)
        self.toVerify.clear()
        if agg is not None:
            oo1 = agg.clone()
            oo1.andNot(self.verifiedSignatures)
            if oo1.cardinality() > 0:
                self._registerUpdate(agg)


class P2PHandel:
    def __init__(self, params=None, config=None):  # :111-117
        self.params = params or P2PHandelParameters()
        self._config = config
        self.network = P2PNetwork(self.params.connectionCount, False, self.params.networkLatencyName, config)

    def copy(self):
        return P2PHandel(self.params, self._config)

    # ---- compressedSize :160-202 / mergeRanges :204-229
    def compressedSize(self, sigs):
        if sigs.length() == self.params.signingNodeCount:
            return 1
        firstOneAt, sigCt, pos = -1, 0, -1
        compressing = wasCompressing = False
        while True:
            pos += 1
            if not pos <= sigs.length() + 1:
                break
            if not sigs.get(pos):
                compressing = False
                sigCt -= self.mergeRanges(firstOneAt, pos)
                firstOneAt = -1
            elif compressing:
                if (pos + 1) % 2 == 0:
                    compressing = False
                    wasCompressing = True
            else:
                sigCt += 1
                if pos % 2 == 0:
                    compressing = True
                    if not wasCompressing:
                        firstOneAt = pos
                    else:
                        wasCompressing = False
        return sigCt

    def mergeRanges(self, firstOneAt, pos):
        if firstOneAt < 0:
            return 0
        if firstOneAt % 4 != 0:
            firstOneAt += 4 - (firstOneAt % 4)
        rangeCt = (pos - firstOneAt) // 2 if pos >= firstOneAt else -((firstOneAt - pos) // 2)  # Java's / truncates toward zero
        if rangeCt < 2:
            return 0
        mx = rangeCt.bit_length() - 1  # MoreMath.log2
        while mx > 0:
            sizeInBlocks = 1 << mx
            size = sizeInBlocks * 2
            if firstOneAt % size == 0:
                return (sizeInBlocks - 1) + self.mergeRanges(firstOneAt + size, pos)
            mx -= 1
        return 0

    def init(self):  # :483-510
        # (the reference registers node i's tasks right after addNode(i); the engine takes its node table whole, so they are
        # registered behind the loop, node by node in the same order — registering draws nothing and node construction does not
        # look at the queue, so the rd sequence and the bucket's push order are the reference's)
        net, params = self.network, self.params
        total = params.signingNodeCount + params.relayingNodeCount
        justRelay = set()
        while len(justRelay) < params.relayingNodeCount:
            justRelay.add(net.rd.nextInt(total))
        nodes = [P2PHandelNode(self, i in justRelay) for i in range(total)]
        for n in nodes:
            net.addNode(n)
        for n in nodes:
            if params.sendState:
                net.registerTask(n.sendStateToPeers, 1, n)
            net.registerPeriodicTask(n.sendSigs, 1, params.sigsSendPeriod, n)
            net.registerConditionalTask(n.checkSigs, 1, params.pairingTime, n, lambda n=n: not n.toVerify.isEmpty(),
                                        lambda n=n: n.doneAt == 0)
        net.setPeers()
