"""Casper IMD (P/CasperIMD.java, with C/Block.java, C/BlockChainNode.java, C/BlockChainNetwork.java) written against
the reference's own protocol API — Network.sendAll / registerTask / registerPeriodicTask, Message.action(), Node — and
run on the engine in host-callback mode (wittgenstein_amd.hostnet): the block tree, the attestation sets and the fork
choice stay host objects exactly as in the reference, while every envelope — the N-destination `sendAll` of each block
and attestation (C/Network.java:341-347, C/Envelope.java:57-155), the periodic and one-shot tasks — its latency
sampling, the LIFO / chain ordering and the shared `rd` live in libwittgpu.so on the MI355X.

This is the host-side Python stand-in for the Java class (no JVM in the build image): class, field
and method names follow the Java source so that it reads side by side with it. `for (CasperBlock b :
blocksToReevaluate)` (:352-356) iterates a HashSet in identity-hash order, which the JDK leaves unspecified; like the
oracle (oracle/casper.hpp) this iterates in ascending block id. ByzBlockProducer / SF / NS (:545-633) are mirrored too (their HashSet pick = the smallest block id, as the oracle)
(unused by init() and by every reference test)."""
from wittgenstein_amd.core import IllegalArgumentException, IllegalStateException
from wittgenstein_amd.hostnet import HostNetwork, Message, Node

SLOT_DURATION = 8000  # :19


class CasperParemeters:  # (sic) :18-70
    def __init__(self, cycleLength=4, randomOnTies=True, blockProducersCount=2, attestersPerRound=20,
                 blockConstructionTime=1000, attestationConstructionTime=1, nodeBuilderName=None,
                 networkLatencyName=None):
        if nodeBuilderName not in (None, "", "RANDOM_SPEED=CONSTANT_TOR=0.00"):
            raise IllegalArgumentException("hostnet.Node builds RANDOM / constant-speed nodes only")
        self.cycleLength, self.randomOnTies, self.blockProducersCount = cycleLength, randomOnTies, blockProducersCount
        self.attestersPerRound, self.attestersCount = attestersPerRound, attestersPerRound * cycleLength
        self.blockConstructionTime, self.attestationConstructionTime = blockConstructionTime, attestationConstructionTime
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName


class CasperBlock:  # C/Block.java:4-117 + :129-175
    def __init__(self, protocol=None, producer=None, height=0, father=None, attestationsByHeight=None, time=0):
        if protocol is None:  # genesis: Block(0)
            self.height, self.proposalTime, self.lastTxId, self.id = 0, 0, 0, 0
            self.parent, self.producer, self.valid, self.attestationsByHeight = None, None, True, {}
            return
        if height <= 0:
            raise IllegalArgumentException("Only the genesis block has a special height")
        if father is not None and time < father.proposalTime:
            raise IllegalArgumentException("bad time: parent is (%r), our time:%d" % (father, time))
        if father is not None and father.height >= height:
            raise IllegalArgumentException("Bad parent")
        self.producer, self.height, self.parent, self.valid = producer, height, father, True
        self.id = protocol._next_block_id  # Block.blockId++ (a JVM-wide static there; only the order matters)
        protocol._next_block_id += 1
        self.lastTxId = self.proposalTime = time
        self.attestationsByHeight = attestationsByHeight

    def hasDirectLink(self, b):  # C/Block.java:86-99
        if b is self:
            return True
        if b.height == self.height:
            return False
        older, young = (self, b) if self.height > b.height else (b, self)
        while older.height > young.height:
            older = older.parent
        return older is young


class Attestation(Message):  # :98-127
    def __init__(self, protocol, attester, height):
        self.attester, self.height, self.head = attester, height, attester.head
        self.hs = set()
        cur = attester.head.parent
        while cur is not None and cur.height >= attester.head.height - protocol.params.cycleLength:
            self.hs.add(cur.id)
            cur = cur.parent

    def action(self, network, frm, to):
        to.onAttestation(self)

    def attests(self, cb):
        return cb.id in self.hs


class SendBlock(Message):  # C/BlockChainNetwork.java:22-38
    def __init__(self, toSend):
        self.toSend = toSend

    def action(self, network, frm, to):
        to.onBlock(self.toSend)


class CasperNode(Node):  # C/BlockChainNode.java:6-75 + :177-368
    def __init__(self, protocol, byzantine=False):
        super().__init__(protocol.network)
        self.p, self.network, self.byzantine = protocol, protocol.network, byzantine
        self.genesis = self.head = protocol.genesis
        self.blocksReceivedByBlockId = {protocol.genesis.id: protocol.genesis}
        self.blocksReceivedByFatherId, self.blocksReceivedByHeight = {}, {}
        self.attestationsByHead = {}
        self.blocksToReevaluate = set()

    def periodicTask(self):
        return None

    def _baseOnBlock(self, b):  # BlockChainNode.onBlock :29-47
        if not b.valid:
            return False
        if b.id in self.blocksReceivedByBlockId:
            return False
        self.blocksReceivedByBlockId[b.id] = b
        self.blocksReceivedByFatherId.setdefault(b.parent.id, set()).add(b)
        self.blocksReceivedByHeight.setdefault(b.height, set()).add(b)
        self.head = self.best(self.head, b)
        return True

    def best(self, o1, o2):  # :186-236
        if o1 is o2:
            return o1
        if o1.height == o2.height:
            raise IllegalStateException("two blocks for the same height")
        if o1.hasDirectLink(o2):
            return o2 if o1.height < o2.height else o1
        b1, b2 = o1, o2
        while b1.parent is not b2.parent:
            if b1.parent.height > b2.parent.height:
                b1 = b1.parent
            else:
                b2 = b2.parent
        h = b1.parent
        b1Votes, b2Votes = self.countAttestations(o1, h), self.countAttestations(o2, h)
        if b1Votes > b2Votes:
            return o1
        if b1Votes < b2Votes:
            return o2
        if self.p.params.randomOnTies:
            return o1 if self.network.rd.nextBoolean() else o2
        return o1 if b1.id >= b2.id else o2

    def countAttestations(self, start, h):  # :241-266
        a1 = set()
        cur = start
        while cur is not h:
            for i in range(cur.height - 1, h.height, -1):
                for a in cur.attestationsByHeight.get(i, ()):
                    if a.attests(h):
                        a1.add(a)
            for a in self.attestationsByHead.get(cur.id, ()):
                if a.attests(h):
                    a1.add(a)
            cur = cur.parent
        return len(a1)

    def onBlock(self, b):  # :276-292
        delta = self.network.time - self.genesis.proposalTime + b.height * SLOT_DURATION
        if delta >= 0:
            self.blocksToReevaluate.add(self.head)
            self.blocksToReevaluate.add(b)
            return self._baseOnBlock(b)
        self.network.registerTask(lambda: self.onBlock(b), delta * -1, self)
        return False

    def onAttestation(self, a):  # :294-337
        self.attestationsByHead.setdefault(a.head.id, set()).add(a)
        if a.head.id in self.blocksReceivedByBlockId:
            self.blocksToReevaluate.add(a.head)

    def reevaluateHead(self):  # :349-356
        for b in sorted(self.blocksToReevaluate, key=lambda blk: blk.id):
            self.head = self.best(self.head, b)
        self.blocksToReevaluate.clear()


class BlockProducer(CasperNode):  # :370-443
    def periodicTask(self):
        def run():
            self.reevaluateHead()
            self.createAndSendBlock(self.network.time // SLOT_DURATION)
        return run

    def buildBlock(self, base, height):  # :389-434
        cl = self.p.params.cycleLength
        res = {}
        i = height - 1
        while i >= 0 and i >= height - cl:
            res[i] = set()
            i -= 1
        allFromBlocks = set()
        cur = base
        while cur is not self.genesis and cur.height >= height - cl:
            for ats in cur.attestationsByHeight.values():
                allFromBlocks |= ats
            cur = cur.parent
        cur = base
        while cur is not None and cur.height >= height - cl:
            for a in self.attestationsByHead.get(cur.id, ()):
                if a.height < height and a not in allFromBlocks:
                    res.setdefault(a.height, set()).add(a)
            cur = cur.parent
        return CasperBlock(self.p, self, height, base, res, self.network.time)

    def createAndSendBlock(self, height):  # :436-442
        self.head = self.buildBlock(self.head, height)
        self.network.sendAll(SendBlock(self.head), self, self.network.time + self.p.params.blockConstructionTime)


class Attester(CasperNode):  # :445-473
    def periodicTask(self):
        return lambda: self.vote(self.network.time // SLOT_DURATION)

    def vote(self, height):
        self.reevaluateHead()
        v = Attestation(self.p, self, height)
        self.network.sendAll(v, self, self.network.time + self.p.params.attestationConstructionTime)


class ByzBlockProducer(BlockProducer):  # :511-581
    def __init__(self, protocol, delay):
        super().__init__(protocol, True)
        self.toSend, self.h, self.delay = 1, 0, delay
        self.onDirectFather = self.onOlderAncestor = self.incNotTheBestFather = 0

    def reevaluateH(self, time):  # :529-543
        self.reevaluateHead()
        while self.head.height >= self.toSend:
            self.head = self.head.parent
        slotTime = time - self.delay
        self.h = int(slotTime / SLOT_DURATION)  # Java int division truncates toward zero
        if self.h != self.toSend:
            raise IllegalStateException("h=%d, toSend=%d" % (self.h, self.toSend))


class ByzBlockProducerPlain(ByzBlockProducer):  # ByzBlockProducer's own periodicTask, :545-563
    def periodicTask(self):
        def run():
            self.reevaluateH(self.network.time)
            if self.head.height == self.h - 1:
                self.onDirectFather += 1
            else:
                self.onOlderAncestor += 1
                # blocksReceivedByHeight.get(h - 1).iterator().next(): the reference's HashSet order is the JVM's identity
                # hashes; the oracle and this mirror take the smallest block id. No block of that height: its NPE
                possibleFather = min(self.blocksReceivedByHeight[self.h - 1], key=lambda b: b.id)
                if possibleFather.parent.height != self.h - 1:
                    self.incNotTheBestFather += 1
            self.createAndSendBlock(self.toSend)
            self.toSend += self.p.params.blockProducersCount
        return run


class ByzBlockProducerSF(ByzBlockProducer):  # :583-604 — skip its father's block
    def periodicTask(self):
        def run():
            self.reevaluateH(self.network.time)
            if self.head.id != 0 and self.head.height == self.h - 1:
                self.head = self.head.parent
                self.onDirectFather += 1
            else:
                self.onOlderAncestor += 1
            self.createAndSendBlock(self.toSend)
            self.toSend += self.p.params.blockProducersCount
        return run


class ByzBlockProducerNS(ByzBlockProducer):  # :610-633 — skip the father if the father skipped the grand father
    def __init__(self, protocol, delay):
        super().__init__(protocol, delay)
        self.skipped = 0

    def periodicTask(self):
        def run():
            self.reevaluateH(self.network.time)
            if self.head.id != 0 and self.head.height == self.h - 1 and self.head.parent.height == self.h - 3:
                b = min(self.blocksReceivedByHeight[self.h - 2], key=lambda blk: blk.id)
                self.head = b
                self.skipped += 1
            self.createAndSendBlock(self.toSend)
            self.toSend += self.p.params.blockProducersCount
        return run


class ByzBlockProducerWF(ByzBlockProducer):  # :635-692
    def __init__(self, protocol, delay):
        super().__init__(protocol, delay)
        self.late = self.onTime = 0

    def periodicTask(self):
        def run():
            if self.head is self.genesis and self.toSend == 1:
                self.reevaluateH(self.network.time)
                self.createAndSendBlock(self.h)
                self.toSend += self.p.params.blockProducersCount
        return run

    def onBlock(self, b):
        if not super().onBlock(b):
            return False
        if b.height == self.toSend - 1:
            perfectDate = SLOT_DURATION * self.toSend + self.delay
            th = self.toSend

            def r():
                self.head = self.buildBlock(b, th)
                self.network.sendAll(SendBlock(self.head), self,
                                     self.network.time + self.p.params.blockConstructionTime)
            self.toSend += self.p.params.blockProducersCount
            if self.network.time >= perfectDate:
                r()
                self.late += 1
            else:
                self.network.registerTask(r, perfectDate, self)
                self.onTime += 1
        return True


class CasperIMD:  # :14-96, :475-509
    def __init__(self, params=None, config=None):
        self.params = params or CasperParemeters()
        self.network = HostNetwork(self.params.networkLatencyName, config)
        self._next_block_id = 1
        self.genesis = CasperBlock()
        self.attesters, self.bps = [], []
        self.observer = CasperNode(self)          # network.addObserver(new CasperNode(false, genesis) {})  :86
        self.network.addNode(self.observer)

    def copy(self):
        return CasperIMD(self.params)

    def init(self, byzantineNode=None):
        p, net = self.params, self.network
        if byzantineNode is None:
            byzantineNode = ByzBlockProducerWF(self, 0)
        self.bps.append(byzantineNode)
        net.addNode(byzantineNode)
        periodic = [(byzantineNode, SLOT_DURATION + byzantineNode.delay, SLOT_DURATION * p.blockProducersCount)]
        for i in range(1, p.blockProducersCount):
            n = BlockProducer(self)
            self.bps.append(n)
            net.addNode(n)
            periodic.append((n, SLOT_DURATION * (i + 1), SLOT_DURATION * p.blockProducersCount))
        for i in range(p.attestersCount):
            n = Attester(self)
            self.attesters.append(n)
            net.addNode(n)
            periodic.append((n, SLOT_DURATION * (1 + i % p.cycleLength) + 4000, SLOT_DURATION * p.cycleLength))
        # (the engine learns the node set at the first registration, so the registrations of :484-508 follow the
        # node constructions; their relative order — the push order of the task envelopes — is the reference's)
        for n, startAt, period in periodic:
            net.registerPeriodicTask(n.periodicTask(), startAt, period, n)
