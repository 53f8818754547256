"""protocols.Dfinity (P/Dfinity.java, with C/Block.java, C/BlockChainNode.java, C/BlockChainNetwork.java) written against the
reference's own protocol API and run on the engine in host-callback mode (wittgenstein_amd.hostnet): the block tree, the votes and
the beacon exchanges stay host objects as in the reference, while every envelope — the beacons' and blocks' `sendAll`, the
proposals / votes / exchanges sent to a shuffled committee (C/Network.java:418-447) — its latency sampling, the ordering and the
shared `rd` (which shuffles every committee before a send) live in libwittgpu.so on the MI355X. Host-side Python stand-in for the
Java classes (no JVM in the build image); class, field and method names follow the Java source. The block id is a
per-protocol counter (a JVM-wide static in the reference, C/Block.java:11) and the three node lists of DfinityParameters (:30-32)
belong to the protocol instance (the reference's copies share and grow them) — as oracle/dfinity.hpp."""
from wittgenstein_amd.core import IllegalArgumentException
from wittgenstein_amd.hostnet import HostNetwork, Message, Node


def shuffle(lst, rd):  # java.util.Collections.shuffle(list, rnd)
    for i in range(len(lst), 1, -1):
        j = rd.nextInt(i)
        lst[i - 1], lst[j] = lst[j], lst[i - 1]


class DfinityParameters:  # :14-71
    roundTime = 3000
    blockProducersPerRound = 5

    def __init__(self, blockProducersCount=10, attestersCount=10, attestersPerRound=10, blockConstructionTime=1,
                 attestationConstructionTime=1, percentageDeadAttester=0, nodeBuilderName=None, networkLatencyName=None):
        self.blockProducersCount = blockProducersCount
        self.blockProducersRound = blockProducersCount // self.blockProducersPerRound
        self.attestersRound = attestersCount // attestersPerRound
        self.attestersCount, self.attestersPerRound = attestersCount, attestersPerRound
        self.randomBeaconCount = attestersPerRound
        self.majority = attestersPerRound // 2 + 1
        self.blockConstructionTime, self.attestationConstructionTime = blockConstructionTime, attestationConstructionTime
        self.percentageDeadAttester = percentageDeadAttester
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName


class DfinityBlock:  # C/Block.java:4-117 + :92-105
    def __init__(self, protocol=None, producer=None, height=0, parent=None, valid=True, time=0):
        if protocol is None:  # createGenesis(): Block(0)
            self.height = self.proposalTime = self.lastTxId = self.id = 0
            self.parent = self.producer = None
            self.valid = True
            return
        if height <= 0:
            raise IllegalArgumentException("Only the genesis block has a special height")
        if parent is not None and time < parent.proposalTime:
            raise IllegalArgumentException("bad time: parent is (%r), our time:%d" % (parent, time))
        if parent is not None and parent.height >= height:
            raise IllegalArgumentException("Bad parent")
        self.producer, self.height, self.parent, self.valid = producer, height, parent, valid
        self.id = protocol._next_block_id
        protocol._next_block_id += 1
        self.lastTxId = self.proposalTime = time

    def hasDirectLink(self, b):  # C/Block.java:86-99
        if b is self:
            return True
        if b.height == self.height:
            return False
        older, young = (self, b) if self.height > b.height else (b, self)
        while older.height > young.height:
            older = older.parent
        return older is young


def compare(o1, o2):  # DfinityBlockComparator :107-130
    if o1 is o2:
        return 0
    if not o2.valid:
        return 1
    if not o1.valid:
        return -1
    if o1.hasDirectLink(o2):
        return -1 if o1.height < o2.height else 1
    if o1.height != o2.height:
        return -1 if o1.height < o2.height else 1
    return 0  # Long.compare(o1.producer.nodeId, o1.producer.nodeId) — o1 against itself (:128)


class BlockProposal(Message):  # :132-144
    def __init__(self, block):
        self.block = block

    def action(self, network, frm, to):
        to.onProposal(self.block)


class Vote(Message):  # :146-157
    def __init__(self, voteFor):
        self.voteFor = voteFor

    def action(self, network, frm, to):
        to.onVote(frm, self.voteFor)


class RandomBeaconExchange(Message):  # :159-171
    def __init__(self, height):
        self.height = height

    def action(self, network, frm, to):
        to.onRandomBeaconExchange(frm, self.height)


class RandomBeaconResult(Message):  # :173-186
    def __init__(self, height, rd):
        self.height, self.rd = height, rd

    def action(self, network, frm, to):
        to.onRandomBeacon(self.height, self.rd)


class SendBlock(Message):  # C/BlockChainNetwork.java:22-38
    def __init__(self, toSend):
        self.toSend = toSend

    def action(self, network, frm, to):
        to.onBlock(self.toSend)


class DfinityNode(Node):  # C/BlockChainNode.java:6-75 + :188-213
    def __init__(self, p):
        super().__init__(p.network)
        self.p = p
        self.genesis = self.head = p.genesis
        self.blocksReceivedByBlockId = {p.genesis.id: p.genesis}
        self.committeeMajorityBlocks, self.committeeMajorityHeight = set(), set()
        self.lastRandomBeacon = 0

    def best(self, o1, o2):  # :194-196
        return o1 if compare(o1, o2) >= 0 else o2

    def _baseOnBlock(self, b):  # BlockChainNode.onBlock C/BlockChainNode.java:29-47
        if not b.valid:
            return False
        if b.id in self.blocksReceivedByBlockId:
            return False
        self.blocksReceivedByBlockId[b.id] = b
        self.head = self.best(self.head, b)
        return True

    def onBlock(self, b):
        return self._baseOnBlock(b)

    def onVote(self, voter, voteFor):  # :202
        pass

    def onRandomBeacon(self, height, rd):  # :205-210
        if self.lastRandomBeacon < height:
            self.lastRandomBeacon = height
            self.onRandomBeaconOnce(height, rd)

    def onRandomBeaconOnce(self, height, rd):  # :212
        pass


class BlockProducerNode(DfinityNode):  # :215-263
    def __init__(self, myRound, p):
        super().__init__(p)
        self.myRound = myRound
        self.waitForBlockHeight = -1

    def createProposal(self, height):  # :225-240
        if self.head.height != height - 1:
            raise IllegalArgumentException()
        net, params = self.p.network, self.p.params
        newBlock = DfinityBlock(self.p, self, height, self.head, True, net.time)
        attestersS = list(self.p.attesters)
        shuffle(attestersS, net.rd)
        net.send(BlockProposal(newBlock), self, attestersS, net.time + params.blockConstructionTime, _force_multi=True)
        self.waitForBlockHeight = -1

    def onBlock(self, b):  # :243-253
        if not self._baseOnBlock(b):
            return False
        if self.head.height == self.waitForBlockHeight:
            self.createProposal(self.waitForBlockHeight + 1)
        return True

    def onRandomBeaconOnce(self, h, rd):  # :256-262
        if rd % self.p.params.blockProducersRound == self.myRound:
            if self.head.height == h - 1:
                self.createProposal(h)


class AttesterNode(DfinityNode):  # :265-351
    def __init__(self, myRound, p):
        super().__init__(p)
        self.votes = {}
        self.proposals = []
        self.myRound = myRound
        self.voteForHeight = -1

    def _voteTo(self, b):  # :309-313, :341-345
        net = self.p.network
        attestersS = list(self.p.attesters)
        shuffle(attestersS, net.rd)
        net.send(Vote(b), self, attestersS, net.time + self.p.params.attestationConstructionTime, _force_multi=True)

    def onVote(self, voter, voteFor):  # :277-284
        voters = self.votes.setdefault(voteFor.id, set())
        if self.voteForHeight == voteFor.height:
            if voter.nodeId not in voters:
                voters.add(voter.nodeId)
                if len(voters) >= self.p.params.majority:
                    self.sendBlock(voteFor)

    def sendBlock(self, voteFor):  # :286-292
        self.committeeMajorityBlocks.add(voteFor.id)
        self.committeeMajorityHeight.add(voteFor.height)
        self.voteForHeight = -1
        self.p.network.sendAll(SendBlock(voteFor), self)

    def onProposal(self, b):  # :298-318
        if self.voteForHeight == b.height:
            voters = self.votes.setdefault(b.id, set())
            if self.nodeId not in voters:
                voters.add(self.nodeId)
                if len(voters) >= self.p.params.majority:
                    self.sendBlock(b)
                else:
                    self._voteTo(b)
        elif b.height > self.head.height:
            self.proposals.append(b)

    def onBlock(self, b):  # :321-332
        if not self._baseOnBlock(b):
            return False
        self.committeeMajorityBlocks.add(b.id)
        self.committeeMajorityHeight.add(b.height)
        if self.voteForHeight == b.height:
            self.voteForHeight = -1
        return True

    def onRandomBeaconOnce(self, h, rd):  # :335-350
        if rd % self.p.params.attestersRound == self.myRound and h not in self.committeeMajorityHeight:
            self.voteForHeight = h
            sent = set()
            for b in self.proposals:
                if b.height == h and b.id not in sent:
                    sent.add(b.id)
                    self._voteTo(b)
            self.proposals.clear()


class RandomBeaconNode(DfinityNode):  # :353-424
    def __init__(self, p):
        super().__init__(p)
        self.rd = 0
        self.height = 1
        self.lastRDSent = 0
        self.exchanged = {}

    def onRandomBeaconExchange(self, frm, height):  # :367-374
        if height >= self.height and height > self.lastRDSent:
            voters = self.exchanged.setdefault(height, set())
            if frm.nodeId not in voters:
                voters.add(frm.nodeId)
                if height == self.height and len(voters) >= self.p.params.majority:
                    self.sendRB()

    def sendRB(self):  # :376-381
        net = self.p.network
        self.rd = self.height
        self.lastRDSent = self.height
        net.sendAll(RandomBeaconResult(self.height, self.rd), self, net.time + self.p.params.attestationConstructionTime)

    def onBlock(self, b):  # :387-410
        if not self._baseOnBlock(b):
            return True
        net, params = self.p.network, self.p.params
        if self.head.height == self.height:
            self.height += 1
            voters = self.exchanged.setdefault(self.height, set())
            added = self.nodeId not in voters
            voters.add(self.nodeId)
            if added and len(voters) >= params.majority:
                self.sendRB()
            else:
                wt = self.head.parent.proposalTime + params.roundTime * 2
                if wt <= net.time:
                    wt = net.time + params.attestationConstructionTime
                rdsSends = list(self.p.rds)
                shuffle(rdsSends, net.rd)
                net.send(RandomBeaconExchange(self.height), self, rdsSends, wt, _force_multi=True)
        return False

    def onRandomBeaconOnce(self, h, rd):  # :417-423
        if h > self.height:
            self.lastRDSent = self.height
            self.height = h
            self.rd = rd


class Dfinity:
    def __init__(self, params=None, config=None):  # :86-90
        self.params = params or DfinityParameters()
        self._config = config
        self.network = HostNetwork(self.params.networkLatencyName, config)
        self.genesis = DfinityBlock()
        self._next_block_id = 1
        self.attesters, self.bps, self.rds = [], [], []
        self.observer = DfinityNode(self)  # (built — and its position drawn — by the constructor, before any rd.setSeed)
        self.network.addNode(self.observer)  # BlockChainNetwork.addObserver C/BlockChainNetwork.java:15-18

    def copy(self):
        return Dfinity(self.params, self._config)

    def init(self):  # :426-450
        net, params = self.network, self.params
        for i in range(params.attestersCount):
            n = AttesterNode(i % params.attestersRound, self)
            self.attesters.append(n)
            net.addNode(n)
        for i in range(params.blockProducersCount):
            n = BlockProducerNode(i % params.blockProducersRound, self)
            self.bps.append(n)
            net.addNode(n)
        for i in range(params.randomBeaconCount):
            n = RandomBeaconNode(self)
            self.rds.append(n)
            net.addNode(n)
        shuffle(self.bps, net.rd)
        for n in self.rds:
            n.sendRB()
