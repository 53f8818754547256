"""protocols.Paxos (P/Paxos.java) written against the reference's own protocol API and run on the engine in host-callback mode
(wittgenstein_amd.hostnet): Propose / Agree / Reject / Commit / Accept / RejectOnCommit, their latency sampling and ordering,
the timeout tasks and the shared `rd` (which shuffles the acceptors before every multi-destination send, :299-303) live in
libwittgpu.so on the MI355X; the acceptors' and proposers' state stays host objects as in the reference. init() sends between node
constructions (:374-387): HostNetwork.deferred_init keeps the reference's rd order. Host-side Python stand-in for the Java classes
(no JVM in the build image); class, field and method names follow the Java source; a null Integer is None."""
from wittgenstein_amd.core import IllegalStateException
from wittgenstein_amd.hostnet import HostNetwork, Message, Node

MAX_VAL = 1000  # :24


def shuffle(lst, rd):  # java.util.Collections.shuffle(list, rnd)
    for i in range(len(lst), 1, -1):
        j = rd.nextInt(i)
        lst[i - 1], lst[j] = lst[j], lst[i - 1]


class PaxosParameters:  # :352-371
    def __init__(self, acceptorCount=3, proposerCount=3, timeout=1000, nodeBuilder=None, latency=None):
        self.acceptorCount, self.proposerCount, self.timeout = acceptorCount, proposerCount, timeout
        self.nodeBuilder, self.latency = nodeBuilder, latency


class Propose(Message):  # :43-54
    def __init__(self, seq):
        self.seq = seq

    def action(self, network, frm, to):
        to.onPropose(frm, self)


class Reject(Message):  # :56-75
    def __init__(self, seqRejected, seqAccepted):
        self.seqRejected, self.seqAccepted = seqRejected, seqAccepted

    def action(self, network, frm, to):
        to.onReject(self.seqRejected, self.seqAccepted)


class Agree(Message):  # :77-96
    def __init__(self, yourSeq, acceptedSeq, acceptedVal):
        self.yourSeq, self.acceptedSeq, self.acceptedVal = yourSeq, acceptedSeq, acceptedVal

    def action(self, network, frm, to):
        to.onAgree(self.yourSeq, self.acceptedSeq, self.acceptedVal)


class Commit(Message):  # :98-115
    def __init__(self, seq, val):
        self.seq, self.val = seq, val

    def action(self, network, frm, to):
        to.onCommit(frm, self.seq, self.val)


class Accept(Message):  # :117-129
    def __init__(self, yourSeq):
        self.yourSeq = yourSeq

    def action(self, network, frm, to):
        to.onAccept(self.yourSeq)


class RejectOnCommit(Message):  # :132-145
    def __init__(self, seqRejected, seqAccepted):
        self.seqRejected, self.seqAccepted = seqRejected, seqAccepted

    def action(self, network, frm, to):
        to.onRejectOnCommit(self.seqRejected, self.seqAccepted)


class PaxosNode(Node):  # :147-151
    def __init__(self, p):
        super().__init__(p.network)
        self.p = p


class AcceptorNode(PaxosNode):  # :153-207
    def __init__(self, p):
        super().__init__(p)
        self.maxAgreed = -1
        self.acceptedSeq = self.acceptedVal = self.agreedTo = None

    def onPropose(self, frm, p):  # :163-177
        net = self.p.network
        if p.seq < self.maxAgreed:
            net.send(Reject(p.seq, self.maxAgreed), self, frm)
        elif p.seq == self.maxAgreed:
            raise IllegalStateException("%r %r" % (self, p))
        else:
            a = Agree(p.seq, self.acceptedSeq, self.acceptedVal)
            self.maxAgreed = p.seq
            self.agreedTo = frm
            net.send(a, self, frm)

    def onCommit(self, frm, seq, val):  # :179-190
        net = self.p.network
        if seq != self.maxAgreed or (self.acceptedVal is not None and self.acceptedVal != val):
            net.send(RejectOnCommit(seq, self.maxAgreed), self, frm)
        else:
            self.acceptedVal = val
            self.acceptedSeq = seq if self.acceptedSeq is None else max(self.acceptedSeq, seq)
            net.send(Accept(seq), self, frm)


class ProposerNode(PaxosNode):  # :209-339
    def __init__(self, rank, p):
        super().__init__(p)
        self.rank = rank
        self.valueProposed = p.network.rd.nextInt(MAX_VAL)
        self.valueAccepted = self.acceptedSeqIP = self.acceptedValIP = None
        self.seqIP = self.agreeCountIP = self.reject1CountIP = self.acceptCountIP = self.reject2CountIP = 0
        self.proposalIP = False
        self.seqAccepted = self.agreeCount = self.reject1Count = self.reject2Count = self.timeoutCount = 0

    def onReject(self, seq, serverCurSeq):  # :238-248
        if seq == self.seqIP:
            self.reject1CountIP += 1
            if self.reject1CountIP == self.p.majority:
                self.proposalIP = False
                self.seqAccepted = max(self.seqAccepted, serverCurSeq)
                self.reject1Count += 1
                self.startNextProposal()

    def onAgree(self, seq, acceptedSeq, acceptedVal):  # :250-268
        if seq == self.seqIP and self.agreeCountIP < self.p.majority:
            self.agreeCountIP += 1
            if acceptedSeq is not None:
                if self.acceptedSeqIP is None or self.acceptedSeqIP < acceptedSeq:
                    self.acceptedSeqIP = acceptedSeq
                    self.acceptedValIP = acceptedVal
            if self.agreeCountIP >= self.p.majority:
                self.agreeCount += 1
                if self.acceptedValIP is None:
                    self.acceptedValIP = self.valueProposed
                self.sendToAcceptors(Commit(self.seqIP, self.acceptedValIP), self.p.network.time + 1)

    def onAccept(self, seq):  # :270-285
        if seq == self.seqIP and self.acceptCountIP < self.p.majority:
            self.acceptCountIP += 1
            if self.acceptCountIP >= self.p.majority:
                self.proposalIP = False
                if self.acceptedValIP is None:
                    raise IllegalStateException()
                if self.valueAccepted is not None:
                    raise IllegalStateException("Already accepted a value")
                self.valueAccepted = self.acceptedValIP
                self.doneAt = self.p.network.time

    def onRejectOnCommit(self, seq, serverCurSeq):  # :287-297
        if seq == self.seqIP:
            self.reject2CountIP += 1
            if self.reject2CountIP == self.p.majority:
                self.proposalIP = False
                self.seqAccepted = max(self.seqAccepted, serverCurSeq)
                self.reject2Count += 1
                self.startNextProposal()

    def sendToAcceptors(self, m, sentTime):  # :299-303
        dest = list(self.p.acceptors)
        shuffle(dest, self.p.network.rd)
        self.p.network.send(m, self, dest, sentTime, _force_multi=True)  # network.send(m, sentTime, this, dest)

    def onTimeout(self, seq):  # :305-311
        if seq == self.seqIP and self.proposalIP:
            self.proposalIP = False
            self.timeoutCount += 1
            self.startNextProposal()

    def startNextProposal(self):  # :313-338
        if self.proposalIP:
            raise IllegalStateException()
        net, params = self.p.network, self.p.params
        self.acceptedSeqIP = self.acceptedValIP = None
        self.proposalIP = True
        self.agreeCountIP = self.reject1CountIP = self.acceptCountIP = self.reject2CountIP = 0
        gap = self.seqAccepted % params.proposerCount
        newSeqIP = self.seqAccepted + params.proposerCount - gap + self.rank
        self.seqIP = newSeqIP if newSeqIP > self.seqIP else self.seqIP + params.proposerCount
        p = Propose(self.seqIP)
        sentTime = net.time + 1
        self.sendToAcceptors(p, sentTime)
        net.registerTask(lambda: self.onTimeout(p.seq), sentTime + params.timeout, self)


class Paxos:
    def __init__(self, params=None, config=None):  # :32-37
        self.params = params or PaxosParameters()
        self.majority = self.params.acceptorCount // 2 + 1
        self._config = config
        self.network = HostNetwork(self.params.latency, config)
        self.acceptors, self.proposers = [], []

    def copy(self):
        return Paxos(self.params, self._config)

    def init(self):  # :374-387
        net = self.network
        with net.deferred_init():  # every proposer starts its first proposal before the next node is built
            for _ in range(self.params.acceptorCount):
                an = AcceptorNode(self)
                net.addNode(an)
                self.acceptors.append(an)
            for i in range(self.params.proposerCount):
                pn = ProposerNode(i, self)
                net.addNode(pn)
                self.proposers.append(pn)
                pn.startNextProposal()

    def finalCheck(self):  # play() :473-486
        val = None
        for pn in self.proposers:
            if val is None:
                val = pn.valueAccepted
            elif val != pn.valueAccepted:
                return False
        return True
