"""protocols.OptimisticP2PSignature (P/OptimisticP2PSignature.java) — "just send the signatures": every node floods its
own signature, a receiver forwards a signature it has not seen to all its peers but the sender — on the engine in
host-callback mode, over the reference's P2P layer (p2p.P2PNetwork = C/P2PNetwork.java with minimum == false, :83). The
second P2PNetwork protocol beside P2PFlood that runs this way. Every forward is one multi-destination
send (C/Network.java:418-447, delaysBetweenMessage == 0) queued, ordered and latency-sampled by libwittgpu.so; the action()
stays here. Host-side Python stand-in for the Java classes (no JVM in the build image); names follow the Java source."""
from wittgenstein_amd.hostnet import Message
from .p2p import P2PNetwork, P2PNode


class OptimisticP2PSignatureParameters:  # :33-72
    def __init__(self, nodeCount=100, threshold=99, connectionCount=20, pairingTime=1, nodeBuilderName=None,
                 networkLatencyName=None):
        self.nodeCount, self.threshold, self.connectionCount, self.pairingTime = nodeCount, threshold, connectionCount, pairingTime
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName


class SendSig(Message):  # :86-103
    def __init__(self, who):
        self.sig = who.nodeId

    def size(self):
        return 4 + 48  # NodeId + sig

    def action(self, network, frm, to):
        to.onSig(frm, self)


class P2PSigNode(P2PNode):  # :105-156
    def __init__(self, p):
        super().__init__(p.network)
        self.p = p
        self.verifiedSignatures = set()  # (a BitSet in the reference: membership and cardinality are all that is asked of it)
        self.done = False

    def onSig(self, frm, ss):  # :114-133
        if self.done or ss.sig in self.verifiedSignatures:
            return
        self.verifiedSignatures.add(ss.sig)
        net = self.p.network
        dests = [n for n in self.peers if n is not frm]
        net.send(ss, self, dests, net.time + 1, _force_multi=True)  # network.send(ss, network.time + 1, this, dests)
        if len(self.verifiedSignatures) >= self.p.params.threshold:
            self.done = True
            self.doneAt = net.time + self.p.params.pairingTime * 2


class OptimisticP2PSignature:
    def __init__(self, params=None, config=None):
        self.params = params or OptimisticP2PSignatureParameters()
        # every node forwards every signature once, each forward one multi-destination envelope that lives until its last
        # peer is reached: up to nodeCount^2 of them in flight (the engine's default ring holds max(4096, 16 x nodes))
        n = self.params.nodeCount
        config = dict(config or {})
        config.setdefault("chain_slots", max(4096, n * n))
        config.setdefault("chain_dests", max(1 << 20, n * n * (self.params.connectionCount + 4)))
        self.network = P2PNetwork(self.params.connectionCount, False, self.params.networkLatencyName, config)  # :76
        self._config = config

    def copy(self):
        return OptimisticP2PSignature(self.params, self._config)

    def init(self):  # :158-167
        # (the reference registers node i's first task right after addNode(i); the engine takes its node table whole, so the
        # tasks are registered behind the loop, in the same order — registerTask draws nothing and node construction does not
        # look at the queue, so the rd sequence and the bucket's push order are the reference's)
        net = self.network
        nodes = [P2PSigNode(self) for _ in range(self.params.nodeCount)]
        for n in nodes:
            net.addNode(n)
        for n in nodes:
            net.registerTask(lambda n=n: n.onSig(n, SendSig(n)), 1, n)
        net.setPeers()
