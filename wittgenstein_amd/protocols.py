"""Host-side mirror of the reference's Protocol objects (C/Protocol.java:9-22) for the resident
protocols: same constructor arguments, copy(), init(), network()."""
import ctypes as C

from . import _lib as L
from .core import Network, _raise


_config = L.make_config


class PingPongParameters:
    """P/PingPong.java:34-50"""

    def __init__(self, nodeCt=1000, nodeBuilderName=None, networkLatencyName=None):
        self.nodeCt, self.nodeBuilderName, self.networkLatencyName = nodeCt, nodeBuilderName, networkLatencyName


class PingPong:
    """P/PingPong.java. `seed` is what RunMultipleTimes does with network.rd.setSeed(i) before init()."""

    def __init__(self, params=None, seed=0, config=None):
        self.params = params or PingPongParameters()
        self.seed, self.config = seed, config
        self._net = None
        self.init_seconds = None

    def copy(self):
        return PingPong(self.params, self.seed, self.config)

    def init(self):
        p = self.params
        h = C.c_void_p()
        cfg = _config(self.config)
        rc = L.lib().wgh_pingpong_create(p.nodeCt, p.nodeBuilderName.encode() if p.nodeBuilderName else None,
                                         p.networkLatencyName.encode() if p.networkLatencyName else None,
                                         C.c_int64(self.seed), C.byref(cfg), C.byref(h))
        if rc != L.WG_OK:
            _raise(rc, L.lib().wgh_last_error().decode())
        self._net = Network(h)
        self.init_seconds = L.lib().wgh_last_init_seconds()

    def network(self):
        return self._net


class HandelParameters:
    """P/Handel.java:97-142 (constructor argument order preserved). byzantineSuicide (:538-559, 577-584, 688-694) and
    hiddenByzantine (:813-817, 840-917) are resident on the device, with the down nodes chosen by Network.chooseBadNodes or
    given explicitly: badNodes = the ids of the BitSet's set bits (any iterable of ints), used by init() instead of the
    draws (:960-964)."""

    def __init__(self, nodeCount, threshold, pairingTime, levelWaitTime, extraCycle, disseminationPeriodMs, fastPath,
                 nodesDown, nodeBuilderName=None, networkLatencyName=None, desynchronizedStart=0,
                 byzantineSuicide=False, hiddenByzantine=False, badNodes=None):
        if byzantineSuicide and hiddenByzantine:
            from .core import IllegalArgumentException
            raise IllegalArgumentException("Only one attack at a time")
        self.badNodes = None if badNodes is None else sorted(set(int(i) for i in badNodes))
        if self.badNodes and not (0 <= self.badNodes[0] and self.badNodes[-1] < nodeCount):
            from .core import IllegalArgumentException
            raise IllegalArgumentException("badNodes: ids must be in [0, nodeCount)")
        self.byzantineSuicide, self.hiddenByzantine = bool(byzantineSuicide), bool(hiddenByzantine)
        self.nodeCount, self.threshold, self.pairingTime, self.levelWaitTime = nodeCount, threshold, pairingTime, levelWaitTime
        self.extraCycle, self.disseminationPeriodMs, self.fastPath, self.nodesDown = extraCycle, disseminationPeriodMs, fastPath, nodesDown
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName
        self.desynchronizedStart = desynchronizedStart


class Handel:
    """P/Handel.java."""

    def __init__(self, params, seed=0, config=None):
        self.params, self.seed, self.config = params, seed, config
        self._net = None
        self.init_seconds = None

    def copy(self):
        return Handel(self.params, self.seed, self.config)

    def init(self):
        p = self.params
        hp = L.wg_handel_params(p.nodeCount, p.threshold, p.pairingTime, p.levelWaitTime, p.extraCycle,
                                p.disseminationPeriodMs, p.fastPath, p.nodesDown, p.desynchronizedStart, 0, 0, 0,
                                int(p.byzantineSuicide), int(p.hiddenByzantine))
        h = C.c_void_p()
        cfg = _config(self.config)
        bad = None
        if p.badNodes is not None:
            bad = (C.c_uint8 * p.nodeCount)()
            for i in p.badNodes:
                bad[i] = 1
        rc = L.lib().wgh_handel_create_bad_nodes(C.byref(hp), bad, p.nodeBuilderName.encode() if p.nodeBuilderName else None,
                                                 p.networkLatencyName.encode() if p.networkLatencyName else None,
                                                 C.c_int64(self.seed), C.byref(cfg), C.byref(h))
        if rc != L.WG_OK:
            _raise(rc, L.lib().wgh_last_error().decode())
        self._net = Network(h)
        self.init_seconds = L.lib().wgh_last_init_seconds()
        self.init_on_device = bool(L.lib().wgh_last_init_on_device())  # the emission lists (P/Handel.java:991-1013)

    def network(self):
        return self._net

    def cont_if(self):
        """Handel.newContIf (P/Handel.java:1044-1053): some live node has doneAt == 0 or addedCycle > 0."""
        v = C.c_int32()
        self._net._ck(L.lib().wg_protocol_cont_if(self._net._h, C.byref(v)))
        return bool(v.value)


class GSFSignatureParameters:
    """P/GSFSignature.java:26-107 (constructor argument order preserved; the ratio overload :86-106 is
    `from_ratios`)."""

    def __init__(self, nodeCount=32768 // 32, threshold=None, pairingTime=3, timeoutPerLevelMs=50, periodDurationMs=10,
                 acceleratedCallsCount=10, nodesDown=0, nodeBuilderName=None, networkLatencyName=None):
        if threshold is None:
            threshold = int(nodeCount * 0.99)
        if nodesDown >= nodeCount or nodesDown < 0 or threshold > nodeCount or nodesDown + threshold > nodeCount:
            from .core import IllegalArgumentException
            raise IllegalArgumentException("nodeCount=%d, threshold=%d" % (nodeCount, threshold))
        self.nodeCount, self.threshold, self.pairingTime = nodeCount, threshold, pairingTime
        self.timeoutPerLevelMs, self.periodDurationMs = timeoutPerLevelMs, periodDurationMs
        self.acceleratedCallsCount, self.nodesDown = acceleratedCallsCount, nodesDown
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName

    @classmethod
    def from_ratios(cls, nodeCount, ratioThreshold, pairingTime, timeoutPerLevelMs, periodDurationMs,
                    acceleratedCallsCount, ratioNodesDown, nodeBuilderName=None, networkLatencyName=None):
        return cls(nodeCount, int(ratioThreshold * nodeCount), pairingTime, timeoutPerLevelMs, periodDurationMs,
                   acceleratedCallsCount, int(ratioNodesDown * nodeCount), nodeBuilderName, networkLatencyName)


class GSFSignature:
    """P/GSFSignature.java."""

    def __init__(self, params=None, seed=0, config=None):
        self.params = params or GSFSignatureParameters()
        self.seed, self.config = seed, config
        self._net = None
        self.init_seconds = None

    def copy(self):
        return GSFSignature(self.params, self.seed, self.config)

    def init(self):
        p = self.params
        gp = L.wg_gsf_params(p.nodeCount, p.threshold, p.pairingTime, p.timeoutPerLevelMs, p.periodDurationMs,
                             p.acceleratedCallsCount, p.nodesDown)
        h = C.c_void_p()
        cfg = _config(self.config)
        rc = L.lib().wgh_gsf_create(C.byref(gp), p.nodeBuilderName.encode() if p.nodeBuilderName else None,
                                    p.networkLatencyName.encode() if p.networkLatencyName else None,
                                    C.c_int64(self.seed), C.byref(cfg), C.byref(h))
        if rc != L.WG_OK:
            _raise(rc, L.lib().wgh_last_error().decode())
        self._net = Network(h)
        self.init_seconds = L.lib().wgh_last_init_seconds()

    def network(self):
        return self._net

    def cont_if(self):
        """GSFSignature.newConfIf (P/GSFSignature.java:670-683): some live node holds < threshold signatures."""
        v = C.c_int32()
        self._net._ck(L.lib().wg_protocol_cont_if(self._net._h, C.byref(v)))
        return bool(v.value)


class SanFerminSignatureParameters:
    """P/SanFerminSignature.java:39-111 (constructor argument order preserved; shuffledLists is read nowhere in the
    protocol)."""

    def __init__(self, nodeCount=32768 // 32, threshold=32768 // 32, pairingTime=2, signatureSize=48, replyTimeout=300,
                 candidateCount=1, shuffledLists=False, nodeBuilderName=None, networkLatencyName=None):
        self.nodeCount, self.threshold, self.pairingTime = nodeCount, threshold, pairingTime
        self.signatureSize, self.replyTimeout, self.candidateCount = signatureSize, replyTimeout, candidateCount
        self.shuffledLists, self.nodeBuilderName, self.networkLatencyName = shuffledLists, nodeBuilderName, networkLatencyName


class SanFerminSignature:
    """P/SanFerminSignature.java resident on the device (wittgenstein_amd/csrc/proto_sanfermin.hip.h). `seed` is
    RunMultipleTimes' rd.setSeed(i) on the copy: the constructor has built the nodes from new Random(0) before it."""

    def __init__(self, params=None, seed=0, config=None):
        self.params = params or SanFerminSignatureParameters()
        self.seed, self.config = seed, config
        self._net = None
        self.init_seconds = None

    def copy(self):
        return SanFerminSignature(self.params, self.seed, self.config)

    def init(self):
        p = self.params
        sp = L.wg_sanfermin_params(p.nodeCount, p.threshold, p.pairingTime, p.signatureSize, p.replyTimeout,
                                   p.candidateCount)
        h = C.c_void_p()
        cfg = _config(self.config)
        rc = L.lib().wgh_sanfermin_create(C.byref(sp), p.nodeBuilderName.encode() if p.nodeBuilderName else None,
                                          p.networkLatencyName.encode() if p.networkLatencyName else None,
                                          C.c_int64(self.seed), C.byref(cfg), C.byref(h))
        if rc != L.WG_OK:
            _raise(rc, L.lib().wgh_last_error().decode())
        self._net = Network(h)
        self.init_seconds = L.lib().wgh_last_init_seconds()

    def network(self):
        return self._net

    def cont_if(self):
        """some live node has not finished (RunMultipleTimes.contUntilDone's shape)"""
        v = C.c_int32()
        self._net._ck(L.lib().wg_protocol_cont_if(self._net._h, C.byref(v)))
        return bool(v.value)


class CasperParemeters:
    """(sic) P/CasperIMD.java:18-70, constructor argument order preserved."""

    def __init__(self, cycleLength=4, randomOnTies=True, blockProducersCount=2, attestersPerRound=20,
                 blockConstructionTime=1000, attestationConstructionTime=1, nodeBuilderName=None,
                 networkLatencyName=None):
        self.cycleLength, self.randomOnTies, self.blockProducersCount = cycleLength, randomOnTies, blockProducersCount
        self.attestersPerRound, self.attestersCount = attestersPerRound, attestersPerRound * cycleLength
        self.blockConstructionTime, self.attestationConstructionTime = blockConstructionTime, attestationConstructionTime
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName


class CasperIMD:
    """P/CasperIMD.java resident on the device (wittgenstein_amd/csrc/proto_casper.hip.h): init() as the reference's,
    with ByzBlockProducerWF(byzDelay). max_slots sizes the attestation / block tables (a slot is 8 s). `seed` is
    RunMultipleTimes' rd.setSeed(i) on the copy (the constructor has built the observer from new Random(0) before)."""

    def __init__(self, params=None, seed=0, config=None, byz_delay=0, max_slots=64):
        self.params = params or CasperParemeters()
        self.seed, self.config, self.byz_delay, self.max_slots = seed, config, byz_delay, max_slots
        self._net = None
        self.init_seconds = None

    def copy(self):
        return CasperIMD(self.params, self.seed, self.config, self.byz_delay, self.max_slots)

    def init(self):
        p = self.params
        cp = L.wg_casper_params(p.cycleLength, int(bool(p.randomOnTies)), p.blockProducersCount, p.attestersPerRound,
                                p.blockConstructionTime, p.attestationConstructionTime, self.byz_delay, self.max_slots)
        h = C.c_void_p()
        cfg = _config(self.config)
        rc = L.lib().wgh_casper_create(C.byref(cp), p.nodeBuilderName.encode() if p.nodeBuilderName else None,
                                       p.networkLatencyName.encode() if p.networkLatencyName else None,
                                       C.c_int64(self.seed), C.byref(cfg), C.byref(h))
        if rc != L.WG_OK:
            _raise(rc, L.lib().wgh_last_error().decode())
        self._net = Network(h)
        self.init_seconds = L.lib().wgh_last_init_seconds()

    def network(self):
        return self._net

    def attester_ids(self):
        p = self.params
        first = 1 + p.blockProducersCount
        return range(first, first + p.cycleLength * p.attestersPerRound)

    def stop_attesters(self, count, seed=0):
        """BASELINE config 5's "+10 % Byzantine": the reference has no such population (init() installs exactly one
        ByzBlockProducerWF, P/CasperIMD.java:473-476), so it is defined as `count` attesters stop()ped
        (C/Node.java:120-123) after init(), chosen the way Network.chooseBadNodes picks (C/Network.java:52-64: draw
        until `count` distinct ones) from a java.util.Random(seed) of their own — the simulation's rd is not touched.
        Returns the node ids, so that a checker can stop the same nodes."""
        ids = choose_attesters(self.attester_ids(), count, seed)
        for i in ids:
            self._net.set_node_down(i, True)
        return ids


class _JavaRandom:
    """java.util.Random's nextInt(bound) (JDK javadoc algorithm) for host-side choices that need no engine"""

    def __init__(self, seed):
        self.s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def _next(self, bits):
        self.s = (self.s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.s >> (48 - bits)
        return v - (1 << 32) if v >= (1 << 31) else v  # (int) of the shifted seed

    def nextInt(self, bound):
        if bound <= 0:
            raise ValueError("bound must be positive")
        r = self._next(31)
        m = bound - 1
        if (bound & m) == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            if u - r + m < (1 << 31):
                return r
            u = self._next(31)


def choose_attesters(attester_ids, count, seed=0):
    ids = list(attester_ids)
    if not 0 <= count < len(ids):
        raise ValueError("count must be in [0, attesters)")
    rd, bad, out = _JavaRandom(seed), set(), []
    while len(out) < count:
        k = rd.nextInt(len(ids))
        if k not in bad:
            bad.add(k)
            out.append(ids[k])
    return out


class P2PFloodParameters:
    """P/P2PFlood.java:41-86, constructor argument order preserved."""

    def __init__(self, nodeCount=100, deadNodeCount=10, delayBeforeResent=50, msgCount=1, msgToReceive=1, peersCount=10,
                 delayBetweenSends=30, nodeBuilderName=None, networkLatencyName=None):
        self.nodeCount, self.deadNodeCount, self.delayBeforeResent = nodeCount, deadNodeCount, delayBeforeResent
        self.msgCount, self.msgToReceive, self.peersCount = msgCount, msgToReceive, peersCount
        self.delayBetweenSends = delayBetweenSends
        self.nodeBuilderName, self.networkLatencyName = nodeBuilderName, networkLatencyName


class P2PFlood:
    """P/P2PFlood.java resident on the device (wittgenstein_amd/csrc/proto_p2pflood.hip.h); the peer graph is built on
    the host as P2PNetwork.setPeers does."""

    def __init__(self, params=None, seed=0, config=None):
        self.params = params or P2PFloodParameters()
        self.seed, self.config = seed, config
        self._net = None
        self.init_seconds = None

    def copy(self):
        return P2PFlood(self.params, self.seed, self.config)

    def init(self):
        p = self.params
        fp = L.wg_p2pflood_params(p.nodeCount, p.deadNodeCount, p.delayBeforeResent, p.msgCount, p.msgToReceive,
                                  p.peersCount, p.delayBetweenSends)
        h = C.c_void_p()
        cfg = _config(self.config)
        rc = L.lib().wgh_p2pflood_create(C.byref(fp), p.nodeBuilderName.encode() if p.nodeBuilderName else None,
                                         p.networkLatencyName.encode() if p.networkLatencyName else None,
                                         C.c_int64(self.seed), C.byref(cfg), C.byref(h))
        if rc != L.WG_OK:
            _raise(rc, L.lib().wgh_last_error().decode())
        self._net = Network(h)
        self.init_seconds = L.lib().wgh_last_init_seconds()

    def network(self):
        return self._net
