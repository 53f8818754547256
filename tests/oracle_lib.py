"""ctypes bindings for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg. The product package (wittgenstein_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_last_error.restype = C.c_char_p
        _LIB.orc_handel_init_seconds.restype = C.c_double
        _LIB.orc_gsf_shape_violations.restype = C.c_uint64
    return _LIB


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("oracle error %d: %s" % (code, msg))
        self.code = code


def _ck(rc):
    if rc != 0:
        raise OracleError(rc, lib().orc_last_error().decode())


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def jrandom_ints(seed, n):
    out = np.zeros(n, np.int32)
    _ck(lib().orc_jrandom_ints(C.c_int64(seed), n, _p(out, C.c_int32)))
    return out


def jrandom_bounded(seed, bound, n):
    out = np.zeros(n, np.int32)
    _ck(lib().orc_jrandom_bounded(C.c_int64(seed), bound, n, _p(out, C.c_int32)))
    return out


def jrandom_doubles(seed, n):
    out = np.zeros(n, np.float64)
    _ck(lib().orc_jrandom_doubles(C.c_int64(seed), n, _p(out, C.c_double)))
    return out


def jrandom_booleans(seed, n):
    out = np.zeros(n, np.uint8)
    _ck(lib().orc_jrandom_booleans(C.c_int64(seed), n, _p(out, C.c_uint8)))
    return out


def jshuffle_iota(seed, n, rounds=1):
    out = np.zeros(n, np.int32)
    _ck(lib().orc_jshuffle_iota(C.c_int64(seed), n, rounds, _p(out, C.c_int32)))
    return out


def pseudo_random(node_id, seed):
    return lib().orc_pseudo_random(C.c_int32(node_id), C.c_int32(seed))


def latency_bydistance(dist, delta):
    return lib().orc_latency_bydistance(dist, delta)


def latency(name, x1, y1, e1, x2, y2, e2, delta, same=False):
    out = C.c_int32()
    _ck(lib().orc_latency(name.encode() if name else None, x1, y1, e1, x2, y2, e2, int(same), delta, C.byref(out)))
    return out.value


def latency_table():
    md = lib().orc_max_dist()
    t = np.zeros((md + 1, 100), np.int32)
    for d in range(md + 1):
        for k in range(100):
            t[d, k] = latency_bydistance(d, k)
    return t


def node_xy(rd_int):
    x, y = C.c_int32(), C.c_int32()
    lib().orc_node_xy(C.c_int32(rd_int), C.byref(x), C.byref(y))
    return x.value, y.value


class PingPong:
    FIELDS = {"pong": 0, "msgReceived": 1, "msgSent": 2, "bytesSent": 3, "bytesReceived": 4, "x": 5, "y": 6,
              "down": 7}

    def __init__(self, node_ct=1000, nb=None, nl=None, seed=0):
        self.h = C.c_void_p()
        self.n = node_ct
        _ck(lib().orc_pingpong_create(node_ct, nb.encode() if nb else None, nl.encode() if nl else None,
                                      C.c_int64(seed), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pingpong_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_pingpong_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_pingpong_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        lib().orc_pingpong_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value}


class Handel:
    FIELDS = {"doneAt": 0, "msgReceived": 1, "msgSent": 2, "bytesSent": 3, "bytesReceived": 4, "sigsChecked": 5,
              "sigQueueSize": 6, "msgFiltered": 7, "currWindowSize": 8, "addedCycle": 9, "down": 10, "x": 11,
              "y": 12, "startAt": 13, "nodePairingTime": 14, "extraLatency": 15}
    LEVEL_FIELDS = {"posInLevel": 0, "outgoingFinished": 1, "queueLen": 2, "suicideBizAfter": 3}
    BITS = {"totalIncoming": 0, "lastAggVerified": 1, "verifiedIndSignatures": 2, "toVerifyInd": 3,
            "finishedPeers": 4, "totalOutgoingLast": 5, "waitedSigs": 6, "blacklist": 7}

    def __init__(self, node_count, threshold, pairing_time, level_wait_time, extra_cycle, period, fast_path,
                 nodes_down, nb=None, nl=None, desync=0, seed=0, byzantine_suicide=False, hidden_byzantine=False, bad_nodes=None):
        ip = (C.c_int32 * 9)(node_count, threshold, pairing_time, level_wait_time, extra_cycle, period, fast_path,
                             nodes_down, desync)
        self.h = C.c_void_p()
        self.n = node_count
        bad = None
        if bad_nodes is not None:  # HandelParameters.badNodes (P/Handel.java:51): the ids of the set bits
            bad = (C.c_uint8 * node_count)()
            for i in bad_nodes:
                bad[i] = 1
        _ck(lib().orc_handel_create_bad(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed),
                                        int(byzantine_suicide), int(hidden_byzantine), bad, C.byref(self.h)))
        self.levels = lib().orc_handel_levels(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_handel_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_handel_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def cont_if(self):
        return bool(lib().orc_handel_cont_if(self.h))

    def init_seconds(self):
        return lib().orc_handel_init_seconds(self.h)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_handel_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def read_level(self, field):
        out = np.zeros((self.n, self.levels), np.int32)
        _ck(lib().orc_handel_read_level(self.h, self.LEVEL_FIELDS[field], _p(out, C.c_int32)))
        return out

    def read_bits(self, which):
        w = (self.n + 63) // 64
        out = np.zeros((self.n, w), np.uint64)
        _ck(lib().orc_handel_read_bits(self.h, self.BITS[which], _p(out, C.c_uint64)))
        return out

    def read_ranks(self, node):
        out = np.zeros(self.n, np.int32)
        lib().orc_handel_read_ranks(self.h, node, _p(out, C.c_int32))
        return out

    def read_peers(self, node, level):
        out = np.zeros(self.n, np.int32)
        cnt = C.c_int32()
        lib().orc_handel_read_peers(self.h, node, level, _p(out, C.c_int32), C.byref(cnt))
        return out[:cnt.value].copy()

    def info(self, with_queue=True):
        t, q, r, d, k = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().orc_handel_info(self.h, C.byref(t), C.byref(q) if with_queue else None, C.byref(r), C.byref(d),
                              C.byref(k))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value, "tasks": k.value}

    def stats(self):
        dl = np.zeros(32, np.uint64)
        qm = np.zeros(32, np.int32)
        lib().orc_handel_stats(self.h, _p(dl, C.c_uint64), _p(qm, C.c_int32))
        return {"deliveredByLevel": dl[:self.levels].copy(), "queueMax": qm[:self.levels].copy()}


class GSFSignature:
    FIELDS = {"doneAt": 0, "msgReceived": 1, "msgSent": 2, "bytesSent": 3, "bytesReceived": 4, "sigChecked": 5,
              "sigQueueSize": 6, "toVerifySize": 7, "verifiedCardinality": 8, "down": 10, "x": 11, "y": 12,
              "nodePairingTime": 14, "extraLatency": 15}
    LEVEL_FIELDS = {"posInLevel": 0, "remainingCalls": 1}
    BITS = {"verifiedSignatures": 0, "levelVerified": 1, "individualSignatures": 2, "indivVerifiedSig": 3,
            "waitedSigs": 4}

    def __init__(self, node_count, threshold, pairing_time, timeout_per_level, period, accelerated_calls,
                 nodes_down, nb=None, nl=None, seed=0):
        ip = (C.c_int32 * 7)(node_count, threshold, pairing_time, timeout_per_level, period, accelerated_calls,
                             nodes_down)
        self.h = C.c_void_p()
        self.n = node_count
        _ck(lib().orc_gsf_create(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed),
                                 C.byref(self.h)))
        self.levels = lib().orc_gsf_levels(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_gsf_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_gsf_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def set_copy_on_delivery(self, on=True):
        lib().orc_gsf_set_copy_on_delivery(self.h, int(on))

    def cont_if(self):
        return bool(lib().orc_gsf_cont_if(self.h))

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_gsf_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def read_level(self, field):
        out = np.zeros((self.n, self.levels), np.int32)
        _ck(lib().orc_gsf_read_level(self.h, self.LEVEL_FIELDS[field], _p(out, C.c_int32)))
        return out

    def read_bits(self, which):
        w = (self.n + 63) // 64
        out = np.zeros((self.n, w), np.uint64)
        _ck(lib().orc_gsf_read_bits(self.h, self.BITS[which], _p(out, C.c_uint64)))
        return out

    def read_peers(self, node, level):
        out = np.zeros(self.n, np.int32)
        cnt = C.c_int32()
        lib().orc_gsf_read_peers(self.h, node, level, _p(out, C.c_int32), C.byref(cnt))
        return out[:cnt.value].copy()

    def info(self, with_queue=True):
        t, q, r, d, k, qm = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
        s = C.c_double()
        lib().orc_gsf_info(self.h, C.byref(t), C.byref(q) if with_queue else None, C.byref(r), C.byref(d),
                           C.byref(k), C.byref(qm), C.byref(s))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value, "tasks": k.value,
                "queueMax": qm.value, "init_s": s.value}

    def stats(self):
        dl = np.zeros(32, np.uint64)
        lib().orc_gsf_stats(self.h, _p(dl, C.c_uint64))
        return {"deliveredByLevel": dl[:self.levels].copy(),
                "shapeViolations": int(lib().orc_gsf_shape_violations(self.h))}


class CasperIMD:
    """oracle/casper.hpp through oracle/capi.cpp; params = (cycleLength, randomOnTies, blockProducersCount,
    attestersPerRound, blockConstructionTime, attestationConstructionTime) — CasperParemeters' ctor order
    (P/CasperIMD.java:52-70)."""
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "headHeight": 4,
              "headProposalTime": 5, "headId": 6, "attestationsByHeadSize": 7, "x": 8, "y": 9, "blocksReceived": 10,
              "attestationsHeld": 11}

    BYZ = {"WF": 0, "plain": 1, "SF": 2, "NS": 3}  # the ByzBlockProducer given to init(badNode), P/CasperIMD.java:481

    def __init__(self, params, nb=None, nl=None, seed=0, byz_delay=0, byz="WF"):
        self.h = C.c_void_p()
        ip = (C.c_int32 * 7)(params[0], int(params[1]), params[2], params[3], params[4], params[5], byz_delay)
        _ck(lib().orc_casper_create_byz(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed),
                                        self.BYZ[byz], C.byref(self.h)))
        self.n = lib().orc_casper_node_count(self.h)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_casper_destroy(self.h)
            self.h = None

    def byz_counters(self):
        """onDirectFather, onOlderAncestor, incNotTheBestFather, skipped, toSend of the byzantine producer (node 1)"""
        out = (C.c_int32 * 5)()
        _ck(lib().orc_casper_byz_counters(self.h, out))
        return dict(zip(("onDirectFather", "onOlderAncestor", "incNotTheBestFather", "skipped", "toSend"), list(out)))

    def stop(self, ids):
        """Node.stop() on the listed nodes (SURVEY.md §8d's definition of config 5's stopped attesters)"""
        a = np.ascontiguousarray(ids, np.int32)
        _ck(lib().orc_casper_stop(self.h, _p(a, C.c_int32), len(a)))

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_casper_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_casper_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d, k = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().orc_casper_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d), C.byref(k))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value, "tasks": k.value}


class Fuzz:
    """oracle/fuzz.hpp — the scheduler stress protocol (test infrastructure on both sides; tests/fuzz_protocol.py is
    its twin on the engine's host-callback mode)."""
    FIELDS = {"h": 0, "c": 1, "msgReceived": 2, "msgSent": 3, "bytesSent": 4, "bytesReceived": 5}
    OPS = {"partition": 0, "endPartition": 1, "stop": 2, "start": 3, "setMsgDiscardTime": 4}

    def __init__(self, n, ttl, nl=None, seed=0):
        self.h, self.n = C.c_void_p(), n
        _ck(lib().orc_fuzz_create(n, ttl, nl.encode() if nl else None, C.c_int64(seed), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_fuzz_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_fuzz_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def op(self, name, arg=0):
        _ck(lib().orc_fuzz_op(self.h, self.OPS[name], int(arg)))

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_fuzz_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d, k = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().orc_fuzz_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d), C.byref(k))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value, "tasks": k.value}


class SanFerminSignature:
    """oracle/sanfermin.hpp; params = (nodeCount, threshold, pairingTime, signatureSize, replyTimeout, candidateCount)
    — SanFerminSignatureParameters' ctor order (P/SanFerminSignature.java:84-104)."""
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "aggValue": 4,
              "currentPrefixLength": 5, "doneAt": 6, "thresholdAt": 7, "sentRequests": 8, "receivedRequests": 9,
              "done": 10, "isSwapping": 11, "x": 12, "y": 13}

    def __init__(self, params, nb=None, nl=None, seed=0):
        self.h, self.n = C.c_void_p(), params[0]
        ip = (C.c_int32 * 6)(*params)
        _ck(lib().orc_sanfermin_create(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed),
                                       C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_sanfermin_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_sanfermin_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_sanfermin_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d, k, f = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
        lib().orc_sanfermin_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d), C.byref(k), C.byref(f))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value, "tasks": k.value,
                "finished": f.value}


class SanFerminCappos:
    """oracle/sanfermin_cappos.hpp; params = (nodeCount, threshold, pairingTime, signatureSize, timeout, candidateCount) —
    SanFerminParameters' ctor order (P/SanFerminCappos.java:87-106)."""
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "totalNumberOfSigs": 4,
              "currentPrefixLength": 5, "doneAt": 6, "thresholdAt": 7, "cachedLevels": 8, "cachedValues": 9,
              "done": 10, "isSwapping": 11, "x": 12, "y": 13}

    def __init__(self, params, nb=None, nl=None, seed=0):
        self.h, self.n = C.c_void_p(), params[0]
        ip = (C.c_int32 * 6)(*params)
        _ck(lib().orc_cappos_create(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed),
                                    C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_cappos_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_cappos_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_cappos_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d, k, f = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
        lib().orc_cappos_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d), C.byref(k), C.byref(f))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value, "tasks": k.value,
                "finished": f.value}


class P2PFlood:
    """oracle/p2pflood.hpp; params = P2PFloodParameters' ctor order (P/P2PFlood.java:63-86): (nodeCount, deadNodeCount,
    delayBeforeResent, msgCount, msgToReceive, peersCount, delayBetweenSends)."""
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "doneAt": 4, "down": 5,
              "received": 6, "peerCount": 7, "peerDigest": 8, "x": 9, "y": 10}

    def __init__(self, params, nb=None, nl=None, seed=0):
        self.h, self.n = C.c_void_p(), params[0]
        ip = (C.c_int32 * 7)(*params)
        _ck(lib().orc_p2pflood_create(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed),
                                      C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_p2pflood_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_p2pflood_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_p2pflood_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        lib().orc_p2pflood_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value}


class OptimisticP2PSignature:
    """oracle/optimistic_p2p.hpp; params = OptimisticP2PSignatureParameters' ctor order (P/OptimisticP2PSignature.java:58-71):
    (nodeCount, threshold, connectionCount, pairingTime)."""
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "doneAt": 4, "done": 5,
              "sigs": 6, "peerCount": 7, "peerDigest": 8, "x": 9, "y": 10}

    def __init__(self, params, nb=None, nl=None, seed=0):
        self.h, self.n = C.c_void_p(), params[0]
        ip = (C.c_int32 * 4)(*params)
        _ck(lib().orc_optp2p_create(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed),
                                    C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_optp2p_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_optp2p_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_optp2p_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        lib().orc_optp2p_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value}


class Dfinity:
    """oracle/dfinity.hpp: P/Dfinity.java; params = DfinityParameters' ctor order (blockProducersCount, attestersCount,
    attestersPerRound, blockConstructionTime, attestationConstructionTime, percentageDeadAttester). Nodes: the observer, the
    attesters, the producers, the beacon nodes; a field of another kind of node reads -2."""
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "x": 4, "y": 5, "headHeight": 6, "headId": 7,
              "headTime": 8, "lastRandomBeacon": 9, "blocksReceived": 10, "majorityBlocks": 11, "majorityHeightSum": 12,
              "voteForHeight": 13, "proposals": 14, "votes": 15, "waitForBlockHeight": 16, "myRound": 17, "rbHeight": 18,
              "lastRDSent": 19, "rbRd": 20, "exchanged": 21}

    def __init__(self, params, nb=None, nl=None, seed=0):
        self.h = C.c_void_p()
        ip = (C.c_int32 * 6)(*params)
        _ck(lib().orc_dfinity_create(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed), C.byref(self.h)))
        self.n = lib().orc_dfinity_node_count(self.h)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_dfinity_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_dfinity_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_dfinity_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        lib().orc_dfinity_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value}


class P2PHandel:
    """oracle/p2phandel.hpp: P/P2PHandel.java; params = P2PHandelParameters' ctor order (signingNodeCount, relayingNodeCount,
    threshold, connectionCount, pairingTime, sigsSendPeriod, doubleAggregateStrategy, sendSigsStrategy, sendState);
    sendSigsStrategy: "all" / "dif" / "cmp_all" / "cmp_diff"."""
    STRATEGY = {"all": 0, "dif": 1, "cmp_all": 2, "cmp_diff": 3}
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "doneAt": 4, "x": 5, "y": 6, "sigs": 7,
              "sigsDigest": 8, "toVerify": 9, "toVerifyCapacity": 10, "toVerifyOrder": 11, "peerCount": 12, "peerDigest": 13,
              "justRelay": 14, "peersState": 15}

    def __init__(self, params, nb=None, nl=None, seed=0):
        self.h, self.n = C.c_void_p(), params[0] + params[1]
        ip = (C.c_int32 * 9)(params[0], params[1], params[2], params[3], params[4], params[5], int(params[6]),
                             self.STRATEGY[params[7]], int(params[8]))
        _ck(lib().orc_p2phandel_create(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_p2phandel_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_p2phandel_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_p2phandel_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        lib().orc_p2phandel_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value}

    def compressed_size(self, binary):
        return lib().orc_p2phandel_compressed_size(self.h, binary.encode())

    def probe(self):
        out = (C.c_int32 * 3)()
        _ck(lib().orc_p2phandel_probe(self.h, out))
        return list(out)


class Paxos:
    """oracle/paxos.hpp: P/Paxos.java; params = PaxosParameters' ctor order (acceptorCount, proposerCount, timeout). Per-node
    reads cover both kinds of node: a field of the other kind reads -2, a null Integer -1."""
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "doneAt": 4, "x": 5, "y": 6, "maxAgreed": 7,
              "acceptedSeq": 8, "acceptedVal": 9, "agreedTo": 10, "valueProposed": 11, "valueAccepted": 12, "seqIP": 13,
              "seqAccepted": 14, "agreeCount": 15, "reject1Count": 16, "reject2Count": 17, "timeoutCount": 18, "proposalIP": 19,
              "agreeCountIP": 20, "acceptCountIP": 21}

    def __init__(self, params, nb=None, nl=None, seed=0):
        self.h, self.n = C.c_void_p(), params[0] + params[1]
        ip = (C.c_int32 * 3)(*params)
        _ck(lib().orc_paxos_create(ip, nb.encode() if nb else None, nl.encode() if nl else None, C.c_int64(seed), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_paxos_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_paxos_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_paxos_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        lib().orc_paxos_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value}


class Slush:
    """oracle/slush.hpp: P/Slush.java, or P/Snowflake.java with snowflake=True; params = (NODES_AV, M, K, A[, B]) in the
    reference's ctor order (SlushParameters :37-47, SnowflakeParameters :36-52)."""
    FIELDS = {"msgReceived": 0, "msgSent": 1, "bytesSent": 2, "bytesReceived": 3, "myColor": 4, "myQueryNonce": 5, "round": 6,
              "cnt": 7, "answersInProgress": 8, "x": 9, "y": 10}

    def __init__(self, params, nb=None, nl=None, seed=0, snowflake=False):
        self.h, self.n = C.c_void_p(), params[0]
        b = params[4] if len(params) > 4 else 0
        ip = (C.c_int32 * 5)(params[0], params[1], params[2], b, 1 if snowflake else 0)
        lib().orc_slush_create.argtypes = [C.c_void_p, C.c_double, C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p]
        _ck(lib().orc_slush_create(ip, C.c_double(params[3]), nb.encode() if nb else None, nl.encode() if nl else None,
                                   C.c_int64(seed), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().orc_slush_destroy(self.h)
            self.h = None

    def run_ms(self, ms):
        d = C.c_int()
        _ck(lib().orc_slush_run_ms(self.h, ms, C.byref(d)))
        return bool(d.value)

    def read(self, field):
        out = np.zeros(self.n, np.int64)
        _ck(lib().orc_slush_read(self.h, self.FIELDS[field], _p(out, C.c_int64)))
        return out

    def info(self):
        t, q, r, d = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        lib().orc_slush_info(self.h, C.byref(t), C.byref(q), C.byref(r), C.byref(d))
        return {"time": t.value, "queue": q.value, "rng": r.value, "delivered": d.value}


# ---- city topology / latency (oracle/geo.hpp): data from tests/golden/city_data.json
_CITY_LOADED = False


def load_city_data(path=None):
    global _CITY_LOADED
    if _CITY_LOADED:
        return
    import json
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "city_data.json")
    d = json.load(open(path))
    lines = ["D\t%s" % c for c in d["dirs"]]
    for i, m in enumerate(d["ping"]):
        lines += ["P\t%d\t%s\t%s" % (i, k, v) for k, v in m.items()]
    lines += ["C\t%s\t%s\t%s\t%s" % tuple(r) for r in d["cities"]]
    _ck(lib().orc_city_data_load("\n".join(lines).encode()))
    _CITY_LOADED = True


def city_builder_table(kind):
    """NodeBuilderWithCity's citiesInfo in entrySet() order: kind "AWS" | "CITIES" -> (names, mercX, mercY, cumulative
    probability (float32), size of the city LIST the builder was given)"""
    load_city_data()
    cap = 1 << 16
    buf = C.create_string_buffer(cap)
    x, y, cum = np.zeros(512, np.int32), np.zeros(512, np.int32), np.zeros(512, np.float32)
    n, ls = C.c_int32(), C.c_int32()
    _ck(lib().orc_city_builder_table(0 if kind == "AWS" else 1, cap, buf, _p(x, C.c_int32), _p(y, C.c_int32),
                                     _p(cum, C.c_float), C.byref(n), C.byref(ls)))
    names = buf.value.decode().split("\n")[:n.value]
    return names, x[:n.value].copy(), y[:n.value].copy(), cum[:n.value].copy(), ls.value


def city_choose(kind, rd_int):
    load_city_data()
    idx = C.c_int32()
    _ck(lib().orc_city_choose(0 if kind == "AWS" else 1, C.c_int32(rd_int), C.byref(idx)))
    return idx.value


class LatencyModel:
    """a named NetworkLatency of the registry (C/RegistryNetworkLatencies.java), probed between two cities"""

    def __init__(self, name):
        load_city_data()
        self.h = C.c_void_p()
        _ck(lib().orc_latency_model_create(name.encode(), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_latency_model_destroy(self.h)
            self.h = None

    def city(self, city_from, city_to, delta, same=False, e1=0, e2=0):
        out = C.c_int32()
        _ck(lib().orc_latency_model_city(self.h, city_from.encode(), city_to.encode(), e1, e2, int(same), delta,
                                         C.byref(out)))
        return out.value
