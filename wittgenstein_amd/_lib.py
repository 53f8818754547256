"""ctypes binding of libwittgpu.so (include/wittgpu.h, include/wittgpu_host.h).

There is no CPU fallback: if the shared library is missing or no HIP device is visible the
package raises instead of computing anything on the host.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WG_LIB selects an investigation build (tools/kprof.sh); the product library otherwise
LIB_PATH = os.environ.get("WG_LIB") or os.path.join(_HERE, "libwittgpu.so")


class wg_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("horizon_ms", C.c_int32), ("bucket_pool_records", C.c_int64),
                ("payload_words", C.c_int64), ("outbox_records", C.c_int64), ("chain_dests", C.c_int64),
                ("chain_slots", C.c_int32), ("queue_cap", C.c_int32),
                ("shard", C.c_int32), ("nshards", C.c_int32), ("allreduce", C.c_void_p), ("allreduce_ctx", C.c_void_p),
                ("rccl_id", C.c_void_p), ("queue_cap_wide", C.c_int32), ("rank_bump_cap", C.c_int32),
                ("alltoallv", C.c_void_p), ("alltoallv_ctx", C.c_void_p)]


def make_config(cfg):
    """wg_config from a dict of its fields; `rccl_id` may be the 128 bytes of shards.rccl_unique_id()"""
    c = wg_config()
    for k, v in (cfg or {}).items():
        if k == "rccl_id" and isinstance(v, (bytes, bytearray)):
            c._rccl_keep = (C.c_uint8 * 128).from_buffer_copy(bytes(v))  # must outlive the wg_create call
            v = C.addressof(c._rccl_keep)
        setattr(c, k, v)
    return c


# int32_t (*wg_allreduce_fn)(void* ctx, void* buf, int64_t count)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64)
# int32_t (*wg_alltoallv_fn)(void* ctx, const void* sendbuf, const int64_t* send_counts, const int64_t* send_offsets,
#                            void* recvbuf, const int64_t* recv_counts, const int64_t* recv_offsets)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                           C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class wg_handel_params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nodeCount", "threshold", "pairingTime", "levelWaitTime", "extraCycle", "disseminationPeriodMs", "fastPath",
        "nodesDown", "desynchronizedStart", "windowInitial", "windowMinimum", "windowMaximum", "byzantineSuicide",
        "hiddenByzantine")]


class wg_gsf_params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nodeCount", "threshold", "pairingTime", "timeoutPerLevelMs", "periodDurationMs", "acceleratedCallsCount",
        "nodesDown")]


class wg_sanfermin_params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nodeCount", "threshold", "pairingTime", "signatureSize", "replyTimeout", "candidateCount")]


class wg_casper_params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "cycleLength", "randomOnTies", "blockProducersCount", "attestersPerRound", "blockConstructionTime",
        "attestationConstructionTime", "byzDelay", "maxSlots")]


class wg_p2pflood_params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nodeCount", "deadNodeCount", "delayBeforeResent", "msgCount", "msgToReceive", "peersCount", "delayBetweenSends")]


class wg_delivery(C.Structure):
    _fields_ = [("kind", C.c_int32), ("time", C.c_int32), ("from_", C.c_int32), ("to", C.c_int32),
                ("msg", C.c_uint32), ("payload", C.c_uint32)]


class wg_step_op(C.Structure):
    _fields_ = [("after", C.c_int32), ("kind", C.c_int32), ("msg", C.c_uint32), ("payload", C.c_uint32),
                ("time", C.c_int32), ("from_", C.c_int32), ("to", C.c_int32), ("n", C.c_int32), ("delay", C.c_int32),
                ("seed", C.c_int32)]


class wg_run_stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("delivered", "tasks", "events", "draws", "simulated_ms", "wall_ns",
                                         "payload_bytes")]


class wg_profile_entry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("spans", C.c_int64), ("total_ns", C.c_double)]


# every symbol include/wittgpu.h and include/wittgpu_host.h declare
ABI_VERSION = 5  # WG_ABI_VERSION of the include/wittgpu.h these ctypes structures restate
ABI_SYMBOLS = [
    "wg_abi_version", "wg_abi_struct_size", "wg_selftest", "wg_read_i32",
    "wg_create", "wg_destroy", "wg_last_error", "wg_add_nodes", "wg_node_count", "wg_set_latency",
    "wg_set_latency_by_name", "wg_set_latency_city", "wg_latency_probe", "wg_set_partitions", "wg_set_node_down", "wg_set_discard_time",
    "wg_rng_set_seed", "wg_rng_get_state", "wg_rng_set_state", "wg_send", "wg_send_arrive_at", "wg_register_task",
    "wg_register_periodic_task", "wg_protocol_load", "wg_run_ms", "wg_time", "wg_queue_size", "wg_queue_size_at",
    "wg_read_i64", "wg_read_level_i32", "wg_read_bits", "wg_levels", "wg_device_bytes", "wg_delivered_by_level",
    "wg_protocol_cont_if", "wg_snapshot", "wg_restore", "wg_snapshot_bytes", "wg_shard_configure", "wg_shard_set_alltoallv", "wg_shard_configure_rccl", "wg_rccl_unique_id", "wg_shard_info", "wg_shard_traffic", "wg_next_delivery", "wg_step_begin", "wg_step_end", "wg_host_released", "wg_set_time", "wg_batch_create", "wg_batch_destroy", "wg_batch_last_error", "wg_batch_size",
    "wg_batch_run_ms", "wg_batch_cont_if", "wg_batch_run_multiple_times", "wg_profile_enable", "wg_profile_read", "wg_profile_set_reference", "wg_profile_read_spans",
    "wgh_pingpong_create", "wgh_handel_create", "wgh_handel_create_bad_nodes", "wgh_gsf_create", "wgh_sanfermin_create", "wgh_casper_create", "wgh_p2pflood_create", "wgh_register_city_builder", "wgh_register_city_latency", "wgh_last_error", "wgh_last_init_seconds", "wgh_last_init_on_device", "wgh_jrandom_ints",
    "wgh_jrandom_skip_ints", "wgh_jrandom_bounded",
]

WG_OK, WG_EINVAL, WG_ESTATE, WG_ENOMEM, WG_EHIP, WG_EUNSUPPORTED, WG_EHOSTINIT = 0, -1, -2, -3, -4, -5, -6

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(wittgenstein_amd has no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.wg_last_error.restype = C.c_char_p
        l.wg_last_error.argtypes = [C.c_void_p]
        l.wgh_last_error.restype = C.c_char_p
        l.wgh_last_init_seconds.restype = C.c_double
        l.wg_destroy.restype = None
        l.wg_destroy.argtypes = [C.c_void_p]
        l.wg_batch_last_error.restype = C.c_char_p
        l.wg_batch_last_error.argtypes = [C.c_void_p]
        l.wg_batch_destroy.restype = None
        l.wg_batch_destroy.argtypes = [C.c_void_p]
        # the structures above must be the library's (they have grown across versions): a stale library is a load error
        if not hasattr(l, "wg_abi_version") or l.wg_abi_version() != ABI_VERSION:
            raise ImportError("%s: ABI version %s, this binding is version %d — rebuild it (__graft_entry__.build())"
                              % (LIB_PATH, l.wg_abi_version() if hasattr(l, "wg_abi_version") else "< 5", ABI_VERSION))
        for k, t in enumerate((wg_config, wg_handel_params, wg_gsf_params, wg_casper_params, wg_sanfermin_params,
                               wg_p2pflood_params, wg_delivery, wg_step_op, wg_run_stats)):
            if l.wg_abi_struct_size(k) != C.sizeof(t):
                raise ImportError("%s: sizeof(%s) is %d in the library, %d in this binding" % (
                    LIB_PATH, t.__name__, l.wg_abi_struct_size(k), C.sizeof(t)))
        _lib = l
    return _lib
