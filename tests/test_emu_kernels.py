"""The product's kernel SOURCES (wittgenstein_amd/csrc/*.hip*) built for the CPU wave emulator in
tests/emu and run against the oracle — so that the engine's logic is exercised by `-m "not gpu"` in the
GPU-less build container too. Test infrastructure only: the package itself never loads the emulator
build (it has no CPU path), this module swaps it in explicitly and restores the real binding after.
The parity claims proper are the `-m gpu` runs of the same test bodies on the MI355X.

The emulator is stricter than the hardware in one respect: lanes of a wavefront do not advance in
lock-step between collectives, so a missing wave barrier shows up here as a mismatch."""
import ctypes as C
import os
import subprocess

import pytest

import wittgenstein_amd._lib as L

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EMU_LIB = os.path.join(EMU_DIR, "libwittgpu_emu.so")


@pytest.fixture(scope="module", autouse=True)
def emulated_kernels(oracle):
    subprocess.run(["make", "-s", "-C", EMU_DIR], check=True)
    saved = (L._lib, L.LIB_PATH)
    L._lib, L.LIB_PATH = None, EMU_LIB
    try:
        L.lib()
        yield
    finally:
        L._lib, L.LIB_PATH = saved


import test_gpu_batch as tb  # noqa: E402
import test_gpu_engine as te  # noqa: E402
import test_gpu_gsf as tg  # noqa: E402
import test_gpu_handel as th  # noqa: E402
import test_gpu_casper as tc  # noqa: E402
import test_gpu_p2pflood_resident as tfr  # noqa: E402
import test_gpu_casper_resident as tcr  # noqa: E402
import test_gpu_sanfermin_resident as tsr  # noqa: E402
import test_gpu_p2pflood as tpf  # noqa: E402
import test_gpu_optimistic_p2p as top  # noqa: E402
import test_gpu_slush as tsl  # noqa: E402
import test_gpu_paxos as tpx  # noqa: E402
import test_gpu_dfinity as tdf  # noqa: E402
import test_gpu_p2phandel as tph  # noqa: E402
import test_gpu_sanfermin as tsf  # noqa: E402
import test_gpu_fuzz as tf  # noqa: E402
import test_gpu_hostmode as thm  # noqa: E402
import test_gpu_snapshot as tsn  # noqa: E402
import test_gpu_city as tcy  # noqa: E402
import test_gpu_handel_hostmode as thh  # noqa: E402
import test_gpu_sanfermin_cappos as tsc  # noqa: E402
import test_gpu_wave_primitives as twp  # noqa: E402

ENGINE = ["test_simple_message_and_time", "test_register_task", "test_all_flavors_of_send",
          "test_multiple_message_with_delays", "test_delays_across_horizon_pages", "test_stats", "test_partitions",
          "test_task_on_stopped_node_and_periodic", "test_delivery_to_down_node_and_down_at_send",
          "test_argument_errors", "test_msg_discard_time", "test_full_bydistance_lut_matches_oracle"]


@pytest.mark.parametrize("name", ENGINE)
def test_engine_semantics(name):  # CT/NetworkTest.java restated through the C ABI
    getattr(te, name)()


@pytest.mark.parametrize("name", ["NetworkLatencyByDistanceWJitter", "IC3NetworkLatency", "NetworkFixedLatency(100)"])
def test_latency_models(name):
    te.test_latency_models_match_oracle(name)


def test_pingpong_reference_run():
    te.test_pingpong_reference_run()


def test_pingpong_chunking():
    te.test_pingpong_chunking_and_seeds(7)


def test_handel_handeltest_params_every_ms():  # PT/HandelTest.java:14-49 parameters
    th.test_handel_test_params_every_ms()


@pytest.mark.parametrize("n", [2, 8, 32])
def test_handel_tiny(n):
    th.test_tiny_networks(n)


def test_handel_256_chunks_of_10():
    th.lockstep(th.ratios(256), step=10)


def test_handel_incremental_check_sigs_wave_items(monkeypatch):  # cached evaluations on the wave-per-item paths, vs the oracle
    th.test_incremental_check_sigs_wave_items(monkeypatch, 512, "1", 1, 500, 0)


@pytest.mark.parametrize("cap,seed", [(0, 0), (256, 5)])
def test_handel_ranks_carried_by_the_senders(monkeypatch, capfd, cap, seed):  # no N x N matrix: initial rank from the sender + the receiver's bumps
    th.test_reception_ranks_carried_by_the_senders(monkeypatch, capfd, cap, seed)


def test_handel_ranks_matrix_form_and_bump_overflow(monkeypatch, capfd):
    th.test_reception_ranks_matrix_form_still_runs_with_device_init(monkeypatch, capfd)
    monkeypatch.delenv("WG_HANDEL_RANKS")
    th.test_rank_bump_table_overflow_is_loud()


def test_handel_emission_lists_on_the_device(monkeypatch):  # k_handel_init_sort / k_handel_init_shuffle vs the oracle's init()
    th.test_256_every_ms()
    th.test_emission_lists_built_on_the_device(1)
    th.test_emission_lists_fall_back_to_the_host(monkeypatch)


def test_handel_reception_ranks_on_the_device(monkeypatch, capfd):  # k_handel_init_scan / _perm / _chain vs the oracle's init()
    th.test_reception_ranks_shuffled_on_the_device(monkeypatch, capfd)


def test_handel_byzantine_suicide_resident():  # P/Handel.java:538-559, 577-584, 688-694 on the device vs the oracle
    th.test_byzantine_suicide_resident(64, 0)
    th.test_byzantine_suicide_resident_hostmode_cases()
    th.test_attack_parameter_checks()


def test_handel_attack_scenarios_wide_levels():
    th.test_attack_scenarios_resident_wide_levels(1024, "byzantine_suicide", 300)
    th.test_attack_scenarios_resident_wide_levels(1024, "hidden_byzantine", 400)


@pytest.mark.parametrize("mode", [None, "byzantine_suicide"])
def test_handel_explicit_bad_nodes_resident(mode):  # HandelParameters.badNodes, P/Handel.java:51, 960-964
    th.test_explicit_bad_nodes_resident(mode)


def test_handel_hidden_byzantine_resident():  # P/Handel.java:813-817, 840-917 on the device vs the oracle
    th.test_hidden_byzantine_resident((64, 50, 4, 50, 5, 20, 10, 6, 0), 2, 1)


def test_handel_device_init_forms_of_big_networks(monkeypatch):
    th.test_device_init_forms_of_more_than_65536_nodes(monkeypatch, 512, "1")
    th.test_device_init_forms_of_more_than_65536_nodes(monkeypatch, 1024, "2")


def test_handel_chunk_size_is_observable():
    th.test_chunk_size_is_observable_and_matches(7)


def test_handel_desynchronized_start():
    th.test_desynchronized_start_and_fast_pairing()


def test_handel_queue_overflow_is_loud():
    th.test_queue_capacity_overflow_is_loud()


def test_batch_every_ms():
    tb.test_batch_every_ms_lockstep_with_oracle()


def test_batch_run_multiple_times_on_device():  # wg_batch_run_multiple_times: the loop condition on the device
    tb.test_run_multiple_times_device_loop_equals_host_loop(64)  # (per-seed oracle runs: tests/test_gpu_graph.py)


def test_batch_of_nine_blocks_dealt_by_xcd():
    """engine_kernels.hip.h wg_place: from 8 members on a batch launch deals its blocks so that an engine stays on one XCD
    (wgBx / wgBy / wgGx instead of blockIdx / gridDim; XCD 0 gets two engines here, the division leaves blocks over) — every
    member still equals its own oracle run. (The plain mapping, WG_XCD_PLACE=0, is what every batch of fewer than 8 members runs.)"""
    tb.test_handel_batch_matches_oracle_per_seed(32, list(range(9)))


def test_gsf_batch_of_nine():  # GSFSignature copies batched: the deal by XCD, the envelopes' latency words
    tb.test_gsf_batch_matches_oracle_per_seed(32, list(range(40, 49)))


def test_graph_replay_keeps_the_profiler(monkeypatch):
    """WG_GRAPH=1: runMs(chunk) of a batch captured once and replayed; the profiler's brackets are then device clock stamps
    (Engine::ProfScope / k_prof_stamp) instead of HIP events — one span per simulated ms for the delivery pass, the same
    delivered count as the enqueued loop"""
    import wittgenstein_amd as w
    import parity
    res = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("WG_GRAPH", graph)
        sims = [parity.handel_pair((64, 57, 4, 50, 10, 20, 10, 6, 0), seed=s)[0] for s in (0, 1)]
        sims[0].network().profile(2)
        d, ms = w.Batch([g.network() for g in sims]).run_multiple_times(chunk=10, maxTime=20000)
        pr = sims[0].network().profile_read()["deliver"]
        res[graph] = (list(d), list(ms), pr["spans"])
        assert pr["spans"] >= max(ms) and pr["total_ns"] > 0, pr
    assert res["0"][:2] == res["1"][:2]


def test_batch_pingpong_active_mask():
    tb.test_pingpong_batch_and_active_mask()


def test_gsf_reference_parameters():  # PT/GSFSignatureTest.java parameters
    tg.test_reference_test_parameters_every_ms()


def test_gsf_dead_nodes_and_long_lists():
    tg.test_simple_threshold_with_dead_nodes()
    tg.lockstep((128, 96, 6, 10, 5, 10, 25), tg.NBG, total=300, config={"queue_cap": 256})


def test_gsf_without_the_rest_list(monkeypatch):
    tg.test_event_order_visit_of_every_active_node(monkeypatch)


def test_gsf_256_to_convergence():
    tg.test_256_chunks_of_10_to_convergence()


def test_gsf_multiword_levels_and_overflow():
    tg.lockstep((512, 500, 3, 50, 10, 10, 0), seed=5, step=10, total=150)
    tg.test_queue_capacity_overflow_is_loud()
    tg.test_unsupported_shapes_are_loud()


@pytest.mark.parametrize("name", ["test_register_task", "test_task_and_stopped_node", "test_periodic_task",
                                  "test_conditional_task", "test_multiple_destinations_with_delay_and_lifo"])
def test_host_callback_mode_reference_vectors(name):  # CT/NetworkTest.java through wg_next_delivery
    getattr(thm, name)()


@pytest.mark.parametrize("name", thm.BATCHED)
def test_batched_steps_reference_vectors(name, monkeypatch):  # ... and through wg_step_begin / wg_step_end
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    getattr(thm, name)()


def test_batched_steps_protocols(monkeypatch):
    monkeypatch.setenv("WG_HOST_BATCH", "1")
    thm.test_pingpong_through_host_callbacks_matches_oracle(120)
    tsf.test_sanfermin_64_matches_oracle()
    tpf.test_p2pflood_three_messages_by_distance()
    tpf.test_empty_destination_list_costs_a_draw()
    tc.lockstep((2, False, 2, 6, 1000, 1), seed=5, chunk=3000, chunks=3)
    tf.test_fuzz_partitions_stops_and_discard(2)
    thm.test_batched_step_errors_are_loud()


def test_handel_attack_scenarios_through_host_callbacks():  # P/Handel.java byzantineSuicide / hiddenByzantine vs the oracle
    thh.test_byzantine_suicide()
    thh.test_hidden_byzantine()
    thh.test_byzantine_suicide_desynchronized_on_batched_steps()


def test_sanfermin_cappos_through_host_callbacks():  # P/SanFerminCappos.java on the engine vs oracle/sanfermin_cappos.hpp
    tsc.test_cappos_64_matches_oracle()
    tsc.test_cappos_many_candidates_short_timeout_on_batched_steps()


def test_host_callback_mode_pingpong():
    thm.test_pingpong_through_host_callbacks_matches_oracle(120)


def test_host_callback_mode_releases_handles(monkeypatch):  # wg_host_released: a binding forgets a Message with its last envelope
    thm.test_handles_are_released_for_dropped_and_repeated_sends(monkeypatch)


def test_host_callback_mode_refused_send_and_deferred_init_rd():  # the handle of a refused send, rd at the end of deferred_init
    thm.test_a_refused_send_gives_its_handle_back()
    thm.test_deferred_init_leaves_rd_where_init_left_it()


def test_casper_through_host_callbacks():  # P/CasperIMD.java on the engine vs oracle/casper.hpp (the full cases: -m gpu)
    tc.lockstep((2, False, 2, 6, 1000, 1), seed=5, chunk=3000, chunks=3)


@pytest.mark.parametrize("byz", ["SF", "NS"])
def test_casper_other_byzantine_producers_through_host_callbacks(byz):  # P/CasperIMD.java:583-633 vs oracle/casper.hpp
    tc.test_casper_other_byzantine_producers(byz, 0, chunks=14)  # (the full cases: -m gpu)


@pytest.mark.parametrize("nl", [None, "NetworkNoLatency"])
def test_scheduler_fuzz_latency(nl):  # oracle/fuzz.hpp vs tests/fuzz_protocol.py on the engine
    tf.test_fuzz_latency_models(nl)


def test_scheduler_fuzz_partitions_stops_discard():
    tf.test_fuzz_partitions_stops_and_discard(2)


def test_sanfermin_through_host_callbacks():  # P/SanFerminSignature.java on the engine vs oracle/sanfermin.hpp
    tsf.test_sanfermin_64_matches_oracle()
    tsf.test_sanfermin_fixed_latency_short_timeout()


def test_p2phandel_through_host_callbacks(monkeypatch):  # P/P2PHandel.java over C/P2PNetwork.java; java.util.HashSet's order
    tph.test_p2phandel_default_shape()
    tph.test_p2phandel_runs_to_done_in_lockstep((20, 0, 20, 3, 2, 50, True, "cmp_diff", True))
    tph.test_p2phandel_runs_to_done_in_lockstep((40, 8, 36, 6, 3, 10, True, "cmp_all", False))
    tph.test_p2phandel_single_best_strategy_batched_steps(monkeypatch)
    tph.test_p2phandel_a_treeified_bucket_is_refused_like_the_oracle()


def test_dfinity_through_host_callbacks(monkeypatch):  # P/Dfinity.java over the block-chain classes
    tdf.test_dfinity_run()
    tdf.test_dfinity_rounds_of_committees_with_latency()
    tdf.test_dfinity_batched_steps(monkeypatch)


def test_handel_dissemination_phase_registered_late():
    th.test_dissemination_phase_registered_after_the_first_run()


def test_paxos_through_host_callbacks(monkeypatch):  # P/Paxos.java; init() sends between node constructions (deferred_init)
    tpx.test_paxos_simple()
    tpx.test_paxos_contended_and_copy((7, 5, 600), 2)
    tpx.test_paxos_timeouts_batched_steps(monkeypatch)


def test_slush_and_snowflake_through_host_callbacks(monkeypatch):  # P/Slush.java, P/Snowflake.java
    tsl.test_slush_simple()
    tsl.test_snowflake_simple()
    tsl.test_copies_agree_and_match_the_oracle(True, (60, 5, 7, 4.0 / 7.0, 3))
    tsl.test_slush_batched_steps_without_latency(monkeypatch)


def test_optimistic_p2p_signature_through_host_callbacks(monkeypatch):  # P/OptimisticP2PSignature.java over C/P2PNetwork.java
    top.test_optimistic_p2p_simple()
    top.test_envelope_ring_overflow_is_loud()
    top.test_optimistic_p2p_without_latency_batched_steps(monkeypatch)


def test_p2pflood_through_host_callbacks():  # C/P2PNetwork.java + FloodMessage + P/P2PFlood.java on the engine
    tpf.test_p2pflood_three_messages_by_distance()
    tpf.test_empty_destination_list_costs_a_draw()


def test_sendall_expanded_on_the_device():
    import test_gpu_send_expand as tse
    tse.test_sendall_expanded_on_the_device_many_tiles()


@pytest.mark.parametrize("n", [2, 8])
def test_sanfermin_resident_tiny(n):  # P/SanFerminSignature.java resident on the device vs oracle/sanfermin.hpp
    tsr.test_tiny_networks(n)


@pytest.mark.parametrize("cand", [1, 3])
def test_sanfermin_resident_candidates(cand):
    tsr.test_candidate_counts_shuffle_draws(cand)


def test_sanfermin_resident_fixed_latency():
    tsr.test_fixed_latency_short_timeout_and_threshold()


def test_casper_resident():  # P/CasperIMD.java resident on the device vs oracle/casper.hpp (two blocks, one WF far task)
    # (block construction 100 ms + a fixed latency keep the bucket ring at 256 ms: the emulator pays per simulated ms)
    tcr.lockstep((2, False, 2, 6, 100, 1), seed=5, chunk=1500, chunks=18, nl="NetworkFixedLatency(20)")


def test_casper_resident_random_on_ties():  # k_casper_mark / k_casper_seq: ties that draw, vs the oracle after every chunk
    tcr.random_on_ties_cases(long=False)


def test_casper_resident_two_blocks_in_one_ms():  # a delayed byzantine build in another producer's ms: k_casper_seq
    tcr.two_blocks_in_one_ms_cases()


def test_casper_resident_stopped_attesters():  # config 5's "+10 %" (SURVEY.md §8d): attesters stop()ped after init()
    tcr.lockstep((2, False, 2, 10, 100, 1), seed=7, chunk=1500, chunks=12, nl="NetworkFixedLatency(20)", stopped=2)


def test_long_chain_runs_one_wavefront_each(monkeypatch):  # k_expand_runs, forced on networks this small
    monkeypatch.setenv("WG_RUN_MIN", "2")
    tcr.lockstep((2, False, 2, 10, 100, 1), seed=3, chunk=1500, chunks=8, nl="NetworkFixedLatency(20)", stopped=1)
    th.lockstep(th.ratios(128), step=10)  # Handel fast-path envelopes (<= 64 destinations) through the same path


def test_p2pflood_resident():  # P/P2PFlood.java resident on the device vs oracle/p2pflood.hpp
    tfr.test_three_messages_by_distance()
    tfr.lockstep((64, 0, 5, 2, 1, 12, 1), "NetworkFixedLatency(7)", seed=9, chunk=1, chunks=150)


@pytest.mark.parametrize("name", ["test_handel_restore_equals_fresh_init", "test_snapshot_only_before_the_first_event",
                                  "test_batch_restore_and_run_multiple_times_again", "test_pingpong_and_gsf_restore"])
def test_snapshot_restore(name):  # wg_snapshot / wg_restore: the init() image
    getattr(tsn, name)()


def test_city_latency_models_and_builders():  # C/NetworkLatency.java:86-233, C/NodeBuilder.java:98-147 on the engine
    for m in ["AwsRegionNetworkLatency", "NetworkLatencyByCity", "NetworkLatencyByCityWJitter"]:
        tcy.test_city_latency_probe_matches_the_oracle(m)
    tcy.test_pingpong_on_city_nodes("CITIES_SPEED=CONSTANT_TOR=0.00", "NetworkLatencyByCityWJitter")
    tcy.test_pingpong_on_city_nodes("AWS_SPEED=CONSTANT_TOR=0.00", "AwsRegionNetworkLatency")
    tcy.test_city_latency_needs_city_nodes()


def test_wave_primitives_shuffle_forms():  # wg_selftest on the emulator: the non-DPP forms of the reductions / scans / group-of-eight steps
    twp.test_wave_reductions_and_scans(64)
    twp.test_wave_reductions_and_scans(17)
    twp.test_lane_broadcasts_and_shuffles()
    twp.test_groups_of_eight_lanes(60)
    twp.test_block_scan_and_sum(256, 200)
    twp.test_ballot_multisplit_rank(3, 256, 256)
