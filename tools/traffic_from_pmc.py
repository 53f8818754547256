#!/usr/bin/env python
"""profiles/traffic.json from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench workload.

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch. MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE =
TCC_EA0_RDREQ x 64 B and undercounts 128-byte requests by half (calibrated on wide coalesced reads); other widths and
WRITE_SIZE are "uncalibrated: calibrate on a known byte count in your own access pattern". That calibration is
tools/micro/pmc_calib.hip (profiles/archive/r02b_pmc_calib.txt), for the access shapes of the delivery kernels:
  reads   one 64-byte line per request (8 lanes x 8 B, or 4 lanes x 16 B): FETCH_SIZE == bytes (x1.00);
          a lane alone in its line (8 or 16 B used): 64 B counted per lane (the line is fetched);
          wide coalesced 16 B/lane streams: x0.50 (the guide's case)
  writes  full lines: WRITE_SIZE == bytes (x1.00, also for the coalesced stream);
          a lane alone in its line (8 or 16 B): 32 B counted per lane (32-byte write granules)
So for these kernels (scattered lines and lane-scattered words, no wide streams) the counters are the memory-side
bytes as they stand: `hbm_bytes_per_launch` = FETCH_SIZE + WRITE_SIZE; `..._reads_doubled` keeps the guide's x2 on
the reads as the upper bound (it would apply only to the part of the reads that are 128-byte requests).
usage: traffic_from_pmc.py <pmc_FETCH_SIZE.md> <pmc_WRITE_SIZE.md> <nodes> <replicas> <out.json> [kernel patterns, comma-separated
                           [<pmc_req.md>]]
(default patterns: Handel's delivery pass — the six kernels bench.py's HIP events bracket; Casper's:
"k_casper_classify,k_casper_attestations,k_deliver<CasperProto"; GSFSignature's: "k_gsf_docycle,k_gsf_lane,k_deliver_inbox<GsfProto")
<replicas> may be the JSON line a PMC pass's bench.py printed (its config.replicas_per_gpu is taken).
<pmc_req.md>: a third pass with TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum (tools/pmc_issue.sh `req`): the memory-side REQUEST
counts — what the chip's scattered-line ceiling (tools/micro/mlp_probe) is a rate of; FETCH_SIZE / 64 overstates them where a kernel
streams (a 128-byte request counts once) — as `ea_requests_per_launch` / `whole_step_ea_requests` (reads + writes)."""
import json
import re
import sys


def per_dispatch(path, counter, kernels):
    """KB per delivery PASS of each kernel: its sum over the run / the number of passes (= the dispatch count of the kernels
    that are launched in every pass). A kernel that is not launched in every pass — k_handel_dissem since round 4: only in a
    ms whose phase some node's dissemination has — must not contribute its mean per OWN dispatch: the HIP events of
    `avg_launch_us` bracket every pass, with or without it."""
    rows = {}
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 5 and c[1] == counter:
            for k in kernels:  # (a pattern that ends in a name character matches whole names only: k_handel_lane is not k_handel_lane2)
                if re.search(re.escape(k) + (r"(?![A-Za-z0-9_])" if re.match(r"[A-Za-z0-9_]", k[-1]) else ""), c[0]):
                    n, sm = rows.get(k, (0, 0.0))
                    rows[k] = (n + int(float(c[2])), sm + float(c[3]))
    passes = max([n for n, _ in rows.values()] or [1])
    return {k: sm / passes for k, (n, sm) in rows.items()}


def whole_step(path, counter):
    """memory-side KB of ONE step (the passes run `bench.py --steps 1 --warmup 0`): the `sum` column over every per-ms kernel —
    everything but init()'s kernels and the runtime's fill / copy kernels (uploads, wg_restore), which sit outside the timed region"""
    tot, per = 0.0, {}
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 5 and c[1] == counter and "_init" not in c[0] and "__amd_rocclr" not in c[0]:
            tot += float(c[3])
            per[c[0].split("(")[0].replace("void ", "")] = float(c[3])
    return tot, per


def replicas_of(arg):
    if arg.isdigit():
        return int(arg)
    line = [l for l in open(arg).read().strip().splitlines() if l.startswith("{")][-1]
    return int(json.loads(line)["config"]["replicas_per_gpu"])


def main():
    fetch, write, nodes, replicas, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), replicas_of(sys.argv[4]), sys.argv[5]
    kernels = sys.argv[6].split(",") if len(sys.argv) > 6 and sys.argv[6] else ["k_handel_lane", "k_handel_update<", "k_handel_lane2", "k_handel_copy", "k_handel_dissem<", "k_handel_wave<"]
    f = per_dispatch(fetch, "FETCH_SIZE", kernels)
    w = per_dispatch(write, "WRITE_SIZE", kernels)
    rd = sum(f.values()) * 1024.0
    wr = sum(w.values()) * 1024.0
    ws_f, ws_f_per = whole_step(fetch, "FETCH_SIZE")
    ws_w, ws_w_per = whole_step(write, "WRITE_SIZE")
    import os
    req = {}
    if len(sys.argv) > 7:
        rq = {c: per_dispatch(sys.argv[7], c, kernels) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_REQ_sum")}
        ws = {c: whole_step(sys.argv[7], c) for c in rq}
        req = {"ea_requests_per_launch": sum(rq["TCC_EA0_RDREQ_sum"].values()) + sum(rq["TCC_EA0_WRREQ_sum"].values()),
               "ea_read_requests_per_launch": sum(rq["TCC_EA0_RDREQ_sum"].values()),
               "ea_write_requests_per_launch": sum(rq["TCC_EA0_WRREQ_sum"].values()),
               "l2_requests_per_launch": sum(rq["TCC_REQ_sum"].values()),
               "whole_step_ea_requests": ws["TCC_EA0_RDREQ_sum"][0] + ws["TCC_EA0_WRREQ_sum"][0],
               "whole_step_l2_requests": ws["TCC_REQ_sum"][0],
               "per_kernel_requests": rq,
               "requests_note": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum in one further pass: requests "
                                "the L2 sent to the memory side (EA) and requests the L2 received, per delivery pass / per step"}
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from wittgenstein_amd.replicas import csrc_hash  # (the sources these counters were taken on: bench.py compares it with its own tree)
    json.dump({**req, "nodes": nodes, "replicas": replicas, "commit": os.environ.get("WG_COMMIT", "unknown"), "csrc_sha": csrc_hash(),
               "kernels": " + ".join(kernels) + " (one launch of each per simulated ms that is not skipped)",
               "fetch_bytes_per_launch_raw": rd, "write_bytes_per_launch_raw": wr,
               "hbm_bytes_per_launch": rd + wr, "hbm_bytes_per_launch_reads_doubled": 2.0 * rd + wr,
               "per_kernel_KB": {"FETCH_SIZE": f, "WRITE_SIZE": w},
               "whole_step_hbm_bytes": (ws_f + ws_w) * 1024.0,
               "whole_step_fetch_bytes": ws_f * 1024.0, "whole_step_write_bytes": ws_w * 1024.0,
               "whole_step_per_kernel_KB": {"FETCH_SIZE": ws_f_per, "WRITE_SIZE": ws_w_per},
               "whole_step_note": "every per-ms kernel of one step (one RunMultipleTimes pass of the batch), init() and the "
                                  "runtime's fill / copy kernels excluded",
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per dispatch means over the run; "
                       "calibrated on known byte counts in these access shapes (profiles/archive/r02b_pmc_calib.txt): scattered "
                       "64-byte-line reads and full-line writes count x1.00, a lane alone in its line counts the 64-byte "
                       "line (read) / a 32-byte granule (write); the guide's x2 (128-byte requests) is kept as "
                       "..._reads_doubled, an upper bound"},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
