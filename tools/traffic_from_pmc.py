#!/usr/bin/env python
"""profiles/traffic.json from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench workload.

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch. MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE =
TCC_EA0_RDREQ x 64 B and undercounts 128-byte requests by half (calibrated on wide coalesced reads); other widths and
WRITE_SIZE are uncalibrated. The delivery kernels issue mostly scattered 64-byte accesses, so both readings are kept:
`hbm_bytes_per_launch` applies the guide's x2 to the reads (an upper bound), `..._uncorrected` does not.
usage: traffic_from_pmc.py <pmc_FETCH_SIZE.md> <pmc_WRITE_SIZE.md> <nodes> <replicas> <out.json>"""
import json
import sys


def per_dispatch(path, counter, kernels):
    tot = {}
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 5 and c[1] == counter:
            for k in kernels:
                if k in c[0]:
                    tot[k] = tot.get(k, 0.0) + float(c[4])
    return tot


def main():
    fetch, write, nodes, replicas, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    kernels = ["k_deliver_msgs<", "k_deliver<"]
    f = per_dispatch(fetch, "FETCH_SIZE", kernels)
    w = per_dispatch(write, "WRITE_SIZE", kernels)
    rd = sum(f.values()) * 1024.0
    wr = sum(w.values()) * 1024.0
    json.dump({"nodes": nodes, "replicas": replicas,
               "kernels": "k_deliver_msgs<HandelProto> + k_deliver<HandelProto> (one launch of each per simulated ms)",
               "fetch_bytes_per_launch_raw": rd, "write_bytes_per_launch_raw": wr,
               "hbm_bytes_per_launch": 2.0 * rd + wr, "hbm_bytes_per_launch_uncorrected": rd + wr,
               "per_kernel_KB": {"FETCH_SIZE": f, "WRITE_SIZE": w},
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; reads doubled per the guide's gfx950 "
                       "correction (upper bound for 64-byte scattered reads); per dispatch means over the run"},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
