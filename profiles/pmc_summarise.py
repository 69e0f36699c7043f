#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass) of
`bench.py --steps 1 --warmup 0 --no-cpu-baseline` into profiles/r01_pmc_traffic_<workload>.json,
the file bench.py reads `roofline.traffic` from.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python profiles/pmc_summarise.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv \
        hqc128_minsum 4096 128 204000

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section; re-verified on
profiles/microbench/rmw_stream with 4 / 8 / 16 B per lane): FETCH_SIZE and WRITE_SIZE are in KiB;
gfx950 tallies 128-byte read requests at 64 B, so FETCH_SIZE is doubled; WRITE_SIZE is exact.
These are L2 <-> fabric bytes: Infinity-Cache hits are counted (an upper bound on HBM bytes).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    """Average per dispatch, over the dispatches of a kernel's MOST COMMON grid size: the run also
    holds bench.py's `isolated` re-measurement, whose launches sweep the whole tile group (twice the
    codewords of the schedule's own launches)."""
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            key = (name, r["Grid_Size"])
            tot[key] += float(r["Counter_Value"])
            cnt[key] += 1
    out = {}
    for (name, grid), n in cnt.items():
        if name not in out or n > out[name][1]:
            out[name] = (tot[(name, grid)] / n, n)
    return out


def main():
    fpath, wpath, workload, batch, swept, E = sys.argv[1:7]
    batch, swept, E = int(batch), int(swept), int(E)
    fetch = per_kernel(fpath, "FETCH_SIZE")
    write = per_kernel(wpath, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fb = fetch.get(k, (0.0, 0))[0] * 1024.0 * 2.0
        wb = write.get(k, (0.0, 0))[0] * 1024.0
        kernels[k] = {"fetch_bytes": fb, "write_bytes": wb, "traffic_bytes": fb + wb,
                      "dispatches": max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1])}
    out = {
        "workload": workload,
        "batch": batch,
        "tile_group_codewords": swept,
        "algorithmic_bytes_per_launch": 8 * E * swept,
        "corrections": {
            "FETCH_SIZE": "KiB, x2 (gfx950 tallies 128-B requests at 64 B; verified on profiles/microbench/rmw_stream for 4/8/16 B per lane: ratio 0.500)",
            "WRITE_SIZE": "KiB, x1 (verified: ratio 1.000)",
        },
        "note": "TCC_EA (L2<->fabric) counters: Infinity-Cache hits are counted, so this is L2-miss traffic, an upper bound on HBM bytes; "
                "averages per dispatch over all dispatches of a kernel in the run (two-stream schedule: launches of 128 codewords)",
        "kernels": kernels,
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"r01_pmc_traffic_{workload}.json")
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:50s} {v['traffic_bytes'] / 1e6:10.2f} MB  x{v['traffic_bytes'] / (8 * E * swept):.3f} of algorithmic  ({v['dispatches']} dispatches)")


if __name__ == "__main__":
    main()
