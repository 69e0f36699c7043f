#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* pass of `bench.py --pmc-child` -> per-kernel averages per dispatch and the
share of wave cycles parked (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY) and issuing
(SQ_ACTIVE_INST_ANY), as MI355X_MICROARCH.md's PMC section defines them (the three are disjoint and
add up to SQ_WAVE_CYCLES; all in quad-cycles).

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM \
        --output-format csv -d gpurun_out/sq -o sq -- python3 bench.py --pmc-child --workload hqc128_minsum --batch 256 --pmc off
    python profiles/sq_summarise.py gpurun_out/sq/sq_counter_collection.csv profiles/r02/sq_counters_hqc128_minsum.json
"""
import csv
import json
import re
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1:3]
    prefixes = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("k_var", "k_check")
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(src)):
        name = re.sub(r"^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0]
        if not name.startswith(prefixes) or re.search(r"^k_check[^<]*<[^,>]+, true", name):
            continue
        key = (name, r["Grid_Size"])
        tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
    best = {}
    for (name, grid), c in cnt.items():
        n = max(c.values())
        if name not in best or n > best[name][1]:
            best[name] = (grid, n)
    out = {}
    for name, (grid, n) in best.items():
        avg = {k: tot[(name, grid)][k] / cnt[(name, grid)][k] for k in tot[(name, grid)]}
        wc = avg.get("SQ_WAVE_CYCLES", 0.0)
        row = {"grid": int(grid), "dispatches": n, "per_dispatch": avg}
        if wc:
            row["share_parked_on_waitcnt"] = avg.get("SQ_WAIT_ANY", 0.0) / wc
            row["share_issue_stalled"] = avg.get("SQ_WAIT_INST_ANY", 0.0) / wc
            row["share_issuing"] = avg.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
            if avg.get("SQ_WAVES"):
                row["quad_cycles_per_wave"] = wc / avg["SQ_WAVES"]
                row["valu_instructions_per_wave"] = avg.get("SQ_INSTS_VALU", 0.0) / avg["SQ_WAVES"]
                if "SQ_INSTS_LDS" in avg:
                    row["lds_instructions_per_wave"] = avg["SQ_INSTS_LDS"] / avg["SQ_WAVES"]
        out[name] = row
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "per_dispatch"})


if __name__ == "__main__":
    main()
