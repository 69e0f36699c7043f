#!/bin/bash
# round 3: L2 (TCC) hit / miss counts of the record-form kernels on the HQC-128 bench geometry (two counters per pass:
# more in one pass exceed what the hardware can collect and rocprofv3 aborts)
export TMPDIR=/tmp; O=gpurun_out/r03at; mkdir -p $O
W=hqc128_minsum
timeout -k 5 90 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_$W -o t -- python3 bench.py --pmc-child --workload $W --batch 256 --pmc off > $O/tcc_$W.log 2>&1; echo "tcc $W rc=$?"
f=$(find $O/tcc_$W -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python profiles/sq_summarise.py $f $O/tcc_counters_${W}_record_form.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
python - <<PY
import json
try:
    d=json.load(open("$O/tcc_counters_${W}_record_form.json"))
    for k,v in d.items(): print(k, {a:round(b) for a,b in v["per_dispatch"].items()})
except Exception as e: print(e); print(open("$O/tcc_$W.log").read()[:600])
PY
