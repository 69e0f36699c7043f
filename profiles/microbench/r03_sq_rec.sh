#!/bin/bash
# round 3: SQ counters of the min-sum record-form kernels on the HQC-128 bench geometry
export TMPDIR=/tmp; O=gpurun_out/r03ao; mkdir -p $O
W=hqc128_minsum
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/sq_$W -o sq -- python3 bench.py --pmc-child --workload $W --batch 256 --pmc off > $O/sq_$W.log 2>&1; echo "sq $W rc=$?"
f=$(find $O/sq_$W -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python profiles/sq_summarise.py $f $O/sq_counters_${W}_record_form.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $O/sq2_$W -o sq -- python3 bench.py --pmc-child --workload $W --batch 256 --pmc off > $O/sq2_$W.log 2>&1; echo "sq2 $W rc=$?"
f=$(find $O/sq2_$W -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python profiles/sq_summarise.py $f $O/sq2_counters_${W}_record_form.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
python - <<PY
import json
for n in ("sq_counters","sq2_counters"):
    try: d=json.load(open("$O/%s_${W}_record_form.json"%n))
    except Exception as e: print(n, e); continue
    for k,v in d.items():
        print(k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items() if a!="per_dispatch"}); print("    ", {a:round(b) for a,b in v["per_dispatch"].items()})
PY
