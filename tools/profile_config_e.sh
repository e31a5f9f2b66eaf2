#!/bin/bash
# GPU box: rocprofv3 kernel stats + MFMA counters of the lock-step path (BASELINE config E) -> gpurun_out/prof_e_<tag>/
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_e_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_configs.py E > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/tools/bench_configs.py E > $OUT/pmc.log 2>&1
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-160 | head -6
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$OUT/pmc_mfma/*/*_counter_collection.csv"))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_summary.csv", "w") as o:
    o.write("kernel,counter,launches,mean_per_launch\n")
    for k, d in acc.items():
        for c, v in d.items():
            o.write(f'"{k}",{c},{len(v)},{sum(v)/len(v)}\n')
            print(k, c, len(v), sum(v) / len(v))
PY
