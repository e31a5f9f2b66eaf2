#!/bin/bash
# GPU box: rocprofv3 kernel stats + instruction counters of BASELINE config B (CartPole, 4096 trees x 100 sims, 2x128) -> gpurun_out/prof_b_<tag>/
# usage: bash tools/profile_config_b.sh <tag>
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_b_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_configs.py B > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_insts -- python3 $REPO/tools/bench_configs.py B > $OUT/pmc.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/tools/bench_configs.py B > $OUT/pmc2.log 2>&1
# HBM side (the tree-walk-bound config's roofline): separate passes, KiB per launch
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/bench_configs.py B > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/bench_configs.py B > $OUT/pmc4.log 2>&1
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-170 | head -4
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for d in ("pmc_insts", "pmc_mfma", "pmc_fetch", "pmc_write"):
    for f in glob.glob("$OUT/" + d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "search_kernel" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/pmc_summary.csv", "w") as o:
    o.write("kernel,counter,launches,mean_per_launch\n")
    for (k, c), v in sorted(acc.items()):
        o.write('"%s",%s,%d,%.1f\n' % (k, c, len(v), sum(v) / len(v)))
print(open("$OUT/pmc_summary.csv").read())
PY
