#!/bin/bash
# GPU box: rocprofv3 kernel stats + counters of the lock-step path (BASELINE config E) -> gpurun_out/prof_e_<tag>/
# usage: bash tools/profile_config_e.sh <tag> [lib]     (lib: a variant library under alphazero_gym_amd/csrc, e.g. libazgym_hip_x_foo.so)
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
if [ -n "$2" ]; then export AZG_HIP_LIB=$REPO/alphazero_gym_amd/csrc/$2; fi
OUT=$REPO/gpurun_out/prof_e_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/time_e.py > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/tools/time_e.py > $OUT/pmc.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc_wait -- python3 $REPO/tools/time_e.py > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_insts -- python3 $REPO/tools/time_e.py > $OUT/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/time_e.py > $OUT/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/time_e.py > $OUT/pmc5.log 2>&1
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-160 | head -6
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_mfma", "pmc_wait", "pmc_insts", "pmc_fetch", "pmc_write"):
    fs = sorted(glob.glob("$OUT/" + d + "/*/*_counter_collection.csv"))
    if not fs:
        print("no counters in", d)
        continue
    for r in csv.DictReader(open(fs[-1])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_summary.csv", "w") as o:
    o.write("kernel,counter,launches,mean_per_launch\n")
    for k, d in acc.items():
        if "rocclr" in k:
            continue
        for c, v in d.items():
            o.write(f'"{k}",{c},{len(v)},{sum(v)/len(v)}\n')
            print(k[:40], c, len(v), sum(v) / len(v))
PY
