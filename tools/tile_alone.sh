#!/bin/bash
# GPU box: how fast does ONE tile workgroup per CU run?  Per-layer launches (AZG_LS_TEAM=0) at 512 trees = 256 tiles per layer = one
# workgroup per CU (1024 trees: two per CU); rocprofv3 kernel stats give the layer kernel's average duration.
# usage: bash tools/tile_alone.sh <variant library suffix or "base"> [trees]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-base}; T=${2:-512}
if [ $V != base ]; then export AZG_HIP_LIB=$REPO/alphazero_gym_amd/csrc/libazgym_hip_x_$V.so; fi
export AZG_LS_TEAM=0 E_TREES=$T
OUT=/tmp/tile_alone_${V}_${T}; rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/time_e.py > $OUT.log 2>&1
grep "ms/search" $OUT.log
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ls_" in r["Name"]:
            print(f'  {r["Name"][:52]:52s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1e3:8.2f} us')
PY
