#!/bin/bash
# A/B timing of engine builds on ONE GPU box (box-to-box variance is 1-2 %, same-box repeatability about 0.1 %).
# Build each variant in-tree next to the product library:
#   make -j8 -C alphazero_gym_amd/csrc variant NAME=foo DEFS=-DAZG_X_FOO      ->  libazgym_hip_x_foo.so
# (or copy the current libazgym_hip.so to libazgym_hip_x_prev.so before changing the sources), then:
#   gpurun -- 'bash tools/ab_variants.sh base foo ...'
# Per variant: the GPU test suite's verdict, two headline bench runs (ms per search), configs B and E.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for v in "$@"; do
  if [ $v = base ]; then unset AZG_HIP_LIB; else export AZG_HIP_LIB=$PWD/alphazero_gym_amd/csrc/libazgym_hip_x_$v.so; fi
  r=$(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -1)
  for i in 1 2; do b=$(python bench.py --no-cpu-baseline --steps 60 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"); r="$r | $b"; done
  c=$(python tools/bench_configs.py B E 2>&1 | tail -2 | cut -c52-90 | tr '\n' ' ')
  echo "$v: $r | $c"
done
