cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset AZG_HIP_LIB; else export AZG_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_x_$v.so; fi
  r=$(timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "default or goldens" 2>&1 | tail -1)
  for i in 1 2; do b=$(python bench.py --no-cpu-baseline --steps 60 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"); r="$r | $b"; done
  c=$(python tools/bench_configs.py B 2>&1 | tail -1 | cut -c52-80)
  echo "$v: $r | $c"
done
