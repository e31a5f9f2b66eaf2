cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do
for cfg in "4 1" "8 1" "8 2"; do set -- $cfg; for b in 4096 8192; do echo "waves=$1 groups=$2 trees=$b"; AZG_WAVES=$1 AZG_GROUPS=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --trees $b 2>&1 | tail -1 | cut -c84-200; done; done; done
