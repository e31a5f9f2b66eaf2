cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "cfg18 or cfg19 or cfg20 or cfg21" 2>&1 | tail -1
AZG_HIP_LIB=$PWD/alphazero_gym_amd/csrc/libazgym_hip_x_prev.so python tools/bench_configs.py E | tail -1
python tools/bench_configs.py E | tail -1
