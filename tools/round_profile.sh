#!/bin/bash
# GPU box: the round's evidence in one call -> gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards
# (tools/digest_profile.py <tag> <tag>; the phase profiles and config B's summary are copied as they are).
# usage: bash tools/round_profile.sh <tag>     (build the stamped libraries first: make -C alphazero_gym_amd/csrc libazgym_hip_stamp.so libazgym_hip_stampa.so)
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
# in-kernel phase stamps (cycles per simulation step and phase): configs C, B and C at 8192 trees
python3 tools/phase_profile.py pendulum 4096 > $OUT/${TAG}_phase_C.txt 2>&1
python3 tools/phase_profile.py cartpole 4096 > $OUT/${TAG}_phase_B.txt 2>&1
python3 tools/phase_profile.py pendulum 8192 > $OUT/${TAG}_phase_C8192.txt 2>&1
# the same with the stamp slots of the network's parts given to tree phase A's (finish leaf | backup | re-scoring)
if [ -f alphazero_gym_amd/csrc/libazgym_hip_stampa.so ]; then
  python3 tools/phase_profile.py pendulum 4096 --phase-a > $OUT/${TAG}_phase_C_treeA.txt 2>&1
  python3 tools/phase_profile.py cartpole 4096 --phase-a > $OUT/${TAG}_phase_B_treeA.txt 2>&1
fi
TEAM_LIB=libazgym_hip_stampa.so python3 tools/team_profile.py 1024 > $OUT/${TAG}_team_profile.txt 2>&1
TEAM_LIB=libazgym_hip_stampa.so python3 tools/team_profile.py 512 > $OUT/${TAG}_team_profile_alone.txt 2>&1
# every part of a step with ONE stamp pair each (builds within a few per cent of the product; tools/build_single_stamps.sh): configs C and B
if [ -f alphazero_gym_amd/csrc/libazgym_hip_ss13.so ]; then
  bash tools/single_stamp_profile.sh pendulum 4096 > $OUT/${TAG}_phase_C_single.txt 2>&1
  bash tools/single_stamp_profile.sh cartpole 4096 > $OUT/${TAG}_phase_B_single.txt 2>&1
fi
# the wide network's hidden-layer tiles by themselves (tools/probes/tile8/README.md)
if [ -x tools/probes/tile8/tile8_probe ]; then
  for d in 1 32; do echo "== activation blocks / $d"; tools/probes/tile8/tile8_probe 200 $d; done > $OUT/${TAG}_tile_probe.txt 2>&1
fi
# BASELINE shapes with the general kernels beside the compile-time specialised ones, same box
( python3 tools/quick_times.py C B C8192 B8192 E X1536 E2048 E3072 X4096; echo "== AZG_NO_SPEC=1"; AZG_NO_SPEC=1 python3 tools/quick_times.py C B C8192 B8192 ) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_quick_times.txt
# end-to-end learning on one GPU (examples/selfplay_train.py)
python3 examples/selfplay_train.py --game CartPole-v0 --games 512 --n-rollouts 32 --iters 80 2>/dev/null | grep "^{" > $OUT/${TAG}_learning_cartpole.jsonl
python3 examples/selfplay_train.py --game Pendulum-v1 --games 512 --n-rollouts 50 --iters 40 --steps-per-iter 200 --train-rows 16384 --batch-size 128 2>/dev/null | grep "^{" > $OUT/${TAG}_learning_pendulum.jsonl
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# rocprofv3: kernel stats + PMC passes of the headline command, and of config B (with the HBM passes)
bash tools/profile_bench.sh $TAG > $OUT/${TAG}_profile_bench.log 2>&1
bash tools/profile_config_b.sh $TAG > $OUT/${TAG}_profile_b.log 2>&1
bash tools/profile_config_e.sh $TAG > $OUT/${TAG}_profile_e.log 2>&1
tail -n 25 $OUT/${TAG}_phase_C.txt
