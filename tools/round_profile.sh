#!/bin/bash
# GPU box: the round's evidence in one call -> gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards
# (tools/digest_profile.py <tag> <tag>; the phase profiles and config B's summary are copied as they are).
# usage: bash tools/round_profile.sh <tag>     (build the stamped libraries first: make -C alphazero_gym_amd/csrc libazgym_hip_stamp.so libazgym_hip_stampa.so)
TAG=${1:-r04}
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
python3 tools/team_profile.py 1024 > $OUT/${TAG}_team_profile.txt 2>&1
# rocprofv3: kernel stats + PMC passes of the headline command, and of config B (with the HBM passes)
bash tools/profile_bench.sh $TAG > $OUT/${TAG}_profile_bench.log 2>&1
bash tools/profile_config_b.sh $TAG > $OUT/${TAG}_profile_b.log 2>&1
bash tools/profile_config_e.sh $TAG > $OUT/${TAG}_profile_e.log 2>&1
tail -n 25 $OUT/${TAG}_phase_C.txt
