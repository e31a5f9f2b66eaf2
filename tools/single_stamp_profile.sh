#!/bin/bash
# GPU box: cycles per step of every part of a simulation step, each measured by a build that carries ONE stamp pair (a few per cent slower
# than the product; tools/build_single_stamps.sh builds them).  usage: bash tools/single_stamp_profile.sh [pendulum|cartpole] [trees]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
MODE=${1:-pendulum}; B=${2:-4096}
echo "== product library"; python3 tools/quick_times.py $([ $MODE = pendulum ] && echo C || echo B) 2>&1 | grep -v amdgpu.ids
for s in 0 1 4 12 5 6 2 4a 5a 3 13 11 14 15; do
  [ -f alphazero_gym_amd/csrc/libazgym_hip_ss$s.so ] || continue
  python3 tools/phase_profile.py $MODE $B --only $s 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | sed 's/kernel search_kernel<[^>]*>//; s/helper waves (4 of 8 per workgroup):/H:/; s/; the lines below are the WALKING waves//; s/over the walking waves//; s/single pair, //'; echo
done
