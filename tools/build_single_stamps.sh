#!/bin/bash
# Builds the single-stamp-pair libraries (alphazero_gym_amd/csrc/Makefile: ss) for tools/single_stamp_profile.sh, eight at a time.
# usage: bash tools/build_single_stamps.sh [TU ...]   (default: dispatch_pendulum_large = config C's kernels)
cd $(dirname $0)/../alphazero_gym_amd/csrc
TUS_="${*:-dispatch_pendulum_large}"
make -j8 libazgym_hip.so > /tmp/azg_build.log 2>&1 || { grep -m5 -A5 "error" /tmp/azg_build.log; echo "BUILD FAILED"; exit 1; }
n=0
for s in 0 1 2 3 4 5 6 11 12 13 14 15; do
  make ss SLOT=$s SSTUS="$TUS_" > /dev/null 2>&1 &
  n=$((n + 1)); if [ $((n % 8)) = 0 ]; then wait; fi
done
for s in 4 5; do make ss SLOT=$s SSTAG=a SSDEFS=-DAZG_STAMPS_A SSTUS="$TUS_" > /dev/null 2>&1 & done
wait
ls -1 libazgym_hip_ss*.so
