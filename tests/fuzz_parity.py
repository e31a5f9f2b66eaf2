#!/usr/bin/env python3
"""One-off soak on the GPU box: the random-configuration parity test of tests/test_hip_parity.py over many more seeds
(python tests/fuzz_parity.py [first_seed] [count]); prints the seeds that disagree with the oracle, if any."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T  # noqa: E402
from alphazero_gym_amd import _native  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 32
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, modes = [], {0: 0, 1: 0}
t0 = time.time()
for seed in range(first, first + count):
    try:
        T.test_hip_bit_exact_vs_oracle_random_configurations(_native, seed)
    except AssertionError as ex:
        bad.append(seed)
        print("seed", seed, "DIFFERS:", str(ex)[:200])
print(f"{count} random configurations from seed {first}: {len(bad)} differ {bad} ({time.time() - t0:.0f} s)")
