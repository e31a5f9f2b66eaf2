#!/usr/bin/env python3
"""Kernel time against the batch size for the two small BASELINE networks (GPU box):  python tools/size_sweep.py
C<n>: Pendulum-v1, n trees x 200 sims, 2x256 ELU;  B<n>: CartPole, n trees x 100 sims, 2x128 ReLU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import quick_times as Q  # noqa: E402

SIZES = (256, 1024, 2048, 4096, 6144, 8192, 16384, 65536)
for n in SIZES:
    Q.SHAPES["C%d" % n] = (Q.PEND, n, 200, 3, [256, 256], "elu")
    Q.SHAPES["B%d" % n] = (Q.CART, n, 100, 4, [128, 128], "relu")
sys.argv = ["x"] + ["C%d" % n for n in SIZES] + ["B%d" % n for n in SIZES]
Q.main()
