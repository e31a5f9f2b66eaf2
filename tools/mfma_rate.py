#!/usr/bin/env python3
"""Diagnostic: what the fp32 matrix pipe sustains on this GPU (register-only v_mfma_f32_16x16x4_f32 loop on every CU).
GPU box only:  python tools/mfma_rate.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from alphazero_gym_amd import _native  # noqa: E402

for wgs, iters, reps in [(256, 64, 50), (256, 64, 1), (256, 640, 20), (256, 6400, 5), (256, 64000, 2), (128, 6400, 5), (512, 6400, 5)]:
    x = np.zeros(8)
    x[:3] = [wgs, iters, reps]
    out = _native.math_selftest(101, x)
    cyc, ticks, ms, tf = out[0], out[1], out[3], out[4]
    mhz = cyc / (ticks / 100.0) if ticks else 0.0
    print(f"wgs {wgs:4d}  MFMAs/wave {iters * 16:8d}  launches {reps:3d}: {ms * 1e3:10.1f} us/launch  {tf:6.1f} TFLOP/s  "
          f"{cyc / (iters * 16):5.1f} cycles/MFMA  shader clock {mhz:6.0f} MHz")
