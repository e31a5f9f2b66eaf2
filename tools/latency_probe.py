#!/usr/bin/env python3
"""Diagnostic: dependent-operation latencies of one wave on this GPU, in shader cycles (what the tree walk is made of).
GPU box only:  python tools/latency_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from alphazero_gym_amd import _native  # noqa: E402

NAMES = ["v_fma_f64", "v_fma_f32", "v_mul_f64 + v_add_f64", "float64 division, engine sequence (8 ops)", "float64 division, compiler sequence (11 ops)",
         "LDS round trip (ds_read_b32)", "LDS round trip (ds_read_b128 + cvt)", "DPP move + add", "ds_bpermute_b32",
         "global load round trip (L2 hit)", "azg_sincos (float64)", "v_mfma_f32_16x16x4_f32, dependent"]
out = _native.math_selftest(102, np.zeros(16))
for nm, v in zip(NAMES, out):
    print(f"{nm:48s} {v:8.1f} cycles")
