#!/usr/bin/env python3
"""Compiler's register / scratch / LDS report for the kernels of one translation unit (no GPU needed):
    python tools/resource_usage.py dispatch_cartpole [filter]      -> one line per kernel instantiation"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alphazero_gym_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt".split()


def main():
    tu = sys.argv[1] if len(sys.argv) > 1 else "dispatch_pendulum_large"
    flt = sys.argv[2] if len(sys.argv) > 2 else "kernel"
    extra = sys.argv[3:]
    p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", tu + ".hip"],
                       cwd=CSRC, capture_output=True, text=True)
    blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
    if not blocks:
        print(p.stderr[-2000:])
    for b in blocks:
        name = b.split("\n")[0].split()[0]
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"^void ", "", dn).split("(")[0]
        if flt not in dn:
            continue
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        scr, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{dn:60s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} spill {g('VGPRs Spill'):>2s} scratch {scr:>3s} occ {occ} LDS {lds}")


if __name__ == "__main__":
    main()
