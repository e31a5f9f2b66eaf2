#!/usr/bin/env python3
"""Compiler's register / scratch / LDS report for the kernels of one translation unit, plus static instruction counts from the
gfx950 ISA it generates (no GPU needed):
    python tools/resource_usage.py dispatch_cartpole [filter] [extra hipcc flags]   -> one line per kernel instantiation
Columns: VGPR / AGPR / SGPR counts, spilled VGPRs and SGPRs (an SGPR spill is a v_writelane / v_readlane pair into a spare VGPR:
no scratch memory, but instructions in the stream), scratch bytes per lane, occupancy, static LDS; then the kernel's static
instruction mix: all | vector ALU (v_*, without MFMA and lane moves) | scalar ALU + branches (s_*, without waits / nops) | MFMA |
LDS (ds_*) | v_readlane + v_writelane | branches alone."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alphazero_gym_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt".split()


def demangle(name):
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    return re.sub(r"^void ", "", dn).split("(")[0]


def instruction_mix(asm_path):
    """{mangled kernel name: dict of static counts} from the device assembly (-S --cuda-device-only)."""
    out, cur = {}, None
    label = re.compile(r"^(_Z\w+):")
    with open(asm_path) as f:
        for line in f:
            m = label.match(line)
            if m:
                cur = out.setdefault(m.group(1), dict(all=0, valu=0, salu=0, mfma=0, lds=0, lane=0, branch=0))
                continue
            if cur is None:
                continue
            t = line.strip()
            if t.startswith(".end_amdhsa_kernel") or t.startswith("s_endpgm"):
                if t.startswith("s_endpgm"):
                    cur["all"] += 1
                continue
            if not t or t[0] in ".;/" or t.endswith(":"):
                continue
            op = t.split()[0]
            if not re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_|flat_)", op):
                continue
            cur["all"] += 1
            if op.startswith("v_mfma") or op.startswith("v_smfma"):
                cur["mfma"] += 1
            elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                cur["lane"] += 1
            elif op.startswith("v_"):
                cur["valu"] += 1
            elif op.startswith("ds_"):
                cur["lds"] += 1
            elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_endpgm")):
                cur["salu"] += 1
                if op.startswith(("s_cbranch", "s_branch")):
                    cur["branch"] += 1
    return out


def main():
    tu = sys.argv[1] if len(sys.argv) > 1 else "dispatch_pendulum_large"
    flt = sys.argv[2] if len(sys.argv) > 2 else "kernel"
    extra = sys.argv[3:]
    if tu in ("dispatch_pendulum_large", "dispatch_team_wide", "dispatch_mcc") and "-licm" not in " ".join(extra):   # (as alphazero_gym_amd/csrc/Makefile builds them)
        extra = extra + ["-mllvm", "-disable-machine-licm"]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, tu + ".s")
        p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-S", "--cuda-device-only", "-o", asm,
                                                                    tu + ".hip"], cwd=CSRC, capture_output=True, text=True)
        mix = instruction_mix(asm) if os.path.exists(asm) else {}
    blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
    if not blocks:
        print(p.stderr[-2000:])
    for b in blocks:
        name = b.split("\n")[0].split()[0]
        dn = demangle(name)
        if flt not in dn:
            continue

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        scr, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        m = mix.get(name)
        tail = (f" | insts {m['all']:5d} valu {m['valu']:5d} salu {m['salu']:5d} mfma {m['mfma']:4d} lds {m['lds']:4d} lane {m['lane']:3d} br {m['branch']:4d}"
                if m else "")
        print(f"{dn:60s} VGPR {g('[^A-Za-z]VGPRs'):>3s} AGPR {g('AGPRs'):>3s} SGPR {g('TotalSGPRs'):>3s} vspill {g('VGPRs Spill'):>2s} sspill {g('SGPRs Spill'):>2s} "
              f"scratch {scr:>3s} occ {occ} LDS {lds}{tail}")


if __name__ == "__main__":
    main()
