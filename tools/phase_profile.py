#!/usr/bin/env python3
"""Diagnostic: where a simulation step spends its cycles (in-kernel s_memtime stamps, -DAZG_STAMPS build).
Read the SHARES, not the absolute time (stamps serialise the schedule).  GPU box only:
    make -C alphazero_gym_amd/csrc libazgym_hip_stamp.so && python tools/phase_profile.py [pendulum|cartpole] [trees]"""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASE_A = "--phase-a" in sys.argv   # slots 4..6 = phase A's parts (libazgym_hip_stampa.so) instead of the network's
# --only SLOT[a]: libazgym_hip_ss<SLOT>[a].so (make -C alphazero_gym_amd/csrc ss SLOT=..): ONE stamp pair, a build that runs close to
# the product's time; prints that slot only ("a": slots 4..6 are phase A's parts).  --env-only = --only 15.
ONLY = None
if "--env-only" in sys.argv:
    sys.argv.remove("--env-only")
    ONLY = "15"
if "--only" in sys.argv:
    i = sys.argv.index("--only")
    ONLY = sys.argv[i + 1]
    del sys.argv[i:i + 2]
if "--phase-a" in sys.argv:
    sys.argv.remove("--phase-a")
if ONLY and ONLY.endswith("a"):
    PHASE_A = True
os.environ["AZG_HIP_LIB"] = os.path.join(ROOT, "alphazero_gym_amd", "csrc",
                                         f"libazgym_hip_ss{ONLY}.so" if ONLY else ("libazgym_hip_stampa.so" if PHASE_A else "libazgym_hip_stamp.so"))
SLOT_NAMES = {0: "barrier wait in front of the network phase", 1: "network phase (whole)", 2: "tree phase A: finish leaf + backup", 3: "tree phase B: select / step / expand",
              4: "mlp: layer 0 + ELU + publish", 5: "mlp: hidden MFMA loop (issue)", 6: "mlp: hidden activation", 7: "B per level: child records + division + U",
              8: "B per level: scores + arg-max", 9: "B per level: chosen record", 10: "B per level: path slot + cold prefetch", 11: "B per level: full level",
              12: "network phase, walking waves: the deferred half of phase B (tree_phase_b2)", 13: "B: whole descent", 14: "B: widening (noise, tanh, edge, child list)", 15: "B: env step + observation"}
SLOT_NAMES_A = {4: "A: finish leaf (head sums, exp, cold store)", 5: "A: backup (return chain, records)", 6: "A: re-scoring + resume"}
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from alphazero_gym_amd import synthetic as O  # noqa: E402  (make_weights)
from alphazero_gym_amd import _capi, _native  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "pendulum"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096   # > 4096 trees: the 8-wave / 32-tree workgroups
    if mode == "pendulum":
        n_sims, hidden = 200, [256, 256]
        e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=n_sims, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
        e.set_weights(_capi.make_desc(3, hidden, 2, "elu"), O.make_weights(34, 3, hidden, 2))
    else:
        n_sims, hidden = 100, [128, 128]
        e = _native.HipEngine(env_id=0, mode=0, n_trees=B, n_sims=n_sims, c_uct=1.5, gamma=1.0, num_actions=2, seed=34)
        e.set_weights(_capi.make_desc(4, hidden, 2, "relu"), O.make_weights(34, 4, hidden, 2))
    e.upload_roots(e.synthetic_roots())
    for _ in range(3):
        e.search_resident()
    e.sync()
    print("kernel ms", e.last_search_ms())
    lib = _native.lib()
    kname = e.search_info()["kernel_name"]
    print("kernel", kname)
    args = kname.split("<")[1].rstrip(">").split(",")
    waves, groups, nt = int(args[5]), int(args[6]), int(args[7])   # search_kernel<ENV, HP, NREG, TLDS, GMM, NW, NG, NT>
    rows = (B + nt * groups - 1) // (nt * groups) * waves   # one row per wave
    buf = np.zeros((max(rows, (B + 3) // 4 * 4), 16), np.uint64)
    lib.azg_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    n = lib.azg_debug_stamps(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), len(buf))
    assert n >= rows
    buf = buf[:rows]
    if waves == 8 and groups == 1:
        # eight waves share the network phase, the first four walk the trees: report the walkers (the helpers wait at the barrier meanwhile)
        walker = (np.arange(rows) % 8) < 4
        print(f"helper waves (4 of 8 per workgroup): barrier wait {buf[~walker, 0].astype(np.float64).mean() / (n_sims + 1):.0f}, network "
              f"{buf[~walker, 1].astype(np.float64).mean() / (n_sims + 1):.0f} cycles/step; the lines below are the WALKING waves")
        bufh = buf[~walker]
        buf = buf[walker]
    else:
        bufh = None
    if ONLY:
        k = int(re.match(r"\d+", ONLY).group(0))   # (suffixes: a = phase A's slots, c0 / c1 ... = A/B builds)
        nm = (SLOT_NAMES_A if PHASE_A and k in SLOT_NAMES_A else SLOT_NAMES)[k]
        v = buf[:, k].astype(np.float64) / (n_sims + 1)
        print(f"single pair, slot {ONLY:>3s}: {nm:48s} mean {v.mean():8.0f} cycles/step  (min {v.min():8.0f}, max {v.max():8.0f} over the walking waves)")
        if bufh is not None:
            vh = bufh[:, k].astype(np.float64) / (n_sims + 1)
            print(f"  same slot on the helper waves: mean {vh.mean():8.0f} (min {vh.min():8.0f}, max {vh.max():8.0f})")
        return
    names = ["barrier wait", "network (MLP)", "finish leaf + backup", "select/step/expand"]
    tot = buf[:, :4].sum(1).mean()
    for i, nm in enumerate(names):
        v = buf[:, i].astype(np.float64)
        print(f"{nm:24s} mean {v.mean() / (n_sims + 1):9.0f} cycles/step  ({100 * v.mean() / tot:5.1f} %)  min {v.min() / (n_sims + 1):8.0f} max {v.max() / (n_sims + 1):8.0f}")
    print(f"total {tot / (n_sims + 1):.0f} shader cycles/step")
    sub = ((4, "  A: finish leaf"), (5, "  A: backup (chain + records)"), (6, "  A: re-scoring + resume")) if PHASE_A else \
          ((4, "  mlp: layer0+ELU+publish"), (5, "  mlp: hidden MFMA loop"), (6, "  mlp: hidden act + store"))
    for i, nm in sub:
        v = buf[:, i].astype(np.float64)
        print(f"{nm:28s} mean {v.mean() / (n_sims + 1):9.0f} cycles/step")
    b = buf.astype(np.float64)
    lv = max(b[:, 12].sum(), 1)
    print(f"  B: whole descent           mean {b[:, 13].mean() / (n_sims + 1):9.0f} cycles/step, {b[:, 12].mean() / (n_sims + 1):.2f} completed levels per step "
          f"(wave-level: max over the wave's trees)")
    for i, nm in ((14, "widening (noise, tanh, edge, child list)"), (15, "env step + observation")):
        print(f"  B: after the descent: {nm:42s} mean {b[:, i].mean() / (n_sims + 1):9.0f} cycles/step")
    for i, nm in ((11, "full level"), (7, "child records + division + U"), (8, "scores + arg-max"), (9, "chosen record"), (10, "path slot + cold prefetch")):
        print(f"  B: per level: {nm:30s} {b[:, i].sum() / lv:8.0f} cycles")


if __name__ == "__main__":
    main()
