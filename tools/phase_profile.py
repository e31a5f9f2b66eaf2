#!/usr/bin/env python3
"""Diagnostic: where a simulation step spends its cycles (in-kernel s_memtime stamps, -DAZG_STAMPS build).
Read the SHARES, not the absolute time (stamps serialise the schedule).  GPU box only:
    make -C alphazero_gym_amd/csrc libazgym_hip_stamp.so && python tools/phase_profile.py [pendulum|cartpole] [trees]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASE_A = "--phase-a" in sys.argv   # slots 4..6 = phase A's parts (libazgym_hip_stampa.so) instead of the network's
ENV_ONLY = "--env-only" in sys.argv  # libazgym_hip_stampe.so: ONE stamp pair (env step + observation of phase B): runs close to the product's time
for flag in ("--phase-a", "--env-only"):
    if flag in sys.argv:
        sys.argv.remove(flag)
os.environ["AZG_HIP_LIB"] = os.path.join(ROOT, "alphazero_gym_amd", "csrc",
                                         "libazgym_hip_stampe.so" if ENV_ONLY else ("libazgym_hip_stampa.so" if PHASE_A else "libazgym_hip_stamp.so"))
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
    name = C.create_string_buffer(256)
    lib.azg_debug_kernel_name(C.c_void_p(e._h.value), name, C.c_size_t(256))
    print("kernel", name.value.decode())
    args = name.value.decode().split("<")[1].rstrip(">").split(",")
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
        buf = buf[walker]
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
