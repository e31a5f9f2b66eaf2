#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations / second, Pendulum-v1, 4096 trees per GPU, n_sims=200, 2x256 ELU MLP
(BASELINE.json `metric`, config C; config D = the same on 8 GPUs, weak scaling, one process per GPU).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one whole search (B trees x n_sims simulations) over synthetic fixed-seed root states that are already
resident in HBM; each step is ONE launch of the fused search kernel.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

N_TREES, N_SIMS, HIDDEN = 4096, 200, [256, 256]
FLOP_PER_SIM = 2 * (3 * 256 + 256 * 256 + 256 * 3)   # 134 144 FLOP per fused policy/value evaluation
PEAK_TFLOPS = 157.3                                    # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)


def profiled_traffic():
    """HBM bytes per launch of the search kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/*_pmc_summary.csv; FETCH_SIZE and WRITE_SIZE are collected in separate passes, in KiB; gfx950 reports half
    of the bytes of wide coalesced reads, so the read side is doubled as the microarch guide prescribes -- an upper bound
    for this kernel's narrow reads).  None when no profile is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")))
    if not files:
        return None
    vals = {}
    for r in csv.DictReader(open(files[-1])):
        vals[r["counter"]] = float(r["mean_per_launch"])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def cpu_baseline(seconds_target=15.0):
    """The C oracle (a scalar port of the reference's per-tree algorithm) on this box's host cores, OpenMP over trees,
    on a bounded sample of the same workload."""
    import oracle_lib as O
    from alphazero_gym_amd import _capi

    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    desc = _capi.make_desc(3, HIDDEN, 2, "elu")
    blob = O.make_weights(34, 3, HIDDEN, 2)
    n = max(cores, 8)
    kw = dict(env_id=2, mode=1, n_sims=N_SIMS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    e = O.OracleEngine(n_trees=n, **kw)
    e.set_weights(desc, blob)
    roots = e.synthetic_roots()
    t0 = time.perf_counter(); e.search(roots); dt = time.perf_counter() - t0
    rate = n * N_SIMS / dt
    e.close()
    n2 = int(min(N_TREES, max(n, rate * seconds_target / N_SIMS)))
    e = O.OracleEngine(n_trees=n2, **kw)
    e.set_weights(desc, blob)
    roots = e.synthetic_roots()
    t0 = time.perf_counter(); e.search(roots); dt = time.perf_counter() - t0
    e.close()
    return {"value": n2 * N_SIMS / dt, "unit": "sims/s", "cores": cores, "kind": "port",
            "sample": f"{n2} of {N_TREES} trees x {N_SIMS} sims, same seeds/weights, C oracle + OpenMP, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--trees", type=int, default=N_TREES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N")
    import torch

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import oracle_lib as O   # make_weights only (numpy); the oracle library itself is used by cpu_baseline() alone
    from alphazero_gym_amd import _capi, _native

    B = args.trees
    eng = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=N_SIMS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34,
                            tree_id_base=rank * B, device_id=local_rank)
    eng.set_weights(_capi.make_desc(3, HIDDEN, 2, "elu"), O.make_weights(34, 3, HIDDEN, 2))
    eng.upload_roots(eng.synthetic_roots())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.search_resident()
    eng.sync()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.search_resident()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-launch kernel duration from HIP events on the engine's stream (separate pass, so the event reads don't
    # serialise the timed loop)
    for _ in range(min(args.steps, 10)):
        eng.search_resident()
        kernel_ms.append(eng.last_search_ms())
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res = eng.results()
    assert (res["counts"].sum(1) == N_SIMS).all()
    eng.close()
    if rank == 0:
        sims = world * B * N_SIMS * args.steps
        kms = float(np.mean(kernel_ms))
        achieved = B * N_SIMS * FLOP_PER_SIM / (kms * 1e-3) / 1e12
        out = {
            "metric": "MCTS sims/sec (whole node), Pendulum-v1 4096 trees n_sims=200, 1/2/4/8 GPU", "value": sims / elapsed, "unit": "sims/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Pendulum-v1 A0C, {B} trees/GPU x {N_SIMS} sims, 2x256 ELU policy/value MLP, c_uct=0.05 c_pw=1 kappa=0.5",
                       "trees_per_gpu": B, "n_sims": N_SIMS, "parallelism": f"{world} independent shards (no data-path collective)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS,
                         "traffic": profiled_traffic(), "kernel": "search_kernel<2, 256, 1, 1, false, 4, 1> (ENV=Pendulum, HP=256, NREG=1, trees in LDS with 8-bit ids, no GMM, 4 waves, 1 tree group)", "kernel_ms": kms,
                         "note": "one launch = one whole search; achieved = trees x sims x 134144 FLOP / mean launch time (HIP events on the "
                                 "engine stream); policy/value MLP in fp32 MFMA, tree statistics in fp64; traffic = HBM bytes per launch from "
                                 "the committed rocprofv3 PMC passes (profiles/)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
