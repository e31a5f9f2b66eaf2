#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations / second, Pendulum-v1, 4096 trees per GPU, n_sims=200, 2x256 ELU MLP
(BASELINE.json `metric`).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N = 1  (config C): a "step" is one whole search (4096 trees x 200 simulations = ONE launch of the fused search kernel) over
       synthetic fixed-seed root states already resident in HBM, return_results of every tree included (written into device
       buffers by the search kernel's epilogue; azg_results_resident hands them out, no copy).  The JSON line also carries `roofline`
       (fp32 MFMA; `traffic` = HBM bytes per launch from two rocprofv3 --pmc child passes of this run), `roofline_hbm` (the tree walk's
       algorithmic bytes, counted from the searched trees, against the HBM peak), `cpu_baseline`, and `extra`: configs B and E and C at
       8192 trees timed the same way (B with its HBM-side block, E with its hardware MFMA-busy share), `mfma_busy` (the hardware's view
       of roofline.frac), `config_d_1rank` (the N > 1 workload on this one GPU, no collectives) and the PCIe-inclusive rate of config C.
       `--config-d` runs the N > 1 loop (below) with one rank: the collectives then go through RCCL with world size 1.
N > 1  (config D): one process per GPU (spawned here when the script was not started by torch.distributed.run), 4096 self-play
       games per GPU keyed by global game id.  A step is one device-resident self-play step (search + final action + env step +
       replay row) PLUS the all-gather of the step's replay rows over RCCL (HBM to HBM, overlapped with the next step's search)
       and a weight broadcast + engine re-sync (device to device: azg_set_weights_device) every --bcast-every steps: the loop
       shape of run_continuous.py:111-155 scaled out.
       `extra.search_only` is the same loop without the collectives; rank 0 adds `cpu_baseline` after the timed region.
       Rendezvous and collectives are bounded (--dist-timeout): a rank that never arrives or dies fails the run instead of hanging it.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

N_TREES, N_SIMS, HIDDEN = 4096, 200, [256, 256]
PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PENDULUM = dict(env_id=2, mode=1, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
CARTPOLE = dict(env_id=0, mode=0, c_uct=1.5, gamma=1.0, num_actions=2, seed=34)


def mlp_flops(in_dim, hidden, n_out):
    """2 * sum(in * out) of one fused policy/value evaluation."""
    dims = [in_dim] + list(hidden)
    return 2 * (sum(a * b for a, b in zip(dims[:-1], dims[1:])) + hidden[-1] * n_out)


FLOP_PER_SIM = mlp_flops(3, HIDDEN, 3)   # 134 144


def profiled_traffic(tag=""):
    """HBM bytes per launch of the search kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/*_pmc_summary.csv, or profiles/*_config<tag>_pmc_summary.csv; FETCH_SIZE and WRITE_SIZE are collected in separate
    passes, in KiB; gfx950 reports half of the bytes of wide coalesced reads, so the read side is doubled as the microarch guide
    prescribes -- an upper bound for this kernel's narrow reads).  (bytes, file name) or (None, None) when no profile is committed."""
    import csv
    import glob
    if tag:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_config{tag}_pmc_summary.csv")))
    else:
        files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")) if "config" not in os.path.basename(f))
    for f in reversed(files):
        vals = {}
        for r in csv.DictReader(open(f)):
            name = r.get("kernel") or r.get("Kernel_Name") or ""
            if name and not ("search_kernel" in name or "ls_team_kernel" in name):
                continue                                   # (the summaries of the wide configs also list the small kernels)
            vals[r["counter"]] = float(r["mean_per_launch"])
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, os.path.relpath(f, ROOT)
    return None, None


PROBE_SHAPES = {   # --probe-config: (engine kwargs, n_sims, in_dim, hidden, activation)
    "C": (PENDULUM, N_SIMS, 3, HIDDEN, "elu"),
    "B": (CARTPOLE, 100, 4, [128, 128], "relu"),
    "E": (PENDULUM, N_SIMS, 3, [1024] * 4, "elu"),
}


def traffic_probe(args):
    """`--traffic-probe` (child of live_traffic below, run under rocprofv3 --pmc): a few searches of one shape, nothing else."""
    from alphazero_gym_amd import _capi, _native
    from alphazero_gym_amd.synthetic import make_weights
    kw, n_sims, in_dim, hidden, act = PROBE_SHAPES[args.probe_config]
    eng = _native.HipEngine(n_trees=args.trees, n_sims=n_sims, **kw)
    eng.set_weights(_capi.make_desc(in_dim, hidden, 2, act), make_weights(34, in_dim, hidden, 2))
    eng.upload_roots(eng.synthetic_roots())
    for _ in range(6):
        eng.search_resident()
    eng.sync()
    eng.close()


def under_profiler():
    """True when this process already runs under rocprofv3 / rocprofiler-sdk (its tool library is preloaded into every child):
    a nested rocprofv3 would then re-exec a process whose GPU runtime is already initialised."""
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        return True
    return any(k.startswith("ROCPROF") for k in os.environ)


def clean_child_env():
    """The parent's environment without anything a profiler put there (LD_PRELOAD, ROCP_*, ROCPROF*): no hop in front of the
    final interpreter of a child run carries a tool library."""
    env = {k: v for k, v in os.environ.items() if not (k == "LD_PRELOAD" or k.startswith("ROCP_") or k.startswith("ROCPROF"))}
    env["TMPDIR"] = "/tmp"
    return env


def live_counters(trees, config, kernel, passes):
    """Hardware counters of `kernel` measured NOW: one child run of this script's --traffic-probe per entry of `passes` (a list of
    counter-name lists: counters that do not fit one pass go into separate ones) under `rocprofv3 --pmc`, averaged over the
    kernel's launches but the first (which also pulls the weights and tables in).  The children are started as ordinary child
    processes from /tmp with TMPDIR=/tmp and an environment without anything a profiler left behind; the program after `--` is the
    interpreter binary itself.  ({counter: mean per launch}, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    if under_profiler():
        return None, "already under a profiler"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    # the program after `--` must be the interpreter binary itself (no wrapper script, no launcher that re-execs under the profiler)
    py = os.path.realpath(sys.executable)
    try:
        if open(py, "rb").read(4) != b"\x7fELF":
            return None, "the interpreter is not an ELF binary"
    except OSError:
        return None, "the interpreter binary cannot be read"
    out = {}
    work = tempfile.mkdtemp(prefix="azg_traffic_", dir="/tmp")
    try:
        for n, counters in enumerate(passes):
            d = os.path.join(work, f"pass{n}")
            cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "--", py, os.path.abspath(__file__), "--traffic-probe",
                                                     "--trees", str(trees), "--probe-config", config]
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=clean_child_env(), capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {' '.join(counters)} timed out"
            if p.returncode != 0:
                return None, f"rocprofv3 --pmc {' '.join(counters)} exited with {p.returncode}"
            vals = {c: [] for c in counters}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") in vals:
                        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c in counters:
                if len(vals[c]) < 2:
                    return None, f"no {c} rows for {kernel}"
                out[c] = float(np.mean(vals[c][1:]))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out, None


def live_traffic(trees, config="C", kernel="search_kernel"):
    """HBM bytes per launch of the search kernel measured NOW: FETCH_SIZE and WRITE_SIZE in separate passes (the two do not fit
    one), values in KiB per dispatch, the read side doubled (gfx950 tallies 128-byte read requests at 64 bytes:
    MI355X_MICROARCH.md, HBM section -- an upper bound for this kernel's narrow reads).  (bytes, None) or (None, reason)."""
    out, why = live_counters(trees, config, kernel, [["FETCH_SIZE"], ["WRITE_SIZE"]])
    if out is None:
        return None, why
    return (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0, None


def live_mfma_busy(trees, config="C", kernel="search_kernel"):
    """Share of the launch during which the matrix pipes were busy, from the hardware: SQ_VALU_MFMA_BUSY_CYCLES (summed over the
    SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024 SIMDs).  (fraction, None) or (None, reason)."""
    out, why = live_counters(trees, config, kernel, [["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]])
    if out is None:
        return None, why
    return out["SQ_VALU_MFMA_BUSY_CYCLES"] / (out["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), None


def physical_cores():
    """Distinct (package, core) pairs; falls back to the logical count."""
    seen = set()
    try:
        for d in os.listdir("/sys/devices/system/cpu"):
            if d.startswith("cpu") and d[3:].isdigit():
                t = f"/sys/devices/system/cpu/{d}/topology/"
                seen.add((open(t + "physical_package_id").read().strip(), open(t + "core_id").read().strip()))
    except OSError:
        pass
    return len(seen) or (os.cpu_count() or 1)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by a cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(short=False):
    """CPU lines next to the GPU number, each on a bounded sample of the same workload:
      * the C oracle (scalar port of the per-tree algorithm, OpenMP over trees, trees allocated by their worker thread) on one
        thread and on one thread per physical core;
      * the Python object-tree restatement with the reference's cost structure (oracle/pytree.py: batch-1 torch-CPU forwards,
        env replay from the root, numpy UCT), P worker processes x 1 torch thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    import pytree
    from alphazero_gym_amd import _capi
    from alphazero_gym_amd.synthetic import make_weights

    cores = min(physical_cores(), usable_cpus())   # one worker per physical core the process is allowed to run on
    desc = _capi.make_desc(3, HIDDEN, 2, "elu")
    blob = make_weights(34, 3, HIDDEN, 2)

    def oracle_rate(threads, n_trees, seconds=0.0):
        """searches of n_trees trees until `seconds` have passed (at least one, after an untimed one that allocates the trees)"""
        O.set_threads(threads)
        e = O.OracleEngine(n_trees=n_trees, n_sims=N_SIMS, **PENDULUM)
        e.set_weights(desc, blob)
        roots = e.synthetic_roots()
        if seconds > 0:
            e.search(roots)
        reps, t0 = 0, time.perf_counter()
        while True:
            e.search(roots)
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= seconds:
                break
        e.close()
        return reps * n_trees * N_SIMS / dt, dt, reps

    r1, _, _ = oracle_rate(1, 16)                                  # ~0.3 s: sizes the samples
    n1 = int(max(16, min(N_TREES, r1 * 2.0 / N_SIMS)))
    r1, dt1, reps1 = oracle_rate(1, n1, 2.0 if short else 4.0)
    nall = int(max(cores, min(N_TREES, r1 * cores * 2.0 / N_SIMS)))
    rall, dtall, repsall = oracle_rate(cores, nall, 4.0 if short else 8.0)
    out = {"value": rall, "unit": "sims/s", "cores": cores, "kind": "port",
           "sample": f"{repsall} searches of {nall} of {N_TREES} trees x {N_SIMS} sims, same seeds/weights, C oracle, OpenMP over trees on "
                     f"{cores} threads (physical cores {physical_cores()}, usable CPUs {usable_cpus()}, logical CPUs {os.cpu_count()}), {dtall:.1f} s",
           "single_thread": {"value": r1, "unit": "sims/s", "cores": 1, "sample": f"{reps1} searches of {n1} trees x {N_SIMS} sims, {dt1:.1f} s"}}
    if short:   # (N > 1 runs: the other ranks are waiting)
        return out
    procs = min(cores, 64)
    rpy, simspy, dtpy = pytree.throughput("pendulum", n_rollouts=N_SIMS, hidden=HIDDEN, processes=procs, trees_per_process=8, seconds=6.0, detail=True)
    out["python_object_tree"] = {"value": rpy, "unit": "sims/s", "cores": procs, "kind": "port",
                                 "sample": f"{procs} processes x 1 torch thread, whole searches of {N_SIMS} sims for {dtpy:.1f} s each ({simspy} simulations in all, "
                                           "start-up outside the clock), oracle/pytree.py (the reference's cost structure: batch-1 torch forwards, env replay "
                                           "from the root, numpy UCT)"}
    return out


def time_search(eng, steps, warmup):
    """(median, mean) kernel milliseconds of `steps` resident searches by HIP events on the engine's stream."""
    for _ in range(warmup):
        eng.search_resident()
    eng.sync()
    ms = []
    for _ in range(steps):
        eng.search_resident()
        ms.append(eng.last_search_ms())
    return float(np.median(ms)), float(np.mean(ms))


def kernel_name(eng):
    """The kernel(s) the engine's last search ran as, in rocprofv3's spelling (the engine's own account, not a constant here)."""
    return eng.search_info()["kernel_name"]


PEAK_HBM_GBS = 8000.0   # MI355X HBM3E (MI355X_MICROARCH.md)
B_KERNEL = "search_kernel<0, 128, 1, 1, false, 4, 1, 16, 1>"   # what config B is expected to run as (the engine reports what it did run)


def tree_walk_bytes(dump, n_actions, n_sims):
    """SURVEY 8d's declared minimal SoA traffic of the tree walk, from the searched trees themselves:
    bytes = 16 L + 12 C (+ 4 C priors, discrete) + 56 L + 92 E + 40 per simulation, with L = levels descended (every traversal of
    an edge increments its count: sum of edge counts), C = children scanned (discrete: all n_actions per level), E = expansions
    (new nodes: records / n_actions).  Returns (bytes per simulation, L, C, E) averaged over all trees (discrete mode)."""
    B = dump["n_records"].shape[0]
    sims = float(B * n_sims)
    L = float(dump["edge_n"].sum()) / sims
    C = n_actions * L
    E = float((dump["n_records"] - 1).sum()) / n_actions / sims
    return 16 * L + 16 * C + 56 * L + 92 * E + 40, L, C, E


def tree_walk_bytes_continuous(dump, n_sims, c_pw, kappa):
    """The same declared traffic for a progressive-widening search (no priors: 12 B per scanned child): L from the edge counts, E
    from the expanded records, C = children scored, reconstructed exactly from the nodes' visit counts: a node's v-th visit either
    widens it (states.py:271-275: ceil(c_pw (v+1)^kappa) > K) or scores its K children (mcts.py:728-741), so the number of
    children a node has scored so far is a function of its visit count alone (the root starts with the child of mcts.py:673)."""
    import math
    B = dump["n_records"].shape[0]
    sims = float(B * n_sims)
    scored = np.zeros((2, n_sims + 2), np.float64)     # [is_root][visits] -> children scored over those visits
    kids = np.zeros((2, n_sims + 2), np.int64)
    for root in (0, 1):
        K, Cacc = root, 0
        for v in range(n_sims + 1):
            scored[root, v], kids[root, v] = Cacc, K
            if math.ceil(c_pw * (min(v, n_sims + 1) + 1) ** kappa) > K:
                K += 1
            else:
                Cacc += K
        scored[root, n_sims + 1], kids[root, n_sims + 1] = Cacc, K
    nn = np.minimum(dump["node_n"], n_sims + 1)
    is_root = np.zeros_like(nn); is_root[:, 0] = 1
    valid = np.arange(nn.shape[1])[None, :] < dump["n_records"][:, None]
    Cn = float((scored[is_root, nn] * valid).sum()) / sims
    L = float((dump["edge_n"] * valid).sum()) / sims
    E = float((((dump["node_flags"] & 1) != 0) & valid).sum() - B) / sims
    return 16 * L + 12 * Cn + 56 * L + 92 * E + 40, L, Cn, E


def hbm_block(per_sim, L, Cn, E, trees, n_sims, ms, traffic, traffic_note):
    gbs = trees * n_sims * per_sim / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": traffic,
            "traffic_frac_of_peak": (traffic / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if traffic else None,
            "bytes_per_sim": per_sim, "levels_per_sim": L, "children_per_sim": Cn, "expansions_per_sim": E,
            "note": "the tree walk against the HBM roofline (SURVEY 8d): achieved = algorithmic bytes (16L + 12C [+ 4C priors] + 56L + 92E + 40 "
                    "per simulation, L / C / E counted from the searched trees) x simulations / launch time; traffic = HBM bytes per "
                    "launch, " + traffic_note + ".  The hot records live in LDS for the whole search, so the counter traffic is below "
                    "the algorithmic bytes and both are far below the HBM peak: the walk is bound by dependent-access latency"}


def extra_config(name, kw, trees, n_sims, in_dim, hidden, n_dist, act, expect, flops_per_sim, note, device_id, hbm=False, live=None):
    """One more configuration timed like the headline.  flops_per_sim None: evaluations per simulation counted from the trees x
    the network's FLOP.  hbm: add SURVEY 8d's HBM-side block.  live = (--probe-config tag, kernel name): measure HBM traffic with two
    rocprofv3 --pmc child passes (falls back to the committed profile)."""
    import ctypes as C
    from alphazero_gym_amd import _capi, _native
    from alphazero_gym_amd.synthetic import make_weights
    eng = _native.HipEngine(n_trees=trees, n_sims=n_sims, device_id=device_id, **kw)
    eng.set_weights(_capi.make_desc(in_dim, hidden, n_dist, act), make_weights(34, in_dim, hidden, n_dist))
    eng.upload_roots(eng.synthetic_roots())
    med, mean = time_search(eng, 7, 2)
    res = eng.results()
    assert (res["counts"].sum(1) == n_sims).all()
    ran = kernel_name(eng)
    fallbacks = eng.search_info()["team_fallbacks"]
    walk = None
    if hbm or flops_per_sim is None:
        walk = tree_walk_bytes(eng.dump_tree(), kw.get("num_actions", 2), n_sims)
    if flops_per_sim is None:
        flops_per_sim = walk[3] * mlp_flops(in_dim, hidden, 1 + n_dist)   # (E evaluations per simulation: traces that end in a terminal node need none)
    eng.close()
    ach = trees * n_sims * flops_per_sim / (med * 1e-3) / 1e12
    out = {"config": name, "ms_per_search": med, "ms_mean": mean, "sims_per_s": trees * n_sims / (med * 1e-3), "trees": trees, "n_sims": n_sims,
           "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS},
           "kernels": [ran], "note": note}
    if ran != expect:
        out["fallback"] = f"expected {expect}; the engine ran {ran} (team-kernel fallbacks: {fallbacks})"
    traffic, tnote = None, "not measured"
    if live and ran == expect:
        traffic, why = live_traffic(trees, live[0], live[1])
        tnote = "measured in this run (two rocprofv3 --pmc child passes, 2 x FETCH_SIZE + WRITE_SIZE)"
        if traffic is None:
            traffic, src = profiled_traffic(live[0])
            tnote = f"live PMC passes unavailable ({why}); from the committed passes ({src})"
        out["roofline"]["traffic"] = traffic
        out["roofline"]["traffic_note"] = tnote
        if not hbm:   # (the MFMA-bound configuration: the hardware's own account of how busy the matrix pipes were)
            busy, why = live_mfma_busy(trees, live[0], live[1])
            out["roofline"]["mfma_busy"] = busy
            out["roofline"]["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), one more rocprofv3 --pmc child pass"
                                                 if busy is not None else f"not measured ({why})")
    if hbm:
        out["roofline_hbm"] = hbm_block(*walk, trees, n_sims, med, traffic, tnote)
    return out


def config_e_leg(args, rank, world, dev, dist, barrier):
    """BASELINE.json configs[4] under N > 1: every rank searches its own shard of `--e-trees` trees (global tree ids rank * e_trees ..) with the
    4x1024 network -- the reference's MCTSContinuous.search (alphazero/search/mcts.py:656-702) at E's sizes, no collective inside: K whole
    searches bracketed by barriers, the max over ranks of the wall time, every rank's kernel form / fall-backs / event times gathered.
    Returns the block rank 0 reports (None on the other ranks)."""
    import ctypes as C
    import torch
    from alphazero_gym_amd import _capi, _native
    from alphazero_gym_amd.synthetic import make_weights
    T, K, hidden = args.e_trees, args.e_searches, [1024] * 4
    e = _native.HipEngine(n_trees=T, n_sims=N_SIMS, tree_id_base=rank * T, device_id=dev, **PENDULUM)
    e.set_weights(_capi.make_desc(3, hidden, 2, "elu"), make_weights(34, 3, hidden, 2))
    e.upload_roots(e.synthetic_roots())
    e.search_resident(); e.results_resident(); e.sync()     # warm-up (first launch of the kernel form, L2-cold weights)
    barrier()
    t0 = time.perf_counter()
    ev = []
    for _ in range(K):
        e.search_resident()
        e.results_resident()
        e.sync()                                              # (a search is 12.7 ms: the event read between searches is noise)
        ev.append(e.last_search_ms())
    barrier()
    wall = time.perf_counter() - t0
    res = e.results()
    assert (res["counts"].sum(1) == N_SIMS).all(), "config E leg: a tree does not hold n_sims visits"
    name = kernel_name(e)
    fallbacks = int(e.search_info()["team_fallbacks"])
    e.close()
    cpu = args.backend != "nccl"
    mine = torch.tensor([wall, float(np.median(ev)), float(min(ev)), float(max(ev)), float(fallbacks)], dtype=torch.float64, device="cpu" if cpu else "cuda")
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    names = [None] * world
    dist.all_gather_object(names, name)
    if rank != 0:
        return None
    a = torch.stack(allr).cpu().numpy()
    wall_max = float(a[:, 0].max())
    ms = a[:, 1]
    flop = mlp_flops(3, hidden, 3)
    ach = T * N_SIMS * flop / (float(np.median(ms)) * 1e-3) / 1e12
    return {"config": f"E: Pendulum-v1, 4x1024 ELU policy/value MLP, {world * T} trees = {T} per GPU x {N_SIMS} sims, {world} ranks (trees sharded by global id, "
                      "no collective inside a search)",
            "sims_per_s": world * T * N_SIMS * K / wall_max, "searches": K, "wall_ms_per_search_max_over_ranks": wall_max / K * 1e3,
            "kernel_ms_per_search": {"min": float(ms.min()), "median": float(np.median(ms)), "max": float(ms.max()),
                                     "note": "every rank's median over its searches (HIP events on the engine stream), then min / median / max over the ranks"},
            "kernels": sorted(set(names)), "team_kernel_fallbacks_per_rank": [int(x) for x in a[:, 4]],
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS,
                         "note": "per GPU, from the median rank's kernel time; 6 303 744 FLOP per simulation"},
            "note": "sims of all ranks / max-over-ranks wall time of K searches + return_results; ranks that share one GPU (--same-device) "
                    "time-slice it and their team kernels may fall back to the per-layer launches: reported, not an error"}


def spawn(args):
    """`--gpus N` without torch.distributed.run around us: start it as a child process (this process has not touched the GPU
    or torch) and exit with its code; the child's rank 0 prints the JSON line."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--trees", str(args.trees), "--bcast-every", str(args.bcast_every), "--gather-every", str(args.gather_every),
           "--backend", args.backend, "--dist-timeout", str(args.dist_timeout), "--e-trees", str(args.e_trees), "--e-searches", str(args.e_searches)]
    for flag, on in (("--same-device", args.same_device), ("--config-d", args.config_d), ("--verify-gather", args.verify_gather),
                     ("--no-cpu-baseline", args.no_cpu_baseline)):
        if on:
            cmd.append(flag)
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--trees", type=int, default=N_TREES, help="trees (games) per GPU")
    ap.add_argument("--bcast-every", type=int, default=10, help="N > 1: weight broadcast + engine re-sync every this many steps")
    ap.add_argument("--gather-every", type=int, default=4, help="N > 1: replay rows are all-gathered in blocks of this many steps")
    ap.add_argument("--backend", default="nccl", help="N > 1: nccl (= RCCL) or gloo (functional test of the loop on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="N > 1: every rank on GPU 0 (functional test on a 1-GPU box, with gloo)")
    ap.add_argument("--config-d", action="store_true", help="run the N > 1 loop (self-play step + all-gather + weight broadcast) also with one rank: "
                                                            "RCCL with world size 1 on the engine-owned ring")
    ap.add_argument("--verify-gather", action="store_true", help="N > 1 loop: check every gathered block against the rows the previous step wrote (slow)")
    ap.add_argument("--traffic-probe", action="store_true", help="(internal) a few searches for the PMC passes of live_traffic()")
    ap.add_argument("--probe-config", default="C", choices=sorted(PROBE_SHAPES), help="(internal) the shape --traffic-probe runs")
    ap.add_argument("--no-live-traffic", action="store_true", help="N = 1: take roofline.traffic from the committed profile instead of measuring it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-timeout", type=float, default=300.0, help="N > 1: seconds before a rendezvous / collective that does not complete "
                                                                      "(a dead rank) fails the run")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: skip the config B / E lines")
    ap.add_argument("--e-trees", type=int, default=1024, help="N > 1: trees per GPU of the config E leg (4x1024 MLP, BASELINE.json configs[4]: 8192 trees over 8 GPUs)")
    ap.add_argument("--e-searches", type=int, default=5, help="N > 1: timed searches of the config E leg (0: skip the leg)")
    args = ap.parse_args()

    if args.traffic_probe:
        return traffic_probe(args)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.config_d):
        spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    dev = 0 if args.same_device else local_rank
    import torch

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        # bounded rendezvous and collectives: a rank that never shows up, or dies, makes the others fail (RCCL's watchdog aborts
        # the process when a collective exceeds the timeout; torch.distributed.run then tears the job down) -- the bench exits
        # non-zero instead of hanging
        import datetime
        limit = datetime.timedelta(seconds=args.dist_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=limit)
        else:
            dist.init_process_group(args.backend, timeout=limit)

    from alphazero_gym_amd import _capi, _native
    from alphazero_gym_amd.synthetic import make_weights

    if dist is not None and os.environ.get("AZG_BENCH_DIE_RANK") == str(rank):
        os._exit(3)   # (test hook, tests/test_bench.py: a rank that dies after the rendezvous must fail the run, not hang it)

    B = args.trees
    desc, blob = _capi.make_desc(3, HIDDEN, 2, "elu"), make_weights(34, 3, HIDDEN, 2)
    eng = _native.HipEngine(n_trees=B, n_sims=N_SIMS, tree_id_base=rank * B, device_id=dev, **PENDULUM)
    eng.set_weights(desc, blob)
    eng.upload_roots(eng.synthetic_roots())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    config_d = world > 1 or args.config_d
    if not config_d:
        # ---------------- config C: K whole searches, roots resident in HBM
        for _ in range(args.warmup):
            eng.search_resident()
            eng.results_resident()
        eng.sync()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.search_resident()      # the whole search: one launch
            eng.results_resident()     # return_results of every tree: device buffers written by the search kernel's epilogue, no copy
        eng.sync()
        barrier()
        elapsed = time.perf_counter() - t0
        kmed, kmean = time_search(eng, min(args.steps, 10), 0)   # separate pass: the event reads would serialise the timed loop
        res = eng.results()
        assert (res["counts"].sum(1) == N_SIMS).all()
        walk_c = tree_walk_bytes_continuous(eng.dump_tree(), N_SIMS, PENDULUM["c_pw"], PENDULUM["kappa"])
        # the boundary with host buffers (azg_search uploads the roots, azg_results downloads the root statistics over PCIe)
        roots = eng.synthetic_roots()
        pc = []
        for _ in range(5):
            t1 = time.perf_counter(); eng.search(roots); eng.results(); pc.append(time.perf_counter() - t1)
        extra["pcie_inclusive"] = {"sims_per_s": B * N_SIMS / float(np.median(pc)), "ms_per_search": float(np.median(pc)) * 1e3,
                                   "note": "azg_search (host roots in) + azg_results (host root statistics out), median of 5; never `value`"}
        # the N > 1 workload (config D's self-play step) on this one GPU, no collectives: what the driver's 1 -> N ratio should be
        # read against (N = 1 times config C = search + return_results; N > 1 times config D = that + final action, env step, replay row)
        eng.selfplay_begin(200, capacity_steps=8, fifo=True)
        for _ in range(args.warmup):
            eng.selfplay_step()
        eng.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng.selfplay_step()
        eng.sync()
        dsp = time.perf_counter() - t1
        extra["config_d_1rank"] = {"sims_per_s": B * N_SIMS * args.steps / dsp, "ms_per_step": dsp / args.steps * 1e3, "steps": args.steps,
                                   "note": "config D's step on one GPU (search + final action + env step + replay row on the device, no collectives): "
                                           "the like-for-like N = 1 leg of the N > 1 lines, whose `value` times this step plus the replay all-gather "
                                           "and the weight broadcast"}
        workload = f"config C: Pendulum-v1 A0C, {B} trees/GPU x {N_SIMS} sims, 2x256 ELU policy/value MLP, c_uct=0.05 c_pw=1 kappa=0.5"
        parallelism = "1 GPU"
    else:
        # ---------------- config D: self-play steps + replay all-gather + weight broadcast
        from alphazero_gym_amd.agent.buffers import DeviceReplay
        K = max(1, args.gather_every)
        eng.selfplay_begin(200, capacity_steps=2 * K, fifo=True)
        ring = DeviceReplay(eng, batch_size=32).ring.reshape(2, K * B, -1)   # zero-copy torch view of the device ring: two blocks of K steps
        RL = ring.shape[-1]
        on_host = args.backend != "nccl"
        gathered = torch.empty((world * K * B, RL), dtype=torch.float32, device="cpu" if on_host else ring.device)
        flat = torch.from_numpy(blob.copy())
        d_flat = None if on_host else torch.from_numpy(blob.copy()).to(ring.device)
        verify = {"checked": 0, "snap": None, "gathers": 0}
        # what each rank's host saw (ms): the all-gather (call to data usable), the weight broadcast + engine re-sync, a whole loop
        # iteration (wait for the previous step + collectives + launching the next one); filled while `timing` is on
        tlog = {"on": False, "gather_ms": [], "bcast_ms": [], "step_ms": []}

        def gather(block):
            """all-gather one block of K finished steps (K * B rows per rank), HBM to HBM"""
            rows = ring[block]
            g0 = time.perf_counter()
            dist.all_gather_into_tensor(gathered, rows.cpu() if on_host else rows)
            torch.cuda.current_stream().synchronize()         # the block is free again before the steps that overwrite it are launched
            if tlog["on"]:
                tlog["gather_ms"].append((time.perf_counter() - g0) * 1e3)
            verify["gathers"] += 1
            if args.verify_gather:
                mine = gathered[rank * K * B:(rank + 1) * K * B]
                assert torch.equal(mine.cpu(), verify["snap"]), "the gathered block is not the block the last K steps wrote"
                verify["checked"] += 1

        def run(steps, collectives):
            """`steps` self-play steps; with collectives, every K steps the block of the K steps just finished is all-gathered
            while the next step searches (fewer, larger collectives: K * B rows per rank and gather).  Which ring slots a step
            writes comes from the engine's own account (azg_selfplay_ring: steps played since begin), not from the loop index:
            the FIFO ring of 2 K steps keeps turning across calls of this function."""
            for s in range(steps):
                it0 = time.perf_counter()
                eng.sync()                                        # the previous step finished (it ran while the host did the last gather)
                total = eng.selfplay_ring()[2]                    # steps played so far = index of the step about to be launched
                if collectives and s > 0 and s % args.bcast_every == 0:
                    b0 = time.perf_counter()
                    if on_host:                                   # functional run over gloo: staged through the host
                        dist.broadcast(flat, src=0)
                        eng.set_weights(desc, flat.numpy())
                    else:                                         # RCCL broadcast into HBM, re-layout by the engine's gather kernel
                        dist.broadcast(d_flat, src=0)
                        torch.cuda.current_stream().synchronize()
                        eng.set_weights_device(desc, d_flat.data_ptr(), d_flat.numel())
                    if tlog["on"]:
                        tlog["bcast_ms"].append((time.perf_counter() - b0) * 1e3)
                full = collectives and total > 0 and total % K == 0   # steps total - K .. total - 1 fill one block, all finished
                if full and args.verify_gather:
                    # the rows those K steps wrote, found without the slot formula: the block that differs from the snapshot taken
                    # K steps ago; it must be the one that is about to be gathered
                    now = ring.clone().cpu()
                    if verify.get("before") is not None:
                        changed = [i for i in range(2) if not torch.equal(now[i], verify["before"][i])]
                        assert changed == [(total // K - 1) % 2], (changed, total, K)
                    verify["snap"], verify["before"] = now[(total // K - 1) % 2], now
                eng.selfplay_step()                               # launches only: this step now runs on the engine's stream
                if full:
                    gather((total // K - 1) % 2)                  # FIFO, capacity 2 K: step k lives in slot k % 2K, block (k / K) % 2
                if tlog["on"]:
                    tlog["step_ms"].append((time.perf_counter() - it0) * 1e3)

        run(args.warmup, True)
        barrier()
        t0 = time.perf_counter()
        tlog["on"] = True
        run(args.steps, True)
        tlog["on"] = False
        barrier()
        elapsed = time.perf_counter() - t0
        barrier()
        t1 = time.perf_counter()
        run(args.steps, False)
        barrier()
        plain = time.perf_counter() - t1
        if verify["gathers"]:
            counts = gathered[:, 3 + eng.kmax:3 + 2 * eng.kmax].sum(1)
            assert bool((counts == N_SIMS).all()), "a gathered replay row does not hold n_sims visits"
        kmed, kmean = time_search(eng, 5, 0)
        if dist is not None and args.e_searches > 0:
            extra["config_e_nrank"] = config_e_leg(args, rank, world, dev, dist, barrier)
        workload = (f"config D: Pendulum-v1 A0C self-play, {world * B} games = {B} per GPU x {N_SIMS} sims per move, 2x256 ELU MLP; per step: "
                    f"search + final action + env step + replay row on the device; every {K} steps an all-gather of those steps' "
                    f"{world * K * B} replay rows ({world * K * B * RL * 4 / 1e6:.1f} MB, HBM to HBM, beside the next step's search); a weight "
                    f"broadcast + engine re-sync (device to device) every {args.bcast_every} steps")
        parallelism = f"{world} ranks x {B} games (games sharded by global id; RCCL all-gather + broadcast outside the search)"
        if args.backend != "nccl":
            parallelism += f" [functional run: backend {args.backend}, rows staged through the host]"
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if args.backend != "nccl" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if config_d:
            # every rank's own mean of each host-side interval -> min / median / max over the ranks (a straggler, or a rank whose
            # collectives take longer than the others', shows here and nowhere else in the line)
            mine = torch.tensor([float(np.mean(tlog[k])) if tlog[k] else -1.0 for k in ("gather_ms", "bcast_ms", "step_ms")] +
                                [float(len(tlog["gather_ms"])), float(len(tlog["bcast_ms"])), float(dist.get_world_size())],
                                dtype=torch.float64, device="cpu" if args.backend != "nccl" else "cuda")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = torch.stack(allr).cpu().numpy()
        if config_d:
            t = torch.tensor([plain], device="cpu" if args.backend != "nccl" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            plain = float(t.item())
    kname = kernel_name(eng)
    eng.close()
    if rank == 0:
        sims = world * B * N_SIMS * args.steps
        achieved = B * N_SIMS * FLOP_PER_SIM / (kmean * 1e-3) / 1e12
        if config_d:
            extra["search_only"] = {"sims_per_s": sims / plain, "ms_per_step": plain / args.steps * 1e3,
                                    "note": "the same self-play loop without the all-gather and the weight broadcast"}
            def spread(col):
                v = per_rank[:, col]
                v = v[v >= 0]
                return None if v.size == 0 else {"min": float(v.min()), "median": float(np.median(v)), "max": float(v.max())}
            extra["collectives"] = {"backend": args.backend, "world_size": world, "gather_every": K, "gathers": verify["gathers"],
                                    "gathers_verified": verify["checked"],
                                    "world_size_seen": sorted(set(int(x) for x in per_rank[:, 5])),   # dist.get_world_size() of every rank
                                    "per_rank_ms": {"gather_ms": spread(0), "bcast_ms": spread(1), "step_ms": spread(2),
                                                    "gathers_timed": int(per_rank[:, 3].min()), "bcasts_timed": int(per_rank[:, 4].min()),
                                                    "note": "host-side means of every rank over the timed run, then min / median / max over the ranks: "
                                                            "all-gather call until its rows are usable, weight broadcast + engine re-sync, one whole "
                                                            "loop iteration (wait for the previous step, collectives, launch of the next step)"},
                                    "weight_sync": "host (gloo functional run)" if on_host else "device to device: RCCL broadcast into HBM + azg_set_weights_device"}
        elif not args.no_extra:
            extra["configs"] = [
                extra_config("C at 8192 trees per GPU (the batch shape of real self-play runs: two 16-tree groups per CU)", PENDULUM, 8192, 200, 3, HIDDEN, 2, "elu",
                             "search_kernel<2, 256, 1, 1, false, 8, 2, 16, 1>", FLOP_PER_SIM,
                             "8-wave / 32-tree workgroups: two 16-tree groups share every network phase and walk together, two waves per SIMD (the second one fills the other's LDS / memory waits; MFMA and vector time add up)", dev),
                extra_config("B: CartPole-v1 discrete, 4096 trees, n_sims=100, 2x128 ReLU", CARTPOLE, 4096, 100, 4, [128, 128], 2, "relu",
                             B_KERNEL, None,
                             "tree-walk bound (8-9 levels per trace, about 0.2 evaluations per simulation -- counted from the trees --: the MFMA "
                             "fraction is small by construction, the HBM-side figure of SURVEY 8d is in roofline_hbm; selections are taken at "
                             "backup time and stored with the nodes, the descent follows them", dev, hbm=True, live=("B", "search_kernel")),
                extra_config("E (per GPU): Pendulum-v1, 1024 trees, n_sims=200, 4x1024 ELU", PENDULUM, 1024, 200, 3, [1024] * 4, 2, "elu",
                             "ls_team_kernel<2, 1024, false, 1, 2, 2, 1, 32>",
                             mlp_flops(3, [1024] * 4, 3), "persistent team kernel: one launch per search, 32 teams of 16 workgroups (one 64-unit slice of every "
                             "layer for the team's 32 trees each), hand-offs through global memory; traffic = L2 misses of the cross-XCD "
                             "activation hand-offs (DESIGN.md section 3), the weights stay L2-resident", dev, live=("E", "ls_team_kernel")),
                extra_config("E's network at 2048 trees per GPU", PENDULUM, 2048, 200, 3, [1024] * 4, 2, "elu",
                             "ls_team_kernel<2, 1024, false, 1, 2, 2, 1, 64>",
                             mlp_flops(3, [1024] * 4, 3), "the team kernel with teams of 64 trees (64 x 64 tiles: a third fewer staged bytes per MFMA than the "
                             "32-tree teams' 32 x 64), two workgroups per CU, short staging chunks", dev),
                extra_config("E's network at 3072 trees per GPU", PENDULUM, 3072, 200, 3, [1024] * 4, 2, "elu",
                             "ls_team_kernel<2, 1024, false, 1, 2, 3, 1, 64>",
                             mlp_flops(3, [1024] * 4, 3), "64-tree teams, three workgroups per CU: while one waits at a hand-off or walks its trees the "
                             "other two keep the matrix pipe busy", dev),
            ]
        traffic, traffic_note = None, ("not measured in the N > 1 loop (the search kernel is the one of the N = 1 line: same launch, same "
                                       "traffic; rocprofv3 child passes are not started from inside a running multi-rank job)")
        if config_d and B == N_TREES:
            traffic, src = profiled_traffic()
            traffic_note += f"; value from the committed N = 1 passes ({src})"
        if not config_d:
            if not args.no_live_traffic:
                traffic, why = live_traffic(B)
                traffic_note = ("measured in this run: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) over 5 launches of this kernel, "
                                "2 x FETCH + WRITE") if traffic is not None else f"live PMC passes unavailable ({why})"
            if traffic is None and B == N_TREES:
                traffic, src = profiled_traffic()
                traffic_note += f"; from the committed passes ({src})"
            if not args.no_live_traffic:
                busy, why = live_mfma_busy(B)
                extra["mfma_busy"] = ({"value": busy, "note": "hardware view of roofline.frac: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) of the "
                                                               "search kernel, one more rocprofv3 --pmc child pass of this run (it also counts the first layer's "
                                                               "and the heads' MFMAs, which the 134 144-FLOP figure leaves out)"}
                                      if busy is not None else {"value": None, "note": f"not measured ({why})"})
        out = {
            "metric": "MCTS sims/sec (whole node), Pendulum-v1 4096 trees n_sims=200, 1/2/4/8 GPU", "value": sims / elapsed, "unit": "sims/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "trees_per_gpu": B, "n_sims": N_SIMS, "parallelism": parallelism},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS,
                         "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": kname + " (template arguments: ENV 2 = Pendulum, HP = padded hidden width, NREG = hidden->hidden layers held in "
                                   "registers, tree storage 1 = LDS with 8-bit ids, mixture head, waves per workgroup -- eight, of which the first four walk the trees --, "
                                   "tree groups per workgroup, trees per group, 1 = compiled for epsilon 0 / lowest-index ties (the general kernel gives the same trees))",
                         "kernel_ms": kmean, "kernel_ms_median": kmed,
                         "note": "one launch = one whole search; achieved = trees x sims x 134144 FLOP / mean launch time (HIP events on the "
                                 "engine stream); policy/value MLP in fp32 MFMA, tree statistics in fp64; traffic = HBM bytes per launch from "
                                 "rocprofv3 PMC passes (see traffic_note); `value` is wall "
                                 "time over K steps of search + return_results (written by the same launch's epilogue) with everything resident in HBM, the PCIe-inclusive "
                                 "rate is extra.pcie_inclusive"},
            "extra": extra,
        }
        if not config_d:
            out["roofline_hbm"] = hbm_block(*walk_c, B, N_SIMS, kmean, traffic, traffic_note)
        if not args.no_cpu_baseline:
            # rank 0 only, after the timed region (the other ranks wait at the final barrier, inside --dist-timeout)
            out["cpu_baseline"] = cpu_baseline(short=config_d)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
