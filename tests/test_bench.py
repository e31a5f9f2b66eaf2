"""bench.py: the pieces that run without a GPU, and (-m gpu) its N > 1 loop end to end on one GPU (gloo, every rank on GPU 0)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_flop_and_core_counts():
    import bench
    assert bench.FLOP_PER_SIM == 134144                            # 2x256 Pendulum
    assert bench.mlp_flops(3, [1024] * 4, 3) == 6303744            # config E
    assert bench.mlp_flops(4, [128, 128], 3) == 34560              # config B, per evaluation
    assert 1 <= bench.physical_cores() <= (os.cpu_count() or 1)


def test_no_profiler_child_passes_under_a_profiler(monkeypatch):
    """bench.py's live PMC passes start rocprofv3 children; when bench.py itself already runs under rocprofv3 the
    tool library is preloaded into every child, and a nested rocprofv3 would re-exec a process whose GPU runtime is initialised.
    live_traffic() must refuse there, and the children's environment must carry nothing a profiler left behind."""
    import bench
    for var, val in (("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so"), ("ROCPROF_COUNTER_COLLECTION", "1"),
                     ("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")):
        monkeypatch.setenv(var, val)
        assert bench.under_profiler()
        assert bench.live_traffic(16) == (None, "already under a profiler")
        env = bench.clean_child_env()
        assert "LD_PRELOAD" not in env and not [k for k in env if k.startswith("ROCP")] and env["TMPDIR"] == "/tmp"
        monkeypatch.delenv(var)
    assert not bench.under_profiler()


def test_walk_bytes_of_a_widening_search_are_counted_from_the_trees():
    """roofline_hbm of the headline line: L / C / E of SURVEY 8d's byte formula reconstructed from a searched batch (oracle double):
    the children scored are a function of every node's visit count under the widening law; the reconstruction must land on the
    survey's probe values (L 2.64, C 11.45, E 1.0 for 200 rollouts) and on an exact recount for a small case."""
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from alphazero_gym_amd import _capi
    e = O.OracleEngine(n_trees=64, n_sims=200, **bench.PENDULUM)
    e.set_weights(_capi.make_desc(3, bench.HIDDEN, 2, "elu"), O.make_weights(34, 3, bench.HIDDEN, 2))
    e.search(e.synthetic_roots())
    per_sim, L, C, E = bench.tree_walk_bytes_continuous(e.dump_tree(), 200, 1.0, 0.5)
    e.close()
    assert E == 1.0 and 2.3 < L < 3.0 and 10.0 < C < 13.0 and 400 < per_sim < 520
    # every node's final child count must equal what the reconstruction's law gives for its visit count (the same walk the formula takes)
    d = None
    e = O.OracleEngine(n_trees=4, n_sims=60, **bench.PENDULUM)
    e.set_weights(_capi.make_desc(3, bench.HIDDEN, 2, "elu"), O.make_weights(34, 3, bench.HIDDEN, 2))
    e.search(e.synthetic_roots())
    d = e.dump_tree()
    e.close()
    import math
    for t in range(4):
        n = int(d["n_records"][t])
        kids = np.bincount(d["parent"][t][1:n], minlength=n)
        for j in range(n):
            K = 1 if j == 0 else 0
            for v in range(int(d["node_n"][t][j])):
                if math.ceil((v + 1) ** 0.5) > K:
                    K += 1
            assert K == kids[j], (t, j, K, kids[j])


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True)
    assert p.returncode != 0 and "does not match WORLD_SIZE" in (p.stderr + p.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_multi_rank_loop_on_one_gpu(ranks):
    """`python bench.py --gpus N` spawns its own ranks; config D's step (self-play step, all-gather of the step's rows, weight
    broadcast + engine re-sync) runs with gloo and every rank on GPU 0: N = 2, 4 and 8 -- eight tree_id_base offsets, eight engines
    on one device, an eight-way gather, 256 games per rank (the shape of the driver's 8-GPU run has run once
    before the driver tries it), followed by the config E leg (4x1024 network, every rank its own shard of trees).  The N > 1 line carries the CPU baseline, a note about the traffic figure and
    every rank's host-side timings of the collectives (min / median / max over the ranks) with the world size each rank saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    trees = "256" if ranks == 8 else "512"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "5", "--warmup", "2", "--trees", trees,
                        "--bcast-every", "2", "--backend", "gloo", "--same-device", "--e-trees", "64", "--e-searches", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == ranks and out["steps"] == 5 and out["scaling"] == "weak"
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"
    assert "traffic_note" in out["roofline"]
    assert out["value"] > 0 and out["extra"]["search_only"]["sims_per_s"] >= out["value"] * 0.5
    assert "config D" in out["config"]["workload"]
    c = out["extra"]["collectives"]
    assert c["world_size"] == ranks and c["world_size_seen"] == [ranks]
    pr = c["per_rank_ms"]
    assert pr["gathers_timed"] >= 1 and pr["bcasts_timed"] >= 1
    for k in ("gather_ms", "bcast_ms", "step_ms"):
        assert 0 < pr[k]["min"] <= pr[k]["median"] <= pr[k]["max"], (k, pr[k])
    # BASELINE.json configs[4] under N > 1: every rank searched its own shard of the 4x1024 network's trees
    # (64 here; ranks that share one GPU may see their team kernels fall back to the per-layer launches: reported, not an error)
    e = out["extra"]["config_e_nrank"]
    assert f"{ranks * 64} trees = 64 per GPU" in e["config"] and e["searches"] == 2
    assert e["sims_per_s"] > 0 and len(e["team_kernel_fallbacks_per_rank"]) == ranks
    km = e["kernel_ms_per_search"]
    assert 0 < km["min"] <= km["median"] <= km["max"]
    assert all(k.startswith(("ls_team_kernel", "lock-step", "ls_")) for k in e["kernels"]), e["kernels"]
    assert 0 < e["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_a_dead_rank_fails_the_run_instead_of_hanging_it():
    """one of two ranks exits right after the rendezvous (AZG_BENCH_DIE_RANK, a test hook).  The survivor's
    collectives are bounded (--dist-timeout) and torch.distributed.run tears the job down: bench.py exits non-zero, well inside
    the test's own time limit, and prints no JSON line."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AZG_BENCH_DIE_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--trees", "256",
                        "--backend", "gloo", "--same-device", "--dist-timeout", "30", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 300


def _bench_json(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("warmup", [3, 4])
def test_gathered_rows_are_the_previous_steps_rows(warmup):
    """The two-slot replay ring keeps turning across the warm-up, the timed loop and the collective-free loop; the slot that is
    all-gathered while step s searches must be the one step s - 1 wrote, whatever the parity of the warm-up (with an
    odd warm-up the loop used to gather the slot being overwritten).  --verify-gather finds the slot a step wrote by comparing
    the ring before and after it and checks every gathered block against it.  (Rows are gathered in blocks of --gather-every
    steps; the ring holds two blocks.)"""
    out = _bench_json(["--gpus", "2", "--steps", "5", "--warmup", str(warmup), "--trees", "256", "--bcast-every", "2", "--gather-every", "2",
                       "--backend", "gloo", "--same-device", "--verify-gather"])
    c = out["extra"]["collectives"]
    assert c["gathers_verified"] == c["gathers"] >= (warmup + 5 - 1) // 2   # a block of two finished steps per gather, across both runs


@pytest.mark.gpu
def test_config_d_loop_over_rccl_with_one_rank():
    """The `nccl` branch on hardware: bench.py --config-d under torch.distributed.run with ONE rank -- all_gather_into_tensor on the
    engine-owned ring (the zero-copy view azg_selfplay_rows_device exports) and the weight broadcast into HBM followed by
    azg_set_weights_device go through RCCL with world size 1.  Proves RCCL accepts engine-owned memory and that the stream
    ordering between the engine's stream and RCCL's holds (every gathered block verified), before a multi-GPU node sees it."""
    out = _bench_json(["--gpus", "1", "--config-d", "--steps", "6", "--warmup", "3", "--trees", "1024", "--bcast-every", "2", "--gather-every", "2",
                       "--backend", "nccl", "--verify-gather"])
    assert out["n_gpus"] == 1 and "config D" in out["config"]["workload"]
    c = out["extra"]["collectives"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["gathers_verified"] == c["gathers"] >= 4
    assert c["weight_sync"].startswith("device to device")
    assert out["value"] > 0
