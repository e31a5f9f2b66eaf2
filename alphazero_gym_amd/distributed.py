"""Multi-GPU self-play: one process per GPU, trees sharded by global tree id, no collective inside the search.

The reference is single-process; independent games shard embarrassingly.  Rank g of G owns the global
tree ids [g*B/G, (g+1)*B/G); RNG streams and synthetic roots are keyed by the global id, so a tree's result does not
depend on G.  Exactly two collectives exist (RCCL over xGMI when the backend is "nccl"; "gloo" in CPU tests):
an all-gather of replay rows after self-play and a broadcast of the updated weights after the optimiser step."""
from typing import List, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) global tree ids of `rank`; sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_replay_rows(states: np.ndarray, actions: np.ndarray, counts: np.ndarray, Q: np.ndarray, v_target: np.ndarray) -> torch.Tensor:
    """One float32 row per game: state | actions[K] | counts[K] | Q[K] | V_target  (196 B for Pendulum, K = 15)."""
    B = states.shape[0]
    return torch.from_numpy(np.concatenate([
        states.reshape(B, -1).astype(np.float32), actions.reshape(B, -1).astype(np.float32), counts.reshape(B, -1).astype(np.float32),
        Q.reshape(B, -1).astype(np.float32), v_target.reshape(B, 1).astype(np.float32)], axis=1))


def unpack_replay_rows(rows: torch.Tensor, state_dim: int, K: int):
    r = rows.cpu().numpy()
    s = r[:, :state_dim]
    a = r[:, state_dim:state_dim + K]
    c = r[:, state_dim + K:state_dim + 2 * K]
    q = r[:, state_dim + 2 * K:state_dim + 3 * K]
    return s, a, c, q, r[:, -1]


def gather_replay_rows(rows: torch.Tensor, device=None, counts=None, total: int = None, rows_per_game: int = None) -> torch.Tensor:
    """All-gather the per-rank row blocks into [B_total, row] ordered by rank (= by global tree id).

    The block lengths are known without asking anybody in two cases: `counts` (rows of every rank, the same list on every rank), or
    `total` + `rows_per_game` -- `total` games of the WHOLE JOB partitioned by `shard_range`, `rows_per_game` rows for each of them
    (1 for one search per game, n_steps for a block of self-play steps): rank r then holds (games of r) x rows_per_game rows, a rank
    without games holds none.  Equal blocks -- config D, the common case -- then cost ONE collective into one tensor, no padding, no
    host sync.  `total` without `rows_per_game` infers the latter from this rank's own block, which is only defined when EVERY rank
    holds at least one game and the same number of rows per game: a precondition on the caller, checked on each rank against its own
    block only (a rank that fails it raises while the others wait in the collective until the process group's timeout).
    When neither is given the lengths are exchanged first (one small all-gather + a host read per rank): any shapes.
    Blocks that differ in length are padded to the longest for the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    rows = (rows.to(device) if device is not None else rows).contiguous()
    if counts is None and total is not None:
        spans = [shard_range(total, r, world) for r in range(world)]
        if rows_per_game is None:
            lo, hi = spans[dist.get_rank()]
            rows_per_game, rem = divmod(rows.shape[0], max(hi - lo, 1))
            if hi - lo == 0 or rem:
                raise ValueError(f"rank {dist.get_rank()} holds {rows.shape[0]} rows for {hi - lo} games of {total}: pass rows_per_game "
                                 "(or counts) when a rank may be without games")
        counts = [(b - a) * int(rows_per_game) for a, b in spans]
    if counts is None:
        n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
        got = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(got, n)
        counts = [int(c.item()) for c in got]
    counts = [int(c) for c in counts]
    if len(counts) != world or counts[dist.get_rank()] != rows.shape[0]:
        raise ValueError(f"counts {counts} do not describe this rank's {rows.shape[0]} rows")
    longest = max(counts)
    if min(counts) == longest:
        out = rows.new_empty((world * longest,) + tuple(rows.shape[1:]))
        dist.all_gather_into_tensor(out, rows)
        return out
    if rows.shape[0] < longest:
        rows = torch.cat([rows, rows.new_zeros((longest - rows.shape[0],) + tuple(rows.shape[1:]))], dim=0)
    out = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(out, rows)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def broadcast_weights(model: torch.nn.Module, src: int = 0) -> None:
    """Rank `src` trained; everyone else receives the parameters (one flat buffer: KBs to a few MB, latency-bound)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    params: List[torch.Tensor] = list(model.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            # in-place copy INTO the Parameter (not its .data alias): bumps Parameter._version, which is what
            # BatchedMCTS.sync_weights watches to decide whether the engine needs the new weights
            p.copy_(flat[off:off + n].view_as(p))
            off += n
