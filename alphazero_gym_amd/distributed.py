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


def gather_replay_rows(rows: torch.Tensor, device=None, counts=None, total: int = None) -> torch.Tensor:
    """All-gather the per-rank row blocks into [B_total, row] ordered by rank (= by global tree id).

    The block lengths are known without asking anybody when the rows are one per game of a `shard_range` partition: pass
    `total` (games of the whole job; rows per game are inferred from this rank's block) or `counts` (rows of every rank).
    Equal blocks -- config D, the common case -- then cost ONE collective into one tensor, no padding, no host sync.  Only when
    neither is given are the lengths exchanged first (one small all-gather + a host read per rank).  Blocks that differ in length
    (by at most one game when `total` is not a multiple of the world size) are padded to the longest for the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    rows = (rows.to(device) if device is not None else rows).contiguous()
    if counts is None and total is not None:
        lo, hi = shard_range(total, dist.get_rank(), world)
        per_game, rem = divmod(rows.shape[0], max(hi - lo, 1))
        if hi - lo == 0 or rem:
            raise ValueError(f"rank {dist.get_rank()} holds {rows.shape[0]} rows for {hi - lo} games of {total}")
        counts = [(b - a) * per_game for a, b in (shard_range(total, r, world) for r in range(world))]
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
