"""Replay storage for the optimiser step.

``ReplayBuffer`` keeps the observable behaviour of the reference's class (alphazero/agent/buffers.py:40-127: slot reuse
order, minibatch sizes, attribute names the run scripts touch) on a columnar ring: one preallocated array per experience
field, written in place and gathered per minibatch with one fancy-index per column (no per-sample Python objects, no
``np.stack`` of tuples).  ``DeviceReplay`` applies the same rules to the engine's device-resident ring (azg_selfplay_*):
rows never leave HBM, minibatches are gathered on the GPU and come out as torch tensors.
"""
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

Experience = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]
N_FIELDS = 5   # state, actions, counts, Qs, value target


class _FifoCursor:
    """Slot bookkeeping shared by both buffers: fill slots 0..max-1 in order, then recycle them cyclically starting at 0
    (buffers.py:75-82).  ``size`` and ``insert`` are the reference's ``size`` / ``insert_index``."""

    __slots__ = ("capacity", "size", "insert")

    def __init__(self, capacity: int):
        self.capacity, self.size, self.insert = int(capacity), 0, 0

    def take(self) -> int:
        if self.size < self.capacity:
            self.size += 1
            return self.size - 1
        slot = self.insert
        self.insert = (slot + 1) % self.size
        return slot

    def reset(self) -> None:
        self.size = self.insert = 0


class _EpochSampler:
    """One epoch = the shuffled slot ids cut into batches of ``batch_size``; a tail shorter than a batch is appended to the
    last full one (so batches have batch_size .. 2*batch_size-1 samples); asking for a batch past the end reshuffles for the
    next epoch and stops the iteration (buffers.py:84-123)."""

    def __init__(self, batch_size: int):
        self.batch_size = int(batch_size)
        self.order: Optional[np.ndarray] = None
        self.pos = 0

    def reshuffle(self, n: int) -> None:
        self.order = np.arange(n)
        np.random.shuffle(self.order)   # numpy's global stream, like the reference (run scripts seed it)
        self.pos = 0

    def next_ids(self, n: int) -> Optional[np.ndarray]:
        start, bs = self.pos, self.batch_size
        if start != 0 and start + bs > n:
            return None
        if self.order is None:
            raise AssertionError("reshuffle() has not been called")
        ids = self.order[start:] if start + 2 * bs > n else self.order[start:start + bs]
        self.pos = start + bs
        return ids


class ReplayBuffer:
    """Host replay buffer.  ``pinned=True`` allocates the columns in page-locked memory (asynchronous H2D copies of minibatches)."""

    def __init__(self, max_size: int, batch_size: int, pinned: bool = False) -> None:
        self.max_size, self.batch_size = int(max_size), int(batch_size)
        self._pinned = pinned
        self._cursor = _FifoCursor(self.max_size)
        self._sampler = _EpochSampler(self.batch_size)
        self._cols: Optional[List[np.ndarray]] = None
        self._keep = []   # pinned torch tensors backing the columns

    # ---- the reference's attribute surface
    @property
    def size(self) -> int:
        return self._cursor.size

    @property
    def insert_index(self) -> int:
        return self._cursor.insert

    @property
    def sample_array(self) -> Optional[np.ndarray]:
        return self._sampler.order

    @property
    def sample_index(self) -> int:
        return self._sampler.pos

    @property
    def experience(self) -> List[Experience]:
        """The stored experiences in slot order, as tuples of array views."""
        if self._cols is None:
            return []
        return [tuple(col[i] for col in self._cols) for i in range(self.size)]   # type: ignore[misc]

    def clear(self) -> None:
        self._cursor.reset()

    def __len__(self) -> int:
        return self.size

    # ---- storage
    def _allocate(self, fields: Sequence[np.ndarray]) -> None:
        self._cols = []
        for f in fields:
            shape = (self.max_size,) + f.shape
            if self._pinned:
                import torch
                t = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, f.dtype)).dtype).pin_memory()
                self._keep.append(t)
                self._cols.append(t.numpy())
            else:
                self._cols.append(np.empty(shape, dtype=f.dtype))

    def store(self, experience: Experience) -> None:
        fields = [np.asarray(f) for f in experience]
        if len(fields) != N_FIELDS:
            raise ValueError(f"an experience has {N_FIELDS} fields: state, actions, counts, Qs, value target")
        if self._cols is None:
            self._allocate(fields)
        slot = self._cursor.take()
        for col, f in zip(self._cols, fields):
            if f.shape != col.shape[1:]:
                raise ValueError(f"experience field of shape {f.shape} in a buffer of {col.shape[1:]}: shapes must stay fixed "
                                 "(they are, for a fixed n_rollouts)")
            col[slot] = f

    # ---- sampling
    def reshuffle(self) -> None:
        self._sampler.reshuffle(self.size)

    def __iter__(self) -> Iterator[Experience]:
        return self

    def __next__(self) -> Experience:
        ids = self._sampler.next_ids(self.size)
        if ids is None:
            self.reshuffle()
            raise StopIteration
        assert self._cols is not None
        return tuple(col[ids] for col in self._cols)   # type: ignore[return-value]

    next = __next__


class _DeviceArray:
    """A device pointer with the CUDA array interface (torch.as_tensor wraps it without a copy)."""

    def __init__(self, ptr: int, shape: Tuple[int, ...]):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}


class DeviceReplay:
    """The engine's device-resident replay ring behind the same sampling rules.

    Rows are ``[obs | actions[K] | counts[K] | Qs[K] | V_target]`` float32 (include/azgym.h), one block of ``n_trees`` rows per
    self-play step; the ring overwrites its oldest step when started with ``fifo=True``.  ``rows()`` is a zero-copy torch view
    of the valid part of the ring; iteration yields minibatches ``(states, actions, counts, Qs, values)`` as device tensors
    gathered on the GPU."""

    def __init__(self, engine, batch_size: int, device=None, wrap=None):
        """``wrap(ptr, shape, device) -> tensor``: how the ring's device pointer becomes a tensor; default: zero-copy through the
        CUDA array interface on the engine's GPU."""
        import torch
        self.engine = engine
        self.batch_size = int(batch_size)
        if device is None:
            device = torch.device("cuda", int(engine.cfg.device_id))
        self.device = torch.device(device)
        self._wrap = wrap if wrap is not None else (lambda ptr, shape, dev: torch.as_tensor(_DeviceArray(ptr, shape), device=dev))
        self._ptr = None
        self._attach()
        self._sampler = _EpochSampler(self.batch_size)

    def _attach(self):
        """(Re-)wrap the engine's ring: azg_selfplay_begin allocates a new one, the view of the old one must not be used again."""
        ptr, cap_rows, row_len = self.engine.selfplay_rows_device()
        if ptr != self._ptr or (cap_rows, row_len) != (self.capacity_rows, self.row_len):
            self._ptr, self.capacity_rows, self.row_len = ptr, cap_rows, row_len
            self.K = (row_len - 1 - self.engine.s_obs) // 3
            self._ring = self._wrap(ptr, (cap_rows, row_len), self.device)

    capacity_rows = row_len = 0

    @property
    def ring(self):
        """The whole ring [capacity_rows, row_len] as a zero-copy tensor (slots beyond ``size`` are undefined)."""
        self._attach()
        return self._ring

    @property
    def size(self) -> int:
        """Valid rows (ReplayBuffer.size in rows)."""
        return self.engine.selfplay_ring()[0] * self.engine.n_trees

    def __len__(self) -> int:
        return self.size

    def rows(self):
        """Zero-copy view [size, row_len] of the stored rows in slot order (waits for the engine's stream first)."""
        self.engine.sync()
        return self.ring[:self.size]

    def split(self, rows):
        so, K = self.engine.s_obs, self.K
        return rows[:, :so], rows[:, so:so + K], rows[:, so + K:so + 2 * K], rows[:, so + 2 * K:so + 3 * K], rows[:, -1]

    def reshuffle(self) -> None:
        self._sampler.reshuffle(self.size)

    def __iter__(self):
        return self

    def __next__(self):
        import torch
        n = self.size
        ids = self._sampler.next_ids(n)
        if ids is None:
            self.reshuffle()
            raise StopIteration
        idx = torch.from_numpy(np.ascontiguousarray(ids)).to(self.device, non_blocking=True)
        return self.split(self.rows().index_select(0, idx))

    next = __next__
