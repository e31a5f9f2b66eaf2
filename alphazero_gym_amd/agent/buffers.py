"""FIFO replay buffer with the reference's overwrite and minibatch rules (alphazero/agent/buffers.py:40-127)."""
from typing import List, Tuple

import numpy as np

Experience = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]


class ReplayBuffer:
    def __init__(self, max_size: int, batch_size: int) -> None:
        self.max_size, self.batch_size = max_size, batch_size
        self.sample_array = None
        self.clear()
        self.sample_index = 0

    def clear(self) -> None:
        self.experience: List[Experience] = []
        self.insert_index = 0
        self.size = 0

    def store(self, experience: Experience) -> None:
        """Append until full, then overwrite the oldest slot (buffers.py:75-82)."""
        if self.size < self.max_size:
            self.experience.append(experience)
            self.size += 1
        else:
            self.experience[self.insert_index] = experience
            self.insert_index += 1
            if self.insert_index >= self.size:
                self.insert_index = 0

    def reshuffle(self) -> None:
        self.sample_array = np.arange(self.size)
        np.random.shuffle(self.sample_array)
        self.sample_index = 0

    def __iter__(self):
        return self

    def __len__(self) -> int:
        return len(self.experience)

    def __next__(self) -> Experience:
        """Minibatches of batch_size; the last batch absorbs the remainder (< 2 batch_size) (buffers.py:108-123)."""
        if (self.sample_index + self.batch_size > self.size) and (not self.sample_index == 0):
            self.reshuffle()
            raise StopIteration
        assert self.sample_array is not None
        if self.sample_index + 2 * self.batch_size > self.size:
            indices = self.sample_array[self.sample_index:]
        else:
            indices = self.sample_array[self.sample_index:self.sample_index + self.batch_size]
        batch = [self.experience[i] for i in indices]
        self.sample_index += self.batch_size
        states, actions, counts, Qs, values = map(np.stack, zip(*batch))
        return states, actions, counts, Qs, values

    next = __next__
