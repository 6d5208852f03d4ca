"""Chunk iterator used by the reference's *_by_chunk methods (src/utils/batch.py:6-43)."""
import numpy as np
import torch


class BatchedData:
    def __init__(self, batch_size, data=None, **kwargs):
        self.batch_size = batch_size
        self.data = data if data is not None else []

    def __len__(self):
        assert self.batch_size is not None, "batch_size is not defined"
        return int(np.ceil(len(self.data) / self.batch_size))

    def __getitem__(self, idx):
        assert self.batch_size is not None, "batch_size is not defined"
        return self.data[idx * self.batch_size:(idx + 1) * self.batch_size]

    def cat(self, data, dim=0):
        self.data = data if len(self.data) == 0 else torch.cat([self.data, data], dim=dim)

    def append(self, data):
        self.data.append(data)

    def stack(self, dim=0):
        self.data = torch.stack(self.data, dim=dim)
