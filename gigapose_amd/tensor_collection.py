"""Dict-of-tensors + pandas `infos` container crossing every boundary of the hot path.

Mirrors the surface of the reference container the callers rely on
(src/megapose/utils/tensor_collection.py:45-212): attribute access to tensors,
register_tensor, __getitem__ (row selection on tensors + infos), cat_df, clone, to/cuda/cpu,
len() == len(infos).  The file-system gather used by MegaPose training is out of scope.
"""
import pandas as pd
import torch


class TensorCollection:
    def __init__(self, **tensors):
        object.__setattr__(self, "_tensors", dict())
        for k, v in tensors.items():
            self.register_tensor(k, v)

    def register_tensor(self, name, tensor):
        self._tensors[name] = tensor

    def delete_tensor(self, name):
        del self._tensors[name]

    @property
    def tensors(self):
        return self._tensors

    @property
    def device(self):
        return next(iter(self._tensors.values())).device

    def __getattr__(self, name):
        tensors = object.__getattribute__(self, "_tensors")
        if name in tensors:
            return tensors[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self._tensors:
            self._tensors[name] = value
        else:
            object.__setattr__(self, name, value)

    def __getitem__(self, ids):
        return TensorCollection(**{k: v[ids] for k, v in self._tensors.items()})

    def to(self, where):
        for k, v in self._tensors.items():
            self._tensors[k] = v.to(where)
        return self

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def float(self):
        return self.to(torch.float)

    def clone(self):
        return TensorCollection(**{k: v.clone() for k, v in self._tensors.items()})

    def __repr__(self):
        rows = [f"    {k}: {tuple(t.shape)} {t.dtype} {t.device}," for k, t in self._tensors.items()]
        return self.__class__.__name__ + "(\n" + "\n".join(rows) + "\n)"


class PandasTensorCollection(TensorCollection):
    def __init__(self, infos, **tensors):
        super().__init__(**tensors)
        self.infos = infos.reset_index(drop=True)
        self.meta = dict()

    def __len__(self):
        return len(self.infos)

    def __getitem__(self, ids):
        infos = self.infos.iloc[ids].reset_index(drop=True)
        return PandasTensorCollection(infos, **{k: v[ids] for k, v in self._tensors.items()})

    def cat_df(self, other):
        for k in list(self._tensors):
            self._tensors[k] = torch.cat([self._tensors[k], other._tensors[k]], dim=0)
        return PandasTensorCollection(infos=self.infos, **self._tensors)

    def clone(self):
        return PandasTensorCollection(self.infos.copy(), **{k: v.clone() for k, v in self._tensors.items()})

    def __repr__(self):
        return super().__repr__()[:-1] + "-" * 40 + "\n    infos:\n" + repr(self.infos) + "\n)"


def concatenate(datas):
    datas = [d for d in datas if len(d) > 0]
    if not datas:
        return PandasTensorCollection(infos=pd.DataFrame())
    infos = pd.concat([d.infos for d in datas], axis=0, sort=False).reset_index(drop=True)
    tensors = {k: torch.cat([getattr(d, k) for d in datas], dim=0) for k in datas[0].tensors}
    return PandasTensorCollection(infos=infos, **tensors)
