"""TEST INFRASTRUCTURE ONLY -- import shim that lets the *unmodified* reference
(`/root/reference`, nv-nguyen/gigapose) run on CPU inside the build container.

Only `oracle/make_goldens.py` (golden-vector generation) and the cross-check
tests that are skipped when `/root/reference` is absent may import this module.
Nothing under `gigapose_amd/` imports it; it does not exist on the GPU box.

What it does:
  * imports `transformers` first (it breaks if it later sees a stub torchvision),
  * installs permissive stub packages for third-party modules the reference
    imports at module scope but that are not installed here,
  * aliases `megapose.*` -> `src.megapose.*` (reference `setup.cfg` does this via
    package_dir),
  * replaces `pytorch_lightning.LightningModule` by a plain nn.Module subclass.
No reference source is copied; the reference is imported where it lies.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("GIGAPOSE_REFERENCE_ROOT", "/root/reference")

_MISSING = {
    "cv2", "omegaconf", "wandb", "pytorch_lightning", "torchvision", "skimage",
    "trimesh", "matplotlib", "bop_toolkit_lib", "hydra", "webdataset", "imageio",
    "pyrender", "distinctipy",
}


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models"))


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = mock.MagicMock(name=f"{self.__name__}.{k}")
        setattr(self, k, v)
        return v


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in _MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


class _MegaposeAlias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name == "megapose" or name.startswith("megapose."):
            return importlib.machinery.ModuleSpec(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("src." + spec.name)

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Idempotently install the shim; returns the reference `src` package."""
    global _installed
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if not _installed:
        import transformers  # noqa: F401  (must precede the torchvision stub)
        from transformers import Dinov2Config, Dinov2Model  # noqa: F401
        import torch

        nthreads = torch.get_num_threads()
        sys.meta_path[:0] = [_MegaposeAlias(), _StubFinder()]
        import pytorch_lightning as pl

        class _LightningModule(torch.nn.Module):
            global_rank = 0
            logger = None

            @property
            def device(self):
                try:
                    return next(self.parameters()).device
                except StopIteration:
                    return torch.device("cpu")

        pl.LightningModule = _LightningModule
        sys.path.insert(0, REFERENCE_ROOT)
        import src  # noqa: F401
        import src.megapose  # noqa: F401  (sets OMP/MKL_NUM_THREADS=1 in os.environ)

        torch.set_num_threads(nthreads)
        _installed = True
    import src

    return src


class HFDinov2Backbone:
    """Factory for the stand-in DINOv2 backbone: HF Dinov2Model,
    random-init, exposing forward_features(x)["x_prenorm"] = hidden_states[-1]."""

    @staticmethod
    def build(hidden_size, num_layers, num_heads, seed=0):
        import torch
        from transformers import Dinov2Config, Dinov2Model

        cfg = Dinov2Config(
            hidden_size=hidden_size, num_hidden_layers=num_layers,
            num_attention_heads=num_heads, image_size=224, patch_size=14,
        )
        torch.manual_seed(seed)
        model = Dinov2Model(cfg).eval()

        class _Adapter(torch.nn.Module):
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward_features(self, x):
                out = self.m(pixel_values=x, output_hidden_states=True)
                return {"x_prenorm": out.hidden_states[-1]}

        return _Adapter(model)
