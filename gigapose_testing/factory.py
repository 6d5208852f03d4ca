"""Assembly helpers shared by bench.py, __graft_entry__.smoke() and the tests: the model wired as
the reference's Hydra config wires it (configs/model/large.yaml, ae_net/dinov2_l.yaml,
ist_net/resnet.yaml) with deterministic random-init weights, plus synthetic template sets/batches
(there is no network for checkpoints or datasets)."""
import tempfile
import types

import numpy as np
import torch

from gigapose_amd.ae_net import AENet
from gigapose_amd.gigaPose import GigaPose
from gigapose_amd.ist_net import ISTNet, Regressor, ResNet
from gigapose_amd.matching import LocalSimilarity
from gigapose_amd.vit import VARIANTS, Dinov2ViT

from . import synthetic as syn

IST_CFG = dict(n_heads=0, input_dim=3, input_size=256, initial_dim=128, block_dims=[128, 192, 256, 512],
               descriptor_size=256)


def build_model(variant="dinov2_vitl14", k=5, device="cuda", seed=0, log_dir=None, numerics=None):
    dim, depth, heads = VARIANTS[variant]
    vit = syn.fill_state_dict(Dinov2ViT(dim, depth, heads), seed + 1)
    ae = AENet(variant, vit, descriptor_size=dim, max_batch_size=64)
    ist = syn.fill_state_dict(ISTNet("resnet", ResNet(dict(IST_CFG)), Regressor(256, 256, True, True), 64), seed + 2)
    metric = LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=3)
    model = GigaPose("large", ae, ist, None, metric, None, 1000, log_dir or tempfile.mkdtemp(prefix="gigapose_"),
                     max_num_dets_per_forward=4, numerics=numerics)
    return model.eval().to(device)


class TemplateSet:
    """Stand-in for the reference TemplateSet (src/dataloader/template.py:55-81): item i has
    .rgb (N,3,224,224) .mask (N,224,224) .K (3,3) .M (N,3,3) .poses (N,4,4)."""

    def __init__(self, O, N, seed):
        tK, tM, tP = syn.template_geometry(seed + 1, O, N)
        self.items, self._rgb, self._mask = [], [], []
        for o in range(O):
            rgb, mask = syn.template_images(seed + 10 + o, N)
            self._rgb.append(rgb)
            self._mask.append(mask)
            self.items.append(types.SimpleNamespace(rgb=torch.from_numpy(rgb), mask=torch.from_numpy(mask),
                                                    K=torch.from_numpy(tK[o]), M=torch.from_numpy(tM[o]),
                                                    poses=torch.from_numpy(tP[o])))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def crops(self, seed, B, device):
        """B query crops = noisy copies of randomly chosen templates, with geometry."""
        rs = np.random.RandomState(seed)
        O, N = len(self.items), self._rgb[0].shape[0]
        labels = rs.randint(1, O + 1, B)
        views = rs.randint(0, N, B)
        img = np.stack([self._rgb[l - 1][v] for l, v in zip(labels, views)])
        msk = np.stack([self._mask[l - 1][v] for l, v in zip(labels, views)])
        img = ((img + 0.1 * rs.standard_normal(img.shape).astype(np.float32)) * msk[:, None]).astype(np.float32)
        K, M = syn.crop_geometry(seed + 2, B)
        t = lambda a: torch.from_numpy(a).to(device)
        return dict(tar_img=t(img), tar_mask=t(msk), tar_K=t(K), tar_M=t(M), labels=torch.from_numpy(labels), views=views)
