"""CPU, build container only (needs /root/reference): the mirror classes a Hydra `_target_` swap instantiates
(reference test.py:43-47; configs/model/large.yaml:1, ae_net/dinov2_l.yaml:1, ist_net/resnet.yaml:1) take the reference's
constructor arguments -- same names, same order, same defaults -- and expose the methods / attributes the reference's
callers use.  Skipped where the reference tree is absent (the GPU box)."""
import inspect

import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="needs /root/reference")

PAIRS = [  # (reference module, class, mirror module, methods the callers use)
    ("src.models.gigaPose", "GigaPose", "gigapose_amd.gigaPose", ["set_template_data", "eval_retrieval", "filter_and_save", "test_step", "on_test_epoch_end"]),
    ("src.models.network.ae_net", "AENet", "gigapose_amd.ae_net", ["forward", "forward_by_chunk"]),
    ("src.models.matching", "LocalSimilarity", "gigapose_amd.matching", ["test", "val"]),
    ("src.models.network.ist_net", "ISTNet", "gigapose_amd.ist_net", ["forward_by_chunk", "inference", "inference_by_chunk"]),
    ("src.models.network.ist_net", "Regressor", "gigapose_amd.ist_net", []),
    ("src.models.network.resnet", "ResNet", "gigapose_amd.ist_net", ["forward"]),
    ("src.models.poses", "ObjectPoseRecovery", "gigapose_amd.poses", ["forward_ransac", "forward_recovery"]),
]


def params(fn):
    return [(p.name, p.kind, p.default) for p in list(inspect.signature(fn).parameters.values())[1:]]  # without self


@pytest.mark.parametrize("ref_mod,cls,mine_mod,methods", PAIRS)
def test_constructor_and_method_signatures_match_the_reference(ref_mod, cls, mine_mod, methods):
    import importlib

    ref_shim.install()
    ref = getattr(importlib.import_module(ref_mod), cls)
    mine = getattr(importlib.import_module(mine_mod), cls)
    r, m = params(ref.__init__), params(mine.__init__)
    assert [(n, k) for n, k, _ in m] == [(n, k) for n, k, _ in r], f"{cls}.__init__ parameters differ: {m} vs {r}"
    for (n, _, dm), (_, _, dr) in zip(m, r):
        assert dm == dr or (dm is inspect.Parameter.empty) == (dr is inspect.Parameter.empty) and repr(dm) == repr(dr), f"{cls}.__init__ default of {n}"
    for name in methods:
        assert hasattr(mine, name), f"{cls}.{name} missing"
        rp, mp = params(getattr(ref, name)), params(getattr(mine, name))
        # the mirror may ADD trailing keyword parameters with defaults; the reference's own must be there, in order
        assert [n for n, _, _ in mp][:len(rp)] == [n for n, _, _ in rp], f"{cls}.{name}: {mp} vs {rp}"
        assert all(d is not inspect.Parameter.empty for _, _, d in mp[len(rp):]), f"{cls}.{name}: extra parameter without a default"


def test_gigapose_accepts_the_hydra_config_kwargs():
    """configs/model/large.yaml passes refiner / checkpoint_path through **kwargs; test.py:67-74 assigns attributes afterwards."""
    import tempfile

    from gigapose_amd.gigaPose import GigaPose
    from gigapose_amd.matching import LocalSimilarity

    m = GigaPose(model_name="large", ae_net=None, ist_net=None, training_loss=None, testing_metric=LocalSimilarity(5, 0.5, 3), optim_config=None,
                 log_interval=1000, log_dir=tempfile.mkdtemp(), max_num_dets_per_forward=None, test_setting="localization",
                 refiner=None, checkpoint_path="gigaPose_v1.ckpt")
    for attr in ["template_datasets", "test_dataset_name", "max_num_dets_per_forward", "run_id", "log_interval"]:
        assert hasattr(m, attr)
