"""CPU, build container (needs the reference's config files): the reference's OWN model config -- configs/model/large.yaml with its
defaults -- instantiates into the gigapose_amd classes after the `_target_` swap of INTEGRATION.md section 1, every YAML key being a
constructor argument the mirror accepts; the resolved config is committed (tests/golden/model_cfg_resolved.json) so that the GPU
box, which has no reference tree, builds the model from the same file (tests/test_gpu_dropin_flow.py)."""
import json
import os

import pytest

import dropin_flow as df
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="needs /root/reference")
FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "model_cfg_resolved.json")


def test_reference_yaml_instantiates_into_the_mirrors(tmp_path):
    cfg = df.compose_reference_model_cfg(ref_shim.REFERENCE_ROOT, save_dir="<save_dir>")
    want = json.loads(json.dumps(cfg))                        # plain JSON types
    if os.environ.get("GIGAPOSE_WRITE_FIXTURES") == "1" or not os.path.exists(FIXTURE):
        json.dump(want, open(FIXTURE, "w"), indent=1, sort_keys=True)
    assert json.load(open(FIXTURE)) == want, "tests/golden/model_cfg_resolved.json is stale: regenerate with GIGAPOSE_WRITE_FIXTURES=1"
    cfg["log_dir"] = str(tmp_path)
    cfg["test_setting"] = "localization"                      # test.py:46 sets it on the config before instantiating
    model = df.instantiate(cfg)
    from gigapose_amd.ae_net import AENet
    from gigapose_amd.gigaPose import GigaPose
    from gigapose_amd.ist_net import ISTNet, Regressor, ResNet
    from gigapose_amd.matching import LocalSimilarity
    from gigapose_amd.vit import Dinov2ViT

    assert isinstance(model, GigaPose) and isinstance(model.ae_net, AENet) and isinstance(model.ae_net.dinov2_model, Dinov2ViT)
    assert isinstance(model.ist_net, ISTNet) and isinstance(model.ist_net.backbone, ResNet) and isinstance(model.ist_net.regressor, Regressor)
    assert isinstance(model.testing_metric, LocalSimilarity) and model.testing_metric.k == 5
    assert model.ae_net.dinov2_model.dim == 1024 and model.ae_net.dinov2_model.depth == 24
    assert model.test_setting == "localization" and model.model_name == "large"
    # what test.py assigns after construction (test.py:67-74)
    for attr in ["template_datasets", "test_dataset_name", "max_num_dets_per_forward", "run_id", "log_interval"]:
        assert hasattr(model, attr)
    # the released checkpoint's parameter names (ae_net.dinov2_model.*, ist_net.backbone.*, ist_net.regressor.*) are the state dict's
    keys = list(model.state_dict())
    assert "ae_net.dinov2_model.blocks.23.mlp.fc2.weight" in keys and "ist_net.backbone.layer4_outconv.weight" in keys
    assert any(k.startswith("ist_net.regressor.scale_predictor") for k in keys)
