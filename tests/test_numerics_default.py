"""CPU: which numerics the drop-in selects (the bench must measure the mode a maintainer gets)."""
import pytest


def test_product_default_is_split(monkeypatch):
    monkeypatch.delenv("GIGAPOSE_NUMERICS", raising=False)
    from gigapose_amd import _lib, factory
    from gigapose_amd.ist_net import ResNet
    from gigapose_amd.matching import LocalSimilarity
    from gigapose_amd.vit import Dinov2ViT

    assert _lib.default_numerics() == "split"
    assert Dinov2ViT(384, 1, 6).numerics == "split"
    assert LocalSimilarity(k=5, sim_threshold=0.5, patch_threshold=3).numerics == "split"
    assert ResNet(dict(factory.IST_CFG)).numerics == "split"
    model = factory.build_model("dinov2_vits14", k=2, device="cpu")
    assert model.ae_net.dinov2_model.numerics == model.testing_metric.numerics == model.ist_net.backbone.numerics == "split"


def test_chain_is_opt_in_by_env_and_by_yaml_key(monkeypatch):
    from gigapose_amd import _lib, factory

    monkeypatch.setenv("GIGAPOSE_NUMERICS", "chain")
    assert _lib.default_numerics() == "chain"
    assert factory.build_model("dinov2_vits14", k=2, device="cpu").testing_metric.numerics == "chain"
    monkeypatch.delenv("GIGAPOSE_NUMERICS")
    m = factory.build_model("dinov2_vits14", k=2, device="cpu", numerics="chain")    # the `numerics:` key of the model YAML
    assert m.ae_net.dinov2_model.numerics == m.testing_metric.numerics == m.ist_net.backbone.numerics == "chain"
    monkeypatch.setenv("GIGAPOSE_NUMERICS", "fp8")
    with pytest.raises(ValueError):
        _lib.default_numerics()
    with pytest.raises(ValueError):
        m.set_numerics("bf16")


def test_second_stream_switch_is_off_by_default_and_read_from_the_environment(monkeypatch):
    """GigaPose.overlap_ist: False unless GIGAPOSE_OVERLAP_IST says 1 / auto (small batches)."""
    from gigapose_amd import factory

    monkeypatch.delenv("GIGAPOSE_OVERLAP_IST", raising=False)
    assert factory.build_model("dinov2_vits14", k=2, device="cpu").overlap_ist is False
    for text, want in (("1", True), ("auto", "auto"), ("AUTO", "auto"), ("0", False), ("", False), ("on", True)):
        monkeypatch.setenv("GIGAPOSE_OVERLAP_IST", text)
        assert factory.build_model("dinov2_vits14", k=2, device="cpu").overlap_ist == want
