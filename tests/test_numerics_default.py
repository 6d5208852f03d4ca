"""CPU: which numerics the drop-in selects (the bench must measure the mode a maintainer gets)."""
import pytest


def test_product_default_is_split(monkeypatch):
    monkeypatch.delenv("GIGAPOSE_NUMERICS", raising=False)
    from gigapose_amd import _lib
    from gigapose_testing import factory
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
    from gigapose_amd import _lib
    from gigapose_testing import factory

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


def test_second_stream_switch_defaults_to_auto_and_is_read_from_the_environment(monkeypatch):
    """GigaPose.overlap_ist: "auto" (IST backbone on a second stream up to 32 crops; round 5 -- the GPU suite runs with it) unless
    GIGAPOSE_OVERLAP_IST says 0 / 1 (small batches)."""
    from gigapose_testing import factory

    monkeypatch.delenv("GIGAPOSE_OVERLAP_IST", raising=False)
    assert factory.build_model("dinov2_vits14", k=2, device="cpu").overlap_ist == "auto"
    for text, want in (("1", True), ("auto", "auto"), ("AUTO", "auto"), ("0", False), ("", False), ("on", True)):
        monkeypatch.setenv("GIGAPOSE_OVERLAP_IST", text)
        assert factory.build_model("dinov2_vits14", k=2, device="cpu").overlap_ist == want


def test_cross_image_accumulation_default_and_keys(monkeypatch):
    """GigaPose.accumulate_crops: 64 by default; GIGAPOSE_ACCUMULATE_CROPS or the `accumulate_crops:` YAML key (read from **kwargs
    like `numerics`, so the constructor keeps the reference's signature) override it; 0 = the reference's per-image flow."""
    import tempfile

    from gigapose_testing import factory
    from gigapose_amd.gigaPose import GigaPose

    monkeypatch.delenv("GIGAPOSE_ACCUMULATE_CROPS", raising=False)
    m = factory.build_model("dinov2_vits14", k=2, device="cpu")
    assert m.accumulate_crops == 64 and m._pending == [] and m._in_flight is None
    monkeypatch.setenv("GIGAPOSE_ACCUMULATE_CROPS", "0")
    assert factory.build_model("dinov2_vits14", k=2, device="cpu").accumulate_crops == 0
    m2 = GigaPose("large", m.ae_net, m.ist_net, None, m.testing_metric, None, 1000, tempfile.mkdtemp(), accumulate_crops=32)
    assert m2.accumulate_crops == 32
