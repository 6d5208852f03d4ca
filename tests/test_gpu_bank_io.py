"""GPU: a model started from a bank file predicts exactly what the model that onboarded the templates predicts;
template shards cut at load time equal the shards cut at onboarding (both numerics)."""
import pytest
import torch

from gigapose_amd import bank_io

from gigapose_testing import factory

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("numerics", ["chain", "split"])
def test_bank_file_round_trip_and_shards(tmp_path, numerics):
    dev = torch.device("cuda", 0)
    tset = factory.TemplateSet(2, 9, seed=70)
    q = tset.crops(71, 4, dev)
    a = factory.build_model("dinov2_vits14", k=4, device=dev, seed=5).set_numerics(numerics)
    a.template_datasets = {"syn": tset}
    a.set_template_data("syn")
    path = str(tmp_path / "syn.gpbank")
    bank_io.save_bank(a, "syn", path)
    ref = a.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")

    b = factory.build_model("dinov2_vits14", k=4, device=dev, seed=5).set_numerics(numerics)
    hdr = bank_io.load_bank(b, "syn", path)                     # no template images, no onboarding
    assert hdr["numerics"] == numerics and (hdr["O"], hdr["N"]) == (2, 9)
    got = b.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
    for n, v in ref.tensors.items():
        assert torch.equal(v, got.tensors[n]), n

    full = a.match_banks["syn"]
    c = factory.build_model("dinov2_vits14", k=4, device=dev, seed=5).set_numerics(numerics)
    for rank in range(2):                                        # shard slices without a process group
        h = bank_io.read_header(path)
        lo, hi = __import__("gigapose_amd.sharding", fromlist=["shard_bounds"]).shard_bounds(9, 2, rank)
        part = torch.from_numpy(bank_io.map_section(path, h, "match_hi" if numerics == "split" else "match_f32")[:, lo:hi].copy())
        want = (full.hi if numerics == "split" else full.features)[:, lo:hi].cpu()
        assert torch.equal(part, want)
    with pytest.raises(ValueError):                               # numerics mismatch is refused
        other = factory.build_model("dinov2_vits14", k=4, device=dev, seed=5).set_numerics("chain" if numerics == "split" else "split")
        bank_io.load_bank(other, "syn", path)


def test_fp16_bank_file_round_trip_and_8_way_shards(tmp_path):
    """BASELINE config 5's bank format: the fp16 (hi-plane-only) matcher bank, 162 templates per object, cut 8-way at load time
    (two ranks with 21, six with 20 templates -- sharding.shard_bounds): the file holds no lo plane (half the matcher bytes), a model
    started from it predicts bit for bit what the onboarding model predicts, and every rank's slice is the onboarded slice."""
    from gigapose_amd.sharding import ShardedMatcher, shard_bounds

    dev = torch.device("cuda", 0)
    tset = factory.TemplateSet(2, 162, seed=72)
    q = tset.crops(73, 4, dev)

    def build():
        m = factory.build_model("dinov2_vits14", k=5, device=dev, seed=5).set_numerics("split")
        m.testing_metric.bank_dtype = "f16"
        return m

    a = build()
    a.template_datasets = {"syn": tset}
    a.set_template_data("syn")
    full = a.match_banks["syn"]
    assert full.bank_dtype == "f16" and full.lo is None
    path = str(tmp_path / "syn16.gpbank")
    hdr = bank_io.save_bank(a, "syn", path)
    assert hdr["bank_dtype"] == "f16" and "match_lo" not in hdr["sections"] and hdr["sections"]["match_hi"]["dtype"] == "float16"
    assert hdr["sections"]["match_hi"]["shape"] == [2, 162, 256, 384]
    ref = a.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
    b = build()
    assert bank_io.load_bank(b, "syn", path)["bank_dtype"] == "f16"
    got = b.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
    for n, v in ref.tensors.items():
        assert torch.equal(v, got.tensors[n]), n
    sizes = []
    for rank in range(8):
        c = build()
        bank_io.load_bank(c, "syn", path, shard=(rank, 8))
        sm = c.match_banks["syn"]
        lo, hi = shard_bounds(162, 8, rank)
        assert isinstance(sm, ShardedMatcher) and sm.lo == lo and sm.bank.N == hi - lo and sm.bank.lo is None and sm.bank.bank_dtype == "f16"
        assert torch.equal(sm.bank.hi, full.hi[:, lo:hi]) and torch.equal(sm.bank.masks, full.masks[:, lo:hi])
        sizes.append(hi - lo)
    assert sum(sizes) == 162 and sorted(sizes, reverse=True) == [21, 21, 20, 20, 20, 20, 20, 20]


def test_bank_file_carries_the_plane_scale_calibration(tmp_path):
    """A model started from a bank file has never seen the templates: the ViT's plane-scale calibration (round 5) travels in the file.
    Weights with planted outliers (synthetic.plant_dinov2_outliers, ViT-L width, 2 blocks): the onboarding model calibrates, saves;
    the loading model adopts the calibration, predicts WITHOUT tripping the range guard (no warning), and equals the onboarding
    model's predictions bit for bit."""
    import warnings

    from gigapose_amd import _lib
    from gigapose_testing import synthetic as syn
    from gigapose_amd.vit import Dinov2ViT
    from test_gpu_guards import _gigapose_with_vit

    dev = torch.device("cuda", 0)

    def make():
        return _gigapose_with_vit(syn.plant_dinov2_outliers(syn.fill_state_dict(Dinov2ViT(1024, 2, 16), 11).eval()).to(dev))

    tset = factory.TemplateSet(1, 64, seed=80)
    q = tset.crops(81, 64, dev)
    a = make()
    a.template_datasets = {"syn": tset}
    a.set_template_data("syn")
    report = a.ae_net.dinov2_model.plane_scale_report()
    assert report, "the planted outliers must have lowered some plane scales"
    path = str(tmp_path / "outliers.gpbank")
    hdr = bank_io.save_bank(a, "syn", path)
    assert "vit_plane_amax" in hdr["sections"]
    ref = a.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
    torch.cuda.synchronize()
    _lib.check_status()
    b = make()
    assert b.ae_net.dinov2_model.plane_scales is None
    bank_io.load_bank(b, "syn", path)
    assert b.ae_net.dinov2_model.plane_scale_report() == report
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = b.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
        torch.cuda.synchronize()
        _lib.check_status()                                       # clean: no range trip on the first query
    for n, v in ref.tensors.items():
        assert torch.equal(v, got.tensors[n]), n
