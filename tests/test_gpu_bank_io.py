"""GPU: a model started from a bank file predicts exactly what the model that onboarded the templates predicts;
template shards cut at load time equal the shards cut at onboarding (both numerics)."""
import pytest
import torch

from gigapose_amd import bank_io, factory

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
