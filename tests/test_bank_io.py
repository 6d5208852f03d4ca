"""CPU: the bank file container (gigapose_amd/bank_io.py): header / section round trip, alignment, shard slices."""
import numpy as np
import pytest

from gigapose_amd import bank_io
from gigapose_amd.sharding import shard_bounds


def test_sections_round_trip_and_are_page_aligned(tmp_path):
    rs = np.random.RandomState(0)
    arrays = {"match_hi": rs.standard_normal((3, 7, 256, 64)).astype(np.float16), "masks": rs.rand(3, 7, 256).astype(np.float32),
              "poses": rs.standard_normal((3, 7, 4, 4)).astype(np.float32), "tiny": np.arange(5, dtype=np.int64)}
    path = str(tmp_path / "b.gpbank")
    bank_io.write_sections(path, dict(numerics="split", O=3, N=7, C=64), arrays)
    h = bank_io.read_header(path)
    assert h["numerics"] == "split" and (h["O"], h["N"], h["C"]) == (3, 7, 64) and h["_base"] % bank_io.ALIGN == 0
    for name, a in arrays.items():
        assert h["sections"][name]["offset"] % bank_io.ALIGN == 0
        np.testing.assert_array_equal(np.asarray(bank_io.map_section(path, h, name)), a)
    lo, hi = shard_bounds(7, 2, 1)                       # a rank's template slice = O contiguous row ranges
    np.testing.assert_array_equal(np.asarray(bank_io.map_section(path, h, "match_hi")[:, lo:hi]), arrays["match_hi"][:, lo:hi])


def test_rejects_foreign_files(tmp_path):
    p = tmp_path / "x.bin"
    p.write_bytes(b"not a bank")
    with pytest.raises(ValueError):
        bank_io.read_header(str(p))
