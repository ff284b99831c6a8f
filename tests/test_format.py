"""CPU tests of the ggml model-file reader/writer (the loader half of the boundary)."""
import os
import struct

import numpy as np
import pytest

from whisper_amd import ggml_format as gf


def test_roundtrip_and_header(tmp_path, tiny_model):
    p = str(tmp_path / "m.bin")
    n = gf.write_model(p, tiny_model)
    assert n == os.path.getsize(p)
    raw = open(p, "rb").read(4 + 44)
    assert struct.unpack("<I", raw[:4])[0] == 0x67676D6C
    assert list(struct.unpack("<11i", raw[4:])) == tiny_model.hparams.as_list()
    back = gf.read_model(p)
    assert back.hparams == tiny_model.hparams
    assert len(back.tensors) == 11 + 15 * 4 + 24 * 4
    for k, v in tiny_model.tensors.items():
        assert back.tensors[k].dtype == v.dtype and np.array_equal(back.tensors[k], v)
    assert np.array_equal(back.filters, tiny_model.filters)
    assert back.vocab == tiny_model.vocab


def test_tensor_inventory_counts():
    for kind, (v, d, h, l) in gf.MODEL_SHAPES.items():
        hp = gf.hparams_for(kind)
        assert len(gf.tensor_specs(hp)) == 11 + 15 * l + 24 * l
        assert d == 64 * h


def test_real_shape_file_sizes():
    """Byte count of a medium-shaped file equals the real ggml-medium.bin (1 533 796 691 B would need the real vocab
    strings; with the stand-in vocabulary only the vocab section differs)."""
    hp = gf.hparams_for("medium")
    payload = sum(int(np.prod(s)) * (2 if f16 else 4) for _, s, f16 in gf.tensor_specs(hp))
    assert 1.52e9 < payload < 1.54e9


def test_mel_filterbank_properties():
    f = gf.mel_filterbank()
    assert f.shape == (80, 201) and f.dtype == np.float32
    assert (f >= 0).all() and (f.sum(axis=1) > 0).all()
    # triangles: each row has a single contiguous support
    for row in f:
        nz = np.nonzero(row)[0]
        assert nz[-1] - nz[0] + 1 == len(nz)


def test_special_tokens():
    en = gf.special_tokens(gf.hparams_for("tiny.en"))
    ml = gf.special_tokens(gf.hparams_for("tiny"))
    assert en["eot"] == 50256 and en["beg"] == 50363 and ml["eot"] == 50257 and ml["beg"] == 50364
    assert en["translate"] == ml["translate"] == 50358


def test_bad_magic(tmp_path):
    p = tmp_path / "bad.bin"
    p.write_bytes(b"\0" * 64)
    with pytest.raises(ValueError):
        gf.read_model(str(p))
