"""TensorFlow tensor-bundle checkpoint I/O.  The golden data are the reference's own
checkpoint index files (test_v1/model/*.index, copied as binary fixtures into tests/golden/v1_index/): the
reader must verify their TensorFlow-written block checksums and the writer must reproduce every file byte for
byte from the parsed entries."""
import glob
import os

import numpy as np
import pytest

from dl_ofdm_amd import tf_bundle as T

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "v1_index", "*.index")))


def test_crc32c_known_answers():
    assert T.crc32c(b"123456789") == 0xE3069283                    # the CRC-32C check value
    assert T.crc32c(bytes(32)) == 0x8A9136AA                        # RFC 3720 B.4
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == 0xE3069283     # incremental
    for v in (0, 1, 0xDEADBEEF, 0xFFFFFFFF):
        assert T.unmask_crc(T.mask_crc(v)) == v


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_reference_index_files_parse_and_rebuild_byte_exact(path):
    raw = open(path, "rb").read()
    entries = T.read_index(path, verify=True)                       # TF-written block CRCs check out
    assert len(entries) == 33 and "global_step" in entries and entries["global_step"]["shape"] == []
    k = entries["fft_like/conv3d/kernel"]
    assert k["dtype"] == 1 and len(k["shape"]) == 5 and k["size"] == 4 * int(np.prod(k["shape"]))
    offs = sorted((e["offset"], e["size"]) for e in entries.values())
    assert offs[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(offs, offs[1:]))   # contiguous, key order
    assert T.index_bytes(entries) == raw


def test_manifest_matches_fixtures():
    import json
    man = json.load(open(os.path.join(HERE, "golden", "v1_index_manifest.json")))
    assert len(FIXTURES) == 8
    for path in FIXTURES:
        name = os.path.basename(path)[:-6]
        ent = T.read_index(path)
        assert {k: v["shape"] for k, v in ent.items()} == {k: v["shape"] for k, v in man[name].items()}


def test_write_read_roundtrip_and_corruption(tmp_path):
    rng = np.random.RandomState(0)
    tensors = {"fft_like/conv3d/kernel": rng.standard_normal((1, 5, 1, 5, 8)).astype(np.float32),
               "demodulation/dense/kernel": rng.standard_normal((40, 12)).astype(np.float32),
               "demodulation/dense/kernel/Adam": np.zeros((40, 12), np.float32),
               "global_step": np.float32(17.0), "beta1_power": np.float32(0.9 ** 17),
               "counts": np.arange(6, dtype=np.int64).reshape(2, 3), "z/empty_dim": np.zeros((0, 3), np.float32)}
    prefix = str(tmp_path / "model" / "OFDM_T")
    T.write_checkpoint(prefix, tensors)
    assert os.path.exists(prefix + ".index") and os.path.exists(T.data_path(prefix))
    assert 'model_checkpoint_path: "OFDM_T"' in open(str(tmp_path / "model" / "checkpoint")).read()
    back = T.read_checkpoint(prefix)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == np.asarray(v).dtype and back[k].shape == np.asarray(v).shape
        assert np.array_equal(back[k], v)
    ent = T.read_index(prefix + ".index")
    assert list(ent) == sorted(tensors, key=lambda s: s.encode())
    blob = bytearray(open(T.data_path(prefix), "rb").read())
    blob[ent["demodulation/dense/kernel"]["offset"] + 5] ^= 0x40
    open(T.data_path(prefix), "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="checksum"):
        T.read_checkpoint(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="checksum"):
        T.read_index(prefix + ".index")


def test_many_entries_span_restarts_and_blocks(tmp_path):
    """> 16 entries per block (restart points) and > 256 KiB of index (several data blocks + a real index block)"""
    names = ["scope_%03d/%s" % (i, "x" * 300) for i in range(1200)]
    tensors = {n: np.float32(i) for i, n in enumerate(names)}
    prefix = str(tmp_path / "big")
    T.write_checkpoint(prefix, tensors, state_file=False)
    assert os.path.getsize(prefix + ".index") > 2 * T.BLOCK_SIZE // 2
    back = T.read_checkpoint(prefix)
    assert len(back) == 1200 and all(float(back[n]) == i for i, n in enumerate(names))


def test_receiver_layout_mapping_is_invertible():
    rng = np.random.RandomState(1)
    p = {"fft_like/conv3d/kernel": rng.standard_normal((80, 128)).astype(np.float32),
         "fft_like/conv3d/kernel/Adam_1": rng.standard_normal((80, 128)).astype(np.float32),
         "demodulation/conv2d/kernel": rng.standard_normal((2, 4)).astype(np.float32),
         "demodulation/dense/bias": rng.standard_normal(640).astype(np.float32), "global_step": np.float32(3)}
    tf = T.rx_to_tf(p, 80)
    assert tf["fft_like/conv3d/kernel"].shape == (1, 80, 1, 80, 128) and tf["demodulation/conv2d/kernel"].shape == (1, 1, 2, 4)
    assert np.count_nonzero(tf["fft_like/conv3d/kernel"][0, :39]) == 0            # dead taps
    back = T.rx_from_tf(tf)
    assert all(np.array_equal(back[k], p[k]) for k in p)
    e = {"Equalizer/conv3d_1/kernel": rng.standard_normal((7, 64, 1, 2)).astype(np.float32),
         "optimizer/Equalizer/conv3d/kernel/Adam": rng.standard_normal((1, 64, 1, 128)).astype(np.float32),
         "Equalizer/dense_3/kernel": rng.standard_normal((8, 8)).astype(np.float32)}
    etf = T.eq_to_tf(e)
    assert etf["Equalizer/conv3d_1/kernel"].shape == (7, 64, 1, 1, 2)
    assert etf["optimizer/Equalizer/conv3d/kernel/Adam"].shape == (1, 64, 1, 1, 128)
    assert all(np.array_equal(T.eq_from_tf(etf)[k], e[k]) for k in e)


def test_v1_bundle_folds_its_second_1x1_conv_and_foreign_variables_raise(tmp_path):
    """The reference's only real checkpoints (test_v1/model/*, shapes pinned in v1_index_manifest.json) hold an extra
    demodulation/conv2d_1 layer right behind conv2d with no activation in between: rx_from_tf folds the pair into one
    affine map (same function), drops what cannot be folded (their Adam slots) and refuses unknown variables."""
    import json
    man = json.load(open(os.path.join(HERE, "golden", "v1_index_manifest.json")))
    name = sorted(k for k in man if "2mod" in k and "cpTrue" in k)[0]
    rng = np.random.RandomState(0)
    tensors = {}
    for var, ent in man[name].items():
        shp = tuple(ent["shape"])
        tensors[var] = (rng.randn(*shp) * 0.3).astype(np.float32) if shp else np.float32(0.5)
    assert "demodulation/conv2d_1/kernel" in tensors
    T.write_checkpoint(str(tmp_path / "v1"), tensors)
    got = T.rx_from_tf(T.read_checkpoint(str(tmp_path / "v1")))
    assert not any(k.startswith("demodulation/conv2d_1") for k in got)
    assert "demodulation/conv2d/kernel/Adam" not in got and "demodulation/dense/kernel/Adam" in got
    w1, b1 = tensors["demodulation/conv2d/kernel"][0, 0].astype(np.float64), tensors["demodulation/conv2d/bias"].astype(np.float64)
    w1b, b1b = tensors["demodulation/conv2d_1/kernel"][0, 0].astype(np.float64), tensors["demodulation/conv2d_1/bias"].astype(np.float64)
    z = rng.randn(50, 2)
    two = (z @ w1 + b1) @ w1b + b1b
    one = z @ got["demodulation/conv2d/kernel"].astype(np.float64) + got["demodulation/conv2d/bias"].astype(np.float64)
    assert np.abs(two - one).max() < 1e-6
    assert got["fft_like/conv3d/kernel"].shape == (80, 128) and got["demodulation/dense/kernel"].shape == (1024, 736)
    bad = dict(tensors)
    bad["demodulation/dense_7/kernel"] = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError):
        T.rx_from_tf(bad)
