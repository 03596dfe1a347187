"""t5x checkpoint directory reader/writer, host only.

The format is restated from memory of t5x (no real checkpoint is available offline), so these
tests pin self-consistency, the zarr-v2 rules (edge chunks stored full size, gzip members, missing
chunk = fill value, C order, '.'-separated chunk keys) against hand-built directories, and the
error behaviour."""
import gzip
import json
import os

import numpy as np
import pytest

from mt3_amd import checkpoints as CK
from mt3_amd import network


def _small_params(seed=0):
    cfg = network.T5Config(emb_dim=64, num_heads=2, head_dim=16, mlp_dim=96, num_encoder_layers=1,
                           num_decoder_layers=1, vocab_size=40, input_depth=24)
    return cfg, network.init_random_params(cfg, seed=seed)


def test_round_trip_all_parameters(tmp_path):
    cfg, params = _small_params()
    CK.save_t5x_checkpoint(str(tmp_path / "ck"), params, step=7)
    assert CK.is_t5x_checkpoint_dir(str(tmp_path / "ck"))
    got = CK.load_t5x_checkpoint(str(tmp_path / "ck"))
    assert set(got) == set(params) == set(network.param_shapes(cfg))
    for k in params:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], params[k]), k
    # names on disk follow t5x: target.<path with dots>
    assert os.path.isdir(tmp_path / "ck" / "target.encoder.layers_0.attention.query.kernel")
    idx = CK.read_index(str(tmp_path / "ck"))
    assert idx["optimizer"]["state"]["step"] == 7
    spec = idx["optimizer"]["target"]["decoder"]["logits_dense"]["kernel"]
    assert spec["driver"] == "zarr" and spec["kvstore"]["path"] == "target.decoder.logits_dense.kernel"


def test_chunked_arrays_and_inline_leaves(tmp_path):
    _, params = _small_params(seed=3)
    CK.save_t5x_checkpoint(str(tmp_path / "ck"), params, inline_below=100, chunk_rows=24)   # 64 rows -> 24+24+16
    d = tmp_path / "ck" / "target.encoder.layers_0.mlp.wi_0.kernel"
    assert sorted(f for f in os.listdir(d) if not f.startswith(".")) == ["0.0", "1.0", "2.0"]
    assert not os.path.exists(tmp_path / "ck" / "target.encoder.encoder_norm.scale")        # 64 elements: inline
    got = CK.load_t5x_checkpoint(str(tmp_path / "ck"))
    for k in params:
        assert np.array_equal(got[k], params[k]), k


def test_directory_without_index_and_legacy_index(tmp_path):
    _, params = _small_params(seed=4)
    CK.save_t5x_checkpoint(str(tmp_path / "ck"), params)
    os.remove(tmp_path / "ck" / "checkpoint")
    got = CK.load_t5x_checkpoint(str(tmp_path / "ck"))
    assert all(np.array_equal(got[k], params[k]) for k in params)
    # older layout: {'target': ...} at the top level
    import msgpack
    CK.save_t5x_checkpoint(str(tmp_path / "ck2"), params)
    idx = CK.read_index(str(tmp_path / "ck2"))
    with open(tmp_path / "ck2" / "checkpoint", "wb") as f:
        f.write(msgpack.packb({"target": idx["optimizer"]["target"]}, use_bin_type=True))
    got = CK.load_t5x_checkpoint(str(tmp_path / "ck2"))
    assert all(np.array_equal(got[k], params[k]) for k in params)


def _hand_zarr(path, arr, chunks, compressor, dtype_str=None, drop=None, sep=None):
    os.makedirs(path)
    meta = {"chunks": list(chunks), "compressor": compressor, "dtype": dtype_str or arr.dtype.str, "fill_value": 0.0,
            "filters": None, "order": "C", "shape": list(arr.shape), "zarr_format": 2}
    if sep:
        meta["dimension_separator"] = sep
    with open(os.path.join(path, ".zarray"), "w") as f:
        json.dump(meta, f)
    grid = [-(-s // c) for s, c in zip(arr.shape, chunks)]
    for idx in np.ndindex(*grid):
        if drop == idx:
            continue
        block = np.zeros(chunks, arr.dtype)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
        block[tuple(slice(0, s.stop - s.start) for s in sel)] = arr[sel]
        raw = block.tobytes()
        if compressor and compressor["id"] == "gzip":
            raw = gzip.compress(raw)                        # an independent gzip writer
        sub = os.path.join(path, (sep or ".").join(map(str, idx)))
        os.makedirs(os.path.dirname(sub), exist_ok=True)
        with open(sub, "wb") as f:
            f.write(raw)


def test_zarr_rules_against_hand_built_arrays(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((10, 7)).astype(np.float32)
    _hand_zarr(str(tmp_path / "g"), a, (4, 3), {"id": "gzip", "level": 5})         # ragged in both axes
    assert np.array_equal(CK.read_zarr_array(str(tmp_path / "g")), a)
    _hand_zarr(str(tmp_path / "raw"), a, (10, 7), None)
    assert np.array_equal(CK.read_zarr_array(str(tmp_path / "raw")), a)
    _hand_zarr(str(tmp_path / "slash"), a, (5, 7), None, sep="/")
    assert np.array_equal(CK.read_zarr_array(str(tmp_path / "slash")), a)
    # a missing chunk reads as fill_value
    _hand_zarr(str(tmp_path / "hole"), a, (5, 7), None, drop=(1, 0))
    want = a.copy()
    want[5:] = 0
    assert np.array_equal(CK.read_zarr_array(str(tmp_path / "hole")), want)
    # float64 and float16 widen/narrow through `dtype=`
    _hand_zarr(str(tmp_path / "target.x.y"), a.astype(np.float64), (10, 7), None)
    got = CK.load_t5x_checkpoint(str(tmp_path))
    assert list(got) == ["x/y"] and got["x/y"].dtype == np.float32 and np.array_equal(got["x/y"], a)
    # bfloat16 storage: upper 16 bits of the float32 pattern
    bits = (a.view(np.uint32) >> 16).astype("<u2")
    _hand_zarr(str(tmp_path / "bf"), bits, (10, 7), None, dtype_str="bfloat16")
    want = (bits.astype(np.uint32) << 16).view(np.float32)
    assert np.array_equal(CK.read_zarr_array(str(tmp_path / "bf")), want)
    # scalar
    CK.write_zarr_array(str(tmp_path / "s"), np.float32(3.5))
    assert CK.read_zarr_array(str(tmp_path / "s")) == np.float32(3.5)


def test_errors(tmp_path):
    with pytest.raises(CK.CheckpointError, match="not a directory"):
        CK.load_t5x_checkpoint(str(tmp_path / "nope"))
    os.makedirs(tmp_path / "empty")
    with pytest.raises(CK.CheckpointError, match="no `target"):
        CK.load_t5x_checkpoint(str(tmp_path / "empty"))
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    _hand_zarr(str(tmp_path / "z"), a, (3, 4), {"id": "blosc"})
    with pytest.raises(CK.CheckpointError, match="compressor"):
        CK.read_zarr_array(str(tmp_path / "z"))
    _hand_zarr(str(tmp_path / "short"), a, (3, 4), None)
    with open(tmp_path / "short" / "0.0", "wb") as f:
        f.write(b"\0" * 40)
    with pytest.raises(CK.CheckpointError, match="expected 48"):
        CK.read_zarr_array(str(tmp_path / "short"))
    CK.write_zarr_array(str(tmp_path / "gone"), a)
    os.remove(tmp_path / "gone" / "0.0")
    with pytest.raises(CK.CheckpointError, match="missing"):
        CK.read_zarr_array(str(tmp_path / "gone"))
    with pytest.raises(CK.CheckpointError, match="no .zarray"):
        CK.read_zarr_array(str(tmp_path / "empty"))


def test_directory_names_with_dots_resolve_against_the_expected_tree(tmp_path):
    """Array directories join tree levels with '.', which is ambiguous for a level whose own name has a dot; with
    the caller's expected names the reader maps `target.enc.v1.5.kernel` to 'enc/v1.5/kernel', not 'enc/v1/5/kernel'."""
    from mt3_amd import checkpoints
    d = tmp_path / "ck"
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    checkpoints.write_zarr_array(str(d / "target.enc.v1.5.kernel"), a)
    checkpoints.write_zarr_array(str(d / "target.enc.plain.kernel"), a + 1)
    naive = checkpoints.load_t5x_checkpoint(str(d))
    assert "enc/v1/5/kernel" in naive
    got = checkpoints.load_t5x_checkpoint(str(d), expected=["enc/v1.5/kernel", "enc/plain/kernel"])
    assert set(got) == {"enc/v1.5/kernel", "enc/plain/kernel"}
    assert np.array_equal(got["enc/v1.5/kernel"], a) and np.array_equal(got["enc/plain/kernel"], a + 1)
