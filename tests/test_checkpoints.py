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


def test_hand_typed_checkpoint_directory(tmp_path):
    """A checkpoint directory in which nothing was produced by this package or by the msgpack library: the index is
    typed out byte by byte from the msgpack specification (fixmap 0x80|n, fixstr 0xA0|n, fixarray 0x90|n, positive
    fixint, bin 8 = 0xC4 len, ext 8 = 0xC7 len type), the array metadata is literal zarr-v2 JSON text
    (https://zarr.readthedocs.io/en/stable/spec/v2.html: shape / chunks / dtype '<f4' / compressor / fill_value / order /
    filters / zarr_format), chunks are raw little-endian bytes (one gzip member written by Python's gzip module).
    Layout after t5x [from memory: PARITY UNPINNED against a real checkpoint, see tests/golden/external/README.md]:
    {'version', 'optimizer': {'target': tree, 'state': {...}}}; a leaf is an inline flax ndarray (ext type 1 holding
    msgpack [shape, dtype name, bytes]) or a TensorStore zarr spec pointing at `target.<dotted path>`."""
    import struct
    d = tmp_path / "hand"
    os.makedirs(d)

    def s(txt):                                  # fixstr
        b = txt.encode()
        assert len(b) < 32
        return bytes([0xA0 | len(b)]) + b

    def s8(txt):                                 # str 8
        b = txt.encode()
        return bytes([0xD9, len(b)]) + b

    scale = struct.pack("<4f", 1.0, 0.5, -2.0, 3.25)
    inline = bytes([0x93, 0x91, 0x04]) + s("float32") + bytes([0xC4, len(scale)]) + scale   # [[4], 'float32', bin]
    ext = bytes([0xC7, len(inline), 0x01]) + inline                                          # ext 8, type 1 = ndarray
    path = "target.decoder.logits_dense.kernel"
    spec = (bytes([0x83]) + s("driver") + s("zarr") +
            s("kvstore") + bytes([0x82]) + s("driver") + s("file") + s("path") + s8(path) +
            s("metadata") + bytes([0x82]) + s("shape") + bytes([0x92, 0x04, 0x06]) + s("dtype") + s("<f4"))
    index = (bytes([0x82]) + s("version") + bytes([0x03]) +
             s("optimizer") + bytes([0x82]) +
             s("target") + bytes([0x81]) + s("decoder") + bytes([0x82]) +
             s("decoder_norm") + bytes([0x81]) + s("scale") + ext +
             s("logits_dense") + bytes([0x81]) + s("kernel") + spec +
             s("state") + bytes([0x81]) + s("step") + bytes([0x05]))
    (d / "checkpoint").write_bytes(index)
    kernel = (np.arange(24, dtype=np.float32).reshape(4, 6) - 7.5) / 4.0
    os.makedirs(d / path)
    (d / path / ".zarray").write_text(
        '{\n  "chunks": [3, 4],\n  "compressor": {"id": "gzip", "level": 6},\n  "dtype": "<f4",\n'
        '  "fill_value": null,\n  "filters": null,\n  "order": "C",\n  "shape": [4, 6],\n  "zarr_format": 2\n}\n')
    for (i, j) in ((0, 0), (0, 1), (1, 0), (1, 1)):                 # edge chunks are stored full size (3 x 4)
        block = np.zeros((3, 4), "<f4")
        part = kernel[3 * i: 3 * i + 3, 4 * j: 4 * j + 4]
        block[: part.shape[0], : part.shape[1]] = part
        (d / path / f"{i}.{j}").write_bytes(gzip.compress(struct.pack("<12f", *block.reshape(-1).tolist())))
    # the msgpack library reads the hand-typed bytes as the tree they spell
    idx = CK.read_index(str(d))
    assert idx["version"] == 3 and idx["optimizer"]["state"]["step"] == 5
    assert idx["optimizer"]["target"]["decoder"]["logits_dense"]["kernel"]["kvstore"]["path"] == path
    got = CK.load_t5x_checkpoint(str(d))
    assert set(got) == {"decoder/decoder_norm/scale", "decoder/logits_dense/kernel"}
    assert got["decoder/decoder_norm/scale"].tolist() == [1.0, 0.5, -2.0, 3.25]
    assert np.array_equal(got["decoder/logits_dense/kernel"], kernel)
    # and what save_t5x_checkpoint writes for the same tree is byte-compatible in the parts the format fixes:
    # the same key set, the same .zarray fields
    CK.save_t5x_checkpoint(str(tmp_path / "ours"), got, step=5, inline_below=5)
    ours = CK.read_index(str(tmp_path / "ours"))
    assert set(ours) == set(idx) and set(ours["optimizer"]) == set(idx["optimizer"])
    assert np.array_equal(ours["optimizer"]["target"]["decoder"]["decoder_norm"]["scale"],
                          idx["optimizer"]["target"]["decoder"]["decoder_norm"]["scale"])
    meta_ours = json.loads((tmp_path / "ours" / path / ".zarray").read_text())
    meta_hand = json.loads((d / path / ".zarray").read_text())
    assert set(meta_ours) == set(meta_hand) and meta_ours["dtype"] == "<f4" and meta_ours["shape"] == [4, 6]


def test_train_state_shaped_checkpoint_bf16_sharded_with_optimizer_state(tmp_path):
    """The shape of a checkpoint t5x writes for a TRAINED model (layout from t5x's documented structure,
    unpinned against a real one): train state {'version', 'optimizer': {'target', 'state': {'step', 'param_states'}}},
    every partitioned parameter a TensorStore zarr spec whose chunks are the shards (smaller than the array, along axis 0),
    values stored as bfloat16, small leaves (norm scales) and ALL optimizer state inline; '/'-separated chunk keys as well.
    `load_t5x_checkpoint` returns the model parameters only, widened to float32 (restore dtype='float32', NB:255-256)."""
    cfg, params = _small_params(seed=9)
    rng = np.random.default_rng(1)
    adafactor = {"param_states": {"decoder": {"logits_dense": {"kernel": {
        "v_row": rng.random(64).astype(np.float32), "v_col": rng.random(40).astype(np.float32), "m": np.zeros(1, np.float32)}}}}}
    want = {k: (CK._bf16_to_f32(CK._f32_to_bf16_bits(v)).reshape(v.shape) if v.size >= 100 else v) for k, v in params.items()}
    for sep in (None, "/"):
        d = str(tmp_path / ("ck" + ("_slash" if sep else "")))
        CK.save_t5x_checkpoint(d, params, step=400000, inline_below=100, chunk_rows=24, store_dtype="bfloat16",
                               dimension_separator=sep, optimizer_state=adafactor)
        wi = os.path.join(d, "target.encoder.layers_0.mlp.wi_0.kernel")                       # [64, 96] in shards of 24 rows
        with open(os.path.join(wi, ".zarray")) as f:
            meta = json.load(f)
        assert meta["dtype"] == "bfloat16" and meta["chunks"] == [24, 96] and meta["shape"] == [64, 96]
        assert meta["compressor"]["id"] == "gzip" and meta.get("dimension_separator") == sep
        assert os.path.isfile(os.path.join(wi, "2/0" if sep else "2.0")) and not os.path.exists(os.path.join(wi, "3/0" if sep else "3.0"))
        idx = CK.read_index(d)
        assert idx["version"] == 3 and idx["optimizer"]["state"]["step"] == 400000
        assert idx["optimizer"]["state"]["param_states"]["decoder"]["logits_dense"]["kernel"]["v_row"].shape == (64,)
        got = CK.load_t5x_checkpoint(d, expected=network.param_shapes(cfg))
        assert set(got) == set(params)                                                      # no optimizer state leaks in
        for k in params:
            assert got[k].dtype == np.float32 and np.array_equal(got[k], want[k]), k
        # bf16 storage costs 2^-9 relative at most
        k = "decoder/logits_dense/kernel"
        assert 0 < np.abs(got[k] - params[k]).max() <= np.abs(params[k]).max() * 2.0 ** -8
