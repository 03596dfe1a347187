"""t5x checkpoint directories <-> the flat parameter dict the engine loads.

The reference restores its weights with t5x (`InferenceModel.restore_from_checkpoint`, notebook cell
"Imports and Definitions", `t5x.utils.RestoreCheckpointConfig(path, mode='specific', dtype='float32')`).
t5x itself is not part of the reference tree, so this is a restatement of its on-disk format
[from memory of t5x/checkpoints.py; no real checkpoint is available offline to pin it]:

    <dir>/checkpoint                        msgpack of the train-state dict {'version', 'optimizer':
                                            {'target': {...}, 'state': {...}}} (older: {'target': ...}).
                                            A leaf is either an inline array (flax.serialization
                                            ext type 1 = msgpack (shape, dtype-name, bytes)) or a
                                            TensorStore spec {'driver': 'zarr', 'kvstore': {'path': ...},
                                            'metadata': {...}} that points at a sibling directory.
    <dir>/target.<a>.<b>...<leaf>/          one zarr-v2 array per parameter: `.zarray` (JSON: shape,
                                            chunks, dtype, compressor {id: gzip|zlib|null}, order C,
                                            dimension_separator '.') and one file per chunk ('0.0', ...).

Only numpy + zlib + msgpack are needed.  `load_t5x_checkpoint` returns {'encoder/layers_0/attention/
query/kernel': f32 array, ...}, i.e. the Flax tree paths joined by '/', which is what
`network.Transformer.load_params` / `mt3_engine_load_weight` take.  `save_t5x_checkpoint` writes the
same layout (used by the tests and to hand weights back to a t5x-side tool).
"""
from __future__ import annotations

import json
import os
import zlib
from typing import Any, Dict, Iterable, Optional, Tuple

import numpy as np

_TARGET_PREFIX = "target."
_FLAX_EXT_NDARRAY, _FLAX_EXT_NPSCALAR = 1, 3


class CheckpointError(ValueError):
    pass


# ---------------------------------------------------------------------------------------- zarr v2
def _np_dtype(name: str) -> Tuple[np.dtype, bool]:
    """zarr dtype string -> (storage dtype, is_bfloat16)."""
    if name in ("bfloat16", "<V2", "|V2"):
        return np.dtype("<u2"), True
    try:
        return np.dtype(name), False
    except TypeError as e:
        raise CheckpointError("unsupported zarr dtype %r" % (name,)) from e


def _bf16_to_f32(u16: np.ndarray) -> np.ndarray:
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _decompress(raw: bytes, compressor: Optional[Dict[str, Any]]) -> bytes:
    if compressor is None:
        return raw
    cid = compressor.get("id")
    if cid == "gzip":
        return zlib.decompress(raw, 16 + zlib.MAX_WBITS)
    if cid == "zlib":
        return zlib.decompress(raw)
    raise CheckpointError("unsupported zarr compressor %r (t5x writes gzip)" % (cid,))


def read_zarr_array(path: str) -> np.ndarray:
    """One zarr-v2 array directory -> ndarray (bfloat16 is widened to float32)."""
    meta_path = os.path.join(path, ".zarray")
    if not os.path.isfile(meta_path):
        raise CheckpointError("%s: not a zarr array (no .zarray)" % path)
    with open(meta_path) as f:
        meta = json.load(f)
    if meta.get("zarr_format", 2) != 2:
        raise CheckpointError("%s: zarr_format %r not supported" % (path, meta.get("zarr_format")))
    if meta.get("filters"):
        raise CheckpointError("%s: zarr filters are not supported" % path)
    shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
    order = meta.get("order", "C")
    dtype, is_bf16 = _np_dtype(meta["dtype"])
    sep = meta.get("dimension_separator", ".")
    fill = meta.get("fill_value")
    out = np.empty(shape, dtype)
    if len(shape) == 0:                                   # scalar: a single chunk named '0'
        grid: Iterable[Tuple[int, ...]] = [()]
    else:
        grid = np.ndindex(*[-(-s // c) for s, c in zip(shape, chunks)])
    n_chunk = int(np.prod(chunks)) if chunks else 1
    for idx in grid:
        key = sep.join(str(i) for i in idx) if idx else "0"
        cpath = os.path.join(path, key)
        if os.path.isfile(cpath):
            with open(cpath, "rb") as f:
                buf = _decompress(f.read(), meta.get("compressor"))
            if len(buf) != n_chunk * dtype.itemsize:
                raise CheckpointError("%s: chunk %s holds %d bytes, expected %d"
                                      % (path, key, len(buf), n_chunk * dtype.itemsize))
            chunk = np.frombuffer(buf, dtype).reshape(chunks, order=order)
        elif fill is not None:
            chunk = np.full(chunks, fill, dtype)         # zarr: a missing chunk is all fill_value
        else:
            raise CheckpointError("%s: chunk %s is missing" % (path, key))
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]   # edge chunks are stored full-size
    return _bf16_to_f32(out) if is_bf16 else out


def _f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bit patterns (round to nearest even), as a '<u2' array"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype("<u2")


def write_zarr_array(path: str, arr: np.ndarray, chunks: Optional[Tuple[int, ...]] = None,
                     compressor: Optional[str] = "gzip", store_dtype: Optional[str] = None,
                     dimension_separator: Optional[str] = None) -> None:
    """store_dtype="bfloat16": the values are rounded to bfloat16 and stored as 2-byte patterns under the zarr dtype
    string "bfloat16" (what TensorStore writes for a bf16 parameter); dimension_separator "/" nests the chunk files."""
    arr = np.asarray(arr)
    dtype_str = arr.dtype.str
    if store_dtype == "bfloat16":
        arr, dtype_str = _f32_to_bf16_bits(arr).reshape(arr.shape), "bfloat16"
    elif store_dtype is not None:
        arr = arr.astype(store_dtype)
        dtype_str = arr.dtype.str
    chunks = tuple(chunks) if chunks is not None else tuple(max(1, s) for s in arr.shape)
    os.makedirs(path, exist_ok=True)
    meta = {"chunks": list(chunks), "compressor": {"id": compressor, "level": 1} if compressor else None,
            "dtype": dtype_str, "fill_value": None, "filters": None, "order": "C",
            "shape": list(arr.shape), "zarr_format": 2}
    if dimension_separator:
        meta["dimension_separator"] = dimension_separator
    with open(os.path.join(path, ".zarray"), "w") as f:
        json.dump(meta, f)
    grid = [()] if arr.ndim == 0 else np.ndindex(*[-(-s // c) for s, c in zip(arr.shape, chunks)])
    for idx in grid:
        block = np.zeros(chunks, arr.dtype)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
        block[tuple(slice(0, s.stop - s.start) for s in sel)] = arr[sel]
        raw = block.tobytes()
        if compressor == "gzip":
            co = zlib.compressobj(1, zlib.DEFLATED, 16 + zlib.MAX_WBITS)
            raw = co.compress(raw) + co.flush()
        elif compressor == "zlib":
            raw = zlib.compress(raw, 1)
        elif compressor is not None:
            raise CheckpointError("unsupported compressor %r" % (compressor,))
        cpath = os.path.join(path, (dimension_separator or ".").join(str(i) for i in idx) if idx else "0")
        os.makedirs(os.path.dirname(cpath), exist_ok=True)
        with open(cpath, "wb") as f:
            f.write(raw)


# ---------------------------------------------------------------------------------------- msgpack index
def _ext_hook(code: int, data: bytes):
    import msgpack
    if code == _FLAX_EXT_NDARRAY:
        shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
        dtype, is_bf16 = _np_dtype(dtype_name)
        a = np.frombuffer(buf, dtype).reshape(shape)
        return _bf16_to_f32(a) if is_bf16 else a
    if code == _FLAX_EXT_NPSCALAR:
        shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, np.dtype(dtype_name)).reshape(shape)[()]
    return msgpack.ExtType(code, data)


def read_index(ckpt_dir: str) -> Optional[Dict[str, Any]]:
    """The msgpack `checkpoint` file as nested dicts, or None if the directory has none."""
    path = os.path.join(ckpt_dir, "checkpoint")
    if not os.path.isfile(path):
        return None
    import msgpack
    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)


def _is_ts_spec(node: Any) -> bool:
    return isinstance(node, dict) and "kvstore" in node and "driver" in node


def _spec_path(spec: Dict[str, Any]) -> str:
    kv = spec["kvstore"]
    return kv["path"] if isinstance(kv, dict) else str(kv)


def _walk(node: Any, prefix: Tuple[str, ...]):
    if isinstance(node, dict) and not _is_ts_spec(node):
        for k, v in node.items():
            yield from _walk(v, prefix + (str(k),))
    else:
        yield prefix, node


def _target_tree(index: Dict[str, Any]) -> Optional[Dict[str, Any]]:
    if "optimizer" in index and isinstance(index["optimizer"], dict) and "target" in index["optimizer"]:
        return index["optimizer"]["target"]
    return index.get("target")


# ---------------------------------------------------------------------------------------- public API
def load_t5x_checkpoint(ckpt_dir: str, dtype=np.float32, expected: Optional[Iterable[str]] = None) -> Dict[str, np.ndarray]:
    """All model parameters (`target.*`) of a t5x checkpoint directory as {'a/b/c': array}.
    `expected`: the parameter names the caller wants ('a/b/c' form).  Array directories are named with '.' between
    tree levels, which is ambiguous for a level name that itself contains a dot; with `expected` a directory
    is mapped to the expected name whose dotted form it equals, and only otherwise by replacing every '.'.

    The msgpack index is authoritative when present (inline leaves, TensorStore specs resolved
    relative to the directory); array directories named `target.*` that the index does not mention
    are picked up as well, so a directory without an index still loads."""
    if not os.path.isdir(ckpt_dir):
        raise CheckpointError("%s: not a directory" % (ckpt_dir,))
    params: Dict[str, np.ndarray] = {}
    index = read_index(ckpt_dir)
    tree = _target_tree(index) if index is not None else None
    if tree is not None:
        for keys, leaf in _walk(tree, ()):
            name = "/".join(keys)
            if _is_ts_spec(leaf):
                sub = os.path.join(ckpt_dir, os.path.basename(_spec_path(leaf).rstrip("/")))
                params[name] = read_zarr_array(sub)
            elif isinstance(leaf, np.ndarray) or np.isscalar(leaf):
                params[name] = np.asarray(leaf)
            elif leaf is None:
                continue
            else:
                raise CheckpointError("%s: leaf %s has unsupported type %s" % (ckpt_dir, name, type(leaf).__name__))
    dotted = {n.replace("/", "."): n for n in expected} if expected is not None else {}
    for entry in sorted(os.listdir(ckpt_dir)):
        if entry.startswith(_TARGET_PREFIX) and os.path.isfile(os.path.join(ckpt_dir, entry, ".zarray")):
            tail = entry[len(_TARGET_PREFIX):]
            name = dotted.get(tail, tail.replace(".", "/"))
            if name not in params:
                params[name] = read_zarr_array(os.path.join(ckpt_dir, entry))
    if not params:
        raise CheckpointError("%s: no `target.*` parameters found (is this a t5x checkpoint directory?)" % ckpt_dir)
    return {k: np.ascontiguousarray(v, dtype=dtype) for k, v in params.items()}


def save_t5x_checkpoint(ckpt_dir: str, params: Dict[str, np.ndarray], step: int = 0,
                        inline_below: int = 0, chunk_rows: Optional[int] = None, store_dtype: Optional[str] = None,
                        dimension_separator: Optional[str] = None, optimizer_state: Optional[Dict[str, Any]] = None) -> None:
    """Write `params` ({'a/b/c': array}) in the layout above.  Arrays with fewer than `inline_below`
    elements go inline into the msgpack index (t5x does this for leaves without partitioning axes);
    `chunk_rows` splits the first axis into chunks of that many rows (t5x chunks per shard: a parameter sharded over
    the 'data' / 'model' mesh axes is one zarr array whose chunks are the shards); store_dtype="bfloat16" stores bf16
    (a checkpoint saved with `dtype='bfloat16'`; the reader widens it, as restoring with dtype='float32' does, NB:255-256);
    optimizer_state: extra entries of `optimizer.state` next to `step` (e.g. Adafactor's `param_states`: arrays inline) --
    everything outside `optimizer.target` is ignored by `load_t5x_checkpoint`."""
    import msgpack
    os.makedirs(ckpt_dir, exist_ok=True)
    tree: Dict[str, Any] = {}
    for name, arr in params.items():
        arr = np.asarray(arr)
        node = tree
        keys = name.split("/")
        for k in keys[:-1]:
            node = node.setdefault(k, {})
        if arr.size < inline_below:
            node[keys[-1]] = msgpack.ExtType(
                _FLAX_EXT_NDARRAY, msgpack.packb((list(arr.shape), arr.dtype.name, arr.tobytes()), use_bin_type=True))
            continue
        sub = _TARGET_PREFIX + ".".join(keys)
        chunks = None
        if chunk_rows and arr.ndim >= 1:
            chunks = (min(chunk_rows, max(1, arr.shape[0])),) + tuple(max(1, s) for s in arr.shape[1:])
        write_zarr_array(os.path.join(ckpt_dir, sub), arr, chunks=chunks, store_dtype=store_dtype,
                         dimension_separator=dimension_separator)
        node[keys[-1]] = {"driver": "zarr", "kvstore": {"driver": "file", "path": sub},
                          "metadata": {"shape": list(arr.shape), "dtype": store_dtype or arr.dtype.str,
                                       "chunks": list(chunks or arr.shape), "compressor": {"id": "gzip"}}}

    def pack_state(node):
        if isinstance(node, dict):
            return {k: pack_state(v) for k, v in node.items()}
        if isinstance(node, np.ndarray):
            return msgpack.ExtType(_FLAX_EXT_NDARRAY,
                                   msgpack.packb((list(node.shape), node.dtype.name, node.tobytes()), use_bin_type=True))
        return node
    state = {"step": int(step)}
    state.update(pack_state(optimizer_state or {}))
    index = {"version": 3, "optimizer": {"target": tree, "state": state}}
    with open(os.path.join(ckpt_dir, "checkpoint"), "wb") as f:
        f.write(msgpack.packb(index, use_bin_type=True))


def is_t5x_checkpoint_dir(path: str) -> bool:
    if not os.path.isdir(path):
        return False
    if os.path.isfile(os.path.join(path, "checkpoint")):
        return True
    return any(e.startswith(_TARGET_PREFIX) for e in os.listdir(path))


# ---------------------------------------------------------------------------------------- compact .npz (repo fixtures)
def save_compact_npz(path: str, params: Dict[str, np.ndarray], meta: Optional[Dict[str, Any]] = None) -> Dict[str, np.ndarray]:
    """A checkpoint small enough to live in the repository (tests/golden/): every matrix as int8 with one f32 scale per
    OUTPUT column (embedding: per row), vectors (norm scales) as f32, zlib-compressed -- 1 byte per weight.  The weights of
    the checkpoint ARE the de-quantised values (`load_compact_npz` returns them as f32; every engine precision starts
    from those same f32 numbers).  Not a t5x format: `save_t5x_checkpoint(load_compact_npz(p))` writes that.
    Returns the de-quantised dict."""
    out, deq = {}, {}
    for name, w in params.items():
        w = np.asarray(w, np.float32)
        if w.ndim != 2:
            out[name + "|f"] = w
            deq[name] = w
            continue
        axis = 1 if name.endswith("/embedding") else 0            # reduce over `axis`: one scale per column (per row)
        s = np.abs(w).max(axis=axis, keepdims=True) / 127.0
        s = np.where(s > 0, s, 1.0).astype(np.float32)
        q = np.clip(np.rint(w / s), -127, 127).astype(np.int8)
        out[name + "|q"], out[name + "|s"] = q, s
        deq[name] = q.astype(np.float32) * s
    if meta:
        out["__meta__"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(path, **out)
    return deq


def load_compact_npz(path: str) -> Dict[str, np.ndarray]:
    """{'a/b/c': f32 array} of a file written by `save_compact_npz`."""
    params: Dict[str, np.ndarray] = {}
    with np.load(path) as z:
        for key in z.files:
            if key.endswith("|f"):
                params[key[:-2]] = np.asarray(z[key], np.float32)
            elif key.endswith("|q"):
                params[key[:-2]] = z[key].astype(np.float32) * z[key[:-2] + "|s"]
    if not params:
        raise CheckpointError("%s: not a compact checkpoint (no '|q' / '|f' entries)" % path)
    return params


def compact_npz_meta(path: str) -> Dict[str, Any]:
    with np.load(path) as z:
        return json.loads(bytes(z["__meta__"]).decode()) if "__meta__" in z.files else {}
