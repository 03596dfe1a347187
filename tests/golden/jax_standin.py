"""A numpy stand-in for the slice of jax / flax that mt3/layers.py and mt3/network.py use.

Purpose: run the reference's REAL network code (module wiring, norm placement, residuals, attention without
1/sqrt(d), fixed position embeddings, the decode-mode K/V cache with its one-hot update, logits head) in a
container that has no JAX, to produce golden vectors for the oracle (make_network_golden.py).  Only the leaf
numerics (einsum, dot_general, softmax, tanh-GELU, rsqrt) are numpy here; everything that decides WHAT is
computed is the reference's own Python.  This is test tooling for the build container, never shipped or
imported by mt3_amd/.

Covered: flax.linen.{Module, compact, Dropout, initializers.*, linear.default_kernel_init, gelu, relu, ...},
flax.linen.partitioning.{param_with_axes, with_sharding_constraint}, flax.struct.dataclass,
jax.{numpy, lax.*, nn.softmax/one_hot, random, vmap}; Module.apply with 'params' and a mutable 'cache'.
"""
import dataclasses
import sys
import types

import numpy as np

_stack = []          # modules whose method is executing (innermost last)
_ctx = None          # the running apply(): params, collections, mutable


class _Context:
    def __init__(self, params, collections, mutable):
        self.params, self.collections, self.mutable = params, collections, set(mutable or ())


class _Variable:
    def __init__(self, store, key):
        self._store, self._key = store, key

    @property
    def value(self):
        return self._store[self._key]

    @value.setter
    def value(self, v):
        self._store[self._key] = v


def _wrap(fn):
    def method(self, *args, **kwargs):
        self._ensure_setup()
        _stack.append(self)
        saved = self.__dict__.get("_autonames")
        object.__setattr__(self, "_autonames", {})
        try:
            return fn(self, *args, **kwargs)
        finally:
            object.__setattr__(self, "_autonames", saved if saved is not None else {})
            _stack.pop()
    method.__name__ = getattr(fn, "__name__", "method")
    method.__wrapped__ = fn
    return method


class Module:
    """flax.linen.Module: dataclass-style fields, children named by `name=` / attribute / ClassName_i."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        ann = dict(cls.__dict__.get("__annotations__", {}))
        ann.pop("name", None)
        ann.pop("parent", None)
        ann["parent"] = object
        ann["name"] = object
        cls.__annotations__ = ann
        cls.parent = None
        cls.name = None
        for k, v in list(cls.__dict__.items()):
            if k in ann:                                  # a field default that happens to be callable (initialisers)
                continue
            if isinstance(v, types.FunctionType) and (k == "__call__" or not k.startswith("_")) and k != "setup":
                setattr(cls, k, _wrap(v))
        dataclasses.dataclass(cls, eq=False, repr=False)

    def __post_init__(self):
        object.__setattr__(self, "_setup_done", False)
        object.__setattr__(self, "_autonames", {})
        if self.parent is None and _stack:
            object.__setattr__(self, "parent", _stack[-1])

    def __setattr__(self, key, value):
        if isinstance(value, Module) and key not in ("parent",) and value.name is None and self.__dict__.get("_in_setup"):
            object.__setattr__(value, "name", key)
            if value.parent is None:
                object.__setattr__(value, "parent", self)
        object.__setattr__(self, key, value)

    # ---- naming / scope
    def _resolved_name(self):
        if self.name is None:
            p = self.parent
            base = type(self).__name__
            i = p._autonames.get(base, 0) if p is not None else 0
            if p is not None:
                p._autonames[base] = i + 1
            object.__setattr__(self, "name", "%s_%d" % (base, i))
        return self.name

    def _path(self):
        if self.parent is None:
            return ()
        return self.parent._path() + (self._resolved_name(),)

    def _ensure_setup(self):
        if self.__dict__.get("_setup_done"):
            return
        object.__setattr__(self, "_setup_done", True)
        if self.parent is not None:
            self._resolved_name()                     # auto-names follow construction/first-use order
        setup = getattr(type(self), "setup", None)
        if setup is not None:
            _stack.append(self)
            object.__setattr__(self, "_in_setup", True)
            try:
                setup(self)
            finally:
                object.__setattr__(self, "_in_setup", False)
                _stack.pop()

    # ---- variables
    def param(self, name, init_fn, *init_args):
        key = "/".join(self._path() + (name,))
        if key not in _ctx.params:
            raise KeyError("parameter %s not provided" % key)
        value = np.asarray(_ctx.params[key])
        if init_args and tuple(int(s) for s in init_args[0]) != tuple(value.shape):
            raise ValueError("parameter %s: shape %s, module expects %s" % (key, value.shape, tuple(init_args[0])))
        return value

    def variable(self, col, name, init_fn=None, *init_args):
        store = _ctx.collections.setdefault(col, {})
        key = "/".join(self._path() + (name,))
        if key not in store:
            if col not in _ctx.mutable:
                raise KeyError("variable %s/%s missing and collection not mutable" % (col, key))
            store[key] = init_fn(*init_args)
        return _Variable(store, key)

    def has_variable(self, col, name):
        return "/".join(self._path() + (name,)) in _ctx.collections.get(col, {})

    def is_mutable_collection(self, col):
        return col in _ctx.mutable

    def make_rng(self, name):
        raise RuntimeError("stochastic paths are not supported by the stand-in (run deterministic)")

    def apply(self, variables, *args, method=None, mutable=False, rngs=None, **kwargs):
        global _ctx
        prev, prev_stack = _ctx, list(_stack)
        cols = {k: dict(v) for k, v in variables.items() if k != "params"}
        _ctx = _Context(variables["params"], cols, mutable if mutable else ())
        del _stack[:]
        try:
            fn = method if method is not None else type(self).__call__
            name = getattr(fn, "__name__", None)
            bound = getattr(type(self), name) if name and hasattr(type(self), name) else fn
            out = bound(self, *args, **kwargs)
        finally:
            result_cols = _ctx.collections
            _ctx = prev
            _stack[:] = prev_stack
        if mutable:
            return out, {k: v for k, v in result_cols.items() if k in set(mutable)}
        return out


def compact(fn):
    return fn


class Dropout(Module):
    rate: float = 0.0
    broadcast_dims: tuple = ()
    deterministic: object = None

    def __call__(self, inputs, deterministic=None):
        det = deterministic if deterministic is not None else self.deterministic
        if not det and self.rate > 0.0:
            raise RuntimeError("dropout must be deterministic in the stand-in")
        return inputs


def _initializer(*a, **k):
    def init(key, shape, dtype=np.float32):
        raise RuntimeError("initializers are not evaluated by the stand-in (parameters are supplied)")
    return init


def _ones(key, shape, dtype=np.float32):
    return np.ones(shape, dtype)


def gelu(x, approximate=True):
    x = np.asarray(x)
    if approximate:                                   # jax.nn.gelu default: tanh approximation
        c = np.sqrt(2.0 / np.pi).astype(x.dtype) if hasattr(np.sqrt(2.0 / np.pi), "astype") else np.sqrt(2.0 / np.pi)
        return (0.5 * x * (1.0 + np.tanh(c * (x + 0.044715 * (x ** 3))))).astype(x.dtype)
    from math import erf
    return (0.5 * x * (1.0 + np.vectorize(erf)(x / np.sqrt(2.0)))).astype(x.dtype)


def relu(x):
    return np.maximum(x, 0)


# ------------------------------------------------------------------ jax.lax / jax.nn
def dot_general(lhs, rhs, dimension_numbers, precision=None, preferred_element_type=None):
    (lc, rc), (lb, rb) = dimension_numbers
    if tuple(lb) or tuple(rb):
        raise NotImplementedError("batched dot_general is not used by the reference network")
    return np.tensordot(lhs, rhs, axes=(tuple(lc), tuple(rc)))


def dynamic_slice(operand, start_indices, slice_sizes):
    idx = []
    for s, n, dim in zip(np.asarray(start_indices).reshape(-1).tolist(), np.asarray(slice_sizes).reshape(-1).tolist(),
                         operand.shape):
        s = int(min(max(int(s), 0), dim - int(n)))      # XLA clamps the start so the slice fits
        idx.append(slice(s, s + int(n)))
    return operand[tuple(idx)]


def dynamic_slice_in_dim(operand, start_index, slice_size, axis=0):
    start = [0] * operand.ndim
    sizes = list(operand.shape)
    start[axis] = int(start_index)
    sizes[axis] = int(slice_size)
    return dynamic_slice(operand, start, sizes)


def softmax(x, axis=-1):
    x = np.asarray(x)
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def one_hot(x, num_classes, dtype=np.float32, axis=-1):
    return (np.asarray(x)[..., None] == np.arange(num_classes)).astype(dtype)


def _vmap(fn, in_axes=0, out_axes=0):
    def not_available(*a, **k):
        raise NotImplementedError("jax.vmap is only used for relative-position bias slicing, which MT3 has none of")
    return not_available


def install():
    """Register the stand-in modules in sys.modules (jax, jax.numpy, jax.lax, jax.nn, jax.random, flax, ...)."""
    jnp = types.ModuleType("jax.numpy")
    for k in dir(np):
        if not k.startswith("__"):
            setattr(jnp, k, getattr(np, k))
    jnp.ndarray = np.ndarray

    lax = types.ModuleType("jax.lax")
    lax.dot_general = dot_general
    lax.rsqrt = lambda x: 1.0 / np.sqrt(x)
    lax.square = np.square
    lax.select = lambda pred, a, b: np.where(pred, a, b)
    lax.iota = lambda dtype, n: np.arange(n, dtype=dtype)
    lax.dynamic_slice = dynamic_slice
    lax.dynamic_slice_in_dim = dynamic_slice_in_dim

    jnn = types.ModuleType("jax.nn")
    jnn.softmax, jnn.one_hot, jnn.gelu, jnn.relu = softmax, one_hot, gelu, relu

    rnd = types.ModuleType("jax.random")
    rnd.bernoulli = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no randomness in the stand-in"))

    jax = types.ModuleType("jax")
    jax.numpy, jax.lax, jax.nn, jax.random, jax.vmap = jnp, lax, jnn, rnd, _vmap
    jax.__path__ = []

    inits = types.SimpleNamespace(variance_scaling=_initializer, normal=_initializer, ones=_ones,
                                  zeros=lambda key, shape, dtype=np.float32: np.zeros(shape, dtype))
    linen = types.ModuleType("flax.linen")
    linen.Module, linen.compact, linen.Dropout, linen.initializers = Module, compact, Dropout, inits
    linen.linear = types.SimpleNamespace(default_kernel_init=_initializer())
    linen.gelu, linen.relu = gelu, relu
    linen.__path__ = []

    part = types.ModuleType("flax.linen.partitioning")

    def param_with_axes(name, init_fn, *init_args, axes=None, module=None):
        return (module or _stack[-1]).param(name, init_fn, *init_args)

    part.param_with_axes = param_with_axes
    part.with_sharding_constraint = lambda x, axes: x
    linen.partitioning = part

    struct = types.ModuleType("flax.struct")
    struct.dataclass = lambda cls: dataclasses.dataclass(cls, frozen=True)

    flax = types.ModuleType("flax")
    flax.linen, flax.struct = linen, struct
    flax.__path__ = []

    for name, mod in (("jax", jax), ("jax.numpy", jnp), ("jax.lax", lax), ("jax.nn", jnn), ("jax.random", rnd),
                      ("flax", flax), ("flax.linen", linen), ("flax.linen.partitioning", part), ("flax.struct", struct)):
        sys.modules[name] = mod
