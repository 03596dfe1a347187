"""The committed TINY model + inputs shared by tests/golden/export_with_reference_stack.py (which runs them through
the real flax/t5x stack on a machine that has it) and tests/test_external_fixtures.py (which runs them through this
repo's oracle / engine).  numpy only, bit-reproducible from the seeds below."""
import numpy as np

CFG = dict(vocab_size=128, emb_dim=128, num_heads=2, head_dim=64, mlp_dim=128, num_encoder_layers=2,
           num_decoder_layers=2)
INPUT_DEPTH, T, L, B, SEED = 64, 256, 24, 6, 11   # T = 256: the product engine can run it too
EOS_BOOST = 1.3          # flattens the logits and lifts EOS so that beam search finishes rows at different lengths


def param_shapes():
    e, hd, f, v = CFG["emb_dim"], CFG["num_heads"] * CFG["head_dim"], CFG["mlp_dim"], CFG["vocab_size"]
    s = {"encoder/continuous_inputs_projection/kernel": (INPUT_DEPTH, e), "encoder/encoder_norm/scale": (e,),
         "decoder/token_embedder/embedding": (v, e), "decoder/decoder_norm/scale": (e,),
         "decoder/logits_dense/kernel": (e, v)}
    for side, n, blocks in (("encoder", CFG["num_encoder_layers"], (("pre_attention_layer_norm", "attention"),)),
                            ("decoder", CFG["num_decoder_layers"],
                             (("pre_self_attention_layer_norm", "self_attention"),
                              ("pre_cross_attention_layer_norm", "encoder_decoder_attention")))):
        for i in range(n):
            p = "%s/layers_%d" % (side, i)
            for norm, att in blocks:
                s["%s/%s/scale" % (p, norm)] = (e,)
                for k in ("query", "key", "value"):
                    s["%s/%s/%s/kernel" % (p, att, k)] = (e, hd)
                s["%s/%s/out/kernel" % (p, att)] = (hd, e)
            s[p + "/pre_mlp_layer_norm/scale"] = (e,)
            s[p + "/mlp/wi_0/kernel"] = (e, f)
            s[p + "/mlp/wi_1/kernel"] = (e, f)
            s[p + "/mlp/wo/kernel"] = (f, e)
    return s


def params():
    rng = np.random.default_rng(SEED)
    out = {}
    for name, shape in sorted(param_shapes().items()):
        if name.endswith("/scale"):
            w = 1.0 + 0.2 * rng.standard_normal(shape)
        elif name.endswith("/embedding"):
            w = rng.standard_normal(shape)
        else:
            w = rng.standard_normal(shape) / np.sqrt(shape[0])
            if name.endswith("query/kernel"):
                w = w / np.sqrt(CFG["head_dim"])
        out[name] = w.astype(np.float32)
    k = out["decoder/logits_dense/kernel"] * 0.3
    k[:, 1] *= EOS_BOOST
    out["decoder/logits_dense/kernel"] = k.astype(np.float32)
    return out


def inputs():
    rng = np.random.default_rng(SEED + 1)
    x = rng.standard_normal((B, T, INPUT_DEPTH)).astype(np.float32)
    x[2, 100:] = 0.0                                            # a short segment: zero rows (F8)
    forced = rng.integers(3, CFG["vocab_size"], (B, L)).astype(np.int32)
    return x, forced


def nested(flat):
    """flat 'a/b/c' dict -> the nested Flax tree"""
    tree = {}
    for k, v in flat.items():
        d = tree
        parts = k.split("/")
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return tree
