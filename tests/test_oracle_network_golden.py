"""The network oracle against golden vectors produced by the reference's REAL mt3/network.py + mt3/layers.py
(run unmodified in the build container on a numpy stand-in for jax/flax: tests/golden/make_network_golden.py).
Pins the wiring of the whole encoder-decoder -- encoder output, teacher-forced logits, and the cached one-token
decode path as t5x drives it -- on a small T5 (2+2 layers, emb 32, 2 heads x 8, gated-GELU 64, vocab 48)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from mt3_amd import network as product_network  # noqa: E402  (parameter names / initialisers only; no GPU code runs)
from oracle import network as ON  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "network_golden.npz")


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    V, emb, H, hd, mlp, ne, nd, depth, seed = (int(v) for v in z["config"])
    pc = product_network.T5Config(dtype="float32", vocab_size=V, emb_dim=emb, num_heads=H, head_dim=hd, mlp_dim=mlp,
                                  num_encoder_layers=ne, num_decoder_layers=nd, input_depth=depth)
    params = product_network.init_random_params(pc, seed=seed, norm_scale_jitter=0.3)
    assert abs(sum(float(np.abs(np.asarray(v, np.float32)).sum()) for v in params.values()) -
               float(z["param_checksum"])) < 1e-3 * float(z["param_checksum"]), "parameter generator changed"
    oc = ON.T5Config(vocab_size=V, emb_dim=emb, num_heads=H, num_encoder_layers=ne, num_decoder_layers=nd, head_dim=hd,
                     mlp_dim=mlp, input_depth=depth)
    return z, ON.Oracle(params, oc)


def test_encoder_output_matches_the_reference_network(gold):
    z, orc = gold
    enc = orc.encode(z["x"]).numpy()
    np.testing.assert_allclose(enc, z["encoded"], rtol=2e-5, atol=2e-5)


def test_teacher_forced_logits_match_the_reference_network(gold):
    z, orc = gold
    logits = orc.decode_logits(torch.from_numpy(z["encoded"]), z["dec_in"]).numpy()
    np.testing.assert_allclose(logits, z["logits_teacher_forced"], rtol=2e-5, atol=5e-5)


def test_cached_single_step_decode_matches_the_reference_cache_path(gold):
    """layers.py:246-314 (cache init pass, one-hot K/V update, cache_index mask) + FixedEmbed decode indexing
    (layers.py:589-596), driven token by token like t5x's `tokens_ids_to_logits`."""
    z, orc = gold
    enc = torch.from_numpy(z["encoded"])
    cache = orc._init_cache(enc)
    dec_in = z["dec_in"]
    steps = []
    for t in range(dec_in.shape[1]):
        steps.append(orc._step(cache, torch.from_numpy(dec_in[:, t].astype(np.int64)), t).numpy())
    got = np.stack(steps, 1)
    np.testing.assert_allclose(got, z["logits_cached"], rtol=2e-5, atol=5e-5)
    # and the reference's two decode paths agree with each other (a property of the fixture itself)
    np.testing.assert_allclose(z["logits_cached"], z["logits_teacher_forced"], atol=5e-6)
