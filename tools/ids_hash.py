"""sha256 of what the library in place decodes (round 6: `tools/ab_r6.py ids` runs this once per build of the library to
hold "same bits" against the builds without the priority / the kernel-argument pin): f32 and bf16 engines, B = 256 on the
product's row groups and on one stream, 96 greedy steps + a beam-1 decode with early exit."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import network, spectrograms, synthetic  # noqa: E402

h = hashlib.sha256()
lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(256, seed=1000), None)
for dtype in ("float32", "bfloat16"):
    cfg = network.T5Config(dtype=dtype)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256)
    eng.load_params(synthetic.boost_note_events(network.init_random_params(cfg, seed=0)))
    eng.encode(lm)
    for kw in (dict(num_steps=96), dict(num_steps=96, single_stream=True), dict(num_steps=160, beam1=True, early_exit=True)):
        ids = eng.decode(**kw)
        torch.cuda.synchronize()
        h.update(ids.cpu().numpy().tobytes())
    del eng
print("ids sha256", h.hexdigest(), flush=True)
