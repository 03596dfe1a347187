"""Encoder probe (GPU): run-to-run determinism of encode(), split-vs-single decode residual stream agreement, and
HIP-event timing of encode at B=256 (for rocprofv3 --kernel-trace --stats runs of the encoder alone).
PROBE_DENSE=fp8_e4m3 runs the MXFP8 encoder."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

B = int(os.environ.get("PROBE_B", "256"))
import dataclasses  # noqa: E402
cfg = dataclasses.replace(network.T5Config(dtype="bfloat16"), dense_dtype=os.environ.get("PROBE_DENSE", ""))   # "fp8_e4m3": MXFP8 encoder
params = network.init_random_params(cfg, seed=0, norm_scale_jitter=0.2)
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B)
eng.load_params(params)
lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=3), None)
outs = [eng.encode(lm, return_encoded=True).clone() for _ in range(4)]
torch.cuda.synchronize()
print("encode run-to-run bit-identical:", [bool(torch.equal(outs[0], o)) for o in outs[1:]],
      "max abs diff", [float((outs[0] - o).abs().max()) for o in outs[1:]])
ids0 = eng.decode(num_steps=8, return_first_logits=True)[1].clone()
ids1 = eng.decode(num_steps=8, return_first_logits=True)[1].clone()
print("step-0 logits run-to-run identical:", bool(torch.equal(ids0, ids1)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    eng.encode(lm)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("encode B=%d: %.3f ms  -> %.0f TF/s" % (B, ms, 12.214 * B / ms))

if os.environ.get("PROBE_SPLIT"):
    # split vs single decode residual stream on two engine INSTANCES (tests/test_gpu_engine.py::test_split_residual...)
    Bs = 34
    res = {}
    for name, opt in (("split", 0), ("single", _lib.OPT_SINGLE_RESIDUAL_STREAM | _lib.OPT_ENCODER_SINGLE_RESIDUAL_STREAM)):
        e2 = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=Bs, options=opt)
        e2.load_params(params)
        enc = e2.encode(lm[:Bs], return_encoded=True).clone()
        ids, lg = e2.decode(num_steps=4, return_first_logits=True)
        res[name] = (enc, lg.clone(), e2.status(2))
        del e2
    a, b = res["split"], res["single"]
    print("status split/single:", a[2], b[2], "enc equal across instances:", bool(torch.equal(a[0], b[0])))
    d = (a[1] - b[1]).double().norm(dim=1) / b[1].double().norm(dim=1)
    print("per-row rel diff of step-0 logits:", [float("%.2e" % v) for v in d.tolist()])
