"""PMC pass over the PRODUCT decode schedule (VERDICT r5 #3a): the f32 engine at the headline shape (B = 256, MT3 shape, four
row groups) decodes STEPS positions with direct launches, so that `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
(separate passes) can be reduced PER KERNEL NAME (tools/pmc_decode_summary.py): where do the dense launches' f32 weights
(91 MB per step and group) come from while the K/V stream goes through the Infinity Cache?
The decode starts at cache depth DEPTH (teacher-forced prefix = garbage ids, the caches hold real rows): the attention
launches then stream what a mid-decode step streams (depth 512 = the mean launch of the headline).
A 1 GiB copy calibrates the counters (gfx950 tallies 128-byte requests as 64).
Usage:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o FETCH_SIZE -- python tools/pmc_decode.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import network, spectrograms, synthetic  # noqa: E402

STEPS = int(os.environ.get("PMC_STEPS", "32"))
DEPTH = int(os.environ.get("PMC_DEPTH", "512"))
B = 256
a = torch.empty(256 * 1024 * 1024, device="cuda")
a.normal_()
for _ in range(2):
    b = a.clone()                                   # calibration: 1 GiB read + 1 GiB written
torch.cuda.synchronize()
del a, b
cfg = network.T5Config(dtype="float32")
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B)
eng.load_params(network.init_random_params(cfg, seed=0))
eng.encode(spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=1000), None))
# DEPTH + STEPS positions, direct launches (counters are read per dispatch; the ids are irrelevant).  Only the LAST `STEPS`
# steps are meant to be read: tools/pmc_decode_summary.py keeps the last STEPS x launches-per-step dispatches of each kernel
ids = eng.decode(num_steps=DEPTH + STEPS, use_graph=False)
torch.cuda.synchronize()
print("decoded", tuple(ids.shape), "groups", eng.status(7), flush=True)
