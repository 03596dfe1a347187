import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import network, spectrograms, synthetic
cfg = network.T5Config(dtype=os.environ.get("DT", "bfloat16"))
params = network.init_random_params(cfg, seed=0)
for B in (8, 64, 256):
    for chains in (1, 2):
        eng = network.Transformer(cfg, max_batch=B, decode_chains=chains)
        eng.load_params(params)
        audio = synthetic.synth_audio(B, seed=11)
        lm = spectrograms.compute_spectrogram_batch(audio, None)
        enc1 = eng.encode(lm, return_encoded=True).clone()
        ids1 = eng.decode(num_steps=48).clone()
        enc1b = eng.encode(lm, return_encoded=True).clone()
        ids1b = eng.decode(num_steps=48).clone()
        perm = torch.randperm(B, device="cuda")
        enc2 = eng.encode(lm[perm].contiguous(), return_encoded=True).clone()
        ids2, lg2 = eng.decode(num_steps=48, return_first_logits=True)
        eng.encode(lm)
        ids3, lg1 = eng.decode(num_steps=48, return_first_logits=True)
        bad = (ids2 != ids1[perm]).any(1)
        first = [(int(r), int((ids2[r] != ids1[perm][r]).nonzero()[0])) for r in bad.nonzero().flatten()[:5]]
        print(f"B={B} chains={chains}: repeat enc equal {torch.equal(enc1, enc1b)}, repeat ids equal {torch.equal(ids1, ids1b)}, "
              f"perm enc equal {torch.equal(enc2, enc1[perm])} (max diff {(enc2 - enc1[perm]).abs().max().item():.3e}), "
              f"perm logits0 max diff {(lg2 - lg1[perm]).abs().max().item():.3e}, perm ids rows differing {int(bad.sum())} first (row, step) {first}")
        del eng
