"""Where do the small decode kernels spend their ~5 us?  Decode with both attention kernels skipped
(caches stay hot: only ~44 MB of weights + activations are touched per step) vs with them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import network, spectrograms, synthetic

B = int(os.environ.get("B", 256))
cfg = network.T5Config()
eng = network.Transformer(cfg, max_batch=B)
eng.load_params(network.init_random_params(cfg, seed=0))
stream = torch.cuda.Stream()
audio = synthetic.synth_audio(B, seed=1)


def ms(**kw):
    with torch.cuda.stream(stream):
        eng.debug_decode(num_steps=2, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        eng.debug_decode(num_steps=1024, **kw)
        b.record(stream)
    b.synchronize()
    return a.elapsed_time(b)


with torch.cuda.stream(stream):
    eng.encode(spectrograms.compute_spectrogram_batch(audio, None))
full = ms()
noself = ms(skip_self_attn=True)
nocross = ms(skip_cross_attn=True)
noattn = ms(skip_self_attn=True, skip_cross_attn=True)
print(f"B={B} full {full:.1f} ms | no self {noself:.1f} | no cross {nocross:.1f} | no attention {noattn:.1f} ms "
      f"-> {noattn / 1024 * 1e3 / 59:.2f} us per small kernel with hot caches; "
      f"with attention running: {(full - (full - noself) - (full - nocross)) / 1024 * 1e3 / 59:.2f} us")
