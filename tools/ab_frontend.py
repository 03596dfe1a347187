"""Log-mel frontend: 32-frame tiles (8 waves, round 3) against 16-frame tiles (4 waves), 256 and 2,048 segments, HIP
events; the two must produce bit-identical log-mels (a frame's arithmetic does not depend on its tile)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, spectrograms, synthetic  # noqa: E402

lib = _lib.load()
outs = {}
for n in (256, 2048):
    audio = torch.cat([synthetic.synth_audio(min(1024, n - s), seed=50 + s) for s in range(0, n, 1024)])
    nf = [256 if i % 5 else 77 for i in range(n)]
    for small in (0, 1):
        _lib.check(lib.mt3_debug_set_knob(_lib.DEBUG_KNOB_FRONTEND_32_FRAME_TILES, 1 - small))
        spectrograms.compute_spectrogram_batch(audio, None)
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                spectrograms.compute_spectrogram_batch(audio, None)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        outs[(n, small)] = (spectrograms.compute_spectrogram_batch(audio, None).clone(),
                            spectrograms.compute_spectrogram_batch(audio, nf).clone())
        print("%4d segments, %s: %.1f us  %.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (
            n, "16-frame tiles" if small else "32-frame tiles", best * 1e3, 655360 * n / (best * 1e-3) / 1e9,
            655360 * n / (best * 1e-3) / 8e12 * 100), flush=True)
    print("   bit-identical across tile sizes:", bool(torch.equal(outs[(n, 0)][0], outs[(n, 1)][0])),
          bool(torch.equal(outs[(n, 0)][1], outs[(n, 1)][1])), flush=True)
lib.mt3_debug_set_knob(_lib.DEBUG_KNOB_FRONTEND_32_FRAME_TILES, 0)
