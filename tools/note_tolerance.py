#!/usr/bin/env python3
"""Note-level F1 of the reduced-precision engines against the f32 engine on one synthetic file.

  python tools/note_tolerance.py [--minutes 10] [--shape mt3|base] [--eos 4.0] [--weights boosted|<ckpt.npz>]

boosted = random-init weights whose logits favour note events (synthetic.boost_note_events): thousands of notes, but the
distributions are as flat as any random-init model's, so one flipped arg-max re-rolls the rest of a row.
<ckpt.npz> = a checkpoint written by tools/train_synthetic.py (peaked distributions, conditioned on the audio).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--shape", default="mt3", choices=["mt3", "base"])
    ap.add_argument("--eos", type=float, default=4.0)
    ap.add_argument("--weights", default="boosted")
    ap.add_argument("--decoding", default="beam1")
    args = ap.parse_args()
    import dataclasses
    from mt3_amd import evaluation, network, synthetic
    shape = dataclasses.replace(network.MT3_BASE if args.shape == "base" else network.MT3_SMALL, dtype="float32")
    truth = None
    if args.weights == "boosted":
        n_seg = int(-(-args.minutes * 60.0 // 2.048))
        wav = synthetic.synth_audio(n_seg, seed=77, tones=6).reshape(-1)[: int(args.minutes * 60.0 * 16000)].cpu().numpy()
        params = synthetic.boost_note_events(network.init_random_params(shape, seed=0), eos=args.eos)
    else:
        from mt3_amd import checkpoints
        params = checkpoints.load_compact_npz(args.weights)
        truth, wav = synthetic.synth_music(args.minutes * 60.0, seed=77)
    t0 = time.perf_counter()
    rep = evaluation.compare_engines(params, wav, shape, decoding=args.decoding, truth=truth)
    rep["wall_s"] = time.perf_counter() - t0
    print("NOTE_TOLERANCE " + json.dumps({"shape": args.shape, "weights": args.weights, "minutes": args.minutes, **rep}))


if __name__ == "__main__":
    main()
