#!/usr/bin/env python3
"""Train the MT3 network (mt3/gin/model.gin shape) on synthetic music, on one MI355X, with plain PyTorch autograd.

WHY: the note-level tolerance of the reduced-precision engines (bf16, e4m3 K/V caches, MXFP8 encoder) can
only be read off PEAKED distributions that are conditioned on the audio -- random-init weights, boosted or not, re-roll the
rest of a row at the first flipped arg-max.  No checkpoint can be downloaded here, so one is made: a few thousand steps on
pairs (audio rendered from random note lists, the reference's own target tokens for those notes).  The result is a TEST
FIXTURE (tests/golden/mt3_synthetic_ckpt.npz, int8 + per-column scales: 1 byte per weight), not a product: it transcribes
`synthetic.render_notes` timbres and nothing else.

Everything the reference's TRAINING path would do is restated minimally and only here (training is out of scope for the
product): network = mt3/network.py + mt3/layers.py (pre-norm T5, RMSNorm, unscaled dot-product attention,
gated-GELU MLP, bias-free DenseGeneral, fixed sinusoidal positions, separate logits layer) as torch modules whose
parameters carry the Flax names; targets = mt3/tasks.py:142-178's preprocessor chain (tokenize -> split -> tie section ->
run-length shifts -> redundant state changes removed -> EOS) through mt3_amd's bit-exact encode side
(run_length_encoding.py / note_sequences.py); loss = softmax cross-entropy over non-pad targets; log-mel = the product's
own frontend kernel.  AdamW instead of Adafactor, no dropout (the data never repeats).

  python tools/train_synthetic.py --steps 6000 --out gpurun_out/mt3_synthetic_ckpt.npz
"""
import argparse
import json
import math
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FILE_SEGMENTS = 4                      # a training "file" = 4 consecutive segments (8.192 s): segments 1..3 start with ties
SEG_FRAMES, HOP = 256, 128
MAX_TARGET = 192                       # tokens per segment incl. EOS (synthetic pieces: mean 60, max ~110)


# ------------------------------------------------------------------------------------------------ data (host workers)
def file_targets(ns, codec, n_frames, seg=SEG_FRAMES):
    """target ids (vocabulary ids: event index + 3, EOS = 1 appended) of every segment of one file"""
    import numpy as np
    from mt3_amd import note_sequences as NS, run_length_encoding as RLE
    times, values = NS.note_sequence_to_onsets_and_offsets_and_programs(ns)
    ft = np.arange(n_frames) / 125.0
    ev, si, ei, se, sidx = RLE.encode_and_index_events(NS.NoteEncodingState(), times, values, NS.note_event_data_to_events,
                                                       codec, ft, NS.note_encoding_state_to_events)
    out = []
    for lo in range(0, n_frames, seg):
        t = RLE.segment_targets(ev, si, ei, se, sidx, lo, min(lo + seg, n_frames), codec, True)
        t = RLE.remove_redundant_state_changes(t, codec, ("velocity", "program"))
        out.append(np.concatenate([t + 3, [1]]).astype(np.int64))
    return out


_CODEC = None


def make_batch(args):
    """one training batch on the host: note arrays of `files` random pieces + the target ids of their segments"""
    seed, files = args
    import numpy as np
    from mt3_amd import synthetic, vocabularies
    global _CODEC
    if _CODEC is None:
        _CODEC = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
    rng = np.random.default_rng(seed)
    on, off, pitch, amp, fidx = [], [], [], [], []
    tgt = np.zeros((files * FILE_SEGMENTS, MAX_TARGET), np.int64)
    counts = np.full(files * FILE_SEGMENTS, SEG_FRAMES, np.int32)           # audio frames per segment (the last may be short)
    for f in range(files):
        # one file in six ends inside its last segment, as real files do (NB:318-335: the last segment is short, its
        # log-mel rows are 0 after the audio): the model has to stop there
        n_frames = FILE_SEGMENTS * SEG_FRAMES if rng.random() > 1.0 / 6.0 else int(rng.integers(3 * SEG_FRAMES + 8, 4 * SEG_FRAMES))
        counts[f * FILE_SEGMENTS + FILE_SEGMENTS - 1] = n_frames - (FILE_SEGMENTS - 1) * SEG_FRAMES
        ns = synthetic.random_music(n_frames * HOP / 16000.0, seed=int(rng.integers(1 << 62)),
                                    notes_per_second=float(rng.uniform(2.0, 9.0)))
        for i, t in enumerate(file_targets(ns, _CODEC, n_frames)):
            t = t[:MAX_TARGET]
            tgt[f * FILE_SEGMENTS + i, : len(t)] = t
        on += [n.start_time for n in ns.notes]
        off += [n.end_time for n in ns.notes]
        pitch += [n.pitch for n in ns.notes]
        fidx += [f] * len(ns.notes)
    amp = rng.uniform(0.3, 1.0, len(on))
    return {"on": np.array(on), "off": np.array(off), "pitch": np.array(pitch), "amp": amp,
            "file": np.array(fidx, np.int64), "targets": tgt, "seed": seed, "counts": counts}


# ------------------------------------------------------------------------------------------------ network (torch)
def build_model(cfg, params, device):
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    def sinusoidal(max_len, d):
        pos = torch.arange(max_len)[:, None].float()
        div = torch.exp(torch.arange(d // 2).float() * (-math.log(10000.0) / (d // 2 - 1)))     # mt3/layers.py:51-82
        return torch.cat([torch.sin(pos * div), torch.cos(pos * div)], 1)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.names = sorted(params)
            self.p = nn.ParameterList([nn.Parameter(torch.from_numpy(params[n]).clone()) for n in self.names])
            self.index = {n: i for i, n in enumerate(self.names)}
            self.register_buffer("pe", sinusoidal(2048, cfg.emb_dim))

        def w(self, name):
            return self.p[self.index[name]]

        def rms(self, x, name):
            xf = x.float()
            return (xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6) * self.w(name)).to(x.dtype)

        def heads(self, x, name):
            B, L, _ = x.shape
            return (x @ self.w(name).to(x.dtype)).view(B, L, cfg.num_heads, cfg.head_dim).transpose(1, 2)

        def mha(self, prefix, xq, xkv, causal):
            q, k, v = (self.heads(xq, prefix + "/query/kernel"), self.heads(xkv, prefix + "/key/kernel"),
                       self.heads(xkv, prefix + "/value/kernel"))
            o = F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=1.0)           # UNSCALED (layers.py:230-234)
            B, H, L, D = o.shape
            return o.transpose(1, 2).reshape(B, L, H * D) @ self.w(prefix + "/out/kernel").to(o.dtype)

        def mlp(self, prefix, x):
            g = F.gelu(x @ self.w(prefix + "/wi_0/kernel").to(x.dtype), approximate="tanh")
            return (g * (x @ self.w(prefix + "/wi_1/kernel").to(x.dtype))) @ self.w(prefix + "/wo/kernel").to(x.dtype)

        def forward(self, logmel, dec_in):
            """logmel f32 [B, T, 512], dec_in int64 [B, L] (BOS-shifted targets) -> logits [B, L, V]"""
            x = logmel.to(torch.bfloat16)
            x = x @ self.w("encoder/continuous_inputs_projection/kernel").to(x.dtype) + self.pe[: x.shape[1]].to(x.dtype)
            x = x.float()                                           # f32 residual stream, bf16 matmuls
            for i in range(cfg.num_encoder_layers):
                L = "encoder/layers_%d" % i
                x = x + self.mha(L + "/attention", *(2 * [self.rms(x, L + "/pre_attention_layer_norm/scale").bfloat16()]),
                                 False).float()
                x = x + self.mlp(L + "/mlp", self.rms(x, L + "/pre_mlp_layer_norm/scale").bfloat16()).float()
            enc = self.rms(x, "encoder/encoder_norm/scale").bfloat16()
            y = self.w("decoder/token_embedder/embedding")[dec_in] + self.pe[: dec_in.shape[1]]
            for i in range(cfg.num_decoder_layers):
                L = "decoder/layers_%d" % i
                h = self.rms(y, L + "/pre_self_attention_layer_norm/scale").bfloat16()
                y = y + self.mha(L + "/self_attention", h, h, True).float()
                h = self.rms(y, L + "/pre_cross_attention_layer_norm/scale").bfloat16()
                y = y + self.mha(L + "/encoder_decoder_attention", h, enc, False).float()
                y = y + self.mlp(L + "/mlp", self.rms(y, L + "/pre_mlp_layer_norm/scale").bfloat16()).float()
            y = self.rms(y, "decoder/decoder_norm/scale").bfloat16()
            return (y @ self.w("decoder/logits_dense/kernel").to(y.dtype)).float()

        def export(self):
            return {n: self.p[i].detach().float().cpu().numpy() for i, n in enumerate(self.names)}

    return Net().to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--files", type=int, default=64, help="files per batch (x 4 segments each)")
    ap.add_argument("--lr", type=float, default=6e-4)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--minutes", type=float, default=0.0, help="stop after this many minutes of training (0: run all steps)")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/mt3_synthetic_ckpt.npz")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    # the host workers fork BEFORE anything touches the GPU runtime
    pool = mp.get_context("fork").Pool(args.workers)

    def batch_stream(ahead=96):
        """batches in seed order, at most `ahead` of them in flight or waiting (the workers outrun the GPU many times over)"""
        import collections
        pending, i = collections.deque(), 0
        while True:
            while len(pending) < ahead:
                pending.append(pool.apply_async(make_batch, ((args.seed * 1000003 + i, args.files),)))
                i += 1
            yield pending.popleft().get()
    batches = batch_stream()

    import numpy as np
    import torch
    import torch.nn.functional as F
    from mt3_amd import checkpoints, network, spectrograms, synthetic
    dev = args.device
    cfg = network.T5Config(dtype="float32", num_encoder_layers=args.layers, num_decoder_layers=args.layers)
    model = build_model(cfg, network.init_random_params(cfg, seed=args.seed), dev)
    decay = [p for n, p in zip(model.names, model.p) if p.ndim == 2]
    other = [p for n, p in zip(model.names, model.p) if p.ndim != 2]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": other, "weight_decay": 0.0}],
                            lr=args.lr, betas=(0.9, 0.98), eps=1e-9)
    n_samples = FILE_SEGMENTS * SEG_FRAMES * HOP

    def to_device(b):
        audio = synthetic.render_notes(b["on"], b["off"], b["pitch"], b["amp"], b["file"], args.files, n_samples,
                                       seed=int(b["seed"]) & 0x7FFFFFFF, device=dev)
        audio = audio.reshape(args.files * FILE_SEGMENTS, SEG_FRAMES * HOP)
        if dev == "cuda":
            for i in np.nonzero(b["counts"] < SEG_FRAMES)[0]:
                audio[i, int(b["counts"][i]) * HOP:] = 0.0                                  # nothing after the end of a file
            logmel = spectrograms.compute_spectrogram_batch(audio, b["counts"])              # the product's frontend kernel
        else:                                                                               # CPU dry run of the script only
            spec = torch.stft(audio, 2048, 128, window=torch.hann_window(2048), center=False, return_complex=True)
            logmel = torch.log(spec.abs().transpose(1, 2)[:, :, :512].clamp_min(1e-5))
            logmel = F.pad(logmel, (0, 0, 0, SEG_FRAMES - logmel.shape[1]))
        tgt = torch.from_numpy(b["targets"]).to(dev)
        L = int((tgt > 0).sum(1).max())
        tgt = tgt[:, : max(8, L)]
        dec_in = F.pad(tgt[:, :-1], (1, 0))                                                 # BOS = 0, shifted right
        return logmel, dec_in, tgt

    def lr_at(step, elapsed=0.0):
        """linear warm-up, then a cosine to 5 % over the run -- the run's length being --steps or --minutes, whichever ends first"""
        if step < args.warmup:
            return args.lr * (step + 1) / args.warmup
        t = (step - args.warmup) / max(1, args.steps - args.warmup)
        if args.minutes:
            t = max(t, elapsed / (args.minutes * 60.0))
        return args.lr * (0.05 + 0.95 * 0.5 * (1.0 + math.cos(math.pi * min(1.0, t))))

    held = [to_device(next(batches)) for _ in range(2)]                                     # held-out batches (never trained on)

    def evaluate(m):
        m.eval()
        with torch.no_grad():
            tot = hit = 0
            loss = 0.0
            for logmel, dec_in, tgt in held:
                lg = m(logmel, dec_in)
                mask = tgt > 0
                loss += float(F.cross_entropy(lg[mask], tgt[mask]))
                hit += int((lg.argmax(-1)[mask] == tgt[mask]).sum())
                tot += int(mask.sum())
        m.train()
        return loss / len(held), hit / tot

    t0 = time.perf_counter()
    log = []
    step = 0
    for step in range(args.steps):
        logmel, dec_in, tgt = to_device(next(batches))
        lr = lr_at(step, time.perf_counter() - t0)
        for g in opt.param_groups:
            g["lr"] = lr
        lg = model(logmel, dec_in)
        mask = tgt > 0
        loss = F.cross_entropy(lg[mask], tgt[mask])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        if step % 100 == 0 or step == args.steps - 1:
            el = time.perf_counter() - t0
            rec = {"step": step, "loss": float(loss.detach()), "lr": lr, "elapsed_s": el}
            if step % 500 == 0 or step == args.steps - 1:
                rec["held_out_loss"], rec["held_out_token_accuracy"] = evaluate(model)
            log.append(rec)
            print("TRAIN " + json.dumps(rec), flush=True)
        if args.minutes and (time.perf_counter() - t0) > args.minutes * 60.0:
            print("TRAIN stopping at step %d: %.1f minutes" % (step, args.minutes), flush=True)
            break
    pool.terminate()

    # ---- export: int8 + per-column scales; the checkpoint IS the de-quantised weights, so score those
    params = model.export()
    ho_loss, ho_acc = evaluate(model)
    meta = {"what": "MT3 (mt3/gin/model.gin shape, %d + %d layers) trained by tools/train_synthetic.py on synthetic music "
                    "(mt3_amd.synthetic.random_music / render_notes); a TEST FIXTURE" % (args.layers, args.layers),
            "steps": step + 1, "files_per_batch": args.files, "segments_per_batch": args.files * FILE_SEGMENTS, "lr": args.lr,
            "seed": args.seed, "train_seconds": time.perf_counter() - t0,
            "held_out_loss_f32_weights": ho_loss, "held_out_token_accuracy_f32_weights": ho_acc}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    deq = checkpoints.save_compact_npz(args.out, params, meta)
    with torch.no_grad():
        for i, n in enumerate(model.names):
            model.p[i].copy_(torch.from_numpy(deq[n]))
    q_loss, q_acc = evaluate(model)
    meta["held_out_loss_int8_weights"], meta["held_out_token_accuracy_int8_weights"] = q_loss, q_acc
    checkpoints.save_compact_npz(args.out, params, meta)
    with open(os.path.splitext(args.out)[0] + "_train_log.json", "w") as f:
        json.dump({"meta": meta, "log": log}, f)
    print("TRAIN_DONE " + json.dumps(meta), flush=True)
    print("checkpoint bytes:", os.path.getsize(args.out), flush=True)


if __name__ == "__main__":
    main()
