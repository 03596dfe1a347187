#!/usr/bin/env python3
"""A ragged corpus through the engine, two ways (VERDICT r4 "next" #1 / #3; SURVEY.md 8(d)'s synthetic EOS schedule on
BASELINE configs[3]): output lengths ~ clipped N(300, 100) imposed PER SEGMENT,

  batch   the reference's shape of loop (NB:295-301): one engine call per `--slots` segments, each call runs until its
          longest row is done (MT3_DECODE_EARLY_EXIT: finished rows are retired and compacted, never refilled)
  refill  mt3_engine_transcribe: `--slots` decode slots, a finished slot restarts on the next encoded segment

Same pipeline inside the clock for both: log-mel of every segment -> encoder -> decode -> ids -> tokens (device) -> host
copy.  Prints one JSON line; `--check` also asserts that the two ways return the same tokens.
"""
import argparse
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEG_SECONDS = 2.048
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=10000)
    ap.add_argument("--slots", type=int, default=1250)
    ap.add_argument("--dtype", default="float32", choices=["float32", "bfloat16"])
    ap.add_argument("--kv-dtype", default="", choices=["", "fp8_e4m3"])
    ap.add_argument("--dense-dtype", default="", choices=["", "fp8_e4m3"])
    ap.add_argument("--model", default="mt3", choices=["mt3", "base"], help="base = gin/ismir2022/base.gin shape (BASELINE configs[4])")
    ap.add_argument("--mode", default="both", choices=["both", "batch", "refill"])
    ap.add_argument("--eos-mean", type=float, default=300.0)
    ap.add_argument("--eos-sd", type=float, default=100.0)
    ap.add_argument("--decode-steps", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--single-stream", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--spin-waits", action="store_true", help="MT3_OPT_SPIN_WAITS: the workers spin as in round 4")
    ap.add_argument("--options", type=int, default=0, help="mt3_engine_config.options bits")
    ap.add_argument("--decode-probe", action="store_true",
                    help="also time one canonical full-length decode of `--slots` rows (ms, host CPU seconds)")
    ap.add_argument("--polls", default="0", help="refill mode: comma list of poll intervals to try (0 = the product's)")
    ap.add_argument("--groups", default="0", help="refill mode: comma list of row-group counts to try (0 = the product's; "
                                                  "+ 16: the refill chunks' encoder passes left out, differential timing)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from mt3_amd import _lib, network, spectrograms, synthetic, vocabularies

    N, S, L = args.segments, args.slots, 1024
    import dataclasses
    cfg = dataclasses.replace(network.MT3_BASE if args.model == "base" else network.MT3_SMALL, dtype=args.dtype,
                              kv_dtype=args.kv_dtype, dense_dtype=args.dense_dtype)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=S,
                              options=(_lib.OPT_SPIN_WAITS if args.spin_waits else 0) | args.options)
    eng.load_params(network.init_random_params(cfg, seed=0))
    vocab = vocabularies.vocabulary_from_codec(vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1)))
    audio = torch.cat([synthetic.synth_audio(min(1024, N - s), seed=1000 + s) for s in range(0, N, 1024)])
    rng = np.random.default_rng(0)
    lens = np.clip(np.rint(rng.normal(args.eos_mean, args.eos_sd, N)), 1, args.decode_steps).astype(np.int32)
    stream = torch.cuda.Stream()

    def logmel_all():
        out = torch.empty((N, 256, 512), device="cuda", dtype=torch.float32)
        for a in range(0, N, 1024):
            out[a:a + 1024] = spectrograms.compute_spectrogram_batch(audio[a:a + 1024], None)
        return out

    def run_batch():
        toks, steps = [], 0
        with torch.cuda.stream(stream):
            for a in range(0, N, S):
                eng.debug_set_eos_schedule(lens[a:a + S])
                eng.encode(spectrograms.compute_spectrogram_batch(audio[a:a + S], None))
                ids = eng.decode(num_steps=args.decode_steps, early_exit=True, single_stream=args.single_stream)
                steps += eng.steps_run
                toks.append(vocab.decode_tf(ids))
            host = torch.cat(toks).cpu().numpy()
        return host, {"decode_steps_run": steps, "calls": -(-N // S)}

    variant = {"poll": 0, "groups": 0}

    def run_refill():
        with torch.cuda.stream(stream):
            eng.debug_set_eos_schedule(lens)
            ids = eng.transcribe(logmel_all(), num_steps=args.decode_steps, single_stream=args.single_stream,
                                 debug_poll_steps=variant["poll"], debug_row_groups=variant["groups"] % 16,
                                 debug_skip_encoder_passes=variant["groups"] >= 16)
            host = vocab.decode_tf(ids).cpu().numpy()
        return host, dict(eng.transcribe_stats)

    def timed(fn):
        fn()                                          # warm-up: graphs captured, staging allocated
        torch.cuda.synchronize()
        best = None
        for _ in range(args.reps):
            r0 = resource.getrusage(resource.RUSAGE_SELF)
            t0 = time.perf_counter()
            host, info = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            r1 = resource.getrusage(resource.RUSAGE_SELF)
            cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
            if best is None or dt < best[0]:
                best = (dt, cpu, host, info)
        return best

    esz = 2 if args.dtype == "bfloat16" else 4
    H, nl = cfg.num_heads, cfg.num_decoder_layers
    kv1 = H * (2.0 * 64 + 8.0) if args.kv_dtype else 2.0 * H * 64 * esz
    qo = 2.0 * H * 64 * esz
    ll = np.minimum(lens, args.decode_steps).astype(np.float64)
    live_bytes = nl * float((kv1 * (ll * (ll + 1) / 2) + (kv1 + qo) * ll).sum() + ((kv1 * 256 + qo) * ll).sum())
    out = {"segments": N, "slots": S, "model": args.model,
           "dtype": args.dtype + ("+fp8kv" if args.kv_dtype else "") + ("+mxfp8 encoder" if args.dense_dtype else ""),
           "spin_waits": args.spin_waits, "options": args.options,
           "lengths": {"mean": float(ll.mean()), "max": int(ll.max())}, "live_row_kv_bytes": live_bytes,
           "single_stream": args.single_stream}
    hosts = {}
    if args.decode_probe:
        with torch.cuda.stream(stream):
            eng.encode(spectrograms.compute_spectrogram_batch(audio[:S], None))
            eng.decode(num_steps=8)
        torch.cuda.synchronize()
        probe = {}
        for key, kw in (("row_groups", {}), ("single_stream", {"single_stream": True})):
            r0 = resource.getrusage(resource.RUSAGE_SELF)
            t0 = time.perf_counter()
            with torch.cuda.stream(stream):
                eng.decode(num_steps=args.decode_steps, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            r1 = resource.getrusage(resource.RUSAGE_SELF)
            probe[key] = {"decode_ms": dt * 1e3, "host_cpu_s": (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime),
                          "groups": eng.status(_lib.STATUS_LAST_DECODE_GROUPS)}
        out["full_length_decode"] = probe
    try:
        for mode, fn in (("batch", run_batch), ("refill", run_refill)):
            if args.mode not in ("both", mode):
                continue
            dt, cpu, host, info = timed(fn)
            hosts[mode] = host
            got_len = np.where((host == -1).any(1), (host == -1).argmax(1) + 1, host.shape[1])
            out[mode] = {"audio_s_per_s": N * SEG_SECONDS / dt, "seconds": dt, "host_cpu_s": cpu,
                         "hbm_frac_on_live_bytes_whole_pass": live_bytes / dt / 1e9 / HBM_PEAK_GBS,
                         "decoded_mean_len": float(got_len.mean()), "row_groups": eng.status(_lib.STATUS_LAST_DECODE_GROUPS),
                         "tokens_sha16": __import__("hashlib").sha256(np.ascontiguousarray(host).tobytes()).hexdigest()[:16],
                         "graph_fallbacks": eng.status(_lib.STATUS_GRAPH_FALLBACKS), **info}
    finally:
        eng.debug_set_eos_schedule(None)
    ab = [(int(p), int(g)) for p in args.polls.split(",") for g in args.groups.split(",") if (int(p), int(g)) != (0, 0)]
    if ab and args.mode in ("both", "refill"):
        out["variants"] = []
        try:
            for p, g in ab:
                variant.update(poll=p, groups=g)
                dt, cpu, host, info = timed(run_refill)
                out["variants"].append({"poll_steps": p, "row_groups": g, "audio_s_per_s": N * SEG_SECONDS / dt, "seconds": dt,
                                        "host_cpu_s": cpu, "steps_run": info["steps_run"], "polls": info["polls"],
                                        "starved_polls": info["starved_polls"],
                                        "same_tokens": bool(np.array_equal(host, hosts.get("refill", host)))})
        finally:
            eng.debug_set_eos_schedule(None)
    if "batch" in out and "refill" in out:
        out["speedup"] = out["refill"]["audio_s_per_s"] / out["batch"]["audio_s_per_s"]
        same = bool(np.array_equal(hosts["batch"], hosts["refill"]))
        out["tokens_identical"] = same
        if args.check:
            assert same, "refill and batch-synchronous tokens differ"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
