#!/usr/bin/env python3
"""bench.py -- audio-seconds transcribed per second on MI355X (BASELINE.json's metric).

One "step" = one full pass of the hot path over one batch of synthetic 16 kHz segments that are
already resident in HBM:  log-mel frontend -> T5 encoder -> cross-K/V -> 1024-step greedy decode
(hipGraph replay per step, NO early exit: random weights never emit EOS reliably)
-> ids->tokens kernel -> (N>1: RCCL all-gather of the int32 token rows) -> host run-length /
note decoding of every row (C++ in libmt3hip.so; on rank 0, on a worker thread that overlaps the next
batch's launches and is joined before the clock stops).

Workload at N=1: BASELINE.json configs[2] ("MT3-base full encoder-decoder greedy decode, batch=256
synthetic segments, 1xMI355X with hipGraph") -- the largest single-GPU configuration and the only
one that *transcribes* (configs[1] is encoder-only and would leave the decoder out of the timed
region).  Scaling is weak: every rank processes `--batch` segments.

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEG_SECONDS = 2.048          # 256 frames * 128 hop / 16 kHz  (mt3.gin:4, spectrograms.py:23-24)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(n_segments: int, decode_steps: int):
    """The oracle (CPU restatement of the reference path: numpy frontend + torch-CPU f32 network +
    pure-Python note decoding) timed on this box's host cores on a bounded sample."""
    import numpy as np
    import torch
    from mt3_amd import network
    from oracle import frontend as OF, network as ON, symbolic as OS
    # the oracle's decode step is a chain of tiny matmuls: more than ~16 threads only adds sync cost
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = network.T5Config(dtype="float32")
    params = network.init_random_params(cfg, seed=0)
    audio = OF.synth_audio(n_segments, seed=0)
    orc = ON.Oracle(params, ON.T5Config())
    t0 = time.perf_counter()
    lm = np.stack([OF.compute_logmel(a, np.float32) for a in audio])
    enc = orc.encode(lm)
    ids = orc.greedy_decode(enc, decode_steps)
    vocab = OS.GenericTokenVocabulary(1388, extra_ids=100)
    toks = vocab.decode_tf(ids)
    codec = OS.build_codec(OS.VocabularyConfig(num_velocity_bins=1))
    preds = [{"est_tokens": OS.trim_eos(t), "start_time": OS.floor_start_time(i * SEG_SECONDS, 100)}
             for i, t in enumerate(toks)]
    OS.event_predictions_to_ns(preds, codec, "ties")
    dt = time.perf_counter() - t0
    return {"value": n_segments * SEG_SECONDS / dt, "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": "%d segments (%.1f s of audio), same path: log-mel + encoder + %d greedy steps + note "
                      "decoding; oracle restatement (numpy/torch-CPU f32), not JAX; %.1f s wall"
                      % (n_segments, n_segments * SEG_SECONDS, decode_steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="segments per GPU per step (c3: 256)")
    ap.add_argument("--decode-steps", type=int, default=1024)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--chains", type=int, default=1,
                    help="independent row groups run as parallel branches of the step graph (measured on "
                         "MI355X/ROCm 7.2 at batch 256: 1 -> 758, 2 -> 744 audio-s/s; "
                         "two chains were +6 %% while the decode GEMMs were 2 us slower)")
    ap.add_argument("--decoding", default="greedy", choices=["greedy", "beam1"],
                    help="token selection: plain greedy (what BASELINE configs[2] names) or the rule of t5x "
                         "beam_search with one beam (what the reference's InferenceModel runs; ~1 %% slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-segments", type=int, default=2)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print("CPU_BASELINE " + json.dumps(cpu_baseline(args.cpu_segments, args.decode_steps)), flush=True)
        return

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mt3_amd import distributed, metrics_utils, network, note_sequences, spectrograms, synthetic, vocabularies

    B, L = args.batch, 1024
    cfg = network.T5Config(dtype=args.dtype)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B, decode_chains=args.chains)
    eng.load_params(network.init_random_params(cfg, seed=0))
    codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
    vocab = vocabularies.vocabulary_from_codec(codec)
    audio = synthetic.synth_audio(B, seed=1000 + rank)                    # [B, 32768] f32 in HBM
    stream = torch.cuda.Stream()                                          # a real (capturable) stream
    start_times = [s * SEG_SECONDS - (s * SEG_SECONDS) % 0.01 for s in range(B * world)]

    # rank 0's host stage (EOS trim + run-length / note decoding in libmt3hip.so) runs on a worker thread, so the
    # NEXT batch's GPU work is already being launched while the previous batch's tokens become notes; every
    # future is joined before the clock stops, so all of it stays inside the timed region
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)
    pending = []

    def host_stage(host):
        eos = host == vocabularies.DECODED_EOS_ID
        n_tok = np.where(eos.any(1), eos.argmax(1), host.shape[1])
        rows = [r[:n] for r, n in zip(host, n_tok)]
        ns, inv, drop = metrics_utils._run(codec, note_sequences.NoteEncodingWithTiesSpec.spec_id, rows, start_times)
        return len(ns.notes)

    def step():
        with torch.cuda.stream(stream):
            logmel = spectrograms.compute_spectrogram_batch(audio, None)
            eng.encode(logmel)
            ids = eng.decode(num_steps=args.decode_steps, beam1=args.decoding == "beam1")
            tokens = vocab.decode_tf(ids)                                  # CUDA int32 [B, L]
            tokens = distributed.gather_token_rows(tokens, world * B)   # RCCL all-gather (identity at N=1)
            if rank == 0:
                host = tokens.cpu().numpy()                                # syncs the stream
                pending.append(pool.submit(host_stage, host))

    def drain():
        n = 0
        while pending:
            n = pending.pop(0).result()
        return n

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    n_notes = drain()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- roofline of the dominant kernel (decode self-attention: HBM streaming of the K/V cache).
    # In-situ and live: HIP events (recorded on the stream the graphs are launched on) around the whole
    # graph-replayed decode, once as it ships and once with that kernel's launches left out of the step
    # graph; the difference / launches = the kernel's average duration inside the real decode loop.
    roof = None
    if rank == 0:
        def decode_ms(**kw):
            with torch.cuda.stream(stream):
                kw["chains"] = 1      # the kernel at full-GPU width, one launch at a time (as rocprofv3 sees it)
                eng.decode(num_steps=2, **kw)                         # capture / warm this graph variant
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                eng.decode(num_steps=args.decode_steps, **kw)
                e1.record(stream)
            e1.synchronize()
            return e0.elapsed_time(e1)

        with torch.cuda.stream(stream):
            eng.encode(spectrograms.compute_spectrogram_batch(audio, None))
        t_full = min(decode_ms(), decode_ms())
        t_noself = min(decode_ms(skip_self_attn=True), decode_ms(skip_self_attn=True))
        t_nocross = min(decode_ms(skip_cross_attn=True), decode_ms(skip_cross_attn=True))
        esize = 2 if args.dtype == "bfloat16" else 4
        H, S, nl = cfg.num_heads, args.decode_steps, cfg.num_decoder_layers
        kv_row = 2.0 * B * H * 64 * esize                            # K+V bytes of one cache position, all rows
        launches = S * nl
        # algorithmic bytes: read the t+1 cached K/V rows + q, write the new row + the output
        self_bytes = nl * sum(kv_row * (t + 1) + kv_row + 2.0 * B * H * 64 * esize for t in range(S))
        cross_bytes = launches * (kv_row * 256 + 2.0 * B * H * 64 * esize)
        self_us = (t_full - t_noself) * 1e3 / launches
        cross_us = (t_full - t_nocross) * 1e3 / launches
        ach = self_bytes / launches / (self_us * 1e-6) / 1e9
        # HBM traffic of that kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
        # passes, FETCH_SIZE x2 on gfx950 -- calibrated on a 1 GiB copy): collected by tools/gpu_pmc.sh at this
        # exact shape, committed as profiles/r1_pmc_summary.json (a bench run cannot wrap itself in rocprofv3)
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r1_pmc_summary.json")) as f:
                pmc = json.load(f)
            if pmc["shape"]["B"] == B and args.dtype == "bfloat16" and args.decode_steps == 1024:
                ratio = pmc["dec_attn_self_append"]["n_keys_513"]["traffic_over_algorithmic"]
                traffic = ratio * self_bytes / launches
                traffic_src = "profiles/r1_pmc_summary.json (measured traffic/algorithmic = %.4f at the mean launch)" % ratio
        except (OSError, KeyError, ValueError):
            pass
        roof = {"bound": "hbm", "kernel": "mt3k::dec_attn_kernel<bf16, APPEND=true> (decode self-attention over the "
                                          "K/V cache)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_launch_us": self_us, "algorithmic_bytes_per_launch": self_bytes / launches, "launches": launches,
                "method": "HIP events on the launch stream around the whole graph-replayed decode, with and "
                          "without this kernel in the step graph; (difference)/launches",
                "decode_ms_single_chain": t_full, "decode_ms_without_self_attn": t_noself,
                "decode_ms_without_cross_attn": t_nocross,
                "cross_attn": {"achieved": cross_bytes / launches / (cross_us * 1e-6) / 1e9, "avg_launch_us": cross_us,
                               "algorithmic_bytes_per_launch": cross_bytes / launches}}

    if rank == 0:
        segs = B * world * args.steps
        value = segs * SEG_SECONDS / dt
        out = {
            "metric": "audio-seconds transcribed/sec (whole node), MT3-base, 1/2/4/8 MI355X",
            "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bfloat16" else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: MT3 (model.gin) random-init, full encoder-decoder %s "
                                   "decode, batch=%d synthetic 2.048 s segments per GPU, %d decode steps (no early "
                                   "exit), hipGraph step replay, ids->tokens + host note decoding included"
                                   % ("greedy" if args.decoding == "greedy" else "beam-1 (t5x beam_search, one beam)",
                                      B, args.decode_steps),
                       "segments_per_gpu": B, "decode_steps": args.decode_steps, "segment_seconds": SEG_SECONDS,
                       "decode_chains": args.chains, "decoding": args.decoding,
                       "parallelism": "dp%d (segments sharded, weights replicated, RCCL all-gather of token rows)"
                                      % world if world > 1 else "single GPU",
                       "notes_decoded_last_step": n_notes},
            "segments_per_s": segs / dt,
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            # separate process, hard wall-clock bound: the bench must finish in minutes on any host
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-segments",
                                    str(args.cpu_segments), "--decode-steps", str(args.decode_steps)],
                                   capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
                out["cpu_baseline"] = json.loads(line[-1][len("CPU_BASELINE "):]) if line else \
                    {"value": None, "unit": "audio-s/s", "cores": 0, "kind": "port", "sample": "failed: " + r.stderr[-300:]}
            except subprocess.TimeoutExpired:
                out["cpu_baseline"] = {"value": None, "unit": "audio-s/s", "cores": 0, "kind": "port",
                                       "sample": "oracle did not finish %d segments in 300 s" % args.cpu_segments}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
