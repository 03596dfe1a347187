#!/usr/bin/env python3
"""bench.py -- audio-seconds transcribed per second on MI355X (BASELINE.json's metric).

One "step" = one full pass of the hot path over one batch of synthetic 16 kHz segments that are
already resident in HBM:  log-mel frontend -> T5 encoder -> cross-K/V -> 1024-step greedy decode
(NO early exit: random weights never emit EOS reliably; at batch >= 128 the decode runs
as 2 or 4 ROW GROUPS, each replaying its own captured step graph on a stream with a hardware queue of
its own -- `config.decode_schedule` says what ran) -> ids->tokens kernel -> (N>1: RCCL gather of the
int32 token rows) -> host run-length / note decoding of every row (C++ in libmt3hip.so; on rank 0, on a
worker thread that overlaps the next batch's launches and is joined before the clock stops).

Precision: the headline computes in FLOAT32, the reference's own precision (mt3/gin/model.gin:50
`dtype = 'float32'`; f32 MFMA operands, f32 K/V caches, token-exact against the oracle).  The bf16
engine, the e4m3-cache engine and BASELINE configs[4] are reported as `extra` blocks with their own
rooflines; `extra.eos_schedule` is SURVEY.md 8(d)'s second figure (synthetic output lengths ~ clipped
N(300,100), early exit + row retirement), `extra.uniform_noise` its pure-noise input run.

Workload at N=1: BASELINE.json configs[2] ("MT3-base full encoder-decoder greedy decode, batch=256
synthetic segments, 1xMI355X with hipGraph") -- the largest single-GPU configuration and the only
one that *transcribes* (configs[1] is encoder-only and would leave the decoder out of the timed
region; it is reported as the driver-timed extra `configs1`).  Default scaling is weak: every rank
processes `--batch` segments per step.  `--corpus N` is BASELINE configs[3]: a fixed N-segment corpus
sharded over the ranks (strong scaling, 10 000 segments = 1250 per GPU at 8 GPUs).

Launch:  python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: spawns N ranks itself)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import math
import os
import resource
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEG_SECONDS = 2.048          # 256 frames * 128 hop / 16 kHz  (mt3.gin:4, spectrograms.py:23-24)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_FP8_PEAK_TFLOPS = 5000.0              # dense MX-scaled fp8 (K = 128 instructions), MI355X_MICROARCH.md
ENC_GFLOP_PER_SEGMENT = 10.603 + 1.611     # encoder + the one-off cross-K/V projections of 8 layers
ENC_ATTN_GFLOP_PER_SEGMENT = 8 * 6 * 2 * 2 * 256 * 256 * 64 / 1e9      # of which QK^T and PV: 8 layers x 6 heads (0.805)
FRONTEND_BYTES_PER_SEGMENT = 655360        # 131072 in + 524288 out
FRONTEND_SOURCES = ("frontend.hip", "frontend_core.h", "frontend_tables.h")


def kernel_source_hash(files=("attention.hip", "device.h", "kernels.h")):
    """Identity of the kernel a PMC file was collected from: sha256 over the CODE of the sources the decode-attention
    kernels (default) are compiled from -- comments and white space removed, so that rewording a comment does not throw a
    measured traffic ratio away while any change to the code does.  Of kernels.h (the launcher interface of ALL kernels)
    only what those kernels see counts: the text of `struct DecAttnArgs`."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mt3_amd", "csrc")
    for f in files:
        with open(os.path.join(d, f), "r", errors="replace") as fh:
            src = fh.read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)          # block comments
        src = re.sub(r"//[^\n]*", " ", src)                         # line comments
        if f == "kernels.h":
            m = re.search(r"struct DecAttnArgs\s*\{.*?\n\};", src, flags=re.S)
            src = m.group(0) if m else src
        h.update(f.encode() + b"\0" + " ".join(src.split()).encode())
    return h.hexdigest()[:16]


# ----------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(n_segments: int, decode_steps: int, enc_segments: int, small_segments: int = 0, parity_file: str = ""):
    """The oracle (CPU restatement of the reference path: numpy frontend + torch-CPU f32 network + pure-Python
    note decoding; NOT JAX -- SURVEY 8c: jax/t5x are not installable here) timed on this box's host cores on a
    bounded sample: (a) the full path on `n_segments` segments as ONE batch (reduced configs[2]; 8 = the reference
    InferenceModel's batch size, NB:190), (b) a batch-scaling probe at `small_segments` segments (first 48 decode steps
    only), (c) configs[1]: log-mel + encoder only on `enc_segments` segments.
    parity_file (written by the GPU leg): the audio of `n_segments` rows of the headline batch (`parity_rows`: spread over all row groups), the product's
    log-mel, ids and notes of those rows as the headline's own schedule produced them -- the timed sample then runs on
    that audio and its tokens / notes are COMPARED with the product's (`parity`): the driver-run line carries parity at
    the credited configuration."""
    import numpy as np
    import torch
    from mt3_amd import network
    from oracle import frontend as OF, network as ON, symbolic as OS
    nproc = os.cpu_count() or 1
    cfg = network.T5Config(dtype="float32")
    params = network.init_random_params(cfg, seed=0)
    audio = OF.synth_audio(max(n_segments, enc_segments), seed=0)
    par = None
    if parity_file:
        with np.load(parity_file) as z:
            par = {k: z[k] for k in z.files}
        n_segments = min(n_segments, par["audio"].shape[0])
        audio = np.concatenate([par["audio"][:n_segments].astype(audio.dtype), audio[n_segments:]])
    orc = ON.Oracle(params, ON.T5Config())
    vocab = OS.GenericTokenVocabulary(1388, extra_ids=100)
    codec = OS.build_codec(OS.VocabularyConfig(num_velocity_bins=1))

    def pick_threads(enc_probe, candidates):
        # the decode step is a chain of small-batch GEMMs whose speed peaks well below a 256-core host's nproc: take
        # the fastest candidate on 4 real decode steps (the probe is not part of the timed sample) and say which
        best = (None, 1e30)
        for th in sorted({min(nproc, c) for c in candidates}):
            torch.set_num_threads(th)
            orc.greedy_decode(enc_probe, 1)
            t0 = time.perf_counter()
            orc.greedy_decode(enc_probe, 4)
            dt = time.perf_counter() - t0
            if dt < best[1]:
                best = (th, dt)
        return best[0]

    def full_path(n, candidates):
        with torch.no_grad():
            torch.set_num_threads(min(nproc, 32))
            enc_probe = orc.encode(np.stack([OF.compute_logmel(a, np.float32, tables="tf32") for a in audio[:n]]))
            threads = pick_threads(enc_probe, candidates)
            torch.set_num_threads(threads)
            t0 = time.perf_counter()
            lm = np.stack([OF.compute_logmel(a, np.float32, tables="tf32") for a in audio[:n]])
            enc = orc.encode(lm)
            ids, logits = orc.greedy_decode(enc, decode_steps, return_logits=True)
            toks = vocab.decode_tf(ids)
            preds = [{"est_tokens": OS.trim_eos(t), "start_time": OS.floor_start_time(i * SEG_SECONDS, 100)}
                     for i, t in enumerate(toks)]
            ns = OS.event_predictions_to_ns(preds, codec, "ties")["est_ns"]
            return time.perf_counter() - t0, threads, (lm, ids, logits, ns)

    def divergences(ids_cpu, logits, ids_gpu):
        """(rows token-exact over all steps, [first divergence of every other row with the oracle's top-2 margin there])"""
        exact, div = 0, []
        for r in range(ids_cpu.shape[0]):
            d = np.nonzero(ids_cpu[r] != ids_gpu[r, : ids_cpu.shape[1]])[0]
            if d.size == 0:
                exact += 1
                continue
            t = int(d[0])
            lg = logits[r, t].double()
            top = torch.topk(lg, 2).values
            div.append({"row": r, "step": t, "oracle_id": int(ids_cpu[r, t]), "product_id": int(ids_gpu[r, t]),
                        "oracle_top2_margin_over_sigma": float((top[0] - top[1]) / lg.std())})
        return exact, div

    t_begin = time.perf_counter()
    dt, threads, (lm_cpu, ids_cpu, logits_cpu, ns_cpu) = full_path(n_segments, (16, 32, 64))
    out = {"value": n_segments * SEG_SECONDS / dt, "unit": "audio-s/s", "cores": threads, "nproc": nproc,
           "kind": "port",
           "sample": "%d segments as one batch (%.1f s of audio; the reference InferenceModel's own batch size, NB:190), "
                     "same path: log-mel + encoder + %d greedy steps + note decoding; oracle restatement (numpy/torch-CPU "
                     "f32), not JAX; %d torch threads (fastest of 16/32/64 of nproc=%d on a 4-step probe); %.1f s wall"
                     % (n_segments, n_segments * SEG_SECONDS, decode_steps, threads, nproc, dt)}
    if par is not None:
        # ---- parity at the credited configuration: the oracle's audio -> tokens -> notes of these rows against the
        # product's, as the HEADLINE's schedule (row groups, graph replay) produced them inside the full batch
        ids_gpu = par["ids"][:n_segments]
        exact, div = divergences(ids_cpu, logits_cpu, ids_gpu)
        lm_gpu = par["logmel"][:n_segments]
        sig = np.exp(lm_cpu) > 1e-2
        got_notes = [tuple(r) for r in par["notes"].tolist()]
        ref_notes = [tuple(float(v) for v in t[:6]) for t in ns_cpu.as_tuples()]
        parity = {"rows": int(n_segments), "steps": int(decode_steps), "token_exact_rows": exact, "first_divergence": div,
                  "notes_equal": got_notes == ref_notes, "notes": len(ref_notes),
                  "logmel_max_abs_diff_where_mel_above_1e-2": float(np.abs(lm_gpu - lm_cpu)[sig].max()),
                  "batch_rows": [int(r) for r in par["rows"]] if "rows" in par else list(range(int(n_segments))),
                  "what": "oracle (numpy frontend -> torch-CPU f32 network -> greedy loop -> pure-Python note decoding) "
                          "on the audio of %d rows of the headline batch spread over all its row groups (batch_rows), "
                          "against the ids / notes the headline engine produced for those rows inside its %d-row batch "
                          "(%s); the rows are taken as consecutive segments of one track on both sides"
                          % (n_segments, int(par["batch"]), str(par["schedule"]))}
        if div and time.perf_counter() - t_begin < 120.0:
            # rows that differ end to end: is it the frontend's 1e-4 or the network?  the oracle's network alone, fed the
            # PRODUCT's log-mel of those rows
            rows = [d["row"] for d in div]
            with torch.no_grad():
                ids2, lg2 = orc.greedy_decode(orc.encode(lm_gpu[rows]), decode_steps, return_logits=True)
            ex2, div2 = divergences(ids2, lg2, ids_gpu[rows])
            for d in div2:
                d["row"] = rows[d["row"]]
            parity["network_on_product_logmel"] = {"rows": len(rows), "token_exact_rows": ex2, "first_divergence": div2}
            parity["token_exact_rows_network"] = exact + ex2
        out["parity"] = parity
    if small_segments and time.perf_counter() - t_begin < 110.0:
        # does a LARGER batch use the host better?  (VERDICT r2 #9 asked for 32 segments: measured on MI355X hosts it is
        # SLOWER per audio-second -- 190 s for the full 1024 steps, 0.35 against 0.52 audio-s/s -- so the full-length
        # sample stays at 8 and the larger batch is reported as a bounded probe: ms per decode step over the first 48
        # steps at both batch sizes, threads = the fastest of 32/64/128 for the large one)
        nb = small_segments
        with torch.no_grad():
            torch.set_num_threads(min(nproc, 32))
            enc_nb = orc.encode(np.stack([OF.compute_logmel(a, np.float32, tables="tf32") for a in audio[:nb]]))
            th_nb = pick_threads(enc_nb, (32, 64, 128))
            torch.set_num_threads(th_nb)
            t0 = time.perf_counter()
            orc.greedy_decode(enc_nb, 48)
            ms_nb = (time.perf_counter() - t0) * 1e3 / 48
            torch.set_num_threads(threads)
            enc_8 = enc_nb[:n_segments]
            t0 = time.perf_counter()
            orc.greedy_decode(enc_8, 48)
            ms_8 = (time.perf_counter() - t0) * 1e3 / 48
        out["batch_probe"] = {"segments": nb, "threads": th_nb, "ms_per_decode_step": ms_nb,
                              "ms_per_decode_step_at_%d_segments" % n_segments: ms_8,
                              "audio_s_per_s_ratio": (nb / ms_nb) / (n_segments / ms_8),
                              "sample": "first 48 cached decode steps only (shallow cache): a probe of batch scaling, "
                                        "not a throughput figure"}
    with torch.no_grad():
        # (c) encoder-only (configs[1]) at all cores: big GEMMs, this one does scale with threads
        torch.set_num_threads(nproc)
        if time.perf_counter() - t_begin > 150.0:      # a slow (shared) host: keep the whole CPU leg bounded
            enc_segments = max(16, enc_segments // 2)
        t1 = time.perf_counter()
        lm2 = np.stack([OF.compute_logmel(a, np.float32, tables="tf32") for a in audio[:enc_segments]])
        orc.encode(lm2)
        dt2 = time.perf_counter() - t1
    out["encoder_only"] = {"value": enc_segments * SEG_SECONDS / dt2, "unit": "audio-s/s",
                           "segments_per_s": enc_segments / dt2, "cores": nproc,
                           "sample": "configs[1] on the CPU: log-mel + encoder, %d segments, %d torch threads, "
                                     "%.1f s wall" % (enc_segments, nproc, dt2)}
    return out


def parity_rows(batch: int, n: int, groups: int):
    """Which rows of the headline batch the oracle is compared on: `n` rows spread over ALL row groups of the decode
    schedule (group g decodes rows [g * batch / groups, (g + 1) * batch / groups): csrc/engine.hip), first / interior /
    last rows of the groups included -- at batch 256, 4 groups, n = 8: 0, 37, 64, 100, 128, 191, 200, 255."""
    if n >= batch:
        return list(range(batch))
    if batch == 256 and n == 8:
        return [0, 37, 64, 100, 128, 191, 200, 255]
    groups = max(1, min(groups, n))
    per = batch // groups
    rows = []
    for i in range(n):
        g = i % groups
        k = i // groups                                   # k-th pick inside group g: first row, then last, then interior
        lo, hi = g * per, (batch if g == groups - 1 else (g + 1) * per) - 1
        rows.append(lo if k == 0 else hi if k == 1 else lo + ((2 * k - 3) * (hi - lo)) // (2 * max(1, n // groups)))
    return sorted(set(rows))[:n]


# ----------------------------------------------------------------------------------------- rank spawning
def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks (one per GPU) under torch.distributed.run."""
    if "--dry-run" not in sys.argv:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but this node exposes %d GPU(s); not faking a multi-GPU line\n" % (n, have))
            return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="segments per GPU per step (c3: 256)")
    ap.add_argument("--decode-steps", type=int, default=1024)
    ap.add_argument("--dtype", default="float32", choices=["bfloat16", "float32"],
                    help="MFMA operand / K/V cache type of the HEADLINE engine: float32 = the reference's own precision "
                         "(mt3/gin/model.gin:50); bfloat16 is always reported as an extra next to it")
    ap.add_argument("--kv-dtype", default="", choices=["", "fp8_e4m3"],
                    help="K/V cache format: '' = the compute dtype; fp8_e4m3 = OCP e4m3 rows + per-row scales "
                         "(BASELINE configs[4]'s fp8 path, with --dtype bfloat16; NOT the default line)")
    ap.add_argument("--dense-dtype", default="", choices=["", "fp8_e4m3"],
                    help="encoder dense layers + cross-K/V projections: '' = the compute dtype; fp8_e4m3 = MXFP8 on the "
                         "block-scaled MFMA (BASELINE configs[4]'s fp8 MFMA path; NOT the default line)")
    ap.add_argument("--model", default="mt3", choices=["mt3", "base"],
                    help="mt3 = gin/model.gin (configs[1..3]); base = gin/ismir2022/base.gin shape (configs[4])")
    ap.add_argument("--chains", type=int, default=1,
                    help="independent row groups run as parallel branches of the step graph (measured on "
                         "MI355X/ROCm 7.2 at batch 256: 1 -> 758, 2 -> 744 audio-s/s)")
    ap.add_argument("--decoding", default="greedy", choices=["greedy", "beam1"],
                    help="token selection: plain greedy (what BASELINE configs[2] names) or the rule of t5x "
                         "beam_search with one beam (what the reference's InferenceModel runs; ~1 %% slower)")
    ap.add_argument("--corpus", type=int, default=0,
                    help="BASELINE configs[3]: a fixed corpus of this many segments sharded over the ranks (strong "
                         "scaling; 10000 = 1250 per GPU at 8 GPUs); one step = one pass over the corpus")
    ap.add_argument("--corpus-batch", type=int, default=1250, help="segments per engine call in --corpus mode")
    ap.add_argument("--file-segments", type=int, default=256,
                    help="the corpus is a list of files of this many consecutive segments (256 x 2.048 s = 8.7 min of "
                         "audio); host note decoding is sequential inside a file and parallel across files")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other-precision blocks and the stage (frontend/encoder) extras")
    ap.add_argument("--eos-mean", type=float, default=300.0, help="synthetic EOS schedule: mean output length (SURVEY 8d)")
    ap.add_argument("--eos-sd", type=float, default=100.0)
    ap.add_argument("--cpu-segments", type=int, default=8)
    ap.add_argument("--cpu-small-segments", type=int, default=32, help="batch of the CPU batch-scaling probe (0: skip)")
    ap.add_argument("--cpu-enc-segments", type=int, default=64)
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check WITHOUT GPUs (tests/test_bench_dry_run.py): the same rank spawning, shard "
                         "arithmetic, ONE gather per pass (gloo instead of RCCL), host note decoding, max-over-ranks timing "
                         "and JSON assembly, with a stub in place of frontend + engine; the line says dry_run: true and "
                         "its value measures nothing")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-parity-file", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print("CPU_BASELINE " + json.dumps(cpu_baseline(args.cpu_segments, args.decode_steps, args.cpu_enc_segments,
                                                        args.cpu_small_segments, args.cpu_parity_file)), flush=True)
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dry = args.dry_run
    dev = "cpu" if dry else "cuda"
    if dry:
        args.no_extras = args.no_cpu_baseline = True
        if world > 1:
            dist.init_process_group("gloo")
    else:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        if world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from mt3_amd import _lib, distributed, metrics_utils, network, note_sequences, spectrograms, synthetic, vocabularies

    L = 1024
    corpus = args.corpus
    if corpus:
        lo, hi = distributed.shard_range(corpus, rank, world)
        n_local = hi - lo
        B = max(1, min(args.corpus_batch, n_local))
    else:
        lo, n_local, B = rank * args.batch, args.batch, args.batch
    n_global = corpus if corpus else args.batch * world
    import dataclasses
    shape = network.MT3_BASE if args.model == "base" else network.MT3_SMALL
    cfg = dataclasses.replace(shape, dtype=args.dtype, kv_dtype=args.kv_dtype, dense_dtype=args.dense_dtype)
    codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
    vocab = vocabularies.vocabulary_from_codec(codec)
    if dry:
        eng = audio = stream = None
    else:
        eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B, decode_chains=args.chains)
        eng.load_params(network.init_random_params(cfg, seed=0))
        # this rank's shard of the synthetic corpus, resident in HBM before the clock starts
        audio = torch.cat([synthetic.synth_audio(min(1024, n_local - s), seed=1000 + lo + s) for s in range(0, n_local, 1024)])
        stream = torch.cuda.Stream()                                      # a real (capturable) stream
    start_times = [s * SEG_SECONDS - (s * SEG_SECONDS) % 0.01 for s in range(n_global)]

    # rank 0's host stage (EOS trim + run-length / note decoding in libmt3hip.so) runs on worker threads, so the
    # NEXT batch's GPU work is already being launched while the previous batch's tokens become notes; every
    # future is joined before the clock stops, so all of it stays inside the timed region.
    # The corpus is a list of FILES of `--file-segments` consecutive segments (the host state machine is sequential
    # inside a file, mt3/metrics_utils.py:92-116); file boundaries do not depend on the number of ranks, each file is
    # decoded on its own worker thread (the C++ decoder runs outside the GIL), so rank 0's host time per step does
    # not grow with N and the notes are the same for every N (tests/test_distributed_gloo.py).
    gather_events = []

    def notes_of_file(rows, first):
        """EOS trim + run-length / note decoding of one file's token rows (C++ in libmt3hip.so; notes as a record array:
        no Python object per note on the job's host stage)"""
        rec, inv, drop, total = metrics_utils.decode_token_rows(codec, note_sequences.NoteEncodingWithTiesSpec, rows,
                                                                start_times[first: first + len(rows)])
        return len(rec)

    def transcribe(first, count, engine=None):
        """frontend -> encode -> decode -> ids -> tokens for global segments [first, first + count): CUDA int32 [count, L]"""
        if dry:
            return torch.from_numpy(synthetic.stub_token_rows(first, count, L))
        chunk = audio[first - lo: first - lo + count]
        e = engine or eng
        e.encode(spectrograms.compute_spectrogram_batch(chunk, None))
        return vocab.decode_tf(e.decode(num_steps=args.decode_steps, beam1=args.decoding == "beam1"))

    class _HostEvent:                                   # dry run: wall-clock stand-in for a HIP event
        def __init__(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    def on_gather(phase):
        if dry:
            ev = _HostEvent()
        else:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
        if phase == 0:
            gather_events.append([ev, None])
        else:
            gather_events[-1][1] = ev

    job = distributed.ShardedTranscriber(n_global, rank, world, transcribe, notes_of_file, call_segments=B,
                                         file_segments=args.file_segments, host_threads=8, on_gather=on_gather)

    def host_stage(host):
        """one step's token rows of THIS rank's shard alone (the single-GPU extras): futures, one per file"""
        return [job._pool.submit(notes_of_file, host[a:b], a) for a, b in distributed.file_ranges(host.shape[0],
                                                                                                   args.file_segments)]

    def step():
        if dry:
            return job.step()
        with torch.cuda.stream(stream):
            job.step()

    def drain():
        return sum(job.drain())

    def device_sync():
        if not dry:
            torch.cuda.synchronize()

    def sync_all():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    for _ in range(args.warmup):
        step()
    drain()
    gather_events.clear()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    n_notes = drain()
    device_sync()
    dt_own = time.perf_counter() - t0                 # this rank's own time to finish its K steps
    sync_all()
    dt = time.perf_counter() - t0
    per_rank_ms, gather_ms = [dt_own * 1e3 / args.steps], None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        mine = torch.tensor([dt_own * 1e3 / args.steps], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(t.item()) for t in allr]
        gather_ms = sum(a.elapsed_time(b) for a, b in gather_events) / max(1, len(gather_events))
    graph_fallbacks = used_graph = decode_groups = partition_fallbacks = 0
    if not dry:
        graph_fallbacks = eng.status(_lib.STATUS_GRAPH_FALLBACKS)
        used_graph = eng.status(_lib.STATUS_LAST_DECODE_USED_GRAPH)
        decode_groups = eng.status(_lib.STATUS_LAST_DECODE_GROUPS)
        partition_fallbacks = eng.status(_lib.STATUS_PARTITION_FALLBACKS)

    # ---- roofline of the dominant kernel (decode self-attention: HBM streaming of the K/V cache).
    # In-situ and live: HIP events (recorded on the stream the graphs are launched on) around the whole
    # graph-replayed decode, once as it ships and once with that kernel's launches left out of the step
    # graph (mt3_debug_engine_decode, include/mt3_hip_debug.h); the difference / launches = the kernel's average
    # duration inside the real decode loop.
    roof, extras = None, {}
    if rank == 0 and not dry:
        Br = min(B, n_local)

        def timed(fn, reps=1):
            with torch.cuda.stream(stream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    fn()
                e1.record(stream)
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        def pmc_ratio(section, key):
            """HBM traffic / algorithmic bytes of a decode-attention kernel from the PMC counters (rocprofv3 --pmc
            FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 on gfx950 -- calibrated on a 1 GiB copy): a bench
            run cannot wrap itself in rocprofv3, so tools/gpu_pmc.sh collects them at this exact shape and
            tools/pmc_summary.py stamps the summary with the hash of the kernel sources it was collected from; a summary
            from OTHER sources is refused (traffic = null) instead of silently carrying an old ratio over a kernel change."""
            for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_summary.json")),
                               reverse=True):
                try:
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        pmc = json.load(f)
                    if key is None:         # the frontend kernel: its own sources
                        if pmc.get("frontend_source_hash") != kernel_source_hash(FRONTEND_SOURCES) or Br != 256:
                            continue
                        return pmc[section]["traffic_over_algorithmic"], name, pmc["frontend_source_hash"]
                    if pmc.get("kernel_source_hash") != kernel_source_hash() or pmc["shape"]["B"] != Br:
                        continue
                    return pmc[section][key]["traffic_over_algorithmic"], name, pmc["kernel_source_hash"]
                except (OSError, KeyError, ValueError, TypeError):
                    continue
            return None, None, None

        def attention_roofline(engine, ecfg, reps=2):
            """roofline block of `engine`'s decode self-attention kernel (plus the cross-attention figures), measured
            on the engine's current encoded batch of Br rows"""
            def decode_ms(**kw):
                with torch.cuda.stream(stream):
                    engine.debug_decode(num_steps=2, chains=1, **kw)          # capture / warm this graph variant
                # the kernel at full-GPU width, one launch at a time (as rocprofv3 sees it)
                return min(timed(lambda: engine.debug_decode(num_steps=args.decode_steps, chains=1, **kw))
                           for _ in range(reps))
            t_full = decode_ms()

            def product_decode(**kw):
                """(ms, host CPU seconds of the whole process, row groups, graph replay?) of one decode as the product
                schedules it: HIP events for the device side, getrusage (user + system, all threads: the engine's group
                workers included) for the host side"""
                with torch.cuda.stream(stream):
                    engine.decode(num_steps=2, **kw)
                torch.cuda.synchronize()
                best = None
                for _ in range(reps):
                    r0 = resource.getrusage(resource.RUSAGE_SELF)
                    ms = timed(lambda: engine.decode(num_steps=args.decode_steps, **kw))
                    r1 = resource.getrusage(resource.RUSAGE_SELF)
                    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
                    if best is None or ms < best[0]:
                        best = (ms, cpu)
                return best + (engine.status(_lib.STATUS_LAST_DECODE_GROUPS),
                               engine.status(_lib.STATUS_LAST_DECODE_USED_GRAPH))
            t_prod, cpu_prod, prod_groups, prod_graph = product_decode()
            t_direct, cpu_direct, _, _ = product_decode(use_graph=False)
            t_noself = decode_ms(skip_self_attn=True)
            t_nocross = decode_ms(skip_cross_attn=True)
            esz = 2 if ecfg.dtype == "bfloat16" else 4
            H, S, nl = ecfg.num_heads, args.decode_steps, ecfg.num_decoder_layers
            # K+V bytes of one cache position, all rows (fp8: 64 e4m3 bytes each + the {k, v} f32 scale pair of the row)
            kv_row = Br * H * (2.0 * 64 + 8.0) if ecfg.kv_dtype else 2.0 * Br * H * 64 * esz
            launches = S * nl
            # algorithmic bytes: read the t cached K/V rows + the new row + q, write the new row + the output
            self_bytes = nl * sum(kv_row * (t + 1) + kv_row + 2.0 * Br * H * 64 * esz for t in range(S))
            cross_bytes = launches * (kv_row * 256 + 2.0 * Br * H * 64 * esz)
            self_us = (t_full - t_noself) * 1e3 / launches
            cross_us = (t_full - t_nocross) * 1e3 / launches
            ach = self_bytes / launches / (self_us * 1e-6) / 1e9
            kname = ("fp8" if ecfg.kv_dtype else ("bf16" if esz == 2 else "f32"))
            traffic, traffic_src = None, "no PMC summary for these kernel sources / this shape (run tools/gpu_pmc.sh)"
            if args.decode_steps == 1024 and (ecfg.num_heads == 6 or (ecfg.num_heads == 12 and ecfg.kv_dtype)):
                ratio, fname, khash = pmc_ratio("dec_attn_self_append_" + kname + ("_h12" if ecfg.num_heads == 12 else ""),
                                                "n_keys_513")
                if ratio is not None:
                    traffic = ratio * self_bytes / launches
                    traffic_src = "profiles/%s (kernel sources %s; measured traffic/algorithmic = %.4f at the mean " \
                                  "launch)" % (fname, khash, ratio)
            return {"bound": "hbm",
                    "kernel": "mt3k::dec_attn_fp8_kernel<APPEND=true> (decode self-attention over the e4m3 K/V cache)"
                    if ecfg.kv_dtype else "mt3k::dec_attn_kernel<%s, APPEND=true> (decode self-attention over the K/V "
                                          "cache)" % kname,
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_us": self_us, "algorithmic_bytes_per_launch": self_bytes / launches,
                    "launches": launches,
                    "method": "HIP events on the launch stream around the whole SINGLE-STREAM graph-replayed debug decode "
                              "(the kernel at full-GPU width, one launch at a time, as rocprofv3 times it; NOT the "
                              "product's row-group schedule, whose clock is decode_ms_product_schedule), with and "
                              "without this kernel in the step graph; (difference)/launches",
                    "decode_ms_single_chain": t_full, "decode_ms_without_self_attn": t_noself,
                    "decode_ms_without_cross_attn": t_nocross,
                    "small_kernel_us_per_step": (t_noself + t_nocross - t_full) * 1e3 / S,
                    "whole_step_hbm_frac": (self_bytes + cross_bytes) / (t_full * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    # the schedule the headline runs (row groups at this batch): same bytes, its own clock
                    "decode_ms_product_schedule": t_prod, "product_schedule_row_groups": prod_groups,
                    "product_schedule_graph_replay": bool(prod_graph),
                    "host_cpu_s_per_decode": cpu_prod,
                    "decode_ms_direct_launches": t_direct, "host_cpu_s_per_decode_direct_launches": cpu_direct,
                    "whole_step_hbm_frac_product_schedule": (self_bytes + cross_bytes) / (t_prod * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "cross_attn": {"achieved": cross_bytes / launches / (cross_us * 1e-6) / 1e9,
                                   "frac": cross_bytes / launches / (cross_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   "avg_launch_us": cross_us, "algorithmic_bytes_per_launch": cross_bytes / launches}}

        with torch.cuda.stream(stream):
            eng.encode(spectrograms.compute_spectrogram_batch(audio[:Br], None))
        roof = attention_roofline(eng, cfg)

        main_line = not corpus and args.model == "mt3" and not args.kv_dtype and not args.dense_dtype and world == 1
        if not args.no_extras and main_line:
            # ---- stage extras, driver-timed (HIP events on the launch stream, inputs in HBM)
            esize = 2 if args.dtype == "bfloat16" else 4
            peak = MFMA_BF16_PEAK_TFLOPS if esize == 2 else MFMA_F32_PEAK_TFLOPS
            a256 = audio[:Br]
            lm256 = spectrograms.compute_spectrogram_batch(a256, None)
            fe_ms = timed(lambda: spectrograms.compute_spectrogram_batch(a256, None), reps=20)
            enc_ms = min(timed(lambda: eng.encode(lm256), reps=5) for _ in range(2))
            fe_gbs = FRONTEND_BYTES_PER_SEGMENT * Br / (fe_ms * 1e-3) / 1e9
            fe_ratio, fe_name, fe_hash = pmc_ratio("logmel_kernel_256_segments", None)
            extras["frontend"] = {"kernel": "mt3::logmel_kernel", "segments": Br, "ms": fe_ms, "bound": "hbm",
                                  "achieved": fe_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fe_gbs / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_segment": FRONTEND_BYTES_PER_SEGMENT,
                                  "traffic": fe_ratio * FRONTEND_BYTES_PER_SEGMENT * Br if fe_ratio else None,
                                  "traffic_source": "profiles/%s (frontend sources %s)" % (fe_name, fe_hash)
                                  if fe_ratio else "no PMC summary for these frontend sources"}
            enc_tf = ENC_GFLOP_PER_SEGMENT * Br / (enc_ms * 1e-3) / 1e3

            def mfma_block(tf):
                """roofline fields of an encoder figure.  The f32 engine's encoder multiplies on the bf16 pipes -- dense layers
                AND attention (round 4): every f32 operand as three exact bf16 terms, six bf16 products per f32 product
                (csrc/gemm.hip: gemm_x6_kernel, csrc/enc_attention_x6.hip) -- so its matrix-instruction work is 6x the
                algorithmic flops and is priced against the BF16 peak; against the f32 instruction's peak the same figure
                would read above 1"""
                if esize == 2:
                    return {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak}
                return {"bound": "mfma", "achieved": 6.0 * tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": 6.0 * tf / MFMA_BF16_PEAK_TFLOPS, "f32_equivalent_tflops": tf,
                        "f32_mfma_peak": MFMA_F32_PEAK_TFLOPS,
                        "note": "achieved = bf16 matrix-instruction work of the whole encoder, dense layers and attention "
                                "(6 bf16 MFMAs per f32 product: three exact bf16 planes per operand), over the whole time; "
                                "f32_equivalent_tflops = algorithmic flops / time"}
            extras["encoder"] = {"segments": Br, "ms": enc_ms, **mfma_block(enc_tf),
                                 "algorithmic_gflop_per_segment": ENC_GFLOP_PER_SEGMENT}
            # BASELINE configs[1]: batch 64, log-mel + encoder (+ cross-K/V), encoder-only throughput
            n1 = min(64, Br)
            a64 = audio[:n1]

            def c1():
                eng.encode(spectrograms.compute_spectrogram_batch(a64, None))
            c1()
            c1_ms = min(timed(c1, reps=10) for _ in range(2))
            c1_tf = ENC_GFLOP_PER_SEGMENT * n1 / (c1_ms * 1e-3) / 1e3
            extras["configs1"] = {"workload": "BASELINE configs[1]: batch=%d synthetic segments, log-mel + encoder "
                                              "(+ cross-K/V projections), no decode" % n1,
                                  "segments_per_s": n1 / (c1_ms * 1e-3), "audio_s_per_s": n1 * SEG_SECONDS / (c1_ms * 1e-3),
                                  "ms": c1_ms, **mfma_block(c1_tf)}
            with torch.cuda.stream(stream):
                eng.encode(lm256)                                    # leave the engine at the bench batch

            free_running = {}          # key -> the token rows [Br, L] of that engine's free-running decode of a256

            def other_engine(key, ecfg, label, n_steps, with_roofline, inputs=None):
                """the SAME pipeline as the headline on another engine configuration: one warm-up step, then `n_steps`
                timed steps (wall clock, synchronised both sides, host note decoding inside).  inputs: (audio, true
                frame counts, [(first, count) per file]) instead of the headline's fixed-length segments"""
                try:
                    e2 = network.Transformer(ecfg, input_length=256, max_decode_length=L, max_batch=Br)
                    e2.load_params(network.init_random_params(ecfg, seed=0))
                    aud, nfr, file_list = inputs if inputs is not None else (a256, None, None)
                    lm_in = spectrograms.compute_spectrogram_batch(aud, nfr)

                    def one_step():
                        with torch.cuda.stream(stream):
                            e2.encode(spectrograms.compute_spectrogram_batch(aud, nfr))
                            ids = e2.decode(num_steps=args.decode_steps)
                            host = vocab.decode_tf(ids).cpu().numpy()
                        free_running[key] = host
                        if file_list is not None:     # one host-stage job per FILE of the ragged corpus
                            return [job._pool.submit(notes_of_file, host[a:a + n], a) for a, n in file_list]
                        return host_stage(host)
                    for f in one_step():
                        f.result()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    futs = []
                    for _ in range(n_steps):
                        futs += one_step()
                    for f in futs:
                        f.result()
                    torch.cuda.synchronize()
                    d = (time.perf_counter() - t1) / n_steps
                    e_ms = min(timed(lambda: e2.encode(lm_in), reps=5) for _ in range(2))
                    rec = {"value": Br * SEG_SECONDS / d, "unit": "audio-s/s", "ms_per_step": d * 1e3, "steps": n_steps,
                           "warmup": 1, "dtype": ("bf16" if ecfg.dtype == "bfloat16" else "f32") +
                           (" compute + fp8 (e4m3) K/V caches" if ecfg.kv_dtype else "") +
                           (" + MXFP8 encoder dense layers" if ecfg.dense_dtype else ""),
                           "workload": "%s, batch=%d, %d greedy steps, same pipeline as the headline"
                                       % (label, Br, args.decode_steps),
                           "encoder_ms": e_ms, "device_bytes": e2.device_bytes,
                           "graph_fallbacks": e2.status(_lib.STATUS_GRAPH_FALLBACKS)}
                    if with_roofline:
                        rec["roofline"] = attention_roofline(e2, ecfg, reps=1)
                    extras[key] = rec
                    del e2
                    return rec
                except Exception as ex:                              # the headline must not die with an extra
                    extras[key] = {"value": None, "error": repr(ex)[:300]}
                    return extras[key]

            # ---- the same workload in the OTHER operand precision (>= 5 timed steps, own roofline block): the headline is
            # float32 = the reference's own (gin/model.gin:50 `dtype = 'float32'`; f32 MFMA operands, f32 K/V caches,
            # token-exact against the oracle: tests/test_gpu_parity_deep.py), so this block is the bf16 engine -- or the
            # f32 one when the line was asked for with --dtype bfloat16
            other = "bfloat16" if args.dtype == "float32" else "float32"
            okey = "bf16" if other == "bfloat16" else "f32"
            r_other = other_engine(okey, network.T5Config(dtype=other), "MT3 (model.gin) shape, %s operands" % okey,
                              max(5, min(args.steps, 8)), True)
            if r_other.get("value"):
                r_other["note"] = ("bf16 MFMA operands and K/V caches, f32 accumulation / residual / softmax: NOT the reference's "
                              "precision -- see extra.divergence_vs_f32 for how far its free-running tokens drift"
                              if other == "bfloat16" else
                              "reference precision (model.gin:50 dtype float32): f32 operands and K/V caches; the decode step on "
                              "v_mfma_f32_16x16x4_f32, the encoder on the bf16 pipes with three exact bf16 planes per operand; "
                              "token-exact vs the oracle")
            # ---- BASELINE configs[4] ingredients, one warm-up + 3 timed steps each (never the headline):
            #   fp8_kv    : this workload with e4m3 K/V caches (half the bytes of the HBM-bound decode stream)
            #   fp8_kv_mx8: the same plus the encoder's dense layers / cross-K/V projections as MXFP8 on the scaled MFMA
            #   configs4  : the ismir2022/base.gin shape (emb 768, 12 heads, 12+12 layers, mlp 2048) with both
            # configs[4] runs on "Slakh-shaped" audio: six tones in every segment, files of 1 .. 8 segments
            # whose last segment is ragged (true frame counts go to the frontend, one host-stage job per file)
            slakh = synthetic.synth_slakh_shaped(Br, seed=4)
            for key, shp, dense, label, wr, inp in (
                    ("fp8_kv", network.MT3_SMALL, "", "MT3 (model.gin) shape", True, None),
                    ("fp8_kv_mx8", network.MT3_SMALL, "fp8_e4m3", "MT3 (model.gin) shape", False, None),
                    ("configs4", network.MT3_BASE, "fp8_e4m3", "BASELINE configs[4]: ismir2022/base.gin shape, Slakh-shaped "
                     "synthetic audio (6 tones, %d files of 1-8 segments, ragged last segments)" % len(slakh[2]), True, slakh)):
                c8 = dataclasses.replace(shp, dtype="bfloat16", kv_dtype="fp8_e4m3", dense_dtype=dense)
                rec = other_engine(key, c8, label, 3, wr, inp)
                if key == "fp8_kv_mx8" and rec.get("value"):
                    tf8 = ENC_GFLOP_PER_SEGMENT * Br / (rec["encoder_ms"] * 1e-3) / 1e3
                    extras["encoder_mx8"] = {"segments": Br, "ms": rec["encoder_ms"], "bound": "mfma", "achieved": tf8,
                                             "peak": MFMA_FP8_PEAK_TFLOPS, "unit": "TFLOP/s",
                                             "frac": tf8 / MFMA_FP8_PEAK_TFLOPS,
                                             "note": "encoder + cross-K/V projections with MXFP8 operands "
                                                     "(v_mfma_scale_f32_16x16x128_f8f6f4); attention stays bf16"}
            # ---- how far the reduced-precision engines' FREE-RUNNING decodes drift from the f32 engine's on the
            # same 256 segments x 1024 steps (random-init weights: one flipped arg-max re-rolls the rest of a row,
            # so this is an upper bound on what trained, peaked distributions would show); VERDICT r2 #1c
            try:
                from mt3_amd import metrics
                with torch.cuda.stream(stream):
                    free_running["f32" if args.dtype == "float32" else "bf16"] = transcribe(lo, Br).cpu().numpy()
                if "f32" in free_running:
                    extras["divergence_vs_f32"] = {
                        k: metrics.token_stream_divergence(free_running["f32"], free_running[k], codec)
                        for k in ("bf16", "fp8_kv", "fp8_kv_mx8") if k in free_running}
                    extras["divergence_vs_f32"]["note"] = (
                        "top-level fields of a mode: free-running greedy decode of the same %d segments x %d steps, reference = "
                        "this repo's f32 engine (token-exact vs the oracle); random-init weights (next to no notes: see "
                        "<mode>.trained / <mode>.boosted for the note-level figures)" % (Br, args.decode_steps))
            except Exception as ex:
                extras["divergence_vs_f32"] = {"error": repr(ex)[:300]}

            # ---- NOTE-LEVEL tolerance of the reduced-precision engines (north_star: "decoded note onsets/offsets
            # within a stated fp tolerance"; SURVEY.md 8(d): mir_eval's rule, mt3/metrics.py:255-290, F1 >= 0.99 bf16 / >= 0.97
            # fp8).  One file through InferenceModel per engine configuration, every mode scored against the f32 engine's notes
            # (token-exact vs the oracle: cpu_baseline.parity, tests/test_gpu_note_tolerance.py):
            #   trained : tests/golden/mt3_synthetic_ckpt.npz -- this network TRAINED on synthetic music (tools/train_synthetic.py,
            #             14 minutes on one MI355X): peaked distributions conditioned on the audio, and a piece whose ground-truth
            #             notes are known (`vs_truth` = transcription accuracy of each engine)
            #   boosted : random-init weights with boosted note-event logits -- tens of thousands of notes but FLAT distributions (a
            #             flipped arg-max re-rolls the rest of a row): a worst case, published as measured; also at the
            #             ismir2022/base.gin shape for the configuration `configs4` times
            def note_tolerance():
                from mt3_amd import checkpoints, evaluation
                dv = extras.setdefault("divergence_vs_f32", {})
                if not isinstance(dv, dict) or "error" in dv:
                    dv = extras["divergence_vs_f32"] = {}

                def spread(rep, label, modes=evaluation.REDUCED_MODES):
                    dv.setdefault("f32_reference", {})[label] = rep.get("f32")
                    for k in modes:
                        if k in rep:
                            dv.setdefault(k, {})[label] = rep[k]
                t1 = time.perf_counter()
                try:
                    ck = os.path.join(ROOT, "tests", "golden", "mt3_synthetic_ckpt.npz")
                    truth, wav = synthetic.synth_music(600.0, seed=77)
                    spread(evaluation.compare_engines(checkpoints.load_compact_npz(ck), wav, network.MT3_SMALL, truth=truth), "trained")
                    dv["trained_checkpoint"] = checkpoints.compact_npz_meta(ck)
                except Exception as ex:
                    dv["trained_error"] = repr(ex)[:300]
                try:
                    wav = synthetic.synth_audio(293, seed=77, tones=6).reshape(-1)[: 600 * 16000].cpu().numpy()
                    prm = synthetic.boost_note_events(network.init_random_params(network.MT3_SMALL, seed=0), eos=4.0)
                    spread(evaluation.compare_engines(prm, wav, network.MT3_SMALL), "boosted")
                    prm = synthetic.boost_note_events(network.init_random_params(network.MT3_BASE, seed=0), eos=4.0)
                    rep = evaluation.compare_engines(prm, wav[: 180 * 16000], network.MT3_BASE, modes=("fp8_kv_mx8",))
                    dv["configs4"] = {"boosted": rep.get("fp8_kv_mx8"), "f32_reference": rep.get("f32"),
                                      "what": "ismir2022/base.gin shape, e4m3 caches + MXFP8 encoder (what extra.configs4 times) "
                                              "against the f32 engine of the same shape, 3-minute file"}
                except Exception as ex:
                    dv["boosted_error"] = repr(ex)[:300]
                dv["note_level_what"] = (
                    "<mode>.trained / <mode>.boosted: one 10-minute file through InferenceModel (beam-1, early exit) per engine "
                    "configuration, notes scored against the f32 engine's notes of the same file with the reference's mir_eval rule "
                    "(onset +-50 ms; offset max(50 ms, 20 %)); pitch as note numbers (the reference's call) and exact (hz); trained "
                    "= tests/golden/mt3_synthetic_ckpt.npz (vs_truth = against the notes the audio was rendered from), boosted = "
                    "random-init weights with boosted note-event logits (flat distributions: worst case)")
                dv["note_level_wall_s"] = time.perf_counter() - t1
            try:
                with torch.cuda.stream(stream):
                    note_tolerance()
            except Exception as ex:
                extras.setdefault("divergence_vs_f32", {})["note_level_error"] = repr(ex)[:300]

            # ---- SURVEY.md 8(d), second figure: a SYNTHETIC EOS SCHEDULE -- output lengths ~ clipped N(300, 100) imposed
            # through the bench-only hook of include/mt3_hip_debug.h (row r's distribution at step len[r] - 1 becomes a
            # point mass on EOS; everything before is the model's own arithmetic), decode with EARLY EXIT = finished rows
            # are retired (no K/V streamed for them, live rows compacted at the 32-step poll).  Same pipeline as the
            # headline; real transcriptions are a few hundred tokens of the 1024, so this is the cost in use, the
            # full-length headline the upper bound.  LABELLED SYNTHETIC: the lengths are imposed, not predicted.
            def eos_schedule(engine, ecfg, key, n_steps=3):
                try:
                    rng = np.random.default_rng(0)
                    lens = np.clip(np.rint(rng.normal(args.eos_mean, args.eos_sd, Br)), 1, args.decode_steps).astype(np.int32)
                    engine.debug_set_eos_schedule(lens)
                    stats = {}

                    def one_step(**kw):
                        with torch.cuda.stream(stream):
                            engine.encode(spectrograms.compute_spectrogram_batch(a256, None))
                            ids = engine.decode(num_steps=args.decode_steps, early_exit=True, **kw)
                            stats["steps_run"] = engine.steps_run
                            stats["compactions"] = engine.status(_lib.STATUS_LAST_DECODE_COMPACTIONS)
                            stats["groups"] = engine.status(_lib.STATUS_LAST_DECODE_GROUPS)
                            host = vocab.decode_tf(ids).cpu().numpy()
                        return host, host_stage(host)

                    def timed_steps(**kw):
                        host, futs = one_step(**kw)
                        for f in futs:
                            f.result()
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        futs = []
                        for _ in range(n_steps):
                            futs += one_step(**kw)[1]
                        for f in futs:
                            f.result()
                        torch.cuda.synchronize()
                        return (time.perf_counter() - t1) / n_steps, host

                    def decode_only(**kw):
                        with torch.cuda.stream(stream):
                            engine.decode(num_steps=args.decode_steps, early_exit=True, **kw)
                        torch.cuda.synchronize()
                        r0 = resource.getrusage(resource.RUSAGE_SELF)
                        t1 = time.perf_counter()
                        with torch.cuda.stream(stream):
                            engine.decode(num_steps=args.decode_steps, early_exit=True, **kw)
                        torch.cuda.synchronize()
                        dt1 = time.perf_counter() - t1
                        r1 = resource.getrusage(resource.RUSAGE_SELF)
                        return dt1 * 1e3, (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
                    d, host = timed_steps()
                    got_len = np.where((host == -1).any(1), (host == -1).argmax(1) + 1, host.shape[1])
                    dec_ms, dec_cpu = decode_only()
                    dec_ms_direct, dec_cpu_direct = decode_only(use_graph=False)
                    dec_ms_single, _ = decode_only(single_stream=True)
                    esz = 2 if ecfg.dtype == "bfloat16" else 4
                    H, nl = ecfg.num_heads, ecfg.num_decoder_layers
                    kv1 = H * (2.0 * 64 + 8.0) if ecfg.kv_dtype else 2.0 * H * 64 * esz        # K + V bytes of one cached position of one row
                    qo = 2.0 * H * 64 * esz
                    ll = lens.astype(np.float64)
                    # a row of length n runs steps t = 0 .. n - 1: reads its t cached positions + appends one, reads the
                    # 256 cross positions; bytes the LIVE rows need (retired rows need none)
                    self_b = nl * float((kv1 * (ll * (ll + 1) / 2) + (kv1 + qo) * ll).sum())
                    cross_b = nl * float(((kv1 * 256 + qo) * ll).sum())
                    full_b = nl * Br * (kv1 * args.decode_steps * (args.decode_steps + 1) / 2 +
                                        (kv1 + qo) * args.decode_steps + (kv1 * 256 + qo) * args.decode_steps)
                    extras.setdefault("eos_schedule", {})[key] = {
                        "value": Br * SEG_SECONDS / d, "unit": "audio-s/s", "ms_per_step": d * 1e3, "steps": n_steps, "warmup": 1,
                        "dtype": "bf16" if ecfg.dtype == "bfloat16" else "f32",
                        "workload": "SYNTHETIC EOS SCHEDULE (SURVEY.md 8(d)): batch=%d, output lengths ~ clipped "
                                    "N(%g, %g) imposed per row (seed 0), greedy decode with early exit + row retirement, "
                                    "same pipeline as the headline (frontend, encoder, ids->tokens, host note decoding)"
                                    % (Br, args.eos_mean, args.eos_sd),
                        "lengths": {"mean": float(ll.mean()), "min": int(lens.min()), "max": int(lens.max()),
                                    "decoded_mean": float(got_len.mean()),
                                    "decoded_not_longer_than_imposed_frac": float((got_len <= lens).mean())},
                        "decode_ms": dec_ms, "decode_steps_run": stats["steps_run"], "row_groups": stats["groups"],
                        "compactions_per_decode": stats["compactions"],
                        "host_cpu_s_per_decode": dec_cpu,
                        "decode_ms_direct_launches": dec_ms_direct, "host_cpu_s_per_decode_direct_launches": dec_cpu_direct,
                        "decode_ms_single_stream": dec_ms_single,
                        "live_row_kv_bytes_per_decode": self_b + cross_b,
                        "live_bytes_over_full_length_bytes": (self_b + cross_b) / full_b,
                        "whole_decode_hbm_frac_on_live_bytes": (self_b + cross_b) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                except Exception as ex:
                    extras.setdefault("eos_schedule", {})[key] = {"value": None, "error": repr(ex)[:300]}
                finally:
                    try:
                        engine.debug_set_eos_schedule(None)
                    except Exception:
                        pass

            eos_schedule(eng, cfg, "f32" if args.dtype == "float32" else "bf16")
            try:
                ocfg = network.T5Config(dtype=other)
                e3 = network.Transformer(ocfg, input_length=256, max_decode_length=L, max_batch=Br)
                e3.load_params(network.init_random_params(ocfg, seed=0))
                eos_schedule(e3, ocfg, okey)
                del e3
            except Exception as ex:
                extras.setdefault("eos_schedule", {})[okey] = {"value": None, "error": repr(ex)[:300]}

            # ---- the same synthetic EOS schedule on BASELINE configs[3]'s corpus: 10,000 independent
            # segments through 1250 decode slots, (a) as the reference's loop has it -- one batch-synchronous engine call per
            # 1250 segments (NB:295-301; early exit + row retirement inside a call) -- and (b) with IN-FLIGHT BATCHING
            # (mt3_engine_transcribe): a finished slot restarts on the next encoded segment.  Same tokens either way.
            def eos_schedule_corpus(ecfg, key, n_corpus=10000, slots=1250):
                try:
                    ce = network.Transformer(ecfg, input_length=256, max_decode_length=L, max_batch=slots)
                    ce.load_params(network.init_random_params(ecfg, seed=0))
                    aud = torch.cat([synthetic.synth_audio(min(1024, n_corpus - a), seed=1000 + a)
                                     for a in range(0, n_corpus, 1024)])
                    lens = np.clip(np.rint(np.random.default_rng(0).normal(args.eos_mean, args.eos_sd, n_corpus)), 1,
                                   args.decode_steps).astype(np.int32)

                    def run_batch(n):
                        toks = []
                        with torch.cuda.stream(stream):
                            for a in range(0, n, slots):
                                ce.debug_set_eos_schedule(lens[a:a + slots])
                                ce.encode(spectrograms.compute_spectrogram_batch(aud[a:min(a + slots, n)], None))
                                toks.append(vocab.decode_tf(ce.decode(num_steps=args.decode_steps, early_exit=True)))
                            return torch.cat(toks).cpu().numpy()

                    def run_refill(n):
                        with torch.cuda.stream(stream):
                            ce.debug_set_eos_schedule(lens[:n])
                            lm = torch.empty((n, 256, 512), device="cuda", dtype=torch.float32)
                            for a in range(0, n, 1024):
                                lm[a:a + 1024] = spectrograms.compute_spectrogram_batch(aud[a:min(a + 1024, n)], None)
                            return vocab.decode_tf(ce.transcribe(lm, num_steps=args.decode_steps)).cpu().numpy()

                    def clocked(fn):
                        fn(min(n_corpus, 2 * slots))                 # warm-up on a fifth of the corpus: graphs, staging ring
                        torch.cuda.synchronize()
                        r0 = resource.getrusage(resource.RUSAGE_SELF)
                        t1 = time.perf_counter()
                        host = fn(n_corpus)
                        torch.cuda.synchronize()
                        d = time.perf_counter() - t1
                        r1 = resource.getrusage(resource.RUSAGE_SELF)
                        return d, (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime), host
                    d_b, cpu_b, host_b = clocked(run_batch)
                    d_r, cpu_r, host_r = clocked(run_refill)
                    st = dict(ce.transcribe_stats)
                    esz = 2 if ecfg.dtype == "bfloat16" else 4
                    H, nl = ecfg.num_heads, ecfg.num_decoder_layers
                    kv1, qo = 2.0 * H * 64 * esz, 2.0 * H * 64 * esz
                    ll = lens.astype(np.float64)
                    live = nl * float((kv1 * (ll * (ll + 1) / 2) + (kv1 + qo) * ll).sum() + ((kv1 * 256 + qo) * ll).sum())
                    extras.setdefault("eos_schedule_corpus", {})[key] = {
                        "value": n_corpus * SEG_SECONDS / d_r, "unit": "audio-s/s", "seconds_per_pass": d_r, "steps": 1, "warmup": 1,
                        "dtype": "bf16" if ecfg.dtype == "bfloat16" else "f32",
                        "workload": "SYNTHETIC EOS SCHEDULE on BASELINE configs[3]'s corpus: %d segments, output lengths ~ clipped "
                                    "N(%g, %g) per segment (seed 0), %d decode slots with IN-FLIGHT BATCHING "
                                    "(mt3_engine_transcribe: finished slots restart on the next encoded segment); log-mel of "
                                    "every segment, encoder passes, greedy decode, ids->tokens and the host copy inside the clock"
                                    % (n_corpus, args.eos_mean, args.eos_sd, slots),
                        "transcribe_stats": st, "host_cpu_s_per_pass": cpu_r,
                        "live_row_kv_bytes_per_pass": live,
                        "whole_pass_hbm_frac_on_live_bytes": live / d_r / 1e9 / HBM_PEAK_GBS,
                        "batch_synchronous": {"value": n_corpus * SEG_SECONDS / d_b, "seconds_per_pass": d_b,
                                              "host_cpu_s_per_pass": cpu_b,
                                              "whole_pass_hbm_frac_on_live_bytes": live / d_b / 1e9 / HBM_PEAK_GBS,
                                              "what": "the reference's shape of loop: one engine call per %d segments, each "
                                                      "until its longest row is done (early exit + row retirement)" % slots},
                        "refill_over_batch_synchronous": d_b / d_r,
                        "tokens_identical": bool(np.array_equal(host_b, host_r))}
                    ce.debug_set_eos_schedule(None)
                    del ce
                except Exception as ex:
                    extras.setdefault("eos_schedule_corpus", {})[key] = {"value": None, "error": repr(ex)[:300]}

            eos_schedule_corpus(cfg, "f32" if args.dtype == "float32" else "bf16")

            # ---- BASELINE configs[0]'s analogue: ONE file through the drop-in class, InferenceModel.__call__
            # (NB:283-308), reference precision.  A 10-minute "Slakh-shaped" file (six tones per segment, ragged last segment);
            # random-init weights whose logits favour note events and EOS (synthetic.boost_note_events: rows end of their own
            # accord, no imposed lengths).  (a) as shipped: the engine sized to the file, one refilled engine call;
            # (b) schedule="batch": the reference's literal loop of batch-synchronous 8-row calls (NB:190,295-301).
            def single_file(minutes=10.0):
                try:
                    from mt3_amd import inference
                    n_seg = int(math.ceil(minutes * 60.0 / SEG_SECONDS))
                    wav = synthetic.synth_audio(n_seg, seed=77, tones=6).reshape(-1)[: int(minutes * 60.0 * 16000)].cpu().numpy()
                    icfg = network.T5Config(dtype=args.dtype)
                    prm = synthetic.boost_note_events(network.init_random_params(icfg, seed=0), eos=4.0)
                    rec = {}
                    notes = {}
                    for key, kw in (("file_sized_engine_refill", {}), ("reference_batches_of_8", {"schedule": "batch"})):
                        m = inference.InferenceModel(prm, "mt3", dtype=args.dtype, decoding="beam1", **kw)
                        with torch.cuda.stream(stream):
                            # warm-up: engine growth and graphs (the refilled call on the whole file: its engine is sized to
                            # it; the 8-row loop on the first 16 segments: its 8-row graphs are the same for every batch)
                            m(wav if not kw else wav[: 16 * 32768])
                            torch.cuda.synchronize()
                            t1 = time.perf_counter()
                            ns = m(wav)
                            torch.cuda.synchronize()
                        d = time.perf_counter() - t1
                        notes[key] = [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum) for n in ns.notes]
                        rec[key] = {"wall_s": d, "audio_s_per_s": len(wav) / 16000.0 / d, "engine_slots": m.engine_slots,
                                    "rows_per_engine_call": m.rows_per_engine_call[:4] + (["..."] if len(m.rows_per_engine_call) > 4 else []),
                                    "engine_calls": len(m.rows_per_engine_call), "batch_size_attribute": m.batch_size,
                                    "notes": len(ns.notes)}
                        del m
                    rec["speedup"] = rec["reference_batches_of_8"]["wall_s"] / rec["file_sized_engine_refill"]["wall_s"]
                    rec["notes_identical"] = notes["file_sized_engine_refill"] == notes["reference_batches_of_8"]
                    rec["workload"] = ("one %.0f-minute 16 kHz file (%d segments, six tones each, ragged last segment) through "
                                       "InferenceModel.__call__: host framing, log-mel, encoder, beam-1 decode with early exit, ids -> "
                                       "tokens, note decoding; %s; random-init weights with boosted note-event / EOS logits (rows "
                                       "end of their own accord); the CPU oracle's rate on this path is cpu_baseline.value"
                                       % (minutes, n_seg, "f32" if args.dtype == "float32" else "bf16"))
                    extras["single_file"] = rec
                except Exception as ex:
                    extras["single_file"] = {"error": repr(ex)[:300]}

            single_file()

            # ---- "also report a pure uniform(-1, 1) noise run, which has no empty-energy bins": the
            # headline pipeline on white-noise segments (3 timed steps after a warm-up)
            try:
                gen = torch.Generator(device="cuda").manual_seed(0)
                noise = (torch.rand((Br, a256.shape[1]), device="cuda", generator=gen) * 2.0 - 1.0).to(torch.float32)

                def noise_step():
                    with torch.cuda.stream(stream):
                        lmn = spectrograms.compute_spectrogram_batch(noise, None)
                        eng.encode(lmn)
                        host = vocab.decode_tf(eng.decode(num_steps=args.decode_steps)).cpu().numpy()
                    return lmn, host_stage(host)
                lmn, futs = noise_step()
                for f in futs:
                    f.result()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                futs = []
                for _ in range(3):
                    futs += noise_step()[1]
                for f in futs:
                    f.result()
                torch.cuda.synchronize()
                dn = (time.perf_counter() - t1) / 3
                extras["uniform_noise"] = {
                    "value": Br * SEG_SECONDS / dn, "unit": "audio-s/s", "ms_per_step": dn * 1e3, "steps": 3, "warmup": 1,
                    "dtype": "f32" if args.dtype == "float32" else "bf16",
                    "workload": "uniform(-1, 1) white-noise segments (no empty-energy mel bins), batch=%d, otherwise the "
                                "headline pipeline" % Br,
                    "logmel_min": float(lmn.min()), "logmel_at_floor_frac": float((lmn <= math.log(1e-5) + 1e-6).float().mean())}
            except Exception as ex:
                extras["uniform_noise"] = {"value": None, "error": repr(ex)[:300]}
            with torch.cuda.stream(stream):
                eng.encode(lm256)                                    # leave the engine at the bench batch

    if rank == 0:
        segs = n_global * args.steps
        value = segs * SEG_SECONDS / dt
        if corpus:
            workload = ("BASELINE configs[3]: MT3 (model.gin) random-init, %d-segment synthetic corpus sharded over "
                        "%d GPU(s) (%d segments on rank 0, %d per engine call), full encoder-decoder %s decode, %d "
                        "decode steps (no early exit), one RCCL all-gather of the token rows per pass, host note "
                        "decoding included" % (corpus, world, n_local, B, args.decoding, args.decode_steps))
        else:
            head = ("BASELINE configs[2]: MT3 (model.gin) random-init" if args.model == "mt3" else
                    "BASELINE configs[4] shape: ismir2022/base.gin random-init")
            workload = ("%s, full encoder-decoder %s decode, batch=%d synthetic 2.048 s segments per GPU, %d decode "
                        "steps (no early exit), ids->tokens + host note decoding included%s"
                        % (head, "greedy" if args.decoding == "greedy" else "beam-1 (t5x beam_search, one beam)",
                           B, args.decode_steps, ", e4m3 K/V caches" if args.kv_dtype else ""))
        out = {
            "metric": "audio-seconds transcribed/sec (whole node), MT3-base, 1/2/4/8 MI355X",
            "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong" if corpus else "weak",
            "vs_baseline": None,
            "dtype": ("bf16" if args.dtype == "bfloat16" else "f32") + ("+fp8kv" if args.kv_dtype else "") +
                     ("+mxfp8 encoder" if args.dense_dtype else ""),
            "data": "synthetic" if not dry else "stub token rows (DRY RUN: no GPU work, the value measures nothing)",
            "dry_run": bool(dry),
            "config": {"workload": workload,
                       "segments_per_gpu": n_local, "segments_total": n_global, "decode_steps": args.decode_steps,
                       "segment_seconds": SEG_SECONDS, "decode_chains": args.chains, "decoding": args.decoding,
                       "file_segments": args.file_segments,
                       "parallelism": "dp%d (segments sharded, weights replicated, ONE RCCL gather of the token rows to rank 0)"
                                      % world if world > 1 else "single GPU",
                       "decode_schedule": ("%d row groups, each on a stream with its own hardware queue and driven by its own "
                                           "engine worker thread, %s" % (decode_groups, "one captured hipGraph per group, "
                                           "replayed per step" if used_graph else "direct launches")) if decode_groups > 1 else
                       ("one stream, hipGraph replay per step" if used_graph else
                        "one stream, DIRECT LAUNCHES (graph capture failed)"),
                       "graph_fallbacks": graph_fallbacks, "partition_fallbacks": partition_fallbacks,
                       "notes_decoded_last_step": n_notes},
            "segments_per_s": segs / dt,
            "rccl_world": dist.get_world_size() if world > 1 else 1,
            "per_rank_ms_per_step": per_rank_ms, "gather_ms": gather_ms,
            # the other operand precision of the same workload (float32 = the reference's; whichever is not the headline)
            "f32_value": value if args.dtype == "float32" else extras.get("f32", {}).get("value"),
            "bf16_value": value if args.dtype == "bfloat16" else extras.get("bf16", {}).get("value"),
            "eos_schedule_value": (extras.get("eos_schedule", {}).get("f32" if args.dtype == "float32" else "bf16") or {}).get("value"),
            "roofline": roof,
            "extra": extras,
        }
        if not args.no_cpu_baseline and world == 1:
            parity_args = []
            if args.dtype == "float32" and args.model == "mt3" and not args.kv_dtype and not args.dense_dtype and not corpus:
                # the rows the oracle will be compared on: the first cpu_segments rows of the headline batch, decoded once
                # more by the headline engine in the headline's schedule inside the full batch
                try:
                    import tempfile
                    n8 = min(args.cpu_segments, B)
                    rows8 = parity_rows(B, n8, eng.status(_lib.STATUS_LAST_DECODE_GROUPS))
                    sel = torch.as_tensor(rows8, device="cuda", dtype=torch.long)
                    with torch.cuda.stream(stream):
                        lm_all = spectrograms.compute_spectrogram_batch(audio[:B], None)
                        eng.encode(lm_all)
                        ids_all = eng.decode(num_steps=args.decode_steps, beam1=False)
                        tok8 = vocab.decode_tf(ids_all[sel]).cpu().numpy()
                    eos8 = tok8 == vocabularies.DECODED_EOS_ID
                    n_tok = np.where(eos8.any(1), eos8.argmax(1), tok8.shape[1])
                    ns8, _, _ = metrics_utils._run(codec, note_sequences.NoteEncodingWithTiesSpec.spec_id,
                                                   [r[:n] for r, n in zip(tok8, n_tok)], start_times[:n8])
                    notes8 = np.array([[n.start_time, n.end_time, n.pitch, n.velocity, n.program, float(n.is_drum)]
                                       for n in ns8.notes], np.float64).reshape(-1, 6)
                    sched = "%d row groups, %s" % (eng.status(_lib.STATUS_LAST_DECODE_GROUPS),
                                                   "graph replay" if eng.status(_lib.STATUS_LAST_DECODE_USED_GRAPH) else "direct launches")
                    pf = os.path.join(tempfile.mkdtemp(prefix="mt3_parity_"), "rows.npz")
                    np.savez(pf, audio=audio[sel].cpu().numpy(), logmel=lm_all[sel].cpu().numpy(),
                             ids=ids_all[sel].cpu().numpy(), notes=notes8, batch=np.int64(B), schedule=np.str_(sched),
                             rows=np.asarray(rows8, np.int64))
                    parity_args = ["--cpu-parity-file", pf]
                except Exception as ex:
                    out["cpu_parity_error"] = repr(ex)[:300]
            # separate process, hard wall-clock bound: the bench must finish in minutes on any host
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-segments",
                                    str(args.cpu_segments), "--cpu-small-segments", str(args.cpu_small_segments),
                                    "--cpu-enc-segments", str(args.cpu_enc_segments),
                                    "--decode-steps", str(args.decode_steps)] + parity_args,
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
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
