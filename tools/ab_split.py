"""The row-group overlap experiment (VERDICT r2, next #2 iii): the bf16 decode of 256 segments as one graph-replayed
chain (the product) against 2 / 3 / 4 row groups, one host thread + one stream each with direct launches, the streams
unmasked / masked to contiguous blocks of the CU-mask bits / masked to interleaved bits (mt3_debug_engine_decode_split).
Prints wall ms of the 1024-step decode loop and whether the ids are identical."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import network, spectrograms, synthetic  # noqa: E402

B = int(os.environ.get("AB_B", "256"))
import dataclasses  # noqa: E402
cfg = dataclasses.replace(network.MT3_BASE if os.environ.get("AB_MODEL") == "base" else network.MT3_SMALL,
                          dtype=os.environ.get("AB_DTYPE", "bfloat16"), kv_dtype=os.environ.get("AB_KV", ""))
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B)
eng.load_params(network.init_random_params(cfg, seed=0))
stream = torch.cuda.Stream()
lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=1000), None)
with torch.cuda.stream(stream):
    eng.encode(lm)
    eng.decode(num_steps=2)
    best = 1e30
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ref = eng.decode(num_steps=1024, single_stream=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print("one chain, hipGraph replay, one stream        : %8.1f ms" % best, flush=True)
    t0 = time.perf_counter()
    d = eng.decode(num_steps=1024)
    torch.cuda.synchronize()
    from mt3_amd import _lib as K
    print("mt3_engine_decode as shipped (row groups)     : %8.1f ms  ids equal %s  (%d groups, %d fallbacks)" % (
        (time.perf_counter() - t0) * 1e3, bool(torch.equal(d, ref)), eng.status(K.STATUS_LAST_DECODE_GROUPS),
        eng.status(K.STATUS_PARTITION_FALLBACKS)), flush=True)
    t0 = time.perf_counter()
    d = eng.decode(num_steps=1024, use_graph=False)
    torch.cuda.synchronize()
    print("one chain, direct launches, one host thread   : %8.1f ms  ids equal %s" % ((time.perf_counter() - t0) * 1e3,
                                                                                  bool(torch.equal(d, ref))), flush=True)
    cases = [(g, m, l) for g in (2, 3, 4) for m, l in ((0, "no CU mask"), (1, "contiguous CU-mask blocks"),
                                                        (2, "interleaved CU-mask bits"))]
    cases += [(2, m, "interleaved, stagger %d us" % us) for m, us in ((3, 8), (4, 15), (5, 25), (6, 40))]
    cases += [(2, 2, "interleaved CU-mask bits (again)")]
    cases += [(2, m, "OVERLAPPING masks, %d/8 of the CUs each" % k) for m, k in ((7, 5), (8, 6), (9, 7))]
    cases += [(g, 10, "FULL mask (all CUs) on every stream") for g in (2, 3, 4)]
    cases += [(2, 2, "interleaved CU-mask bits (3rd)")]
    if os.environ.get("AB_SPLIT_ONLY") == "product":
        G = int(os.environ.get("AB_GROUPS", "4"))
        cases = [(G, 10, "FULL masks"), (G, 11, "FULL masks, caller drives group 0"),
                 (G, 12, "FULL masks, own done slots + events"), (G, 13, "FULL masks, streams kept (1st use)"),
                 (G, 13, "FULL masks, streams kept (2nd use)"), (G, 14, "FULL masks, threads do not sync"),
                 (G, 15, "decode_partitioned() itself"), (G, 16, "decode_partitioned() on fresh streams"),
                 (G, 17, "this loop on the engine's streams"), (G, 10, "FULL masks (again)")]
        cases = [(G, 15, "decode_partitioned() itself"), (G, 32 + 3, "caller drives + events"),
                 (G, 32 + 5, "caller drives + threads do not sync"), (G, 32 + 6, "events + threads do not sync"),
                 (G, 32 + 7, "all three"), (G, 32 + 15, "all three + only the caller's stream is synchronised"),
                 (G, 32 + 14, "events, no thread sync, only the caller's stream synchronised"), (G, 10, "FULL masks (again)")]
    if os.environ.get("AB_SPLIT_ONLY") == "product":
        pass
    elif os.environ.get("AB_SPLIT_ONLY") == "overlap":
        cases = [c for c in cases if c[1] in (7, 8, 9, 10) or (c[0] == 2 and c[1] == 2)][1:]
    elif os.environ.get("AB_SPLIT_ONLY") == "full":
        cases = [c for c in cases if c[1] == 10 or (c[0] == 2 and c[1] == 2)][1:]
    elif os.environ.get("AB_SPLIT_ONLY"):
        cases = [c for c in cases if c[0] == 2 and c[1] >= 2]
    for groups, mode, label in cases:
        if True:
            try:
                eng.debug_decode_split(num_steps=8, groups=groups, mask_mode=mode)
                ids, ms = eng.debug_decode_split(num_steps=1024, groups=groups, mask_mode=mode)
                print("%d groups, %-34s    : %8.1f ms  ids equal %s" % (groups, label, ms, bool(torch.equal(ids, ref))),
                      flush=True)
            except Exception as ex:
                print("%d groups, %s: FAILED %r" % (groups, label, ex), flush=True)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = eng.decode(num_steps=1024)
        torch.cuda.synchronize()
        print("mt3_engine_decode as shipped, at the end      : %8.1f ms  ids equal %s  (%d groups)" % (
            (time.perf_counter() - t0) * 1e3, bool(torch.equal(d, ref)), eng.status(K.STATUS_LAST_DECODE_GROUPS)), flush=True)
