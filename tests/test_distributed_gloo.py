"""The N>1 path on CPU: world_size-2 gloo processes shard a segment list, all-gather the token
rows (mt3_amd.distributed) and rank 0 decodes notes -- must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_rows(n, L=64, seed=0):
    rng = np.random.default_rng(seed)
    rows = rng.integers(-2, 1400, (n, L)).astype(np.int32)
    rows[:, 0] = 1131                                   # tie token first: valid MT3-style rows
    return rows


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mt3_amd import distributed, metrics_utils, note_sequences, vocabularies
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = _make_rows(n_items)
    lo, hi = distributed.shard_range(n_items, rank, world)
    local = torch.from_numpy(rows[lo:hi])
    allrows = distributed.gather_token_rows(local, n_items)
    assert torch.equal(allrows, torch.from_numpy(rows))
    if rank == 0:
        codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
        preds = [{"est_tokens": r.numpy(), "start_time": i * 2.04} for i, r in enumerate(allrows)]
        res = metrics_utils.event_predictions_to_ns(preds, codec, note_sequences.NoteEncodingWithTiesSpec)
        q.put((len(res["est_ns"].notes), res["est_invalid_events"], res["est_dropped_events"],
               res["est_ns"].total_time))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7])          # even and ragged shards
def test_two_rank_gather_and_decode(n_items):
    from mt3_amd import distributed, metrics_utils, note_sequences, vocabularies
    assert distributed.shard_range(7, 0, 2) == (0, 4) and distributed.shard_range(7, 1, 2) == (4, 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = q.get(timeout=10)
    rows = _make_rows(n_items)
    codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
    preds = [{"est_tokens": r, "start_time": i * 2.04} for i, r in enumerate(rows)]
    res = metrics_utils.event_predictions_to_ns(preds, codec, note_sequences.NoteEncodingWithTiesSpec)
    assert got == (len(res["est_ns"].notes), res["est_invalid_events"], res["est_dropped_events"],
                   res["est_ns"].total_time)


# ------------------------------------------------------------------------------------------------------------
# The job bench.py runs at N > 1 (mt3_amd.distributed.ShardedTranscriber: shard -> engine calls -> ONE gather to
# rank 0 -> per-file host note decoding), driven by a STUB engine on gloo: rank 0's notes must equal a single-rank run
# of the same corpus, for a ragged corpus (10,001 segments: shards of 5001 + 5000; 40 files, the last of 17 segments),
# with start times mapped per global segment.  Reference split: NB:270-275 (batch axis only).
L_STUB = 48


def _stub_rows(first, count):
    """deterministic token rows of global segments [first, first + count): MT3-style (tie token, then shift / pitch /
    program tokens drawn from a per-segment generator, a -1 tail of random length)"""
    out = np.empty((count, L_STUB), np.int32)
    for i in range(count):
        rng = np.random.default_rng(1_000_003 * (first + i) + 7)
        row = rng.integers(0, 1388, L_STUB).astype(np.int32)
        row[0] = 1131
        row[rng.integers(L_STUB // 2, L_STUB + 1):] = -1
        out[i] = row
    return out


def _notes_of_file_factory():
    from mt3_amd import metrics_utils, note_sequences, vocabularies
    codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))

    def notes_of_file(rows, first):
        eos = rows == -1
        n_tok = np.where(eos.any(1), eos.argmax(1), rows.shape[1])
        starts = [(first + i) * 2.048 - ((first + i) * 2.048) % 0.01 for i in range(len(rows))]
        ns, inv, drop = metrics_utils._run(codec, note_sequences.NoteEncodingWithTiesSpec.spec_id,
                                           [r[:n] for r, n in zip(rows, n_tok)], starts)
        h = hash(tuple((n.start_time, n.end_time, n.pitch, n.program, n.is_drum) for n in ns.notes))
        return (first, len(ns.notes), inv, drop, h)
    return notes_of_file


def _job_worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mt3_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def transcribe(first, count):
        calls.append((first, count))
        return torch.from_numpy(_stub_rows(first, count))
    collectives = []
    job = distributed.ShardedTranscriber(n_items, rank, world, transcribe, _notes_of_file_factory(), call_segments=1250,
                                         file_segments=256, host_threads=4, on_gather=lambda ph: collectives.append(ph))
    job.step()
    job.step()                                             # two passes: the pending queue and the pool are reused
    res = job.drain()
    lo, hi = distributed.shard_range(n_items, rank, world)
    assert calls[: len(calls) // 2] == [(s, min(1250, hi - s)) for s in range(lo, hi, 1250)], calls
    assert collectives == [0, 1, 0, 1]                     # exactly one collective per pass
    if rank == 0:
        q.put(res)
    else:
        assert res == []
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_job_matches_a_single_rank_run_on_a_ragged_corpus():
    from mt3_amd import distributed
    n_items = 10_001
    assert distributed.shard_range(n_items, 0, 2) == (0, 5001) and distributed.shard_range(n_items, 1, 2) == (5001, 10001)
    assert distributed.file_ranges(n_items, 256)[-1] == (9984, 10001) and len(distributed.file_ranges(n_items, 256)) == 40
    # single rank, no process group: gather is the identity
    single = distributed.ShardedTranscriber(n_items, 0, 1, lambda f, c: torch.from_numpy(_stub_rows(f, c)),
                                            _notes_of_file_factory(), call_segments=1250, file_segments=256, host_threads=4)
    single.step()
    want = single.drain()
    assert len(want) == 40 and want[0][0] == 0 and want[-1][0] == 9984 and sum(w[1] for w in want) > 1_000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got == want


def test_gather_needs_no_size_exchange_and_rejects_wrong_shards():
    """Shard sizes come from shard_range on every rank; without a process group the call is the identity."""
    from mt3_amd import distributed
    t = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    assert distributed.gather_token_rows(t, 3) is t and distributed.gather_token_rows(t, 3, dst=0) is t
    src = open(os.path.join(ROOT, "mt3_amd", "distributed.py")).read()
    assert ".item()" not in src and src.count("dist.all_gather(") == 1 and "dist.gather(" in src


# ------------------------------------------------------------------------------------------------------------
# A corpus SMALLER than the world: the rank without a shard must still enter the gather (it used to die
# in torch.cat([]) while the other rank waited in dist.gather forever), or the job must refuse on every rank.
def _tiny_job_worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mt3_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        distributed.ShardedTranscriber(n_items, rank, world, None, None, call_segments=4, file_segments=4)
        refused = False
    except ValueError:
        refused = True                                    # on EVERY rank: nobody is left waiting in a collective
    calls = []

    def transcribe(first, count):
        calls.append((first, count))
        return torch.from_numpy(_stub_rows(first, count))
    job = distributed.ShardedTranscriber(n_items, rank, world, transcribe, _notes_of_file_factory(), call_segments=4,
                                         file_segments=4, host_threads=2, row_length=L_STUB)
    job.step()
    res = job.drain()
    lo, hi = distributed.shard_range(n_items, rank, world)
    assert calls == ([(lo, hi - lo)] if hi > lo else [])
    q.put((rank, refused, res))
    dist.barrier()
    dist.destroy_process_group()


def test_a_corpus_smaller_than_the_world_neither_hangs_nor_loses_rows():
    from mt3_amd import distributed
    assert distributed.shard_range(1, 0, 2) == (0, 1) and distributed.shard_range(1, 1, 2) == (1, 1)
    single = distributed.ShardedTranscriber(1, 0, 1, lambda f, c: torch.from_numpy(_stub_rows(f, c)),
                                            _notes_of_file_factory(), call_segments=4, file_segments=4, host_threads=2)
    single.step()
    want = single.drain()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tiny_job_worker, args=(r, 2, port, 1, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == [(0, True, want), (1, True, [])]
