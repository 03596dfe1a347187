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
