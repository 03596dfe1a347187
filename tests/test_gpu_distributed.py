"""The N>1 path on real GPUs: 2 ranks, backend "nccl" (= RCCL over xGMI), one process per GPU.  Exercises the CUDA
branch of `distributed.gather_token_rows` (all_gather_into_tensor) that the gloo test cannot reach, on the rows the
engine itself decodes: each rank encodes + decodes ITS shard of a segment list, the gathered rows must equal a
single-GPU decode of the whole list, bit for bit (segments are independent units).
Skipped on 1-GPU boxes (gpurun's); the driver's multi-GPU node runs it."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _decode_rows(audio, steps):
    from mt3_amd import network, spectrograms, vocabularies
    cfg = network.T5Config(dtype="bfloat16")
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=audio.shape[0])
    eng.load_params(network.init_random_params(cfg, seed=0))
    eng.encode(spectrograms.compute_spectrogram_batch(audio, None))
    ids = eng.decode(num_steps=steps)
    vocab = vocabularies.vocabulary_from_codec(vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1)))
    return vocab.decode_tf(ids)


def _worker(rank, world, port, n_items, steps, q):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    from mt3_amd import distributed, synthetic
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    assert dist.get_world_size() == world
    audio = synthetic.synth_audio(n_items, seed=77)                       # same list on every rank
    lo, hi = distributed.shard_range(n_items, rank, world)
    local = _decode_rows(audio[lo:hi].contiguous(), steps)
    allrows = distributed.gather_token_rows(local, n_items)               # CUDA branch: all_gather_into_tensor
    assert allrows.is_cuda and allrows.shape == (n_items, 1024)
    to0 = distributed.gather_token_rows(local, n_items, dst=0)            # what bench.py issues: ONE gather to rank 0
    assert (to0 is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(to0, allrows)
        q.put(allrows.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (one RCCL rank per GPU)")
@pytest.mark.parametrize("n_items", [16, 13])                             # even and ragged shards
def test_two_rank_rccl_gather_equals_single_gpu(n_items):
    steps = 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from mt3_amd import synthetic
    torch.cuda.set_device(0)
    want = _decode_rows(synthetic.synth_audio(n_items, seed=77), steps).cpu().numpy()
    assert np.array_equal(got, want)
