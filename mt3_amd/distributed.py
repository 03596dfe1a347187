"""Multi-GPU sharding of the MT3 path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

Segments are independent units (the reference itself only shards the batch axis:
PartitionSpec('data',) in the notebook's `_get_predict_fn`, NB:270-275), so weights are replicated, each rank
takes a contiguous slice of the global segment list, and the ONLY collective is ONE gather of the decoded int32
token rows to the rank that runs the host note decoding.  Every rank computes every shard size locally from
`shard_range` -- no size exchange, no host synchronisation in front of the payload collective.  The payload is tiny
(B x 1024 x 4 bytes per rank: 1 MB at B = 256, 5 MB at 1250 segments), i.e. latency-bound: one flat collective.

The only cross-segment dependency of the path is the HOST state machine that walks the segments of one FILE in
start-time order (mt3/metrics_utils.py:92-116): files are runs of consecutive segments whose boundaries do not depend
on the number of ranks, so the notes a job produces are identical for every world size (tests/test_distributed_gloo.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_token_rows(tokens, n_items: Optional[int] = None, dst: Optional[int] = None):
    """ONE collective: per-rank token rows [n_local, L] (int32 tensor, CUDA or CPU; n_local = this rank's
    `shard_range(n_items, rank, world)` size) -> the global [n_items, L] tensor in shard order.
    dst = None: all-gather (every rank gets the rows); dst = r: gather to rank r only (the others return None).
    Shard sizes come from `shard_range` on every rank (they differ by at most one row, so the padded send buffer has
    base + 1 rows); nothing is exchanged or synchronised before the payload.  Without an initialised process group:
    identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tokens
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_items is None:
        raise ValueError("gather_token_rows needs the global row count (shard sizes are computed from it, not exchanged)")
    counts = [hi - lo for lo, hi in (shard_range(n_items, r, world) for r in range(world))]
    if tokens.shape[0] != counts[rank]:
        raise RuntimeError("rank %d holds %d rows, its shard of %d items has %d" % (rank, tokens.shape[0], n_items, counts[rank]))
    m = max(counts)
    send = tokens.contiguous()
    if send.shape[0] < m:
        send = torch.cat([send, torch.zeros((m - send.shape[0], send.shape[1]), device=send.device, dtype=send.dtype)], 0)
    if dst is None:
        out = torch.empty((world * m, send.shape[1]), device=send.device, dtype=send.dtype)
        if send.is_cuda:
            dist.all_gather_into_tensor(out, send)
        else:
            parts = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(parts, send)
            out = torch.cat(parts, 0)
        parts = [out[r * m: r * m + counts[r]] for r in range(world)]
    else:
        parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, parts, dst=dst)
        if rank != dst:
            return None
        parts = [p[: counts[r]] for r, p in enumerate(parts)]
    return parts[0] if all(c == 0 for c in counts[1:]) else torch.cat(parts, 0)


def file_ranges(n_items: int, file_segments: int) -> List[Tuple[int, int]]:
    """The corpus as files of `file_segments` consecutive segments (the last one ragged)."""
    return [(a, min(a + file_segments, n_items)) for a in range(0, n_items, file_segments)]


class ShardedTranscriber:
    """One rank's share of a segment corpus: transcribe the shard in engine-sized calls, ONE gather of the token rows
    to rank 0, host note decoding per file on rank 0's worker threads (overlapping the next pass's launches).

    transcribe(first, count) -> int32 tensor [count, L] of `decode_tf` tokens for global segments first .. first+count-1
    (the caller binds the frontend + engine; tests bind a stub).  notes(rows, first) -> whatever the host stage
    returns for the file whose segments start at global index `first` (rows: numpy [n, L])."""

    def __init__(self, n_items: int, rank: int, world: int, transcribe: Callable, notes: Callable, call_segments: int,
                 file_segments: int, host_threads: int = 8, on_gather: Optional[Callable] = None,
                 row_length: Optional[int] = None, device=None):
        """row_length / device: shape and placement of the token rows `transcribe` returns.  Only needed when a rank
        can end up with an EMPTY shard (n_items < world): that rank has no `transcribe` result to take them from and
        still has to enter the gather with a [0, row_length] tensor, or the other ranks wait for it forever.  Without
        them a corpus smaller than the world is refused here -- on EVERY rank (all of them see the same n_items and
        world), so the job fails loudly instead of hanging."""
        from concurrent.futures import ThreadPoolExecutor
        if n_items <= 0:
            raise ValueError("ShardedTranscriber: empty corpus")
        if n_items < world and row_length is None:
            raise ValueError("ShardedTranscriber: %d segments over %d ranks leaves ranks without a shard; pass "
                             "row_length (and device) so that they can still enter the gather" % (n_items, world))
        self.n_items, self.rank, self.world = n_items, rank, world
        self.row_length, self.device = row_length, device
        self.lo, self.hi = shard_range(n_items, rank, world)
        self.call_segments = max(1, call_segments)
        self.files = file_ranges(n_items, file_segments)
        self._transcribe, self._notes, self._on_gather = transcribe, notes, on_gather
        self._pool = ThreadPoolExecutor(max_workers=max(1, host_threads)) if rank == 0 else None
        self._pending: List[list] = []

    def step(self):
        """one pass over this rank's shard; rank 0 also queues the host stage of the gathered rows"""
        import torch
        parts = [self._transcribe(s, min(self.call_segments, self.hi - s))
                 for s in range(self.lo, self.hi, self.call_segments)]
        if not parts:       # an empty shard (n_items < world): this rank still enters the collective
            tokens = torch.zeros((0, self.row_length), dtype=torch.int32, device=self.device or "cpu")
        else:
            tokens = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        if self.world > 1:
            if self._on_gather:
                self._on_gather(0)
            tokens = gather_token_rows(tokens, self.n_items, dst=0)
            if self._on_gather:
                self._on_gather(1)
        if self.rank == 0:
            host = tokens.cpu().numpy()                                  # syncs the stream
            self._pending.append([self._pool.submit(self._notes, host[a:b], a) for a, b in self.files])

    def drain(self) -> Sequence:
        """join every queued host stage; returns the per-file results of the LAST pass (rank 0; [] elsewhere)"""
        last: list = []
        while self._pending:
            last = [f.result() for f in self._pending.pop(0)]
        return last
