"""The queue between the encoder passes and the row groups of mt3_engine_transcribe (mt3_amd/csrc/feed.h: the staging
ring's host protocol -- one producer, up to four consumers, everything under one mutex) under a thread stress test
WITHOUT a GPU: tests/host/feed_stress.cpp models a ring entry as an int stamped with the segment it holds and checks that
every segment is handed out exactly once with the right entry, that no chunk is overwritten before every consumer has
given it back (consumers read their entries only at their NEXT poll, as the refill copies do), that the padded last
chunk never offers its padding, that everybody terminates, and that a failing producer wakes everybody."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stress(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("feed") / "libfeed_stress.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "mt3_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "feed_stress.cpp"), "-o", out])
    lib = ctypes.CDLL(out)
    lib.feed_stress.restype = ctypes.c_int
    lib.feed_stress.argtypes = [ctypes.c_int] * 5 + [ctypes.c_uint] + [ctypes.c_int] * 4
    return lib


FAST = (300, 200, 0)         # (producer us, consumer us, most slots per poll): consumers keep up, the ring stays shallow
SLOW = (20, 400, 5)          # slow, small polls: the ring fills and the producer has to wait for chunks to come back


@pytest.mark.parametrize("n_total,slots,cap,consumers", [
    (10000, 1250, 64, 4),     # BASELINE configs[3] through 1250 slots: 137 chunks through the 8-chunk ring
    (2560, 256, 64, 4), (700, 24, 24, 1), (29, 8, 8, 1), (300, 40, 40, 2),
    (1003, 17, 17, 3),        # a last chunk of 1 segment, padded to 8
    (5, 16, 16, 1), (64, 64, 64, 4)])           # nothing to refill
def test_every_segment_once_no_chunk_overwritten_before_release(stress, n_total, slots, cap, consumers):
    for seed in range(4):
        for pace in (FAST, SLOW) if n_total <= 3000 else (FAST,):
            rc = stress.feed_stress(n_total, slots, cap, min(8, cap), consumers, seed, 0, *pace)
            assert rc == 0, (rc, seed, pace)


def test_a_failing_producer_wakes_everybody(stress):
    for seed in range(4):
        assert stress.feed_stress(5000, 100, 64, 8, 4, seed, 7, *FAST) == 0
        assert stress.feed_stress(5000, 100, 16, 8, 4, seed, 12, *SLOW) == 0          # ... also one that sits waiting for the ring


def test_the_stress_test_sees_a_broken_protocol(tmp_path):
    """Mutation check: a producer that does not wait for its ring chunk to be given back must be CAUGHT (an entry
    overwritten while a consumer still holds it, or a segment lost), otherwise the test above proves nothing."""
    src = open(os.path.join(ROOT, "mt3_amd", "csrc", "feed.h")).read()
    needle = "f.failed || q < kStageChunks || ch.released == ch.n"
    assert needle in src
    (tmp_path / "feed.h").write_text(src.replace(needle, "true"))
    out = str(tmp_path / "libmut.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I", str(tmp_path),
                           os.path.join(ROOT, "tests", "host", "feed_stress.cpp"), "-o", out])
    lib = ctypes.CDLL(out)
    lib.feed_stress.restype = ctypes.c_int
    lib.feed_stress.argtypes = [ctypes.c_int] * 5 + [ctypes.c_uint] + [ctypes.c_int] * 4
    assert any(lib.feed_stress(700, 24, 24, 8, 2, seed, 0, *SLOW) != 0 for seed in range(4))
