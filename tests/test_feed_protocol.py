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
    lib.feed_stress.argtypes = [ctypes.c_int] * 5 + [ctypes.c_uint, ctypes.c_int]
    return lib


@pytest.mark.parametrize("n_total,slots,cap,consumers", [
    (10000, 1250, 64, 4),     # BASELINE configs[3] through 1250 slots: 137 chunks through the 8-chunk ring
    (2560, 256, 64, 4), (700, 24, 24, 1), (29, 8, 8, 1), (300, 40, 40, 2),
    (1003, 17, 17, 3),        # a last chunk of 1 segment, padded to 8
    (5, 16, 16, 1), (64, 64, 64, 4)])           # nothing to refill
def test_every_segment_once_no_chunk_overwritten_before_release(stress, n_total, slots, cap, consumers):
    for seed in range(6):
        rc = stress.feed_stress(n_total, slots, cap, min(8, cap), consumers, seed, 0)
        assert rc == 0, (rc, seed)


def test_a_failing_producer_wakes_everybody(stress):
    for seed in range(4):
        assert stress.feed_stress(5000, 100, 64, 8, 4, seed, 7) == 0
