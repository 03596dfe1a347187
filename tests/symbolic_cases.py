"""Known-answer cases for the symbolic stage, transcribed from the reference's
own unit tests (literals only), shared by the oracle tests and the product
(C++ host library) tests.

Each case: codec ranges (excluding the implicit shift range), max_shift_steps,
mode, segments [(start_time, tokens, max_time-or-"auto")], expected notes
(start, end, pitch, velocity, program, is_drum), invalid, dropped, total_time.
"""
PITCH = ("pitch", 0, 127)
VEL = ("velocity", 0, 127)
DRUM = ("drum", 0, 127)
PROG = ("program", 0, 127)
TIE = ("tie", 0, 0)

# note_sequences_test.py:28-40 codec
NS_CODEC = [PITCH, VEL, DRUM, PROG, TIE]

# single-segment decode_events KATs: note_sequences_test.py:290-501
SINGLE = [
    dict(name="onsets", ranges=NS_CODEC, mode="onsets", tokens=[25, 161, 50, 162], start=0, max_time=None,
         notes=[(0.25, 0.26, 60, 100, 0, False), (0.50, 0.51, 61, 100, 0, False)], invalid=0, dropped=0, total=0.51),
    dict(name="onsets_only", ranges=NS_CODEC, mode="onsets", tokens=[5, 161, 25, 162], start=0, max_time=None,
         notes=[(0.05, 0.06, 60, 100, 0, False), (0.25, 0.26, 61, 100, 0, False)], invalid=0, dropped=0, total=0.26),
    dict(name="velocity", ranges=NS_CODEC, mode="notes", tokens=[5, 356, 161, 25, 229, 161], start=0, max_time=None,
         notes=[(0.05, 0.25, 60, 127, 0, False)], invalid=0, dropped=0, total=0.25),
    dict(name="missing_offset", ranges=NS_CODEC, mode="notes", tokens=[5, 356, 161, 10, 161, 25, 229, 161],
         start=0, max_time=None,
         notes=[(0.05, 0.10, 60, 127, 0, False), (0.10, 0.25, 60, 127, 0, False)], invalid=0, dropped=0, total=0.25),
    dict(name="multitrack", ranges=NS_CODEC, mode="notes",
         tokens=[5, 525, 356, 161, 15, 356, 394, 25, 525, 229, 161], start=0, max_time=None,
         notes=[(0.15, 0.16, 37, 127, 0, True), (0.05, 0.25, 60, 127, 40, False)], invalid=0, dropped=0, total=0.25,
         instruments=[9, 0]),
    dict(name="invalid_tokens", ranges=NS_CODEC, mode="onsets", tokens=[5, -1, 161, -2, 25, 162, 9999], start=0,
         max_time=None,
         notes=[(0.05, 0.06, 60, 100, 0, False), (0.25, 0.26, 61, 100, 0, False)], invalid=3, dropped=0, total=0.26),
    dict(name="exactly_max_time", ranges=NS_CODEC, mode="onsets", tokens=[161, 25, 162], start=1.0, max_time=1.25,
         notes=[(1.00, 1.01, 60, 100, 0, False), (1.25, 1.26, 61, 100, 0, False)], invalid=0, dropped=0, total=1.26),
    dict(name="dropped", ranges=NS_CODEC, mode="onsets", tokens=[5, 161, 30, 162], start=1.0, max_time=1.25,
         notes=[(1.05, 1.06, 60, 100, 0, False)], invalid=0, dropped=2, total=1.06),
    dict(name="invalid_events", ranges=NS_CODEC, mode="onsets", tokens=[25, 230, 50, 161], start=0, max_time=None,
         notes=[(0.50, 0.51, 60, 100, 0, False)], invalid=1, dropped=0, total=0.51),
]

# multi-segment combiner KATs: metrics_utils_test.py:28-238
COMBINE = [
    dict(name="onsets", ranges=[PITCH], mode="onsets",
         segments=[(0.0, [20, 160]), (0.4, [20, 161, 50, 162]), (0.8, [163, 20, 164])],
         notes=[(0.20, 0.21, 59, 100, 0, False), (0.60, 0.61, 60, 100, 0, False),
                (0.80, 0.81, 62, 100, 0, False), (1.00, 1.01, 63, 100, 0, False)],
         invalid=0, dropped=2, total=1.01),
    dict(name="offsets", ranges=[PITCH, VEL], mode="notes",
         segments=[(0.0, [20, 356, 160]), (0.4, [20, 292, 161]), (0.8, [20, 229, 160, 161])],
         notes=[(0.20, 1.00, 59, 127, 0, False), (0.60, 1.00, 60, 63, 0, False)],
         invalid=0, dropped=0, total=1.00),
    dict(name="multitrack", ranges=[PITCH, VEL, DRUM, PROG], mode="notes",
         segments=[(0.0, [20, 517, 356, 160]), (0.4, [20, 356, 399]), (0.8, [20, 517, 229, 160])],
         notes=[(0.60, 0.61, 42, 127, 0, True), (0.20, 1.00, 59, 127, 32, False)],
         invalid=0, dropped=0, total=1.00, instruments=[9, 0]),
    dict(name="multitrack_ties", ranges=[PITCH, VEL, DRUM, PROG, TIE], mode="ties",
         segments=[(0.0, [613, 20, 517, 356, 160]), (0.4, [517, 160, 613, 20, 356, 399]), (0.8, [613])],
         notes=[(0.60, 0.61, 42, 127, 0, True), (0.20, 0.80, 59, 127, 32, False)],
         invalid=0, dropped=0, total=0.80, instruments=[9, 0]),
]
