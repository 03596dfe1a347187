"""CPU oracle for the MT3 inference hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference path
(magenta/mt3: audio -> log-mel -> T5 encoder-decoder -> event tokens -> notes).
It exists to *check* the HIP product in ``mt3_amd/``; it is never the thing
that is shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``mt3_amd/`` imports it (``tests/test_layout.py`` enforces that).

Pinning status (see DESIGN.md "Oracle"):
  * symbolic stage (codec / vocabulary / run-length decode / note state
    machine / segment combiner): PINNED -- checked against every literal of the
    reference's own unit tests and against golden vectors produced by running
    the reference's real Python modules in the build container
    (tests/golden/make_symbolic_golden.py).
  * layers (attention math, masks, DenseGeneral, MLP): pinned by the literals
    of mt3/layers_test.py that do not need JAX to evaluate.
  * log-mel frontend (tf.signal.*) and the t5x decode loop: PARITY UNPINNED --
    TensorFlow / JAX / t5x are not installable here, so those parts restate the
    published algorithms and are anchored only on the reference's call sites.
"""
