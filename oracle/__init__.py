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
  * network (layers + encoder-decoder wiring + the decode-mode K/V cache path): PINNED on golden vectors
    produced by the reference's REAL mt3/layers.py + mt3/network.py, run unmodified in the build container on a
    numpy stand-in for the jax/flax entry points they use (tests/golden/make_network_golden.py,
    tests/golden/jax_standin.py -> tests/golden/network_golden.npz: encoder output, teacher-forced logits,
    cached one-token decode as t5x drives it), plus the literals of mt3/layers_test.py that do not need JAX.
    The leaf numerics (einsum, softmax, tanh-GELU) are numpy in that run, not XLA.
  * log-mel frontend: composition and parameters PINNED on the reference's REAL mt3/spectrograms.py +
    mt3/spectral_ops.py, run unmodified on a numpy stand-in for TensorFlow (tests/golden/make_frontend_golden.py,
    tf_standin.py -> frontend_golden.npz).  The tf.signal leaves (frame / stft / hann / HTK filterbank) are
    third-party code outside the reference tree: restated from the TensorFlow documentation (independently in the
    stand-in and in oracle/frontend.py) and cross-checked against torch.stft and scipy
    (tests/test_oracle_cross_checks.py) -- PARITY UNPINNED against TensorFlow itself.
  * t5x decode loop (beam_search): PARITY UNPINNED -- t5x is neither installable here nor in the reference tree;
    restated from memory of its source.
"""
