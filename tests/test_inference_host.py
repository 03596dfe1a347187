"""Host-only pieces of InferenceModel (no engine, no GPU): framing of the audio, segment start times, EOS trimming.
The reference code is the class in mt3/colab/music_transcription_with_transformers.ipynb ("Imports and Definitions":
`_audio_to_frames` ipynb:318, `postprocess` :346, `_trim_eos` :359) and mt3/spectrograms.py:55-61 (`split_audio`)."""
import numpy as np

from mt3_amd import inference, spectrograms, vocabularies


def _host_model():
    """an InferenceModel with the host attributes only (its __init__ builds the GPU engine)"""
    m = object.__new__(inference.InferenceModel)
    m.spectrogram_config = spectrograms.SpectrogramConfig()
    m.codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
    m.inputs_length = 256
    return m


def test_audio_to_frames_pads_like_the_notebook():
    m = _host_model()
    hop = m.spectrogram_config.hop_width
    for n in (1, hop - 1, hop, hop + 1, 5 * hop, 16000 * 3 + 17):
        audio = np.arange(1, n + 1, dtype=np.float32)
        frames, times = m._audio_to_frames(audio)
        # the notebook pads by hop - n % hop: a FULL hop when n is already aligned
        want = n // hop + 1
        assert frames.shape == (want, hop) and times.shape == (want,)
        flat = frames.reshape(-1)
        assert np.array_equal(flat[:n], audio) and not flat[n:].any()
        assert np.array_equal(times, np.arange(want) / m.spectrogram_config.frames_per_second)
        assert times.dtype == np.float64


def test_split_audio_is_frame_with_pad_end():
    cfg = spectrograms.SpectrogramConfig()
    x = np.arange(300, dtype=np.float32)
    f = spectrograms.split_audio(x, cfg)
    assert f.shape == (3, 128) and np.array_equal(f.reshape(-1)[:300], x) and not f.reshape(-1)[300:].any()
    assert spectrograms.split_audio(np.zeros(0, np.float32), cfg).shape == (0, 128)
    assert cfg.frames_per_second == 125.0 and cfg.abbrev_str == ""
    assert spectrograms.SpectrogramConfig(hop_width=64, num_mel_bins=256).abbrev_str == "hw64mb256"


def test_postprocess_floors_the_start_time_to_the_codec_step_in_float64():
    m = _host_model()
    assert m.codec.steps_per_second == 100
    fps = m.spectrogram_config.frames_per_second
    for seg in (0, 1, 2, 3, 7, 25, 100, 1171, 4882):
        t0 = (seg * 256) / fps                                   # input_times[0] of segment `seg`: seg * 2.048
        out = m.postprocess(np.array([5, 6, 1, 0, 0], np.int32) - 0, {"input_times": np.array([t0, t0 + 1 / fps])})
        want = t0 - t0 % (1 / 100)                               # the notebook's expression, evaluated in float64
        assert out["start_time"] == want and isinstance(out["start_time"], float)
        assert 0 <= t0 - out["start_time"] < 0.01 + 1e-12
        assert out["raw_inputs"] == []
    # segment 1 starts at 2.048 s and is decoded from 2.04 s
    out = m.postprocess(np.array([3, -1, -1], np.int32), {"input_times": np.array([2.048])})
    assert abs(out["start_time"] - 2.04) < 1e-12 and np.array_equal(out["est_tokens"], [3])


def test_trim_eos():
    eos = vocabularies.DECODED_EOS_ID
    assert eos == -1
    assert np.array_equal(inference.trim_eos([4, 5, eos, 7, eos]), [4, 5])
    assert np.array_equal(inference.trim_eos([4, 5, 6]), [4, 5, 6])
    assert inference.trim_eos([eos]).size == 0 and inference.trim_eos([]).size == 0
    assert inference.InferenceModel._trim_eos([9, eos]).dtype == np.int32


def test_the_engine_is_sized_to_the_job_while_batch_size_stays_the_references_8(monkeypatch):
    """`batch_size` is what `input_shapes` reports (NB:190); the engine behind `predict_tokens` grows to the job in
    powers of two up to `max_slots` and never shrinks.  A stub stands in for the engine: no GPU."""
    from mt3_amd import network
    built = []

    class StubEngine:
        def __init__(self, cfg, input_length, max_decode_length, max_batch):
            self.max_batch = max_batch
            built.append(max_batch)

        def load_params(self, params):
            self.params = params

    monkeypatch.setattr(network, "Transformer", StubEngine)
    m = inference.InferenceModel({"w": 0}, "mt3", max_slots=256)
    assert built == [8] and m.batch_size == 8 and m.engine_slots == 8
    assert m.input_shapes == {"encoder_input_tokens": (8, 256), "decoder_input_tokens": (8, 1024)}
    for n, want in ((3, 8), (8, 8), (9, 16), (12, 16), (293, 256), (40, 256), (5000, 256)):
        m._ensure_slots(n)
        assert m.engine_slots == want and m.batch_size == 8, (n, m.engine_slots)
    assert built == [8, 16, 256] and m.model.params == {"w": 0}
    small = inference.InferenceModel({"w": 0}, "mt3", max_slots=4, batch_size=8)        # max_slots below batch_size: ignored
    small._ensure_slots(100)
    assert small.engine_slots == 8
    import pytest
    with pytest.raises(ValueError):
        inference.InferenceModel({"w": 0}, "mt3", schedule="nope")
