"""Mask helpers (mt3/layers_test.py:117-283 literals) and the feature converter."""
import numpy as np

from mt3_amd import layers, models


def test_make_attention_mask_literals():
    tgt = np.array([[7, 0, 0], [8, 5, 0]])
    m = layers.make_attention_mask(tgt > 0, tgt > 0, dtype=np.int32)
    assert m.shape == (2, 1, 3, 3)
    np.testing.assert_array_equal(m[0, 0], [[1, 0, 0], [0, 0, 0], [0, 0, 0]])
    np.testing.assert_array_equal(m[1, 0], [[1, 1, 0], [1, 1, 0], [0, 0, 0]])
    seg = np.array([[1, 1, 2, 2, 2, 0], [1, 1, 1, 2, 0, 0]])
    m = layers.make_attention_mask(seg, seg, pairwise_fn=np.equal, dtype=np.int32)
    np.testing.assert_array_equal(m[0, 0], [[1, 1, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0], [0, 0, 1, 1, 1, 0],
                                            [0, 0, 1, 1, 1, 0], [0, 0, 1, 1, 1, 0], [0, 0, 0, 0, 0, 1]])
    np.testing.assert_array_equal(m[1, 0], [[1, 1, 1, 0, 0, 0], [1, 1, 1, 0, 0, 0], [1, 1, 1, 0, 0, 0],
                                            [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 1], [0, 0, 0, 0, 1, 1]])


def test_causal_and_combine_literals():
    y = layers.make_causal_mask(np.array([[7, 0, 0], [8, 5, 0]]))
    assert y.shape == (2, 1, 3, 3)
    np.testing.assert_allclose(y[0], [[[1., 0., 0.], [1., 1., 0.], [1., 1., 1.]]])
    assert layers.make_causal_mask(np.ones((3, 3, 5)), extra_batch_dims=2).shape == (1, 1, 3, 3, 1, 5, 5)
    f = np.float32
    np.testing.assert_allclose(layers.combine_masks(np.array([0, 1, 0, 1], f), None, np.array([1, 1, 1, 1], f),
                                                    np.array([1, 1, 1, 0], f)), [0, 1, 0, 0])
    np.testing.assert_allclose(layers.combine_biases(np.array([0, 1, 0, 1], f), None, np.array([0, 1, 1, 1], f),
                                                     np.array([0, 1, 1, 0], f)), [0, 3, 2, 2])
    assert layers.combine_masks(None) is None and layers.combine_biases() is None


def test_make_decoder_mask_literals():
    m = layers.make_decoder_mask(np.array([6, 7, 3, 0]), np.float32)
    np.testing.assert_array_equal(m, [[[1, 0, 0, 0], [1, 1, 0, 0], [1, 1, 1, 0], [0, 0, 0, 0]]])
    m = layers.make_decoder_mask(np.array([[6, 7, 3, 4, 5, 0]]), np.float32,
                                 decoder_segment_ids=np.array([[1, 1, 1, 2, 2, 0]]))
    np.testing.assert_array_equal(m, [[[[1, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0],
                                        [0, 0, 0, 1, 0, 0], [0, 0, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0]]]])
    m = layers.make_decoder_mask(np.array([[5, 6, 7, 3, 4, 0]]), np.float32,
                                 decoder_causal_attention=np.array([[1, 1, 1, 0, 0, 0]]))
    np.testing.assert_array_equal(m, [[[[1, 1, 1, 0, 0, 0], [1, 1, 1, 0, 0, 0], [1, 1, 1, 0, 0, 0],
                                        [1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 0], [0, 0, 0, 0, 0, 0]]]])
    m = layers.make_decoder_mask(np.array([[5, 6, 7, 8, 3, 4, 0]]), np.float32,
                                 decoder_causal_attention=np.array([[1, 1, 0, 1, 1, 0, 0]]),
                                 decoder_segment_ids=np.array([[1, 1, 1, 2, 2, 2, 0]]))
    np.testing.assert_array_equal(m, [[[[1, 1, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0],
                                        [0, 0, 0, 1, 1, 0, 0], [0, 0, 0, 1, 1, 0, 0], [0, 0, 0, 1, 1, 1, 0],
                                        [0, 0, 0, 0, 0, 0, 0]]]])
    m = layers.make_decoder_mask(np.array([[6, 7, 3, 4, 8, 9, 0]]), np.float32,
                                 decoder_causal_attention=np.array([[1, 1, 0, 0, 1, 1, 0]]))
    np.testing.assert_array_equal(m[0, 0], [[1, 1, 0, 0, 1, 1, 0], [1, 1, 0, 0, 1, 1, 0], [1, 1, 1, 0, 0, 0, 0],
                                            [1, 1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 0],
                                            [0, 0, 0, 0, 0, 0, 0]])


def test_feature_converter():
    ex = [{"inputs": np.ones((256, 512), np.float32), "targets": np.zeros((0,), np.int32)},
          {"inputs": np.full((100, 512), 2.0, np.float32), "targets": np.array([5, 6, 1], np.int32)},
          {"inputs": np.ones((300, 512), np.float32)}]
    b = models.convert_features(ex, {"inputs": 256, "targets": 1024})
    assert b["encoder_input_tokens"].shape == (3, 256, 512) and b["decoder_input_tokens"].shape == (3, 1024)
    assert np.all(b["encoder_input_tokens"][1, 100:] == 0.0) and np.all(b["encoder_input_tokens"][1, :100] == 2.0)
    np.testing.assert_array_equal(b["decoder_target_tokens"][1, :4], [5, 6, 1, 0])
    np.testing.assert_array_equal(b["decoder_input_tokens"][1, :4], [0, 5, 6, 1])
    assert not b["decoder_input_tokens"][0].any()
