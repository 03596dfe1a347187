"""Vocabulary layer (mirror of mt3/vocabularies.py:29-282) over libmt3hip.so."""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from typing import Sequence

import numpy as np

from . import _lib
from . import event_codec

DECODED_EOS_ID = -1
DECODED_INVALID_ID = -2
DEFAULT_STEPS_PER_SECOND = 100
DEFAULT_MAX_SHIFT_SECONDS = 10
DEFAULT_NUM_VELOCITY_BINS = 127
DEFAULT_EXTRA_IDS = 100        # t5.data.DEFAULT_EXTRA_IDS (used at vocabularies.py:145)
MAX_MIDI_VELOCITY = 127


@dataclasses.dataclass
class VocabularyConfig:
    steps_per_second: int = DEFAULT_STEPS_PER_SECOND
    max_shift_seconds: int = DEFAULT_MAX_SHIFT_SECONDS
    num_velocity_bins: int = DEFAULT_NUM_VELOCITY_BINS


def num_velocity_bins_from_codec(codec: event_codec.Codec) -> int:
    lo, hi = codec.event_type_range("velocity")
    return hi - lo


def velocity_to_bin(velocity: int, num_velocity_bins: int) -> int:
    return 0 if velocity == 0 else math.ceil(num_velocity_bins * velocity / MAX_MIDI_VELOCITY)


def bin_to_velocity(velocity_bin: int, num_velocity_bins: int) -> int:
    return 0 if velocity_bin == 0 else int(MAX_MIDI_VELOCITY * velocity_bin / num_velocity_bins)


def build_codec(vocab_config: VocabularyConfig) -> event_codec.Codec:
    """shift | pitch | velocity | tie | program | drum, laid out by `mt3_build_codec`."""
    desc = _lib.CodecDesc()
    _lib.check(_lib.load().mt3_build_codec(vocab_config.steps_per_second, vocab_config.max_shift_seconds,
                                           vocab_config.num_velocity_bins, C.byref(desc)))
    return event_codec.Codec.from_desc(desc)


class GenericTokenVocabulary:
    """PAD=0, EOS=1, UNK=2, then `regular_ids` tokens, then `extra_ids` sentinels."""

    def __init__(self, regular_ids: int, extra_ids: int = 0):
        self._num_special_tokens = 3
        self._num_regular_tokens = int(regular_ids)
        self.extra_ids = int(extra_ids)

    pad_id, eos_id, unk_id = 0, 1, 2

    @property
    def _base_vocab_size(self) -> int:
        return self._num_special_tokens + self._num_regular_tokens

    @property
    def vocab_size(self) -> int:
        return self._base_vocab_size + self.extra_ids

    def encode(self, token_ids: Sequence[int]):
        out = []
        for t in token_ids:
            if not 0 <= t < self._num_regular_tokens:
                raise ValueError(f"token_id {t} does not fall within valid range of [0, {self._num_regular_tokens})")
            out.append(int(t) + self._num_special_tokens)
        return out

    def decode(self, ids: Sequence[int]):
        """Python-list path: truncates after the first EOS."""
        row = np.asarray(list(ids), np.int32).reshape(1, -1)
        toks = self.decode_tf(row)[0].tolist()
        return toks[: toks.index(DECODED_EOS_ID) + 1] if DECODED_EOS_ID in toks else toks

    def decode_tf(self, ids):
        """`_decode_tf` (mt3/vocabularies.py:241-271).  A CUDA int32 torch tensor goes through the
        `mt3_ids_to_tokens` kernel (returned as a tensor, no sync: the hot path).  Anything host-side (numpy,
        lists, CPU tensors: the `vocabulary.decode_tf(x).numpy()` use of the notebook and the t5x write_fn in
        inference.py) is remapped on the host with the same three-line rule, so host-only postprocessing needs
        no GPU and empty rows are fine."""
        try:
            import torch
            if isinstance(ids, torch.Tensor) and ids.is_cuda:
                shape, orig_dtype = ids.shape, ids.dtype
                if ids.numel() == 0:
                    return ids.clone()
                t2 = ids.reshape(-1, shape[-1]).to(dtype=torch.int32).contiguous()
                out = torch.empty_like(t2)
                _lib.check(_lib.load().mt3_ids_to_tokens(t2.data_ptr(), t2.shape[0], t2.shape[1],
                                                          self._num_regular_tokens, out.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream))
                out = out.reshape(shape)
                return out.to(orig_dtype) if orig_dtype != torch.int32 else out
            if isinstance(ids, torch.Tensor):
                return torch.from_numpy(self._decode_host(ids.numpy()))
        except ImportError:
            pass
        return self._decode_host(np.asarray(ids))

    def _decode_host(self, ids: np.ndarray) -> np.ndarray:
        src = np.asarray(ids)
        a = src.astype(np.int64)
        eos = a == self.eos_id
        after = np.cumsum(eos, axis=-1) > 0                     # from the first EOS to the end of the row
        ok = (a >= self._num_special_tokens) & (a < self._base_vocab_size)
        out = np.where(after, DECODED_EOS_ID, np.where(ok, a - self._num_special_tokens, DECODED_INVALID_ID))
        return out.astype(src.dtype if np.issubdtype(src.dtype, np.integer) else np.int32)

    def __eq__(self, other):
        return (self.extra_ids == other.extra_ids and self._num_regular_tokens == other._num_regular_tokens)


def vocabulary_from_codec(codec: event_codec.Codec) -> GenericTokenVocabulary:
    return GenericTokenVocabulary(codec.num_classes, extra_ids=DEFAULT_EXTRA_IDS)


def num_embeddings(vocabulary: GenericTokenVocabulary) -> int:
    return 128 * math.ceil(vocabulary.vocab_size / 128)


# mt3/vocabularies.py:94-117: the NoteSequence side of a program granularity (`program_map_fn`, idempotent); the token
# side (`tokens_map_fn`: drop_programs / programs_to_midi_classes) belongs to the training data pipeline, which is out of
# this path's scope.
PROGRAM_MAP_FNS = {
    "flat": lambda program: 0,                             # programs ignored
    "midi_class": lambda program: 8 * (program // 8),      # first program of the MIDI class
    "full": lambda program: program,
}
