/*
 * mt3_hip_debug.h -- measurement and fault-injection entry points of libmt3hip.so.
 *
 * NOT part of the product ABI (include/mt3_hip.h): nothing here has a counterpart in the reference
 * (magenta/mt3), a drop-in caller never needs it, and results of the "skip" variants are meaningless.
 * bench.py uses mt3_debug_engine_decode for the in-situ duration of the decode-attention kernels
 * (a difference of whole-decode times instead of 8192 per-launch event pairs) and
 * mt3_debug_engine_set_eos_schedule for its synthetic-length figure; tests use
 * mt3_debug_engine_poison_caches to prove that stale cache contents cannot leak into results.
 * Same conventions as mt3_hip.h.
 */
#ifndef MT3_HIP_DEBUG_H_
#define MT3_HIP_DEBUG_H_

#include <stdint.h>

#include "mt3_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* mt3_engine_decode with kernels LEFT OUT of every step (ids are garbage): `skip` is a bit set of the values
 * below.  flags as for mt3_engine_decode. */
enum { MT3_DEBUG_SKIP_SELF_ATTN = 1, MT3_DEBUG_SKIP_CROSS_ATTN = 2 };
int mt3_debug_engine_decode(mt3_engine* e, int32_t batch, int32_t num_steps, int32_t flags, int32_t skip,
                            int32_t* d_ids, void* stream);

/* SYNTHETIC EOS SCHEDULE (bench.py's `eos_schedule` figure, SURVEY.md 8(d): "a second figure with a synthetic EOS
 * schedule (lengths ~ clipped N(300,100))"; tests of the row retirement): random-init weights never emit EOS reliably, so
 * the output LENGTH of row r is imposed -- at decode step h_lengths[r] - 1 (0-based) the model's distribution of that
 * row is replaced by a point mass on EOS: the greedy decode emits EOS there (the row's output has h_lengths[r] tokens,
 * the last one EOS), the beam-1 search finishes `prefix + EOS` with log-probability 0 and closes.  Everything before
 * that step is the model's own arithmetic.  h_lengths [n] host int32, each >= 1 (rows >= n: never forced);
 * n may exceed max_batch: in mt3_engine_transcribe entry i is the length of SEGMENT i (the call refuses a schedule
 * shorter than its n_segments).  NULL switches the schedule off.  Synchronous (a setup call); applies to every later mt3_engine_decode of the engine,
 * not to mt3_engine_decode_forced. */
int mt3_debug_engine_set_eos_schedule(mt3_engine* e, const int32_t* h_lengths, int32_t n);

/* mt3_engine_transcribe with the two schedule parameters the product fixes, for A/B measurements: poll_steps = decode
 * steps between two refill polls of a row group (0: the product's 4), row_groups = 1 .. 4 (0: the product's rule).  Same
 * ids whatever the values (tests/test_gpu_transcribe.py).  skip_encoder_passes != 0: the encoder passes of the refill
 * chunks are LEFT OUT -- differential timing only, the ids of refilled segments are garbage (under an imposed EOS
 * schedule the decode does the same work, so the difference of two runs is what those passes cost the job). */
int mt3_debug_engine_transcribe(mt3_engine* e, const float* d_inputs, int32_t n_segments, int32_t num_steps, int32_t flags,
                                int32_t poll_steps, int32_t row_groups, int32_t skip_encoder_passes, int32_t* d_ids,
                                mt3_transcribe_stats* h_stats, void* stream);

/* Fill the engine's self-attention K/V caches (and, with fp8 caches, their scale arrays) with the byte `pattern`
 * (0xFF = NaN in bf16 / f32 / e4m3; 0x7F.. etc.), and with cross != 0 also the cross-attention K/V buffers
 * (call it BEFORE mt3_engine_encode then: encode rewrites the rows of its batch).  A decode that follows must
 * return exactly the ids it returns over zero-filled caches. */
int mt3_debug_engine_poison_caches(mt3_engine* e, int32_t pattern, int32_t cross, void* stream);

/* (Rounds 2-3 had thirteen process-wide launch-shape knobs here -- mt3_debug_set_knob -- and a row-group experiment
 * entry, mt3_debug_engine_decode_split.  What they measured is recorded in DESIGN.md sections 3 and 5 and under
 * profiles/r3_ab_*; the variants that lost are no longer compiled into the library, the ones that won are the code.) */

#ifdef __cplusplus
}
#endif
#endif /* MT3_HIP_DEBUG_H_ */
