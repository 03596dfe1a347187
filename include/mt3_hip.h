/*
 * mt3_hip.h -- C ABI of libmt3hip.so, the MI355X (gfx950) engine for the MT3
 * audio -> notes inference path.
 *
 * The reference (magenta/mt3) has no FFI layer: its boundary is a Python class,
 * `InferenceModel` (colab/music_transcription_with_transformers.ipynb, cell
 * "Imports and Definitions"), plus the t5x write_fn in mt3/inference.py:34-138.
 * Each entry point below replaces the *compiled work* behind one reference call;
 * the Python mirror in mt3_amd/ keeps the reference's names on top of it
 * (see INTEGRATION.md for the ctypes binding a reference maintainer would add).
 *
 * Conventions
 *   - every function returns int: 0 = MT3_OK, negative = mt3_status;
 *     mt3_last_error() gives a message for the calling thread.
 *   - no exception crosses the ABI; no torch / C++ types in signatures.
 *   - `d_*` pointers are DEVICE pointers owned by the caller (torch tensors in
 *     the Python mirror); `h_*` are host pointers.  The library never frees or
 *     reallocates caller memory.  Work is enqueued on `stream` (a hipStream_t
 *     passed as void*) and the call returns without synchronising, EXCEPT:
 *     mt3_engine_load_weight / mt3_engine_finalize (setup), the pure-host
 *     functions, and mt3_engine_decode -- which (a) with MT3_DECODE_EARLY_EXIT
 *     waits for the device at every poll, and (b) on the row-group schedule
 *     (batches of >= 128 rows, see "Schedule" there) returns only when the decode
 *     has FINISHED on the device, unless the caller passes MT3_DECODE_ASYNC and
 *     joins with mt3_engine_decode_wait (the caller's thread is free in between);
 *     and mt3_engine_transcribe, which returns when the whole job is done (the
 *     calling thread drives the encoder passes meanwhile).  While they wait for the
 *     device the engine's threads SLEEP (event polls between naps), they do not spin.
 *   - one engine per (device, stream); an engine is not thread-safe, distinct
 *     engines are independent.  An engine owns up to four worker threads (one
 *     per row group; created with the first decode that needs them, joined by
 *     mt3_engine_destroy) and four streams with hardware queues of their own.
 *     Those streams are BLOCKING streams in HIP's sense: work the application
 *     puts on the legacy NULL stream while a decode runs serialises with them
 *     (pass an explicit stream, as every caller of this header does anyway).
 *     mt3_engine_decode must not be called while `stream` is being captured.
 */
#ifndef MT3_HIP_H_
#define MT3_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mt3_status {
  MT3_OK = 0,
  MT3_ERR_INVALID = -1,   /* bad argument / shape / state                      */
  MT3_ERR_HIP = -2,       /* a HIP runtime call failed (no GPU, OOM, launch)   */
  MT3_ERR_CAPACITY = -3,  /* caller's output buffer too small; size returned   */
  MT3_ERR_MISSING = -4    /* weight not loaded / unknown weight name           */
} mt3_status;

const char* mt3_last_error(void);
int mt3_abi_version(void);

/* ------------------------------------------------------------------ frontend
 * Replaces spectrograms.compute_spectrogram -> spectral_ops.compute_logmel
 * (mt3/spectrograms.py:64-73, mt3/spectral_ops.py:29-88) as called per segment by
 * preprocessors.compute_spectrograms (mt3/preprocessors.py:613-618), plus the
 * zero-fill of short segments done later by the feature converter
 * (mt3/models.py:48-98): frames >= n_frames[s] are written as 0.0, not log(eps).
 */
typedef struct mt3_frontend_config {
  int32_t sample_rate;   /* 16000  spectrograms.py:23 */
  int32_t hop_width;     /* 128    spectrograms.py:24 */
  int32_t num_mel_bins;  /* 512    spectrograms.py:25 */
  int32_t fft_size;      /* 2048   spectrograms.py:28 */
  float lo_hz;           /* 20.0   spectrograms.py:29 */
  float hi_hz;           /* 7600.0 spectral_ops.py:79 */
  int32_t table_dtype;   /* arithmetic the Hann window and the mel matrix are BUILT in.  0 (default): float32 in
                            TensorFlow's op order -- tf.signal.stft's window and linear_to_mel_weight_matrix default to
                            dtype=float32 and the reference passes none (spectral_ops.py:42-47,69-71), so this is the side
                            of the <= 2.8e-3 log-domain gap between the two evaluations the reference most likely sits on
                            [TF's op order restated from memory: still unpinned against TensorFlow itself];
                            1: float64, rounded once (rounds 1-4) */
} mt3_frontend_config;

typedef struct mt3_frontend mt3_frontend;

int mt3_frontend_create(const mt3_frontend_config* cfg, mt3_frontend** out);
void mt3_frontend_destroy(mt3_frontend* fe);
/* number of non-zero mel weights and a copy of the dense [fft/2+1, mel] f32 matrix
 * the kernel was built from (for parity tests) */
int mt3_frontend_mel_matrix(const mt3_frontend* fe, float* h_out /*[(fft/2+1)*mel]*/, int64_t* nnz);
/* d_audio  [n_segments, frames_per_segment*hop] f32 (segment s uses its first
 *          h_n_frames[s]*hop samples; the rest is ignored)
 * d_logmel [n_segments, frames_per_segment, num_mel_bins] f32 */
int mt3_frontend_logmel(mt3_frontend* fe, const float* d_audio, int32_t n_segments,
                        int32_t frames_per_segment, const int32_t* h_n_frames /* may be NULL = all full */,
                        float* d_logmel, void* stream);
/* mt3_frontend_logmel passes h_n_frames to the kernel BY VALUE (1024 counts per launch; longer calls are issued in
 * pieces): the caller's host buffer is free when the call returns, nothing is allocated, copied, locked or waited for on
 * the call path, any number of host threads and streams may call, and a stream may be under capture.
 * mt3_frontend_logmel_dev: the same with the counts already in DEVICE memory (caller-owned, must stay valid until the
 * launch has run). */
int mt3_frontend_logmel_dev(mt3_frontend* fe, const float* d_audio, int32_t n_segments,
                            int32_t frames_per_segment, const int32_t* d_n_frames /* may be NULL */,
                            float* d_logmel, void* stream);

/* -------------------------------------------------------------------- engine
 * Replaces network.Transformer (mt3/network.py:265-409, layers in mt3/layers.py)
 * as driven by t5x predict_batch_with_aux through
 * models.ContinuousInputsEncoderDecoderModel (mt3/models.py:121-152):
 * encode once, cross-K/V once, then up to `max_decode_len` cached decode steps.
 */
typedef enum mt3_dtype { MT3_BF16 = 0, MT3_F32 = 1, MT3_FP8_E4M3 = 2 /* K/V caches and the encoder's dense layers only: OCP e4m3fn */ } mt3_dtype;

typedef struct mt3_engine_config {   /* network.T5Config (network.py:25-41), model.gin:47-59 */
  int32_t vocab_size;           /* 1536 (mt3) / 1664 (ismir2021): vocabularies.num_embeddings */
  int32_t emb_dim;              /* 512  */
  int32_t num_heads;            /* 6    */
  int32_t head_dim;             /* 64 (only 64 is supported by the attention kernels) */
  int32_t mlp_dim;              /* 1024 */
  int32_t num_encoder_layers;   /* 8    */
  int32_t num_decoder_layers;   /* 8    */
  int32_t input_depth;          /* 512 = spectrograms.input_depth */
  int32_t input_length;         /* T: 256 (mt3) / 512 (ismir2021) encoder frames */
  int32_t max_decode_len;       /* L: 1024 */
  int32_t max_batch;            /* segments per call the workspaces are sized for */
  int32_t compute_dtype;        /* mt3_dtype: MFMA operand type; accumulation is always f32 */
  int32_t decode_chains;        /* 0/1: one chain; n <= 8: the decode batch is dealt to n independent row groups
                                   that run as parallel branches of the step graph (same results, bit for bit) */
  int32_t kv_cache_dtype;       /* 0: K/V caches in the compute dtype.  MT3_FP8_E4M3 (with MT3_BF16 compute): self- and
                                   cross-attention K/V rows are cached as OCP e4m3 bytes + one power-of-two scale per
                                   (row, head, position) -- half the bytes the HBM-bound decode step streams
                                   (BASELINE configs[4] "fp8 path"; tolerances in DESIGN.md section 4) */
  int32_t dense_dtype;          /* 0: every dense layer in the compute dtype.  MT3_FP8_E4M3 (with MT3_BF16 compute): the
                                   ENCODER's dense layers and the cross-attention K/V projections (the MFMA-bound 99 % of
                                   the encoder's FLOPs) run as MXFP8 -- e4m3 operands with one E8M0 scale per 32 K
                                   elements on v_mfma_scale_f32_16x16x128_f8f6f4, weights quantised once at finalize,
                                   activations by the producing epilogue; the decode step's M = batch GEMMs stay bf16
                                   (they are launch-latency-bound) */
  int32_t options;              /* bit set of MT3_OPT_* below; 0 = the defaults every number in DESIGN.md is quoted on */
} mt3_engine_config;

/* mt3_engine_config.options: numerics-relevant choices of HOW the same function is evaluated (all variants stay inside
 * the tolerances of DESIGN.md section 4; the tests compare them with each other) */
enum {
  /* DECODER: keep the residual stream as ONE f32 stream with in-kernel RMSNorm statistics (the f32 engine always
   * does; the bf16 engine otherwise carries f32 rows + bf16 copy + per-16-column sums of squares) */
  MT3_OPT_SINGLE_RESIDUAL_STREAM = 1,
  /* the same choice for the ENCODER's residual rows (the split form is what feeds the LDS-DMA staged GEMM tile) */
  MT3_OPT_ENCODER_SINGLE_RESIDUAL_STREAM = 4,
  /* decoder: the projections that consume a freshly updated residual row (cross-attention query; next layer's
   * q/k/v) get a launch of their own instead of riding as extra output columns in the neighbouring launches
   * (linearity of the residual update) */
  MT3_OPT_SEPARATE_PROJECTIONS = 2,
  /* decoder: only the q / k / v (+ cross-query) projection of a layer's INPUT row keeps its own launch (otherwise it
   * rides in the previous layer's MLP out-projection launch, and layer 0's comes from two table rows); the
   * cross-attention query stays folded */
  MT3_OPT_SEPARATE_QKV_PROJECTION = 8,
  /* never use the row-group decode schedule (see mt3_engine_decode): every decode stays on the caller's stream */
  MT3_OPT_NO_ROW_GROUPS = 16,
  /* f32 engine: keep the encoder's dense layers and attention on the f32 matrix instruction (v_mfma_f32_16x16x4_f32).  By
   * default they multiply on the bf16 pipes with every f32 operand split EXACTLY into three bf16 terms and the six
   * significant products accumulated in f32 -- at least as exact as the f32 instruction (measured 1.3e-7 against 2.1e-7
   * of sum |p| on a K = 512 dot product) at 2.7x its rate; DESIGN.md section 3 */
  MT3_OPT_ENCODER_F32_MFMA = 32,
  /* host side only (no numerics): the engine's worker threads SPIN while they wait for the device (hipStreamSynchronize,
   * the runtime's own queue back-pressure) as they did up to round 4.  By default a worker keeps at most two windows
   * of 16 decode steps enqueued ahead of the device and, for the older one, POLLS an event (hipEventQuery) between naps of
   * 20 us growing to 200 us (hipEventSynchronize spins on this runtime even for hipEventBlockingSync events, so the nap is
   * explicit); the final wait of a decode and the polls of MT3_DECODE_EARLY_EXIT / mt3_engine_transcribe sleep the same
   * way.  Cost: a completion is noticed up to one nap (<= 200 us) late -- once at the end of a call, and per early-exit
   * poll on the path that decides when to stop (1.25 instead of 5.5 CPU-seconds per 1.10 s decode, same decode time) */
  MT3_OPT_SPIN_WAITS = 64
};

typedef struct mt3_engine mt3_engine;

int mt3_engine_create(const mt3_engine_config* cfg, mt3_engine** out);
void mt3_engine_destroy(mt3_engine* e);
/* Flax parameter names of the reference tree joined by '/', e.g.
 * "encoder/layers_0/attention/query/kernel" (SURVEY.md A.3); h_data is f32,
 * row-major, in the reference's own [in, out] orientation. */
int mt3_engine_load_weight(mt3_engine* e, const char* name, const float* h_data,
                           const int64_t* shape, int32_t ndim);
/* folds norm scales, fuses QKV / gate matrices, converts to compute dtype, uploads */
int mt3_engine_finalize(mt3_engine* e);
int64_t mt3_engine_device_bytes(const mt3_engine* e);

/* Transformer.encode (network.py:275-301) + cross-attention K/V of every decoder
 * layer (layers.py:239-240, hoisted out of the decode loop).
 * d_inputs [batch, T, input_depth] f32.  d_encoded_f32 [batch, T, emb] f32 or NULL.
 * Reproducibility across batch sizes: the f32 engine gives a segment the SAME bits whatever batch it is encoded in (one
 * tile family at every size since round 5).  The bf16 engine has two tile families -- passes of fewer than 2048 rows
 * (8 segments at T = 256) take the decode-sized tiles, whose f32 sums are rounded to bf16 in other places -- so a
 * segment's bf16 encoder output can differ in the last bf16 bit between a pass of < 8 and one of >= 8 segments (both within
 * the bf16 bounds of DESIGN.md section 4); mt3_engine_transcribe pads its chunks to 8 segments for that reason. */
int mt3_engine_encode(mt3_engine* e, const float* d_inputs, int32_t batch,
                      float* d_encoded_f32, void* stream);

/* Autoregressive decode (BOS=0, EOS=1; ids after a row's EOS are 0): the loop t5x
 * `predict_batch_with_aux` drives over Transformer.decode (network.py:303-343).
 * Default: greedy until EOS.  MT3_DECODE_BEAM1: the token selection of t5x
 * `decoding.beam_search` with num_decodes=1, alpha=0.6 (what the reference's
 * InferenceModel.predict_tokens runs): per step the top-2 of log_softmax; the live
 * hypothesis follows the best non-EOS token, an EOS candidate finishes
 * prefix+EOS with score logp/((5+len)/6)^alpha; a row stops once its best finished
 * score exceeds live_logp/((5+L+1)/6)^alpha; the best finished hypothesis is
 * returned, or the live one if none finished.
 * Runs `num_steps` (<= L) steps; each step is one hipGraph replay (per row group) unless
 * flags & MT3_DECODE_NO_GRAPH.  d_ids [batch, L] int32 (columns >= num_steps
 * are zero-filled).  d_first_logits: [batch, vocab] f32 logits of step 0, or NULL.
 * With MT3_DECODE_EARLY_EXIT the host polls a device counter every 32 steps and
 * stops once every row has emitted EOS / finished its search (this synchronises the stream), and finished rows
 * are RETIRED, as the reference's beam_search stops extending finished rows (mt3/models.py:126-127; everything past
 * a row's EOS is cut by _trim_eos anyway, NB:358-363): from the step after a row finishes, the attention kernels stream
 * nothing for it (a wave-uniform exit before the first cache request) and the token kernel skips it; at a poll where
 * the live rows of a row group fit fewer 32-row GEMM tiles than the group occupies, the live rows' per-step state is
 * compacted to the front of the group (a slot -> row map finds their K/V caches, which never move), so attention
 * grids AND the GEMMs' M shrink with the live set.  The ids of every row up to and including its EOS are bit-identical
 * to the schedule without EARLY_EXIT (rows are independent; tests/test_gpu_retire.py).  Without EARLY_EXIT every row
 * runs all `num_steps` steps (the canonical full-length workload bench.py's headline is quoted on).
 * Schedule: a batch of >= 128 rows is decoded as 2 or 4 ROW GROUPS (bf16 operands: 2 from 128 rows, 4 from 512; f32:
 * 2 from 128, 4 from 256 -- with MT3_DECODE_EARLY_EXIT the bf16 rule in f32 as well: the ragged regime is launch latency,
 * two groups measure faster there), each on an engine-owned stream with a hardware queue of its own (created with
 * hipExtStreamCreateWithCUMask and a mask of all compute units: two plain HIP streams serialise)
 * and driven by one of the engine's worker threads (one captured step graph per group, replayed per step;
 * MT3_DECODE_NO_GRAPH: direct launches), so that one group's HBM-bound attention runs beside the other groups'
 * latency-bound GEMMs (+6 % at batch 256 in bf16, +7 % in f32); the groups start after an event on the caller's stream,
 * the ids are bit-identical to the single-stream schedule (rows are independent), and the decode is complete when all
 * groups have FINISHED (each group's thread waits for its stream: a stream nobody waits on runs 7 % slower) -- which is
 * when mt3_engine_decode returns, or, with MT3_DECODE_ASYNC, when mt3_engine_decode_wait does.
 * MT3_DECODE_SINGLE_STREAM / _CHAINS(n), decode_chains > 1 or MT3_OPT_NO_ROW_GROUPS keep everything on `stream`. */
enum {
  MT3_DECODE_NO_GRAPH = 1,
  MT3_DECODE_EARLY_EXIT = 2,
  MT3_DECODE_BEAM1 = 4,
  /* keep the whole decode on the caller's stream (no helper streams): see "Schedule" above */
  MT3_DECODE_SINGLE_STREAM = 8,
  /* return as soon as the decode has been handed to the engine's worker threads; the caller MUST call
   * mt3_engine_decode_wait before it reads d_ids, enqueues anything else on `stream`, or calls any other function of
   * this engine (they fail with MT3_ERR_INVALID while a decode is in flight).  h_steps_run is not written by the
   * asynchronous call (mt3_engine_decode_wait reports it). */
  MT3_DECODE_ASYNC = 16
  /* bits 8..11: number of decode chains for this call (1..8); 0 = the engine's configured default.
   * Any other bit is rejected with MT3_ERR_INVALID (profiling variants live in mt3_hip_debug.h). */
};
#define MT3_DECODE_CHAINS(n) (((n) & 0xF) << 8)
int mt3_engine_decode(mt3_engine* e, int32_t batch, int32_t num_steps, int32_t flags,
                      int32_t* d_ids, float* d_first_logits, int32_t* h_steps_run, void* stream);
/* Joins the decode an MT3_DECODE_ASYNC call started: blocks (sleeping, not spinning) until the engine's worker threads
 * are done, then enqueues the copy of the ids into that call's d_ids on that call's stream (and the beam-1
 * finalisation before it) and reports errors of the decode loop.  MT3_ERR_INVALID when no decode is in flight. */
int mt3_engine_decode_wait(mt3_engine* e, int32_t* h_steps_run);

/* Streaming transcription with IN-FLIGHT BATCHING: encode + decode of n_segments independent segments (any number;
 * the reference's loop over `.batch(8)` calls of predict_batch_with_aux, NB:295-301, mt3/models.py:121-152) through the
 * engine's max_batch decode SLOTS.  The reference's decode is batch-synchronous -- a batch ends when its LAST row has
 * terminated (t5x decoding.beam_search's while_loop) -- so a finished row idles until then; here a finished slot hands
 * its id row to the caller and restarts at position 0 on the next segment: every slot carries its own position counter
 * (as the reference's cache index does per call, mt3/layers.py:246-314), self-attention cache rows past a slot's
 * position are discarded by position, and the segment's cross-attention K/V arrive from an encoder pass that ran ahead
 * on the caller's stream (chunks of up to 64 segments into a staging ring; the first min(n_segments, max_batch)
 * segments are encoded straight into the caches).  Row count per row group is constant, so ONE captured step graph per
 * group serves the whole job; when the queue of segments is empty the remaining rows are retired and compacted as
 * under MT3_DECODE_EARLY_EXIT.
 * d_inputs  [n_segments, T, input_depth] f32 (log-mel);  d_ids [n_segments, L] int32: row i = the ids of segment i,
 * bit-identical to what mt3_engine_encode + mt3_engine_decode(MT3_DECODE_EARLY_EXIT) return for that segment (ids after
 * a row's EOS are 0; a row without EOS has num_steps ids).  flags: MT3_DECODE_BEAM1, MT3_DECODE_NO_GRAPH,
 * MT3_DECODE_SINGLE_STREAM (one row group); anything else is rejected.  The call BLOCKS until every segment is done
 * (the calling thread drives the encoder passes, the engine's workers the row groups); d_inputs / d_ids must stay
 * valid until it returns.  h_stats (may be NULL) reports what ran. */
typedef struct mt3_transcribe_stats {
  int32_t slots;           /* decode slots in use = min(n_segments, max_batch)                                   */
  int32_t groups;          /* row groups                                                                         */
  int32_t steps_run;       /* decode steps of the group that ran longest                                         */
  int32_t polls;           /* refill polls (all groups)                                                          */
  int32_t refills;         /* slots restarted on a new segment (= n_segments - slots)                            */
  int32_t starved_polls;   /* polls at which finished slots found no encoded segment waiting (queue not empty)   */
  int32_t encoder_chunks;  /* encoder passes after the first                                                     */
  int32_t compactions;     /* live-row compactions once the queue was empty                                      */
  int32_t used_graph;      /* 1: every group replayed a captured step graph                                      */
  int32_t reserved[7];
} mt3_transcribe_stats;
int mt3_engine_transcribe(mt3_engine* e, const float* d_inputs, int32_t n_segments, int32_t num_steps, int32_t flags,
                          int32_t* d_ids, mt3_transcribe_stats* h_stats, void* stream);

/* Teacher-forced cached decode: Transformer.decode (mt3/network.py:303-361) on GIVEN decoder inputs, driven one
 * token per call through the same cached step (layers.py:246-314) the autoregressive loop uses -- the input of
 * step 0 is BOS, the input of step t+1 is d_forced_ids[b][t] (i.e. decoder_input_tokens = shift_right(forced),
 * seqio autoregressive_inputs as in mt3/models.py:96).  d_forced_ids [batch, L] int32.  d_step_logits (may be
 * NULL): [num_steps, batch, vocab] f32, the logits of EVERY step (the parity tests compare them with the
 * reference's teacher-forced logits at all cache depths).  d_ids [batch, L]: the arg-max of each step (no EOS
 * bookkeeping).  flags: MT3_DECODE_NO_GRAPH, MT3_DECODE_CHAINS(n); not BEAM1 / EARLY_EXIT. */
int mt3_engine_decode_forced(mt3_engine* e, int32_t batch, int32_t num_steps, int32_t flags,
                             const int32_t* d_forced_ids, float* d_step_logits, int32_t* d_ids, void* stream);

/* Engine facts a caller cannot see from results alone (a negative return is an mt3_status).
 * GRAPH_FALLBACKS: decode calls so far whose step graph could not be captured/instantiated and that therefore
 * ran as direct launches (same ids, slower) -- the fallback is counted, never silent;
 * LAST_DECODE_USED_GRAPH: 1/0 for the most recent decode; RESIDUAL_SPLIT: 1 if the bf16 decode loop carries the
 * residual rows as f32 + bf16 copy + partial sums of squares. */
enum { MT3_STATUS_GRAPH_FALLBACKS = 0, MT3_STATUS_LAST_DECODE_USED_GRAPH = 1, MT3_STATUS_RESIDUAL_SPLIT = 2,
       MT3_STATUS_KV_FP8 = 3, MT3_STATUS_Q_FOLD = 4 /* cross q-projection folded into the neighbouring launches */,
       MT3_STATUS_DENSE_FP8 = 5 /* encoder dense layers on the MXFP8 path */,
       MT3_STATUS_QKV_FOLD = 6 /* the decoder layers' q/k/v projections folded into the preceding launches */,
       MT3_STATUS_LAST_DECODE_GROUPS = 7 /* row groups of the most recent decode (2 or 4: the row-group schedule); 1: on the caller's stream */,
       MT3_STATUS_PARTITION_FALLBACKS = 8 /* decodes that wanted the row-group schedule but could not set it up */,
       MT3_STATUS_LAST_DECODE_COMPACTIONS = 9 /* live-row compactions of the most recent decode (all row groups) */ };
int mt3_engine_status(const mt3_engine* e, int32_t what);

/* GenericTokenVocabulary._decode_tf (mt3/vocabularies.py:241-271): -1 from the
 * first EOS(1) to the end of the row, id-3 for 3 <= id < 3+num_regular, else -2. */
int mt3_ids_to_tokens(const int32_t* d_ids, int32_t batch, int32_t length, int32_t num_regular,
                      int32_t* d_tokens, void* stream);

/* ------------------------------------------------ kernel-level entry points
 * The same kernels the engine launches, exposed one at a time so that parity
 * tests can check each against the oracle.  dtype: mt3_dtype of A/W/out. */
enum { MT3_EPI_STORE = 0, MT3_EPI_RESID = 1, MT3_EPI_GEGLU = 2, MT3_EPI_POS = 3, MT3_EPI_F32 = 4, MT3_EPI_HEADS = 5 };
/* out = epilogue( [rms(A)] * A[M,K] @ Wt[N,K]^T );  a_is_f32: A is f32 (converted on load);
 * norm: multiply rows by rsqrt(mean(A^2)+1e-6) (requires a_is_f32).  aux: pos table (EPI_POS, f32 [T,N]);
 * seq_len: T for EPI_POS / EPI_HEADS; heads for EPI_HEADS = N/(2*64).  small: use the decode-sized tile. */
int mt3_op_gemm(int32_t dtype, const void* d_A, int32_t a_is_f32, int32_t norm, const void* d_Wt,
                void* d_out, int32_t M, int32_t N, int32_t K, int32_t epilogue, const float* d_aux,
                int32_t seq_len, int32_t small, void* stream);
/* The same with the split residual stream: norm = 2 takes A as the COMPUTE-TYPE copy of the
 * rows plus d_a_ss [M][K/16], the exact f32 sums of squares of each 16-column group (what the producer of the rows
 * left), instead of accumulating statistics from an f32 A; with EPI_RESID, d_out_ct / d_out_ss (both or neither)
 * receive the compute-type copy [M][N] and the partial sums [M][N/16] of the updated rows.  small = 0 with bf16
 * operands in memory selects the LDS-DMA staged 128x128x64 tile. */
int mt3_op_gemm_ex(int32_t dtype, const void* d_A, int32_t a_is_f32, int32_t norm, const void* d_Wt,
                   void* d_out, int32_t M, int32_t N, int32_t K, int32_t epilogue, const float* d_aux,
                   int32_t seq_len, int32_t small, const float* d_a_ss, void* d_out_ct, float* d_out_ss,
                   void* stream);
/* x f32 [rows][dim] -> compute-type copy [rows][dim] + per-16-column sums of squares [rows][dim/16] */
int mt3_op_residual_split(int32_t dtype, const float* d_x, void* d_x_ct, float* d_x_ss, int32_t rows, int32_t dim,
                          void* stream);
/* encoder self-attention over qkv [B, T, 3, H, 64] -> out [B, T, H*64]; unscaled logits, softmax f32 */
int mt3_op_encoder_attention(int32_t dtype, const void* d_qkv, void* d_out, int32_t B, int32_t T, int32_t H,
                             void* stream);
/* single-query attention: q [B, H*64] (row stride q_stride elements) against cache K/V [B, H, cap, 64],
 * attending positions 0..n_keys-1; if d_new_kv != NULL its [B, 2, H, 64]-strided K/V rows (row stride
 * kv_stride, K at +0 and V at +H*64) are first appended at position n_keys-1.  With d_step != NULL the key
 * count is read PER ROW from device memory: n_keys = d_step[b] + 1.  The kernel requests its first group of
 * keys before it knows the row's length; what lies past the row's length is discarded BY POSITION (selected away,
 * never multiplied by a zero weight), so cache rows beyond n_keys may hold anything, NaN / Inf patterns included
 * (they only have to be addressable up to `cap`). */
int mt3_op_decode_attention(int32_t dtype, const void* d_q, int32_t q_stride, void* d_kcache, void* d_vcache,
                            int32_t cap, const void* d_new_k, const void* d_new_v, int32_t kv_stride,
                            const int32_t* d_step, int32_t n_keys, void* d_out, int32_t B, int32_t H, void* stream);
/* The same over an fp8 cache (kv_cache_dtype MT3_FP8_E4M3): d_kcache / d_vcache hold OCP e4m3 bytes [B, H, cap, 64],
 * d_kv_scale [B, H, cap] pairs of f32 {k_scale, v_scale} (row value = byte value * scale); q, the new rows and
 * out are bf16.  An appended row is quantised by the kernel (power-of-two scale from the row's amax per head). */
int mt3_op_decode_attention_fp8(const void* d_q, int32_t q_stride, void* d_kcache, void* d_vcache, void* d_kv_scale,
                                int32_t cap, const void* d_new_k, const void* d_new_v, int32_t kv_stride,
                                const int32_t* d_step, int32_t n_keys, void* d_out, int32_t B, int32_t H,
                                void* stream);
/* d_src bf16 [2][rows][64] (K rows, then V rows) -> d_dst e4m3 [2][rows][64] + d_scales [rows] f32 pairs */
int mt3_op_kv_quantize_fp8(const void* d_src, void* d_dst, void* d_scales, int32_t rows, void* stream);

/* MXFP8 dense path (dense_dtype MT3_FP8_E4M3; no counterpart in the reference, whose DenseGeneral is f32,
 * mt3/layers.py:311-360): operands are OCP e4m3fn bytes [rows][K] with one E8M0 power-of-two scale per 32 consecutive
 * K elements [rows][K/32]: scale = 2^(floor(log2 amax) - 7), so amax / scale lies in [128, 256) and nothing
 * saturates; elements are rounded to nearest even.  mt3_host_mx8_quantize is the host (weight) side of that rule,
 * mt3_op_mx8_quantize the device (activation) side: in_is_f32 ? f32 : bf16 rows [M][K], K a multiple of 64;
 * d_ss (f32 input only, may be NULL) receives the per-16-column sums of squares [M][K/16]. */
int mt3_host_mx8_quantize(const float* h_w, int64_t rows, int64_t K, uint8_t* h_q, uint8_t* h_sc);
int mt3_op_mx8_quantize(const void* d_in, int32_t in_is_f32, int32_t M, int32_t K, uint8_t* d_q, uint8_t* d_sc,
                        float* d_ss, void* stream);
/* out = epilogue( [rms] * A[M,K] @ W[N,K]^T ) on v_mfma_scale_f32_16x16x128_f8f6f4, f32 accumulation; K and N
 * multiples of 128.  d_a_ss != NULL: fused RMSNorm from the [M][K/16] sums of squares of the rows A was quantised
 * from (K <= 1024).  epilogue: MT3_EPI_STORE (bf16 d_out [M][N]), MT3_EPI_HEADS (bf16 [2][B][H][seq_len][64]),
 * MT3_EPI_RESID (f32 d_out [M][N] += product; the updated rows also leave as MXFP8 d_out_q / d_out_sc and their
 * per-16-column sums of squares d_out_ss), MT3_EPI_GEGLU (W rows interleaved gate/linear in 16s as for mt3_op_gemm;
 * the result gelu(gate) * linear leaves ONLY as MXFP8 d_out_q [M][N/2] / d_out_sc [M][N/64]). */
int mt3_op_gemm_mx8(const uint8_t* d_A, const uint8_t* d_a_sc, const uint8_t* d_W, const uint8_t* d_w_sc, void* d_out,
                    int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t seq_len, const float* d_a_ss,
                    uint8_t* d_out_q, uint8_t* d_out_sc, float* d_out_ss, void* stream);

/* --------------------------------------------------- symbolic stage (host CPU)
 * Replaces metrics_utils.event_predictions_to_ns (mt3/metrics_utils.py:59-146) =
 * decode_and_combine_predictions + run_length_encoding.decode_events
 * (mt3/run_length_encoding.py:371-423) + the note state machine
 * (mt3/note_sequences.py:262-446) + event_codec.Codec.decode_event_index
 * (mt3/event_codec.py:103-112).  Integer results are bit-exact with the reference;
 * times are computed in double with the reference's own expressions.
 */
enum { MT3_EV_SHIFT = 0, MT3_EV_PITCH = 1, MT3_EV_VELOCITY = 2, MT3_EV_TIE = 3, MT3_EV_PROGRAM = 4, MT3_EV_DRUM = 5 };
enum { MT3_SPEC_ONSETS = 0, MT3_SPEC_NOTES = 1, MT3_SPEC_TIES = 2 };  /* note_sequences.py:416-446 */

typedef struct mt3_event_range { int32_t type, min_value, max_value; } mt3_event_range;
typedef struct mt3_codec {
  double steps_per_second;
  int32_t num_ranges;               /* ranges[0] must be MT3_EV_SHIFT with min 0 */
  mt3_event_range ranges[8];
} mt3_codec;
typedef struct mt3_note {
  double start_time, end_time;
  int32_t pitch, velocity, program, is_drum, instrument, reserved;
} mt3_note;

/* vocabularies.build_codec (mt3/vocabularies.py:119-140) */
int mt3_build_codec(int32_t steps_per_second, int32_t max_shift_seconds, int32_t num_velocity_bins,
                    mt3_codec* out);
int mt3_codec_num_classes(const mt3_codec* c);
int mt3_codec_decode_event(const mt3_codec* c, int32_t index, int32_t* type, int32_t* value);  /* MT3_ERR_INVALID = ValueError */
int mt3_codec_encode_event(const mt3_codec* c, int32_t type, int32_t value, int32_t* index);
/* tokens: concatenated per-segment token rows (already trimmed at EOS); seg_offsets [n_segments+1];
 * h_has_max_time==NULL -> the combiner rule (max_time = next segment's start, none for the last);
 * otherwise explicit per-segment max_time (decode_events' own argument). */
int mt3_notes_decode(const mt3_codec* c, int32_t spec, int32_t n_segments, const int32_t* h_tokens,
                     const int64_t* h_seg_offsets, const double* h_start_times,
                     const int32_t* h_has_max_time, const double* h_max_times,
                     mt3_note* h_notes, int64_t notes_capacity, int64_t* n_notes,
                     int64_t* invalid_events, int64_t* dropped_events, double* total_time);

#ifdef __cplusplus
}
#endif
#endif /* MT3_HIP_H_ */
