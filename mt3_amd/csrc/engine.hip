// The MT3 encoder-decoder engine behind mt3_engine_* (include/mt3_hip.h).
//
// Replaces network.Transformer (mt3/network.py:265-409) as t5x's predict_batch_with_aux drives it
// (mt3/models.py:121-152): `encode` once, then a cached single-token `decode` per step.
//
// MI355X design notes
//   * weights live on the device in the layouts the kernels want, built once at finalize():
//       - every DenseGeneral kernel [in,out] (layers.py:373-418) is stored output-major Wt[out][in];
//       - Q|K|V (layers.py:238-240) are fused into one [3*H*64][emb] matrix, wi_0|wi_1
//         (layers.py:460-468) are interleaved in 16-row groups for the GEGLU epilogue;
//       - every pre-norm scale (network.py:54-56,71,104-106,126-128,142,244) is folded into the
//         rows of the matrix that consumes the normalised activations (the rsqrt part is computed
//         inside that GEMM), so the only standalone norm left is `encoder_norm`.
//   * cross-attention K/V (layers.py:239-240 via network.py:129-135) are computed ONCE per encode
//     for all decoder layers, head-major for the streaming decode kernel; the reference recomputes
//     them inside every decode step unless XLA hoists them.
//   * the self-attention cache is [B][H][L][64] written in place at position t (the reference
//     rewrites the whole [B,H,64,L] cache per step: layers.py:272-292).
//   * one decode step = 8 x 8 + 2 kernels with per-row position counters in DEVICE memory (the argmax
//     kernel also writes the next step's embedding row), captured once per
//     batch size into a hipGraph and replayed L times.  The batch can be dealt to `decode_chains`
//     independent row groups, one graph BRANCH each (fork/join capture over several streams), so that one
//     chain's latency-bound GEMMs run beside another chain's HBM-bound K/V streaming; it paid +6 % with the
//     first kernels and nothing with the current ones, so the default is one chain.
//   * sized for 288 GB HBM: all workspaces for max_batch are allocated up front
//     (B=256: ~3.3 GB KV cache + ~0.9 GB cross K/V + ~0.5 GB activations in bf16).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "feed.h"
#include "kernels.h"

namespace mt3k {
// enc_attention_x6.hip: the f32 engine's encoder attention with Q, K, V and P as three bf16 planes each
int launch_encoder_attention_x6(const void* qkv, void* out, int B, int T, int H, hipStream_t s);
}  // namespace mt3k
#include "mx8.h"
#include "mt3_hip.h"
#include "mt3_hip_debug.h"

namespace {

constexpr int kMaxPos = 2048;   // FixedEmbed.max_length, layers.py:565
constexpr int kMaxChains = 8;

uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                                       // RNE
  return static_cast<uint16_t>(u >> 16);
}

struct HostWeight {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct LayerDev {
  void* wqkv = nullptr;      // [3HD][emb]
  void* wo = nullptr;        // [emb][HD]
  void* wq_x = nullptr;      // decoder cross query [HD][emb]
  void* wkv_x = nullptr;     // decoder cross key|value [2HD][emb]
  void* wo_x = nullptr;      // decoder cross out [emb][HD]
  // q-fold (bf16 decode): the cross-attention q-projection rides in the two neighbouring launches
  void* wqkv_ext = nullptr;  // decoder [3HD + HD][emb]: wqkv rows, then the cross query rows (pre_cross norm scale folded)
  void* wo_ext = nullptr;    // decoder [emb + HD][HD]: self out-projection rows, then (Wo . (s2 * Wq_x))^T
  // qkv-fold (round 3): the NEXT layer's q | k | v | cross-q projections ride in this layer's MLP out-projection launch
  void* w_fold = nullptr;    // decoder [emb + 4HD][mlp + emb]: rows < emb = [Wo_mlp^T | 0]; rows >= emb =
                             // [(Wo_mlp . Wext)^T | Wext^T] with Wext = the next layer's scaled [q | k | v | cross-q]
  void* wi = nullptr;        // [2*mlp][emb] interleaved gate/linear
  void* wo_mlp = nullptr;    // [emb][mlp]
  void* self_k = nullptr;    // decoder [Bm][H][L][64]
  void* self_v = nullptr;
  void* cross_kv = nullptr;  // decoder [2][B][H][T][64]
  // MXFP8 dense path (dense_dtype MT3_FP8_E4M3): the same matrices as e4m3 bytes + E8M0 block scales [rows][K / 32]
  uint8_t *wqkv_q = nullptr, *wqkv_sc = nullptr, *wo_q = nullptr, *wo_sc = nullptr, *wi_q = nullptr, *wi_sc = nullptr,
          *wo_mlp_q = nullptr, *wo_mlp_sc = nullptr, *wkv_x_q = nullptr, *wkv_x_sc = nullptr;
  // f32 engine, encoder-sized launches (gemm_x6_kernel): the same matrices as three bf16 planes (hi / mid / lo) [N][K]
  void *wqkv_p[3] = {}, *wo_p[3] = {}, *wi_p[3] = {}, *wo_mlp_p[3] = {}, *wkv_x_p[3] = {};
  // fp8 (e4m3) K/V caches only: {k_scale, v_scale} per cached row
  float2* self_scale = nullptr;    // [Bm][H][L]
  float2* cross_scale = nullptr;   // [B][H][T]
};

// Variant bits of one decode step (index of a captured step graph): 1 / 2 = mt3_debug_engine_decode's skipped kernels,
// 4 = beam-1 token selection, 8 = teacher forcing, 16 = row retirement (finished slots cost nothing, the slot map is
// in use), 32 = the synthetic EOS schedule (mt3_debug_engine_set_eos_schedule)
// 64 = in-flight batching (mt3_engine_transcribe: finished slots restart on new segments, the slot -> segment map is in use)
constexpr int kVarBeam = 4, kVarForced = 8, kVarRetire = 16, kVarEos = 32, kVarStream = 64, kNumVariants = 128;
// not a step variant of its own (never an index into graph_exec): set in GroupRun::variant when the decode runs as SEVERAL
// row groups, so that a step knows it runs beside other groups' launches (GemmArgs::concurrent) and the group-graph cache
// keeps such steps apart from a lone stream's steps of the same shape
constexpr int kVarBeside = 128;
constexpr int kMaxGroups = 4;
// staging ring of mt3_engine_transcribe: cross-attention K/V of segments that wait for a slot, kStageChunks chunks of up
// to kStageChunkCap segments each (one encoder pass per chunk)
constexpr int kStageChunks = mt3feed::kStageChunks, kStageChunkCap = 64, kStageMinBatch = 8;
constexpr int kStreamPollSteps = 4;     // steps between two refill polls of a row group.  Drained polls, f32, 10,000 ragged segments:
                                        // 32 / 16 / 8 steps -> 2395 / 2417 / 2423 audio-s/s at 1250 slots, 1861 / 1889 / 1905 at 256
                                        // (a finished slot idles half an interval on average, a drained poll costs a bubble of
                                        // ~50 us, more when the host sleeps); the pipelined poll has no bubble and reacts one
                                        // interval later: a slot idles 1.5 intervals on average
constexpr int kThrottleWindow = 16;     // steps per window of the sleeping enqueue throttle (mt3_engine::wait_ev)

// One persistent host thread per row group (created with the first decode that needs it, joined at destroy): a
// decode call hands each group's loop to one of them instead of spawning threads per call, and with
// MT3_DECODE_ASYNC the caller gets its own thread back while they run.
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, quit = false;
};

// a captured decode step of rows [row0, row0 + rows) of a `batch`-row decode
struct GroupGraph {
  int variant, batch, row0, rows, slot, max_len;
  hipGraph_t graph;
  hipGraphExec_t exec;
};

// what mt3_engine_decode left for mt3_engine_decode_wait
struct PendingDecode {
  bool active = false;
  int posted = 0;               // workers that hold a job of this decode
  int groups = 1;
  bool beam1 = false;
  bool used_graph[kMaxGroups] = {};
  int batch = 0;
  int32_t* d_ids = nullptr;
  hipStream_t s = nullptr;
  int rcs[kMaxGroups] = {};
  int ran[kMaxGroups] = {};
  std::string errs[kMaxGroups];
};

}  // namespace

struct mt3_engine {
  mt3_engine_config cfg{};
  std::map<std::string, HostWeight> raw;
  std::vector<void*> allocs;
  int64_t device_bytes = 0;
  bool finalized = false;
  int esize = 2;
  int kv_esize = 2;              // bytes per cached K/V element: esize, or 1 with the fp8 (e4m3) caches
  bool kv_fp8 = false;
  void* cross_stage = nullptr;   // fp8 caches: bf16 [2][Bm][H][T][64] landing buffer of the cross-K/V GEMM (one layer)
  // MXFP8 encoder (gemm_mx8.hip): every GEMM operand of the encoder as e4m3 + E8M0 block scales
  bool dense_fp8 = false;
  uint8_t *x_q = nullptr, *x_sc = nullptr;         // residual rows [M][emb], written by the RESID epilogues
  uint8_t *attn_q = nullptr, *attn_sc = nullptr;   // attention output [M][HD]
  uint8_t *h_q = nullptr, *h_sc = nullptr;         // GEGLU output [M][mlp], written by the GEGLU epilogue
  uint8_t *enc_q = nullptr, *enc_sc = nullptr;     // normed encoder output [M][emb] (A of the cross-K/V projections)

  void* enc_in = nullptr;        // [emb][input_depth]
  void* enc_in_p[3] = {};        // f32 engine: its three bf16 planes
  bool x6 = false;               // f32 engine: the encoder's dense layers multiply on the bf16 pipes (three planes per operand)
  float* enc_norm = nullptr;     // [emb] f32
  float* embedding = nullptr;    // [V][emb] f32
  void* logits_w = nullptr;      // [V][emb] (decoder_norm folded)
  float* pos_table = nullptr;    // [kMaxPos][emb] f32
  std::vector<LayerDev> enc, dec;

  // encoder workspaces
  float* x = nullptr;
  // bf16 path: the encoder's residual rows also travel split (bf16 copy + partial sums of squares), so that every
  // encoder GEMM reads bf16 operands through the LDS-DMA staged tile
  void* x_ct = nullptr;          // [max_batch * T][emb] bf16
  float* x_ss = nullptr;         // [max_batch * T][emb / 16]
  bool x_split = false;
  void* qkv = nullptr;
  void* attn = nullptr;
  void* hbuf = nullptr;
  void* enc_out = nullptr;
  // decode workspaces
  float* y = nullptr;
  // bf16 decode path: compute-type copy of the residual rows + their per-16-column sums of squares, kept up to
  // date by every producer of `y` (embed / argmax / the RESID GEMMs), read by the RMSNorm-fused GEMMs
  void* y_ct = nullptr;          // [max_batch][emb] bf16
  float* y_ss = nullptr;         // [max_batch][emb/16]
  bool y_split = false;
  // q-fold: y_new . Wq' = y_old . Wq' + attn . (Wo . Wq') by linearity, so the first term is 384 extra output columns
  // of the QKV launch (same A operand), the second 384 extra columns of the self out-projection launch (same A
  // operand), and the row's 1/rms is applied by the cross-attention kernel from the partial sums the out-projection's
  // RESID epilogue leaves anyway: 8 launches per step fewer
  float* qf = nullptr;           // [max_batch][HD] f32, unnormalised cross-attention query
  bool q_fold = false;
  // qkv-fold: y_in(l+1) . W = y2(l) . W + h(l) . (Wo_mlp . W) by the same linearity, W = the next layer's
  // [q | k | v | cross-q]: a two-source K = mlp + emb product that rides as 4HD extra output columns in layer l's MLP
  // out-projection launch (kEpiResidS); layer 0's row comes from two table rows (embedding . W, position table . W)
  // written by the kernel that creates the decoder input row.  The self-attention kernel applies 1/rms and rounds
  // q / k / v itself.  8 more launches per step gone.
  bool qkv_fold = false;
  float* qkvf = nullptr;         // [max_batch][4HD] f32 unnormalised q | k | v | cross-q of the current layer's input row
  float* ew0 = nullptr;          // [vocab][4HD] f32: embedding . Wext(layer 0)
  float* pw0 = nullptr;          // [kMaxPos][4HD] f32: position table . Wext(layer 0)
  // the two-source launch reads the compute-type residual rows as an operand while its RESID tiles replace them: the
  // rows alternate between two buffers from layer to layer (bf16: the bf16 copy; f32: the f32 rows themselves)
  void* y_ct_alt = nullptr;
  float* y_alt = nullptr;
  void* qkv_d = nullptr;
  void* attn_d = nullptr;
  void* q_d = nullptr;
  void* h_d = nullptr;
  float* logits = nullptr;
  int* ids = nullptr;
  int* cur_tok = nullptr;
  int* done = nullptr;
  int* step = nullptr;
  int* n_done = nullptr;
  int* h_pinned = nullptr;
  int* forced = nullptr;         // [max_batch][L] teacher-forcing tokens of the current mt3_engine_decode_forced call
  std::atomic<int> graph_fallbacks{0};   // decode loops whose step graph could not be captured (ran as direct launches)
  int last_used_graph = 0;
  // Row retirement (MT3_DECODE_EARLY_EXIT): per-row-of-the-batch state lives in SLOTS; slot_row maps a slot to the row
  // whose caches / ids it works on.  Live slots are compacted to the front of their row group at the early-exit poll
  // (launch_compact), so attention grids and the GEMMs' M shrink with the live set while the caches stay in place.
  int* slot_row = nullptr;       // [max_batch]
  int* eos_at = nullptr;         // [eos_cap] synthetic EOS schedule (mt3_debug_engine_set_eos_schedule): per row, or -- in
                                 // mt3_engine_transcribe -- per segment
  int eos_cap = 0;
  bool eos_on = false;
  // In-flight batching (mt3_engine_transcribe): slot_seg maps a slot to the SEGMENT it is decoding (-1: none); a
  // finished slot hands its id row to the caller's output and restarts on the next encoded segment (launch_refill)
  int* slot_seg = nullptr;       // [max_batch]
  int* cs_seg = nullptr;         // compaction scratch
  int* refill_plan = nullptr;    // [max_batch + kMaxChains]: rows + 1 entries per group
  int stream_max_len = 0;        // steps per segment of the transcribe call in flight (a kernel argument of its step graphs)
  int stage_cap = 0;             // segments per staging chunk (0: staging not allocated yet)
  std::vector<void*> stage_kv;       // per decoder layer: [kStageChunks][2][stage_cap][H][T][64] cache elements
  std::vector<float2*> stage_scale;  // e4m3 caches: [kStageChunks][stage_cap][H][T]
  float* cs_y = nullptr;         // compaction scratch: same shapes as y / y_ct / y_ss / qkvf, 4 ints + 2 floats per slot
  void* cs_y_ct = nullptr;
  float* cs_y_ss = nullptr;
  float* cs_qkvf = nullptr;
  int* cs_int = nullptr;
  float* cs_beam = nullptr;
  int* cs_perm = nullptr;        // [max_batch + kMaxChains]: rows + 1 entries per group
  int compactions = 0;           // compactions of the most recent decode (all groups; written at the end of the decode)
  std::atomic<int> compactions_now{0};
  // step graphs of single row groups (the single-stream schedule is the group [0, batch))
  std::vector<GroupGraph> group_graphs;
  std::mutex graph_mu;
  Worker* workers[kMaxGroups] = {};
  int device = 0;
  PendingDecode pending;
  float* beam_f = nullptr;       // [2][max_batch]: live log-prob | best finished score (MT3_DECODE_BEAM1)
  int* beam_len = nullptr;       // [max_batch] per slot
  int* beam_len_row = nullptr;   // [max_batch] per row (read by the finalisation)
  float* beam_cfg = nullptr;     // [0] brevity penalty of the loop bound, [1 + n] brevity_penalty(n)

  // Row-group decode schedule (round 3): the batch as 2 or 4 row groups (row_groups_for), each on an engine-owned stream
  // with a hardware queue of its own, each driven by its own host thread with direct launches -- one group's HBM-bound
  // attention kernels run beside the other groups' latency-bound GEMMs.  The streams are created with
  // hipExtStreamCreateWithCUMask and a mask of ALL compute units: measured at B = 256 (bf16), two PLAIN streams take
  // 818 ms per 1024-step decode (HIP multiplexes them onto its queue pool and the groups serialise), two masked ones
  // 588 ms whether the masks are disjoint halves, overlap, or cover every CU (one graph-replayed chain: 626 ms).
  hipStream_t part_stream[kMaxGroups] = {};
  hipEvent_t part_begin = nullptr;
  // Sleeping waits (round 5): a group's worker keeps at most two windows of kThrottleWindow steps enqueued ahead of the
  // device and waits for the older one on a BLOCKING-SYNC event -- the thread sleeps until the interrupt instead of
  // spinning in the runtime's queue back-pressure / hipStreamSynchronize (round 4: five cores busy for the length of a
  // decode).  Index kMaxGroups = the caller's stream (single-stream schedule, the encoder passes of transcribe).
  // events the workers NAP-POLL (sleep_until: hipEventQuery between 20..200 us naps; not blocking-sync waits):
  hipEvent_t wait_ev[kMaxGroups + 1][4] = {};     // [0] waits, [1] throttle, [2], [3] the two poll snapshots of transcribe
  bool spin_waits = false;       // MT3_OPT_SPIN_WAITS: round 4's behaviour
  int part_failed = 0;           // partitioned decodes that fell back to the single-stream schedule (stream creation failed)
  int last_groups = 1;           // row groups of the most recent decode

  int cur_batch = 0;             // batch of the last encode
  hipStream_t cap_stream[8] = {};     // one capture stream per chain (kMaxChains)
  hipEvent_t cap_event[8] = {};
  // one captured decode step per (batch, variant); variant bits: 1 = no self-attention, 2 = no
  // cross-attention (differential profiling only), 4 = beam-1 token selection, 8 = teacher forcing
  hipGraphExec_t graph_exec[kNumVariants][9] = {};   // [variant][chains]: the step graph with `chains` parallel branches
  hipGraph_t graph[kNumVariants][9] = {};
  int graph_batch = 0;

  int HD() const { return cfg.num_heads * cfg.head_dim; }
};

namespace {

// t5x decoding.brevity_penalty(alpha = 0.6, length): ((5 + length) / 6) ^ alpha
float brevity_penalty(int length) { return static_cast<float>(std::pow((5.0 + length) / 6.0, 0.6)); }

int dmalloc(mt3_engine* e, void** p, size_t bytes) {
  MT3_HIP_CHECK(hipMalloc(p, bytes));
  e->allocs.push_back(*p);
  e->device_bytes += static_cast<int64_t>(bytes);
  return MT3_OK;
}

// upload a host f32 matrix as the compute type
int upload_ct(mt3_engine* e, const std::vector<float>& h, void** d) {
  int rc = dmalloc(e, d, h.size() * e->esize);
  if (rc) return rc;
  if (e->cfg.compute_dtype == MT3_BF16) {
    std::vector<uint16_t> t(h.size());
    for (size_t i = 0; i < h.size(); ++i) t[i] = f32_to_bf16_bits(h[i]);
    MT3_HIP_CHECK(hipMemcpy(*d, t.data(), t.size() * 2, hipMemcpyHostToDevice));
  } else {
    MT3_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  return MT3_OK;
}

// upload a host f32 matrix [rows][K] as MXFP8 (only when the engine runs the MXFP8 dense path)
int upload_mx8(mt3_engine* e, const std::vector<float>& h, int64_t rows, int64_t K, uint8_t** q, uint8_t** sc) {
  if (!e->dense_fp8) return MT3_OK;
  std::vector<uint8_t> hq(h.size()), hs(static_cast<size_t>(rows) * (K / 32));
  int rc = mt3_host_mx8_quantize(h.data(), rows, K, hq.data(), hs.data());
  if (rc) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(q), hq.size()))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(sc), hs.size()))) return rc;
  MT3_HIP_CHECK(hipMemcpy(*q, hq.data(), hq.size(), hipMemcpyHostToDevice));
  MT3_HIP_CHECK(hipMemcpy(*sc, hs.data(), hs.size(), hipMemcpyHostToDevice));
  return MT3_OK;
}

// upload a host f32 matrix as its three bf16 planes: hi = rne(w), mid = rne(w - hi), lo = rne(w - hi - mid) (both
// differences exact); only when the engine multiplies f32 operands that way (mt3_engine::x6)
int upload_planes(mt3_engine* e, const std::vector<float>& h, void* (&p)[3]) {
  if (!e->x6) return MT3_OK;
  std::vector<uint16_t> pl[3];
  for (auto& v : pl) v.resize(h.size());
  for (size_t i = 0; i < h.size(); ++i) {
    auto val = [](uint16_t b) {
      const uint32_t u = static_cast<uint32_t>(b) << 16;
      float f;
      std::memcpy(&f, &u, 4);
      return f;
    };
    const uint16_t hi = f32_to_bf16_bits(h[i]);
    const float r1 = h[i] - val(hi);
    const uint16_t mi = f32_to_bf16_bits(r1);
    const float r2 = r1 - val(mi);
    pl[0][i] = hi;
    pl[1][i] = mi;
    pl[2][i] = f32_to_bf16_bits(r2);
  }
  for (int k = 0; k < 3; ++k) {
    int rc = dmalloc(e, &p[k], h.size() * 2);
    if (rc) return rc;
    MT3_HIP_CHECK(hipMemcpy(p[k], pl[k].data(), h.size() * 2, hipMemcpyHostToDevice));
  }
  return MT3_OK;
}

int upload_f32(mt3_engine* e, const std::vector<float>& h, float** d) {
  int rc = dmalloc(e, reinterpret_cast<void**>(d), h.size() * 4);
  if (rc) return rc;
  MT3_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  return MT3_OK;
}

const HostWeight* find(mt3_engine* e, const std::string& name, int64_t d0, int64_t d1) {
  auto it = e->raw.find(name);
  if (it == e->raw.end()) {
    mt3::fail(MT3_ERR_MISSING, "weight not loaded: " + name);
    return nullptr;
  }
  const HostWeight& w = it->second;
  const bool ok = (d1 < 0) ? (w.shape.size() == 1 && w.shape[0] == d0)
                           : (w.shape.size() == 2 && w.shape[0] == d0 && w.shape[1] == d1);
  if (!ok) {
    mt3::fail(MT3_ERR_INVALID, "weight has the wrong shape: " + name);
    return nullptr;
  }
  return &w;
}

// dst rows [row0, row0+out) of an output-major matrix with `in` columns  <-  W[in][out] (* scale[in])
void put_transposed(std::vector<float>& dst, int in, int row0, const HostWeight& w, const float* scale) {
  const int out = static_cast<int>(w.shape[1]);
  for (int k = 0; k < in; ++k) {
    const float s = scale ? scale[k] : 1.f;
    const float* src = w.data.data() + static_cast<size_t>(k) * out;
    for (int n = 0; n < out; ++n) dst[static_cast<size_t>(row0 + n) * in + k] = src[n] * s;
  }
}

int build_attention(mt3_engine* e, const std::string& prefix, const float* scale, bool cross, LayerDev* L,
                    bool encoder = false) {
  const int emb = e->cfg.emb_dim, hd = e->HD();
  const HostWeight *q = find(e, prefix + "/query/kernel", emb, hd), *k = find(e, prefix + "/key/kernel", emb, hd),
                   *v = find(e, prefix + "/value/kernel", emb, hd), *o = find(e, prefix + "/out/kernel", hd, emb);
  if (!q || !k || !v || !o) return MT3_ERR_MISSING;
  int rc;
  std::vector<float> ot(static_cast<size_t>(emb) * hd);
  put_transposed(ot, hd, 0, *o, nullptr);
  if (!cross) {
    std::vector<float> t(static_cast<size_t>(3) * hd * emb);
    put_transposed(t, emb, 0, *q, scale);
    put_transposed(t, emb, hd, *k, scale);
    put_transposed(t, emb, 2 * hd, *v, scale);
    if ((rc = upload_ct(e, t, &L->wqkv))) return rc;
    if ((rc = upload_ct(e, ot, &L->wo))) return rc;
    if (encoder) {
      if ((rc = upload_planes(e, t, L->wqkv_p))) return rc;
      if ((rc = upload_planes(e, ot, L->wo_p))) return rc;
      if ((rc = upload_mx8(e, t, 3 * hd, emb, &L->wqkv_q, &L->wqkv_sc))) return rc;
      if ((rc = upload_mx8(e, ot, emb, hd, &L->wo_q, &L->wo_sc))) return rc;
    }
  } else {
    std::vector<float> tq(static_cast<size_t>(hd) * emb), tkv(static_cast<size_t>(2) * hd * emb);
    put_transposed(tq, emb, 0, *q, scale);         // query side sees the pre_cross_attention norm
    put_transposed(tkv, emb, 0, *k, nullptr);      // key/value side sees `encoded` (already normed)
    put_transposed(tkv, emb, hd, *v, nullptr);
    if ((rc = upload_ct(e, tq, &L->wq_x))) return rc;
    if ((rc = upload_ct(e, tkv, &L->wkv_x))) return rc;
    if ((rc = upload_planes(e, tkv, L->wkv_x_p))) return rc;
    if ((rc = upload_mx8(e, tkv, 2 * hd, emb, &L->wkv_x_q, &L->wkv_x_sc))) return rc;
    if ((rc = upload_ct(e, ot, &L->wo_x))) return rc;
  }
  return MT3_OK;
}

// q-fold matrices of one decoder layer (see mt3_engine::qf)
int build_q_fold(mt3_engine* e, const std::string& P, const float* s1, const float* s2, LayerDev* L) {
  const int emb = e->cfg.emb_dim, hd = e->HD();
  const std::string S = P + "/self_attention", X = P + "/encoder_decoder_attention";
  const HostWeight *q = find(e, S + "/query/kernel", emb, hd), *k = find(e, S + "/key/kernel", emb, hd),
                   *v = find(e, S + "/value/kernel", emb, hd), *o = find(e, S + "/out/kernel", hd, emb),
                   *qx = find(e, X + "/query/kernel", emb, hd);
  if (!q || !k || !v || !o || !qx) return MT3_ERR_MISSING;
  std::vector<float> t(static_cast<size_t>(4) * hd * emb);
  put_transposed(t, emb, 0, *q, s1);
  put_transposed(t, emb, hd, *k, s1);
  put_transposed(t, emb, 2 * hd, *v, s1);
  put_transposed(t, emb, 3 * hd, *qx, s2);
  int rc;
  if ((rc = upload_ct(e, t, &L->wqkv_ext))) return rc;
  // rows [0, emb): Wo^T;  rows [emb, emb + hd): P^T with P[k][n] = sum_e Wo[k][e] * s2[e] * Wq_x[e][n]  (double)
  std::vector<float> u(static_cast<size_t>(emb + hd) * hd);
  put_transposed(u, hd, 0, *o, nullptr);
  std::vector<double> acc(hd);
  for (int kk = 0; kk < hd; ++kk) {
    std::fill(acc.begin(), acc.end(), 0.0);
    const float* orow = o->data.data() + static_cast<size_t>(kk) * emb;
    for (int ee = 0; ee < emb; ++ee) {
      const double w = static_cast<double>(orow[ee]) * s2[ee];
      const float* qrow = qx->data.data() + static_cast<size_t>(ee) * hd;
      for (int n = 0; n < hd; ++n) acc[n] += w * qrow[n];
    }
    for (int n = 0; n < hd; ++n) u[static_cast<size_t>(emb + n) * hd + kk] = static_cast<float>(acc[n]);
  }
  return upload_ct(e, u, &L->wo_ext);
}

mt3k::GemmArgs gemm_args(const void* A, const void* Wt, void* out, int M, int N, int K, int ldo);
const float* scale_of(mt3_engine* e, const std::string& name);

// qkv-fold matrices and tables (see mt3_engine::qkv_fold).  The matrix products (Wo_mlp . Wext per layer: 0.8 GFLOP;
// embedding / position table . Wext of layer 0) run on the device with the engine's own f32 GEMM (exact f32 products,
// f32 accumulation: ~1e-6 of an entry, far below what the operand formats keep) instead of seconds of host loops.
int build_qkv_fold(mt3_engine* e) {
  const mt3_engine_config& c = e->cfg;
  const int emb = c.emb_dim, hd = e->HD(), mlp = c.mlp_dim, n4 = 4 * hd, nl = c.num_decoder_layers;
  // Wext^T of layer l: output-major [4HD][emb] = scaled q | k | v of the self-attention, scaled cross-attention query
  auto wext_t = [&](int l, std::vector<float>* t) -> int {
    const std::string P = "decoder/layers_" + std::to_string(l);
    const float* s1 = scale_of(e, P + "/pre_self_attention_layer_norm/scale");
    const float* s2 = scale_of(e, P + "/pre_cross_attention_layer_norm/scale");
    const HostWeight *q = find(e, P + "/self_attention/query/kernel", emb, hd),
                     *k = find(e, P + "/self_attention/key/kernel", emb, hd),
                     *v = find(e, P + "/self_attention/value/kernel", emb, hd),
                     *qx = find(e, P + "/encoder_decoder_attention/query/kernel", emb, hd);
    if (!s1 || !s2 || !q || !k || !v || !qx) return MT3_ERR_MISSING;
    t->assign(static_cast<size_t>(n4) * emb, 0.f);
    put_transposed(*t, emb, 0, *q, s1);
    put_transposed(*t, emb, hd, *k, s1);
    put_transposed(*t, emb, 2 * hd, *v, s1);
    put_transposed(*t, emb, 3 * hd, *qx, s2);
    return MT3_OK;
  };
  float *d_wext = nullptr, *d_wo = nullptr, *d_prod = nullptr;
  auto cleanup = [&]() {
    if (d_wext) (void)hipFree(d_wext);
    if (d_wo) (void)hipFree(d_wo);
    if (d_prod) (void)hipFree(d_prod);
  };
  int rc = MT3_OK;
  const int n_max = n4 > c.vocab_size ? n4 : c.vocab_size;
  hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_wext), static_cast<size_t>(n_max) * emb * 4);
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&d_wo), static_cast<size_t>(mlp) * emb * 4);
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&d_prod), static_cast<size_t>(n_max) * mlp * 4);
  std::vector<float> wt, prod, w;
  // layer l's MLP out-projection launch carries the NEXT consumer of its output row: layer l + 1's q | k | v | cross-q
  // (4HD columns), or -- last layer -- the logits projection (vocab columns, decoder_norm scale folded)
  for (int l = 0; l < nl && he == hipSuccess && rc == MT3_OK; ++l) {
    const bool last = l + 1 == nl;
    const int nx = last ? c.vocab_size : n4;
    const HostWeight* wo = find(e, "decoder/layers_" + std::to_string(l) + "/mlp/wo/kernel", mlp, emb);
    if (!wo) rc = MT3_ERR_MISSING;
    if (rc == MT3_OK && !last) rc = wext_t(l + 1, &wt);
    if (rc == MT3_OK && last) {
      const float* sn = scale_of(e, "decoder/decoder_norm/scale");
      const HostWeight* wl = find(e, "decoder/logits_dense/kernel", emb, c.vocab_size);
      if (!sn || !wl) rc = MT3_ERR_MISSING;
      else {
        wt.assign(static_cast<size_t>(nx) * emb, 0.f);
        put_transposed(wt, emb, 0, *wl, sn);
      }
    }
    if (rc != MT3_OK) break;
    he = hipMemcpy(d_wext, wt.data(), wt.size() * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(d_wo, wo->data.data(), wo->data.size() * 4, hipMemcpyHostToDevice);
    if (he != hipSuccess) break;
    // prod[n][k] = sum_e Wext^T[n][e] * Wo_mlp[k][e]: "A" = Wext^T rows, "Wt" = Wo_mlp rows (K = emb for both)
    mt3k::GemmArgs g = gemm_args(d_wext, d_wo, d_prod, nx, mlp, emb, mlp);
    rc = mt3k::launch_gemm(MT3_F32, g, false, 0, MT3_EPI_F32, false, nullptr);
    if (rc != MT3_OK) break;
    prod.resize(static_cast<size_t>(nx) * mlp);
    he = hipMemcpy(prod.data(), d_prod, prod.size() * 4, hipMemcpyDeviceToHost);
    if (he != hipSuccess) break;
    const size_t K = static_cast<size_t>(mlp) + emb;
    w.assign((static_cast<size_t>(emb) + nx) * K, 0.f);
    for (int k = 0; k < mlp; ++k)
      for (int n = 0; n < emb; ++n) w[static_cast<size_t>(n) * K + k] = wo->data[static_cast<size_t>(k) * emb + n];
    for (int n = 0; n < nx; ++n) {
      float* row = w.data() + (static_cast<size_t>(emb) + n) * K;
      std::memcpy(row, prod.data() + static_cast<size_t>(n) * mlp, static_cast<size_t>(mlp) * 4);
      std::memcpy(row + mlp, wt.data() + static_cast<size_t>(n) * emb, static_cast<size_t>(emb) * 4);
    }
    rc = upload_ct(e, w, &e->dec[l].w_fold);
  }
  // layer 0: its input row is Embed(tok) + FixedEmbed[t], so its projection is the sum of two table rows
  if (he == hipSuccess && rc == MT3_OK) rc = wext_t(0, &wt);
  if (he == hipSuccess && rc == MT3_OK) {
    he = hipMemcpy(d_wext, wt.data(), wt.size() * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) rc = dmalloc(e, reinterpret_cast<void**>(&e->ew0), static_cast<size_t>(c.vocab_size) * n4 * 4);
    if (he == hipSuccess && rc == MT3_OK)
      rc = dmalloc(e, reinterpret_cast<void**>(&e->pw0), static_cast<size_t>(kMaxPos) * n4 * 4);
    if (he == hipSuccess && rc == MT3_OK)
      rc = mt3k::launch_gemm(MT3_F32, gemm_args(e->embedding, d_wext, e->ew0, c.vocab_size, n4, emb, n4), false, 0,
                             MT3_EPI_F32, false, nullptr);
    if (he == hipSuccess && rc == MT3_OK)
      rc = mt3k::launch_gemm(MT3_F32, gemm_args(e->pos_table, d_wext, e->pw0, kMaxPos, n4, emb, n4), false, 0,
                             MT3_EPI_F32, false, nullptr);
    if (he == hipSuccess && rc == MT3_OK) he = hipDeviceSynchronize();
  }
  cleanup();
  if (rc != MT3_OK) return rc;
  if (he != hipSuccess) return mt3::fail(MT3_ERR_HIP, std::string("qkv-fold tables: ") + hipGetErrorString(he));
  return MT3_OK;
}

int build_mlp(mt3_engine* e, const std::string& prefix, const float* scale, LayerDev* L, bool encoder = false) {
  const int emb = e->cfg.emb_dim, mlp = e->cfg.mlp_dim;
  const HostWeight *w0 = find(e, prefix + "/wi_0/kernel", emb, mlp), *w1 = find(e, prefix + "/wi_1/kernel", emb, mlp),
                   *wo = find(e, prefix + "/wo/kernel", mlp, emb);
  if (!w0 || !w1 || !wo) return MT3_ERR_MISSING;
  // rows [32q, 32q+16) = gate columns 16q.., rows [32q+16, 32q+32) = linear columns 16q..
  std::vector<float> t(static_cast<size_t>(2) * mlp * emb);
  for (int k = 0; k < emb; ++k) {
    const float s = scale[k];
    for (int n = 0; n < mlp; ++n) {
      const int q = n >> 4, r = n & 15;
      t[static_cast<size_t>(32 * q + r) * emb + k] = w0->data[static_cast<size_t>(k) * mlp + n] * s;
      t[static_cast<size_t>(32 * q + 16 + r) * emb + k] = w1->data[static_cast<size_t>(k) * mlp + n] * s;
    }
  }
  std::vector<float> ot(static_cast<size_t>(emb) * mlp);
  put_transposed(ot, mlp, 0, *wo, nullptr);
  int rc;
  if ((rc = upload_ct(e, t, &L->wi))) return rc;
  if (encoder) {
    if ((rc = upload_planes(e, t, L->wi_p))) return rc;
    if ((rc = upload_planes(e, ot, L->wo_mlp_p))) return rc;
    if ((rc = upload_mx8(e, t, 2 * mlp, emb, &L->wi_q, &L->wi_sc))) return rc;
    if ((rc = upload_mx8(e, ot, emb, mlp, &L->wo_mlp_q, &L->wo_mlp_sc))) return rc;
  }
  return upload_ct(e, ot, &L->wo_mlp);
}

const float* scale_of(mt3_engine* e, const std::string& name) {
  const HostWeight* w = find(e, name, e->cfg.emb_dim, -1);
  return w ? w->data.data() : nullptr;
}

mt3k::GemmArgs gemm_args(const void* A, const void* Wt, void* out, int M, int N, int K, int ldo) {
  mt3k::GemmArgs g{};
  g.A = A;
  g.Wt = Wt;
  g.out = out;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = K;
  g.ldo = ldo;
  return g;
}

#define MT3_TRY(expr)           \
  do {                          \
    int _rc = (expr);           \
    if (_rc != MT3_OK) return _rc; \
  } while (0)

// One decode step of rows [row0, row0 + rows) -- a "chain".  The batch of a decode call is split into
// `e->chains` such chains; they are data-independent (a segment never looks at another segment), so
// the step graph has one branch per chain and the GPU overlaps one chain's latency-bound GEMMs with
// another chain's HBM-bound attention streams.  Every row-indexed workspace is simply offset.
// Op `op` of the decode step of rows [row0, row0 + rows): per decoder layer 8 ops
//   0 QKV GEMM (fused RMSNorm)   1 self-attention (+cache append)   2 out-proj + residual
//   3 q-proj (fused RMSNorm)     4 cross-attention                  5 out-proj + residual
//   6 GEGLU GEMM (fused RMSNorm) 7 wo + residual
// then  8*L: logits GEMM (fused decoder_norm),  8*L + 1: argmax / EOS / position += 1 / next input row.
// Ops 1 and 4 are the HBM-streaming ("heavy") ones; everything else is latency-bound.
inline int chain_num_ops(const mt3_engine* e) { return 8 * e->cfg.num_decoder_layers + 2; }
inline bool chain_op_is_heavy(const mt3_engine* e, int op) {
  return op < 8 * e->cfg.num_decoder_layers && ((op & 7) == 1 || (op & 7) == 4);
}

int enqueue_chain_op(mt3_engine* e, int row0, int rows, int B_total, int skip, int op, hipStream_t s, int done_slot = 0) {
  const mt3_engine_config& c = e->cfg;
  const int dt = c.compute_dtype, emb = c.emb_dim, hd = e->HD(), H = c.num_heads, T = c.input_length;
  const int Lmax = c.max_decode_len;
  const size_t es = e->esize, kes = e->kv_esize;
  const bool small = true;
  const int nl = c.num_decoder_layers;
  const bool split = e->y_split, fold = e->qkv_fold;
  // row retirement: the step's slots [row0, row0 + rows) reach their cache rows / id rows through e->slot_row, so the
  // cache and id base pointers stay those of the whole batch
  const bool retire = (skip & kVarRetire) != 0;
  const size_t crow0 = retire ? 0 : static_cast<size_t>(row0);
  // Which of the two residual buffers is current (qkv-fold only: the two-source launch at the end of layer l reads the
  // rows it replaces, so the rows move to the other buffer there): layer l works on buffer l & 1; the logits and the
  // arg-max read the last layer's, the arg-max writes the next step's input row into buffer 0.
  const int layer = op < 8 * nl ? (op >> 3) : nl - 1;
  const int cur = fold ? (layer & 1) : 0;
  const bool f32 = dt == MT3_F32;
  auto y_buf = [&](int which) -> float* {            // f32 residual rows
    return ((fold && f32 && which) ? e->y_alt : e->y) + static_cast<size_t>(row0) * emb;
  };
  auto yct_buf = [&](int which) -> char* {           // compute-type rows the norm-fused GEMMs read (f32: the rows themselves)
    if (!split) return nullptr;
    if (f32) return reinterpret_cast<char*>(y_buf(which));
    return static_cast<char*>((fold && which) ? e->y_ct_alt : e->y_ct) + static_cast<size_t>(row0) * emb * es;
  };
  // the decoder input row of this step (Embed(tok) + FixedEmbed[t]) is already in `y`: written by the
  // embed launch before the first step and by the previous step's argmax kernel afterwards
  float* y = y_buf(cur);
  // residual rows as the norm-fused GEMMs see them: f32 with in-kernel statistics, or (split form) the
  // compute-type rows with the producer's partial sums
  char* y_ct = yct_buf(cur);
  char* y_copy = split && !f32 ? y_ct : nullptr;     // where producers of a residual row leave its bf16 copy
  float* y_ss = split ? e->y_ss + static_cast<size_t>(row0) * (emb / 16) : nullptr;
  auto normed = [&](const void* Wt, void* out, int N, int ldo) {
    mt3k::GemmArgs g = gemm_args(split ? static_cast<const void*>(y_ct) : static_cast<const void*>(y), Wt, out, rows, N,
                                 emb, ldo);
    g.a_ss = y_ss;
    return g;
  };
  // this step belongs to one of several row groups / graph chains (GemmArgs::concurrent).  Under row retirement
  // rows < B_total also holds for a LONE stream after a compaction, hence the explicit bit
  const int beside = ((skip & kVarBeside) || (!retire && rows < B_total)) ? 1 : 0;
  auto resid = [&](const void* A, const void* Wt, int K) {
    mt3k::GemmArgs g = gemm_args(A, Wt, y, rows, emb, K, emb);
    g.out_ct = y_copy;
    g.out_ss = y_ss;
    g.concurrent = beside;
    return g;
  };
  const int nrm = split ? 2 : 1;
  char* qkv_d = static_cast<char*>(e->qkv_d) + static_cast<size_t>(row0) * 3 * hd * es;
  char* attn_d = static_cast<char*>(e->attn_d) + static_cast<size_t>(row0) * hd * es;
  char* q_d = static_cast<char*>(e->q_d) + static_cast<size_t>(row0) * hd * es;
  char* h_d = static_cast<char*>(e->h_d) + static_cast<size_t>(row0) * c.mlp_dim * es;
  float* qkvf = fold ? e->qkvf + static_cast<size_t>(row0) * 4 * hd : nullptr;
  float* logits = e->logits + static_cast<size_t>(row0) * c.vocab_size;
  int* step = e->step + row0;                    // per-row position counters
  if (op == 8 * nl) {
    if (fold) return MT3_OK;                  // the logits projection rode in the last layer's MLP out-projection launch
    return mt3k::launch_gemm(dt, normed(e->logits_w, logits, c.vocab_size, c.vocab_size), !split, nrm, MT3_EPI_F32,
                             small, s);
  }
  if (op == 8 * nl + 1) {
    const mt3k::BeamState beam{e->beam_f + row0, e->beam_len + row0, e->beam_cfg, c.max_batch, e->beam_len_row + crow0};
    const mt3k::RowProj rp{e->ew0, e->pw0, qkvf, 4 * hd};
    // folded logits arrive unnormalised: the row scale comes from the final residual row's partial sums
    const mt3k::LogitScale ls{fold ? y_ss : nullptr, emb / 16, emb};
    const bool streaming = (skip & kVarStream) != 0;
    const mt3k::StepRetire rt{retire ? 1 : 0, retire ? e->slot_row + row0 : nullptr,
                              (skip & kVarEos) ? e->eos_at + crow0 : nullptr, streaming ? e->slot_seg + row0 : nullptr,
                              streaming ? e->stream_max_len : 0};
    return mt3k::launch_argmax_step(logits, c.vocab_size, e->ids + crow0 * Lmax, Lmax,
                                    e->cur_tok + row0, e->done + row0, e->n_done + done_slot, step, e->embedding, e->pos_table,
                                    kMaxPos, y_buf(0), split && !f32 ? yct_buf(0) : nullptr, y_ss, emb, rows,
                                    (skip & kVarBeam) ? &beam : nullptr,
                                    (skip & kVarForced) ? e->forced + static_cast<size_t>(row0) * Lmax : nullptr, Lmax, rp, ls,
                                    rt, s);
  }
  LayerDev& L = e->dec[op >> 3];
  switch (op & 7) {
    case 0:
      if (fold) return MT3_OK;                // rode in the previous layer's MLP out-projection launch / the row's producer
      if (e->q_fold) {
        mt3k::GemmArgs g = normed(L.wqkv_ext, qkv_d, 4 * hd, 3 * hd);
        g.out2 = e->qf + static_cast<size_t>(row0) * hd;
        g.n_split = 3 * hd;
        return mt3k::launch_gemm(dt, g, false, 2, mt3k::kEpiStoreQ, small, s);
      }
      return mt3k::launch_gemm(dt, normed(L.wqkv, qkv_d, 3 * hd, 3 * hd), !split, nrm, MT3_EPI_STORE, small, s);
    case 1: {
      if (skip & 1) return MT3_OK;
      mt3k::DecAttnArgs a{};
      a.q = qkv_d;
      a.q_stride = 3 * hd;
      a.kcache = static_cast<char*>(L.self_k) + crow0 * H * Lmax * 64 * kes;
      a.vcache = static_cast<char*>(L.self_v) + crow0 * H * Lmax * 64 * kes;
      a.kv_scale = e->kv_fp8 ? L.self_scale + crow0 * H * Lmax : nullptr;
      if (retire) {
        a.done = e->done + row0;
        a.cache_row = e->slot_row + row0;
      }
      a.cap = Lmax;
      a.new_k = qkv_d + static_cast<size_t>(hd) * es;
      a.new_v = qkv_d + static_cast<size_t>(2 * hd) * es;
      a.kv_stride = 3 * hd;
      if (fold) {                             // unnormalised f32 q | k | v + the input row's partial sums of squares
        a.q = nullptr;
        a.q_f32 = qkvf;
        a.q_stride = a.kv_stride = 4 * hd;
        a.new_k = qkvf + hd;
        a.new_v = qkvf + 2 * hd;
        a.q_ss = y_ss;
        a.q_ss_n = emb / 16;
      }
      a.step = step;
      a.out = attn_d;
      a.B = rows;
      a.H = H;
      return mt3k::launch_decode_attention(dt, a, s);
    }
    case 2:
      if (e->q_fold) {
        mt3k::GemmArgs g = resid(attn_d, L.wo_ext, hd);
        g.N = emb + hd;
        g.out2 = fold ? qkvf + 3 * hd : e->qf + static_cast<size_t>(row0) * hd;
        g.ld2 = fold ? 4 * hd : 0;
        g.n_split = emb;
        return mt3k::launch_gemm(dt, g, false, 0, mt3k::kEpiResidQ, small, s);
      }
      return mt3k::launch_gemm(dt, resid(attn_d, L.wo, hd), false, 0, MT3_EPI_RESID, small, s);
    case 3:
      if (e->q_fold) return MT3_OK;           // folded into ops 0 and 2
      return mt3k::launch_gemm(dt, normed(L.wq_x, q_d, hd, hd), !split, nrm, MT3_EPI_STORE, small, s);
    case 4: {
      if (skip & 2) return MT3_OK;
      mt3k::DecAttnArgs x{};
      x.q = q_d;
      x.q_stride = hd;
      if (e->q_fold) {
        x.q_f32 = fold ? qkvf + 3 * hd : e->qf + static_cast<size_t>(row0) * hd;
        if (fold) x.q_stride = 4 * hd;
        x.q_ss = y_ss;
        x.q_ss_n = emb / 16;
      }
      x.kcache = static_cast<char*>(L.cross_kv) + crow0 * H * T * 64 * kes;
      x.vcache = static_cast<char*>(L.cross_kv) + (static_cast<size_t>(B_total) + crow0) * H * T * 64 * kes;
      x.kv_scale = e->kv_fp8 ? L.cross_scale + crow0 * H * T : nullptr;
      if (retire) {
        x.done = e->done + row0;
        x.cache_row = e->slot_row + row0;
      }
      x.cap = T;
      x.n_keys = T;
      x.out = attn_d;
      x.B = rows;
      x.H = H;
      return mt3k::launch_decode_attention(dt, x, s);
    }
    case 5:
      return mt3k::launch_gemm(dt, resid(attn_d, L.wo_x, hd), false, 0, MT3_EPI_RESID, small, s);
    case 6: {
      mt3k::GemmArgs g = normed(L.wi, h_d, 2 * c.mlp_dim, c.mlp_dim);
      g.concurrent = beside;
      return mt3k::launch_gemm(dt, g, !split, nrm, MT3_EPI_GEGLU, small, s);
    }
    default:
      if (fold) {
        // MLP out-projection + residual, and -- as extra output columns with a two-source K = mlp + emb -- what consumes
        // the updated row next, unnormalised: the NEXT layer's q | k | v | cross-q (4HD columns into qkvf), or after
        // the last layer the logits (vocab columns).  The residual rows move to the other buffer (see `cur`).
        const bool last = (op >> 3) + 1 == nl;
        const int nx = last ? c.vocab_size : 4 * hd;
        mt3k::GemmArgs g = gemm_args(h_d, L.w_fold, y_buf(cur ^ 1), rows, emb + nx, c.mlp_dim + emb, emb);
        g.lda = c.mlp_dim;
        g.resid_src = y;
        g.out_ct = f32 ? nullptr : yct_buf(cur ^ 1);
        g.out_ss = y_ss;
        g.out2 = last ? logits : qkvf;
        g.ld2 = nx;
        g.n_split = emb;
        g.A2 = y_ct;
        g.lda2 = emb;
        g.k_split = c.mlp_dim;
        g.concurrent = beside;
        return mt3k::launch_gemm(dt, g, false, 0, mt3k::kEpiResidS, small, s);
      }
      return mt3k::launch_gemm(dt, resid(h_d, L.wo_mlp, c.mlp_dim), false, 0, MT3_EPI_RESID, small, s);
  }
}

int enqueue_chain_step(mt3_engine* e, int row0, int rows, int chain, int B_total, int skip, hipStream_t s,
                       int done_slot = 0) {
  (void)chain;
  for (int op = 0, n = chain_num_ops(e); op < n; ++op)
    MT3_TRY(enqueue_chain_op(e, row0, rows, B_total, skip, op, s, done_slot));
  return MT3_OK;
}

// rows of chain k when `B` rows are dealt to `n` chains
inline void chain_rows(int B, int n, int k, int* row0, int* rows) {
  const int q = B / n, r = B % n;
  *row0 = k * q + (k < r ? k : r);
  *rows = q + (k < r ? 1 : 0);
}

int chains_for(const mt3_engine* e, int B, int requested) {
  int n = requested > 0 ? requested : (e->cfg.decode_chains > 0 ? e->cfg.decode_chains : 1);
  if (n > kMaxChains) n = kMaxChains;
  while (n > 1 && B / n < 16) --n;          // keep at least one MFMA row-fragment per chain
  return n;
}

// direct (un-captured) launch of one whole step on one stream: chains back to back
int enqueue_decode_step(mt3_engine* e, int B, int skip, int n, hipStream_t s) {
  for (int k = 0; k < n; ++k) {
    int row0, rows;
    chain_rows(B, n, k, &row0, &rows);
    MT3_TRY(enqueue_chain_step(e, row0, rows, k, B, skip, s));
  }
  return MT3_OK;
}

void drop_graph(mt3_engine* e) {
  for (int v = 0; v < kNumVariants; ++v)
    for (int n = 0; n < 9; ++n) {
      if (e->graph_exec[v][n]) (void)hipGraphExecDestroy(e->graph_exec[v][n]);
      if (e->graph[v][n]) (void)hipGraphDestroy(e->graph[v][n]);
      e->graph_exec[v][n] = nullptr;
      e->graph[v][n] = nullptr;
    }
  e->graph_batch = 0;
}

// capture one decode step for batch B (on the engine's private stream: the caller's stream may be
// the legacy default stream, which cannot be captured)
int ensure_graph(mt3_engine* e, int B, int skip, int n) {
  if (e->graph_batch != B) drop_graph(e);
  if (e->graph_exec[skip][n]) return MT3_OK;
  for (int k = 0; k < n; ++k) {
    if (!e->cap_stream[k]) MT3_HIP_CHECK(hipStreamCreateWithFlags(&e->cap_stream[k], hipStreamNonBlocking));
    if (!e->cap_event[k]) MT3_HIP_CHECK(hipEventCreateWithFlags(&e->cap_event[k], hipEventDisableTiming));
  }
  hipStream_t origin = e->cap_stream[0];
  MT3_HIP_CHECK(hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal));
  int rc = MT3_OK;
  hipError_t he = hipSuccess;
  // fork: every other chain's stream joins the capture by waiting on an event of the origin
  if (n > 1) he = hipEventRecord(e->cap_event[0], origin);
  for (int k = 1; k < n && he == hipSuccess; ++k) he = hipStreamWaitEvent(e->cap_stream[k], e->cap_event[0], 0);
  for (int k = 0; k < n && he == hipSuccess && rc == MT3_OK; ++k) {
    int row0, rows;
    chain_rows(B, n, k, &row0, &rows);
    rc = enqueue_chain_step(e, row0, rows, k, B, skip, e->cap_stream[k]);
  }
  // join
  for (int k = 1; k < n && he == hipSuccess && rc == MT3_OK; ++k) {
    he = hipEventRecord(e->cap_event[k], e->cap_stream[k]);
    if (he == hipSuccess) he = hipStreamWaitEvent(origin, e->cap_event[k], 0);
  }
  hipGraph_t g = nullptr;
  const hipError_t end = hipStreamEndCapture(origin, &g);
  if (rc != MT3_OK || he != hipSuccess || end != hipSuccess) {
    if (g) (void)hipGraphDestroy(g);
    if (rc != MT3_OK) return rc;
    return mt3::fail(MT3_ERR_HIP, std::string("decode graph capture: ") +
                                      hipGetErrorString(he != hipSuccess ? he : end));
  }
  e->graph[skip][n] = g;
  MT3_HIP_CHECK(hipGraphInstantiate(&e->graph_exec[skip][n], g, nullptr, nullptr, 0));
  e->graph_batch = B;
  return MT3_OK;
}

}  // namespace

extern "C" {

static void workers_stop(mt3_engine* e);
static void drop_group_graphs(mt3_engine* e);

int mt3_engine_create(const mt3_engine_config* cfg, mt3_engine** out) {
  if (!cfg || !out) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: null argument");
  if (cfg->head_dim != 64) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: head_dim must be 64");
  if (cfg->compute_dtype != MT3_BF16 && cfg->compute_dtype != MT3_F32)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: compute_dtype must be MT3_BF16 or MT3_F32");
  if (cfg->input_length != 256 && cfg->input_length != 512)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: input_length must be 256 (mt3) or 512 (ismir2021)");
  if (cfg->emb_dim % 128 || cfg->mlp_dim % 128 || cfg->vocab_size % 128 || cfg->input_depth % 64 ||
      (cfg->num_heads * 64) % 128)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: emb/mlp/vocab/heads*64 must be multiples of 128");
  if (cfg->decode_chains < 0 || cfg->decode_chains > kMaxChains)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: decode_chains must be in [0, 8]");
  if (cfg->max_batch <= 0 || cfg->max_decode_len <= 0 || cfg->max_decode_len > kMaxPos ||
      cfg->num_encoder_layers <= 0 || cfg->num_decoder_layers <= 0)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: bad sizes");
  if (cfg->kv_cache_dtype != 0 && cfg->kv_cache_dtype != MT3_FP8_E4M3)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: kv_cache_dtype must be 0 (= compute dtype) or MT3_FP8_E4M3");
  if (cfg->kv_cache_dtype == MT3_FP8_E4M3 && cfg->compute_dtype != MT3_BF16)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: the fp8 K/V caches go with compute_dtype MT3_BF16");
  if (cfg->dense_dtype != 0 && cfg->dense_dtype != MT3_FP8_E4M3)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: dense_dtype must be 0 (= compute dtype) or MT3_FP8_E4M3");
  if (cfg->dense_dtype == MT3_FP8_E4M3 && (cfg->compute_dtype != MT3_BF16 || cfg->emb_dim > 1024))
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: the MXFP8 dense path goes with compute_dtype MT3_BF16 and emb_dim <= 1024");
  if (cfg->options & ~(MT3_OPT_SINGLE_RESIDUAL_STREAM | MT3_OPT_SEPARATE_PROJECTIONS | MT3_OPT_ENCODER_SINGLE_RESIDUAL_STREAM |
                       MT3_OPT_SEPARATE_QKV_PROJECTION | MT3_OPT_NO_ROW_GROUPS | MT3_OPT_ENCODER_F32_MFMA | MT3_OPT_SPIN_WAITS))
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_create: unknown bit in options");
  mt3_engine* e = new (std::nothrow) mt3_engine();
  if (!e) return mt3::fail(MT3_ERR_INVALID, "out of host memory");
  e->cfg = *cfg;
  if (hipGetDevice(&e->device) != hipSuccess) {
    (void)hipGetLastError();
    e->device = 0;                 // no GPU: the failure is reported by the first call that needs one (finalize)
  }
  e->dense_fp8 = cfg->dense_dtype == MT3_FP8_E4M3;
  e->x6 = cfg->compute_dtype == MT3_F32 && !(cfg->options & MT3_OPT_ENCODER_F32_MFMA);
  e->spin_waits = (cfg->options & MT3_OPT_SPIN_WAITS) != 0;
  e->esize = cfg->compute_dtype == MT3_BF16 ? 2 : 4;
  e->kv_fp8 = cfg->kv_cache_dtype == MT3_FP8_E4M3;
  e->kv_esize = e->kv_fp8 ? 1 : e->esize;
  *out = e;
  return MT3_OK;
}

void mt3_engine_destroy(mt3_engine* e) {
  if (!e) return;
  workers_stop(e);                 // joins a decode that is still in flight
  drop_graph(e);
  drop_group_graphs(e);
  for (int k = 0; k < kMaxChains; ++k) {
    if (e->cap_stream[k]) (void)hipStreamDestroy(e->cap_stream[k]);
    if (e->cap_event[k]) (void)hipEventDestroy(e->cap_event[k]);
  }
  for (int g = 0; g < kMaxGroups; ++g)
    if (e->part_stream[g]) (void)hipStreamDestroy(e->part_stream[g]);
  if (e->part_begin) (void)hipEventDestroy(e->part_begin);
  for (auto& pair : e->wait_ev)
    for (hipEvent_t ev : pair)
      if (ev) (void)hipEventDestroy(ev);
  if (e->h_pinned) (void)hipHostFree(e->h_pinned);
  for (void* p : e->allocs) (void)hipFree(p);
  delete e;
}

int mt3_engine_load_weight(mt3_engine* e, const char* name, const float* h_data, const int64_t* shape, int32_t ndim) {
  if (!e || !name || !h_data || !shape || ndim < 1 || ndim > 2)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_load_weight: bad arguments");
  if (e->finalized) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_load_weight: engine already finalized");
  HostWeight w;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] <= 0) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_load_weight: bad shape");
    w.shape.push_back(shape[i]);
    n *= static_cast<size_t>(shape[i]);
  }
  w.data.assign(h_data, h_data + n);
  e->raw[name] = std::move(w);
  return MT3_OK;
}

int64_t mt3_engine_device_bytes(const mt3_engine* e) { return e ? e->device_bytes : 0; }

int mt3_engine_finalize(mt3_engine* e) {
  if (!e) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_finalize: null engine");
  if (e->finalized) return MT3_OK;
  const mt3_engine_config& c = e->cfg;
  const int emb = c.emb_dim, hd = e->HD(), Bm = c.max_batch, T = c.input_length, L = c.max_decode_len;
  int rc;

  // ---- encoder
  {
    const HostWeight* w = find(e, "encoder/continuous_inputs_projection/kernel", c.input_depth, emb);
    if (!w) return MT3_ERR_MISSING;
    std::vector<float> t(static_cast<size_t>(emb) * c.input_depth);
    put_transposed(t, c.input_depth, 0, *w, nullptr);
    if ((rc = upload_ct(e, t, &e->enc_in))) return rc;
    if ((rc = upload_planes(e, t, e->enc_in_p))) return rc;
  }
  e->enc.resize(c.num_encoder_layers);
  for (int l = 0; l < c.num_encoder_layers; ++l) {
    const std::string P = "encoder/layers_" + std::to_string(l);
    const float* s1 = scale_of(e, P + "/pre_attention_layer_norm/scale");
    const float* s2 = scale_of(e, P + "/pre_mlp_layer_norm/scale");
    if (!s1 || !s2) return MT3_ERR_MISSING;
    if ((rc = build_attention(e, P + "/attention", s1, false, &e->enc[l], true))) return rc;
    if ((rc = build_mlp(e, P + "/mlp", s2, &e->enc[l], true))) return rc;
  }
  {
    const HostWeight* w = find(e, "encoder/encoder_norm/scale", emb, -1);
    if (!w) return MT3_ERR_MISSING;
    if ((rc = upload_f32(e, w->data, &e->enc_norm))) return rc;
  }
  // ---- decoder
  {
    const HostWeight* w = find(e, "decoder/token_embedder/embedding", c.vocab_size, emb);
    if (!w) return MT3_ERR_MISSING;
    if ((rc = upload_f32(e, w->data, &e->embedding))) return rc;
  }
  e->dec.resize(c.num_decoder_layers);
  const bool single_stream = (c.options & MT3_OPT_SINGLE_RESIDUAL_STREAM) != 0;
  // (emb = 768, the ismir2022/base.gin shape: the split residual form exists for the bf16 tiles only -- their 48 partial
  // sums fit the NPV = 16 registers; an f32 engine of that shape keeps the single f32 stream with in-kernel statistics)
  const bool q_fold = emb % 64 == 0 && (emb <= 512 || (emb == 768 && c.compute_dtype == MT3_BF16)) && !single_stream &&
                      !(c.options & MT3_OPT_SEPARATE_PROJECTIONS);
  e->q_fold = q_fold;
  for (int l = 0; l < c.num_decoder_layers; ++l) {
    const std::string P = "decoder/layers_" + std::to_string(l);
    const float* s1 = scale_of(e, P + "/pre_self_attention_layer_norm/scale");
    const float* s2 = scale_of(e, P + "/pre_cross_attention_layer_norm/scale");
    const float* s3 = scale_of(e, P + "/pre_mlp_layer_norm/scale");
    if (!s1 || !s2 || !s3) return MT3_ERR_MISSING;
    if ((rc = build_attention(e, P + "/self_attention", s1, false, &e->dec[l]))) return rc;
    if ((rc = build_attention(e, P + "/encoder_decoder_attention", s2, true, &e->dec[l]))) return rc;
    if (q_fold && (rc = build_q_fold(e, P, s1, s2, &e->dec[l]))) return rc;
    if ((rc = build_mlp(e, P + "/mlp", s3, &e->dec[l]))) return rc;
    const size_t kvb = static_cast<size_t>(Bm) * c.num_heads * L * 64 * e->kv_esize;
    if ((rc = dmalloc(e, &e->dec[l].self_k, kvb))) return rc;
    if ((rc = dmalloc(e, &e->dec[l].self_v, kvb))) return rc;
    // zero-filled once: the decode-attention kernels request their first key group before they know the row's
    // length and mask what lies past it -- those bytes must be finite (0 x NaN would poison the accumulators)
    MT3_HIP_CHECK(hipMemset(e->dec[l].self_k, 0, kvb));
    MT3_HIP_CHECK(hipMemset(e->dec[l].self_v, 0, kvb));
    if ((rc = dmalloc(e, &e->dec[l].cross_kv, static_cast<size_t>(2) * Bm * c.num_heads * T * 64 * e->kv_esize)))
      return rc;
    if (e->kv_fp8) {
      if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->dec[l].self_scale),
                        static_cast<size_t>(Bm) * c.num_heads * L * sizeof(float2))))
        return rc;
      MT3_HIP_CHECK(hipMemset(e->dec[l].self_scale, 0, static_cast<size_t>(Bm) * c.num_heads * L * sizeof(float2)));
      if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->dec[l].cross_scale),
                        static_cast<size_t>(Bm) * c.num_heads * T * sizeof(float2))))
        return rc;
    }
  }
  if (e->kv_fp8 && (rc = dmalloc(e, &e->cross_stage, static_cast<size_t>(2) * Bm * c.num_heads * T * 64 * 2))) return rc;
  {
    const float* sn = scale_of(e, "decoder/decoder_norm/scale");
    const HostWeight* w = find(e, "decoder/logits_dense/kernel", emb, c.vocab_size);
    if (!sn || !w) return MT3_ERR_MISSING;
    std::vector<float> t(static_cast<size_t>(c.vocab_size) * emb);
    put_transposed(t, emb, 0, *w, sn);
    if ((rc = upload_ct(e, t, &e->logits_w))) return rc;
  }
  // ---- sinusoidal table (layers.py:51-82): [sin | cos] halves, scale = -ln(10000)/(emb/2 - 1)
  {
    std::vector<float> pe(static_cast<size_t>(kMaxPos) * emb);
    const int half = emb / 2;
    const double sf = -std::log(10000.0) / (half - 1);
    for (int p = 0; p < kMaxPos; ++p)
      for (int j = 0; j < half; ++j) {
        const double div = std::exp(j * sf);
        pe[static_cast<size_t>(p) * emb + j] = static_cast<float>(std::sin(p * div));
        pe[static_cast<size_t>(p) * emb + half + j] = static_cast<float>(std::cos(p * div));
      }
    if ((rc = upload_f32(e, pe, &e->pos_table))) return rc;
  }
  // ---- qkv-fold (needs the embedding and the position table on the device, and the raw weights still on the host)
  e->qkv_fold = q_fold && !(c.options & MT3_OPT_SEPARATE_QKV_PROJECTION) && emb % 512 == 0 && c.mlp_dim % 512 == 0;
  if (e->qkv_fold && (rc = build_qkv_fold(e))) return rc;
  // ---- workspaces
  const size_t M = static_cast<size_t>(Bm) * T;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->x), M * emb * 4))) return rc;
  e->x_split = c.compute_dtype == MT3_BF16 && emb % 64 == 0 && emb <= 1024 &&
               !(c.options & MT3_OPT_ENCODER_SINGLE_RESIDUAL_STREAM) && !e->dense_fp8;
  if (e->x_split) {
    if ((rc = dmalloc(e, &e->x_ct, M * emb * 2))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->x_ss), M * (emb / 16) * 4))) return rc;
  }
  if (e->dense_fp8) {
    if (!e->x_ss && (rc = dmalloc(e, reinterpret_cast<void**>(&e->x_ss), M * (emb / 16) * 4))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->x_q), M * emb))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->x_sc), M * (emb / 32)))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->attn_q), M * hd))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->attn_sc), M * (hd / 32)))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->h_q), M * c.mlp_dim))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->h_sc), M * (c.mlp_dim / 32)))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->enc_q), M * emb))) return rc;
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->enc_sc), M * (emb / 32)))) return rc;
  }
  if ((rc = dmalloc(e, &e->qkv, M * 3 * hd * e->esize))) return rc;
  if ((rc = dmalloc(e, &e->attn, M * hd * e->esize))) return rc;
  if ((rc = dmalloc(e, &e->hbuf, M * c.mlp_dim * e->esize))) return rc;
  if ((rc = dmalloc(e, &e->enc_out, M * emb * e->esize))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->y), static_cast<size_t>(Bm) * emb * 4))) return rc;
  // the split residual form in the decode loop.  bf16: f32 rows + bf16 copy + per-16-column sums of squares; f32 (round
  // 3): the compute-type rows ARE the f32 rows, so only the sums travel with them (y_ct aliases y and is never written
  // as a copy) -- the norm-fused GEMMs then need no statistics pass and the folded projections work as in bf16
  e->y_split = emb % 64 == 0 && (emb <= 512 || (emb == 768 && c.compute_dtype == MT3_BF16)) && !single_stream;
  if (e->y_split) {
    if (c.compute_dtype == MT3_BF16) {
      if ((rc = dmalloc(e, &e->y_ct, static_cast<size_t>(Bm) * emb * 2))) return rc;
    } else {
      e->y_ct = e->y;
    }
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->y_ss), static_cast<size_t>(Bm) * (emb / 16) * 4))) return rc;
  }
  if (e->q_fold && !e->y_split) e->q_fold = false;
  if (!e->q_fold) e->qkv_fold = false;
  if (e->q_fold && (rc = dmalloc(e, reinterpret_cast<void**>(&e->qf), static_cast<size_t>(Bm) * hd * 4))) return rc;
  if (e->qkv_fold) {
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->qkvf), static_cast<size_t>(Bm) * 4 * hd * 4))) return rc;
    if (c.compute_dtype == MT3_BF16) {
      if ((rc = dmalloc(e, &e->y_ct_alt, static_cast<size_t>(Bm) * emb * 2))) return rc;
    } else if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->y_alt), static_cast<size_t>(Bm) * emb * 4))) {
      return rc;
    }
  }
  if ((rc = dmalloc(e, &e->qkv_d, static_cast<size_t>(Bm) * 3 * hd * e->esize))) return rc;
  if ((rc = dmalloc(e, &e->attn_d, static_cast<size_t>(Bm) * hd * e->esize))) return rc;
  if ((rc = dmalloc(e, &e->q_d, static_cast<size_t>(Bm) * hd * e->esize))) return rc;
  if ((rc = dmalloc(e, &e->h_d, static_cast<size_t>(Bm) * c.mlp_dim * e->esize))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->logits), static_cast<size_t>(Bm) * c.vocab_size * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->ids), static_cast<size_t>(Bm) * L * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->cur_tok), static_cast<size_t>(Bm) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->done), static_cast<size_t>(Bm) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->step), static_cast<size_t>(Bm) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->n_done), 4 * kMaxChains))) return rc;   // one counter per row group
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->forced), static_cast<size_t>(Bm) * L * 4))) return rc;
  // row retirement: slot map, synthetic EOS schedule, compaction scratch
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->slot_row), static_cast<size_t>(Bm) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->eos_at), static_cast<size_t>(Bm) * 4))) return rc;
  e->eos_cap = Bm;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->slot_seg), static_cast<size_t>(Bm) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->cs_seg), static_cast<size_t>(Bm) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->refill_plan), (static_cast<size_t>(Bm) + kMaxChains) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->cs_y), static_cast<size_t>(Bm) * emb * 4))) return rc;
  if (e->y_split && c.compute_dtype == MT3_BF16 && (rc = dmalloc(e, &e->cs_y_ct, static_cast<size_t>(Bm) * emb * 2))) return rc;
  if (e->y_split && (rc = dmalloc(e, reinterpret_cast<void**>(&e->cs_y_ss), static_cast<size_t>(Bm) * (emb / 16) * 4))) return rc;
  if (e->qkv_fold && (rc = dmalloc(e, reinterpret_cast<void**>(&e->cs_qkvf), static_cast<size_t>(Bm) * 4 * hd * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->cs_int), static_cast<size_t>(Bm) * 4 * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->cs_beam), static_cast<size_t>(Bm) * 2 * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->cs_perm), (static_cast<size_t>(Bm) + kMaxChains) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->beam_f), static_cast<size_t>(2) * Bm * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->beam_len), static_cast<size_t>(Bm) * 4))) return rc;
  if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->beam_len_row), static_cast<size_t>(Bm) * 4))) return rc;
  {
    // beam_cfg[0]: loop bound of the current call; beam_cfg[1 + n] = brevity_penalty(n), n = 0 .. L + 1
    std::vector<float> bp(static_cast<size_t>(L) + 3, 0.f);
    for (int n = 0; n <= L + 1; ++n) bp[1 + n] = brevity_penalty(n);
    if ((rc = dmalloc(e, reinterpret_cast<void**>(&e->beam_cfg), bp.size() * 4))) return rc;
    MT3_HIP_CHECK(hipMemcpy(e->beam_cfg, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  }
  MT3_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->h_pinned), 64, hipHostMallocDefault));
  e->raw.clear();
  e->finalized = true;
  return MT3_OK;
}

// Where an encoder pass leaves the cross-attention K/V of its batch: the engine's caches (rows 0 .. batch - 1: what
// mt3_engine_encode does), or a staging chunk of mt3_engine_transcribe ([2][batch][H][T][64] per layer, scales [batch][H][T])
struct CrossDst {
  void* const* kv = nullptr;         // per decoder layer (nullptr: the caches)
  float2* const* scale = nullptr;
};

static int encode_impl(mt3_engine* e, const float* d_inputs, int32_t batch, float* d_encoded_f32, const CrossDst& dst,
                       hipStream_t s) {
  const mt3_engine_config& c = e->cfg;
  auto ckv = [&](int l) -> void* { return dst.kv ? dst.kv[l] : e->dec[l].cross_kv; };
  auto csc = [&](int l) -> float2* { return dst.scale ? dst.scale[l] : e->dec[l].cross_scale; };
  const int dt = c.compute_dtype, emb = c.emb_dim, hd = e->HD(), T = c.input_length;
  const int M = batch * T;
  const bool small = M < 2048;
  // (gemm_x6_kernel addresses its operands with 32-bit byte offsets: a batch whose widest f32 activation matrix reaches
  // 4 GB -- 4096 segments at the MT3 shape -- takes the f32-instruction path below instead of failing)
  const int widest = std::max(std::max(c.input_depth, emb), std::max(hd, c.mlp_dim));
  const bool x6_fits = static_cast<size_t>(M) * widest * 4 < (1ull << 32);
  if (e->x6 && x6_fits) {
    // f32 engine: every dense layer and the attention on the bf16 pipes with three planes per operand (gemm.hip,
    // gemm_x6_kernel: at least as exact as the f32 matrix instruction, 2.7x its rate).  Since round 5 at EVERY batch size
    // (up to round 4 passes of fewer than 2048 rows took the f32 instruction's decode-sized tiles, so a segment's f32
    // encoder output depended on the batch it sat in by ~1e-7 -- ADVICE r4): a file's last short batch, a refill chunk of
    // mt3_engine_transcribe and a 1250-segment corpus call now give a segment the same bits
    auto x6 = [&](const void* A, void* (&W)[3], void* out, int N, int K, int ldo, bool norm, int epi, int seq) {
      mt3k::GemmArgs g = gemm_args(A, W[0], out, M, N, K, ldo);
      g.aux = e->pos_table;
      g.seq_len = seq;
      return mt3k::launch_gemm_x6(g, W[1], W[2], norm, epi, s);
    };
    MT3_TRY(x6(d_inputs, e->enc_in_p, e->x, emb, c.input_depth, emb, false, MT3_EPI_POS, T));
    for (int l = 0; l < c.num_encoder_layers; ++l) {
      LayerDev& L = e->enc[l];
      MT3_TRY(x6(e->x, L.wqkv_p, e->qkv, 3 * hd, emb, 3 * hd, true, MT3_EPI_STORE, 0));
      MT3_TRY(mt3k::launch_encoder_attention_x6(e->qkv, e->attn, batch, T, c.num_heads, s));
      MT3_TRY(x6(e->attn, L.wo_p, e->x, emb, hd, emb, false, MT3_EPI_RESID, 0));
      MT3_TRY(x6(e->x, L.wi_p, e->hbuf, 2 * c.mlp_dim, emb, c.mlp_dim, true, MT3_EPI_GEGLU, 0));
      MT3_TRY(x6(e->hbuf, L.wo_mlp_p, e->x, emb, c.mlp_dim, emb, false, MT3_EPI_RESID, 0));
    }
    MT3_TRY(mt3k::launch_rmsnorm(dt, e->x, e->enc_norm, e->enc_out, d_encoded_f32, M, emb, s));
    for (int l = 0; l < c.num_decoder_layers; ++l)
      MT3_TRY(x6(e->enc_out, e->dec[l].wkv_x_p, ckv(l), 2 * hd, emb, 2 * hd, false, MT3_EPI_HEADS, T));
    return MT3_OK;
  }
  {
    mt3k::GemmArgs g = gemm_args(d_inputs, e->enc_in, e->x, M, emb, c.input_depth, emb);
    g.aux = e->pos_table;
    g.seq_len = T;
    MT3_TRY(mt3k::launch_gemm(dt, g, true, false, MT3_EPI_POS, small, s));
  }
  if (e->dense_fp8) {
    // MXFP8 encoder: x travels as f32 rows + e4m3/E8M0 copy + per-16-column sums of squares; every GEMM operand is
    // MXFP8, written in that form by its producer (the quantiser below, the RESID / GEGLU epilogues)
    auto mx = [&](const uint8_t* A, const uint8_t* a_sc, const uint8_t* W, const uint8_t* w_sc, int N, int K) {
      mt3k::Mx8Args g{};
      g.A = A;
      g.a_sc = a_sc;
      g.W = W;
      g.w_sc = w_sc;
      g.M = M;
      g.N = N;
      g.K = K;
      g.ldo = N;
      return g;
    };
    auto resid = [&](const uint8_t* A, const uint8_t* a_sc, const uint8_t* W, const uint8_t* w_sc, int K) {
      mt3k::Mx8Args g = mx(A, a_sc, W, w_sc, emb, K);
      g.out = e->x;
      g.out_q = e->x_q;
      g.out_sc = e->x_sc;
      g.out_ss = e->x_ss;
      return mt3k::launch_gemm_mx8(g, MT3_EPI_RESID, s);
    };
    MT3_TRY(mt3k::launch_mx8_quantize(e->x, true, M, emb, e->x_q, e->x_sc, e->x_ss, s));
    for (int l = 0; l < c.num_encoder_layers; ++l) {
      LayerDev& L = e->enc[l];
      {
        mt3k::Mx8Args g = mx(e->x_q, e->x_sc, L.wqkv_q, L.wqkv_sc, 3 * hd, emb);
        g.out = e->qkv;
        g.a_ss = e->x_ss;
        MT3_TRY(mt3k::launch_gemm_mx8(g, MT3_EPI_STORE, s));
      }
      MT3_TRY(mt3k::launch_encoder_attention(dt, e->qkv, e->attn, batch, T, c.num_heads, s));
      MT3_TRY(mt3k::launch_mx8_quantize(e->attn, false, M, hd, e->attn_q, e->attn_sc, nullptr, s));
      MT3_TRY(resid(e->attn_q, e->attn_sc, L.wo_q, L.wo_sc, hd));
      {
        mt3k::Mx8Args g = mx(e->x_q, e->x_sc, L.wi_q, L.wi_sc, 2 * c.mlp_dim, emb);
        g.ldo = c.mlp_dim;
        g.a_ss = e->x_ss;
        g.out_q = e->h_q;
        g.out_sc = e->h_sc;
        MT3_TRY(mt3k::launch_gemm_mx8(g, MT3_EPI_GEGLU, s));
      }
      MT3_TRY(resid(e->h_q, e->h_sc, L.wo_mlp_q, L.wo_mlp_sc, c.mlp_dim));
    }
    MT3_TRY(mt3k::launch_rmsnorm(dt, e->x, e->enc_norm, e->enc_out, d_encoded_f32, M, emb, s));
    MT3_TRY(mt3k::launch_mx8_quantize(e->enc_out, false, M, emb, e->enc_q, e->enc_sc, nullptr, s));
    for (int l = 0; l < c.num_decoder_layers; ++l) {
      mt3k::Mx8Args g = mx(e->enc_q, e->enc_sc, e->dec[l].wkv_x_q, e->dec[l].wkv_x_sc, 2 * hd, emb);
      g.out = e->kv_fp8 ? e->cross_stage : ckv(l);
      g.seq_len = T;
      MT3_TRY(mt3k::launch_gemm_mx8(g, MT3_EPI_HEADS, s));
      if (e->kv_fp8)
        MT3_TRY(mt3k::launch_kv_quantize_fp8(e->cross_stage, ckv(l), csc(l),
                                             batch * c.num_heads * T, s));
    }
    return MT3_OK;
  }
  // the split residual form feeds the LDS-DMA tile (K <= 1024); the decode-sized tile a small batch selects holds the
  // partial sums of K <= 512 or K = 768 only, so a small batch of a wider model stays on the single f32 stream
  const bool xs = e->x_split && !(small && !(emb <= 512 || emb == 768));
  if (xs) MT3_TRY(mt3k::launch_residual_split(e->x, e->x_ct, e->x_ss, M, emb, s));
  auto normed = [&](const void* Wt, void* out, int N, int ldo) {
    mt3k::GemmArgs g = gemm_args(xs ? static_cast<const void*>(e->x_ct) : static_cast<const void*>(e->x), Wt, out, M,
                                 N, emb, ldo);
    g.a_ss = xs ? e->x_ss : nullptr;
    return g;
  };
  auto resid = [&](const void* A, const void* Wt, int K) {
    mt3k::GemmArgs g = gemm_args(A, Wt, e->x, M, emb, K, emb);
    g.out_ct = xs ? e->x_ct : nullptr;
    g.out_ss = xs ? e->x_ss : nullptr;
    return g;
  };
  const int nrm = xs ? 2 : 1;
  for (int l = 0; l < c.num_encoder_layers; ++l) {
    LayerDev& L = e->enc[l];
    MT3_TRY(mt3k::launch_gemm(dt, normed(L.wqkv, e->qkv, 3 * hd, 3 * hd), !xs, nrm, MT3_EPI_STORE, small, s));
    MT3_TRY(mt3k::launch_encoder_attention(dt, e->qkv, e->attn, batch, T, c.num_heads, s));
    MT3_TRY(mt3k::launch_gemm(dt, resid(e->attn, L.wo, hd), false, 0, MT3_EPI_RESID, small, s));
    MT3_TRY(mt3k::launch_gemm(dt, normed(L.wi, e->hbuf, 2 * c.mlp_dim, c.mlp_dim), !xs, nrm, MT3_EPI_GEGLU, small, s));
    MT3_TRY(mt3k::launch_gemm(dt, resid(e->hbuf, L.wo_mlp, c.mlp_dim), false, 0, MT3_EPI_RESID, small, s));
  }
  MT3_TRY(mt3k::launch_rmsnorm(dt, e->x, e->enc_norm, e->enc_out, d_encoded_f32, M, emb, s));
  for (int l = 0; l < c.num_decoder_layers; ++l) {
    mt3k::GemmArgs g = gemm_args(e->enc_out, e->dec[l].wkv_x, e->kv_fp8 ? e->cross_stage : ckv(l), M,
                                 2 * hd, emb, 2 * hd);
    g.seq_len = T;
    MT3_TRY(mt3k::launch_gemm(dt, g, false, false, MT3_EPI_HEADS, small, s));
    if (e->kv_fp8)     // bf16 [2][B][H][T][64] -> e4m3 rows + one power-of-two scale per (row, head, position)
      MT3_TRY(mt3k::launch_kv_quantize_fp8(e->cross_stage, ckv(l), csc(l),
                                           batch * c.num_heads * T, s));
  }
  return MT3_OK;
}

int mt3_engine_encode(mt3_engine* e, const float* d_inputs, int32_t batch, float* d_encoded_f32, void* stream) {
  if (!e || !e->finalized) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_encode: engine not finalized");
  if (!d_inputs || batch <= 0 || batch > e->cfg.max_batch)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_encode: batch out of range");
  if (e->pending.active)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_encode: a decode is in flight (MT3_DECODE_ASYNC): call mt3_engine_decode_wait first");
  MT3_TRY(encode_impl(e, d_inputs, batch, d_encoded_f32, CrossDst{}, static_cast<hipStream_t>(stream)));
  e->cur_batch = batch;
  return MT3_OK;
}

// ---------------------------------------------------------------------------------------------- the decode loop
// Row groups of the product decode schedule (mt3_engine::part_stream).  Measured on MI355X, ms per 1024-step decode,
// 1 / 2 / 4 groups (profiles/r3_ab_row_groups_*.txt): bf16 B = 256: 626 / 588 / 608, B = 512: 1113 / 1048 / 1013;
// f32 B = 128: 747 / 684 / 737, B = 256: 1178 / 1132 / 1098 -- groups of ~128 rows for bf16 operands, ~64 for f32.
// (Three groups are never better than two or four.  EIGHT groups of 32 rows -- round 4, f32, B = 256: 2307 ms against
// 1098 with four, the EOS-schedule decode 1052 against 380 ms, profiles/r4_ab_eight_row_groups.txt.  The cliff is at
// exactly FIVE: 5 / 6 groups 2421 / 2306 ms against 1106 with four (3: 1144), the same with GPU_MAX_HW_QUEUES=8 in the
// environment, profiles/r4_ab_five_six_row_groups.txt -- as if a fifth busy queue shared one of four compute pipes with
// another one and the two took turns (the decode waits for a group running at half speed).  Four it is.)
// Under MT3_DECODE_EARLY_EXIT (the ragged regime: rows retire, the step is launch latency, not bandwidth) f32 follows
// the bf16 rule: at B = 256 two groups 370.5 ms, four 380.7, one stream 377.5 (profiles/r4_ab_tall_tiles_groups_wait.txt,
// profiles/r4_bench_driver_like.json eos_schedule); bf16 two groups 212.3 against 220.0 on one stream.
static int row_groups_for(const mt3_engine_config& c, int batch, bool early_exit = false) {
  const bool f32 = c.compute_dtype != MT3_BF16;
  // (round 6, with the dense tiles' priority in place: four groups under early exit from 256 f32 rows on -- 368-372 ms per
  // EOS-schedule decode against 362-367 with two: profiles/r6_ab_priorities_and_kernarg_pin.txt, block 4; the rule stays)
  if (batch >= (f32 && !early_exit ? 256 : 512)) return 4;
  return batch >= 128 ? 2 : 1;
}

// Row groups of mt3_engine_transcribe (in-flight batching keeps every group's rows full, so the rule of the canonical
// full-length schedule applies, not the ragged one's)
static int stream_row_groups_for(const mt3_engine_config& c, int slots) { return row_groups_for(c, slots); }

// ---- persistent group workers
static void worker_main(Worker* w, int dev) {
  (void)hipSetDevice(dev);
  std::unique_lock<std::mutex> lk(w->mu);
  for (;;) {
    w->cv.wait(lk, [&] { return w->has_job || w->quit; });
    if (!w->has_job) return;                       // quit (a posted job is always run first)
    std::function<void()> job = std::move(w->job);
    lk.unlock();
    try {
      job();
    } catch (...) {                                // nothing may leave a thread of a C library
    }
    lk.lock();
    w->has_job = false;
    w->cv.notify_all();
  }
}

// false: the worker thread could not be created (the caller falls back to the calling thread)
static bool worker_post(mt3_engine* e, int g, std::function<void()> fn) {
  if (!e->workers[g]) {
    Worker* w = new (std::nothrow) Worker();
    if (!w) return false;
    try {
      w->th = std::thread(worker_main, w, e->device);
    } catch (...) {
      delete w;
      return false;
    }
    e->workers[g] = w;
  }
  Worker* w = e->workers[g];
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->job = std::move(fn);
    w->has_job = true;
  }
  w->cv.notify_all();
  return true;
}

static void worker_wait(mt3_engine* e, int g) {
  Worker* w = e->workers[g];
  if (!w) return;
  std::unique_lock<std::mutex> lk(w->mu);
  w->cv.wait(lk, [&] { return !w->has_job; });
}

static void workers_stop(mt3_engine* e) {
  for (int g = 0; g < kMaxGroups; ++g) {
    Worker* w = e->workers[g];
    if (!w) continue;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return !w->has_job; });
      w->quit = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
    e->workers[g] = nullptr;
  }
}

static void drop_group_graphs(mt3_engine* e) {
  for (GroupGraph& g : e->group_graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  e->group_graphs.clear();
}

// The captured step of slots [row0, row0 + rows) (nullptr: capture failed, the caller launches directly).  Captured on
// an engine-owned stream in thread-local mode -- several group threads may be here at once, the cache is guarded.
static hipGraphExec_t group_graph(mt3_engine* e, int variant, int batch, int row0, int rows, int slot) {
  std::lock_guard<std::mutex> lk(e->graph_mu);
  const int max_len = (variant & kVarStream) ? e->stream_max_len : 0;      // baked into the step's token kernel
  for (const GroupGraph& g : e->group_graphs)
    if (g.variant == variant && g.batch == batch && g.row0 == row0 && g.rows == rows && g.slot == slot &&
        g.max_len == max_len)
      return g.exec;
  if (!e->cap_stream[slot] && hipStreamCreateWithFlags(&e->cap_stream[slot], hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  hipStream_t cs = e->cap_stream[slot];
  if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  const int rc = enqueue_chain_step(e, row0, rows, slot, batch, variant, cs, slot);
  hipGraph_t g = nullptr;
  hipGraphExec_t x = nullptr;
  const hipError_t end = hipStreamEndCapture(cs, &g);
  if (rc != MT3_OK || end != hipSuccess || hipGraphInstantiate(&x, g, nullptr, nullptr, 0) != hipSuccess) {
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    return nullptr;
  }
  e->group_graphs.push_back(GroupGraph{variant, batch, row0, rows, slot, max_len, g, x});
  return x;
}

// One row group's decode loop: slots [row0, row0 + rows) of a `batch`-row decode on stream `s` (the single-stream
// schedule is the group [0, batch) on the caller's stream).
struct GroupRun {
  int row0, rows, batch, variant, num_steps, slot;
  bool early, use_graph;
  hipStream_t s;
  float* d_first_logits;      // single-stream schedule only
  float* d_step_logits;
  int ran;
  bool used_graph;
};

// ---- sleeping waits (mt3_engine::wait_ev)
static hipEvent_t wait_event(mt3_engine* e, int slot, int which) {
  hipEvent_t& ev = e->wait_ev[slot][which];
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    ev = nullptr;
  }
  return ev;
}

// The calling thread SLEEPS until `ev` has happened: hipEventQuery between naps of 20 .. 200 us (measured in round 5:
// hipEventSynchronize spins a core for the length of the wait on this runtime even on an event created with
// hipEventBlockingSync -- the CPU seconds of a decode did not move -- so the nap is explicit).  The naps start short so
// that a poll's answer is picked up within tens of microseconds and grow while nothing happens.
static hipError_t sleep_until(hipEvent_t ev) {
  long nap_ns = 20000;
  for (;;) {
    const hipError_t he = hipEventQuery(ev);
    if (he != hipErrorNotReady) return he;
    (void)hipGetLastError();                               // NotReady is sticky in the thread's last-error slot
    std::this_thread::sleep_for(std::chrono::nanoseconds(nap_ns));
    if (nap_ns < 200000) nap_ns += nap_ns / 2;
  }
}

// everything enqueued on `s` so far has run; the calling thread sleeps meanwhile (spins with MT3_OPT_SPIN_WAITS or when
// no event could be created)
static hipError_t wait_stream(mt3_engine* e, int slot, hipStream_t s) {
  hipEvent_t ev = e->spin_waits ? nullptr : wait_event(e, slot, 0);
  if (!ev) return hipStreamSynchronize(s);
  hipError_t he = hipEventRecord(ev, s);
  return he == hipSuccess ? sleep_until(ev) : he;
}

// after step t has been enqueued: at the end of every window, sleep until the window before it is done (at most two
// windows of kThrottleWindow steps are ever enqueued ahead of the device: the runtime's own back-pressure, which spins,
// is never reached)
struct Throttle {
  mt3_engine* e;
  int slot;
  hipStream_t s;
  int recorded = 0;
  hipError_t tick(long t) {
    if (e->spin_waits || t % kThrottleWindow != kThrottleWindow - 1) return hipSuccess;
    hipEvent_t ev = wait_event(e, slot, 1);
    if (!ev) return hipSuccess;
    hipError_t he = hipSuccess;
    if (recorded) he = sleep_until(ev);                  // the window recorded one window ago
    if (he == hipSuccess) he = hipEventRecord(ev, s);
    recorded = 1;
    return he;
  }
};

static int compact_group(mt3_engine* e, const GroupRun& r, int cur) {
  const mt3_engine_config& c = e->cfg;
  const size_t r0 = static_cast<size_t>(r.row0);
  const int emb = c.emb_dim, n4 = 4 * e->HD();
  mt3k::CompactArgs a{};
  a.done = e->done + r0;
  a.slot_row = e->slot_row + r0;
  a.step = e->step + r0;
  a.cur_tok = e->cur_tok + r0;
  if (r.variant & kVarBeam) {
    a.beam_f = e->beam_f + r0;
    a.beam_len = e->beam_len + r0;
    a.beam_rows = c.max_batch;
    a.s_beam = e->cs_beam + 2 * r0;
  }
  a.y = e->y + r0 * emb;                                  // the arg-max kernel leaves the next input row in buffer 0
  a.s_y = e->cs_y + r0 * emb;
  if (e->y_split && c.compute_dtype == MT3_BF16) {
    a.y_ct = static_cast<char*>(e->y_ct) + r0 * emb * 2;
    a.s_y_ct = static_cast<char*>(e->cs_y_ct) + r0 * emb * 2;
  }
  if (e->y_split) {
    a.y_ss = e->y_ss + r0 * (emb / 16);
    a.s_y_ss = e->cs_y_ss + r0 * (emb / 16);
  }
  if (e->qkv_fold) {
    a.qkvf = e->qkvf + r0 * n4;
    a.s_qkvf = e->cs_qkvf + r0 * n4;
  }
  a.emb = emb;
  a.q_n = n4;
  a.s_int = e->cs_int + 4 * r0;
  a.perm = e->cs_perm + r0 + r.slot;
  a.rows = cur;
  if (r.variant & kVarStream) {
    a.slot_seg = e->slot_seg + r0;
    a.s_seg = e->cs_seg + r0;
  }
  return mt3k::launch_compact(a, r.s);
}

static int run_group(mt3_engine* e, GroupRun& r) {
  const mt3_engine_config& c = e->cfg;
  const bool retire = (r.variant & kVarRetire) != 0;
  const bool whole = r.row0 == 0 && r.rows == r.batch;
  int cur = r.rows;                              // slots in use: shrinks with the live rows in 32-row (GEMM tile) steps
  hipGraphExec_t exec = nullptr;
  int exec_rows = -1;
  r.ran = 0;
  r.used_graph = r.use_graph;
  const int wslot = r.s == e->part_stream[r.slot] ? r.slot : kMaxGroups;
  Throttle throttle{e, wslot, r.s};
  for (int t = 0; t < r.num_steps; ++t) {
    if (r.use_graph && exec_rows != cur) {
      exec = group_graph(e, r.variant, r.batch, r.row0, cur, r.slot);
      exec_rows = cur;
      if (!exec) {                               // direct launches give the same ids; the fallback is RECORDED
        r.use_graph = r.used_graph = false;
        ++e->graph_fallbacks;
      }
    }
    if (r.use_graph) MT3_HIP_CHECK(hipGraphLaunch(exec, r.s));
    else MT3_TRY(enqueue_chain_step(e, r.row0, cur, r.slot, r.batch, r.variant, r.s, r.slot));
    ++r.ran;
    if (whole && t == 0 && r.d_first_logits)
      MT3_HIP_CHECK(hipMemcpyAsync(r.d_first_logits, e->logits, static_cast<size_t>(r.batch) * c.vocab_size * 4,
                                   hipMemcpyDeviceToDevice, r.s));
    if (whole && r.d_step_logits)
      MT3_HIP_CHECK(hipMemcpyAsync(r.d_step_logits + static_cast<size_t>(t) * r.batch * c.vocab_size, e->logits,
                                   static_cast<size_t>(r.batch) * c.vocab_size * 4, hipMemcpyDeviceToDevice, r.s));
    if (!r.early) MT3_HIP_CHECK(throttle.tick(t));
    if (r.early && t % 32 == 31) {
      // the poll: how many rows of THIS group are finished (every group stops as soon as its own rows are)
      MT3_HIP_CHECK(hipMemcpyAsync(e->h_pinned + r.slot, e->n_done + r.slot, 4, hipMemcpyDeviceToHost, r.s));
      MT3_HIP_CHECK(wait_stream(e, wslot, r.s));
      const int live = r.rows - e->h_pinned[r.slot];
      if (live <= 0) break;
      if (retire) {
        int want = (live + 31) & ~31;
        if (want > r.rows) want = r.rows;
        if (want < cur) {                        // the live rows fit fewer GEMM row tiles: move them to the front
          MT3_TRY(compact_group(e, r, cur));
          cur = want;
          ++e->compactions_now;
        }
      }
    }
  }
  return MT3_OK;
}

// hardware-queue streams of the row groups (see mt3_engine::part_stream); MT3_ERR_CAPACITY = could not be set up
static int ensure_group_streams(mt3_engine* e, int groups) {
  int n_cu = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || n_cu < 2 * groups)
    return MT3_ERR_CAPACITY;
  for (int g = 0; g < groups; ++g) {
    if (e->part_stream[g]) continue;
    // a mask of ALL compute units: the stream is created through the CU-mask entry point for the hardware queue of
    // its own that comes with it, not to restrict it (see mt3_engine::part_stream)
    std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
    for (int i = 0; i < n_cu; ++i) mask[i >> 5] |= 1u << (i & 31);
    if (hipExtStreamCreateWithCUMask(&e->part_stream[g], static_cast<uint32_t>(mask.size()), mask.data()) != hipSuccess) {
      e->part_stream[g] = nullptr;
      (void)hipGetLastError();
      return MT3_ERR_CAPACITY;
    }
  }
  if (!e->part_begin && hipEventCreateWithFlags(&e->part_begin, hipEventDisableTiming) != hipSuccess)
    return MT3_ERR_CAPACITY;
  return MT3_OK;
}

// Joins the decode that is in flight: waits for its workers, then (on the caller's stream) the beam-1 finalisation
// and the copy of the ids.  Every group thread has waited for its stream, so the caller's stream needs no event.
static int decode_finish(mt3_engine* e, int32_t* h_steps_run) {
  PendingDecode& p = e->pending;
  if (!p.active) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode_wait: no decode in flight");
  for (int g = 0; g < p.posted; ++g) worker_wait(e, g);
  p.active = false;
  e->compactions = e->compactions_now.exchange(0);
  int most = 0;
  for (int g = 0; g < p.groups; ++g) {
    if (p.rcs[g] != MT3_OK)
      return mt3::fail(p.rcs[g], "mt3_engine_decode (row group " + std::to_string(g) + "): " + p.errs[g]);
    most = p.ran[g] > most ? p.ran[g] : most;
  }
  const int L = e->cfg.max_decode_len;
  if (p.beam1) MT3_TRY(mt3k::launch_beam1_finalize(e->ids, L, e->beam_len_row, p.batch, p.s));
  MT3_HIP_CHECK(hipMemcpyAsync(p.d_ids, e->ids, static_cast<size_t>(p.batch) * L * 4, hipMemcpyDeviceToDevice, p.s));
  e->last_groups = p.groups;
  e->last_used_graph = 1;
  for (int g = 0; g < p.groups; ++g)
    if (!p.used_graph[g]) e->last_used_graph = 0;
  if (h_steps_run) *h_steps_run = most;
  return MT3_OK;
}

// shared body of mt3_engine_decode / mt3_engine_decode_forced
static int decode_impl(mt3_engine* e, int32_t batch, int32_t num_steps, int32_t flags, int32_t debug_skip,
                       const int32_t* d_forced, float* d_step_logits, int32_t* d_ids, float* d_first_logits,
                       int32_t* h_steps_run, void* stream) {
  if (!e || !e->finalized) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode: engine not finalized");
  if (e->pending.active)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode: a decode is in flight (MT3_DECODE_ASYNC): call mt3_engine_decode_wait first");
  if (batch <= 0 || batch != e->cur_batch)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode: batch must equal the batch of the preceding encode");
  const mt3_engine_config& c = e->cfg;
  if (num_steps <= 0 || num_steps > c.max_decode_len || !d_ids)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode: num_steps out of range or null ids");
  if (flags & ~(MT3_DECODE_NO_GRAPH | MT3_DECODE_EARLY_EXIT | MT3_DECODE_BEAM1 | MT3_DECODE_SINGLE_STREAM |
                MT3_DECODE_ASYNC | MT3_DECODE_CHAINS(0xF)))
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode: unknown flag bit");
  const bool beam1 = (flags & MT3_DECODE_BEAM1) != 0, early = (flags & MT3_DECODE_EARLY_EXIT) != 0;
  const bool async = (flags & MT3_DECODE_ASYNC) != 0;
  if (d_forced && (beam1 || early || async))
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode_forced: not combinable with BEAM1 / EARLY_EXIT / ASYNC");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = c.max_decode_len;
  MT3_HIP_CHECK(hipMemsetAsync(e->step, 0, static_cast<size_t>(batch) * 4, s));
  MT3_HIP_CHECK(hipMemsetAsync(e->n_done, 0, 4 * kMaxChains, s));
  MT3_HIP_CHECK(hipMemsetAsync(e->done, 0, static_cast<size_t>(batch) * 4, s));
  MT3_HIP_CHECK(hipMemsetAsync(e->cur_tok, 0, static_cast<size_t>(batch) * 4, s));     // BOS = 0
  MT3_HIP_CHECK(hipMemsetAsync(e->ids, 0, static_cast<size_t>(batch) * L * 4, s));
  if (d_forced)    // engine-owned copy: the step graph holds ITS address, whatever buffer the caller passes
    MT3_HIP_CHECK(hipMemcpyAsync(e->forced, d_forced, static_cast<size_t>(batch) * L * 4, hipMemcpyDeviceToDevice, s));
  // decoder input of step 0: Embed(BOS) + FixedEmbed[0]; later steps get theirs from the argmax kernel
  {
    const mt3k::RowProj rp{e->ew0, e->pw0, e->qkv_fold ? e->qkvf : nullptr, 4 * e->HD()};
    MT3_TRY(mt3k::launch_embed(e->embedding, e->pos_table, e->cur_tok, e->step, e->y,
                               c.compute_dtype == MT3_BF16 ? e->y_ct : nullptr, e->y_ss, batch, c.emb_dim, rp, s));
  }
  // Row retirement comes with the early exit: rows that have finished cost nothing from then on (mt3_hip.h).  Without
  // EARLY_EXIT every row runs every step -- the canonical full-length schedule the headline is quoted on.
  const bool retire = early && debug_skip == 0;
  if (retire) MT3_TRY(mt3k::launch_iota(e->slot_row, batch, s));

  // step variant: bits 1 / 2 = mt3_debug_engine_decode's skipped kernels (mt3_hip_debug.h; never set by the product
  // entry points), then kVar*
  const int variant = (debug_skip & 3) | (beam1 ? kVarBeam : 0) | (d_forced ? kVarForced : 0) | (retire ? kVarRetire : 0) |
                      (e->eos_on && !d_forced ? kVarEos : 0);
  if (beam1) {
    // t5x beam_search(alpha = 0.6): live log-prob 0, nothing finished; the loop bound uses the brevity
    // penalty of max_decode_len + 1 (the dummy start token extends the length by one).  The value travels as a
    // kernel argument: no host buffer has to stay alive behind an asynchronous copy.
    MT3_TRY(mt3k::launch_set_float(e->beam_cfg, brevity_penalty(num_steps + 1), s));
    MT3_HIP_CHECK(hipMemsetAsync(e->beam_f, 0, static_cast<size_t>(2) * c.max_batch * 4, s));
    MT3_HIP_CHECK(hipMemsetAsync(e->beam_len, 0xFF, static_cast<size_t>(c.max_batch) * 4, s));   // -1
    MT3_HIP_CHECK(hipMemsetAsync(e->beam_len_row, 0xFF, static_cast<size_t>(c.max_batch) * 4, s));
  }
  if (e->group_graphs.size() > 96) drop_group_graphs(e);      // (no worker is running here)
  PendingDecode& p = e->pending;
  p = PendingDecode();
  p.beam1 = beam1;
  p.batch = batch;
  p.d_ids = d_ids;
  p.s = s;
  const bool use_graph = !(flags & MT3_DECODE_NO_GRAPH);
  const int req_chains = (flags >> 8) & 0xF;

  // ---- the row-group schedule (see mt3_engine::part_stream): batches of >= 128 rows, unless the caller asked for
  // one stream / graph chains, or wants per-step logits (those live on the caller's stream)
  int groups = row_groups_for(c, batch, early);
  if (groups > 1 && !(flags & MT3_DECODE_SINGLE_STREAM) && req_chains == 0 && e->cfg.decode_chains <= 1 &&
      !(c.options & MT3_OPT_NO_ROW_GROUPS) && debug_skip == 0 && !d_forced && !d_step_logits && !d_first_logits) {
    if (ensure_group_streams(e, groups) == MT3_OK) {
      MT3_HIP_CHECK(hipEventRecord(e->part_begin, s));
      p.groups = groups;
      p.active = true;
      for (int g = 0; g < groups; ++g) {
        auto body = [e, g, groups, batch, variant, num_steps, early, use_graph]() {
          PendingDecode& q = e->pending;
          GroupRun r{};
          chain_rows(batch, groups, g, &r.row0, &r.rows);
          r.batch = batch;
          r.variant = variant | kVarBeside;            // groups > 1 here
          r.num_steps = num_steps;
          r.slot = g;
          r.early = early;
          r.use_graph = use_graph;
          r.s = e->part_stream[g];
          hipError_t he = hipStreamWaitEvent(r.s, e->part_begin, 0);
          if (he == hipSuccess) {
            q.rcs[g] = run_group(e, r);
            if (q.rcs[g] != MT3_OK) q.errs[g] = mt3_last_error();
          }
          q.ran[g] = r.ran;
          q.used_graph[g] = r.used_graph;
          // The group's host thread WAITS for its stream.  Measured (r3, f32, B = 256, 4 groups): 1095 ms per 1024-step
          // decode when every group stream has a host thread in hipStreamSynchronize, 1166-1170 ms when only the
          // caller's stream (waiting for events of the groups) is synchronised -- a stream nobody waits on retires its
          // commands through the runtime's interrupt path.
          if (he == hipSuccess) he = wait_stream(e, g, r.s);
          if (q.rcs[g] == MT3_OK && he != hipSuccess) {
            q.rcs[g] = MT3_ERR_HIP;
            q.errs[g] = hipGetErrorString(he);
          }
        };
        if (worker_post(e, g, body)) {
          p.posted = g + 1;
        } else {            // no thread to be had: this group runs on the calling thread (after the posted ones started)
          body();
        }
      }
      if (async) return MT3_OK;
      return decode_finish(e, h_steps_run);
    }
    ++e->part_failed;                             // the masked streams could not be created: single stream, recorded
  }
  p.groups = 1;
  const int chains = chains_for(e, batch, req_chains);
  if (chains > 1) {
    // the step graph with `chains` parallel branches (r1's schedule, kept for comparison): inline on the caller's stream
    bool graph = use_graph;
    if (graph && ensure_graph(e, batch, variant, chains) != MT3_OK) {
      graph = false;
      ++e->graph_fallbacks;
    }
    p.used_graph[0] = graph;
    int ran = 0;
    for (int t = 0; t < num_steps; ++t) {
      if (graph) MT3_HIP_CHECK(hipGraphLaunch(e->graph_exec[variant][chains], s));
      else MT3_TRY(enqueue_decode_step(e, batch, variant, chains, s));
      ++ran;
      if (t == 0 && d_first_logits)
        MT3_HIP_CHECK(hipMemcpyAsync(d_first_logits, e->logits, static_cast<size_t>(batch) * c.vocab_size * 4,
                                     hipMemcpyDeviceToDevice, s));
      if (d_step_logits)
        MT3_HIP_CHECK(hipMemcpyAsync(d_step_logits + static_cast<size_t>(t) * batch * c.vocab_size, e->logits,
                                     static_cast<size_t>(batch) * c.vocab_size * 4, hipMemcpyDeviceToDevice, s));
      if (early && (t % 32 == 31)) {
        MT3_HIP_CHECK(hipMemcpyAsync(e->h_pinned, e->n_done, 4, hipMemcpyDeviceToHost, s));
        MT3_HIP_CHECK(wait_stream(e, kMaxGroups, s));
        if (e->h_pinned[0] >= batch) break;
      }
    }
    p.ran[0] = ran;
    p.active = true;
    if (async) return MT3_OK;
    return decode_finish(e, h_steps_run);
  }
  // one group = the whole batch, on the caller's stream
  auto body = [e, batch, variant, num_steps, early, use_graph, s, d_first_logits, d_step_logits]() {
    PendingDecode& q = e->pending;
    GroupRun r{};
    r.row0 = 0;
    r.rows = r.batch = batch;
    r.variant = variant;
    r.num_steps = num_steps;
    r.slot = 0;
    r.early = early;
    r.use_graph = use_graph;
    r.s = s;
    r.d_first_logits = d_first_logits;
    r.d_step_logits = d_step_logits;
    q.rcs[0] = run_group(e, r);
    if (q.rcs[0] != MT3_OK) q.errs[0] = mt3_last_error();
    q.ran[0] = r.ran;
    q.used_graph[0] = r.used_graph;
  };
  p.active = true;
  if (async && worker_post(e, 0, body)) {
    p.posted = 1;
    return MT3_OK;
  }
  body();
  if (async) return MT3_OK;
  return decode_finish(e, h_steps_run);
}

// ------------------------------------------------------------------------------------------- in-flight batching
// mt3_engine_transcribe (mt3_hip.h): the engine's max_batch decode slots stay full while there are segments left.  The
// queue between the encoder passes (producer: the calling thread) and the row groups (consumers) is csrc/feed.h.
using mt3feed::Feed;
using mt3feed::FeedRange;
using mt3feed::feed_fail;
using mt3feed::feed_pop;
using mt3feed::feed_release;
using mt3feed::feed_wait;

static int ensure_stage(mt3_engine* e) {
  if (e->stage_cap) return MT3_OK;
  const mt3_engine_config& c = e->cfg;
  const int cap = c.max_batch < kStageChunkCap ? c.max_batch : kStageChunkCap;
  const size_t row = static_cast<size_t>(c.num_heads) * c.input_length * 64;
  e->stage_kv.assign(c.num_decoder_layers, nullptr);
  e->stage_scale.assign(c.num_decoder_layers, nullptr);
  for (int l = 0; l < c.num_decoder_layers; ++l) {
    MT3_TRY(dmalloc(e, &e->stage_kv[l], static_cast<size_t>(kStageChunks) * 2 * cap * row * e->kv_esize));
    if (e->kv_fp8)
      MT3_TRY(dmalloc(e, reinterpret_cast<void**>(&e->stage_scale[l]),
                      static_cast<size_t>(kStageChunks) * cap * c.num_heads * c.input_length * sizeof(float2)));
  }
  e->stage_cap = cap;
  return MT3_OK;
}

// the refill launches of one run of staged segments for row group `r` (rg == nullptr: finished slots only hand their
// ids over -- the queue is empty)
static int refill_group(mt3_engine* e, const GroupRun& r, int cur, const FeedRange* rg, int32_t* d_out) {
  const mt3_engine_config& c = e->cfg;
  const size_t r0 = static_cast<size_t>(r.row0);
  const int emb = c.emb_dim, n4 = 4 * e->HD();
  mt3k::RefillArgs a{};
  a.done = e->done + r0;
  a.slot_row = e->slot_row + r0;
  a.slot_seg = e->slot_seg + r0;
  a.step = e->step + r0;
  a.cur_tok = e->cur_tok + r0;
  a.n_done = e->n_done + r.slot;
  if (r.variant & kVarBeam) {
    a.beam_f = e->beam_f + r0;
    a.beam_len = e->beam_len + r0;
    a.beam_len_row = e->beam_len_row;
    a.beam_rows = c.max_batch;
  }
  a.y = e->y + r0 * emb;                                   // the arg-max kernel leaves the next input row in buffer 0
  if (e->y_split && c.compute_dtype == MT3_BF16) a.y_ct = static_cast<char*>(e->y_ct) + r0 * emb * 2;
  if (e->y_split) a.y_ss = e->y_ss + r0 * (emb / 16);
  a.emb = emb;
  a.table = e->embedding;
  a.pos = e->pos_table;
  a.rp = mt3k::RowProj{e->ew0, e->pw0, e->qkv_fold ? e->qkvf + r0 * n4 : nullptr, n4};
  a.ids = e->ids;
  a.ids_stride = c.max_decode_len;
  a.out_ids = d_out;
  a.plan = e->refill_plan + r0 + r.slot;
  a.rows = cur;
  if (rg) {
    a.n_new = rg->n;
    a.first_seg = rg->first_seg;
    a.n_layers = c.num_decoder_layers;
    a.row_bytes = static_cast<size_t>(c.num_heads) * c.input_length * 64 * e->kv_esize;
    a.sc_bytes = static_cast<size_t>(c.num_heads) * c.input_length * sizeof(float2);
    const size_t chunk = static_cast<size_t>(rg->seq % kStageChunks);
    for (int l = 0; l < c.num_decoder_layers; ++l) {
      a.src[l] = static_cast<const char*>(e->stage_kv[l]) + chunk * 2 * e->stage_cap * a.row_bytes;
      a.dst[l] = static_cast<char*>(e->dec[l].cross_kv);
      if (e->kv_fp8) {
        a.src_sc[l] = reinterpret_cast<const char*>(e->stage_scale[l]) + chunk * e->stage_cap * a.sc_bytes;
        a.dst_sc[l] = reinterpret_cast<char*>(e->dec[l].cross_scale);
      }
    }
    a.src_batch = rg->batch;
    a.src_entry0 = rg->entry0;
    a.dst_batch = r.batch;
  }
  return mt3k::launch_refill(a, r.s);
}

// One row group's loop of mt3_engine_transcribe: as run_group, but a finished slot restarts on the next staged segment
// at the poll, and the loop ends when the queue is empty for good and every slot of the group has finished.
static int run_group_stream(mt3_engine* e, GroupRun& r, Feed& f, int32_t* d_out, int kPoll) {
  int cur = r.rows;
  hipGraphExec_t exec = nullptr;
  int exec_rows = -1, flushed_at = -1;
  std::vector<FeedRange> held, got(kStageChunks + 2);
  r.ran = 0;
  r.used_graph = r.use_graph;
  // The poll is PIPELINED: at the end of interval j the group's counter of finished slots is copied to pinned memory and
  // an event recorded behind the copy -- and the host goes straight on to enqueue interval j + 1; it looks at snapshot j
  // only after that (sleeping until the event if need be), so the device always has an interval of steps queued and never
  // waits for the host's answer (round 5: with drained polls a sleeping host cost 1.3 % at 256 slots).  What the
  // snapshot triggers -- refills, the hand-over of finished ids, a compaction -- is enqueued BEHIND interval j + 1 and in
  // front of snapshot j + 1, so every snapshot already accounts for it.  The kernels act on the device's state when they
  // run (decode_ops.hip: the plan kernels scan the done flags), the snapshot only says how many staged segments to hand
  // out: it may be an interval stale, never wrong (finished slots stay finished until a refill restarts them).
  int parity = 0;
  bool pending = false;
  // Watchdog on PROGRESS, not on the step count (a group that is starved by a slow encoder keeps stepping its live
  // slots, so its counter grows with wall time): a live slot finishes within num_steps steps of its (re)start and a
  // snapshot is at most two poll intervals old, so `stall_limit` steps without a refill, a newly finished slot or a
  // compaction can only mean a slot that never terminates.
  const long stall_limit = static_cast<long>(r.num_steps) + 4L * kPoll + 64;
  long last_progress = 0;
  int seen_fin = 0;
  for (long t = 0;; ++t) {
    if (t - last_progress > stall_limit)
      return mt3::fail(MT3_ERR_INVALID, "mt3_engine_transcribe: a row group made no progress for num_steps + 4 polls");
    if (r.use_graph && exec_rows != cur) {
      exec = group_graph(e, r.variant, r.batch, r.row0, cur, r.slot);
      exec_rows = cur;
      if (!exec) {
        r.use_graph = r.used_graph = false;
        ++e->graph_fallbacks;
      }
    }
    if (r.use_graph) MT3_HIP_CHECK(hipGraphLaunch(exec, r.s));
    else MT3_TRY(enqueue_chain_step(e, r.row0, cur, r.slot, r.batch, r.variant, r.s, r.slot));
    ++r.ran;
    if (t % kPoll != kPoll - 1) continue;
    // ---- the previous interval's snapshot
    if (pending) {
      hipEvent_t pev = wait_event(e, r.slot, 2 + (parity ^ 1));
      MT3_HIP_CHECK(e->spin_waits ? hipEventSynchronize(pev) : sleep_until(pev));
      feed_release(f, held);                     // what was taken one snapshot ago was copied out in front of this one
      held.clear();
      int n_fin = e->h_pinned[r.slot + kMaxGroups * (parity ^ 1)];   // finished slots among the group's r.rows (dropped ones included)
      if (n_fin != seen_fin) last_progress = t;
      bool dry = false;
      for (;;) {
        if (mt3feed::feed_failed(f)) return MT3_OK;           // somebody else reports the error
        const int refillable = n_fin - (r.rows - cur);
        const int nr = feed_pop(f, refillable, got.data(), static_cast<int>(got.size()), &dry);
        for (int i = 0; i < nr; ++i) {
          MT3_TRY(refill_group(e, r, cur, &got[i], d_out));
          held.push_back(got[i]);
          n_fin -= got[i].n;
          last_progress = t;
        }
        if (dry || n_fin < r.rows) break;
        feed_wait(f);                            // nothing live and the encoder is behind: sleep, do not spin through empty steps
      }
      if (dry) {
        // the queue is empty for good: finished slots hand over their ids, the live ones are compacted as under EARLY_EXIT
        const int live = r.rows - n_fin;
        int want = (live + 31) & ~31;
        if (want > r.rows) want = r.rows;
        const bool compact = live > 0 && want < cur;
        // (the hand-over runs on the device's own done flags, directly in front of a compaction or of the exit: no slot
        // that finished after the snapshot can be dropped with its ids still in the engine)
        if ((n_fin - (r.rows - cur) > 0 && n_fin != flushed_at) || compact || live <= 0) {
          MT3_TRY(refill_group(e, r, cur, nullptr, d_out));
          flushed_at = n_fin;
        }
        if (live <= 0) break;
        if (compact) {
          MT3_TRY(compact_group(e, r, cur));
          cur = want;
          ++e->compactions_now;
          last_progress = t;
        }
      }
      seen_fin = n_fin;
    }
    // ---- this interval's snapshot
    hipEvent_t ev = wait_event(e, r.slot, 2 + parity);
    if (!ev) return mt3::fail(MT3_ERR_HIP, "mt3_engine_transcribe: could not create a poll event");
    MT3_HIP_CHECK(hipMemcpyAsync(e->h_pinned + r.slot + kMaxGroups * parity, e->n_done + r.slot, 4, hipMemcpyDeviceToHost, r.s));
    MT3_HIP_CHECK(hipEventRecord(ev, r.s));
    pending = true;
    parity ^= 1;
  }
  MT3_HIP_CHECK(wait_stream(e, r.slot, r.s));
  feed_release(f, held);
  return MT3_OK;
}

// the producer side of the feed (calling thread, caller's stream)
static int produce_chunks(mt3_engine* e, Feed& f, const float* d_inputs, hipStream_t s, bool skip_encoder) {
  const mt3_engine_config& c = e->cfg;
  const size_t seg_floats = static_cast<size_t>(c.input_length) * c.input_depth;
  const size_t row = static_cast<size_t>(c.num_heads) * c.input_length * 64;
  const int min_batch = e->stage_cap < kStageMinBatch ? e->stage_cap : kStageMinBatch;
  std::vector<void*> kv(c.num_decoder_layers);
  std::vector<float2*> sc(c.num_decoder_layers, nullptr);
  int rc = MT3_OK;
  for (int q = 0; rc == MT3_OK; ++q) {
    int first, n, pad;
    if (!mt3feed::feed_claim(f, q, e->stage_cap, min_batch, &first, &n, &pad)) break;
    const size_t chunk = static_cast<size_t>(q % kStageChunks);
    for (int l = 0; l < c.num_decoder_layers; ++l) {
      kv[l] = static_cast<char*>(e->stage_kv[l]) + chunk * 2 * e->stage_cap * row * e->kv_esize;
      if (e->kv_fp8) sc[l] = e->stage_scale[l] + chunk * e->stage_cap * c.num_heads * c.input_length;
    }
    CrossDst dst;
    dst.kv = kv.data();
    dst.scale = e->kv_fp8 ? sc.data() : nullptr;
    // (skip_encoder: mt3_debug_engine_transcribe's differential timing -- the chunk goes on offer with whatever the staging
    // ring holds; under an imposed EOS schedule the decode does exactly the same work)
    if (!skip_encoder) rc = encode_impl(e, d_inputs + static_cast<size_t>(first - pad) * seg_floats, pad + n, nullptr, dst, s);
    if (rc == MT3_OK && wait_stream(e, kMaxGroups, s) != hipSuccess) rc = mt3::fail(MT3_ERR_HIP, "mt3_engine_transcribe: encoder pass failed");
    if (rc != MT3_OK) break;
    mt3feed::feed_publish(f, q, first, n, pad);
  }
  mt3feed::feed_finish(f, rc != MT3_OK);
  return rc;
}

// poll_steps / groups_override: 0 = the product's choice (mt3_debug_engine_transcribe sets them for A/B runs)
static int transcribe_impl(mt3_engine* e, const float* d_inputs, int32_t n_segments, int32_t num_steps, int32_t flags,
                           int32_t* d_ids, mt3_transcribe_stats* h_stats, void* stream, int poll_steps, int groups_override,
                           bool skip_encoder = false) {
  if (!e || !e->finalized) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_transcribe: engine not finalized");
  if (e->pending.active)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_transcribe: a decode is in flight (MT3_DECODE_ASYNC): call mt3_engine_decode_wait first");
  const mt3_engine_config& c = e->cfg;
  if (!d_inputs || !d_ids || n_segments <= 0) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_transcribe: null buffer or no segments");
  if (num_steps <= 0 || num_steps > c.max_decode_len) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_transcribe: num_steps out of range");
  if (flags & ~(MT3_DECODE_NO_GRAPH | MT3_DECODE_BEAM1 | MT3_DECODE_SINGLE_STREAM))
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_transcribe: flags are MT3_DECODE_NO_GRAPH | MT3_DECODE_BEAM1 | MT3_DECODE_SINGLE_STREAM");
  if (c.num_decoder_layers > mt3k::kRefillMaxLayers) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_transcribe: at most 16 decoder layers");
  if (e->eos_on && n_segments > e->eos_cap)
    return mt3::fail(MT3_ERR_INVALID, "mt3_engine_transcribe: the synthetic EOS schedule is shorter than n_segments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = c.max_decode_len, S = n_segments < c.max_batch ? n_segments : c.max_batch;
  const bool beam1 = (flags & MT3_DECODE_BEAM1) != 0;
  if (n_segments > S) MT3_TRY(ensure_stage(e));
  int groups = ((flags & MT3_DECODE_SINGLE_STREAM) || (c.options & MT3_OPT_NO_ROW_GROUPS)) ? 1 : stream_row_groups_for(c, S);
  if (groups_override > 0) groups = groups_override;
  while (groups > 1 && S / groups < 16) --groups;
  const int poll = poll_steps > 0 ? poll_steps : kStreamPollSteps;
  if (ensure_group_streams(e, groups) != MT3_OK)
    return mt3::fail(MT3_ERR_HIP, "mt3_engine_transcribe: could not create the row groups' streams");

  // ---- the first S segments go straight into the caches; every slot starts as in mt3_engine_decode
  MT3_TRY(encode_impl(e, d_inputs, S, nullptr, CrossDst{}, s));
  e->cur_batch = S;
  MT3_HIP_CHECK(hipMemsetAsync(d_ids, 0, static_cast<size_t>(n_segments) * L * 4, s));
  MT3_HIP_CHECK(hipMemsetAsync(e->step, 0, static_cast<size_t>(S) * 4, s));
  MT3_HIP_CHECK(hipMemsetAsync(e->n_done, 0, 4 * kMaxChains, s));
  MT3_HIP_CHECK(hipMemsetAsync(e->done, 0, static_cast<size_t>(S) * 4, s));
  MT3_HIP_CHECK(hipMemsetAsync(e->cur_tok, 0, static_cast<size_t>(S) * 4, s));
  MT3_HIP_CHECK(hipMemsetAsync(e->ids, 0, static_cast<size_t>(S) * L * 4, s));
  {
    const mt3k::RowProj rp{e->ew0, e->pw0, e->qkv_fold ? e->qkvf : nullptr, 4 * e->HD()};
    MT3_TRY(mt3k::launch_embed(e->embedding, e->pos_table, e->cur_tok, e->step, e->y,
                               c.compute_dtype == MT3_BF16 ? e->y_ct : nullptr, e->y_ss, S, c.emb_dim, rp, s));
  }
  MT3_TRY(mt3k::launch_iota(e->slot_row, S, s));
  MT3_TRY(mt3k::launch_iota(e->slot_seg, S, s));         // slot i starts on segment i
  if (beam1) {
    MT3_TRY(mt3k::launch_set_float(e->beam_cfg, brevity_penalty(num_steps + 1), s));
    MT3_HIP_CHECK(hipMemsetAsync(e->beam_f, 0, static_cast<size_t>(2) * c.max_batch * 4, s));
    MT3_HIP_CHECK(hipMemsetAsync(e->beam_len, 0xFF, static_cast<size_t>(c.max_batch) * 4, s));
    MT3_HIP_CHECK(hipMemsetAsync(e->beam_len_row, 0xFF, static_cast<size_t>(c.max_batch) * 4, s));
  }
  if (e->group_graphs.size() > 96) drop_group_graphs(e);
  e->stream_max_len = num_steps;
  const int variant = (beam1 ? kVarBeam : 0) | kVarRetire | kVarStream | (e->eos_on ? kVarEos : 0);
  const bool use_graph = !(flags & MT3_DECODE_NO_GRAPH);
  MT3_HIP_CHECK(hipEventRecord(e->part_begin, s));

  Feed feed;
  feed.n_total = n_segments;
  feed.next_seg = S;
  feed.finished = n_segments == S;
  PendingDecode& p = e->pending;
  p = PendingDecode();
  p.groups = groups;
  p.active = true;                                       // every other entry point of the engine refuses meanwhile
  bool posted_all = true;
  for (int g = 0; g < groups && posted_all; ++g) {
    auto body = [e, g, groups, S, variant, num_steps, use_graph, &feed, d_ids, poll]() {
      PendingDecode& q = e->pending;
      GroupRun r{};
      chain_rows(S, groups, g, &r.row0, &r.rows);
      r.batch = S;
      r.variant = variant | (groups > 1 ? kVarBeside : 0);
      r.num_steps = num_steps;
      r.slot = g;
      r.early = true;
      r.use_graph = use_graph;
      r.s = e->part_stream[g];
      hipError_t he = hipStreamWaitEvent(r.s, e->part_begin, 0);
      if (he == hipSuccess) {
        q.rcs[g] = run_group_stream(e, r, feed, d_ids, poll);
        if (q.rcs[g] != MT3_OK) q.errs[g] = mt3_last_error();
        he = wait_stream(e, g, r.s);
      }
      if (q.rcs[g] == MT3_OK && he != hipSuccess) {
        q.rcs[g] = MT3_ERR_HIP;
        q.errs[g] = hipGetErrorString(he);
      }
      q.ran[g] = r.ran;
      q.used_graph[g] = r.used_graph;
      if (q.rcs[g] != MT3_OK) feed_fail(feed);           // nobody may wait for this group's entries any more
    };
    if (worker_post(e, g, body)) p.posted = g + 1;
    else posted_all = false;
  }
  int rc = MT3_OK;
  if (!posted_all) {
    feed_fail(feed);
    rc = mt3::fail(MT3_ERR_HIP, "mt3_engine_transcribe: could not start a row group's worker thread");
  } else if (n_segments > S) {
    rc = produce_chunks(e, feed, d_inputs, s, skip_encoder);
  }
  const std::string producer_err = rc != MT3_OK ? mt3_last_error() : "";
  for (int g = 0; g < p.posted; ++g) worker_wait(e, g);
  p.active = false;
  e->compactions = e->compactions_now.exchange(0);
  e->last_groups = groups;
  int most = 0;
  e->last_used_graph = 1;
  for (int g = 0; g < p.posted; ++g) {
    if (rc == MT3_OK && p.rcs[g] != MT3_OK)
      rc = mt3::fail(p.rcs[g], "mt3_engine_transcribe (row group " + std::to_string(g) + "): " + p.errs[g]);
    most = p.ran[g] > most ? p.ran[g] : most;
    if (!p.used_graph[g]) e->last_used_graph = 0;
  }
  mt3_transcribe_stats st{};
  st.slots = S;
  st.groups = groups;
  st.steps_run = most;
  st.polls = feed.polls;
  st.refills = feed.refills;
  st.starved_polls = feed.starved;
  st.encoder_chunks = feed.produced;
  st.compactions = e->compactions;
  st.used_graph = e->last_used_graph;
  if (h_stats) *h_stats = st;
  if (rc != MT3_OK && !producer_err.empty()) return mt3::fail(rc, producer_err);
  return rc;
}

int mt3_engine_transcribe(mt3_engine* e, const float* d_inputs, int32_t n_segments, int32_t num_steps, int32_t flags,
                          int32_t* d_ids, mt3_transcribe_stats* h_stats, void* stream) {
  return transcribe_impl(e, d_inputs, n_segments, num_steps, flags, d_ids, h_stats, stream, 0, 0);
}

int mt3_debug_engine_transcribe(mt3_engine* e, const float* d_inputs, int32_t n_segments, int32_t num_steps, int32_t flags,
                                int32_t poll_steps, int32_t row_groups, int32_t skip_encoder_passes, int32_t* d_ids,
                                mt3_transcribe_stats* h_stats, void* stream) {
  if (poll_steps < 0 || poll_steps > 1024 || row_groups < 0 || row_groups > kMaxGroups)
    return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_transcribe: poll_steps in [0, 1024], row_groups in [0, 4]");
  return transcribe_impl(e, d_inputs, n_segments, num_steps, flags, d_ids, h_stats, stream, poll_steps, row_groups,
                         skip_encoder_passes != 0);
}

int mt3_engine_decode(mt3_engine* e, int32_t batch, int32_t num_steps, int32_t flags, int32_t* d_ids,
                      float* d_first_logits, int32_t* h_steps_run, void* stream) {
  return decode_impl(e, batch, num_steps, flags, 0, nullptr, nullptr, d_ids, d_first_logits, h_steps_run, stream);
}

int mt3_engine_decode_wait(mt3_engine* e, int32_t* h_steps_run) {
  if (!e || !e->finalized) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode_wait: engine not finalized");
  return decode_finish(e, h_steps_run);
}

int mt3_engine_decode_forced(mt3_engine* e, int32_t batch, int32_t num_steps, int32_t flags,
                             const int32_t* d_forced_ids, float* d_step_logits, int32_t* d_ids, void* stream) {
  if (!d_forced_ids) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_decode_forced: null forced ids");
  return decode_impl(e, batch, num_steps, flags, 0, d_forced_ids, d_step_logits, d_ids, nullptr, nullptr, stream);
}

// ---- mt3_hip_debug.h
int mt3_debug_engine_decode(mt3_engine* e, int32_t batch, int32_t num_steps, int32_t flags, int32_t skip,
                            int32_t* d_ids, void* stream) {
  if (skip & ~(MT3_DEBUG_SKIP_SELF_ATTN | MT3_DEBUG_SKIP_CROSS_ATTN))
    return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_decode: unknown skip bit");
  if (flags & MT3_DECODE_ASYNC) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_decode: not with MT3_DECODE_ASYNC");
  return decode_impl(e, batch, num_steps, flags, skip, nullptr, nullptr, d_ids, nullptr, nullptr, stream);
}

int mt3_debug_engine_set_eos_schedule(mt3_engine* e, const int32_t* h_lengths, int32_t n) {
  if (!e || !e->finalized) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_set_eos_schedule: engine not finalized");
  if (e->pending.active) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_set_eos_schedule: a decode is in flight");
  if (!h_lengths) {
    e->eos_on = false;
    return MT3_OK;
  }
  if (n <= 0) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_set_eos_schedule: n must be positive");
  if (n > e->eos_cap) {
    // a schedule per SEGMENT of an mt3_engine_transcribe call: a larger array (the captured step graphs hold the old
    // address, so they go; the old array stays in the engine's allocation list until destroy)
    int* grown = nullptr;
    MT3_TRY(dmalloc(e, reinterpret_cast<void**>(&grown), static_cast<size_t>(n) * 4));
    drop_graph(e);
    drop_group_graphs(e);
    e->eos_at = grown;
    e->eos_cap = n;
  }
  std::vector<int32_t> h(static_cast<size_t>(e->eos_cap), 0x7fffffff);       // rows / segments past n: never
  for (int i = 0; i < n; ++i) {
    if (h_lengths[i] < 1) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_set_eos_schedule: lengths must be >= 1");
    h[i] = h_lengths[i];
  }
  MT3_HIP_CHECK(hipMemcpy(e->eos_at, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  e->eos_on = true;
  return MT3_OK;
}

int mt3_debug_engine_poison_caches(mt3_engine* e, int32_t pattern, int32_t cross, void* stream) {
  if (!e || !e->finalized) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_poison_caches: engine not finalized");
  if (e->pending.active) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_engine_poison_caches: a decode is in flight");
  const mt3_engine_config& c = e->cfg;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t heads = static_cast<size_t>(c.max_batch) * c.num_heads;
  for (LayerDev& L : e->dec) {
    const size_t kvb = heads * c.max_decode_len * 64 * e->kv_esize;
    MT3_HIP_CHECK(hipMemsetAsync(L.self_k, pattern, kvb, s));
    MT3_HIP_CHECK(hipMemsetAsync(L.self_v, pattern, kvb, s));
    if (L.self_scale) MT3_HIP_CHECK(hipMemsetAsync(L.self_scale, pattern, heads * c.max_decode_len * sizeof(float2), s));
    if (cross) {
      MT3_HIP_CHECK(hipMemsetAsync(L.cross_kv, pattern, 2 * heads * c.input_length * 64 * e->kv_esize, s));
      if (L.cross_scale) MT3_HIP_CHECK(hipMemsetAsync(L.cross_scale, pattern, heads * c.input_length * sizeof(float2), s));
    }
  }
  return MT3_OK;
}

int mt3_engine_status(const mt3_engine* e, int32_t what) {
  if (!e) return mt3::fail(MT3_ERR_INVALID, "mt3_engine_status: null engine");
  switch (what) {
    case MT3_STATUS_GRAPH_FALLBACKS: return e->graph_fallbacks.load();
    case MT3_STATUS_LAST_DECODE_USED_GRAPH: return e->last_used_graph;
    case MT3_STATUS_RESIDUAL_SPLIT: return e->y_split ? 1 : 0;
    case MT3_STATUS_KV_FP8: return e->kv_fp8 ? 1 : 0;
    case MT3_STATUS_DENSE_FP8: return e->dense_fp8 ? 1 : 0;
    case MT3_STATUS_Q_FOLD: return e->q_fold ? 1 : 0;
    case MT3_STATUS_QKV_FOLD: return e->qkv_fold ? 1 : 0;
    case MT3_STATUS_LAST_DECODE_GROUPS: return e->last_groups;
    case MT3_STATUS_PARTITION_FALLBACKS: return e->part_failed;
    case MT3_STATUS_LAST_DECODE_COMPACTIONS: return e->compactions;
    default: return mt3::fail(MT3_ERR_INVALID, "mt3_engine_status: unknown item");
  }
}

}  // extern "C"
