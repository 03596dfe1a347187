// Internal launcher interface between the engine (engine.hip) and the kernel files.
#ifndef MT3_KERNELS_H_
#define MT3_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mt3_hip.h"

namespace mt3k {

struct GemmArgs {
  const void* A;      // [M, lda]  f32 or compute type
  const void* Wt;     // [N, K]    compute type
  void* out;          // see epilogue
  const float* aux;   // EPI_POS: positional table [seq_len, N]
  int M, N, K;
  int lda, ldo;
  int seq_len;        // EPI_POS / EPI_HEADS: rows per batch item
  // bf16 decode path: the residual stream travels as f32 rows PLUS a compute-type copy and, per row, the sums of
  // squares of every 16-column group (exact f32), so that the RMSNorm-fused GEMMs read half the A bytes and need
  // no statistics pass of their own
  const float* a_ss;  // norm == 2: [M][K/16] partial sums of squares of the rows whose compute-type copy is A
  void* out_ct;       // EPI_RESID: also write the updated rows in the compute type here [M][ldo] (nullptr: no)
  float* out_ss;      // EPI_RESID with out_ct: [M][N/16] partial sums of squares of the updated rows
  // kEpiStoreQ / kEpiResidQ: output columns [n_split, N) are a SECOND product riding in the same launch -- they go to
  // out2 (f32 [M][N - n_split], unscaled; ResidQ accumulates into it), columns [0, n_split) take the STORE / RESID path
  float* out2;
  int n_split;
  int ld2;            // row stride of out2 in floats (0: N - n_split)
  // kEpiResidS: the tiles past n_split multiply a TWO-SOURCE row [A (K = k_split columns) | A2 (K - k_split columns)]
  // with their K-wide weight rows and STORE the product to out2; the tiles before n_split are a plain RESID over
  // the first k_split columns of their weight rows (the rest of those rows is never read)
  const void* A2;     // [M, lda2] compute type
  int lda2;
  int k_split;
  // RESID family: where the OLD value of the f32 output region is read from (nullptr: `out` itself, the update in
  // place).  The two-source launch of the f32 engine reads the rows it updates as an operand of its other tiles, so
  // there the update goes out of place
  const float* resid_src;
  // tile -> XCD dealing: 0 = an XCD owns a run of row blocks (all weight columns pass through its L2), 1 = an XCD
  // owns a run of weight-column tiles for ALL row blocks (its L2 sees 1/8 of the weights; set by launch_gemm)
  int n_major;
  // decode-sized f32 launches: this launch is one of several row groups running side by side (set by the engine), so
  // a row block of >= 256 rows takes the 64 x 32 tiles (gemm.hip: launch_tile); alone on the chip the 32-row tiles win
  int concurrent;
};
// internal epilogues (not part of the C ABI): STORE / RESID with a second f32 output region, see GemmArgs::out2
constexpr int kEpiStoreQ = 6, kEpiResidQ = 7, kEpiResidS = 8;

// norm: 0 none, 1 fused RMSNorm with statistics from the f32 A stream, 2 fused RMSNorm from g.a_ss (A = compute type)
int launch_gemm(int dtype, const GemmArgs& g, bool a_f32, int norm, int epi, bool small, hipStream_t s);

// f32 operands as three bf16 planes (gemm_x6_kernel, the f32 engine's encoder): A f32 [M][lda]; g.Wt / Wm / Wl = the hi /
// mid / lo bf16 planes [N][K] of the weight; outputs f32.  (norm, epi): (true, STORE | GEGLU), (false, RESID | POS | HEADS)
int launch_gemm_x6(const GemmArgs& g, const void* Wm, const void* Wl, bool norm, int epi, hipStream_t s);

// encoder self-attention, qkv [B, T, 3, H, 64] -> out [B, T, H*64]
int launch_encoder_attention(int dtype, const void* qkv, void* out, int B, int T, int H, hipStream_t s);

struct DecAttnArgs {
  const void* q;        // [B, q_stride] compute type; head h at +h*64
  int q_stride;
  void* kcache;         // [B, H, cap, 64]
  void* vcache;
  int cap;
  const void* new_k;    // [B, kv_stride] rows; head h at +h*64 (NULL: no append)
  const void* new_v;
  int kv_stride;
  const int* step;      // device, per row: n_keys = step[b] + 1 (NULL: use n_keys)
  int n_keys;
  void* out;            // [B, H*64] compute type
  int B, H;
  // non-null: the cache holds OCP e4m3 bytes [B, H, cap, 64] and this is its side array [B, H, cap] of
  // {k_scale, v_scale} (power-of-two row scales); q / new rows / out are bf16
  float2* kv_scale;
  // bf16 path, no append: the query arrives as UNNORMALISED f32 rows q_f32 [B][q_stride] plus the partial sums of
  // squares q_ss [B][q_ss_n] of the residual row it was projected from (emb = 16 * q_ss_n); the kernel applies
  // rsqrt(mean + 1e-6) itself (the cross-attention q-projection folded into the neighbouring GEMM launches)
  const float* q_f32;
  const float* q_ss;
  int q_ss_n;
  // Row retirement (mt3_engine_decode with MT3_DECODE_EARLY_EXIT): `done` [B] per SLOT -- a workgroup whose slot has
  // finished (EOS emitted / beam search closed) returns before it requests a single cache byte; `cache_row` [B] maps a
  // slot to the row of kcache / vcache / kv_scale it owns (the live slots are compacted to the front of the batch while
  // the caches stay where they are).  Both nullptr: slot b = row b, every row is attended (the canonical schedule).
  const int* done;
  const int* cache_row;
};
// bf16 K rows then V rows ([2][rows][64]) -> e4m3 [2][rows][64] + float2 scales [rows]
int launch_kv_quantize_fp8(const void* src_bf16, void* dst_fp8, void* scales, int rows, hipStream_t s);
int launch_decode_attention(int dtype, const DecAttnArgs& a, hipStream_t s);

// out_ct[row] = x[row] * rsqrt(mean(x^2)+eps) * scale ; optional f32 copy
int launch_rmsnorm(int dtype, const float* x, const float* scale, void* out_ct, float* out_f32, int rows, int dim,
                   hipStream_t s);
// x f32 [rows][dim] -> bf16 copy + per-16-column sums of squares (the split residual form, see GemmArgs)
int launch_residual_split(const float* x, void* x_ct, float* x_ss, int rows, int dim, hipStream_t s);
// y[b] = table[tok[b]] + pos[step[b]]
// y_ct / y_ss (both or neither): bf16 copy of the rows and their per-16-column sums of squares (see GemmArgs)
// first-layer projections by table lookup (the decoder's first QKV launch folded away): when q_out != nullptr the
// kernels that produce a decoder input row Embed(tok) + FixedEmbed[t] also write its UNNORMALISED projection
// q_out[b][0 .. q_n) = ew[tok] + pw[t]  (ew = embedding . W, pw = position table . W, both f32 [rows][q_n])
struct RowProj {
  const float* ew;
  const float* pw;
  float* q_out;
  int q_n;
};
// logits that arrive UNNORMALISED (the logits projection folded into the last layer's MLP out-projection launch): the
// kernel that picks the token applies the decoder_norm row scale rsqrt(sum(ss[b][0 .. n_ss)) / dim + 1e-6) itself and
// writes the scaled logits back (ss == nullptr: the logits are final)
struct LogitScale {
  const float* ss;
  int n_ss;
  int dim;
};
int launch_embed(const float* table, const float* pos, const int* tok, const int* step, float* y, void* y_ct,
                 float* y_ss, int B, int dim, const RowProj& rp, hipStream_t s);
// per-row state of the beam-1 search (t5x beam_search, num_decodes = 1): f = [live_logp | best finished
// score], the second array `rows` floats after the first; len = prefix length of the best finished
// hypothesis or -1; cfg[0] = brevity_penalty(max_len + 1), cfg[1 + n] = brevity_penalty(n) (device memory)
struct BeamState {
  float* f;
  int* len;
  const float* cfg;
  int rows;
  // a copy of `len` indexed by the ROW a slot decodes (what the finalisation reads once the loop is over: under row
  // retirement the slot-indexed state of a finished row is gone by then); the base of the step's rows, or -- with a
  // slot map -- of the whole batch
  int* len_row;
};
// Row retirement and the synthetic EOS schedule of one step (all nullptr / 0: the canonical schedule).
//   retire  : a block whose slot is already done returns at once (its ids stay 0, its position counter stops)
//   slot_row: [B] slot -> row of `ids` / `eos_at` (nullptr: identity); with it `ids` is the UN-offset base of the batch
//   eos_at  : [rows] bench / test hook (mt3_debug_engine_set_eos_schedule): row r's distribution at step eos_at[r] - 1
//             is replaced by a point mass on EOS -- greedy emits EOS there, the beam-1 search finishes prefix + EOS with
//             log-prob 0 and closes (its live hypothesis drops to -inf)
//   slot_seg: [B] in-flight batching (mt3_engine_transcribe): the SEGMENT a slot is decoding -- `eos_at` is then indexed by
//             segment, not by row (a cache row serves many segments in turn)
//   max_len : > 0: a slot whose position counter reaches max_len is finished whether or not it emitted EOS (a refilled slot
//             starts at position 0 at an arbitrary step of its group's loop, so the loop bound cannot do it)
struct StepRetire {
  int retire;
  const int* slot_row;
  const int* eos_at;
  const int* slot_seg;
  int max_len;
};
// token pick + bookkeeping for one decode step (see decode_ops.hip); beam == nullptr: greedy;
// forced != nullptr (greedy only): teacher forcing, the next input token is forced[b * forced_stride + t]
int launch_argmax_step(float* logits, int vocab, int* ids, int ids_stride, int* cur_tok, int* done,
                       int* n_done, int* step, const float* table, const float* pos_table, int max_pos,
                       float* y_next, void* y_ct, float* y_ss, int dim, int B, const BeamState* beam,
                       const int* forced, int forced_stride, const RowProj& rp, const LogitScale& ls,
                       const StepRetire& rt, hipStream_t s);
// Compaction of the live slots of one row group to the front of the group (row retirement): the per-slot state that
// lives from one step to the next -- the next step's input row in its three forms, layer 0's projected row, position
// counter, current token, beam-search state, the slot -> row map -- moves from slot perm[i] to slot i (i < n_live) by
// way of a scratch copy; slots [n_live, rows) are marked done.  All pointers are those of the group's first slot.
struct CompactArgs {
  int* done;
  int* slot_row;
  int* step;
  int* cur_tok;
  float* beam_f;        // [2][beam_rows] (nullptr: greedy)
  int* beam_len;
  int beam_rows;
  float* y;             // [rows][emb] f32 input rows of the next step
  void* y_ct;           // bf16 copy (nullptr: f32 engine)
  float* y_ss;          // [rows][emb / 16] (nullptr: single residual stream)
  float* qkvf;          // [rows][q_n] (nullptr: no qkv-fold)
  int emb, q_n;
  // scratch of the same shapes (slot-indexed from the group's first slot as well)
  float* s_y;
  void* s_y_ct;
  float* s_y_ss;
  float* s_qkvf;
  int* s_int;           // [rows][4]: slot_row, step, cur_tok, beam_len
  float* s_beam;        // [rows][2]
  int* perm;            // [rows + 1]: perm[i] = source slot of new slot i; perm[rows] = n_live
  int rows;             // slots of the group in use before the compaction
  // in-flight batching: the slot -> segment map travels with the slot; a dropped slot decodes nothing (-1)
  int* slot_seg;        // (nullptr: not in use)
  int* s_seg;           // scratch [rows]
};
int launch_compact(const CompactArgs& c, hipStream_t s);
// Refill of finished slots (in-flight batching, mt3_engine_transcribe): at a poll of a row group's loop every FINISHED
// slot of the group hands its id row to the caller's output (row = the segment it decoded; the beam-1 finalisation of
// that row applied on the way) and the first `n_new` of them, in ascending slot order, restart at position 0 on segments
// first_seg, first_seg + 1, ...: BOS input row in its three forms, layer 0's projected row, counters, beam state, a
// zeroed id row, and the segment's cross-attention K/V copied from the staging chunk an encoder pass left them in into
// the cache rows the slot owns (the self-attention cache needs nothing: what lies past a row's position is discarded by
// position).  The others decode nothing from then on (slot_seg = -1) until a later refill.  All slot-indexed pointers are
// those of the group's first slot; ids / beam_len_row / the caches are batch bases (reached through slot_row).
constexpr int kRefillMaxLayers = 16;
struct RefillArgs {
  int* done;
  int* slot_row;
  int* slot_seg;
  int* step;
  int* cur_tok;
  int* n_done;          // the group's counter of finished slots: decremented by the number of slots refilled
  float* beam_f;        // [2][beam_rows] (nullptr: greedy)
  int* beam_len;
  int* beam_len_row;    // batch base
  int beam_rows;
  float* y;             // [rows][emb] f32 input rows of the next step
  void* y_ct;           // bf16 copy (nullptr: f32 engine)
  float* y_ss;          // (nullptr: single residual stream)
  int emb;
  const float* table;   // token embedding (row 0 = BOS) and position table (row 0)
  const float* pos;
  RowProj rp;           // q_out = the group's qkvf rows (nullptr: no qkv-fold)
  int* ids;             // engine id rows [max_batch][ids_stride] (batch base)
  int ids_stride;
  int* out_ids;         // caller's [n_segments][ids_stride]
  int* plan;            // [rows + 1] scratch: plan[i] = i-th finished slot (ascending), plan[rows] = how many
  int rows;             // slots of the group in use
  int n_new;            // segments handed out by this call (<= finished slots)
  int first_seg;
  // cross-attention K/V: per decoder layer, staging chunk [2][src_batch][row_bytes] -> cache [2][dst_batch][row_bytes];
  // with e4m3 caches also the scale rows [src_batch][sc_bytes] -> [dst_batch][sc_bytes]
  int n_layers;
  const char* src[kRefillMaxLayers];
  char* dst[kRefillMaxLayers];
  const char* src_sc[kRefillMaxLayers];
  char* dst_sc[kRefillMaxLayers];
  int src_batch, src_entry0, dst_batch;
  size_t row_bytes, sc_bytes;
};
int launch_refill(const RefillArgs& a, hipStream_t s);
int launch_iota(int* dst, int n, hipStream_t s);
int launch_set_float(float* dst, float v, hipStream_t s);
int launch_beam1_finalize(int* ids, int L, const int* beam_len, int B, hipStream_t s);
int launch_ids_to_tokens(const int* ids, int B, int L, int num_regular, int* out, hipStream_t s);

}  // namespace mt3k
#endif  // MT3_KERNELS_H_
