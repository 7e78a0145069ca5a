// common.h — shared host/device declarations for libmarie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/marie_hip.h"

// ------------------------------------------------------------------ kernel ids (profiling)
enum MhipKernelId {
  MHIP_K_CONV_FIRST = 0,  // u8 -> normalise -> conv3x3(1->64)+ReLU+maxpool2x2
  MHIP_K_CONV_IGEMM = 1,  // NHWC implicit-GEMM conv / GEMM on MFMA, fused scale/bias/ReLU/pool
  MHIP_K_LSTM_REC = 2,    // BiLSTM recurrence (persistent over T, batch-sliced)
  MHIP_K_CTC_DECODE = 3,  // wave-shuffle argmax + softmax-max + collapse
  MHIP_K_IMAGE_OPS = 4,   // resize / max-pool / bilinear up-sample (HBM-bound, 16 B per lane)
  MHIP_K_CCL = 5,         // score-map binarise + connected components + per-component statistics
  MHIP_K_CROP_BATCH = 6,  // fragment -> gray -> Pillow-exact bicubic to height 32 -> replicate pad
  MHIP_K_ATTN = 7,        // attention decoder pieces (context, LSTM cell, arg-max)
  MHIP_K_ATTN_FLASH = 8,  // ViT softmax attention: S^T = K Q^T -> online softmax -> O^T = V^T P^T on MFMA
  MHIP_K_VIT_OPS = 9,     // LayerNorm, patch extraction, token/map moves (HBM-bound)
  MHIP_K_DET_OPS = 10,    // detector heads: anchors/decode, top-k, NMS, ROIAlign
  MHIP_K_DEC_OPS = 11,    // text decoder steps: embedding, single-query attention over caches, beam candidates
  // conv_igemm by tile shape: their launches are ALSO counted in MHIP_K_CONV_IGEMM (ProfSlot::parent), so that the fraction of
  // the MFMA peak can be recomputed per variant from a profile (names as the kernels appear in a rocprofv3 trace)
  MHIP_K_IGEMM_T64 = 12,    // conv_igemm_kernel<.., 64, ..>   512 x 64 tile
  MHIP_K_IGEMM_T128 = 13,   // conv_igemm_kernel<.., 128, ..>  256 x 128 tile
  MHIP_K_IGEMM_T256 = 14,   // conv_igemm_kernel<.., 256, ..>  256 x 256 tile
  MHIP_K_IGEMM_S128 = 15,   // conv_igemm_kernel<.., 1128, ..> 128 x 128 tile (few-row GEMMs)
  MHIP_K_IGEMM_PATCH = 16,  // conv3x3_patch_kernel            3x3 / pad 1 convolutions
  MHIP_K_CROSS_ATTN = 17,   // decoder encoder-attention over the encoder tokens themselves (absorbed K / V projections)
  MHIP_K_COUNT = 18
};

constexpr int MHIP_ZERO_BYTES = 65536;

struct ProfSlot {
  double total_ms = 0.0;
  int64_t launches = 0;
  double flops = 0.0;     // algorithmic FLOPs of the launches timed so far (MFMA kernels only)
  int parent = -1;        // slot that also accumulates this one's time / launches / flops
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct mhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // workspace (grown on demand, never inside a steady-state forward)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  void* zeros = nullptr;  // MHIP_ZERO_BYTES of zeros: source of padding taps / rows beyond M, N for LDS-DMA loads
  bool profiling = false;
  ProfSlot prof[MHIP_K_COUNT];
  std::vector<hipEvent_t> event_pool;
  // pinned staging for small host temporaries that must reach the device without draining the stream (descriptor tables of a
  // call): a ring of buffers, each guarded by the event of the copy that last read it (mhip_stage_h2d)
  struct PinnedRing {
    static constexpr int N = 4;
    void* buf[N] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[N] = {0, 0, 0, 0};
    hipEvent_t ev[N] = {nullptr, nullptr, nullptr, nullptr};
    int next = 0;
  } stage;
};
// dst_dev <- src_host (bytes) on ctx->stream through pinned memory; src_host may die when the call returns, the stream is not drained
int mhip_stage_h2d(mhip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);

int mhip_fail(mhip_ctx* ctx, int code, const char* fmt, ...);
// Teardown: wait for the device, not for ctx->stream — the caller's stream (a torch stream handed in by mhip_set_stream) may
// already be gone when an object is destroyed late (interpreter shutdown), and synchronising a dead handle aborts the process
inline void mhip_quiesce(const mhip_ctx* ctx) {
  if (ctx) (void)hipSetDevice(ctx->device);      // the device this object lives on, not whatever the calling thread last used
  (void)hipDeviceSynchronize();
}
int mhip_ensure_workspace(mhip_ctx* ctx, size_t bytes);
extern "C" int mhip_gate_signal(mhip_gate* g, mhip_ctx* ctx);   // phase_gate.hip
void mhip_prof_begin(mhip_ctx* ctx, int kid, hipEvent_t* e0);
void mhip_prof_end(mhip_ctx* ctx, int kid, hipEvent_t e0);

#define MHIP_HIP(ctx, call)                                                              \
  do {                                                                                   \
    hipError_t _e = (call);                                                              \
    if (_e != hipSuccess)                                                                \
      return mhip_fail((ctx), MHIP_EHIP, "%s failed: %s (%s:%d)", #call,                 \
                       hipGetErrorString(_e), __FILE__, __LINE__);                       \
  } while (0)

// RAII-less launch bracket: PROF_LAUNCH(ctx, kid, kernel<<<...>>>(...));
#define PROF_LAUNCH(ctx, kid, ...)                    \
  do {                                                \
    hipEvent_t _e0 = nullptr;                         \
    if ((ctx)->profiling) mhip_prof_begin((ctx), (kid), &_e0); \
    __VA_ARGS__;                                      \
    if ((ctx)->profiling) mhip_prof_end((ctx), (kid), _e0);    \
  } while (0)

// ------------------------------------------------------------------ conv / GEMM launcher
// Implicit-GEMM convolution over NHWC activations:  out[m][n] = act( scale[n] * sum_k A[m][k] W[n][k] + bias[n] )
// with A[m][k] = in[b][y+dy-pad][x+dx-pad][c], k = (dy*KW+dx)*Cin + c, and an optional max-pool
// folded into the epilogue (rows are enumerated so that a pooling window is 4 / 2 consecutive m).
enum PoolMode { POOL_NONE = 0, POOL_2x2 = 1, POOL_2x1 = 2 };
enum ActMode { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };   // values of ConvDesc::relu

struct ConvDesc {
  const void* in = nullptr;     // [B][H][W][Cin]  element type T
  const void* w = nullptr;      // [N][KH*KW*Cin]  element type T (K contiguous)
  const float* scale = nullptr; // [N] or nullptr (= 1)
  const float* bias = nullptr;  // [N] or nullptr (= 0)
  void* out = nullptr;          // [B][Hp][Wp][N]  T, or fp32 if out_f32
  int B = 0, H = 0, W = 0, Cin = 0;
  int KH = 1, KW = 1, pad = 0;
  int N = 0;
  int pool = POOL_NONE;
  int relu = 0;                 // ActMode
  int out_f32 = 0;
  int sy = 1;                   // vertical stride (horizontal stride is 1)
  int pad_x = -1;               // horizontal padding; -1 = same as `pad`
  const void* res = nullptr;    // residual (same shape/type as out: fp32 when out_f32), added before the ReLU; unpooled only
  int ldc = 0;                  // output row pitch in elements; 0 = N
  int pad_cols_writable = 0;    // 1: with a pitch >= roundup(N, 8) the columns [N, roundup(N, 8)) of every output row belong to this
                                //    call and may be overwritten (with zeros): lets a ragged N take the 16-byte store path
  int row_period = 0;           // > 0 (unpooled GEMMs): output row = (q / period) * row_stride + row_offset + q % period,
  int row_stride = 0;           //      residual row = q % period (see igemm_common.h)
  int row_offset = 0;
  int dil = 1;                  // filter dilation
  const void* in2 = nullptr;    // optional second input: channels [Cin1, Cin) of a 1x1 conv over cat(in, in2)
  int Cin1 = 0;
  // ---- LayerNorm folded around a plain f16 GEMM (the ViT encoder blocks, vit_api.hip) ----
  // The residual stream x is kept as two f16 planes, x = hi + lo (hi = f16(x), lo = f16(x - hi): ~22 significant bits in the
  // bytes of one fp32).  hi IS the GEMM operand of the next product: LN(x) W^T = rstd (hi (g*W)^T - mean colsum(g*W)) + W b
  // up to the rounding of x, so no normalised copy of the stream is ever written.
  //   EPI_SPLIT    (producer: proj / fc2 / patch embedding) out = hi plane, out2 = lo plane, res / res2 the residual's planes;
  //                `stats` receives (sum, centred sum of squares) of every output row per 64-column chunk
  //   EPI_LN_ROWS  (consumer, tokens are GEMM rows: q|k, fc1)  out = act(rstd[m] acc - (mean rstd)[m] cs[n] + bias[n])
  //   EPI_LN_COLS  (consumer, tokens are GEMM columns: V^T = W_v X^T)  out = rstd[n] acc - (mean rstd)[n] cs[m] + row_bias[m]
  int epi = 0;
  const float* ln_a = nullptr;
  const float* ln_b = nullptr;
  const float* ln_cs = nullptr;
  const float* row_bias = nullptr;
  void* out2 = nullptr;
  const void* res2 = nullptr;
  float* stats = nullptr;
  int stats_ld = 0;
};
enum { EPI_NONE = 0, EPI_LN_ROWS = 1, EPI_LN_COLS = 2, EPI_SPLIT = 3 };
// precision: MHIP_PREC_F16 / MHIP_PREC_F32.  Returns 0 or negative error.
int mhip_launch_conv_igemm(mhip_ctx* ctx, int precision, const ConvDesc& d);
double mhip_conv_flops(const ConvDesc& d);

// first layer: u8 [B][H][W] -> normalise -> conv3x3 pad1 (1->64) + bias + ReLU + maxpool 2x2 -> [B][H/2][W/2][64] T
int mhip_launch_conv_first(mhip_ctx* ctx, int precision, const uint8_t* crops, const float* w9x64,
                           const float* bias64, void* out, int B, int H, int W);

// BiLSTM recurrence.  xproj fp32 [B][T][2][1024] with each direction's 1024 columns ordered
// [wave][u][unit16][gate] (see mhip_lstm_xproj_row; input projection incl. both biases),
// wpack = W_hh of both directions in MFMA fragment order, hseq out [B][T][512] T.
int mhip_launch_lstm_rec(mhip_ctx* ctx, int precision, const float* xproj, const void* wpack, void* hseq,
                         int B, int T);
size_t mhip_lstm_wpack_bytes(int precision);
// PyTorch gate-matrix row that lands in (per-direction) xproj column `col` (gate-interleaved layout)
int mhip_lstm_xproj_row(int col);
// host-side packing of W_hh (fwd, bwd: [1024][256] fp32, gate order i,f,g,o) into fragment order
void mhip_lstm_pack_whh(int precision, const float* whh_fwd, const float* whh_bwd, void* dst);

// greedy CTC decode
int mhip_launch_ctc_decode(mhip_ctx* ctx, const float* logits, int n, int T, int C, int32_t* argmax,
                           int32_t* tokens, int32_t* lengths, float* conf);

// ------------------------------------------------------------------ detector image ops (image_ops.hip)
void mhip_resize_linear_tables(int src, int dst, std::vector<int>& ofs, std::vector<short>& coef);
int mhip_launch_resize_linear_u8(mhip_ctx* ctx, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw,
                                 const int* xofs, const short* xa, const int* yofs, const short* yb);
int mhip_launch_conv_rgb_first(mhip_ctx* ctx, int precision, const uint8_t* img, int th, int tw, int H, int W,
                               const float* w27x64, const float* scale, const float* bias, void* out);
int mhip_launch_maxpool(mhip_ctx* ctx, int precision, int k, const void* in, void* out, int B, int H, int W, int C);
int mhip_launch_score_head(mhip_ctx* ctx, int precision, const void* in, int in_stride, const float* w1,
                           const float* b1, const float* w2, const float* b2, float* scores, long long npix);
int mhip_launch_upsample_bilinear(mhip_ctx* ctx, int precision, const void* in, void* out, int B, int Hi, int Wi,
                                  int C, int Ho, int Wo);

// ------------------------------------------------------------------ score-map post-processing (ccl.hip)
// scores fp32 [H][W][2] (text, link) -> flags u8 (bit0 text > low_text, bit1 link > link_thr), labels int32
// (0 = background, components numbered in raster order of their first pixel), n_labels (incl. background) and
// per-component stats int32 [n][6] = {left, top, right, bottom, area, max text score as ordered int bits}.
struct CclBuffers {
  uint8_t* flags;   // [H*W]
  int* parent;      // [H*W]  union-find forest, then root index per pixel
  int* labels;      // [H*W]
  int* blocksum;    // [ceil(H*W/2048) + 1]
  int* stats;       // [max_labels][6]
  int* n_labels;    // [1]
};
size_t mhip_ccl_workspace_bytes(int H, int W);
void mhip_ccl_carve(char* base, int H, int W, CclBuffers* out);
int mhip_launch_ccl(mhip_ctx* ctx, const float* scores, int H, int W, float low_text, float link_thr,
                    const CclBuffers& b);
float mhip_ordered_bits_to_float(int bits);

// ------------------------------------------------------------------ crop batcher (crop_batch.hip)
int mhip_crop_resized_width(int w, int h, int img_h, int img_w);
size_t mhip_crop_scratch_bytes(const mhip_crop_desc* descs, int n, int img_h, int img_w);
int mhip_launch_crop_batch(mhip_ctx* ctx, const uint8_t* base_dev, const mhip_crop_desc* descs, int n, int img_h,
                           int img_w, void* scratch_dev, uint8_t* out_dev);

// ------------------------------------------------------------------ production ICR recognizer ops (icr_ops.hip)
int mhip_launch_conv_gray_first(mhip_ctx* ctx, int precision, int in_is_u8, const void* img, int B, int H, int W,
                                int C, const float* w9xC, const float* scale, const float* bias, void* out);
int mhip_launch_maxpool_s21_p01(mhip_ctx* ctx, int precision, const void* in, void* out, int B, int H, int W, int C);
int mhip_launch_avgpool_hw(mhip_ctx* ctx, int precision, const void* in, void* out, int B, int HW, int C);
int mhip_launch_tps_sample(mhip_ctx* ctx, const uint8_t* crops, const float* cprime, const float* inv_delta_c,
                           const float* p_hat, float* out, int B, int H, int W, int F);
int mhip_launch_attn_context(mhip_ctx* ctx, int precision, const float* hproj, const float* hp, int ld_hp,
                             const float* score_w, const void* H, void* ctx_out, int B, int Tn);
int mhip_launch_attn_cell(mhip_ctx* ctx, int precision, const float* gctx, const float* ghid, int ld_hp,
                          const float* w_onehot, const int* chars, float* c, void* h, int B);
int mhip_launch_argmax_rows(mhip_ctx* ctx, const float* logits, int ld, int C, int* idx, int B);
int mhip_launch_rowmax_softmax(mhip_ctx* ctx, const float* logits, int rows, int C, int* idx, float* pmax);

// ------------------------------------------------------------------ ViT encoder ops (vit_ops.hip)
// x: the residual stream, fp32 — or f16 (x_f16, f16 mode only)
// x_lo: the low plane when the stream is kept as two f16 planes (x = hi + lo; ConvDesc::epi), x then being the high plane
int mhip_launch_layernorm(mhip_ctx* ctx, int precision, const void* x, const float* g, const float* b, void* out,
                          int rows, int D, float eps, int x_f16 = 0, const void* x_lo = nullptr);
// the split stream (two f16 planes + row statistics per 64-column chunk, stats[(chunk * stats_ld + row) * 2]):
int mhip_launch_token_init_split(mhip_ctx* ctx, void* hi, void* lo, const float* cls_row, const float* cls_stats, float* stats,
                                 int stats_ld, int B, int npad, int n_tok, int D);
int mhip_launch_ln_finalize(mhip_ctx* ctx, const float* stats, int chunks, int ld, float* rstd, float* mur, int rows, int D, float eps);
int mhip_launch_split_f16(mhip_ctx* ctx, const float* in, void* hi, void* lo, long long n);
int mhip_launch_join_f16(mhip_ctx* ctx, const void* hi, const void* lo, float* out, long long n);
// softmax(Q K^T) V for `images` x `heads` independent (head_dim 64) problems; q is pre-scaled by head_dim^-0.5 * log2(e).
struct AttnDesc {
  const void* q = nullptr;    // [images*npad_q][ldq] T, head h at column h*64
  const void* k = nullptr;    // [images*npad_k][ldk] T
  const void* vt = nullptr;   // [heads*64][ldv] T: V transposed, column = image*npad_k + key
  void* out = nullptr;        // [images*npad_q][ldo] T
  int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
  int images = 0, heads = 0;
  int npad_q = 0, npad_k = 0;  // rows per image (npad_k % 8 == 0).  The last 128-query block / 64-key tile of an image may
                               // read up to 127 rows past it: q, k need that many finite rows of slack, vt 64 elements
  int n_queries = 0, n_keys = 0;
};
int mhip_launch_attention(mhip_ctx* ctx, int precision, const AttnDesc& d);
double mhip_attention_flops(const AttnDesc& d);
int mhip_launch_patchify(mhip_ctx* ctx, int precision, const uint8_t* imgs, int B, int th, int tw, int hp, int wp, int P,
                         int swap_rb, float mean, float stdv, void* out, int ld);
int mhip_launch_token_init(mhip_ctx* ctx, void* x, const float* cls_row, int B, int npad, int n_tok, int D, int x_f16 = 0);
int mhip_launch_tokens_to_map(mhip_ctx* ctx, int precision, const void* x, void* out, int B, int npad, int np, int D, int x_f16 = 0);
int mhip_launch_posemb_bicubic(mhip_ctx* ctx, const float* tab, int gh, int gw, float* out, int hp, int wp, int D);
int mhip_launch_convert_rows(mhip_ctx* ctx, int precision, const void* in, float* out, int rows, int D);
int mhip_launch_narrow_f16(mhip_ctx* ctx, const float* in, void* out, long long n);
int mhip_launch_unnest(mhip_ctx* ctx, int precision, const void* in, const void* coarse, void* out, int out_f32, int B,
                       int H, int W, int C, int nest);

// ------------------------------------------------------------------ Pillow-exact resize (pil_resize.hip)
size_t mhip_pil_resize_scratch_bytes(int sh, int sw, int dh, int dw, int filter);
int mhip_launch_pil_resize_rgb(mhip_ctx* ctx, const uint8_t* src, int sh, int sw, size_t src_stride, uint8_t* dst, int dh,
                               int dw, int filter, void* scratch);

// ------------------------------------------------------------------ detector heads (det_ops.hip)
void mhip_rpn_cell_anchors(const float sizes[5], const float ratios[3], float out[5][3][4]);
struct RpnDesc {
  const float* head[5];       // per level fp32 [images][H*W][16]: 3 objectness logits, 3 x 4 deltas, 1 pad
  int H[5], W[5], stride[5];
  float cell[5][3][4];
  int images = 1;
  int img_h = 0, img_w = 0;   // resized image size (proposal clip)
  float nms_thr = 0.7f;
  int post_topk = 1000;       // <= 1000
  float* lvl_boxes = nullptr;   // scratch [images][5][1000][4]
  float* lvl_scores = nullptr;  // scratch [images][5][1000]
  int* lvl_counts = nullptr;    // scratch [images][5]
  float* out_boxes = nullptr;   // [images][post_topk][4]
  float* out_scores = nullptr;  // [images][post_topk]
  int* out_counts = nullptr;    // [images]
};
int mhip_launch_rpn_proposals(mhip_ctx* ctx, const RpnDesc& d);
struct RoiDesc {
  const void* feat[4];   // p2..p5 NHWC T [images][H][W][C]
  int H[4], W[4];
  float scale[4];
  const float* rois = nullptr;   // [images][max_rois][4]
  const int* counts = nullptr;
  int images = 1, max_rois = 1000, C = 256;
  void* out = nullptr;           // [images][max_rois][49*C] T
};
int mhip_launch_roi_align(mhip_ctx* ctx, int precision, const RoiDesc& d);
struct DetFinalDesc {
  const float* head = nullptr;   // [images][max_rois][8]
  const float* rois = nullptr;
  const int* counts = nullptr;
  int images = 1, max_rois = 1000;
  int img_h = 0, img_w = 0, out_h = 0, out_w = 0;
  float score_thr = 0.05f, nms_thr = 0.5f;
  int max_det = 2000;
  float* out_boxes = nullptr;    // [images][max_rois][4]
  float* out_scores = nullptr;   // [images][max_rois]
  int* out_count = nullptr;      // [images]
};
int mhip_launch_det_final(mhip_ctx* ctx, const DetFinalDesc& d);
int mhip_launch_blackout(mhip_ctx* ctx, uint8_t* page, int H, int W, const int* boxes_dev, int n, int* changed_dev);
int mhip_launch_subsample2(mhip_ctx* ctx, int precision, const void* in, void* out, int B, int H, int W, int C);

// ------------------------------------------------------------------ TrOCR decoder steps (trocr_ops.hip)
int mhip_launch_layernorm2(mhip_ctx* ctx, int precision, const float* x, const float* g, const float* b, float* y_f32,
                           void* y_t, int rows, int D, float eps);
int mhip_launch_embed_step(mhip_ctx* ctx, int precision, const int* tokens, const void* emb, const float* pos_row, float scale,
                           const float* g, const float* b, float* x, void* xt, int rows, int D, float eps);
struct DecAttnDesc {
  const void* q = nullptr;   // [groups*nq][ldq] T, pre-scaled
  const void* k = nullptr;
  const void* v = nullptr;
  void* out = nullptr;       // [groups*nq][ldo] T
  const int* anc = nullptr;  // self-attention: [rows][anc_ld] cache slot per past step (k/v laid out [step][slots][ldk])
  int anc_ld = 0, slots = 0;
  int kv_rows = 0;           // cross-attention: k/v rows per group
  int ldq = 0, ldk = 0, ldo = 0;
  int heads = 0, groups = 0, nq = 1, n_keys = 0;
};
int mhip_launch_decode_attention(mhip_ctx* ctx, int precision, const DecAttnDesc& d);
// Encoder-attention of one decoder layer with the key / value projections absorbed (cross_attn.hip; f16 only):
// q [crops*beam][ldq] (pre-scaled) -> qt = W_k,h^T q_h -> attention over E -> ct -> ao = W_v,h ct_h + b_v.
struct CrossAbsorbDesc {
  const void* q = nullptr;     // [crops*beam][ldq] f16
  int ldq = 0;
  const void* E = nullptr;     // [crops][kv_rows][enc_dim] f16 encoder tokens
  int kv_rows = 0, n_keys = 0, enc_dim = 0;
  const void* wkt = nullptr;   // [heads][enc_dim][64] f16: W_k[h*64 + j][d] * log2(e) at [h][d][j]
  const void* wv = nullptr;    // [heads*64][enc_dim] f16 (the checkpoint's layout)
  const float* bv = nullptr;   // [heads*64]
  void* qt = nullptr;          // scratch [crops*beam][16][enc_dim] f16
  void* ct = nullptr;          // scratch [crops*beam][16][enc_dim] f16
  void* ao = nullptr;          // [crops*beam][ldo] f16
  int ldo = 0;
  int crops = 0, beam = 1, heads = 0;
};
bool mhip_cross_absorb_supported(int enc_dim, int beam, int heads);
int mhip_launch_cross_absorbed(mhip_ctx* ctx, const CrossAbsorbDesc& d);
struct BeamCandDesc {
  const void* logits = nullptr;    // [bsz*beam][ld], fp32 or f16
  int logits_f16 = 0;
  int ld = 0, vocab = 0, beam = 1, bsz = 0;
  const float* cum = nullptr;      // [bsz*beam]
  int step = 0, max_len = 0, min_len = 1, pad = 1, eos = 2;
  float* cand_scores = nullptr;    // [bsz][2*beam]
  int* cand_tokens = nullptr;
  int* cand_beams = nullptr;
};
int mhip_launch_beam_candidates(mhip_ctx* ctx, const BeamCandDesc& d);
int mhip_launch_ancestry(mhip_ctx* ctx, const int* anc_old, int* anc_new, const int* parent, int rows, int ld, int step);
// Generator bookkeeping of one beam-search step on the device (TextRecognitionGenerator._generate, generator.py:208-362:
// finalize_hypos over the eos candidates among the first `beam`, the `beam` best live candidates become the next rows).
// One thread per crop; finished crops keep their rows (no batch compaction).  All arrays live in HBM.
struct BeamState {
  int bsz = 0, beam = 1, max_len = 0, pad = 1, eos = 2;
  const float* cand_scores = nullptr;   // [bsz][2*beam] (mhip_launch_beam_candidates)
  const int* cand_tokens = nullptr;
  const int* cand_beams = nullptr;
  int* tokens[2] = {nullptr, nullptr};  // [bsz*beam][max_len + 2] hypothesis tokens incl. the leading eos; read [cur], written [cur^1]
  int* last_tok = nullptr;              // [bsz*beam] token every row feeds to the next step
  int* parent = nullptr;                // [bsz*beam] row of the previous step a row continues
  float* cum = nullptr;                 // [bsz*beam] cumulative log-probability of a row
  unsigned char* ignore = nullptr;      // [bsz*beam] cands_to_ignore
  unsigned char* finished = nullptr;    // [bsz]
  int* fin_count = nullptr;             // [bsz] finalized hypotheses so far
  int* fin_tokens = nullptr;            // [bsz][beam][max_len + 1]
  int* fin_len = nullptr;               // [bsz][beam]
  float* fin_score = nullptr;           // [bsz][beam] normalised scores, in finalisation order
  int* remaining = nullptr;             // [1] crops not finished yet
};
size_t mhip_beam_state_bytes(int bsz, int beam, int max_len);
int mhip_beam_state_carve(void* base, int bsz, int beam, int max_len, int pad, int eos, BeamState* st);   // pointers into `base`
int mhip_launch_beam_init(mhip_ctx* ctx, const BeamState& st, int* anc0, int anc_ld);
int mhip_launch_beam_select(mhip_ctx* ctx, const BeamState& st, int cur, int step);
// best hypothesis per crop -> tokens_out [bsz][max_len + 1] (padded), lengths_out [bsz], scores_out [bsz] (device arrays)
int mhip_launch_beam_best(mhip_ctx* ctx, const BeamState& st, int* tokens_out, int* lengths_out, float* scores_out);
size_t mhip_pil_resize_fragments_scratch(const mhip_crop_desc* descs, int n, int dh, int dw, int filter);
int mhip_pil_resize_fragments(mhip_ctx* ctx, const uint8_t* base_dev, const mhip_crop_desc* descs, int n, uint8_t* dst, int dh,
                              int dw, int filter, void* scratch, size_t scratch_bytes);
