/*
 * marie_hip.h — C ABI of libmarie_hip.so, the MI355X (gfx950) OCR hot path.
 *
 * The reference (gregbugaj/marie-icr) has NO FFI on this path: its boundary is
 * three Python abstract classes.  This header is the C boundary
 * the Python mirrors of those classes (the marie_icr_amd Python package) bind with ctypes; each
 * entry point names the reference code it replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MHIP_E* code; nothing
 *     throws across the ABI; mhip_last_error(ctx) holds the message.
 *   - plain pointers and sizes only.  "_dev" pointers are device (HBM) addresses,
 *     "_host" pointers are host addresses; the caller owns both.
 *   - one ctx per (process, GPU); a ctx and its models are NOT thread-safe.
 *   - device entry points enqueue on the ctx stream (mhip_set_stream) and return
 *     without synchronising; *_host entry points synchronise before returning.
 */
#ifndef MARIE_HIP_H
#define MARIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHIP_OK 0
#define MHIP_EINVAL (-22)  /* bad argument / shape */
#define MHIP_ENOMEM (-12)  /* device allocation failed */
#define MHIP_EHIP (-5)     /* HIP runtime error (see mhip_last_error) */
#define MHIP_ESTATE (-1)   /* call order violated (e.g. forward before finalize) */

/* arithmetic type of the recognizer's conv/GEMM contractions */
#define MHIP_PREC_F16 0 /* f16 operands, fp32 MFMA accumulate (v_mfma_f32_16x16x32_f16)   */
#define MHIP_PREC_F32 1 /* exact fp32 operands and accumulate (v_mfma_f32_16x16x4_f32)    */

typedef struct mhip_ctx mhip_ctx;
typedef struct mhip_crnn mhip_crnn;

/* ---- context ------------------------------------------------------------------------ */
/* replaces: device selection in marie/models/utils.py:initialize_device_settings and the
 * per-processor `.to(device)` calls (marie/document/craft_ocr_processor.py:142-146).     */
int mhip_init(int device_id, mhip_ctx** out);
int mhip_destroy(mhip_ctx* ctx);
const char* mhip_last_error(mhip_ctx* ctx);
/* hipStream_t to enqueue on (NULL = the null stream).  The caller keeps it alive. */
int mhip_set_stream(mhip_ctx* ctx, void* hip_stream);
int mhip_synchronize(mhip_ctx* ctx);
/* "gfx950", CU count and HBM bytes of the bound device */
int mhip_device_info(mhip_ctx* ctx, char* arch, size_t arch_len, int* cu_count, size_t* hbm_bytes);
/* Device-to-device copy on the ctx stream (e.g. weight arena <-> an RCCL broadcast buffer). */
int mhip_memcpy_dev(mhip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);

/* Phase gate: orders two streams that are fed by two host threads.  replaces: nothing in the reference — its page loop is
 * serial (marie/ocr/ocr_engine.py:172-221: detect, then recognize, page by page); here the detector of page batch k + 1 runs
 * under the HBM-bound decode phase of the recognizer of batch k, and the gate is where that phase starts.
 *   mhip_gate_signal  records an event on ctx's stream and counts it (signal n);
 *   mhip_gate_wait    blocks the calling thread until signal `seq` exists (at most timeout_ms), then makes ctx's stream wait
 *                     for it: returns 1 (stream ordered), 0 (gate open or timed out: the caller proceeds unordered), < 0 error;
 *   mhip_gate_open    1: every present and future wait returns 0 at once (error / shutdown path), 0: closes it again.      */
typedef struct mhip_gate mhip_gate;
int mhip_gate_create(mhip_ctx* ctx, mhip_gate** out);
int mhip_gate_destroy(mhip_gate* g);
int mhip_gate_signal(mhip_gate* g, mhip_ctx* ctx);
long long mhip_gate_count(mhip_gate* g);
int mhip_gate_open(mhip_gate* g, int open);
int mhip_gate_wait(mhip_gate* g, mhip_ctx* ctx, long long seq, int timeout_ms);

/* Per-kernel timing with HIP events on the ctx stream (bench.py's roofline leg).
 * replaces: marie/logging_core/profile.py TimeContextCuda around model calls.
 * While enabled every kernel launch is bracketed by an event pair; mhip_profile_read
 * synchronises and returns the accumulated device time and launch count of one kernel id. */
int mhip_profile_enable(mhip_ctx* ctx, int enable);
int mhip_profile_reset(mhip_ctx* ctx);
int mhip_profile_read(mhip_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);
/* Algorithmic FLOPs (2*MAC) of the launches of an MFMA kernel id timed since the last reset (0 for HBM-bound ids). */
int mhip_profile_flops(mhip_ctx* ctx, int kernel_id, double* flops);
int mhip_kernel_count(void);
const char* mhip_kernel_name(int kernel_id);

/* ---- NHWC convolution / GEMM primitive -------------------------------------------------- */
/* replaces: every nn.Conv2d (+ folded BatchNorm2d + ReLU + MaxPool2d) and nn.Linear on the path,
 * e.g. marie/models/icr/modules/feature_extraction.py:13-25, marie/models/craft/craft.py:14-28,
 * marie/models/craft/basenet/vgg16_bn.py:23-74.
 *   out[b][yp][xp][n] = pool( act( scale[n] * sum_{dy,dx,c} in[b][y+dy-pad][x+dx-pad][c] * w[n][dy][dx][c] + bias[n] ) )
 * in/w/out element type = precision (f16 or f32); out is fp32 when out_f32 != 0.  Stride 1, zero padding.
 * Cin must be a multiple of 64 (f16) / 32 (f32).  pool: 0 none, 1 = 2x2/2 (floor), 2 = (2,1)/(2,1). */
typedef struct mhip_conv_desc {
  int32_t B, H, W, Cin; /* input  [B][H][W][Cin] */
  int32_t KH, KW, pad;  /* filter [N][KH][KW][Cin] */
  int32_t N;            /* output channels */
  int32_t pool, relu, out_f32;
  int32_t dil;          /* filter dilation, 0/1 = dense */
  int32_t Cin1;         /* with in2_dev: channels [0,Cin1) come from in_dev, [Cin1,Cin) from in2_dev */
  int32_t ldc;          /* output row pitch in elements, 0 = N (rows must stay 16-byte aligned; unpooled outputs only)        */
  int32_t pad_cols_writable; /* 1: columns [N, roundup(N, 8)) of a pitched row are the call's own and may be zeroed          */
} mhip_conv_desc;
/* in2_dev (may be NULL) is the second tensor of a channel-concatenated input, torch.cat([a, b], dim=1) followed
 * by a 1x1 conv (marie/models/craft/craft.py:63-64,67-69): the concatenation is never materialised.      */
int mhip_conv2d_nhwc(mhip_ctx* ctx, int precision, const mhip_conv_desc* d, const void* in_dev,
                     const void* in2_dev, const void* w_dev, const float* scale_dev, const float* bias_dev,
                     void* out_dev);

/* ---- crop batcher --------------------------------------------------------------------------- */
/* One text fragment inside a device buffer: first pixel at base + src_offset, h rows of w pixels, `channels` = 3
 * (BGR, OpenCV order) or 1 (gray), rows row_stride bytes apart (a fragment may be a window of a whole page). */
typedef struct mhip_crop_desc {
  uint64_t src_offset;
  int32_t h, w, row_stride, channels;
} mhip_crop_desc;
/* replaces: MemoryDataset.__getitem__ (BGR -> RGB -> PIL "L", marie/models/icr/memory_dataset.py:40-55) and
 * AlignCollate/NormalizePAD with keep_ratio_with_pad (marie/models/icr/dataset.py:275-324): every fragment becomes a
 * 32 x img_w uint8 line — Pillow-exact bicubic resize to height 32 keeping the aspect ratio (width capped at img_w),
 * last column replicated.  out_dev: uint8 [n][32][img_w].  Scratch comes from the ctx workspace.              */
int mhip_crop_batch(mhip_ctx* ctx, const uint8_t* base_dev, const mhip_crop_desc* descs_host, int n, int img_w,
                    uint8_t* out_dev);

/* ---- CRNN-family recognizer: None-VGG-BiLSTM-CTC ---------------------------------- */
/* replaces: Model(opt) construction, marie/models/icr/model.py:27-68 (Trans=None, Feat=VGG,
 * Seq=BiLSTM, Pred=CTC; imgH=32, input_channel=1, output_channel=512, hidden_size=256).  */
int mhip_crnn_create(mhip_ctx* ctx, int precision, int num_class, mhip_crnn** out);
int mhip_crnn_destroy(mhip_crnn* m);

/* replaces: model.load_state_dict(torch.load(...)), marie/document/craft_ocr_processor.py:146.
 * `key` is the reference state_dict key ("FeatureExtraction.ConvNet.0.weight", ... an optional
 * "module." prefix is ignored); `data` is host fp32 in the checkpoint's own layout
 * (conv: [Cout][Cin][kh][kw]; LSTM/Linear: [out][in]).  Unknown keys return MHIP_EINVAL,
 * "num_batches_tracked" is accepted and ignored.                                          */
int mhip_crnn_set_tensor(mhip_crnn* m, const char* key, const float* data_host, const int64_t* shape, int ndim);
/* Repack every tensor into the kernels' layouts (NHWC taps, BN folded to scale/shift, LSTM
 * gates in MFMA-fragment order) inside ONE device arena and upload it.  Fails with MHIP_ESTATE
 * if any tensor is missing.                                                                */
int mhip_crnn_finalize(mhip_crnn* m);
/* Allocate the (identically laid out) arena without filling it — for ranks that receive the
 * weights by an RCCL broadcast of the arena instead of reading a checkpoint.              */
int mhip_crnn_alloc_arena(mhip_crnn* m);
int mhip_crnn_arena(mhip_crnn* m, void** arena_dev, size_t* bytes);

/* T (time steps) for a crop of width w: w/4 - 1 (VGG stack, imgH = 32). */
int mhip_crnn_seq_len(int w);

/* Forward + greedy CTC decode of n pre-cropped grayscale lines, all 32 x w (w % 4 == 0, w >= 8).
 * replaces: AlignCollate's ToTensor/normalise for a full-width crop
 * (marie/models/icr/dataset.py:275-283), model(image, text_for_pred)
 * (marie/document/craft_ocr_processor.py:236-237), preds.max(2) (:240),
 * CTCLabelConverter.decode (marie/models/icr/utils.py:41-54) and the confidence
 * softmax/max/cumprod (:255-271).
 *   crops_dev   uint8  [n][32][w]
 *   logits_dev  fp32   [n][T][num_class]          or NULL
 *   argmax_dev  int32  [n][T]      raw per-step argmax (first max on ties)
 *   tokens_dev  int32  [n][T]      collapsed sequence (blank and repeats removed), 0-padded
 *   lengths_dev int32  [n]         number of valid tokens per line
 *   conf_dev    fp32   [n]         product over all T steps of the max softmax probability */
int mhip_crnn_forward(mhip_crnn* m, const uint8_t* crops_dev, int n, int w, float* logits_dev,
                      int32_t* argmax_dev, int32_t* tokens_dev, int32_t* lengths_dev, float* conf_dev);
/* Same with host buffers: H2D of the crops, forward, D2H of the outputs, stream sync. */
int mhip_crnn_forward_host(mhip_crnn* m, const uint8_t* crops_host, int n, int w, float* logits_host,
                           int32_t* argmax_host, int32_t* tokens_host, int32_t* lengths_host,
                           float* conf_host);
/* Crop batcher + forward + decode in one call: fragments described inside base_dev (e.g. word boxes on a page
 * that is already in HBM) -> host outputs.  replaces: CraftOcrProcessor.recognize_from_fragments' DataLoader +
 * model + decode loop (marie/document/craft_ocr_processor.py:184-286).                                        */
int mhip_crnn_forward_crops(mhip_crnn* m, const uint8_t* base_dev, const mhip_crop_desc* descs_host, int n, int img_w,
                            float* logits_host, int32_t* argmax_host, int32_t* tokens_host, int32_t* lengths_host,
                            float* conf_host);
/* Same with the fragments packed back to back in a HOST buffer (src_offset relative to packed_host). */
int mhip_crnn_forward_fragments_host(mhip_crnn* m, const uint8_t* packed_host, size_t packed_bytes,
                                     const mhip_crop_desc* descs_host, int n, int img_w, float* logits_host,
                                     int32_t* argmax_host, int32_t* tokens_host, int32_t* lengths_host,
                                     float* conf_host);
/* Bytes of ctx workspace one forward of n lines of width w needs (activations, gate buffers). */
size_t mhip_crnn_workspace_bytes(mhip_crnn* m, int n, int w);
/* Algorithmic FLOPs (2*MAC) of one forward of n lines of width w, per kernel id — what
 * bench.py divides by the measured kernel time for the roofline line.                    */
double mhip_crnn_kernel_flops(mhip_crnn* m, int kernel_id, int n, int w);

/* ---- production ICR recognizer: TPS-ResNet-BiLSTM-Attn ---------------------------------------- */
typedef struct mhip_icr mhip_icr;
/* replaces: Model(opt) with Transformation="TPS", FeatureExtraction="ResNet", SequenceModeling="BiLSTM",
 * Prediction="Attn" as configured by CraftOcrProcessor (marie/document/craft_ocr_processor.py:49-70,103-146): imgH 32,
 * imgW 100, 20 fiducial points, hidden 256, batch_max_length 48 (49 decode steps), num_class = 94 chars + [GO] + [s].
 * Keys are the reference state_dict names ("Transformation.LocalizationNetwork.conv.0.weight", ...,
 * "Transformation.GridGenerator.P_hat", "FeatureExtraction.ConvNet.layer3.4.bn2.running_var", ...,
 * "Prediction.attention_cell.rnn.weight_ih", "Prediction.generator.bias"; a "module." prefix is ignored).          */
int mhip_icr_create(mhip_ctx* ctx, int precision, int num_class, mhip_icr** out);
int mhip_icr_destroy(mhip_icr* m);
int mhip_icr_set_tensor(mhip_icr* m, const char* key, const float* data_host, const int64_t* shape, int ndim);
int mhip_icr_finalize(mhip_icr* m);
int mhip_icr_alloc_arena(mhip_icr* m);
int mhip_icr_arena(mhip_icr* m, void** arena_dev, size_t* bytes);
int mhip_icr_steps(void); /* 49 */
/* replaces: model(image, text_for_pred, is_train=False) (craft_ocr_processor.py:245): TPS rectification
 * (marie/models/icr/modules/transformation.py:32-42), ResNet-45 (feature_extraction.py:212-246), BiLSTM x2, and the
 * 49-step greedy attention decoder (prediction.py:47-61,72-83).  crops_dev uint8 [n][32][100] (already resized and
 * padded, e.g. by mhip_crop_batch with img_w = 100).  Outputs: logits fp32 [n][49][num_class]; argmax int32 [n][49];
 * pmax fp32 [n][49] = softmax probability of the arg-max (for the confidence product); rectified fp32 [n][32][100] or
 * NULL.  The [s]-cut / confidence rule (craft_ocr_processor.py:259-271) is the caller's, as in the reference.       */
int mhip_icr_forward(mhip_icr* m, const uint8_t* crops_dev, int n, float* logits_dev, int32_t* argmax_dev,
                     float* pmax_dev, float* rectified_dev);
int mhip_icr_forward_host(mhip_icr* m, const uint8_t* crops_host, int n, float* logits_host, int32_t* argmax_host,
                          float* pmax_host, float* rectified_host);

/* ---- CRAFT text detector -------------------------------------------------------------------- */
typedef struct mhip_craft mhip_craft;
/* replaces: CRAFT() + load_state_dict(copyStateDict(torch.load(...))), marie/boxes/craft_box_processor.py:260-285.
 * Keys are CRAFT.state_dict() names ("basenet.slice1.0.weight", "upconv1.conv.0.weight", "conv_cls.8.bias", ...;
 * a "module." prefix is ignored).                                                                           */
int mhip_craft_create(mhip_ctx* ctx, int precision, mhip_craft** out);
int mhip_craft_destroy(mhip_craft* m);
int mhip_craft_set_tensor(mhip_craft* m, const char* key, const float* data_host, const int64_t* shape, int ndim);
int mhip_craft_finalize(mhip_craft* m);
int mhip_craft_alloc_arena(mhip_craft* m);
int mhip_craft_arena(mhip_craft* m, void** arena_dev, size_t* bytes);
/* replaces: resize_aspect_ratio's size arithmetic, marie/models/craft/imgproc.py:45-71.
 * (h, w) page -> ratio, resized (th, tw), /32 canvas (H32, W32); the score maps are H32/2 x W32/2.          */
int mhip_craft_geometry(int h, int w, int canvas_size, double mag_ratio, double* ratio, int* th, int* tw,
                        int* H32, int* W32);
size_t mhip_craft_workspace_bytes(mhip_craft* m, int h, int w, int canvas_size, double mag_ratio);
/* Algorithmic FLOPs (2*MAC over the checkpoint's real channel counts) of one forward of an h x w page, per kernel id. */
double mhip_craft_kernel_flops(mhip_craft* m, int kernel_id, int h, int w, int canvas_size, double mag_ratio);
/* replaces: get_prediction's resize + normalizeMeanVariance + net(x), craft_box_processor.py:94-110.
 * page_dev uint8 [h][w][3] (channel order as given, the reference feeds BGR) -> scores_dev fp32 [H32/2][W32/2][2]
 * (channel 0 = text/region score, 1 = link/affinity score).                                                  */
int mhip_craft_forward(mhip_craft* m, const uint8_t* page_dev, int h, int w, int canvas_size, double mag_ratio,
                       float* scores_dev);
/* replaces: get_prediction incl. getDetBoxes_core (marie/models/craft/craft_utils.py:25-98): forward, threshold,
 * 4-connected components with statistics (GPU), then per component dilate + minAreaRect + diamond fix + clockwise
 * order.  boxes_host gets n_boxes x 4 corners x (x, y) fp32 in SCORE-MAP coordinates, in OpenCV label order
 * (adjustResultCoordinates is the caller's, as in the reference).  scores_host (may be NULL) receives the maps. */
int mhip_craft_detect(mhip_craft* m, const uint8_t* page_dev, int h, int w, int canvas_size, double mag_ratio,
                      float text_threshold, float link_threshold, float low_text, float* boxes_host,
                      int max_boxes, int* n_boxes, float* scores_host, double* ratio_out);
int mhip_craft_detect_host(mhip_craft* m, const uint8_t* page_host, int h, int w, int canvas_size,
                           double mag_ratio, float text_threshold, float link_threshold, float low_text,
                           float* boxes_host, int max_boxes, int* n_boxes, float* scores_host, double* ratio_out);

/* ---- Pillow-exact 8-bit resize (antialiased BILINEAR / BICUBIC) of RGB images ------------------------------------------ */
#define MHIP_PIL_BILINEAR 2   /* PIL.Image.BILINEAR */
#define MHIP_PIL_BICUBIC 3    /* PIL.Image.BICUBIC  */
/* replaces: ResizeShortestEdge/ResizeTransform's PIL resize (marie/detectron/detector.py:103-105) and TrOCR's
 * im.resize((384, 384), BICUBIC) (marie/document/trocr_ocr_processor.py:116-118).  Host u8 [sh][sw][3] -> [dh][dw][3]. */
int mhip_pil_resize_rgb_host(mhip_ctx* ctx, const uint8_t* src_host, int sh, int sw, uint8_t* dst_host, int dh, int dw,
                             int filter);

/* ---- ViT encoder: the DiT detector backbone (BEiT + fpn1..4) and the TrOCR image encoder ------------------------- */
/* replaces: BEiT.forward_features, marie/boxes/dit/ditod/beit.py:706-748 (dit_base_patch16 :787-800, dit_large_patch16
 * :803-816), and AdaptedVisionTransformer.forward_features, marie/models/unilm/trocr/deit.py:105-146.            */
typedef struct mhip_vit mhip_vit;
typedef struct mhip_vit_config {
  int dim, depth, heads;   /* 768/12/12 base, 1024/24/16 large; head dim must be 64                                     */
  int patch;               /* 16                                                                                    */
  int pos_h, pos_w;        /* pre-training position grid: 14 x 14 (DiT, IMG_SIZE 224), 24 x 24 (TrOCR, 384)          */
  int layer_scale;         /* blocks carry gamma_1 / gamma_2 (DiT: init_values 0.1 / 1e-5)                          */
  int qkv_bias;            /* 0 none (TrOCR DeiT), 1 BEiT q_bias + v_bias, 2 a full qkv.bias                        */
  int final_norm;          /* apply norm.{weight,bias} to the output tokens (TrOCR encoder)                         */
  int fpn;                 /* 1: fpn1..fpn4 on the outputs of blocks taps[0..3] (DiT: 3,5,7,11 / 7,11,15,23)        */
  int taps[4];
  float ln_eps;            /* 1e-6                                                                                  */
} mhip_vit_config;
int mhip_vit_create(mhip_ctx* ctx, int precision, const mhip_vit_config* cfg, mhip_vit** out);
int mhip_vit_destroy(mhip_vit* m);
/* keys relative to the BEiT / VisionTransformer module: cls_token, pos_embed, patch_embed.proj.*, blocks.N.*, norm.*, fpnK.* */
int mhip_vit_set_tensor(mhip_vit* m, const char* key, const float* data, const int64_t* shape, int ndim);
int mhip_vit_finalize(mhip_vit* m);
int mhip_vit_alloc_arena(mhip_vit* m);
int mhip_vit_arena(mhip_vit* m, void** arena_dev, size_t* bytes);
/* B host images u8 [th][tw][3], each placed top-left on a zero (post-normalisation) H32 x W32 canvas; pixel -> (p-127.5)/127.5,
 * swap_rb = 1 reverses the stored channel order.  Any output may be NULL.  tokens_out fp32 [B][1 + np][dim];
 * fpnK fp32 NHWC [B][h_K][w_K][dim] at strides 4, 8, 16, 32.                                                          */
int mhip_vit_forward_host(mhip_vit* m, const uint8_t* imgs_host, int B, int th, int tw, int H32, int W32, int swap_rb,
                          float* tokens_out, float* fpn0, float* fpn1, float* fpn2, float* fpn3);

/* ---- DiT Mask R-CNN text detector (BoxProcessorUlimDit's model) ------------------------------------------------------- */
/* replaces: OptimizedDetectronPredictor.invoke_model, marie/detectron/detector.py:83-147 (model built by
 * build_vit_fpn_backbone, marie/boxes/dit/ditod/backbone.py:131-153; config/zoo/unilm/dit/text_detection/ YAMLs).       */
typedef struct mhip_dit mhip_dit;
typedef struct mhip_dit_config {
  int model;                 /* 0 dit_base_patch16 (mask_rcnn_dit_base.yaml), 1 dit_large_patch16 (mask_rcnn_dit_prod.yaml) */
  int min_size_test;         /* INPUT.MIN_SIZE_TEST 800                                                             */
  int max_size_test;         /* INPUT.MAX_SIZE_TEST 1333 (default) / 4000 (prod)                                    */
  int detections_per_image;  /* TEST.DETECTIONS_PER_IMAGE 2000 / 2500                                               */
  float anchor_sizes[5];     /* ANCHOR_GENERATOR.SIZES 4, 8, 16, 32, 64                                             */
  float aspect_ratios[3];    /* ANCHOR_GENERATOR.ASPECT_RATIOS 1.5, 3.5, 6.5                                        */
  float rpn_nms_thresh;      /* RPN.NMS_THRESH 0.7                                                                  */
  float score_thresh;        /* ROI_HEADS.SCORE_THRESH_TEST 0.05                                                    */
  float nms_thresh;          /* ROI_HEADS.NMS_THRESH_TEST 0.5                                                       */
} mhip_dit_config;
int mhip_dit_default_config(int model, mhip_dit_config* cfg);
/* ResizeShortestEdge output (nh, nw) and the /32 canvas for an h x w page */
int mhip_dit_resized_shape(const mhip_dit_config* cfg, int h, int w, int* nh, int* nw, int* H32, int* W32);
int mhip_dit_create(mhip_ctx* ctx, int precision, const mhip_dit_config* cfg, mhip_dit** out);
int mhip_dit_destroy(mhip_dit* m);
/* detectron2 checkpoint keys ("backbone.bottom_up.backbone.*", "backbone.fpn_*", "proposal_generator.rpn_head.*",
 * "roi_heads.box_head.*", "roi_heads.box_predictor.*"; mask-head keys are accepted and ignored)                          */
int mhip_dit_set_tensor(mhip_dit* m, const char* key, const float* data, const int64_t* shape, int ndim);
int mhip_dit_finalize(mhip_dit* m);
int mhip_dit_alloc_arena(mhip_dit* m);
int mhip_dit_arena(mhip_dit* m, int which /* 0 backbone, 1 heads */, void** arena_dev, size_t* bytes);
size_t mhip_dit_workspace_bytes(mhip_dit* m, int B, int h, int w);
/* B device pages u8 BGR [h][w][3] of one size -> per page up to 1000 boxes xyxy fp32 in page coordinates, score-ordered.
 * boxes_host [B][1000][4], scores_host [B][1000] (may be NULL), counts_host [B].                                      */
int mhip_dit_detect(mhip_dit* m, const uint8_t* const* pages_dev, int B, int h, int w, float* boxes_host,
                    float* scores_host, int* counts_host);
int mhip_dit_detect_host(mhip_dit* m, const uint8_t* pages_host, int B, int h, int w, float* boxes_host,
                         float* scores_host, int* counts_host);
/* one page plus the intermediates the parity tests compare: FPN maps p2..p6 fp32 NHWC and the RPN proposals (any NULL) */
int mhip_dit_debug_host(mhip_dit* m, const uint8_t* page_host, int h, int w, float* boxes_host, float* scores_host,
                        int* count_host, float* p2, float* p3, float* p4, float* p5, float* p6, float* prop_boxes,
                        float* prop_scores, int* prop_count);
/* same plus the inputs of the two discrete stages as this run computed them: fpn5[l] as above, rpn_head5[l] fp32
 * [H*W][16] for p2..p6, box_head fp32 [1000][8] (rows past *prop_count undefined); any pointer may be NULL             */
int mhip_dit_debug_taps_host(mhip_dit* m, const uint8_t* page_host, int h, int w, float* boxes_host, float* scores_host,
                             int* count_host, float* const* fpn5, float* const* rpn_head5, float* prop_boxes,
                             float* prop_scores, int* prop_count, float* box_head);
/* The detectron2 stages of the detector on caller-supplied host inputs (each also used by the parity tests):
 * RPN.predict_proposals for one image — heads_host[l] fp32 [H[l]*W[l]][16] (3 objectness logits, 3 x 4 deltas, 1 pad) for
 * p2..p6 -> up to 1000 proposals xyxy + logits, score-ordered;                                                          */
int mhip_rpn_proposals_host(mhip_ctx* ctx, const float* const* heads_host, const int* H, const int* W, const int* strides,
                            const float* anchor_sizes, const float* aspect_ratios, int img_h, int img_w, float nms_thresh,
                            float* boxes_out, float* scores_out, int* count_out);
/* ROIPooler (7x7 ROIAlignV2, sampling_ratio 0, levels by box size) — feats_host[l] fp32 NHWC at strides 4..32,
 * rois [n][4] (n <= 1000) -> pooled fp32 [n][49*C] with k = bin*C + c;                                                  */
int mhip_roi_align_host(mhip_ctx* ctx, const float* const* feats_host, const int* H, const int* W, int C,
                        const float* rois_host, int n, float* pooled_out);
/* FastRCNNOutputLayers.inference + detector_postprocess — head_host [n][8] (2 class scores, 4 deltas, 2 pad).           */
int mhip_det_final_host(mhip_ctx* ctx, const float* head_host, const float* rois_host, int n, int img_h, int img_w,
                        int page_h, int page_w, float score_thresh, float nms_thresh, int max_det, float* boxes_out,
                        float* scores_out, int* count_out);
/* ---- overlay cleaner: pix2pixHD LocalEnhancer generator + blend -------------------------------- */
/* replaces: OverlayProcessor.__extract_segmentation_mask / model.test() (marie/overlay/overlay.py:165-189;
 * marie/models/pix2pix/models/networks_hd.py:24-213, netG "local", instance norm, spectral-normed convolutions) — page in HBM
 * -> generator -> image in HBM, no PNG round trip.  ngf: base width (64 in the reference; f16 needs a multiple of 64, f32 32). */
typedef struct mhip_overlay mhip_overlay;
int mhip_overlay_create(mhip_ctx* ctx, int precision, int ngf, mhip_overlay** out);
int mhip_overlay_destroy(mhip_overlay* m);
/* state_dict keys of the reference's netG ("model.1.weight_orig" / ".weight_u" / ".weight_v" / ".bias", "model1_2.3.*",
 * "downsample.weight" ...; an optional "netG." / "module." prefix is ignored); data fp32 in the checkpoint's layout            */
int mhip_overlay_set_tensor(mhip_overlay* m, const char* key, const float* data, const int64_t* shape, int ndim);
int mhip_overlay_finalize(mhip_overlay* m);
/* OverlayProcessor.preprocess (overlay.py:147-163): the white canvas the page is placed on (both sides to the next multiple of
 * 32 when either is ragged)                                                                                                    */
int mhip_overlay_padded_shape(int h, int w, int* H, int* W);
/* page u8 BGR [h][w][3] (device) -> the generator's image u8 RGB [H][W][3] on the padded canvas = tensor2im(fake)             */
int mhip_overlay_forward(mhip_overlay* m, const uint8_t* page_dev, int h, int w, uint8_t* fake_rgb_dev);
int mhip_overlay_forward_host(mhip_overlay* m, const uint8_t* page_host, int h, int w, uint8_t* fake_rgb_host, float* raw_host);
/* replaces: OverlayProcessor.blend_to_text (overlay.py:247-291): real BGR + generator image -> text-only image (BGR)          */
int mhip_overlay_blend(mhip_ctx* ctx, const uint8_t* real_bgr_dev, const uint8_t* mask_dev, uint8_t* out_dev, size_t pixels);
/* replaces: blackout_bboxes, marie/boxes/dit/ulim_dit_box_processor.py:161-198, in place on a device page (BGR).        */
int mhip_blackout_bboxes(mhip_ctx* ctx, uint8_t* page_dev, int h, int w, const int32_t* boxes_xyxy_host, int n,
                         int* changed);
/* replaces: the OpenCV chain of crop_to_content (marie/utils/image_utils.py:190-252, behind OcrEngine.extract(crop_to_content=True),
 * marie/ocr/ocr_engine.py:169-176) and of crop_to_content_box (marie/boxes/dit/ulim_dit_box_processor.py:291-352, behind
 * psm_sparse(bbox_optimization=True), :608-626): BGR2GRAY, then content_aware ? close_2x3(otsu(divide(gray, blur5x5(gray), 255)))
 * : otsu(gray).  For each of n rectangles (x, y, w, h, inside the h x w BGR device page) ext[i] = {xmin, ymin, xmax, ymax, count}
 * of the pixels that come out 0, in rectangle coordinates (count == 0: no such pixel, the rest is 0).  The callers' padding
 * rules stay on the host (marie_icr_amd/content.py).                                                                           */
int mhip_content_extents(mhip_ctx* ctx, const uint8_t* page_dev, int h, int w, const int32_t* rects_xywh_host, int n,
                         int content_aware, int32_t* ext_host);

/* ---- TrOCR recognizer (image encoder + text decoder + beam search) ------------------------------------------------------ */
/* replaces: TrOcrProcessor's model path, marie/document/trocr_ocr_processor.py:116-180 (preprocess_image, get_text), with
 * TrOCREncoder (marie/models/unilm/trocr/trocr_models.py:508-524), fairseq's TransformerDecoder (built :137-147) and
 * TextRecognitionGenerator._generate (marie/models/unilm/trocr/generator.py:11-374).                                      */
typedef struct mhip_trocr mhip_trocr;
typedef struct mhip_trocr_config {
  int enc_dim, enc_depth, enc_heads;            /* trocr_base: beit_base_patch16_384 = 768 / 12 / 12 (trocr_models.py:423-434) */
  int dec_dim, dec_layers, dec_heads, dec_ffn;  /* 1024 / 12 / 16 / 4096                                                      */
  int vocab;                                    /* len(target dictionary)                                                     */
  int max_positions;                            /* decoder max_positions (512): embed_positions has max_positions + pad + 1 rows */
  int beam;                                     /* 3 (trocr_ocr_processor.py:228)                                             */
  int max_len_b;                                /* generation max_len_b (200); max_len = min(max_len_b, max_positions - 1)     */
  int min_len;                                  /* 1                                                                          */
  int pad, eos;                                 /* fairseq dictionary: 1, 2                                                   */
  float embed_scale;                            /* 1.0 with RoBERTa's no_scale_embedding, else sqrt(dec_dim)                  */
  int img_size;                                 /* 384                                                                        */
} mhip_trocr_config;
int mhip_trocr_default_config(int model /* 0 base, 1 large */, mhip_trocr_config* cfg);
int mhip_trocr_max_len(const mhip_trocr_config* cfg);
int mhip_trocr_create(mhip_ctx* ctx, int precision, const mhip_trocr_config* cfg, mhip_trocr** out);
int mhip_trocr_destroy(mhip_trocr* m);
/* fairseq checkpoint keys: "encoder.deit.*" (timm VisionTransformer) and "decoder.*" (TransformerDecoder)                  */
int mhip_trocr_set_tensor(mhip_trocr* m, const char* key, const float* data, const int64_t* shape, int ndim);
int mhip_trocr_finalize(mhip_trocr* m);
int mhip_trocr_alloc_arena(mhip_trocr* m);
int mhip_trocr_arena(mhip_trocr* m, int which /* 0 encoder, 1 decoder */, void** arena_dev, size_t* bytes);
size_t mhip_trocr_workspace_bytes(mhip_trocr* m, int n);
/* `gate` (may be NULL) is signalled inside every generate call at the point of the stream where the image encoder ends and
 * the autoregressive decode begins (TextRecognitionGenerator._generate's step loop, generator.py:182-362).  The caller owns it. */
int mhip_trocr_set_decode_gate(mhip_trocr* m, mhip_gate* gate);
/* n device crops u8 [img][img][3] -> best hypothesis per crop: tokens_out [n][max_len + 1] (eos included, pad-filled),
 * lengths_out [n], scores_out [n] = length-normalised log-probability (the reference reports exp(score)).               */
int mhip_trocr_generate(mhip_trocr* m, const uint8_t* crops_dev, int n, int swap_rb, int32_t* tokens_out,
                        int32_t* lengths_out, float* scores_out);
/* host crops; optional parity taps: enc_tokens_out fp32 [n][577][enc_dim], step0_logits_out fp32 [n][vocab]               */
int mhip_trocr_generate_host(mhip_trocr* m, const uint8_t* crops_host, int n, int swap_rb, int32_t* tokens_out,
                             int32_t* lengths_out, float* scores_out, float* enc_tokens_out, float* step0_logits_out);
/* The recognizer in two halves, for callers whose fragments arrive in batches (OcrEngine's batched path: page batches leave the
 * detector one after the other, where the reference's loop calls its recognizer page by page, marie/ocr/ocr_engine.py:172-199):
 * the image encoder (marie/models/unilm/trocr/deit.py:105-146) runs per batch of fragments, the beam search
 * (generator.py:127-362) once over everything encoded since encode_begin — results as mhip_trocr_generate_fragments on the
 * concatenated batches.  encode_begin(max_crops) reserves the token store (grow-only; more than max_crops may follow).       */
int mhip_trocr_encode_begin(mhip_trocr* m, int max_crops);
int mhip_trocr_encode_fragments(mhip_trocr* m, const uint8_t* base_dev, const mhip_crop_desc* descs_host, int n, int swap_rb);
int mhip_trocr_encoded(mhip_trocr* m);      /* crops encoded since encode_begin                                               */
int mhip_trocr_decode(mhip_trocr* m, int32_t* tokens_out, int32_t* lengths_out, float* scores_out);
/* fragments of any size (3 channels) inside one device buffer -> Pillow bicubic to img x img -> generate                    */
int mhip_trocr_generate_fragments(mhip_trocr* m, const uint8_t* base_dev, const mhip_crop_desc* descs_host, int n,
                                  int swap_rb, int32_t* tokens_out, int32_t* lengths_out, float* scores_out);
/* mhip_trocr_generate_host plus the candidate list of every beam-search step as the generator saw it (BeamSearch.step's
 * top 2*beam of cumulative score, generator.py:208-223): trace_* are host arrays [max_len + 1][n][2 * beam] (scores, token ids,
 * source beams), *steps_out the number of steps that ran.  For parity tests: the oracle's search is walked beside it. */
int mhip_trocr_generate_trace_host(mhip_trocr* m, const uint8_t* crops_host, int n, int swap_rb, int32_t* tokens_out,
                                   int32_t* lengths_out, float* scores_out, float* trace_scores, int32_t* trace_tokens,
                                   int32_t* trace_beams, int* steps_out);
/* The decoder's encoder-attention stage alone (one layer, one step) on host inputs, through the f16 kernels that attend over
 * the encoder tokens themselves (key / value projections absorbed: cross_attn.hip).  replaces: fairseq MultiheadAttention
 * (encoder_attn) as TextRecognitionGenerator drives it, marie/models/unilm/trocr/generator.py:127-362.  q [crops*beam][heads*64]
 * (projected, scaled), enc [crops][n_tok][enc_dim], wk / wv [heads*64][enc_dim], bv [heads*64] -> out [crops*beam][heads*64]. */
int mhip_cross_attention_host(mhip_ctx* ctx, const float* q, const float* enc, const float* wk, const float* wv,
                              const float* bv, int crops, int beam, int heads, int n_tok, int enc_dim, float* out);

/* ---- word-box / line geometry of the DiT box processor (host, pure functions; no ctx) --------------------------------- */
/* replaces: merge_boxes, marie/utils/overlap.py:268-330 (find_overlap_horizontal(center_y_overlap=0.5) :106-183,
 * merge_bboxes_as_block :186-204).  xyxy fp32 [n][4] -> out_xyxy fp32 (capacity n rows), *n_out rows.       */
int mhip_merge_boxes(const float* xyxy, int n, float* out_xyxy, int* n_out);
/* replaces: line_merge, marie/boxes/line_processor.py:105-171.  xywh int32 [n][4] -> merged lines sorted by y
 * (capacity n rows).  Equal-y boxes keep input order (the reference leaves ties to numpy's unstable sort).   */
int mhip_line_merge(const int32_t* xywh, int n, int32_t* out_xywh, int* n_out);
/* replaces: find_line_number per box, marie/boxes/line_processor.py:15-44.  out[i] = 1-based line, -1 if no lines. */
int mhip_find_line_numbers(const int32_t* lines_xywh, int n_lines, const int32_t* boxes_xywh, int n, int32_t* out);
/* replaces: lines_from_bboxes, marie/boxes/dit/ulim_dit_box_processor.py:201-288 (rectangle mask, horizontal
 * erode+dilate, 4-connected components with stats, size filter, line_merge) for a height x width page.
 * Returns MHIP_ENOMEM with *n_out = needed rows when cap is too small.                                       */
int mhip_lines_from_bboxes(const float* xyxy, int n, int height, int width, int32_t* out_xywh, int cap, int* n_out);

/* ---- page ingest (the step before the detector) ------------------------------------------------------------------------ */
/* replaces: the shape rule of ensure_max_page_size, marie/utils/image_utils.py:275-310, for one frame (orientation-aware
 * maximum, expanded by expand_ratio; aspect-preserving, int() truncation).  Returns 1 and the new size when the frame is
 * too large, 0 (and the old size) when it is kept.  Host only, no ctx.                                                     */
int mhip_max_page_size(int width, int height, int max_w_portrait, int max_h_portrait, double expand_ratio, int* new_w,
                       int* new_h);
/* replaces: cv2.resize(frame, (new_width, new_height), interpolation=cv2.INTER_AREA), image_utils.py:313-315 — 8-bit,
 * 1 or 3 channels, shrinking only (dh <= sh, dw <= sw).  Device buffers; src rows src_pitch bytes apart, dst packed.        */
int mhip_resize_area_u8(mhip_ctx* ctx, const uint8_t* src_dev, int sh, int sw, int cn, size_t src_pitch, uint8_t* dst_dev,
                        int dh, int dw);
/* replaces: cv2.resize(image, (new_w, new_h), interpolation=cv2.INTER_CUBIC) in resize_image's shrink branch,
 * marie/utils/resize_image.py:53-61 (small pages / region crops framed for the DiT detector,
 * marie/boxes/dit/ulim_dit_box_processor.py:524-540).  8-bit, 1 or 3 channels, any scale.                                  */
int mhip_resize_cubic_u8(mhip_ctx* ctx, const uint8_t* src_dev, int sh, int sw, int cn, size_t src_pitch,
                         uint8_t* dst_dev, int dh, int dw);
int mhip_resize_cubic_u8_host(mhip_ctx* ctx, const uint8_t* src_host, int sh, int sw, int cn, uint8_t* dst_host, int dh,
                              int dw);
/* the same on packed host buffers (stages through the context workspace). */
int mhip_resize_area_u8_host(mhip_ctx* ctx, const uint8_t* src_host, int sh, int sw, int cn, uint8_t* dst_host, int dh,
                             int dw);

#ifdef __cplusplus
}
#endif
#endif /* MARIE_HIP_H */
