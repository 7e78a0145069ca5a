// vit_internal.h — the ViT encoder object as the composite models (DiT detector, TrOCR recognizer) see it.
#pragma once
#include "weights_util.h"

struct mhip_vit {
  mhip_ctx* ctx = nullptr;
  int precision = MHIP_PREC_F16;
  mhip_vit_config cfg{};
  TensorStore store;
  Arena arena;
  // position tables resized to the patch grids seen so far (mixed-DPI streams alternate between a few page sizes);
  // built once per geometry, never inside a steady-state forward
  // dev16: the same table in f16 (f16 residual stream) — or, when the stream is split, its high plane followed by its low plane
  struct PosTable { int hp, wp; float* dev; void* dev16; };
  bool x16 = false;            // f16 mode with an f16 residual stream (as the reference's .half() path): MARIE_HIP_RESIDUAL_F16
  // f16 mode, default: the residual stream is kept as two f16 planes (x = hi + lo, ~22 significant bits) and every LayerNorm in
  // front of a GEMM is folded around that GEMM (common.h, ConvDesc::epi): hi is the operand, the row statistics come out of the
  // producing GEMM's epilogue.  MARIE_HIP_NO_LN_FOLD=1 when the model is created restores the fp32 stream + LayerNorm passes.
  bool fold = false;
  std::vector<PosTable> pos_tables;
  bool ready = false;
  size_t esz() const { return precision == MHIP_PREC_F16 ? 2 : 4; }
};

struct VitGeom {
  int hp, wp, np;   // patch grid
  int n_tok;        // 1 + np
  int npad;         // rows per image (multiple of 8)
};

struct VitRun {
  VitGeom g;
  void* x = nullptr;          // residual stream [B*npad][D]: fp32, or f16 when the model keeps an f16 stream, or (split) the high plane
  void* x_lo = nullptr;       // split stream: the low plane
  char* tap[4] = {nullptr};   // T [B*np][D] patch tokens after blocks cfg.taps[j]
  char* tokens = nullptr;     // T [B*npad][D] after the final norm (final_norm models)
  char* tokens_dst = nullptr; // set by the caller: the final norm writes here instead of into the workspace (its own slack rows)
};

struct VitFpnOut {
  char* level[4];   // T, D channels: strides 4, 8, 16, 32
  int nest[4];      // 2 / 1 / 0: nested 2x2 row order depth (see vit_fpn)
  int h[4], w[4];
};

void vit_geometry(const mhip_vit* m, int H32, int W32, VitGeom* g);
size_t vit_workspace_bytes(const mhip_vit* m, int B, const VitGeom& g);
size_t vit_fpn_workspace_bytes(const mhip_vit* m, int B, const VitGeom& g);
// imgs: B device images u8 [th][tw][3] placed on a zero canvas H32 x W32 (after normalisation)
int vit_encode(mhip_vit* m, Carver& ws, const uint8_t* imgs, int B, int th, int tw, int H32, int W32, int swap_rb,
               VitRun* run);
int vit_fpn(mhip_vit* m, Carver& ws, int B, const VitRun& run, VitFpnOut* out);

int mhip_gemm(mhip_ctx* ctx, int prec, const void* in, const void* w, long long M, int N, int K, const float* scale,
              const float* bias, void* out, int act, int out_f32, const void* res = nullptr, int ldc = 0, int pad_cols_writable = 0);
