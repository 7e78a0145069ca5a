// conv3x3_patch.hip — 3x3 / pad 1 / stride 1 NHWC convolution on the matrix cores, input patch resident in LDS.
//
// Same contract as conv_igemm (out = pool(act(scale * conv + bias)), NHWC, f16 or exact-fp32 MFMA) for the layer shape
// that carries ~90 % of the path's FLOPs: nn.Conv2d(k=3, padding=1) in
// marie/models/craft/basenet/vgg16_bn.py:27-43, marie/models/craft/craft.py:14-28,46-48 and
// marie/models/icr/modules/feature_extraction.py:13-24.
//
// Why a second kernel: measured on MI355X, the generic tap-gather kernel is bound by the LDS-DMA path (~50 GB/s per
// CU of the ~70 GB/s the path sustains): every filter tap re-stages the whole A tile, 9x per 64-channel slice.
// Here a workgroup owns a PH x PW output patch (256 pixels) of ONE image and stages the (PH+2) x (PW+2) input halo
// patch of a 64-channel slice ONCE; the nine taps are nine shifted ds_read_b128 views of that LDS image.  LDS-DMA
// bytes per FLOP drop 1.8x (N tile 256) to 4.8x (N tile 64); border handling is a zero page at staging time, so the
// tap loop has no bounds checks at all.
//   * 8 waves, every wave owns 64 output columns; N tile 64 / 128 / 256 = 8x1 / 4x2 / 2x4 waves.
//   * K loop = (channel slice) x (tap).  Per iteration only the 128-byte-row weight slice of the next tap and at most
//     one 1-KiB piece of the next channel slice's patch are in flight (counted vmcnt(1), raw s_barrier).
//   * LDS rows are 128 B; the 16-B slot is XOR-swizzled with a function of the halo coordinates chosen so that the 16
//     rows of any fragment read (16 consecutive x, or 2 rows x 8 x for pooled layers), at any tap shift, hit 16
//     distinct slots of the 256-B bank row.
//   * Pixels inside a patch are enumerated pool-first, so max-pool stays an in-register max of a lane's accumulators.
#include "igemm_common.h"

using namespace igemm;

namespace {

template <int BN_>
struct PCfg {
  static constexpr int BN = BN_;
  static constexpr int WN = BN_ / 64;
  static constexpr int WM = 8 / WN;
  static constexpr int MT = 256 / WM / 16;     // 16-pixel MFMA tiles per wave (2, 4 or 8)
  static constexpr int WCHUNKS = BN_ / 64;     // weight chunks staged per thread per tap
  static constexpr int B_BYTES = BN_ * ROWB;
  static constexpr int EPW = BN_ < 128 ? BN_ : 128;
};

// Slot swizzle of a halo row.  ds_read_b128 is served in four 16-lane groups that each mix 8 rows of one k-chunk
// with 8 rows of its neighbour chunk ({0-3,12-15,20-27}, ...); with 128-byte rows, `hx & 7` keeps every group on 16
// distinct 16-B slots for all nine tap shifts, for row-major and for pool-first (2 rows x 8 columns) fragments alike
// (checked exhaustively against the bank model of MI355X_MICROARCH.md; (row>>1)&7 is only conflict-free for fragments
// that start on a multiple of 16 rows, which a tap shift destroys).
template <int POOL>
__device__ __forceinline__ int halo_swz(int hy, int hx) {
  (void)hy;
  return hx & 7;
}

// patch pixel index (pool-first order) -> coordinates inside the PH x PW patch
template <int POOL>
__device__ __forceinline__ void patch_xy(int pm, int pw_shift, int& py, int& px) {
  if (POOL == POOL_NONE) {
    py = pm >> pw_shift;
    px = pm & ((1 << pw_shift) - 1);
  } else if (POOL == POOL_2x2) {
    const int sub = pm & 3, q = pm >> 2;
    const int qx = q & ((1 << (pw_shift - 1)) - 1), qy = q >> (pw_shift - 1);
    py = 2 * qy + (sub >> 1);
    px = 2 * qx + (sub & 1);
  } else {
    const int sub = pm & 1, q = pm >> 1;
    px = q & ((1 << pw_shift) - 1);
    py = 2 * (q >> pw_shift) + sub;
  }
}

template <typename T, int POOL, int BN_>
__global__ __launch_bounds__(NTHREADS) void conv3x3_patch_kernel(IgemmArgs p, int pw_shift, int np_pad, int npieces) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef PCfg<BN_> C;
  typedef typename Tr<T>::chunk_t chunk_t;
  constexpr int E = Tr<T>::E;
  constexpr int BKE = ROWB / sizeof(T);
  constexpr int MT = C::MT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int PW = p.PW, PH = p.PH, HP = PW + 2;
  const int NP = (PH + 2) * HP;
  const int patch_bytes = np_pad * ROWB;
  char* pbuf = smem;                          // [2][np_pad][128]
  char* bbuf = smem + 2 * patch_bytes;        // [2][BN][128]

  // ---- XCD-aware tile assignment ---------------------------------------------------------
  int nt, tx, ty, b;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    nt = L % p.ntiles;
    int t = L / p.ntiles;
    tx = t % p.tiles_x;
    t /= p.tiles_x;
    ty = t % p.tiles_y;
    b = t / p.tiles_y;
  }
  const int y0 = ty * PH, x0 = tx * PW, n0 = nt * C::BN;

  // ---- staging set-up ----------------------------------------------------------------------
  // patch: chunk id c = piece*512 + tid -> LDS row c>>3, physical slot c&7 (lane-linear per wave).  The source
  // address is recomputed per piece (one piece per K iteration) instead of being kept in a runtime-indexed array.
  auto piece_src = [&](int k) -> const char* {
    const int c = k * NTHREADS + tid;
    const int row = c >> 3, pc = c & 7;
    const int hy = row / HP, hx = row - hy * HP;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    if (row < NP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
      const int lc = pc ^ halo_swz<POOL>(hy, hx);
      return p.in + ((((size_t)b * p.H + iy) * p.W + ix) * p.Cin + (size_t)lc * E) * sizeof(T);
    }
    return nullptr;
  };
  // weights: rows n0 + q*64 + srow, same layout and swizzle as conv_igemm
  const int srow = wave * 8 + (lane >> 3);
  const int wl = (lane & 7) ^ ((srow >> 1) & 7);
  const char* w_src[C::WCHUNKS];
#pragma unroll
  for (int q = 0; q < C::WCHUNKS; ++q) {
    const int n = n0 + q * 64 + srow;
    w_src[q] = (n < p.N) ? p.w + ((size_t)n * p.Ktot + (size_t)wl * E) * sizeof(T) : nullptr;
  }
  auto stage_patch_piece = [&](int cs, int k, int buf) {
    const char* base = piece_src(k);
    const char* src = base ? base + (size_t)cs * ROWB : p.zeros;
    glds16(src, pbuf + buf * patch_bytes + (k * NTHREADS + wave * 64) * 16);
  };
  auto stage_w = [&](int cs, int tap, int buf) {
    const size_t off = ((size_t)tap * p.Cin + (size_t)cs * BKE) * sizeof(T);
    char* lb = bbuf + buf * C::B_BYTES + wave * 8 * ROWB;
#pragma unroll
    for (int q = 0; q < C::WCHUNKS; ++q) glds16(w_src[q] ? w_src[q] + off : p.zeros, lb + q * 64 * ROWB);
  };

  // ---- fragment geometry -------------------------------------------------------------------
  const int wr = wave / C::WN, wc = wave % C::WN;
  const int frow = lane & 15, fg = lane >> 4;
  int a_py[MT], a_px[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) patch_xy<POOL>(wr * (MT * 16) + t * 16 + frow, pw_shift, a_py[t], a_px[t]);
  int b_off0[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int rb = wc * 64 + t * 16 + frow;
    b_off0[t] = rb * ROWB + ((fg ^ ((rb >> 1) & 7)) << 4);
  }

  float4v acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

  // ---- main loop over (channel slice, tap) -------------------------------------------------
  const int CS = p.Cin / BKE;
  const int total = CS * 9;
  for (int k = 0; k < npieces; ++k) stage_patch_piece(0, k, 0);
  stage_w(0, 0, 0);
  int cs = 0, tap = 0;
  bool piece_after_w = false;   // did the previous iteration issue a patch piece after its weight slice?
  for (int it = 0; it < total; ++it) {
    if (tap != 0 && piece_after_w) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < total) {
      const int ntap = (tap == 8) ? 0 : tap + 1;
      stage_w(ntap == 0 ? cs + 1 : cs, ntap, (it + 1) & 1);
    }
    piece_after_w = (cs + 1 < CS) && (tap < npieces);
    if (piece_after_w) stage_patch_piece(cs + 1, tap, (cs + 1) & 1);

    const char* pa = pbuf + (cs & 1) * patch_bytes;
    const char* pb = bbuf + (it & 1) * C::B_BYTES;
    const int dy = tap / 3, dx = tap - dy * 3;
    int a_off[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int hy = a_py[t] + dy, hx = a_px[t] + dx;
      a_off[t] = (hy * HP + hx) * ROWB + ((fg ^ halo_swz<POOL>(hy, hx)) << 4);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      chunk_t a[MT], bq[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bq[t] = *(const chunk_t*)(pb + (b_off0[t] ^ (s << 6)));
#pragma unroll
      for (int t = 0; t < MT; ++t) a[t] = *(const chunk_t*)(pa + (a_off[t] ^ (s << 6)));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Tr<T>::mma(a[i], bq[j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
    if (tap == 8) {
      tap = 0;
      ++cs;
    } else {
      ++tap;
    }
  }

  // ---- epilogue: scale/bias, ReLU, in-register max-pool -> LDS -> coalesced NHWC stores -----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  constexpr int PF = (POOL == POOL_2x2) ? 4 : (POOL == POOL_2x1) ? 2 : 1;
  constexpr int RQ = 256 / PF;
  constexpr int EPW = C::EPW;
  const int oe = p.out_f32 ? 4 : (int)sizeof(T);
  const int pitch = EPW * oe + 16;
  const size_t grow = (size_t)p.N * oe;
#pragma unroll
  for (int pass = 0; pass < C::BN / EPW; ++pass) {
    __syncthreads();
    if ((wc >> 1) == pass) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lc = (wc & 1) * 64 + j * 16 + frow;
        const int n = n0 + pass * EPW + lc;
        const float sc = (p.scale && n < p.N) ? p.scale[n] : 1.f;
        const float bi = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = acc[i][j][r] * sc + bi;
            v[r] = p.relu ? fmaxf(t, 0.f) : t;
          }
          const int lr4 = wr * (MT * 16) + i * 16 + fg * 4;
          if (POOL == POOL_2x2) {
            float o = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            if (p.out_f32) lds_put<float>(smem, pitch, lr4 >> 2, lc, o);
            else lds_put<T>(smem, pitch, lr4 >> 2, lc, o);
          } else if (POOL == POOL_2x1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float o = fmaxf(v[2 * h], v[2 * h + 1]);
              if (p.out_f32) lds_put<float>(smem, pitch, (lr4 >> 1) + h, lc, o);
              else lds_put<T>(smem, pitch, (lr4 >> 1) + h, lc, o);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (p.out_f32) lds_put<float>(smem, pitch, lr4 + r, lc, v[r]);
              else lds_put<T>(smem, pitch, lr4 + r, lc, v[r]);
            }
          }
        }
      }
    }
    __syncthreads();
    const int nbase = n0 + pass * EPW;
    const int cpr = EPW * oe / 16, epc = 16 / oe;
    const bool vec = (grow & 15) == 0;
    const int items = vec ? RQ * cpr : RQ * EPW;
    for (int c = tid; c < items; c += NTHREADS) {
      const int row = vec ? c / cpr : c / EPW;
      const int sub = vec ? c - row * cpr : c - row * EPW;
      // staged row -> output pixel
      int oy, ox;
      bool ok;
      if (POOL == POOL_NONE) {
        oy = y0 + (row >> pw_shift);
        ox = x0 + (row & (PW - 1));
        ok = oy < p.Ho && ox < p.Wo;
      } else if (POOL == POOL_2x2) {
        oy = (y0 >> 1) + (row >> (pw_shift - 1));
        ox = (x0 >> 1) + (row & ((PW >> 1) - 1));
        ok = oy < p.Hp && ox < p.Wp;
      } else {
        oy = (y0 >> 1) + (row >> pw_shift);
        ox = x0 + (row & (PW - 1));
        ok = oy < p.Hp && ox < p.Wp;
      }
      if (!ok) continue;
      const size_t q = ((size_t)b * p.Hp + oy) * p.Wp + ox;
      if (vec) {
        const int n = nbase + sub * epc;
        if (n < p.N) *(uint4v*)(p.out + q * grow + (size_t)n * oe) = *(const uint4v*)(smem + row * pitch + sub * 16);
      } else {
        const int n = nbase + sub;
        if (n < p.N) {
          if (oe == 4) *(float*)(p.out + q * grow + (size_t)n * 4) = *(const float*)(smem + row * pitch + sub * 4);
          else *(T*)(p.out + q * grow + (size_t)n * sizeof(T)) = *(const T*)(smem + row * pitch + sub * (int)sizeof(T));
        }
      }
    }
  }
}

template <typename T, int BN_>
int launch_patch(mhip_ctx* ctx, const IgemmArgs& a, int pool, int pw_shift, int np_pad, int npieces, size_t lds) {
  const long long blocks = (long long)a.B * a.tiles_x * a.tiles_y * a.ntiles;
  if (blocks > 0x7fffffffLL) return mhip_fail(ctx, MHIP_EINVAL, "conv3x3_patch: grid too large");
  dim3 grid((unsigned)blocks), block(NTHREADS);
  static std::once_flag attr_set;
  std::call_once(attr_set, [&] {
#define SETATTR(...) (void)hipFuncSetAttribute((const void*)__VA_ARGS__, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)
    SETATTR(conv3x3_patch_kernel<T, POOL_NONE, BN_>);
    SETATTR(conv3x3_patch_kernel<T, POOL_2x2, BN_>);
    SETATTR(conv3x3_patch_kernel<T, POOL_2x1, BN_>);
#undef SETATTR
  });
  switch (pool) {
    case POOL_NONE:
      PROF_LAUNCH(ctx, MHIP_K_IGEMM_PATCH, hipLaunchKernelGGL((conv3x3_patch_kernel<T, POOL_NONE, BN_>), grid, block, lds,
                                                             ctx->stream, a, pw_shift, np_pad, npieces));
      break;
    case POOL_2x2:
      PROF_LAUNCH(ctx, MHIP_K_IGEMM_PATCH, hipLaunchKernelGGL((conv3x3_patch_kernel<T, POOL_2x2, BN_>), grid, block, lds,
                                                             ctx->stream, a, pw_shift, np_pad, npieces));
      break;
    default:
      PROF_LAUNCH(ctx, MHIP_K_IGEMM_PATCH, hipLaunchKernelGGL((conv3x3_patch_kernel<T, POOL_2x1, BN_>), grid, block, lds,
                                                             ctx->stream, a, pw_shift, np_pad, npieces));
      break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "conv3x3_patch launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace

int mhip_try_launch_conv3x3_patch(mhip_ctx* ctx, int precision, const ConvDesc& d, IgemmArgs& a) {
  if (d.KH != 3 || d.KW != 3 || d.pad != 1 || a.dil != 1 || d.in2 || a.sy != 1 || a.pad_x != 1 || a.res || a.ldc)
    return 1;
  // Measured (profiles/r01): the LDS-resident patch removes 45-80 % of the LDS-DMA bytes but the loop stays bound by
  // the one-iteration-deep weight prefetch, so it only ties the tap-gather kernel.  Kept selectable for the next
  // round's deeper weight ring; the default path is conv_igemm.
  if (!getenv("MARIE_HIP_PATCH_KERNEL")) return 1;
  // patch shape: 256 pixels, as square as the output height allows (pooled layers need even PH, PW)
  int PH = 16;
  while (PH > 2 && PH > a.Ho) PH >>= 1;
  if (d.pool != POOL_NONE && PH < 2) return 1;
  if (a.Ho < 2 && d.pool != POOL_NONE) return 1;
  if (PH < 4) return 1;                         // 2 x 128 patches need 9 staging pieces: leave those to conv_igemm
  const int PW = 256 / PH;
  int pw_shift = 0;
  while ((1 << pw_shift) < PW) ++pw_shift;
  const int NP = (PH + 2) * (PW + 2);
  const int np_pad = (NP + 63) / 64 * 64;
  const int npieces = np_pad * 8 / NTHREADS;
  if (npieces > 7) return 1;
  int bn = (a.N > 128) ? 256 : (a.N > 64 ? 128 : 64);
  auto lds_need = [&](int bn_) {
    const size_t ring = (size_t)2 * np_pad * ROWB + (size_t)2 * bn_ * ROWB;
    const size_t epi = (size_t)256 * ((bn_ < 128 ? bn_ : 128) * 4 + 16);
    return ring > epi ? ring : epi;
  };
  if (lds_need(bn) > 163840) bn = 128;
  if (lds_need(bn) > 163840) return 1;
  a.PH = PH;
  a.PW = PW;
  a.tiles_x = (a.Wo + PW - 1) / PW;
  a.tiles_y = (a.Ho + PH - 1) / PH;
  a.ntiles = (a.N + bn - 1) / bn;
  const size_t lds = lds_need(bn);
  if (precision == MHIP_PREC_F16) {
    if (bn == 256) return launch_patch<_Float16, 256>(ctx, a, d.pool, pw_shift, np_pad, npieces, lds);
    if (bn == 128) return launch_patch<_Float16, 128>(ctx, a, d.pool, pw_shift, np_pad, npieces, lds);
    return launch_patch<_Float16, 64>(ctx, a, d.pool, pw_shift, np_pad, npieces, lds);
  }
  if (bn == 256) return launch_patch<float, 256>(ctx, a, d.pool, pw_shift, np_pad, npieces, lds);
  if (bn == 128) return launch_patch<float, 128>(ctx, a, d.pool, pw_shift, np_pad, npieces, lds);
  return launch_patch<float, 64>(ctx, a, d.pool, pw_shift, np_pad, npieces, lds);
}
