// Swin-Transformer encoder / UPerNet decoder kernels around the token GEMM (gemm.hip): evaluation-mode forward of the
// reference's default architecture `swin_*-upernet` (configs/train/config_models.yaml:5,
// configs/config_model_zonal_segmentation.yaml:26 -> flair_hub/models/monotemp_model.py:64-92 ->
// smp.create_model("upernet", "tu-swin_...")).  timm / segmentation_models_pytorch are not vendored by the reference;
// the arithmetic restated here is the published one (Liu et al. 2021, timm's swin_transformer.py; smp 0.4.0
// decoders/upernet), see oracle/swin_upernet.py.
//
// Token tensors are NHWC [B][H][W][C] (a token = a pixel of the stage's map), C a multiple of 8.
//   * space_to_depth      -- PatchEmbed's Conv2d(k = s = 4) as a gather + token GEMM
//   * layer_norm          -- nn.LayerNorm over C; the PatchMerging variant gathers the 2x2 neighbourhood first
//   * window_attention    -- W-MSA / SW-MSA: cyclic shift, padding to the window grid, relative position bias, the
//                            shifted-window mask, softmax, P V, window reverse + un-shift, all by index arithmetic on
//                            the [B][H][W][3C] qkv tensor (no rolled / partitioned copy is materialised)
//   * gelu                -- exact (erf) GELU for the f32 parity mode (the bf16 GEMM applies it in its epilogue)
//   * adaptive_avg_pool   -- PSP module's nn.AdaptiveAvgPool2d
//   * bilinear_slice      -- F.interpolate(bilinear) with either corner convention, written into a channel slice of a
//                            wider tensor (the FPN concat never exists as a separate copy), optional addend (FPN top-down)
#include "ffa_common.h"

#define FFA_TF_THREADS 256

static inline int tf_grid(long long items) {
  long long g = (items + FFA_TF_THREADS - 1) / FFA_TF_THREADS;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------
// space to depth: out[b][y][x][(dy*ps + dx)*C + c] = in[b][y*ps + dy][x*ps + dx][c]

template <typename T>
__global__ void space_to_depth_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Ho, int Wo, int C,
                                      int ps) {
  const int CG = C / 8;
  const int per_tok = ps * ps * CG;
  const long long total = (long long)B * Ho * Wo * per_tok;
  const int Wi = Wo * ps, Hi = Ho * ps;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int r = (int)(i % per_tok);
    long long t = i / per_tok;
    const int g = r % CG;
    r /= CG;
    const int dx = r % ps, dy = r / ps;
    const int x = (int)(t % Wo);
    t /= Wo;
    const int y = (int)(t % Ho);
    const long long b = t / Ho;
    float v[8];
    ffa_load8<T>(in + ((b * Hi + (long long)y * ps + dy) * Wi + (long long)x * ps + dx) * C + g * 8, v);
    ffa_store8<T>(out + i * 8, v);
  }
}

extern "C" int ffa_space_to_depth(int dtype, const void* in, void* out, int B, int Ho, int Wo, int C, int ps,
                                  hipStream_t stream) {
  FFA_REQUIRE(in && out && B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0 && ps > 0, "space_to_depth: bad arguments");
  const long long items = (long long)B * Ho * Wo * ps * ps * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(space_to_depth_kernel<ffa_bf16>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const ffa_bf16*)in, (ffa_bf16*)out, B, Ho, Wo, C, ps);
  else
    hipLaunchKernelGGL(space_to_depth_kernel<float>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const float*)in, (float*)out, B, Ho, Wo, C, ps);
  return ffa_check_launch("space_to_depth");
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension.  A row is shared by LPR lanes (16 / 32 / 64: the smallest power of two covering
// its C / 8 sixteen-byte pieces, so a 128-channel row keeps 16 lanes busy and a wave normalises four rows at once); each
// lane keeps its GPL pieces in registers: one read and one write of the tensor.  Mean first, then the variance of the
// centred values (the two-pass form nn.LayerNorm's f32 kernel is equivalent to), reduced by xor-shuffles inside the group.
// MERGE: the row is PatchMerging's gather (timm swin_transformer.py PatchMerging.forward:
// reshape(B, H/2, 2, W/2, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3)):
// row (b, y, x) of width 4C = [ x(2y, 2x) | x(2y+1, 2x) | x(2y, 2x+1) | x(2y+1, 2x+1) ]; Ho / Wo are the merged sizes.

template <typename T, bool MERGE, int LPR, int GPL>
__global__ void __launch_bounds__(256) layer_norm_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ stats,
                                                         long long rows, int C, int Ho, int Wo, float eps) {
  constexpr int RPB = 256 / LPR;  // rows per block
  const int sub = threadIdx.x % LPR;
  const long long row = blockIdx.x * (long long)RPB + threadIdx.x / LPR;
  const bool live = row < rows;
  const int CG = C / 8;  // output row width in 16-byte pieces
  const int Cs = MERGE ? C / 4 : C;
  const int CGs = Cs / 8;
  long long base[4] = {0, 0, 0, 0};
  if (live) {
    if (MERGE) {
      const int xo = (int)(row % Wo);
      const long long t = row / Wo;
      const int yo = (int)(t % Ho);
      const long long b = t / Ho;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        base[s] = ((b * (2 * Ho) + 2 * yo + (s & 1)) * (2LL * Wo) + 2 * xo + (s >> 1)) * Cs;
    } else {
      base[0] = row * (long long)C;
    }
  }
  float v[GPL][8];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    const int g = sub + LPR * j;
    if (live && g < CG) {
      const T* src = MERGE ? x + base[g / CGs] + (g % CGs) * 8 : x + base[0] + g * 8;
      ffa_load8<T>(src, v[j]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[j][e];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    if (sub + LPR * j < CG) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[j][e] - mean;
        sq += d * d;
      }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  if (!live) return;
  if (stats && sub == 0) {  // kept for the backward pass (training)
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    const int g = sub + LPR * j;
    if (g < CG) {
      float ga[8], be[8], o[8];
      ffa_load8<float>(gamma + g * 8, ga);
      ffa_load8<float>(beta + g * 8, be);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mean) * rstd * ga[e] + be[e];
      ffa_store8<T>(y + row * (long long)C + g * 8, o);
    }
  }
}

template <typename T, bool MERGE>
static int layer_norm_launch(const void* x, void* y, const float* gamma, const float* beta, float* stats, long long rows,
                             int C, int Ho, int Wo, float eps, hipStream_t stream) {
  const int CG = C / 8;
#define FFA_LN_CASE(LPR, GPL)                                                                                        \
  hipLaunchKernelGGL((layer_norm_kernel<T, MERGE, LPR, GPL>), dim3((unsigned)((rows + 256 / LPR - 1) / (256 / LPR))), \
                     dim3(256), 0, stream, (const T*)x, (T*)y, gamma, beta, stats, rows, C, Ho, Wo, eps)
  if (CG <= 16) FFA_LN_CASE(16, 1);
  else if (CG <= 32) FFA_LN_CASE(32, 1);
  else if (CG <= 64) FFA_LN_CASE(64, 1);
  else if (CG <= 128) FFA_LN_CASE(64, 2);
  else if (CG <= 256) FFA_LN_CASE(64, 4);
  else if (CG <= 512) FFA_LN_CASE(64, 8);
  else {
    ffa_set_error("layer_norm: C = %d exceeds 4096", C);
    return FFA_ERR_UNSUPPORTED;
  }
#undef FFA_LN_CASE
  return ffa_check_launch(MERGE ? "patch_merge_norm" : "layer_norm");
}

extern "C" int ffa_layer_norm(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats,
                              long long rows, int C, float eps, hipStream_t stream) {
  FFA_REQUIRE(x && y && gamma && beta && rows > 0 && C > 0 && C % 8 == 0, "layer_norm: bad arguments");
  if (dtype == FFA_BF16) return layer_norm_launch<ffa_bf16, false>(x, y, gamma, beta, stats, rows, C, 0, 0, eps, stream);
  return layer_norm_launch<float, false>(x, y, gamma, beta, stats, rows, C, 0, 0, eps, stream);
}

extern "C" int ffa_patch_merge_norm(int dtype, const void* x, void* y, const float* gamma, const float* beta,
                                    float* stats, int B, int H, int W, int C, float eps, hipStream_t stream) {
  FFA_REQUIRE(x && y && gamma && beta && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "patch_merge_norm: bad arguments");
  FFA_REQUIRE(H % 2 == 0 && W % 2 == 0, "patch_merge_norm: odd map %d x %d (the padded variant is not implemented)", H, W);
  const long long rows = (long long)B * (H / 2) * (W / 2);
  if (dtype == FFA_BF16)
    return layer_norm_launch<ffa_bf16, true>(x, y, gamma, beta, stats, rows, 4 * C, H / 2, W / 2, eps, stream);
  return layer_norm_launch<float, true>(x, y, gamma, beta, stats, rows, 4 * C, H / 2, W / 2, eps, stream);
}

// ---- LayerNorm backward.  dx: the forward's lane layout (one read of x and dy, one write), row statistics taken from
// the forward pass:  xhat = (x - mean) rstd, g = dy gamma, dx = rstd (g - mean_c(g) - xhat mean_c(g xhat)).
// MERGE: x is read through PatchMerging's gather and dx scattered back through it (every input element belongs to
// exactly one merged row).  dgamma / dbeta: a second pass over row chunks (per-thread 8 channels x a strided set of
// rows, block partials to the workspace) and a fixed-order sum of the chunk partials -- deterministic.

template <typename T, bool MERGE, int LPR, int GPL>
__global__ void __launch_bounds__(256) layer_norm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ stats,
                                                             const T* __restrict__ dres, T* __restrict__ dx,
                                                             long long rows, int C, int Ho, int Wo) {
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const long long row = blockIdx.x * (long long)RPB + threadIdx.x / LPR;
  const bool live = row < rows;
  const int CG = C / 8;
  const int Cs = MERGE ? C / 4 : C;
  const int CGs = Cs / 8;
  long long base[4] = {0, 0, 0, 0};
  float mean = 0.f, rstd = 0.f;
  if (live) {
    if (MERGE) {
      const int xo = (int)(row % Wo);
      const long long t = row / Wo;
      const int yo = (int)(t % Ho);
      const long long b = t / Ho;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        base[s] = ((b * (2 * Ho) + 2 * yo + (s & 1)) * (2LL * Wo) + 2 * xo + (s >> 1)) * Cs;
    } else {
      base[0] = row * (long long)C;
    }
    mean = stats[row * 2];
    rstd = stats[row * 2 + 1];
  }
  float xh[GPL][8], gg[GPL][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    const int g = sub + LPR * j;
    if (live && g < CG) {
      const long long off = MERGE ? base[g / CGs] + (g % CGs) * 8 : base[0] + g * 8;
      float xv[8], dv[8], ga[8];
      ffa_load8<T>(x + off, xv);
      ffa_load8<T>(dy + row * (long long)C + g * 8, dv);
      ffa_load8<float>(gamma + g * 8, ga);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[j][e] = (xv[e] - mean) * rstd;
        gg[j][e] = dv[e] * ga[e];
        s1 += gg[j][e];
        s2 += gg[j][e] * xh[j][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) xh[j][e] = gg[j][e] = 0.f;
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (!live) return;
  const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    const int g = sub + LPR * j;
    if (g < CG) {
      const long long off = MERGE ? base[g / CGs] + (g % CGs) * 8 : base[0] + g * 8;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (gg[j][e] - m1 - xh[j][e] * m2);
      if (dres) {  // the residual connection around the normalised branch: dx = dres + LayerNorm'(dy)
        float r[8];
        ffa_load8<T>(dres + off, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += r[e];
      }
      ffa_store8<T>(dx + off, o);
    }
  }
}

// partial[chunk][0][c] = sum_r dy xhat, partial[chunk][1][c] = sum_r dy over the chunk's rows
template <typename T, bool MERGE>
__global__ void __launch_bounds__(256) layer_norm_bwd_params_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    const float* __restrict__ stats,
                                                                    float* __restrict__ partial, long long rows, int C,
                                                                    int Ho, int Wo, long long rows_per_chunk) {
  __shared__ float red[8][32][16];
  const int CG = C / 8;
  const int Cs = MERGE ? C / 4 : C;
  const int CGs = Cs / 8;
  const int gl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int g = blockIdx.x * 32 + gl;
  const long long r0 = blockIdx.y * rows_per_chunk;
  long long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  if (g < CG) {
    for (long long row = r0 + rl; row < r1; row += 8) {
      long long off;
      if (MERGE) {
        const int xo = (int)(row % Wo);
        const long long t = row / Wo;
        const int yo = (int)(t % Ho);
        const long long b = t / Ho;
        const int s = g / CGs;
        off = ((b * (2 * Ho) + 2 * yo + (s & 1)) * (2LL * Wo) + 2 * xo + (s >> 1)) * Cs + (g % CGs) * 8;
      } else {
        off = row * (long long)C + g * 8;
      }
      float xv[8], dv[8];
      ffa_load8<T>(x + off, xv);
      ffa_load8<T>(dy + row * (long long)C + g * 8, dv);
      const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc[e] += dv[e] * ((xv[e] - mean) * rstd);
        acc[8 + e] += dv[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[rl][gl][e] = acc[e];
  __syncthreads();
  if (rl == 0 && g < CG) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) v += red[r][gl][e];
      partial[((long long)blockIdx.y * 2 + (e >> 3)) * C + g * 8 + (e & 7)] = v;
    }
  }
}

__global__ void __launch_bounds__(256) layer_norm_bwd_reduce_kernel(const float* __restrict__ partial,
                                                                    float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, int C, int chunks) {
  // 32 lanes per channel over the chunks, combined by xor-shuffles (fixed order)
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int l = threadIdx.x & 31;
  float a = 0.f, b = 0.f;
  if (c < C)
    for (int k = l; k < chunks; k += 32) {
      a += partial[((long long)k * 2) * C + c];
      b += partial[((long long)k * 2 + 1) * C + c];
    }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (c < C && l == 0) {
    dgamma[c] = a;
    dbeta[c] = b;
  }
}

static inline int ln_bwd_chunks(long long rows) {
  long long c = (rows + 255) / 256;
  if (c > 512) c = 512;
  if (c < 1) c = 1;
  return (int)c;
}

extern "C" long long ffa_layer_norm_bwd_workspace_bytes(long long rows, int C) {
  return (long long)ln_bwd_chunks(rows) * 2 * C * (long long)sizeof(float);
}

template <typename T, bool MERGE>
static int layer_norm_bwd_launch(const void* x, const void* dy, const float* gamma, const float* stats,
                                 const void* dres, void* dx, float* dgamma, float* dbeta, long long rows, int C, int Ho, int Wo, void* workspace,
                                 long long workspace_bytes, hipStream_t stream) {
  const int CG = C / 8;
  if (workspace_bytes < ffa_layer_norm_bwd_workspace_bytes(rows, C) || !workspace) {
    ffa_set_error("layer_norm_bwd: workspace of %lld bytes needed", ffa_layer_norm_bwd_workspace_bytes(rows, C));
    return FFA_ERR_WORKSPACE;
  }
#define FFA_LNB_CASE(LPR, GPL)                                                                                            \
  hipLaunchKernelGGL((layer_norm_bwd_kernel<T, MERGE, LPR, GPL>), dim3((unsigned)((rows + 256 / LPR - 1) / (256 / LPR))), \
                     dim3(256), 0, stream, (const T*)x, (const T*)dy, gamma, stats, (const T*)dres, (T*)dx, rows, C, Ho, Wo)
  if (CG <= 16) FFA_LNB_CASE(16, 1);
  else if (CG <= 32) FFA_LNB_CASE(32, 1);
  else if (CG <= 64) FFA_LNB_CASE(64, 1);
  else if (CG <= 128) FFA_LNB_CASE(64, 2);
  else if (CG <= 256) FFA_LNB_CASE(64, 4);
  else if (CG <= 512) FFA_LNB_CASE(64, 8);
  else {
    ffa_set_error("layer_norm_bwd: C = %d exceeds 4096", C);
    return FFA_ERR_UNSUPPORTED;
  }
#undef FFA_LNB_CASE
  const int chunks = ln_bwd_chunks(rows);
  const long long rpc = (rows + chunks - 1) / chunks;
  hipLaunchKernelGGL((layer_norm_bwd_params_kernel<T, MERGE>), dim3((unsigned)((CG + 31) / 32), (unsigned)chunks), dim3(256),
                     0, stream, (const T*)x, (const T*)dy, stats, (float*)workspace, rows, C, Ho, Wo, rpc);
  hipLaunchKernelGGL(layer_norm_bwd_reduce_kernel, dim3((unsigned)((C + 7) / 8)), dim3(256), 0, stream,
                     (const float*)workspace, dgamma, dbeta, C, chunks);
  return ffa_check_launch(MERGE ? "patch_merge_norm_bwd" : "layer_norm_bwd");
}

extern "C" int ffa_layer_norm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* stats,
                                  const void* dres, void* dx, float* dgamma, float* dbeta, long long rows, int C, void* workspace,
                                  long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(x && dy && gamma && stats && dx && dgamma && dbeta && rows > 0 && C > 0 && C % 8 == 0,
              "layer_norm_bwd: bad arguments");
  if (dtype == FFA_BF16)
    return layer_norm_bwd_launch<ffa_bf16, false>(x, dy, gamma, stats, dres, dx, dgamma, dbeta, rows, C, 0, 0, workspace,
                                                  workspace_bytes, stream);
  return layer_norm_bwd_launch<float, false>(x, dy, gamma, stats, dres, dx, dgamma, dbeta, rows, C, 0, 0, workspace,
                                             workspace_bytes, stream);
}

extern "C" int ffa_patch_merge_norm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* stats,
                                        void* dx, float* dgamma, float* dbeta, int B, int H, int W, int C, void* workspace,
                                        long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(x && dy && gamma && stats && dx && dgamma && dbeta && B > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0 &&
                  H % 2 == 0 && W % 2 == 0, "patch_merge_norm_bwd: bad arguments");
  const long long rows = (long long)B * (H / 2) * (W / 2);
  if (dtype == FFA_BF16)
    return layer_norm_bwd_launch<ffa_bf16, true>(x, dy, gamma, stats, nullptr, dx, dgamma, dbeta, rows, 4 * C, H / 2, W / 2,
                                                 workspace, workspace_bytes, stream);
  return layer_norm_bwd_launch<float, true>(x, dy, gamma, stats, nullptr, dx, dgamma, dbeta, rows, 4 * C, H / 2, W / 2,
                                            workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// GELU (exact): 0.5 x (1 + erf(x / sqrt 2))

__device__ __forceinline__ float ffa_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename T>
__global__ void gelu_kernel(const T* __restrict__ x, T* __restrict__ y, long long n8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    ffa_load8<T>(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ffa_gelu(v[e]);
    ffa_store8<T>(y + i * 8, v);
  }
}

extern "C" int ffa_gelu(int dtype, const void* x, void* y, long long n, hipStream_t stream) {
  FFA_REQUIRE(x && y && n > 0 && n % 8 == 0, "gelu: bad arguments");
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(gelu_kernel<ffa_bf16>, dim3(tf_grid(n / 8)), dim3(FFA_TF_THREADS), 0, stream, (const ffa_bf16*)x,
                       (ffa_bf16*)y, n / 8);
  else
    hipLaunchKernelGGL(gelu_kernel<float>, dim3(tf_grid(n / 8)), dim3(FFA_TF_THREADS), 0, stream, (const float*)x,
                       (float*)y, n / 8);
  return ffa_check_launch("gelu");
}

// ------------------------------------------------------------------------------------------------
// Window attention (timm swin_transformer.py SwinTransformerBlock._attn + WindowAttention.forward).
//
// Shifted / padded frame: position (hp, wp), hp < Hp = ceil(H / ws) ws.  hp < H and wp < W: the token at
// ((hp + shift) mod H, (wp + shift) mod W) of the map (torch.roll by -shift, THEN F.pad at the bottom / right);
// otherwise a padding token, whose q / k / v are the qkv bias (norm1's output is padded with zeros before the qkv
// projection).  Window (wy, wx) holds positions [wy ws, (wy+1) ws) x [wx ws, (wx+1) ws), token t = ty ws + tx.
//   attn[i][j] = scale q_i . k_j + table[(yi - yj + ws - 1)(2 ws - 1) + (xi - xj + ws - 1)][head]
//                + (region(i) != region(j) ? -100 : 0)                              (shift > 0 only)
// region = 3 rh + rw with rh = 0 / 1 / 2 for hp in [0, Hp - ws) / [Hp - ws, Hp - shift) / [Hp - shift, Hp).
// The result of a real token goes back to its un-shifted place in out[B][H][W][C]; padding queries are dropped.

struct WinAttnArgs {
  const void* qkv;
  void* out;
  const float* qkv_bias;  // [3C]
  const float* table;     // [(2 ws - 1)^2][heads]
  int B, H, W, C, heads, ws, shift, nwy, nwx;
  float scale;
};

__device__ __forceinline__ int win_region(int p, int P, int ws, int shift) {
  return p < P - ws ? 0 : (p < P - shift ? 1 : 2);
}

// token -> element offset of its q row in qkv (or -1 for a padding token), region id, relative-position line index
struct WinTok {
  long long off;
  int rid, lin;
};
__device__ __forceinline__ WinTok win_token(const WinAttnArgs& a, int b, int wy, int wx, int t) {
  const int ty = t / a.ws, tx = t % a.ws;
  const int hp = wy * a.ws + ty, wp = wx * a.ws + tx;
  WinTok r;
  r.lin = ty * (2 * a.ws - 1) + tx;
  r.rid = a.shift ? win_region(hp, a.nwy * a.ws, a.ws, a.shift) * 3 + win_region(wp, a.nwx * a.ws, a.ws, a.shift) : 0;
  if (hp < a.H && wp < a.W) {
    int y = hp + a.shift, x = wp + a.shift;
    if (y >= a.H) y -= a.H;
    if (x >= a.W) x -= a.W;
    r.off = ((long long)b * a.H + y) * a.W + x;
  } else {
    r.off = -1;
  }
  return r;
}

// ---- f32 parity path: one block per (window, head), one wave per query, plain FMA loops
__global__ void __launch_bounds__(256) window_attention_f32_kernel(WinAttnArgs a) {
  extern __shared__ unsigned char smem_raw[];
  const int N = a.ws * a.ws;
  float* sq = reinterpret_cast<float*>(smem_raw);  // [N][33]
  float* sk = sq + N * 33;
  float* sv = sk + N * 33;
  float* sp = sv + N * 33;                          // [4 waves][N]
  float* stab = sp + 4 * N;                         // [(2ws-1)^2]
  int* slin = reinterpret_cast<int*>(stab + (2 * a.ws - 1) * (2 * a.ws - 1));  // [N]
  int* srid = slin + N;                                                         // [N]
  long long* soff = reinterpret_cast<long long*>((reinterpret_cast<uintptr_t>(srid + N) + 7) & ~(uintptr_t)7);  // [N]
  const int head = blockIdx.y;
  int w = blockIdx.x;
  const int wx = w % a.nwx;
  w /= a.nwx;
  const int wy = w % a.nwy;
  const int b = w / a.nwy;
  const float* qkv = (const float*)a.qkv;
  const int C3 = 3 * a.C;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const WinTok tk = win_token(a, b, wy, wx, i);
    slin[i] = tk.lin;
    srid[i] = tk.rid;
    soff[i] = tk.off;
  }
  const int TS = (2 * a.ws - 1) * (2 * a.ws - 1);
  for (int i = threadIdx.x; i < TS; i += blockDim.x) stab[i] = a.table[i * a.heads + head];
  __syncthreads();
  for (int i = threadIdx.x; i < N * 32; i += blockDim.x) {
    const int t = i >> 5, d = i & 31;
    const long long off = soff[t];
    const int c = head * 32 + d;
    float q, k, v;
    if (off >= 0) {
      const float* p = qkv + off * C3;
      q = p[c];
      k = p[a.C + c];
      v = p[2 * a.C + c];
    } else {
      q = a.qkv_bias[c];
      k = a.qkv_bias[a.C + c];
      v = a.qkv_bias[2 * a.C + c];
    }
    sq[t * 33 + d] = q * a.scale;
    sk[t * 33 + d] = k;
    sv[t * 33 + d] = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* p = sp + wave * N;
  const int off0 = (a.ws - 1) * (2 * a.ws - 1) + (a.ws - 1);
  for (int i = wave; i < N; i += 4) {
    if (soff[i] < 0) continue;  // wave-uniform
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 32; ++d) s += sq[i * 33 + d] * sk[j * 33 + d];
      s += stab[slin[i] - slin[j] + off0];
      if (srid[i] != srid[j]) s += -100.0f;
      p[j] = s;
      mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float e = expf(p[j] - mx);
      p[j] = e;
      sum += e;
    }
    sum = ffa_wave_sum(sum);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): p[] written by the whole wave before it is read across lanes
    __builtin_amdgcn_wave_barrier();
    // lanes 0..31 own one output channel each, the upper half-wave takes the odd keys
    const int d = lane & 31, half = lane >> 5;
    float acc = 0.f;
    for (int j = half; j < N; j += 2) acc += p[j] * sv[j * 33 + d];
    acc += __shfl_xor(acc, 32, 64);
    if (half == 0) ((float*)a.out)[soff[i] * a.C + head * 32 + d] = acc / sum;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- bf16 path: MFMA 16x16x32.  One block per (window, head), NW waves, each wave takes 16-query tiles round-robin.
// LDS holds only what the waves share: k [NP][32] bf16 at an 80-byte row pitch (conflict-free 16-byte fragment reads),
// v row-major [NP][32] at a 96-byte pitch (conflict-free ds_read_b64_tr_b16 transposed reads), one packed word per key
// (relative-position line index | region id << 16) and the head's bias table.  A lane's q fragment comes straight from
// global memory.  S^T = K Q^T per tile (lane: query n = lane % 16, keys 16 t + 4 (lane / 16) + i), softmax across the
// four lane groups of a query by two xor-shuffles, then O^T = V^T P^T with P^T taken straight from the S^T registers:
// the k-slot order of a 32-key MFMA block is [tile 2u keys 4g..4g+3 | tile 2u+1 keys 4g..4g+3] for lane group g, which
// is what two transposed reads of V (4 keys x 16 channels each) deliver.
__device__ __forceinline__ ffa_s16x4 attn_read_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) ffa_s16x4*)(const_cast<unsigned char*>(p)));
}

template <int NTP, int NW>  // padded key tiles of 16 (even): 4 (ws <= 8) or 10 (ws <= 12); waves per block
__global__ void __launch_bounds__(64 * NW) window_attention_bf16_kernel(WinAttnArgs a) {
  constexpr int NP = NTP * 16;
  constexpr int KP = 80;  // k row pitch in bytes
  constexpr int VP = 96;  // v row pitch in bytes
  __shared__ __attribute__((aligned(16))) unsigned char sk[NP * KP];
  __shared__ __attribute__((aligned(16))) unsigned char sv[NP * VP];
  __shared__ __attribute__((aligned(16))) int skey[NP];
  __shared__ float stab[23 * 23];
  const int N = a.ws * a.ws;
  const int head = blockIdx.y;
  int w = blockIdx.x;
  const int wx = w % a.nwx;
  w /= a.nwx;
  const int wy = w % a.nwy;
  const int b = w / a.nwy;
  const ffa_bf16* qkv = (const ffa_bf16*)a.qkv;
  const int C3 = 3 * a.C;
  const int TS = (2 * a.ws - 1) * (2 * a.ws - 1);
  // everything in log2 units: softmax(x) = 2^(x log2e - max) / sum, one v_exp_f32 per element and no multiply
  constexpr float LOG2E = 1.4426950408889634f;
  for (int i = threadIdx.x; i < TS; i += 64 * NW) stab[i] = a.table[i * a.heads + head] * LOG2E;
  const float scale2 = a.scale * LOG2E;
  // only the windows of the last window row / column hold more than one region of the shifted-window mask
  const bool masked = a.shift > 0 && (wy == a.nwy - 1 || wx == a.nwx - 1);
  // stage k and v: 4 pieces of 8 channels per key and operand; keys past the window are zero and flagged
  for (int i = threadIdx.x; i < NP * 4; i += 64 * NW) {
    const int t = i >> 2, pc = i & 3;
    const int c = head * 32 + pc * 8;
    ffa_u32x4 k4 = {0u, 0u, 0u, 0u}, v4 = {0u, 0u, 0u, 0u};
    int packed = 0xffff0000;  // region id -1: excluded from the softmax
    if (t < N) {
      const WinTok tk = win_token(a, b, wy, wx, t);
      packed = tk.lin | (tk.rid << 16);
      if (tk.off >= 0) {
        const ffa_bf16* p = qkv + tk.off * C3 + a.C + c;
        k4 = *reinterpret_cast<const ffa_u32x4*>(p);
        v4 = *reinterpret_cast<const ffa_u32x4*>(p + a.C);
      } else {
        // a padding token's projection is the bias itself, rounded like every other element of the qkv tensor
        float kf[8], vf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          kf[e] = a.qkv_bias[a.C + c + e];
          vf[e] = a.qkv_bias[2 * a.C + c + e];
        }
        ffa_store8<ffa_bf16>(reinterpret_cast<ffa_bf16*>(&k4), kf);
        ffa_store8<ffa_bf16>(reinterpret_cast<ffa_bf16*>(&v4), vf);
      }
    }
    *reinterpret_cast<ffa_u32x4*>(sk + t * KP + pc * 16) = k4;
    *reinterpret_cast<ffa_u32x4*>(sv + t * VP + pc * 16) = v4;
    if (pc == 0) skey[t] = packed;
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int off0 = (a.ws - 1) * (2 * a.ws - 1) + (a.ws - 1);
  const int nqt = (N + 15) / 16;
  // transposed-read address of this lane inside a 4-key x 16-channel block: row (lane / 4) % 4, channels 4 (lane % 4)
  const int tr_off = ((lane >> 2) & 3) * VP + (lane & 3) * 8;
  for (int qt = wave; qt < nqt; qt += NW) {  // wave-uniform trip count: EXEC stays full for the transposed reads
    const int qi = qt * 16 + n;  // this lane's query (may be >= N: computed, never stored)
    long long qoff = -1;
    int qlin = off0, qrid = -2;
    ffa_bf16x8 qf;
    {
      ffa_u32x4 q4 = {0u, 0u, 0u, 0u};
      if (qi < N) {
        const WinTok tk = win_token(a, b, wy, wx, qi);
        qlin += tk.lin;
        qrid = tk.rid;
        qoff = tk.off;
        if (qoff >= 0) {
          q4 = *reinterpret_cast<const ffa_u32x4*>(qkv + qoff * C3 + head * 32 + g * 8);
        } else {
          float qv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) qv[e] = a.qkv_bias[head * 32 + g * 8 + e];
          ffa_store8<ffa_bf16>(reinterpret_cast<ffa_bf16*>(&q4), qv);
        }
      }
      qf = __builtin_bit_cast(ffa_bf16x8, q4);
    }
    ffa_f32x4 s[NTP];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      if (t >= nqt) {  // a tile of padding keys only (block-uniform): probability 0, no work
        s[t] = ffa_f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        continue;
      }
      const ffa_bf16x8 kf = *reinterpret_cast<const ffa_bf16x8*>(sk + (t * 16 + n) * KP + g * 16);
      const int4 key4 = *reinterpret_cast<const int4*>(&skey[t * 16 + g * 4]);
      ffa_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, acc, 0, 0, 0);
      const int kk[4] = {key4.x, key4.y, key4.z, key4.w};
      const bool ragged = (t + 1) * 16 > N;  // the one tile that mixes keys and padding (block-uniform)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int klin = kk[i] & 0xffff;
        float v = acc[i] * scale2 + stab[qlin - klin];
        if (masked && (kk[i] >> 16) != qrid) v += -100.0f * LOG2E;
        if (ragged && kk[i] < 0) v = -INFINITY;
        acc[i] = v;
        mx = fmaxf(mx, v);
      }
      s[t] = acc;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      if (t >= nqt) {
        s[t] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};
        continue;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float e = __builtin_amdgcn_exp2f(s[t][i] - mx);
        s[t][i] = e;
        sum += e;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    ffa_f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NTP / 2; ++u) {
      ffa_bf16x8 pf;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pf[i] = (__bf16)s[2 * u][i];
        pf[4 + i] = (__bf16)s[2 * u + 1][i];
      }
      // lane group g: keys 32u + 4g .. +3 (tile 2u) and 32u + 16 + 4g .. +3 (tile 2u + 1), channels 0-15 / 16-31
      const unsigned char* vb = sv + (u * 32 + g * 4) * VP + tr_off;
      const ffa_s16x4 a00 = attn_read_tr16(vb);
      const ffa_s16x4 a01 = attn_read_tr16(vb + 16 * VP);
      const ffa_s16x4 a10 = attn_read_tr16(vb + 32);
      const ffa_s16x4 a11 = attn_read_tr16(vb + 16 * VP + 32);
      ffa_u32x4 v0, v1;
      v0.x = __builtin_bit_cast(ffa_u32x2, a00).x; v0.y = __builtin_bit_cast(ffa_u32x2, a00).y;
      v0.z = __builtin_bit_cast(ffa_u32x2, a01).x; v0.w = __builtin_bit_cast(ffa_u32x2, a01).y;
      v1.x = __builtin_bit_cast(ffa_u32x2, a10).x; v1.y = __builtin_bit_cast(ffa_u32x2, a10).y;
      v1.z = __builtin_bit_cast(ffa_u32x2, a11).x; v1.w = __builtin_bit_cast(ffa_u32x2, a11).y;
      o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ffa_bf16x8, v0), pf, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ffa_bf16x8, v1), pf, o1, 0, 0, 0);
    }
    // lane: query n, channels 4g..4g+3 (o0) and 16+4g..16+4g+3 (o1)
    if (qoff >= 0) {
      const float inv = 1.0f / sum;
      ffa_bf16* dst = (ffa_bf16*)a.out + qoff * a.C + head * 32 + g * 4;
      uint2 lo, hi;
      lo.x = ffa_pack_bf16x2(o0[0] * inv, o0[1] * inv);
      lo.y = ffa_pack_bf16x2(o0[2] * inv, o0[3] * inv);
      hi.x = ffa_pack_bf16x2(o1[0] * inv, o1[1] * inv);
      hi.y = ffa_pack_bf16x2(o1[2] * inv, o1[3] * inv);
      *reinterpret_cast<uint2*>(dst) = lo;
      *reinterpret_cast<uint2*>(dst + 16) = hi;
    }
  }
}

extern "C" int ffa_window_attention(int dtype, const void* qkv, void* out, const float* qkv_bias, const float* table,
                                    int B, int H, int W, int C, int heads, int ws, int shift, float scale,
                                    hipStream_t stream) {
  FFA_REQUIRE(qkv && out && qkv_bias && table && B > 0 && H > 0 && W > 0 && heads > 0, "window_attention: bad arguments");
  FFA_REQUIRE(C == heads * 32, "window_attention: head dimension %d (only 32 is built)", heads ? C / heads : 0);
  FFA_REQUIRE(ws >= 1 && ws <= 12 && shift >= 0 && shift < ws, "window_attention: window %d / shift %d", ws, shift);
  WinAttnArgs a;
  a.qkv = qkv;
  a.out = out;
  a.qkv_bias = qkv_bias;
  a.table = table;
  a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.ws = ws; a.shift = shift;
  a.nwy = (H + ws - 1) / ws;
  a.nwx = (W + ws - 1) / ws;
  a.scale = scale;
  const long long nwin = (long long)B * a.nwy * a.nwx;
  FFA_REQUIRE(nwin < (1LL << 31) && heads < 65536, "window_attention: grid too large");
  const dim3 grid((unsigned)nwin, (unsigned)heads);
  const int N = ws * ws;
  if (dtype == FFA_BF16) {
    if (N <= 64)
      hipLaunchKernelGGL((window_attention_bf16_kernel<4, 4>), grid, dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL((window_attention_bf16_kernel<10, 3>), grid, dim3(192), 0, stream, a);
  } else {
    const int TS = (2 * ws - 1) * (2 * ws - 1);
    const size_t lds = (size_t)(3 * N * 33 + 4 * N + TS) * 4 + (size_t)(2 * N + (N & 1)) * 4 + (size_t)N * 8 + 16;
    FFA_REQUIRE(lds <= 64 * 1024, "window_attention: window %d needs %zu bytes of LDS in f32 mode", ws, lds);
    hipLaunchKernelGGL(window_attention_f32_kernel, grid, dim3(256), lds, stream, a);
  }
  return ffa_check_launch("window_attention");
}

// ------------------------------------------------------------------------------------------------
// nn.AdaptiveAvgPool2d(S): out[b][i][j] = mean of in[b][floor(i H / S) : ceil((i+1) H / S)][... same for W]

// one block per output cell: 256 threads = (C / 8 pieces, up to 32) x (8 or more pixel lanes) walk the cell's region
// together and combine through LDS in a fixed order (the regions are up to the whole map for S = 1)
template <typename T>
__global__ void __launch_bounds__(256) adaptive_avg_pool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H,
                                                                int W, int C, int S) {
  __shared__ float red[256][8];
  const int CG = C / 8;
  int cell = blockIdx.x;
  const int ox = cell % S;
  cell /= S;
  const int oy = cell % S;
  const long long b = cell / S;
  const int y0 = (oy * H) / S, y1 = ((oy + 1) * H + S - 1) / S;
  const int x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
  const int rw = x1 - x0, npix = (y1 - y0) * rw;
  const float inv = 1.0f / (float)npix;
  for (int g0 = blockIdx.y * 32; g0 < CG; g0 += gridDim.y * 32) {
    const int g = g0 + (threadIdx.x & 31), pl = threadIdx.x >> 5;  // piece, pixel lane 0..7
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (g < CG)
      for (int p = pl; p < npix; p += 8) {
        const int yy = y0 + p / rw, xx = x0 + p % rw;
        float v[8];
        ffa_load8<T>(x + ((b * H + yy) * W + xx) * C + g * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (pl == 0 && g < CG) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) v += red[r * 32 + (threadIdx.x & 31)][e];
        o[e] = v * inv;
      }
      ffa_store8<T>(y + ((b * S + oy) * S + ox) * (long long)C + g * 8, o);
    }
  }
}

extern "C" int ffa_adaptive_avg_pool(int dtype, const void* x, void* y, int B, int H, int W, int C, int S,
                                     hipStream_t stream) {
  FFA_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && S > 0, "adaptive_avg_pool: bad arguments");
  const int CG = C / 8;
  const dim3 grid((unsigned)(B * S * S), (unsigned)((CG + 31) / 32 < 8 ? (CG + 31) / 32 : 8));
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(adaptive_avg_pool_kernel<ffa_bf16>, grid, dim3(256), 0, stream, (const ffa_bf16*)x, (ffa_bf16*)y,
                       B, H, W, C, S);
  else
    hipLaunchKernelGGL(adaptive_avg_pool_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, (float*)y, B, H, W,
                       C, S);
  return ffa_check_launch("adaptive_avg_pool");
}

// ------------------------------------------------------------------------------------------------
// Bilinear resize into a channel slice: y[b][oy][ox][y_off + c] = bilinear(x)[b][oy][ox][c] (+ addend[b][oy][ox][c]).
// align_corners = 0: ATen's area_pixel_compute_source_index as in resample_loss.hip; = 1 (nn.UpsamplingBilinear2d of
// smp's SegmentationHead): src = dst * (in - 1) / (out - 1), the ratio formed in f32 first like ATen does.

__device__ __forceinline__ void bilinear_src2(int dst, float scale, int in_size, int align, int& i0, int& i1, float& l0,
                                              float& l1) {
  float src = align ? scale * (float)dst : scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

template <typename T>
__global__ void bilinear_slice_kernel(const T* __restrict__ x, const T* __restrict__ addend, T* __restrict__ y, int B,
                                      int Hi, int Wi, int Ho, int Wo, int C, int y_pitch, int y_off, int align,
                                      float sy, float sx) {
  const int CG = C / 8;
  const long long total = (long long)B * Ho * Wo * CG;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % CG);
    long long p = i / CG;
    const long long pix = p;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const long long b = p / Ho;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilinear_src2(oy, sy, Hi, align, y0, y1, ly0, ly1);
    bilinear_src2(ox, sx, Wi, align, x0, x1, lx0, lx1);
    float v00[8], v01[8], v10[8], v11[8], o[8];
    ffa_load8<T>(x + ((b * Hi + y0) * Wi + x0) * C + g * 8, v00);
    ffa_load8<T>(x + ((b * Hi + y0) * Wi + x1) * C + g * 8, v01);
    ffa_load8<T>(x + ((b * Hi + y1) * Wi + x0) * C + g * 8, v10);
    ffa_load8<T>(x + ((b * Hi + y1) * Wi + x1) * C + g * 8, v11);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
    if (addend) {
      float ad[8];
      ffa_load8<T>(addend + i * 8, ad);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += ad[e];
    }
    ffa_store8<T>(y + pix * y_pitch + y_off + g * 8, o);
  }
}

extern "C" int ffa_bilinear_slice(int dtype, const void* x, const void* addend, void* y, int B, int Hi, int Wi, int Ho,
                                  int Wo, int C, int y_pitch, int y_off, int align_corners, hipStream_t stream) {
  FFA_REQUIRE(x && y && C > 0 && C % 8 == 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && B > 0, "bilinear_slice: bad arguments");
  FFA_REQUIRE(y_pitch % 8 == 0 && y_off % 8 == 0 && y_off >= 0 && y_off + C <= y_pitch,
              "bilinear_slice: slice [%d, %d) does not fit pitch %d", y_off, y_off + C, y_pitch);
  float sy, sx;
  if (align_corners) {
    sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  } else {
    sy = (float)Hi / (float)Ho;
    sx = (float)Wi / (float)Wo;
  }
  const long long items = (long long)B * Ho * Wo * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(bilinear_slice_kernel<ffa_bf16>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const ffa_bf16*)x, (const ffa_bf16*)addend, (ffa_bf16*)y, B, Hi, Wi, Ho, Wo, C, y_pitch, y_off,
                       align_corners ? 1 : 0, sy, sx);
  else
    hipLaunchKernelGGL(bilinear_slice_kernel<float>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const float*)x, (const float*)addend, (float*)y, B, Hi, Wi, Ho, Wo, C, y_pitch, y_off,
                       align_corners ? 1 : 0, sy, sx);
  return ffa_check_launch("bilinear_slice");
}

// ---- backward of ffa_bilinear_slice in GATHER form (the scheme of bilinear_bwd_kernel in resample_loss.hip, with
// either corner convention and the gradient read from a channel slice of a wider tensor): one thread per 8 channels of
// a SOURCE pixel walks the destination pixels whose footprint contains it; membership and weights come from the very
// bilinear_src2() the forward uses; fixed summation order, no atomics.
template <typename T>
__global__ void bilinear_slice_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Hi, int Wi, int Ho,
                                          int Wo, int C, int y_pitch, int y_off, int align, float sy, float sx) {
  const int CG = C / 8;
  const long long total = (long long)B * Hi * Wi * CG;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % CG);
    long long p = i / CG;
    const int x = (int)(p % Wi);
    p /= Wi;
    const int y = (int)(p % Hi);
    const long long b = p / Hi;
    // candidate destination rows / columns: source coordinate within one pixel of (y, x), one extra on each side for
    // the float rounding of the bounds; exact membership is decided per candidate below
    int oy_lo = 0, oy_hi = Ho - 1, ox_lo = 0, ox_hi = Wo - 1;
    if (sy > 1e-12f) {
      const float a = align ? ((float)y - 1.0f) / sy : ((float)y - 0.5f) / sy - 0.5f;
      const float c = align ? ((float)y + 1.0f) / sy : ((float)y + 1.5f) / sy - 0.5f;
      oy_lo = (int)floorf(a) - 1;
      oy_hi = (int)ceilf(c) + 1;
    }
    if (sx > 1e-12f) {
      const float a = align ? ((float)x - 1.0f) / sx : ((float)x - 0.5f) / sx - 0.5f;
      const float c = align ? ((float)x + 1.0f) / sx : ((float)x + 1.5f) / sx - 0.5f;
      ox_lo = (int)floorf(a) - 1;
      ox_hi = (int)ceilf(c) + 1;
    }
    oy_lo = oy_lo < 0 ? 0 : oy_lo;
    ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi;
    ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly0, ly1;
      bilinear_src2(oy, sy, Hi, align, y0, y1, ly0, ly1);
      if (y0 != y && y1 != y) continue;
      const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);  // y0 == y1 on the clamped last row
      float row[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) row[e] = 0.f;
      const T* drow = dy + ((b * Ho + oy) * Wo) * y_pitch + y_off + g * 8;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        bilinear_src2(ox, sx, Wi, align, x0, x1, lx0, lx1);
        if (x0 != x && x1 != x) continue;
        const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
        float gv[8];
        ffa_load8<T>(drow + (long long)ox * y_pitch, gv);
#pragma unroll
        for (int e = 0; e < 8; ++e) row[e] += wx * gv[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += wy * row[e];
    }
    ffa_store8<T>(dx + i * 8, acc);
  }
}

extern "C" int ffa_bilinear_slice_bwd(int dtype, const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                      int y_pitch, int y_off, int align_corners, hipStream_t stream) {
  FFA_REQUIRE(dy && dx && C > 0 && C % 8 == 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && B > 0,
              "bilinear_slice_bwd: bad arguments");
  FFA_REQUIRE(y_pitch % 8 == 0 && y_off % 8 == 0 && y_off >= 0 && y_off + C <= y_pitch,
              "bilinear_slice_bwd: slice [%d, %d) does not fit pitch %d", y_off, y_off + C, y_pitch);
  float sy, sx;
  if (align_corners) {
    sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  } else {
    sy = (float)Hi / (float)Ho;
    sx = (float)Wi / (float)Wo;
  }
  const long long items = (long long)B * Hi * Wi * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(bilinear_slice_bwd_kernel<ffa_bf16>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const ffa_bf16*)dy, (ffa_bf16*)dx, B, Hi, Wi, Ho, Wo, C, y_pitch, y_off, align_corners ? 1 : 0,
                       sy, sx);
  else
    hipLaunchKernelGGL(bilinear_slice_bwd_kernel<float>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const float*)dy, (float*)dx, B, Hi, Wi, Ho, Wo, C, y_pitch, y_off, align_corners ? 1 : 0, sy,
                       sx);
  return ffa_check_launch("bilinear_slice_bwd");
}

// ---- backward of nn.AdaptiveAvgPool2d(S): dx[b][y][x] = sum over the output cells whose region holds (y, x) of
// dy[b][oy][ox] / area(oy, ox) (regions overlap when S does not divide the map)
template <typename T>
__global__ void adaptive_avg_pool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C,
                                             int S) {
  const int CG = C / 8;
  const long long total = (long long)B * H * W * CG;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % CG);
    long long p = i / CG;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const long long b = p / H;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int oy = 0; oy < S; ++oy) {
      const int y0 = (oy * H) / S, y1 = ((oy + 1) * H + S - 1) / S;
      if (y < y0 || y >= y1) continue;
      for (int ox = 0; ox < S; ++ox) {
        const int x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
        if (x < x0 || x >= x1) continue;
        const float inv = 1.0f / (float)((y1 - y0) * (x1 - x0));
        float gv[8];
        ffa_load8<T>(dy + ((b * S + oy) * S + ox) * C + g * 8, gv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += gv[e] * inv;
      }
    }
    ffa_store8<T>(dx + i * 8, acc);
  }
}

extern "C" int ffa_adaptive_avg_pool_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int S,
                                         hipStream_t stream) {
  FFA_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && S > 0, "adaptive_avg_pool_bwd: bad arguments");
  const long long items = (long long)B * H * W * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(adaptive_avg_pool_bwd_kernel<ffa_bf16>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const ffa_bf16*)dy, (ffa_bf16*)dx, B, H, W, C, S);
  else
    hipLaunchKernelGGL(adaptive_avg_pool_bwd_kernel<float>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const float*)dy, (float*)dx, B, H, W, C, S);
  return ffa_check_launch("adaptive_avg_pool_bwd");
}

// ---- out[c] = sum_m x[m][c] (f32): nn.Linear's bias gradient for any width (ffa_channel_sums stops at 2048 channels).
// Chunk partials in the workspace, summed in a fixed order: deterministic.
template <typename T>
__global__ void __launch_bounds__(256) column_sums_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                          long long rows, int C, long long rows_per_chunk) {
  __shared__ float red[8][32][8];
  const int CG = C / 8;
  const int gl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int g = blockIdx.x * 32 + gl;
  const long long r0 = blockIdx.y * rows_per_chunk;
  long long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (g < CG) {
    for (long long row = r0 + rl; row < r1; row += 8) {
      float v[8];
      ffa_load8<T>(x + row * (long long)C + g * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][gl][e] = acc[e];
  __syncthreads();
  if (rl == 0 && g < CG) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) v += red[r][gl][e];
      partial[(long long)blockIdx.y * C + g * 8 + e] = v;
    }
  }
}

// out[c] = sum_k partial[k][c]: 32 lanes per column take the chunks k = lane, lane + 32, ... and are combined by
// xor-shuffles -- a fixed summation order, 16 dependent loads instead of 512
__global__ void __launch_bounds__(256) column_sums_reduce_kernel(const float* __restrict__ partial,
                                                                 float* __restrict__ out, int C, int chunks, int pitch) {
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int l = threadIdx.x & 31;
  float a = 0.f;
  if (c < C)
    for (int k = l; k < chunks; k += 32) a += partial[(long long)k * pitch + c];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (c < C && l == 0) out[c] = a;
}

extern "C" long long ffa_column_sums_workspace_bytes(long long rows, int C) {
  return (long long)ln_bwd_chunks(rows) * C * (long long)sizeof(float);
}

extern "C" int ffa_column_sums(int dtype, const void* x, float* out, long long rows, int C, void* workspace,
                               long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(x && out && rows > 0 && C > 0 && C % 8 == 0, "column_sums: bad arguments");
  if (!workspace || workspace_bytes < ffa_column_sums_workspace_bytes(rows, C)) {
    ffa_set_error("column_sums: workspace of %lld bytes needed", ffa_column_sums_workspace_bytes(rows, C));
    return FFA_ERR_WORKSPACE;
  }
  const int chunks = ln_bwd_chunks(rows);
  const long long rpc = (rows + chunks - 1) / chunks;
  const dim3 grid((unsigned)((C / 8 + 31) / 32), (unsigned)chunks);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(column_sums_kernel<ffa_bf16>, grid, dim3(256), 0, stream, (const ffa_bf16*)x, (float*)workspace,
                       rows, C, rpc);
  else
    hipLaunchKernelGGL(column_sums_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, (float*)workspace, rows, C,
                       rpc);
  hipLaunchKernelGGL(column_sums_reduce_kernel, dim3((unsigned)((C + 7) / 8)), dim3(256), 0, stream,
                     (const float*)workspace, out, C, chunks, C);
  return ffa_check_launch("column_sums");
}

// ------------------------------------------------------------------------------------------------
// Window attention backward (bf16, MFMA 16x16x32).  One block per (window, head); q, k, v and dO of the window's
// tokens are staged once in LDS (row-major [NP][32] at a 96-byte pitch: row fragments by ds_read_b128, transposed
// fragments by ds_read_b64_tr_b16).  Two passes over the recomputed probabilities, as in flash-attention's backward:
//   pass A, lanes = queries:  S^T = K Q^T, row max / sum, dP^T = V dO^T, delta = sum_k P dP, dS = P (dP - delta),
//           d table[idx(q, k)] += dS (LDS atomics, flushed once per block), dQ^T = scale K^T dS^T;
//           row max, 1 / sum and delta go to LDS for pass B
//   pass B, lanes = keys:     S = Q K^T again in the other orientation, P, dP = dO V^T, dS,
//           dV^T += dO^T P, dK^T += scale Q^T dS   (the query index is the MFMA's k dimension, taken from registers)
// Every real token belongs to exactly one window, so dqkv is written without atomics; the gradients of padding tokens
// (whose q / k / v are the qkv bias) go to dbias_pad[3C] with atomics, like the bias-table gradient (nn.Parameter
// indexing has an atomic backward in torch as well).
template <int NTP, int NW>
__global__ void __launch_bounds__(64 * NW) window_attention_bwd_kernel(WinAttnArgs a, const ffa_bf16* __restrict__ dout,
                                                                       ffa_bf16* __restrict__ dqkv,
                                                                       float* __restrict__ table_partial, int table_pitch,
                                                                       float* __restrict__ pad_partial) {
  constexpr int NP = NTP * 16;
  constexpr int RP = 96;  // row pitch in bytes
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * NP * RP + NP * 16 + 2 * 529 * 4 + 96 * 4];
  unsigned char* sq = smem;
  unsigned char* sk = sq + NP * RP;
  unsigned char* sv = sk + NP * RP;
  unsigned char* sdo = sv + NP * RP;
  int* skey = reinterpret_cast<int*>(sdo + NP * RP);          // [NP] lin | rid << 16
  float* sm = reinterpret_cast<float*>(skey + NP);            // [NP] row max (log2 units)
  float* sli = sm + NP;                                       // [NP] 1 / row sum
  float* sdelta = sli + NP;                                   // [NP]
  float* stab = sdelta + NP;                                  // [529]
  float* sdtab = stab + 529;                                  // [529]
  float* spad = sdtab + 529;                                  // [3][32] gradients of the window's padding tokens
  const int N = a.ws * a.ws;
  const int head = blockIdx.y;
  int w = blockIdx.x;
  const int wx = w % a.nwx;
  w /= a.nwx;
  const int wy = w % a.nwy;
  const int b = w / a.nwy;
  const ffa_bf16* qkv = (const ffa_bf16*)a.qkv;
  const int C3 = 3 * a.C;
  const int TS = (2 * a.ws - 1) * (2 * a.ws - 1);
  for (int i = threadIdx.x; i < TS; i += 64 * NW) {
    stab[i] = a.table[i * a.heads + head] * LOG2E;
    sdtab[i] = 0.f;
  }
  if (threadIdx.x < 96) spad[threadIdx.x] = 0.f;
  const float scale2 = a.scale * LOG2E;
  const bool masked = a.shift > 0 && (wy == a.nwy - 1 || wx == a.nwx - 1);
  for (int i = threadIdx.x; i < NP * 4; i += 64 * NW) {
    const int t = i >> 2, pc = i & 3;
    const int c = head * 32 + pc * 8;
    ffa_u32x4 q4 = {0u, 0u, 0u, 0u}, k4 = q4, v4 = q4, d4 = q4;
    int packed = 0xffff0000;
    if (t < N) {
      const WinTok tk = win_token(a, b, wy, wx, t);
      packed = tk.lin | (tk.rid << 16);
      if (tk.off >= 0) {
        const ffa_bf16* p = qkv + tk.off * C3 + c;
        q4 = *reinterpret_cast<const ffa_u32x4*>(p);
        k4 = *reinterpret_cast<const ffa_u32x4*>(p + a.C);
        v4 = *reinterpret_cast<const ffa_u32x4*>(p + 2 * a.C);
        d4 = *reinterpret_cast<const ffa_u32x4*>(dout + tk.off * a.C + c);
      } else {
        float qf[8], kf[8], vf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          qf[e] = a.qkv_bias[c + e];
          kf[e] = a.qkv_bias[a.C + c + e];
          vf[e] = a.qkv_bias[2 * a.C + c + e];
        }
        ffa_store8<ffa_bf16>(reinterpret_cast<ffa_bf16*>(&q4), qf);
        ffa_store8<ffa_bf16>(reinterpret_cast<ffa_bf16*>(&k4), kf);
        ffa_store8<ffa_bf16>(reinterpret_cast<ffa_bf16*>(&v4), vf);
      }
    }
    *reinterpret_cast<ffa_u32x4*>(sq + t * RP + pc * 16) = q4;
    *reinterpret_cast<ffa_u32x4*>(sk + t * RP + pc * 16) = k4;
    *reinterpret_cast<ffa_u32x4*>(sv + t * RP + pc * 16) = v4;
    *reinterpret_cast<ffa_u32x4*>(sdo + t * RP + pc * 16) = d4;
    if (pc == 0) skey[t] = packed;
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const int off0 = (a.ws - 1) * (2 * a.ws - 1) + (a.ws - 1);
  const int nqt = (N + 15) / 16;
  const int tr_off = ((lane >> 2) & 3) * RP + (lane & 3) * 8;
  auto tr_frag = [&](const unsigned char* img, int u, int dhalf) -> ffa_bf16x8 {
    // transposed 16 x 32 fragment: rows (tokens) 32u + 4g .. +3 and 32u + 16 + 4g .. +3, channels 16 dhalf + lane % 16
    const unsigned char* p = img + (u * 32 + g * 4) * RP + tr_off + dhalf * 32;
    const ffa_s16x4 lo = attn_read_tr16(p);
    const ffa_s16x4 hi = attn_read_tr16(p + 16 * RP);
    ffa_u32x4 v;
    v.x = __builtin_bit_cast(ffa_u32x2, lo).x; v.y = __builtin_bit_cast(ffa_u32x2, lo).y;
    v.z = __builtin_bit_cast(ffa_u32x2, hi).x; v.w = __builtin_bit_cast(ffa_u32x2, hi).y;
    return __builtin_bit_cast(ffa_bf16x8, v);
  };
  // token -> (element offset or -1) for the output stores of this lane's token (tile * 16 + n)
  auto token_off = [&](int t) -> long long { return t < N ? win_token(a, b, wy, wx, t).off : -2; };
  auto store_grad = [&](int t, int which, const ffa_f32x4& lo, const ffa_f32x4& hi, float mul) {
    const long long off = token_off(t);
    if (off >= 0) {
      ffa_bf16* dst = dqkv + off * C3 + which * a.C + head * 32 + g * 4;
      uint2 l2, h2;
      l2.x = ffa_pack_bf16x2(lo[0] * mul, lo[1] * mul);
      l2.y = ffa_pack_bf16x2(lo[2] * mul, lo[3] * mul);
      h2.x = ffa_pack_bf16x2(hi[0] * mul, hi[1] * mul);
      h2.y = ffa_pack_bf16x2(hi[2] * mul, hi[3] * mul);
      *reinterpret_cast<uint2*>(dst) = l2;
      *reinterpret_cast<uint2*>(dst + 16) = h2;
    } else if (off == -1) {  // padding token inside the window: its q / k / v are the bias
      float* dst = spad + which * 32 + g * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsafeAtomicAdd(dst + e, lo[e] * mul);       // ds_add_f32
        unsafeAtomicAdd(dst + 16 + e, hi[e] * mul);
      }
    }
  };

  // ---------------- pass A: lanes = queries
  for (int qt = wave; qt < nqt; qt += NW) {
    const int qi = qt * 16 + n;
    const ffa_bf16x8 qf = *reinterpret_cast<const ffa_bf16x8*>(sq + qi * RP + g * 16);
    const ffa_bf16x8 dof = *reinterpret_cast<const ffa_bf16x8*>(sdo + qi * RP + g * 16);
    const int qk = skey[qi];
    const int qlin = off0 + (qk & 0xffff), qrid = qk >> 16;
    ffa_f32x4 s[NTP], dp[NTP];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      if (t >= nqt) {
        s[t] = ffa_f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        dp[t] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};
        continue;
      }
      const ffa_bf16x8 kf = *reinterpret_cast<const ffa_bf16x8*>(sk + (t * 16 + n) * RP + g * 16);
      const ffa_bf16x8 vf = *reinterpret_cast<const ffa_bf16x8*>(sv + (t * 16 + n) * RP + g * 16);
      const int4 key4 = *reinterpret_cast<const int4*>(&skey[t * 16 + g * 4]);
      ffa_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, acc, 0, 0, 0);
      acd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof, acd, 0, 0, 0);
      const int kk[4] = {key4.x, key4.y, key4.z, key4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[i] * scale2 + stab[qlin - (kk[i] & 0xffff)];
        if (masked && (kk[i] >> 16) != qrid) v += -100.0f * LOG2E;
        if (kk[i] < 0) v = -INFINITY;
        acc[i] = v;
        mx = fmaxf(mx, v);
      }
      s[t] = acc;
      dp[t] = acd;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float e = __builtin_amdgcn_exp2f(s[t][i] - mx);
        s[t][i] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[t][i] *= inv;
        delta += s[t][i] * dp[t][i];
      }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (g == 0) {
      sm[qi] = mx;
      sli[qi] = inv;
      sdelta[qi] = delta;
    }
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      if (t >= nqt) continue;
      const int4 key4 = *reinterpret_cast<const int4*>(&skey[t * 16 + g * 4]);
      const int kk[4] = {key4.x, key4.y, key4.z, key4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ds = s[t][i] * (dp[t][i] - delta);
        s[t][i] = ds;
        if (ds != 0.f) unsafeAtomicAdd(&sdtab[qlin - (kk[i] & 0xffff)], ds);  // ds_add_f32
      }
    }
    ffa_f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NTP / 2; ++u) {
      ffa_bf16x8 df;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        df[i] = (__bf16)s[2 * u][i];
        df[4 + i] = (__bf16)s[2 * u + 1][i];
      }
      q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sk, u, 0), df, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sk, u, 1), df, q1, 0, 0, 0);
    }
    store_grad(qi, 0, q0, q1, a.scale);
  }
  __syncthreads();

  // ---------------- pass B: lanes = keys
  for (int kt = wave; kt < nqt; kt += NW) {
    const int ki = kt * 16 + n;
    const ffa_bf16x8 kfb = *reinterpret_cast<const ffa_bf16x8*>(sk + ki * RP + g * 16);
    const ffa_bf16x8 vfb = *reinterpret_cast<const ffa_bf16x8*>(sv + ki * RP + g * 16);
    const int kk = skey[ki];
    const int klin = kk & 0xffff, krid = kk >> 16;
    ffa_f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
#pragma unroll
    for (int u = 0; u < NTP / 2; ++u) {
      ffa_bf16x8 pf, df;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t = 2 * u + h;
        ffa_f32x4 p = {0.f, 0.f, 0.f, 0.f}, ds = p;
        if (t < nqt) {  // block-uniform
          const ffa_bf16x8 qa = *reinterpret_cast<const ffa_bf16x8*>(sq + (t * 16 + n) * RP + g * 16);
          const ffa_bf16x8 da = *reinterpret_cast<const ffa_bf16x8*>(sdo + (t * 16 + n) * RP + g * 16);
          const int4 q4 = *reinterpret_cast<const int4*>(&skey[t * 16 + g * 4]);
          const float4 m4 = *reinterpret_cast<const float4*>(&sm[t * 16 + g * 4]);
          const float4 l4 = *reinterpret_cast<const float4*>(&sli[t * 16 + g * 4]);
          const float4 d4 = *reinterpret_cast<const float4*>(&sdelta[t * 16 + g * 4]);
          ffa_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = acc;
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kfb, acc, 0, 0, 0);   // rows: queries 4g + i, column: key n
          acd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vfb, acd, 0, 0, 0);
          const int qq[4] = {q4.x, q4.y, q4.z, q4.w};
          const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
          const float ll[4] = {l4.x, l4.y, l4.z, l4.w};
          const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = acc[i] * scale2 + stab[off0 + (qq[i] & 0xffff) - klin];
            if (masked && (qq[i] >> 16) != krid) v += -100.0f * LOG2E;
            float pr = __builtin_amdgcn_exp2f(v - mm[i]) * ll[i];
            if (krid < 0) pr = 0.f;
            p[i] = pr;
            ds[i] = pr * (acd[i] - dd[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pf[4 * h + i] = (__bf16)p[i];
          df[4 * h + i] = (__bf16)ds[i];
        }
      }
      v0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sdo, u, 0), pf, v0, 0, 0, 0);
      v1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sdo, u, 1), pf, v1, 0, 0, 0);
      k0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sq, u, 0), df, k0, 0, 0, 0);
      k1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sq, u, 1), df, k1, 0, 0, 0);
    }
    store_grad(ki, 1, k0, k1, a.scale);
    store_grad(ki, 2, v0, v1, 1.0f);
  }
  __syncthreads();
  // this block's share of the bias-table gradient and of the padding tokens' gradient: one row per window, the
  // columns of this head; the rows are summed by column_sums_kernel afterwards (no global atomics, fixed order)
  float* trow = table_partial + (long long)blockIdx.x * table_pitch;
  for (int i = threadIdx.x; i < TS; i += 64 * NW) trow[i * a.heads + head] = sdtab[i];
  if (head == 0)
    for (int i = TS * a.heads + threadIdx.x; i < table_pitch; i += 64 * NW) trow[i] = 0.f;
  if (pad_partial) {
    const bool last_row = wy == a.nwy - 1, last_col = wx == a.nwx - 1;
    if (last_row || last_col) {  // the only windows that can hold padding tokens
      const int slot = b * (a.nwx + a.nwy - 1) + (last_row ? wx : a.nwx + wy);
      if (threadIdx.x < 96)
        pad_partial[(long long)slot * (3 * a.C) + (threadIdx.x >> 5) * a.C + head * 32 + (threadIdx.x & 31)] =
            spad[threadIdx.x];
    }
  }
}

// ---- f32 parity path of the backward: one block per (window, head), plain FMA, the two passes of the bf16 kernel
// (pass A: one wave per query row -> row max, 1 / sum, delta, dQ, the bias-table gradient; pass B: one wave per key
// column, probabilities recomputed from the saved row statistics -> dK, dV).  Same outputs and partial-row layout.
__global__ void __launch_bounds__(256) window_attention_bwd_f32_kernel(WinAttnArgs a, const float* __restrict__ dout,
                                                                       float* __restrict__ dqkv,
                                                                       float* __restrict__ table_partial, int table_pitch,
                                                                       float* __restrict__ pad_partial) {
  constexpr int NMAX = 144;
  __shared__ float sq[NMAX * 33], sk[NMAX * 33], sv[NMAX * 33], sdo[NMAX * 33];
  __shared__ float sp[4][NMAX], sds[4][NMAX];
  __shared__ float sm[NMAX], sli[NMAX], sdelta[NMAX];
  __shared__ float stab[529], sdtab[529], spad[96];
  __shared__ int slin[NMAX], srid[NMAX];
  __shared__ long long soff[NMAX];
  const int N = a.ws * a.ws;
  const int head = blockIdx.y;
  int w = blockIdx.x;
  const int wx = w % a.nwx;
  w /= a.nwx;
  const int wy = w % a.nwy;
  const int b = w / a.nwy;
  const float* qkv = (const float*)a.qkv;
  const int C3 = 3 * a.C;
  const int TS = (2 * a.ws - 1) * (2 * a.ws - 1);
  for (int i = threadIdx.x; i < N; i += 256) {
    const WinTok tk = win_token(a, b, wy, wx, i);
    slin[i] = tk.lin;
    srid[i] = tk.rid;
    soff[i] = tk.off;
  }
  for (int i = threadIdx.x; i < TS; i += 256) {
    stab[i] = a.table[i * a.heads + head];
    sdtab[i] = 0.f;
  }
  if (threadIdx.x < 96) spad[threadIdx.x] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < N * 32; i += 256) {
    const int t = i >> 5, d = i & 31;
    const long long off = soff[t];
    const int c = head * 32 + d;
    float q, k, v, g = 0.f;
    if (off >= 0) {
      const float* p = qkv + off * C3;
      q = p[c];
      k = p[a.C + c];
      v = p[2 * a.C + c];
      g = dout[off * a.C + c];
    } else {
      q = a.qkv_bias[c];
      k = a.qkv_bias[a.C + c];
      v = a.qkv_bias[2 * a.C + c];
    }
    sq[t * 33 + d] = q;
    sk[t * 33 + d] = k;
    sv[t * 33 + d] = v;
    sdo[t * 33 + d] = g;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int off0 = (a.ws - 1) * (2 * a.ws - 1) + (a.ws - 1);
  const int d = lane & 31, half = lane >> 5;
  float* p = sp[wave];
  float* ds = sds[wave];
  // pass A: queries
  for (int i = wave; i < N; i += 4) {
    if (soff[i] < 0) {  // a padding query has no output row: it contributes nothing (wave-uniform)
      if (lane == 0) {
        sm[i] = 0.f;
        sli[i] = 0.f;
        sdelta[i] = 0.f;
      }
      continue;
    }
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
      float sdot = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) sdot += sq[i * 33 + e] * sk[j * 33 + e];
      float sv_ = sdot * a.scale + stab[slin[i] - slin[j] + off0];
      if (srid[i] != srid[j]) sv_ += -100.0f;
      p[j] = sv_;
      mx = fmaxf(mx, sv_);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float e = expf(p[j] - mx);
      p[j] = e;
      sum += e;
    }
    sum = ffa_wave_sum(sum);
    const float inv = 1.0f / sum;
    float delta = 0.f;
    for (int j = lane; j < N; j += 64) {
      float dp = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) dp += sdo[i * 33 + e] * sv[j * 33 + e];
      const float pr = p[j] * inv;
      p[j] = pr;
      ds[j] = dp;
      delta += pr * dp;
    }
    delta = ffa_wave_sum(delta);
    for (int j = lane; j < N; j += 64) {
      const float g = p[j] * (ds[j] - delta);
      ds[j] = g;
      unsafeAtomicAdd(&sdtab[slin[i] - slin[j] + off0], g);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float acc = 0.f;
    for (int j = half; j < N; j += 2) acc += ds[j] * sk[j * 33 + d];
    acc += __shfl_xor(acc, 32, 64);
    if (half == 0) dqkv[soff[i] * C3 + head * 32 + d] = acc * a.scale;
    if (lane == 0) {
      sm[i] = mx;
      sli[i] = inv;
      sdelta[i] = delta;
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // pass B: keys
  for (int j = wave; j < N; j += 4) {
    for (int i = lane; i < N; i += 64) {
      float pr = 0.f, g = 0.f;
      if (sli[i] != 0.f) {
        float sdot = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          sdot += sq[i * 33 + e] * sk[j * 33 + e];
          dp += sdo[i * 33 + e] * sv[j * 33 + e];
        }
        float sv_ = sdot * a.scale + stab[slin[i] - slin[j] + off0];
        if (srid[i] != srid[j]) sv_ += -100.0f;
        pr = expf(sv_ - sm[i]) * sli[i];
        g = pr * (dp - sdelta[i]);
      }
      p[i] = pr;
      ds[i] = g;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float dv = 0.f, dk = 0.f;
    for (int i = half; i < N; i += 2) {
      dv += p[i] * sdo[i * 33 + d];
      dk += ds[i] * sq[i * 33 + d];
    }
    dv += __shfl_xor(dv, 32, 64);
    dk += __shfl_xor(dk, 32, 64);
    dk *= a.scale;
    if (half == 0) {
      const long long off = soff[j];
      if (off >= 0) {
        dqkv[off * C3 + a.C + head * 32 + d] = dk;
        dqkv[off * C3 + 2 * a.C + head * 32 + d] = dv;
      } else {
        unsafeAtomicAdd(&spad[32 + d], dk);
        unsafeAtomicAdd(&spad[64 + d], dv);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  float* trow = table_partial + (long long)blockIdx.x * table_pitch;
  for (int i = threadIdx.x; i < TS; i += 256) trow[i * a.heads + head] = sdtab[i];
  if (head == 0)
    for (int i = TS * a.heads + threadIdx.x; i < table_pitch; i += 256) trow[i] = 0.f;
  if (pad_partial) {
    const bool last_row = wy == a.nwy - 1, last_col = wx == a.nwx - 1;
    if (last_row || last_col) {
      const int slot = b * (a.nwx + a.nwy - 1) + (last_row ? wx : a.nwx + wy);
      if (threadIdx.x < 96)
        pad_partial[(long long)slot * (3 * a.C) + (threadIdx.x >> 5) * a.C + head * 32 + (threadIdx.x & 31)] =
            spad[threadIdx.x];
    }
  }
}

static inline long long attn_bwd_table_pitch(int heads, int ws) {
  const long long n = (long long)(2 * ws - 1) * (2 * ws - 1) * heads;
  return (n + 7) / 8 * 8;
}

extern "C" long long ffa_window_attention_bwd_workspace_bytes(int B, int H, int W, int C, int heads, int ws) {
  const int nwy = (H + ws - 1) / ws, nwx = (W + ws - 1) / ws;
  const long long nwin = (long long)B * nwy * nwx;
  const long long pitch = attn_bwd_table_pitch(heads, ws);
  const long long pad_rows = (H % ws || W % ws) ? (long long)B * (nwx + nwy - 1) : 0;
  const long long red = ln_bwd_chunks(nwin) * (pitch > 3LL * C ? pitch : 3LL * C);
  return (nwin * pitch + pad_rows * 3 * C + red) * (long long)sizeof(float);
}

__global__ void zero_f32_kernel(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

extern "C" int ffa_window_attention_bwd(int dtype, const void* qkv, const void* dout, void* dqkv, const float* qkv_bias,
                                        const float* table, float* dtable, float* dbias_pad, int B, int H, int W, int C,
                                        int heads, int ws, int shift, float scale, void* workspace,
                                        long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "window_attention_bwd: dtype %d", dtype);
  FFA_REQUIRE(qkv && dout && dqkv && qkv_bias && table && dtable && dbias_pad && B > 0 && H > 0 && W > 0 && heads > 0,
              "window_attention_bwd: bad arguments");
  FFA_REQUIRE(C == heads * 32, "window_attention_bwd: head dimension %d (only 32 is built)", heads ? C / heads : 0);
  FFA_REQUIRE(ws >= 1 && ws <= 12 && shift >= 0 && shift < ws, "window_attention_bwd: window %d / shift %d", ws, shift);
  if (!workspace || workspace_bytes < ffa_window_attention_bwd_workspace_bytes(B, H, W, C, heads, ws)) {
    ffa_set_error("window_attention_bwd: workspace of %lld bytes needed",
                  ffa_window_attention_bwd_workspace_bytes(B, H, W, C, heads, ws));
    return FFA_ERR_WORKSPACE;
  }
  WinAttnArgs a;
  a.qkv = qkv;
  a.out = nullptr;
  a.qkv_bias = qkv_bias;
  a.table = table;
  a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.ws = ws; a.shift = shift;
  a.nwy = (H + ws - 1) / ws;
  a.nwx = (W + ws - 1) / ws;
  a.scale = scale;
  const long long nwin = (long long)B * a.nwy * a.nwx;
  FFA_REQUIRE(nwin < (1LL << 31) && heads < 65536, "window_attention_bwd: grid too large");
  const int pitch = (int)attn_bwd_table_pitch(heads, ws);
  const long long pad_rows = (H % ws || W % ws) ? (long long)B * (a.nwx + a.nwy - 1) : 0;
  float* table_partial = (float*)workspace;
  float* pad_partial = pad_rows ? table_partial + nwin * pitch : nullptr;
  float* red = table_partial + nwin * pitch + pad_rows * 3 * C;
  const dim3 grid((unsigned)nwin, (unsigned)heads);
  if (dtype == FFA_F32)
    hipLaunchKernelGGL(window_attention_bwd_f32_kernel, grid, dim3(256), 0, stream, a, (const float*)dout, (float*)dqkv,
                       table_partial, pitch, pad_partial);
  else if (ws * ws <= 64)
    hipLaunchKernelGGL((window_attention_bwd_kernel<4, 4>), grid, dim3(256), 0, stream, a, (const ffa_bf16*)dout,
                       (ffa_bf16*)dqkv, table_partial, pitch, pad_partial);
  else
    hipLaunchKernelGGL((window_attention_bwd_kernel<10, 3>), grid, dim3(192), 0, stream, a, (const ffa_bf16*)dout,
                       (ffa_bf16*)dqkv, table_partial, pitch, pad_partial);
  // fixed-order sums over the windows: d table [TS][heads] and the padding tokens' share of d qkv_bias [3C]
  {
    const int chunks = ln_bwd_chunks(nwin);
    const long long rpc = (nwin + chunks - 1) / chunks;
    hipLaunchKernelGGL(column_sums_kernel<float>, dim3((unsigned)((pitch / 8 + 31) / 32), (unsigned)chunks), dim3(256), 0,
                       stream, (const float*)table_partial, red, nwin, pitch, rpc);
    const int TSH = (2 * ws - 1) * (2 * ws - 1) * heads;
    hipLaunchKernelGGL(column_sums_reduce_kernel, dim3((unsigned)((TSH + 7) / 8)), dim3(256), 0, stream,
                       (const float*)red, dtable, TSH, chunks, pitch);
  }
  if (pad_rows) {
    const int chunks = ln_bwd_chunks(pad_rows);
    const long long rpc = (pad_rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(column_sums_kernel<float>, dim3((unsigned)((3 * C / 8 + 31) / 32), (unsigned)chunks), dim3(256), 0,
                       stream, (const float*)pad_partial, red, pad_rows, 3 * C, rpc);
    hipLaunchKernelGGL(column_sums_reduce_kernel, dim3((unsigned)((3 * C + 7) / 8)), dim3(256), 0, stream,
                       (const float*)red, dbias_pad, 3 * C, chunks, 3 * C);
  } else {
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((3 * C + 255) / 256)), dim3(256), 0, stream, dbias_pad, 3 * C);
  }
  return ffa_check_launch("window_attention_bwd");
}


// ---- y[m][c] = x[m][c] * row_scale[m / rows_per_scale]: the DropPath factor on a gradient before the weight-gradient and
// bias-gradient reductions (the GEMM epilogues apply it everywhere else)
template <typename T>
__global__ void scale_rows_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ row_scale,
                                  long long rows, int C, int rows_per_scale) {
  const int CG = C / 8;
  const long long total = rows * CG;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const float sc = row_scale[(i / CG) / rows_per_scale];
    float v[8];
    ffa_load8<T>(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sc;
    ffa_store8<T>(y + i * 8, v);
  }
}

extern "C" int ffa_scale_rows(int dtype, const void* x, void* y, const float* row_scale, long long rows, int C,
                              int rows_per_scale, hipStream_t stream) {
  FFA_REQUIRE(x && y && row_scale && rows > 0 && C > 0 && C % 8 == 0 && rows_per_scale > 0, "scale_rows: bad arguments");
  const long long items = rows * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(scale_rows_kernel<ffa_bf16>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const ffa_bf16*)x, (ffa_bf16*)y, row_scale, rows, C, rows_per_scale);
  else
    hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream, (const float*)x,
                       (float*)y, row_scale, rows, C, rows_per_scale);
  return ffa_check_launch("scale_rows");
}

// ---- F.interpolate(F.interpolate(x, 2x, bilinear), 1/2, bilinear) in one pass.  UPerNet's last FPN stage (the 0-channel
// placeholder feature of a transformer encoder) upsamples the pyramid map to stride 2 without adding anything, and the
// decoder then resizes it back to stride 4: the composition of the two align_corners=False resizes is the separable
// 3-tap filter [1/8, 3/4, 1/8] with replicated edges (up[2i] = x[i-1]/4 + 3x[i]/4, up[2i+1] = 3x[i]/4 + x[i+1]/4,
// down[i] = (up[2i] + up[2i+1]) / 2), a symmetric operator -- so its backward is the same kernel on the gradient.
// Neither the stride-2 map (1 GB at batch 32) nor its gradient is ever written.  Input and output may both be channel
// slices of wider tensors.
template <typename T>
__global__ void blur3_slice_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int x_pitch,
                                   int x_off, int y_pitch, int y_off) {
  const int CG = C / 8;
  const long long total = (long long)B * H * W * CG;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % CG);
    long long p = i / CG;
    const long long pix = p;
    const int xx = (int)(p % W);
    p /= W;
    const int yy = (int)(p % H);
    const long long b = p / H;
    const int ys[3] = {yy > 0 ? yy - 1 : 0, yy, yy < H - 1 ? yy + 1 : H - 1};
    const int xs[3] = {xx > 0 ? xx - 1 : 0, xx, xx < W - 1 ? xx + 1 : W - 1};
    const float wt[3] = {0.125f, 0.75f, 0.125f};
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float row[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) row[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v[8];
        ffa_load8<T>(x + ((b * H + ys[a]) * W + xs[c]) * x_pitch + x_off + g * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) row[e] += wt[c] * v[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += wt[a] * row[e];
    }
    ffa_store8<T>(y + pix * y_pitch + y_off + g * 8, acc);
  }
}

extern "C" int ffa_updown2x_slice(int dtype, const void* x, void* y, int B, int H, int W, int C, int x_pitch, int x_off,
                                  int y_pitch, int y_off, hipStream_t stream) {
  FFA_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "updown2x_slice: bad arguments");
  FFA_REQUIRE(x_pitch % 8 == 0 && x_off % 8 == 0 && x_off >= 0 && x_off + C <= x_pitch && y_pitch % 8 == 0 &&
                  y_off % 8 == 0 && y_off >= 0 && y_off + C <= y_pitch, "updown2x_slice: slices do not fit their pitches");
  const long long items = (long long)B * H * W * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(blur3_slice_kernel<ffa_bf16>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const ffa_bf16*)x, (ffa_bf16*)y, B, H, W, C, x_pitch, x_off, y_pitch, y_off);
  else
    hipLaunchKernelGGL(blur3_slice_kernel<float>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream, (const float*)x,
                       (float*)y, B, H, W, C, x_pitch, x_off, y_pitch, y_off);
  return ffa_check_launch("updown2x_slice");
}
