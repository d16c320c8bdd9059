// Kernels of the U-TAE Sentinel time-series branch, NHWC, bf16 or f32 storage, f32 math.
//
// Replaces the torch ops the reference's UTAE reaches in flair_hub/models/multitemp_model.py:
//   reflect padding of nn.Conv2d(padding_mode="reflect") :473-482        -> ffa_reflect_pad1 (+ ffa_conv2d, pad 0)
//   nn.GroupNorm(4) of ConvLayer :464-468, nn.GroupNorm(16) of LTAE2d :224-231,256,276 -> ffa_group_norm
//   PositionalEncoder :287-313                                           -> ffa_positional_encoding
//   "out + positional_encoder(bp)" :270                                  -> ffa_add_rowvec
//   MultiHeadAttention + ScaledDotProductAttention :337-403              -> ffa_ltae_attention
//   Temporal_Aggregator(mode="att_group") :609-628,640-654               -> ffa_temporal_aggregate
//   TemporallySharedBlock.smart_forward's pad-date handling :432-443     -> ffa_mask_images
// The tensors are tiny (Sentinel patches are ~10 x 10 pixels x T dates): these kernels are written for
// correctness and launch count, none of them is on the bandwidth or MFMA roofline.
#include "ffa_common.h"

#include <math.h>

#define FFA_T_THREADS 256

// ------------------------------------------------------------------------------------------------
// reflect padding by one pixel: out[n][y][x][:] = in[n][refl(y-1)][refl(x-1)][:],  refl(-1) = 1, refl(H) = H-2

template <typename T>
__global__ void reflect_pad1_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int C8) {
  const long long total = (long long)N * (H + 2) * (W + 2) * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    long long p = i / C8;
    const int x = (int)(p % (W + 2));
    p /= (W + 2);
    const int y = (int)(p % (H + 2));
    const long long n = p / (H + 2);
    int sy = y - 1, sx = x - 1;
    sy = sy < 0 ? -sy : (sy >= H ? 2 * H - 2 - sy : sy);
    sx = sx < 0 ? -sx : (sx >= W ? 2 * W - 2 - sx : sx);
    float v[8];
    ffa_load8<T>(in + (((n * H + sy) * W + sx) * C8 + c) * 8, v);
    ffa_store8<T>(out + i * 8, v);
  }
}

extern "C" int ffa_reflect_pad1(int dtype, const void* in, void* out, int N, int H, int W, int C, hipStream_t stream) {
  FFA_REQUIRE(in && out && N > 0 && H >= 2 && W >= 2 && C % 8 == 0, "reflect_pad1: bad arguments (H, W >= 2, C % 8 == 0)");
  const long long total = (long long)N * (H + 2) * (W + 2) * (C / 8);
  const int grid = (int)((total + FFA_T_THREADS - 1) / FFA_T_THREADS < 4096 ? (total + FFA_T_THREADS - 1) / FFA_T_THREADS : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(reflect_pad1_kernel<ffa_bf16>, dim3(grid), dim3(FFA_T_THREADS), 0, stream, (const ffa_bf16*)in,
                       (ffa_bf16*)out, N, H, W, C / 8);
  else
    hipLaunchKernelGGL(reflect_pad1_kernel<float>, dim3(grid), dim3(FFA_T_THREADS), 0, stream, (const float*)in,
                       (float*)out, N, H, W, C / 8);
  return ffa_check_launch("reflect_pad1");
}

// ------------------------------------------------------------------------------------------------
// GroupNorm over strided samples.  Sample s (0 .. S-1) starts at element (s / Q) * stride_hi + (s % Q) * stride_lo and
// has `inner` positions `inner_stride` elements apart, C channels each (C contiguous); statistics per (sample, group)
// over inner x C/G values in double-free two-pass f32 (mean first, then centred squares: torch's CPU kernel does the
// same in double).   y = [residual +] relu?((x - mean) * rstd * gamma[c] + beta[c])
//   conv feature maps [N][H][W][C]:         S = N, Q = 1, stride_hi = H*W*C, inner = H*W, inner_stride = C
//   per-pixel sequences [B][T][h][w][C]:    S = B*h*w, Q = h*w, stride_hi = T*h*w*C, stride_lo = C, inner = T,
//                                           inner_stride = h*w*C

template <typename T>
__global__ void __launch_bounds__(FFA_T_THREADS) group_norm_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                                  T* __restrict__ y, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int Q,
                                                                  long long stride_hi, long long stride_lo, int inner,
                                                                  long long inner_stride, int C, int G, float eps,
                                                                  int relu) {
  const int s = blockIdx.x / G, g = blockIdx.x % G;
  const int cg = C / G;
  const long long base = (long long)(s / Q) * stride_hi + (long long)(s % Q) * stride_lo + (long long)g * cg;
  const int n = inner * cg;
  __shared__ float red[FFA_T_THREADS / 64];
  __shared__ float bc[2];
  auto block_sum = [&](float v) {
    v = ffa_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < FFA_T_THREADS / 64; ++w) t += red[w];
    return t;
  };
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += FFA_T_THREADS)
    a += ffa_load_elem<T>(x + base + (long long)(i / cg) * inner_stride + (i % cg));
  const float mean = block_sum(a) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += FFA_T_THREADS) {
    const float d = ffa_load_elem<T>(x + base + (long long)(i / cg) * inner_stride + (i % cg)) - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(block_sum(q) / (float)n + eps);
  for (int i = threadIdx.x; i < n; i += FFA_T_THREADS) {
    const int c = g * cg + (i % cg);
    const long long off = base + (long long)(i / cg) * inner_stride + (i % cg);
    float v = (ffa_load_elem<T>(x + off) - mean) * rstd * gamma[c] + beta[c];
    if (relu) v = fmaxf(v, 0.f);
    if (res) v += ffa_load_elem<T>(res + off);
    ffa_store_elem<T>(y + off, v);
  }
}

extern "C" int ffa_group_norm(int dtype, const void* x, const void* residual, void* y, const float* gamma,
                              const float* beta, long long samples, int Q, long long stride_hi, long long stride_lo,
                              int inner, long long inner_stride, int C, int groups, float eps, int relu,
                              hipStream_t stream) {
  FFA_REQUIRE(x && y && gamma && beta, "group_norm: null pointer");
  FFA_REQUIRE(samples > 0 && Q > 0 && inner > 0 && C > 0 && groups > 0 && C % groups == 0 &&
                  samples * groups < (1LL << 31),
              "group_norm: bad geometry");
  const int grid = (int)(samples * groups);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(group_norm_kernel<ffa_bf16>, dim3(grid), dim3(FFA_T_THREADS), 0, stream, (const ffa_bf16*)x,
                       (const ffa_bf16*)residual, (ffa_bf16*)y, gamma, beta, Q, stride_hi, stride_lo, inner,
                       inner_stride, C, groups, eps, relu);
  else
    hipLaunchKernelGGL(group_norm_kernel<float>, dim3(grid), dim3(FFA_T_THREADS), 0, stream, (const float*)x,
                       (const float*)residual, (float*)y, gamma, beta, Q, stride_hi, stride_lo, inner, inner_stride, C,
                       groups, eps, relu);
  return ffa_check_launch("group_norm");
}

// ------------------------------------------------------------------------------------------------
// sinusoidal date encoding: out[n][r*d + j] = sin / cos (pos[n] / T^(2*(j/2)/d)),  even j -> sin, odd j -> cos,
// repeated `repeat` times along the channel axis

__global__ void positional_encoding_kernel(const float* __restrict__ pos, float* __restrict__ out, int n, int d,
                                           int repeat, float period) {
  const int total = n * d * repeat;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i % d;
    const int row = i / (d * repeat);
    const float denom = powf(period, 2.0f * (float)(j / 2) / (float)d);
    const float a = pos[row] / denom;
    out[i] = (j & 1) ? cosf(a) : sinf(a);
  }
}

extern "C" int ffa_positional_encoding(const float* pos, float* out, int n, int d, int repeat, float period,
                                       hipStream_t stream) {
  FFA_REQUIRE(pos && out && n > 0 && d > 0 && repeat > 0, "positional_encoding: bad arguments");
  const int total = n * d * repeat;
  hipLaunchKernelGGL(positional_encoding_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, pos, out, n, d, repeat,
                     period);
  return ffa_check_launch("positional_encoding");
}

// x[n][p][c] += vec[n][c]   (n images of P pixels, C channels; vec f32)
template <typename T>
__global__ void add_rowvec_kernel(T* __restrict__ x, const float* __restrict__ vec, long long total, int P, int C) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long n = i / ((long long)P * C);
    ffa_store_elem<T>(x + i, ffa_load_elem<T>(x + i) + vec[n * C + c]);
  }
}

extern "C" int ffa_add_rowvec(int dtype, void* x, const float* vec, int N, int P, int C, hipStream_t stream) {
  FFA_REQUIRE(x && vec && N > 0 && P > 0 && C > 0, "add_rowvec: bad arguments");
  const long long total = (long long)N * P * C;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(add_rowvec_kernel<ffa_bf16>, dim3(grid), dim3(256), 0, stream, (ffa_bf16*)x, vec, total, P, C);
  else
    hipLaunchKernelGGL(add_rowvec_kernel<float>, dim3(grid), dim3(256), 0, stream, (float*)x, vec, total, P, C);
  return ffa_check_launch("add_rowvec");
}

// ------------------------------------------------------------------------------------------------
// L-TAE attention with one learnt query per head, per pixel:  k [B][T][P][NH*DK], v [B][T][P][NH*DV],
//   score[t] = <Q[h], k[b][t][p][h*DK ..]> / sqrt(DK), padded dates -> -1e3, attn = softmax_t(score),
//   out[b][p][h*DV + j] = sum_t attn[t] * v[b][t][p][h*DV + j];   attn is also written as f32 [NH][B][T][P]
// one thread per (b, p, head); T is small (tens of dates)

template <typename T>
__global__ void ltae_attention_kernel(const T* __restrict__ k, const T* __restrict__ v, const float* __restrict__ Q,
                                      const unsigned char* __restrict__ pad, T* __restrict__ out,
                                      float* __restrict__ attn, int B, int Tn, int P, int NH, int DK, int DV) {
  const long long total = (long long)B * P * NH;
  const float inv_temp = 1.0f / sqrtf((float)DK);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i % NH);
    const long long bp = i / NH;
    const int p = (int)(bp % P);
    const int b = (int)(bp / P);
    float m = -INFINITY;
    for (int t = 0; t < Tn; ++t) {
      float s = 0.f;
      const T* kk = k + (((long long)(b * Tn + t) * P + p) * NH + h) * DK;
      for (int j = 0; j < DK; ++j) s += Q[h * DK + j] * ffa_load_elem<T>(kk + j);
      s *= inv_temp;
      if (pad[b * Tn + t]) s = -1e3f;
      attn[(((long long)h * B + b) * Tn + t) * P + p] = s;
      m = fmaxf(m, s);
    }
    float se = 0.f;
    for (int t = 0; t < Tn; ++t) {
      const long long ai = (((long long)h * B + b) * Tn + t) * P + p;
      const float e = expf(attn[ai] - m);
      attn[ai] = e;
      se += e;
    }
    float acc[32];
    for (int j = 0; j < DV; ++j) acc[j] = 0.f;
    for (int t = 0; t < Tn; ++t) {
      const long long ai = (((long long)h * B + b) * Tn + t) * P + p;
      const float a = attn[ai] / se;
      attn[ai] = a;
      const T* vv = v + (((long long)(b * Tn + t) * P + p) * NH + h) * DV;
      for (int j = 0; j < DV; ++j) acc[j] += a * ffa_load_elem<T>(vv + j);
    }
    T* o = out + (((long long)b * P + p) * NH + h) * DV;
    for (int j = 0; j < DV; ++j) ffa_store_elem<T>(o + j, acc[j]);
  }
}

extern "C" int ffa_ltae_attention(int dtype, const void* k, const void* v, const float* Q, const unsigned char* pad,
                                  void* out, float* attn, int B, int T, int P, int n_head, int d_k, int d_v,
                                  hipStream_t stream) {
  FFA_REQUIRE(k && v && Q && pad && out && attn, "ltae_attention: null pointer");
  FFA_REQUIRE(B > 0 && T > 0 && P > 0 && n_head > 0 && d_k > 0 && d_v > 0 && d_v <= 32, "ltae_attention: bad geometry");
  const long long total = (long long)B * P * n_head;
  const int grid = (int)((total + 127) / 128 < 4096 ? (total + 127) / 128 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(ltae_attention_kernel<ffa_bf16>, dim3(grid), dim3(128), 0, stream, (const ffa_bf16*)k,
                       (const ffa_bf16*)v, Q, pad, (ffa_bf16*)out, attn, B, T, P, n_head, d_k, d_v);
  else
    hipLaunchKernelGGL(ltae_attention_kernel<float>, dim3(grid), dim3(128), 0, stream, (const float*)k, (const float*)v,
                       Q, pad, (float*)out, attn, B, T, P, n_head, d_k, d_v);
  return ffa_check_launch("ltae_attention");
}

// ------------------------------------------------------------------------------------------------
// attention-weighted temporal mean of a skip feature map, channel groups sharing a head's mask:
//   out[b][p][c] = sum_t attn[c / (C/NH)][b][t][p] * (pad[b][t] && use_pad ? 0 : 1) * x[b][t][p][c]
// attn f32 [NH][B][T][P] already at the resolution of x

template <typename T>
__global__ void temporal_aggregate_kernel(const T* __restrict__ x, const float* __restrict__ attn,
                                          const unsigned char* __restrict__ pad, T* __restrict__ out, int B, int Tn,
                                          int P, int C, int NH, int use_pad) {
  const long long total = (long long)B * P * C;
  const int cg = C / NH;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long bp = i / C;
    const int p = (int)(bp % P);
    const int b = (int)(bp / P);
    const int h = c / cg;
    float acc = 0.f;
    for (int t = 0; t < Tn; ++t) {
      float a = attn[(((long long)h * B + b) * Tn + t) * P + p];
      if (use_pad && pad[b * Tn + t]) a = 0.f;
      acc += a * ffa_load_elem<T>(x + ((long long)(b * Tn + t) * P + p) * C + c);
    }
    ffa_store_elem<T>(out + i, acc);
  }
}

extern "C" int ffa_temporal_aggregate(int dtype, const void* x, const float* attn, const unsigned char* pad, void* out,
                                      int B, int T, int P, int C, int n_head, int use_pad, hipStream_t stream) {
  FFA_REQUIRE(x && attn && pad && out && B > 0 && T > 0 && P > 0 && C > 0 && n_head > 0 && C % n_head == 0,
              "temporal_aggregate: bad arguments");
  const long long total = (long long)B * P * C;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(temporal_aggregate_kernel<ffa_bf16>, dim3(grid), dim3(256), 0, stream, (const ffa_bf16*)x, attn,
                       pad, (ffa_bf16*)out, B, T, P, C, n_head, use_pad);
  else
    hipLaunchKernelGGL(temporal_aggregate_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, attn, pad,
                       (float*)out, B, T, P, C, n_head, use_pad);
  return ffa_check_launch("temporal_aggregate");
}

// ------------------------------------------------------------------------------------------------
// images n with pad[n] != 0 are overwritten with `value` (TemporallySharedBlock: padded dates come out as pad_value)

template <typename T>
__global__ void mask_images_kernel(T* __restrict__ x, const unsigned char* __restrict__ pad, long long total,
                                   long long per_image, float value) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    if (pad[i / per_image]) ffa_store_elem<T>(x + i, value);
}

extern "C" int ffa_mask_images(int dtype, void* x, const unsigned char* pad, int N, long long per_image, float value,
                               hipStream_t stream) {
  FFA_REQUIRE(x && pad && N > 0 && per_image > 0, "mask_images: bad arguments");
  const long long total = (long long)N * per_image;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(mask_images_kernel<ffa_bf16>, dim3(grid), dim3(256), 0, stream, (ffa_bf16*)x, pad, total,
                       per_image, value);
  else
    hipLaunchKernelGGL(mask_images_kernel<float>, dim3(grid), dim3(256), 0, stream, (float*)x, pad, total, per_image,
                       value);
  return ffa_check_launch("mask_images");
}

// pad[n] = every element of image n equals `value` (NCHW or NHWC alike: a flat run of per_image f32 values) --
// "(input == pad_value).all(-1).all(-1).all(-1)" of UTAE.forward :133-135
__global__ void detect_pad_kernel(const float* __restrict__ x, unsigned char* __restrict__ pad, long long per_image,
                                  float value) {
  __shared__ int any_diff;
  if (threadIdx.x == 0) any_diff = 0;
  __syncthreads();
  const float* p = x + (long long)blockIdx.x * per_image;
  int d = 0;
  for (long long i = threadIdx.x; i < per_image; i += blockDim.x) d |= (p[i] != value);
  if (d) any_diff = 1;
  __syncthreads();
  if (threadIdx.x == 0) pad[blockIdx.x] = any_diff ? 0 : 1;
}

extern "C" int ffa_detect_pad_images(const float* x, unsigned char* pad, int N, long long per_image, float value,
                                     hipStream_t stream) {
  FFA_REQUIRE(x && pad && N > 0 && per_image > 0, "detect_pad_images: bad arguments");
  hipLaunchKernelGGL(detect_pad_kernel, dim3(N), dim3(256), 0, stream, x, pad, per_image, value);
  return ffa_check_launch("detect_pad_images");
}

// ================================================================================================
// Backward kernels of the U-TAE branch (training).  Same geometry conventions as the forward kernels above.

// ---- reflect padding backward: dx[n][y][x] = sum of dpad over the padded positions that read (y, x)
template <typename T>
__global__ void reflect_pad1_bwd_kernel(const T* __restrict__ dpad, T* __restrict__ dx, int N, int H, int W, int C8) {
  const long long total = (long long)N * H * W * C8;
  const int Hp = H + 2, Wp = W + 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % C8);
    long long p = i / C8;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const long long n = p / H;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + 1;
    if (y == 1) ys[ny++] = 0;
    if (y == H - 2) ys[ny++] = Hp - 1;
    xs[nx++] = x + 1;
    if (x == 1) xs[nx++] = 0;
    if (x == W - 2) xs[nx++] = Wp - 1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        float v[8];
        ffa_load8<T>(dpad + ((n * Hp + ys[a]) * Wp + xs[b]) * (long long)(C8 * 8) + g * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    ffa_store8<T>(dx + i * 8, acc);
  }
}

extern "C" int ffa_reflect_pad1_bwd(int dtype, const void* dpad, void* dx, int N, int H, int W, int C,
                                    hipStream_t stream) {
  FFA_REQUIRE(dpad && dx && N > 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, "reflect_pad1_bwd: bad arguments");
  const long long total = (long long)N * H * W * (C / 8);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(reflect_pad1_bwd_kernel<ffa_bf16>, dim3(grid), dim3(256), 0, stream, (const ffa_bf16*)dpad,
                       (ffa_bf16*)dx, N, H, W, C / 8);
  else
    hipLaunchKernelGGL(reflect_pad1_bwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)dpad, (float*)dx,
                       N, H, W, C / 8);
  return ffa_check_launch("reflect_pad1_bwd");
}

// ---- GroupNorm backward, one block per (sample, group) like the forward.  y = [res +] relu?(xhat gamma + beta):
//   g = dy * (relu ? y_pre > 0 : 1) * gamma,  dx = rstd (g - mean(g) - xhat mean(g xhat)),
//   partial[s][0][c] = sum_inner dy' xhat, partial[s][1][c] = sum_inner dy'  (dy' = masked dy); the caller sums the
//   partial rows over the samples (ffa_column_sums: fixed order).  256 % (C / G) == 0: a thread always meets the same
//   channel, so the per-channel sums accumulate in registers and are combined through LDS in a fixed order.
template <typename T>
__global__ void __launch_bounds__(FFA_T_THREADS) group_norm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                      T* __restrict__ dx, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta,
                                                                      float* __restrict__ partial, int Q,
                                                                      long long stride_hi, long long stride_lo, int inner,
                                                                      long long inner_stride, int C, int G, float eps,
                                                                      int relu) {
  const int s = blockIdx.x / G, g = blockIdx.x % G;
  const int cg = C / G;
  const long long base = (long long)(s / Q) * stride_hi + (long long)(s % Q) * stride_lo + (long long)g * cg;
  const int n = inner * cg;
  __shared__ float red[FFA_T_THREADS / 64];
  __shared__ float chan[2][FFA_T_THREADS];
  auto block_sum = [&](float v) {
    v = ffa_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < FFA_T_THREADS / 64; ++w) t += red[w];
    return t;
  };
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += FFA_T_THREADS)
    a += ffa_load_elem<T>(x + base + (long long)(i / cg) * inner_stride + (i % cg));
  const float mean = block_sum(a) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += FFA_T_THREADS) {
    const float d = ffa_load_elem<T>(x + base + (long long)(i / cg) * inner_stride + (i % cg)) - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(block_sum(q) / (float)n + eps);
  const int c = g * cg + (threadIdx.x % cg);  // this thread's channel (256 % cg == 0)
  const float ga = gamma[c], be = beta[c];
  float s1 = 0.f, s2 = 0.f, dgam = 0.f, dbet = 0.f;
  for (int i = threadIdx.x; i < n; i += FFA_T_THREADS) {
    const long long off = base + (long long)(i / cg) * inner_stride + (i % cg);
    const float xh = (ffa_load_elem<T>(x + off) - mean) * rstd;
    float d = ffa_load_elem<T>(dy + off);
    if (relu && xh * ga + be <= 0.f) d = 0.f;
    dgam += d * xh;
    dbet += d;
    s1 += d * ga;
    s2 += d * ga * xh;
  }
  const float m1 = block_sum(s1) / (float)n, m2 = block_sum(s2) / (float)n;
  for (int i = threadIdx.x; i < n; i += FFA_T_THREADS) {
    const long long off = base + (long long)(i / cg) * inner_stride + (i % cg);
    const float xh = (ffa_load_elem<T>(x + off) - mean) * rstd;
    float d = ffa_load_elem<T>(dy + off);
    if (relu && xh * ga + be <= 0.f) d = 0.f;
    ffa_store_elem<T>(dx + off, rstd * (d * ga - m1 - xh * m2));
  }
  __syncthreads();
  chan[0][threadIdx.x] = dgam;
  chan[1][threadIdx.x] = dbet;
  __syncthreads();
  if (threadIdx.x < cg) {
    float u = 0.f, v = 0.f;
    for (int t = threadIdx.x; t < FFA_T_THREADS; t += cg) {
      u += chan[0][t];
      v += chan[1][t];
    }
    partial[((long long)s * 2) * C + g * cg + threadIdx.x] = u;
    partial[((long long)s * 2 + 1) * C + g * cg + threadIdx.x] = v;
  }
}

extern "C" int ffa_group_norm_bwd(int dtype, const void* x, const void* dy, void* dx, const float* gamma,
                                  const float* beta, float* partial, long long samples, int Q, long long stride_hi,
                                  long long stride_lo, int inner, long long inner_stride, int C, int groups, float eps,
                                  int relu, hipStream_t stream) {
  FFA_REQUIRE(x && dy && dx && gamma && beta && partial, "group_norm_bwd: null pointer");
  FFA_REQUIRE(samples > 0 && Q > 0 && inner > 0 && C > 0 && groups > 0 && C % groups == 0 &&
                  samples * groups < (1LL << 31), "group_norm_bwd: bad geometry");
  FFA_REQUIRE(FFA_T_THREADS % (C / groups) == 0, "group_norm_bwd: %d channels per group do not divide the block size",
              C / groups);
  const int grid = (int)(samples * groups);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(group_norm_bwd_kernel<ffa_bf16>, dim3(grid), dim3(FFA_T_THREADS), 0, stream, (const ffa_bf16*)x,
                       (const ffa_bf16*)dy, (ffa_bf16*)dx, gamma, beta, partial, Q, stride_hi, stride_lo, inner,
                       inner_stride, C, groups, eps, relu);
  else
    hipLaunchKernelGGL(group_norm_bwd_kernel<float>, dim3(grid), dim3(FFA_T_THREADS), 0, stream, (const float*)x,
                       (const float*)dy, (float*)dx, gamma, beta, partial, Q, stride_hi, stride_lo, inner, inner_stride,
                       C, groups, eps, relu);
  return ffa_check_launch("group_norm_bwd");
}

// ---- L-TAE attention, training: forward with the attention dropout (drop = 0 or 1 / (1 - p) per (head, b, t, pixel))
// applied to the probabilities -- the dropped-out masks are what the reference returns and feeds its
// Temporal_Aggregator (:399-403) --, the clean probabilities kept for the backward pass.
template <typename T>
__global__ void ltae_attention_train_kernel(const T* __restrict__ k, const T* __restrict__ v, const float* __restrict__ Q,
                                            const unsigned char* __restrict__ pad, const float* __restrict__ drop,
                                            T* __restrict__ out, float* __restrict__ attn, float* __restrict__ prob,
                                            int B, int Tn, int P, int NH, int DK, int DV) {
  const long long total = (long long)B * P * NH;
  const float inv_temp = 1.0f / sqrtf((float)DK);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i % NH);
    const long long bp = i / NH;
    const int p = (int)(bp % P);
    const int b = (int)(bp / P);
    float m = -INFINITY;
    for (int t = 0; t < Tn; ++t) {
      float s = 0.f;
      const T* kk = k + (((long long)(b * Tn + t) * P + p) * NH + h) * DK;
      for (int j = 0; j < DK; ++j) s += Q[h * DK + j] * ffa_load_elem<T>(kk + j);
      s *= inv_temp;
      if (pad[b * Tn + t]) s = -1e3f;
      prob[(((long long)h * B + b) * Tn + t) * P + p] = s;
      m = fmaxf(m, s);
    }
    float se = 0.f;
    for (int t = 0; t < Tn; ++t) {
      const long long ai = (((long long)h * B + b) * Tn + t) * P + p;
      const float e = expf(prob[ai] - m);
      prob[ai] = e;
      se += e;
    }
    float acc[32];
    for (int j = 0; j < DV; ++j) acc[j] = 0.f;
    for (int t = 0; t < Tn; ++t) {
      const long long ai = (((long long)h * B + b) * Tn + t) * P + p;
      const float pr = prob[ai] / se;
      prob[ai] = pr;
      const float a = drop ? pr * drop[ai] : pr;
      attn[ai] = a;
      const T* vv = v + (((long long)(b * Tn + t) * P + p) * NH + h) * DV;
      for (int j = 0; j < DV; ++j) acc[j] += a * ffa_load_elem<T>(vv + j);
    }
    T* o = out + (((long long)b * P + p) * NH + h) * DV;
    for (int j = 0; j < DV; ++j) ffa_store_elem<T>(o + j, acc[j]);
  }
}

extern "C" int ffa_ltae_attention_train(int dtype, const void* k, const void* v, const float* Q, const unsigned char* pad,
                                        const float* drop, void* out, float* attn, float* prob, int B, int T, int P,
                                        int n_head, int d_k, int d_v, hipStream_t stream) {
  FFA_REQUIRE(k && v && Q && pad && out && attn && prob, "ltae_attention_train: null pointer");
  FFA_REQUIRE(B > 0 && T > 0 && P > 0 && n_head > 0 && d_k > 0 && d_v > 0 && d_v <= 32, "ltae_attention_train: bad geometry");
  const long long total = (long long)B * P * n_head;
  const int grid = (int)((total + 127) / 128 < 4096 ? (total + 127) / 128 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(ltae_attention_train_kernel<ffa_bf16>, dim3(grid), dim3(128), 0, stream, (const ffa_bf16*)k,
                       (const ffa_bf16*)v, Q, pad, drop, (ffa_bf16*)out, attn, prob, B, T, P, n_head, d_k, d_v);
  else
    hipLaunchKernelGGL(ltae_attention_train_kernel<float>, dim3(grid), dim3(128), 0, stream, (const float*)k,
                       (const float*)v, Q, pad, drop, (float*)out, attn, prob, B, T, P, n_head, d_k, d_v);
  return ffa_check_launch("ltae_attention_train");
}

// backward: one thread per (b, pixel, head).  a = prob * drop, out = sum_t a v:
//   da[t] = <dout, v[t]> + dattn_ext[t] (what the aggregators send back to the returned masks), dv[t] = a[t] dout,
//   dp = da * drop, ds[t] = prob[t] (dp[t] - sum prob dp), padded dates: score was replaced by a constant -> ds = 0,
//   dk[t] = ds[t] Q[h] / temp, dQ[h] += sum ds[t] k[t] / temp  (per-block partial rows, summed by the caller)
template <typename T>
__global__ void __launch_bounds__(128) ltae_attention_bwd_kernel(
    const T* __restrict__ k, const T* __restrict__ v, const float* __restrict__ Q, const unsigned char* __restrict__ pad,
    const float* __restrict__ drop, const float* __restrict__ prob, const T* __restrict__ dout,
    const float* __restrict__ dattn_ext, T* __restrict__ dk, T* __restrict__ dv, float* __restrict__ dq_partial, int B,
    int Tn, int P, int NH, int DK, int DV) {
  extern __shared__ float sdq[];  // [NH * DK]
  for (int i = threadIdx.x; i < NH * DK; i += blockDim.x) sdq[i] = 0.f;
  __syncthreads();
  const long long total = (long long)B * P * NH;
  const float inv_temp = 1.0f / sqrtf((float)DK);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i % NH);
    const long long bp = i / NH;
    const int p = (int)(bp % P);
    const int b = (int)(bp / P);
    float go[32];
    const T* dd = dout + (((long long)b * P + p) * NH + h) * DV;
    for (int j = 0; j < DV; ++j) go[j] = ffa_load_elem<T>(dd + j);
    float dot = 0.f;  // sum_t prob[t] dp[t]
    for (int t = 0; t < Tn; ++t) {
      const long long ai = (((long long)h * B + b) * Tn + t) * P + p;
      const T* vv = v + (((long long)(b * Tn + t) * P + p) * NH + h) * DV;
      float da = dattn_ext ? dattn_ext[ai] : 0.f;
      for (int j = 0; j < DV; ++j) da += go[j] * ffa_load_elem<T>(vv + j);
      const float dr = drop ? drop[ai] : 1.f;
      dot += prob[ai] * da * dr;
    }
    float dqh[8];
    for (int j = 0; j < DK; ++j) dqh[j] = 0.f;
    for (int t = 0; t < Tn; ++t) {
      const long long ai = (((long long)h * B + b) * Tn + t) * P + p;
      const long long row = ((long long)(b * Tn + t) * P + p) * NH + h;
      const T* vv = v + row * DV;
      const float dr = drop ? drop[ai] : 1.f;
      const float pr = prob[ai];
      float da = dattn_ext ? dattn_ext[ai] : 0.f;
      for (int j = 0; j < DV; ++j) da += go[j] * ffa_load_elem<T>(vv + j);
      const float a = pr * dr;
      for (int j = 0; j < DV; ++j) ffa_store_elem<T>(dv + row * DV + j, a * go[j]);
      float ds = pr * (da * dr - dot);
      if (pad[b * Tn + t]) ds = 0.f;
      ds *= inv_temp;
      const T* kk = k + row * DK;
      for (int j = 0; j < DK; ++j) {
        ffa_store_elem<T>(dk + row * DK + j, ds * Q[h * DK + j]);
        dqh[j] += ds * ffa_load_elem<T>(kk + j);
      }
    }
    for (int j = 0; j < DK; ++j) unsafeAtomicAdd(&sdq[h * DK + j], dqh[j]);  // ds_add_f32
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NH * DK; i += blockDim.x) dq_partial[(long long)blockIdx.x * NH * DK + i] = sdq[i];
}

extern "C" int ffa_ltae_attention_bwd_blocks(int B, int P, int n_head) {
  const long long total = (long long)B * P * n_head;
  return (int)((total + 127) / 128 < 1024 ? (total + 127) / 128 : 1024);
}

extern "C" int ffa_ltae_attention_bwd(int dtype, const void* k, const void* v, const float* Q, const unsigned char* pad,
                                      const float* drop, const float* prob, const void* dout, const float* dattn_ext,
                                      void* dk, void* dv, float* dq_partial, int B, int T, int P, int n_head, int d_k,
                                      int d_v, hipStream_t stream) {
  FFA_REQUIRE(k && v && Q && pad && prob && dout && dk && dv && dq_partial, "ltae_attention_bwd: null pointer");
  FFA_REQUIRE(B > 0 && T > 0 && P > 0 && n_head > 0 && d_k > 0 && d_k <= 8 && d_v > 0 && d_v <= 32,
              "ltae_attention_bwd: bad geometry");
  const int grid = ffa_ltae_attention_bwd_blocks(B, P, n_head);
  const size_t lds = (size_t)n_head * d_k * sizeof(float);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(ltae_attention_bwd_kernel<ffa_bf16>, dim3(grid), dim3(128), lds, stream, (const ffa_bf16*)k,
                       (const ffa_bf16*)v, Q, pad, drop, prob, (const ffa_bf16*)dout, dattn_ext, (ffa_bf16*)dk,
                       (ffa_bf16*)dv, dq_partial, B, T, P, n_head, d_k, d_v);
  else
    hipLaunchKernelGGL(ltae_attention_bwd_kernel<float>, dim3(grid), dim3(128), lds, stream, (const float*)k,
                       (const float*)v, Q, pad, drop, prob, (const float*)dout, dattn_ext, (float*)dk, (float*)dv,
                       dq_partial, B, T, P, n_head, d_k, d_v);
  return ffa_check_launch("ltae_attention_bwd");
}

// ---- temporal aggregation backward: dx[b][t][p][c] = attn' dout[b][p][c], dattn[h][b][t][p] = sum_{c in group h}
// x[b][t][p][c] dout[b][p][c] (0 for a padded date when use_pad); one thread per (b, t, p, head)
template <typename T>
__global__ void temporal_aggregate_bwd_kernel(const T* __restrict__ x, const float* __restrict__ attn,
                                              const unsigned char* __restrict__ pad, const T* __restrict__ dout,
                                              T* __restrict__ dx, float* __restrict__ dattn, int B, int Tn, int P, int C,
                                              int NH, int use_pad) {
  const long long total = (long long)B * Tn * P * NH;
  const int cg = C / NH;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i % NH);
    long long r = i / NH;
    const int p = (int)(r % P);
    r /= P;
    const int t = (int)(r % Tn);
    const int b = (int)(r / Tn);
    const long long ai = (((long long)h * B + b) * Tn + t) * P + p;
    const bool dead = use_pad && pad[b * Tn + t];
    const float a = dead ? 0.f : attn[ai];
    const T* xx = x + ((long long)(b * Tn + t) * P + p) * C + h * cg;
    const T* dd = dout + ((long long)b * P + p) * C + h * cg;
    T* dxx = dx + ((long long)(b * Tn + t) * P + p) * C + h * cg;
    float acc = 0.f;
    for (int c = 0; c < cg; ++c) {
      const float d = ffa_load_elem<T>(dd + c);
      acc += ffa_load_elem<T>(xx + c) * d;
      ffa_store_elem<T>(dxx + c, a * d);
    }
    dattn[ai] = dead ? 0.f : acc;
  }
}

extern "C" int ffa_temporal_aggregate_bwd(int dtype, const void* x, const float* attn, const unsigned char* pad,
                                          const void* dout, void* dx, float* dattn, int B, int T, int P, int C,
                                          int n_head, int use_pad, hipStream_t stream) {
  FFA_REQUIRE(x && attn && pad && dout && dx && dattn && B > 0 && T > 0 && P > 0 && C > 0 && n_head > 0 &&
                  C % n_head == 0, "temporal_aggregate_bwd: bad arguments");
  const long long total = (long long)B * T * P * n_head;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(temporal_aggregate_bwd_kernel<ffa_bf16>, dim3(grid), dim3(256), 0, stream, (const ffa_bf16*)x,
                       attn, pad, (const ffa_bf16*)dout, (ffa_bf16*)dx, dattn, B, T, P, C, n_head, use_pad);
  else
    hipLaunchKernelGGL(temporal_aggregate_bwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, attn, pad,
                       (const float*)dout, (float*)dx, dattn, B, T, P, C, n_head, use_pad);
  return ffa_check_launch("temporal_aggregate_bwd");
}

// ---- y = x * m elementwise (dropout with a pre-scaled keep mask: 0 or 1 / (1 - p); its own backward)
template <typename T>
__global__ void mul_kernel(const T* __restrict__ x, const T* __restrict__ m, T* __restrict__ y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    ffa_store_elem<T>(y + i, ffa_load_elem<T>(x + i) * ffa_load_elem<T>(m + i));
}

extern "C" int ffa_mul(int dtype, const void* x, const void* m, void* y, long long n, hipStream_t stream) {
  FFA_REQUIRE(x && m && y && n > 0, "mul: bad arguments");
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(mul_kernel<ffa_bf16>, dim3(grid), dim3(256), 0, stream, (const ffa_bf16*)x, (const ffa_bf16*)m,
                       (ffa_bf16*)y, n);
  else
    hipLaunchKernelGGL(mul_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (const float*)m, (float*)y, n);
  return ffa_check_launch("mul");
}

// ---- y = (x0 + x1 + ... + x(n-1)) / divisor elementwise, n <= 4 (torch.mean(torch.stack(maps), dim=0) of FusionHandler's
// case 3, reference flair_model.py:496-501: left-to-right f32 sum, one division, one rounding; with n = 1 and
// divisor = n_branches it is that mean's backward)
template <typename T>
__global__ void mean_stack_kernel(const T* __restrict__ x0, const T* __restrict__ x1, const T* __restrict__ x2,
                                  const T* __restrict__ x3, int n, float divisor, T* __restrict__ y, long long numel) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) {
    float acc = ffa_load_elem<T>(x0 + i);
    if (n > 1) acc += ffa_load_elem<T>(x1 + i);
    if (n > 2) acc += ffa_load_elem<T>(x2 + i);
    if (n > 3) acc += ffa_load_elem<T>(x3 + i);
    ffa_store_elem<T>(y + i, acc / divisor);
  }
}

extern "C" int ffa_mean_stack(int dtype, const void* const* xs, int n, float divisor, void* y, long long numel,
                              hipStream_t stream) {
  FFA_REQUIRE(xs && y && n >= 1 && n <= 4 && numel > 0 && divisor != 0.f, "mean_stack: bad arguments (1 to 4 operands)");
  for (int i = 0; i < n; ++i) FFA_REQUIRE(xs[i], "mean_stack: null operand");
  const void* p[4] = {xs[0], n > 1 ? xs[1] : xs[0], n > 2 ? xs[2] : xs[0], n > 3 ? xs[3] : xs[0]};
  const int grid = (int)((numel + 255) / 256 < 4096 ? (numel + 255) / 256 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(mean_stack_kernel<ffa_bf16>, dim3(grid), dim3(256), 0, stream, (const ffa_bf16*)p[0],
                       (const ffa_bf16*)p[1], (const ffa_bf16*)p[2], (const ffa_bf16*)p[3], n, divisor, (ffa_bf16*)y, numel);
  else
    hipLaunchKernelGGL(mean_stack_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)p[0], (const float*)p[1],
                       (const float*)p[2], (const float*)p[3], n, divisor, (float*)y, numel);
  return ffa_check_launch("mean_stack");
}
