// Decoder resampling and the per-pixel loss / prediction kernels (all HBM-bound, NHWC).
//
//   * nearest x2 upsample + channel concat (smp UnetDecoder's DecoderBlock: F.interpolate(scale 2,
//     'nearest') then torch.cat([x, skip], 1)) forward and backward
//   * bilinear resize, align_corners=False (flair_hub/models/flair_model.py:318-327 interpolate_map)
//     forward and backward
//   * weighted softmax cross-entropy forward + dlogits + argmax in ONE pass over the logits
//     (flair_hub/tasks/module_setup.py:150-161 nn.CrossEntropyLoss(weight=w);
//      flair_hub/tasks/tasks_module.py:155,158 loss + argmax(softmax(logits)))
//   * margin crop + argmax -> uint8 (flair_zonal_detection/inference.py:300 +
//     flair_zonal_detection/postprocess.py:25-27) and class_prob -> round(softmax*255) (:19-23)
//   * one-hot -> index (tasks_module.py:153)
#include "ffa_common.h"

#include <stdlib.h>

#define FFA_EW_THREADS 256

static inline int ew_grid(long long items) {
  long long g = (items + FFA_EW_THREADS - 1) / FFA_EW_THREADS;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------
// nearest x2 + concat

template <typename T>
__global__ void up2_concat_fwd_kernel(const T* __restrict__ lo, const T* __restrict__ skip, T* __restrict__ out, int B,
                                      int Hl, int Wl, int C1, int C2) {
  const int H = Hl * 2, W = Wl * 2, C = C1 + C2;
  const int CG = C / 8, CG1 = C1 / 8;
  const long long total = (long long)B * H * W * CG;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % CG);
    long long p = i / CG;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const long long b = p / H;
    ffa_u32x4 v0, v1;
    const T* src;
    if (g < CG1)
      src = lo + ((b * Hl + (y >> 1)) * Wl + (x >> 1)) * C1 + g * 8;
    else
      src = skip + ((b * H + y) * W + x) * C2 + (g - CG1) * 8;
    v0 = reinterpret_cast<const ffa_u32x4*>(src)[0];
    if (sizeof(T) == 4) v1 = reinterpret_cast<const ffa_u32x4*>(src)[1];
    ffa_u32x4* dst = reinterpret_cast<ffa_u32x4*>(out + i * 8);
    dst[0] = v0;
    if (sizeof(T) == 4) dst[1] = v1;
  }
}

template <typename T>
__global__ void up2_concat_bwd_kernel(const T* __restrict__ dcat, T* __restrict__ dlo, T* __restrict__ dskip, int B,
                                      int Hl, int Wl, int C1, int C2) {
  // item space: first the low-res pixels x C1 groups (sum of the 4 children), then the skip copy
  const int H = Hl * 2, W = Wl * 2, C = C1 + C2;
  const int CG1 = C1 / 8, CG2 = C2 / 8;
  const long long n_lo = (long long)B * Hl * Wl * CG1;
  const long long n_sk = dskip ? (long long)B * H * W * CG2 : 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_lo + n_sk;
       i += (long long)gridDim.x * blockDim.x) {
    if (i < n_lo) {
      const int g = (int)(i % CG1);
      long long p = i / CG1;
      const int x = (int)(p % Wl);
      p /= Wl;
      const int y = (int)(p % Hl);
      const long long b = p / Hl;
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
      for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx) {
          float v[8];
          ffa_load8<T>(dcat + ((b * H + 2 * y + dyy) * W + 2 * x + dxx) * C + g * 8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
      ffa_store8<T>(dlo + i * 8, acc);
    } else {
      const long long k = i - n_lo;
      const int g = (int)(k % CG2);
      const long long p = k / CG2;
      float v[8];
      ffa_load8<T>(dcat + p * C + C1 + g * 8, v);
      ffa_store8<T>(dskip + k * 8, v);
    }
  }
}

extern "C" int ffa_upsample_nearest2x_concat_fwd(int dtype, const void* lo, const void* skip, void* out, int B, int Hl,
                                                 int Wl, int C1, int C2, hipStream_t stream) {
  FFA_REQUIRE(lo && out && C1 % 8 == 0 && C2 % 8 == 0 && (C2 == 0 || skip), "up2_concat_fwd: bad arguments");
  const long long items = (long long)B * Hl * 2 * Wl * 2 * ((C1 + C2) / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(up2_concat_fwd_kernel<ffa_bf16>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)lo, (const ffa_bf16*)skip, (ffa_bf16*)out, B, Hl, Wl, C1, C2);
  else
    hipLaunchKernelGGL(up2_concat_fwd_kernel<float>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)lo, (const float*)skip, (float*)out, B, Hl, Wl, C1, C2);
  return ffa_check_launch("up2_concat_fwd");
}

extern "C" int ffa_upsample_nearest2x_concat_bwd(int dtype, const void* dcat, void* dlo, void* dskip, int B, int Hl,
                                                 int Wl, int C1, int C2, hipStream_t stream) {
  // dskip may be null with C2 > 0: the caller then uses the channel slice dcat[..., C1:] itself as the skip gradient
  // (it is only ever summed with the encoder-side gradient of the same feature map, which reads strided input fine)
  FFA_REQUIRE(dcat && dlo && C1 % 8 == 0 && C2 % 8 == 0, "up2_concat_bwd: bad arguments");
  const long long items =
      (long long)B * Hl * Wl * (C1 / 8) + (dskip ? (long long)B * Hl * 2 * Wl * 2 * (C2 / 8) : 0LL);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(up2_concat_bwd_kernel<ffa_bf16>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)dcat, (ffa_bf16*)dlo, (ffa_bf16*)(C2 ? dskip : nullptr), B, Hl, Wl, C1, C2);
  else
    hipLaunchKernelGGL(up2_concat_bwd_kernel<float>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)dcat, (float*)dlo, (float*)(C2 ? dskip : nullptr), B, Hl, Wl, C1, C2);
  return ffa_check_launch("up2_concat_bwd");
}

// ------------------------------------------------------------------------------------------------
// bilinear, align_corners=False.  Source index as ATen computes it (area_pixel_compute_source_index):
//   src = max(0, scale * (dst + 0.5) - 0.5), scale = in / out in f32, i0 = (int)src,
//   i1 = i0 + (i0 < in - 1), lambda1 = src - i0, lambda0 = 1 - lambda1

__device__ __forceinline__ void bilinear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l0,
                                             float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

template <typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Hi, int Wi, int Ho, int Wo,
                                    int C, float sy, float sx) {
  const int CG = C / 8;
  const long long total = (long long)B * Ho * Wo * CG;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % CG);
    long long p = i / CG;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const long long b = p / Ho;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilinear_src(oy, sy, Hi, y0, y1, ly0, ly1);
    bilinear_src(ox, sx, Wi, x0, x1, lx0, lx1);
    float v00[8], v01[8], v10[8], v11[8], o[8];
    ffa_load8<T>(x + ((b * Hi + y0) * Wi + x0) * C + g * 8, v00);
    ffa_load8<T>(x + ((b * Hi + y0) * Wi + x1) * C + g * 8, v01);
    ffa_load8<T>(x + ((b * Hi + y1) * Wi + x0) * C + g * 8, v10);
    ffa_load8<T>(x + ((b * Hi + y1) * Wi + x1) * C + g * 8, v11);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
    ffa_store8<T>(y + i * 8, o);
  }
}

// Backward of the bilinear resize in GATHER form: one thread per 8 channels of a SOURCE pixel walks the destination
// pixels whose footprint contains it (rows oy with y0(oy) == y or y1(oy) == y, same for columns; membership and
// weights come from the very bilinear_src() the forward uses) and sums w_y * w_x * dy in a fixed order -- no atomics,
// no f32 scratch image, reproducible bits.  The scatter form this replaces spent 2.5 ms per call (37 % of the
// two-modality training step, tools/bench_fusion.py) in 64 contended f32 atomics per source value when
// FusionHandler aligns a 4x coarser modality (flair_model.py:528-529).
template <typename T>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Hi, int Wi, int Ho,
                                    int Wo, int C, float sy, float sx) {
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
    // destination rows / columns whose source coordinate lies within one pixel of (y, x), one extra on each side
    // for the float rounding of the bounds; exact membership is decided per candidate below
    int oy_lo = (int)floorf(((float)y - 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)y + 1.5f) / sy - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)x - 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)x + 1.5f) / sx - 0.5f) + 1;
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
      bilinear_src(oy, sy, Hi, y0, y1, ly0, ly1);
      const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);  // y0 == y1 on the clamped last row
      if (y0 != y && y1 != y) continue;
      float row[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) row[e] = 0.f;
      const T* drow = dy + ((b * Ho + oy) * Wo) * C + g * 8;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        bilinear_src(ox, sx, Wi, x0, x1, lx0, lx1);
        if (x0 != x && x1 != x) continue;
        const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
        float gv[8];
        ffa_load8<T>(drow + (long long)ox * C, gv);
#pragma unroll
        for (int e = 0; e < 8; ++e) row[e] += wx * gv[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += wy * row[e];
    }
    ffa_store8<T>(dx + i * 8, acc);
  }
}

extern "C" int ffa_bilinear_fwd(int dtype, const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                hipStream_t stream) {
  FFA_REQUIRE(x && y && C % 8 == 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bilinear_fwd: bad arguments");
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  const long long items = (long long)B * Ho * Wo * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(bilinear_fwd_kernel<ffa_bf16>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)x, (ffa_bf16*)y, B, Hi, Wi, Ho, Wo, C, sy, sx);
  else
    hipLaunchKernelGGL(bilinear_fwd_kernel<float>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)x, (float*)y, B, Hi, Wi, Ho, Wo, C, sy, sx);
  return ffa_check_launch("bilinear_fwd");
}

// (kept in the ABI: earlier builds scattered into an f32 scratch image of this size; the gather-form backward needs
// no workspace and accepts a null pointer)
extern "C" long long ffa_bilinear_bwd_workspace_bytes(int B, int Hi, int Wi, int C) {
  (void)B; (void)Hi; (void)Wi; (void)C;
  return 0;
}

extern "C" int ffa_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                void* workspace, long long workspace_bytes, hipStream_t stream) {
  (void)workspace; (void)workspace_bytes;
  FFA_REQUIRE(dy && dx && C % 8 == 0 && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bilinear_bwd: bad arguments");
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  const long long items = (long long)B * Hi * Wi * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(bilinear_bwd_kernel<ffa_bf16>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)dy, (ffa_bf16*)dx, B, Hi, Wi, Ho, Wo, C, sy, sx);
  else
    hipLaunchKernelGGL(bilinear_bwd_kernel<float>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)dy, (float*)dx, B, Hi, Wi, Ho, Wo, C, sy, sx);
  return ffa_check_launch("bilinear_bwd");
}

// ------------------------------------------------------------------------------------------------
// weighted softmax cross-entropy.  One thread per pixel; logits [npix][Cp] (K real classes, the
// Cp - K pad channels are ignored on read and written as zeros in dlogits so the head dgrad never
// sees garbage).  Reduction: loss = sum_p w[t_p] * (lse_p - z_p[t_p]) / sum_p w[t_p]
// (torch.nn.CrossEntropyLoss(weight=w, reduction='mean')).  Targets are uint8 class indices;
// targets >= K contribute nothing (weight 0), matching ignore semantics for padded tiles.
//
// Pass A (ce_weight_sum_kernel) reduces sum_p w[t_p] from the targets alone (1 B/pixel);
// pass B reads the logits once and writes per-block loss partials, dlogits (already divided by the
// weight sum and multiplied by *grad_scale) and the per-pixel argmax.

#define FFA_CE_MAXK 32
#define FFA_CE_BLOCKS 1024

// Sum of the target weights.  vec16: the target pointer is 16-byte aligned -- a thread then takes 16 consecutive
// targets per 16-byte load (one byte per lane per load made this 24 us for 8.4 M targets) and looks the weights up in
// LDS; the ragged tail and unaligned tensors take the byte loop.  Fixed thread <-> pixel assignment and order either way.
__global__ void ce_weight_sum_kernel(const uint8_t* __restrict__ tgt, const float* __restrict__ w, int K,
                                     long long npix, float* __restrict__ parts, int vec16) {
  __shared__ float red[FFA_EW_THREADS / 64];
  __shared__ float wl[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) wl[i] = (i < K) ? w[i] : 0.f;
  __syncthreads();
  float s = 0.f;
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long done = 0;
  if (vec16) {
    const long long n16 = npix / 16;
    for (long long i = gid; i < n16; i += stride) {
      const uint4 v = reinterpret_cast<const uint4*>(tgt)[i];
      const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < 4; ++b) s += wl[(q[j] >> (8 * b)) & 0xff];
    }
    done = n16 * 16;
  }
  for (long long i = done + gid; i < npix; i += stride) s += wl[tgt[i]];
  s = ffa_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int k = 0; k < FFA_EW_THREADS / 64; ++k) tot += red[k];
    parts[blockIdx.x] = tot;
  }
}

// one wave; lane l adds partials l, l+64, ... in double, then a fixed xor-shuffle tree: reproducible order
__device__ __forceinline__ double ce_sum_partials(const float* __restrict__ parts, int n) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)parts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

__global__ void ce_finalize_sum_kernel(const float* __restrict__ parts, int n, float* __restrict__ out) {
  const double s = ce_sum_partials(parts, n);
  if (threadIdx.x == 0) out[0] = (float)s;
}

__global__ void ce_finalize_loss_kernel(const float* __restrict__ parts, int n, const float* __restrict__ wsum,
                                        float* __restrict__ loss) {
  const double s = ce_sum_partials(parts, n);
  if (threadIdx.x == 0) loss[0] = (float)(s / (double)wsum[0]);
}

template <typename T>
__global__ void __launch_bounds__(FFA_EW_THREADS)
softmax_ce_kernel(const T* __restrict__ logits, const uint8_t* __restrict__ tgt, const float* __restrict__ w,
                  const float* __restrict__ wsum, const float* __restrict__ grad_scale, T* __restrict__ dlogits,
                  uint8_t* __restrict__ pred, float* __restrict__ parts, long long npix, int K, int Cp) {
  __shared__ float red[FFA_EW_THREADS / 64];
  float lsum = 0.f;
  float gs = 0.f;
  if (dlogits) gs = grad_scale[0] / wsum[0];
  const int nv = Cp / 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    float z[FFA_CE_MAXK];
#pragma unroll
    for (int v = 0; v < FFA_CE_MAXK / 8; ++v) {
      if (v < nv) {
        float tmp[8];
        ffa_load8<T>(logits + i * Cp + v * 8, tmp);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[v * 8 + e] = tmp[e];
      }
    }
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int k = 0; k < FFA_CE_MAXK; ++k) {
      if (k < K && z[k] > m) {  // strict '>' keeps the lowest index on ties (torch.argmax)
        m = z[k];
        am = k;
      }
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < FFA_CE_MAXK; ++k) {
      if (k < K) {
        z[k] = __expf(z[k] - m);
        se += z[k];
      }
    }
    const int t = tgt[i];
    const float wt = (t < K) ? w[t] : 0.f;
    const float inv = 1.f / se;
    // -log softmax[t] = log(se) - (z_t - m); z[k] now holds exp(z_k - m)
    float zt = 1.f;
#pragma unroll
    for (int k = 0; k < FFA_CE_MAXK; ++k)
      if (k == t) zt = z[k];
    if (wt != 0.f) lsum += wt * (__logf(se) - __logf(zt));
    if (pred) pred[i] = (uint8_t)am;
    if (dlogits) {
      const float c = wt * gs;
#pragma unroll
      for (int v = 0; v < FFA_CE_MAXK / 8; ++v) {
        if (v < nv) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = v * 8 + e;
            o[e] = (k < K) ? c * (z[k] * inv - (k == t ? 1.f : 0.f)) : 0.f;
          }
          ffa_store8<T>(dlogits + i * Cp + v * 8, o);
        }
      }
    }
  }
  lsum = ffa_wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int k = 0; k < FFA_EW_THREADS / 64; ++k) tot += red[k];
    parts[blockIdx.x] = tot;
  }
}

// The same per-pixel arithmetic with the tensor traffic staged through LDS (the default; FFA_CE_TILED=0 selects the
// kernel above).  A thread that owns a pixel reads its 64 bytes as four 16-byte loads 64 bytes apart, so one wave
// instruction touches 32 cache lines for 1 KB -- and writes dlogits the same way.  Here a block moves a tile of 256
// pixels as whole 16-byte pieces in memory order (lane l <-> piece l: 1 KB per wave instruction, 8 lines), parks them in
// LDS at an odd pitch (conflict-free ds_read_b128 when thread = pixel), each thread then works on its own pixel in
// place, and the gradient leaves LDS in memory order again.  The next tile's pieces are requested before the current
// tile's arithmetic.  Thread <-> pixel assignment, the arithmetic and the order of the loss partial sums are those of
// softmax_ce_kernel: results are bit-identical.
// SUMS: the kernel also leaves, per block, the per-class sums of the gradient values it stores (as stored: rounded to
// T) in parts_b[block][FFA_CE_MAXK] -- the bias gradient of the convolution that produced the logits is the column
// sum of dlogits, otherwise a separate pass over the whole tensor (ffa_channel_sums: 537 MB at batch 32 x 512^2).
template <typename T, bool SUMS>
__global__ void __launch_bounds__(FFA_EW_THREADS)
softmax_ce_tiled_kernel(const T* __restrict__ logits, const uint8_t* __restrict__ tgt, const float* __restrict__ w,
                        const float* __restrict__ wsum, const float* __restrict__ grad_scale, T* __restrict__ dlogits,
                        uint8_t* __restrict__ pred, float* __restrict__ parts, float* __restrict__ parts_b,
                        long long npix, int K, int Cp) {
  constexpr int EPP = 16 / (int)sizeof(T);   // elements per 16-byte piece
  constexpr int MAXP = FFA_CE_MAXK / EPP;    // pieces per pixel at the largest pitch: 4 (bf16) / 8 (f32)
  __shared__ __align__(16) unsigned char tile[FFA_EW_THREADS * (MAXP + 1) * 16];
  __shared__ float red[FFA_EW_THREADS / 64];
  const int tid = threadIdx.x;
  const int np = Cp / EPP;   // pieces per pixel
  const int slots = np | 1;  // LDS pitch of a pixel in 16-byte slots: odd
  // SUMS: FFA_EW_THREADS % np == 0 (the launcher checks), so every piece a thread moves out in the store phase holds
  // the same EPP classes, (tid % np) * EPP ..: EPP running sums per thread
  float bsum[SUMS ? EPP : 1];
#pragma unroll
  for (int k = 0; k < (SUMS ? EPP : 1); ++k) bsum[k] = 0.f;
  float lsum = 0.f;
  float gs = 0.f;
  if (dlogits) gs = grad_scale[0] / wsum[0];
  const long long ntiles = (npix + FFA_EW_THREADS - 1) / FFA_EW_THREADS;
  int loff[MAXP];  // LDS byte offset of piece tid + 256 k of a tile
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int idx = tid + FFA_EW_THREADS * k;
    loff[k] = ((idx / np) * slots + idx % np) * 16;
  }
  uint4 pre[MAXP];
  auto gload = [&](long long ti) {
    const long long base = ti * FFA_EW_THREADS;
    const long long left = npix - base;
    const int npc = (int)(left < FFA_EW_THREADS ? left : FFA_EW_THREADS) * np;
    const uint4* src = reinterpret_cast<const uint4*>(logits + base * Cp);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int idx = tid + FFA_EW_THREADS * k;
      if (k < np) pre[k] = idx < npc ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  long long ti = blockIdx.x;
  if (ti < ntiles) gload(ti);
  for (; ti < ntiles; ti += gridDim.x) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k)
      if (k < np) *reinterpret_cast<uint4*>(tile + loff[k]) = pre[k];
    __syncthreads();
    if (ti + gridDim.x < ntiles) gload(ti + gridDim.x);
    const long long i = ti * FFA_EW_THREADS + tid;
    if (i < npix) {
      unsigned char* mine = tile + tid * slots * 16;
      float z[FFA_CE_MAXK];
#pragma unroll
      for (int v = 0; v < MAXP; ++v) {
        if (v < np) {
          const uint4 u = *reinterpret_cast<const uint4*>(mine + v * 16);
          if constexpr (sizeof(T) == 2) {
            z[v * 8 + 0] = __uint_as_float(u.x << 16);
            z[v * 8 + 1] = __uint_as_float(u.x & 0xffff0000u);
            z[v * 8 + 2] = __uint_as_float(u.y << 16);
            z[v * 8 + 3] = __uint_as_float(u.y & 0xffff0000u);
            z[v * 8 + 4] = __uint_as_float(u.z << 16);
            z[v * 8 + 5] = __uint_as_float(u.z & 0xffff0000u);
            z[v * 8 + 6] = __uint_as_float(u.w << 16);
            z[v * 8 + 7] = __uint_as_float(u.w & 0xffff0000u);
          } else {
            z[v * 4 + 0] = __uint_as_float(u.x);
            z[v * 4 + 1] = __uint_as_float(u.y);
            z[v * 4 + 2] = __uint_as_float(u.z);
            z[v * 4 + 3] = __uint_as_float(u.w);
          }
        }
      }
      float m = -INFINITY;
      int am = 0;
#pragma unroll
      for (int k = 0; k < FFA_CE_MAXK; ++k) {
        if (k < K && z[k] > m) {  // strict '>' keeps the lowest index on ties (torch.argmax)
          m = z[k];
          am = k;
        }
      }
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < FFA_CE_MAXK; ++k) {
        if (k < K) {
          z[k] = __expf(z[k] - m);
          se += z[k];
        }
      }
      const int t = tgt[i];
      const float wt = (t < K) ? w[t] : 0.f;
      const float inv = 1.f / se;
      float zt = 1.f;
#pragma unroll
      for (int k = 0; k < FFA_CE_MAXK; ++k)
        if (k == t) zt = z[k];
      if (wt != 0.f) lsum += wt * (__logf(se) - __logf(zt));
      if (pred) pred[i] = (uint8_t)am;
      if (dlogits) {
        const float c = wt * gs;
#pragma unroll
        for (int v = 0; v < MAXP; ++v) {
          if (v < np) {
            float o[EPP];
#pragma unroll
            for (int e = 0; e < EPP; ++e) {
              const int k = v * EPP + e;
              o[e] = (k < K) ? c * (z[k] * inv - (k == t ? 1.f : 0.f)) : 0.f;
            }
            uint4 u;
            if constexpr (sizeof(T) == 2) {
              u.x = ffa_pack_bf16x2(o[0], o[1]);
              u.y = ffa_pack_bf16x2(o[2], o[3]);
              u.z = ffa_pack_bf16x2(o[4], o[5]);
              u.w = ffa_pack_bf16x2(o[6], o[7]);
            } else {
              u = make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
            }
            *reinterpret_cast<uint4*>(mine + v * 16) = u;
          }
        }
      }
    }
    if (dlogits) {
      __syncthreads();
      const long long base = ti * FFA_EW_THREADS;
      const long long left = npix - base;
      const int npc = (int)(left < FFA_EW_THREADS ? left : FFA_EW_THREADS) * np;
      uint4* dst = reinterpret_cast<uint4*>(dlogits + base * Cp);
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        const int idx = tid + FFA_EW_THREADS * k;
        if (k < np && idx < npc) {
          const uint4 u = *reinterpret_cast<const uint4*>(tile + loff[k]);
          dst[idx] = u;
          if constexpr (SUMS) {  // the values as the consumers of dlogits will read them
            if constexpr (sizeof(T) == 2) {
              bsum[0] += __uint_as_float(u.x << 16);
              bsum[1] += __uint_as_float(u.x & 0xffff0000u);
              bsum[2] += __uint_as_float(u.y << 16);
              bsum[3] += __uint_as_float(u.y & 0xffff0000u);
              bsum[4] += __uint_as_float(u.z << 16);
              bsum[5] += __uint_as_float(u.z & 0xffff0000u);
              bsum[6] += __uint_as_float(u.w << 16);
              bsum[7] += __uint_as_float(u.w & 0xffff0000u);
            } else {
              bsum[0] += __uint_as_float(u.x);
              bsum[1] += __uint_as_float(u.y);
              bsum[2] += __uint_as_float(u.z);
              bsum[3] += __uint_as_float(u.w);
            }
          }
        }
      }
    }
    __syncthreads();  // the tile buffer is free for the next fill
  }
  lsum = ffa_wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  if constexpr (SUMS) {  // lanes l, l + np, l + 2 np, ... of a wave hold the same classes (np is a power of two <= 8)
    float* wsums = reinterpret_cast<float*>(tile);  // the tile buffer is free after the loop's last barrier
    if (threadIdx.x < (FFA_EW_THREADS / 64) * FFA_CE_MAXK) wsums[threadIdx.x] = 0.f;  // classes beyond the pitch
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPP; ++e) {
      float t = bsum[e];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
        if (o >= np) t += __shfl_xor(t, o, 64);
      if ((threadIdx.x & 63) < np) wsums[(threadIdx.x >> 6) * FFA_CE_MAXK + (threadIdx.x & 63) * EPP + e] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int k = 0; k < FFA_EW_THREADS / 64; ++k) tot += red[k];
    parts[blockIdx.x] = tot;
  }
  if constexpr (SUMS) {
    if (threadIdx.x < FFA_CE_MAXK) {
      const float* wsums = reinterpret_cast<const float*>(tile);
      float t = 0.f;
      for (int k = 0; k < FFA_EW_THREADS / 64; ++k) t += wsums[k * FFA_CE_MAXK + threadIdx.x];
      parts_b[(size_t)blockIdx.x * FFA_CE_MAXK + threadIdx.x] = t;
    }
  }
}

// dlogit_sums[c] = fixed-order sum over the blocks' partial rows: one block per class, thread l adds rows l, l + 256,
// ... in double (all loads in flight), thread 0 then adds the 256 thread sums in order
__global__ void __launch_bounds__(256) ce_finalize_cols_kernel(const float* __restrict__ parts_b, int nb, int Cp,
                                                               float* __restrict__ out) {
  __shared__ double sh[256];
  const int k = blockIdx.x;
  float v[FFA_CE_BLOCKS / 256];
#pragma unroll
  for (int j = 0; j < FFA_CE_BLOCKS / 256; ++j) {
    const int i = threadIdx.x + 256 * j;
    v[j] = i < nb ? parts_b[(size_t)i * FFA_CE_MAXK + k] : 0.f;
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < FFA_CE_BLOCKS / 256; ++j) s += (double)v[j];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int j = 0; j < 256; ++j) t += sh[j];
    out[k] = (float)t;
  }
}

extern "C" long long ffa_softmax_ce_workspace_bytes(void) {
  // block partials of the weight sum and of the loss, then [FFA_CE_BLOCKS][FFA_CE_MAXK] per-class gradient sums
  return (long long)(2 * FFA_CE_BLOCKS + 4 + FFA_CE_BLOCKS * FFA_CE_MAXK) * sizeof(float);
}

// loss[0] = weighted-mean CE; wsum_out[0] = sum of target weights; optional dlogits (scaled by
// grad_scale[0], a device scalar, default 1 when null is not allowed -> pass a device 1.0f) and pred.
extern "C" int ffa_softmax_ce_sums(int dtype, const void* logits, const uint8_t* targets, const float* class_weights,
                                   const float* grad_scale, float* loss, float* wsum_out, void* dlogits, uint8_t* pred,
                                   float* dlogit_sums, long long npix, int K, int Cp, void* workspace,
                                   long long workspace_bytes, hipStream_t stream);

extern "C" int ffa_softmax_ce(int dtype, const void* logits, const uint8_t* targets, const float* class_weights,
                              const float* grad_scale, float* loss, float* wsum_out, void* dlogits, uint8_t* pred,
                              long long npix, int K, int Cp, void* workspace, long long workspace_bytes,
                              hipStream_t stream) {
  return ffa_softmax_ce_sums(dtype, logits, targets, class_weights, grad_scale, loss, wsum_out, dlogits, pred, nullptr,
                             npix, K, Cp, workspace, workspace_bytes, stream);
}

// ffa_softmax_ce that also returns dlogit_sums[Cp] = per-class sums over all pixels of the dlogits values it wrote (the
// bias gradient of the layer that produced the logits, for the same upstream gradient of grad_scale[0]); needs dlogits
// and 16-byte aligned tensors, FFA_ERR_UNSUPPORTED otherwise (callers then take ffa_channel_sums over dlogits).
extern "C" int ffa_softmax_ce_sums(int dtype, const void* logits, const uint8_t* targets, const float* class_weights,
                                   const float* grad_scale, float* loss, float* wsum_out, void* dlogits, uint8_t* pred,
                                   float* dlogit_sums, long long npix, int K, int Cp, void* workspace,
                                   long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(logits && targets && class_weights && loss && wsum_out && workspace, "softmax_ce: null pointer");
  FFA_REQUIRE(K >= 1 && K <= FFA_CE_MAXK && Cp % 8 == 0 && Cp >= K && Cp <= FFA_CE_MAXK,
              "softmax_ce: unsupported class count %d (pitch %d)", K, Cp);
  FFA_REQUIRE(!dlogits || grad_scale, "softmax_ce: dlogits needs grad_scale");
  if (workspace_bytes < ffa_softmax_ce_workspace_bytes()) {
    ffa_set_error("softmax_ce: workspace too small");
    return FFA_ERR_WORKSPACE;
  }
  float* parts_w = static_cast<float*>(workspace);
  float* parts_l = parts_w + FFA_CE_BLOCKS;
  long long nb = (npix + FFA_EW_THREADS - 1) / FFA_EW_THREADS;
  if (nb > FFA_CE_BLOCKS) nb = FFA_CE_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(ce_weight_sum_kernel, dim3((int)nb), dim3(FFA_EW_THREADS), 0, stream, targets, class_weights, K,
                     npix, parts_w, (reinterpret_cast<uintptr_t>(targets) & 15) == 0 ? 1 : 0);
  hipLaunchKernelGGL(ce_finalize_sum_kernel, dim3(1), dim3(64), 0, stream, parts_w, (int)nb, wsum_out);
  const char* ct = getenv("FFA_CE_TILED");  // A/B switch, read per call (the tests flip it)
  const bool tiled = !(ct && ct[0] == '0');
  const bool aligned = ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
  float* parts_b = parts_l + FFA_CE_BLOCKS + 4;
  const int pieces = Cp * (dtype == FFA_BF16 ? 2 : 4) / 16;  // 16-byte pieces per pixel
  if (dlogit_sums && !(tiled && aligned && dlogits && FFA_EW_THREADS % pieces == 0)) {
    ffa_set_error("softmax_ce_sums: needs dlogits, 16-byte aligned tensors, the tiled kernel and a pitch of 1, 2, 4 or 8 pieces");
    return FFA_ERR_UNSUPPORTED;
  }
  if (tiled && aligned && dlogit_sums) {
    if (dtype == FFA_BF16)
      hipLaunchKernelGGL((softmax_ce_tiled_kernel<ffa_bf16, true>), dim3((int)nb), dim3(FFA_EW_THREADS), 0, stream,
                         (const ffa_bf16*)logits, targets, class_weights, wsum_out, grad_scale, (ffa_bf16*)dlogits, pred,
                         parts_l, parts_b, npix, K, Cp);
    else
      hipLaunchKernelGGL((softmax_ce_tiled_kernel<float, true>), dim3((int)nb), dim3(FFA_EW_THREADS), 0, stream,
                         (const float*)logits, targets, class_weights, wsum_out, grad_scale, (float*)dlogits, pred,
                         parts_l, parts_b, npix, K, Cp);
    hipLaunchKernelGGL(ce_finalize_cols_kernel, dim3(Cp), dim3(256), 0, stream, parts_b, (int)nb, Cp, dlogit_sums);
  } else if (tiled && aligned) {
    if (dtype == FFA_BF16)
      hipLaunchKernelGGL((softmax_ce_tiled_kernel<ffa_bf16, false>), dim3((int)nb), dim3(FFA_EW_THREADS), 0, stream,
                         (const ffa_bf16*)logits, targets, class_weights, wsum_out, grad_scale, (ffa_bf16*)dlogits, pred,
                         parts_l, nullptr, npix, K, Cp);
    else
      hipLaunchKernelGGL((softmax_ce_tiled_kernel<float, false>), dim3((int)nb), dim3(FFA_EW_THREADS), 0, stream,
                         (const float*)logits, targets, class_weights, wsum_out, grad_scale, (float*)dlogits, pred,
                         parts_l, nullptr, npix, K, Cp);
  } else if (dtype == FFA_BF16)
    hipLaunchKernelGGL(softmax_ce_kernel<ffa_bf16>, dim3((int)nb), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)logits, targets, class_weights, wsum_out, grad_scale, (ffa_bf16*)dlogits, pred,
                       parts_l, npix, K, Cp);
  else
    hipLaunchKernelGGL(softmax_ce_kernel<float>, dim3((int)nb), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)logits, targets, class_weights, wsum_out, grad_scale, (float*)dlogits, pred,
                       parts_l, npix, K, Cp);
  hipLaunchKernelGGL(ce_finalize_loss_kernel, dim3(1), dim3(64), 0, stream, parts_l, (int)nb, wsum_out, loss);
  return ffa_check_launch("softmax_ce");
}

// x *= scale[0] (device scalar), skipped entirely when the scalar is exactly 1: the loss forward already wrote
// dlogits for an upstream gradient of 1, which is what loss.backward() supplies unless the caller scales the
// loss -- the usual step then costs one 4-byte read per block instead of a second pass over the logits.
template <typename T>
__global__ void scale_inplace_kernel(T* __restrict__ x, long long nvec, const float* __restrict__ scale) {
  const float s = scale[0];
  if (s == 1.f) return;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    ffa_load8<T>(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= s;
    ffa_store8<T>(x + i * 8, v);
  }
}

extern "C" int ffa_scale_inplace(int dtype, void* x, long long n, const float* scale, hipStream_t stream) {
  FFA_REQUIRE(x && scale && n % 8 == 0, "scale_inplace: bad arguments");
  const long long nvec = n / 8;
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(scale_inplace_kernel<ffa_bf16>, dim3(ew_grid(nvec)), dim3(FFA_EW_THREADS), 0, stream,
                       (ffa_bf16*)x, nvec, scale);
  else
    hipLaunchKernelGGL(scale_inplace_kernel<float>, dim3(ew_grid(nvec)), dim3(FFA_EW_THREADS), 0, stream, (float*)x,
                       nvec, scale);
  return ffa_check_launch("scale_inplace");
}

// ------------------------------------------------------------------------------------------------
// prediction conversion for the zonal tile loop

template <typename T>
__global__ void argmax_crop_kernel(const T* __restrict__ logits, uint8_t* __restrict__ out, int B, int H, int W, int K,
                                   int Cp, int y0, int x0, int h, int w) {
  const long long total = (long long)B * h * w;
  const int nv = Cp / 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    long long p = i / w;
    const int y = (int)(p % h);
    const long long b = p / h;
    const T* src = logits + ((b * H + y0 + y) * W + x0 + x) * Cp;
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int v = 0; v < FFA_CE_MAXK / 8; ++v) {
      if (v < nv) {
        float tmp[8];
        ffa_load8<T>(src + v * 8, tmp);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = v * 8 + e;
          if (k < K && tmp[e] > m) {
            m = tmp[e];
            am = k;
          }
        }
      }
    }
    out[i] = (uint8_t)am;
  }
}

template <typename T>
__global__ void class_prob_crop_kernel(const T* __restrict__ logits, uint8_t* __restrict__ out, int B, int H, int W,
                                       int K, int Cp, int y0, int x0, int h, int w) {
  // out is [B][K][h][w] uint8 = rint(softmax * 255), rint = round-half-even like numpy.round
  const long long total = (long long)B * h * w;
  const int nv = Cp / 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    long long p = i / w;
    const int y = (int)(p % h);
    const long long b = p / h;
    const T* src = logits + ((b * H + y0 + y) * W + x0 + x) * Cp;
    float z[FFA_CE_MAXK];
#pragma unroll
    for (int v = 0; v < FFA_CE_MAXK / 8; ++v) {
      if (v < nv) {
        float tmp[8];
        ffa_load8<T>(src + v * 8, tmp);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[v * 8 + e] = tmp[e];
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < FFA_CE_MAXK; ++k)
      if (k < K) m = fmaxf(m, z[k]);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < FFA_CE_MAXK; ++k)
      if (k < K) {
        z[k] = expf(z[k] - m);
        se += z[k];
      }
#pragma unroll
    for (int k = 0; k < FFA_CE_MAXK; ++k)
      if (k < K) out[((b * K + k) * h + y) * (long long)w + x] = (uint8_t)rintf(z[k] / se * 255.f);
  }
}

// mode 0: argmax -> out [B][h][w]; mode 1: class_prob -> out [B][K][h][w]
extern "C" int ffa_predict_u8(int dtype, int mode, const void* logits, uint8_t* out, int B, int H, int W, int K,
                              int Cp, int y0, int x0, int h, int w, hipStream_t stream) {
  FFA_REQUIRE(logits && out, "predict_u8: null pointer");
  FFA_REQUIRE(K >= 1 && K <= FFA_CE_MAXK && Cp % 8 == 0 && Cp >= K && Cp <= FFA_CE_MAXK,
              "predict_u8: unsupported class count %d (pitch %d)", K, Cp);
  FFA_REQUIRE(y0 >= 0 && x0 >= 0 && h > 0 && w > 0 && y0 + h <= H && x0 + w <= W, "predict_u8: crop outside the tile");
  FFA_REQUIRE(mode == 0 || mode == 1, "predict_u8: unknown mode %d", mode);
  const long long items = (long long)B * h * w;
#define FFA_PRED(KER, TT) \
  hipLaunchKernelGGL(KER<TT>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream, (const TT*)logits, out, B, H, W, K, Cp, y0, x0, h, w)
  if (mode == 0) {
    if (dtype == FFA_BF16) FFA_PRED(argmax_crop_kernel, ffa_bf16); else FFA_PRED(argmax_crop_kernel, float);
  } else {
    if (dtype == FFA_BF16) FFA_PRED(class_prob_crop_kernel, ffa_bf16); else FFA_PRED(class_prob_crop_kernel, float);
  }
#undef FFA_PRED
  return ffa_check_launch("predict_u8");
}

__global__ void onehot_to_index_kernel(const float* __restrict__ onehot, uint8_t* __restrict__ idx, int B, int K,
                                       long long hw) {
  // NCHW one-hot (the reference's label format, flair_hub/data/utils_data/label.py:3-14) -> class index,
  // first maximum wins like torch.argmax
  const long long total = (long long)B * hw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / hw, p = i % hw;
    float m = -INFINITY;
    int am = 0;
    for (int k = 0; k < K; ++k) {
      const float v = onehot[(b * K + k) * hw + p];
      if (v > m) {
        m = v;
        am = k;
      }
    }
    idx[i] = (uint8_t)am;
  }
}

extern "C" int ffa_onehot_to_index(const float* onehot, uint8_t* idx, int B, int K, int H, int W, hipStream_t stream) {
  FFA_REQUIRE(onehot && idx && K >= 1 && K <= 255, "onehot_to_index: bad arguments");
  const long long items = (long long)B * H * W;
  hipLaunchKernelGGL(onehot_to_index_kernel, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream, onehot, idx, B, K,
                     (long long)H * W);
  return ffa_check_launch("onehot_to_index");
}

// ------------------------------------------------------------------------------------------------
// K x K confusion matrix of uint8 predictions vs uint8 targets, accumulated into int64 counts
// (rows = target, columns = prediction): the state behind the IoU metrics that the reference's
// training_step / validation_step update every batch (flair_hub/tasks/tasks_module.py:210-212, :273-275 via
// torchmetrics).  Block-local histogram in LDS, one integer atomic per non-empty bin per block: exact,
// order independent, no host synchronisation (torch.bincount needs one, which would break graph capture).

__global__ void __launch_bounds__(FFA_EW_THREADS)
confusion_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ tgt, long long n, int K,
                 unsigned long long* __restrict__ out, int vec16) {
  __shared__ unsigned int hist[FFA_CE_MAXK * FFA_CE_MAXK];
  const int bins = K * K;
  for (int i = threadIdx.x; i < bins; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long done = 0;
  if (vec16) {  // both byte streams 16-byte aligned: 16 pixels per pair of loads (one byte per lane per load: 38 us)
    const long long n16 = n / 16;
    for (long long i = gid; i < n16; i += stride) {
      const uint4 tv = reinterpret_cast<const uint4*>(tgt)[i];
      const uint4 pv = reinterpret_cast<const uint4*>(pred)[i];
      const uint32_t tq[4] = {tv.x, tv.y, tv.z, tv.w}, pq[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int t = (tq[j] >> (8 * b)) & 0xff, p = (pq[j] >> (8 * b)) & 0xff;
          if (t < K && p < K) atomicAdd(&hist[t * K + p], 1u);
        }
    }
    done = n16 * 16;
  }
  for (long long i = done + gid; i < n; i += stride) {
    const int t = tgt[i], p = pred[i];
    if (t < K && p < K) atomicAdd(&hist[t * K + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += blockDim.x)
    if (hist[i]) atomicAdd(&out[i], (unsigned long long)hist[i]);
}

extern "C" int ffa_confusion_matrix(const uint8_t* pred, const uint8_t* target, long long n, int K, long long* counts,
                                    hipStream_t stream) {
  FFA_REQUIRE(pred && target && counts && K >= 1 && K <= FFA_CE_MAXK, "confusion_matrix: bad arguments");
  long long nb = (n + FFA_EW_THREADS * 16 - 1) / (FFA_EW_THREADS * 16);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  const int vec16 = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target)) & 15) == 0 ? 1 : 0;
  hipLaunchKernelGGL(confusion_kernel, dim3((int)nb), dim3(FFA_EW_THREADS), 0, stream, pred, target, n, K,
                     reinterpret_cast<unsigned long long*>(counts), vec16);
  return ffa_check_launch("confusion_matrix");
}
