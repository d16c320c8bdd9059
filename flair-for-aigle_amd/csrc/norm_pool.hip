// BatchNorm2d (train / eval, forward + backward), MaxPool2d 3x3 s2 p1 (forward + backward) and the
// NCHW f32 <-> NHWC layout hand-over, all NHWC, bf16 or f32 storage with f32 arithmetic.
//
// These replace the ATen batch_norm / relu / max_pool2d calls that segmentation_models_pytorch's
// ResNet-34 encoder and UnetDecoder make (reached from flair_hub/models/flair_model.py:376 and
// :417-419; BN eps 1e-5, momentum 0.1).  All are HBM-bound: every thread moves
// 8 consecutive channels (16 B of bf16) per access, per-channel reductions are two-stage
// (per-block partials in a caller workspace, then a fixed-order finalize) so results are
// bitwise reproducible run to run -- no float atomics.
#include "ffa_common.h"

#define FFA_EW_THREADS 256
#define FFA_MAX_PARTIALS 1024
#ifndef FFA_EW_UNROLL
#define FFA_EW_UNROLL 4
#endif

static inline int ew_grid(long long items) {
  long long g = (items + FFA_EW_THREADS - 1) / FFA_EW_THREADS;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// grid for the per-channel streaming kernels: (grid * FFA_EW_THREADS) % (C / 8) == 0, so that a thread of the
// grid-stride loop stays on one channel group
static inline int ew_grid_c(long long nvec, int C) {
  int a = C / 8, b = FFA_EW_THREADS;
  while (b) {  // a = gcd(C / 8, FFA_EW_THREADS)
    const int r = a % b;
    a = b;
    b = r;
  }
  const int m = (C / 8) / a;
  const int g = ew_grid((nvec + FFA_EW_UNROLL - 1) / FFA_EW_UNROLL) / m * m;  // a thread moves FFA_EW_UNROLL vectors per iteration
  return g < m ? m : g;
}

// ------------------------------------------------------------------------------------------------
// layout hand-over at the model boundary

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W,
                                    int Cp) {
  // one thread per (pixel, 8-channel group): reads are coalesced along x within each plane
  const int groups = Cp / 8;
  const long long total = (long long)B * H * W * groups;
  const long long hw = (long long)H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    // pixel fastest so that neighbouring threads read neighbouring x of one plane
    const long long pix = i % ((long long)B * hw);
    const int g = (int)(i / ((long long)B * hw));
    const long long b = pix / hw, p = pix % hw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = g * 8 + e;
      v[e] = (c < C) ? src[(b * C + c) * hw + p] : 0.f;
    }
    ffa_store8<T>(dst + pix * Cp + g * 8, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W,
                                    int Cp) {
  const long long hw = (long long)H * W;
  const long long total = (long long)B * C * hw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long p = i % hw;
    const long long bc = i / hw;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    dst[i] = ffa_load_elem<T>(src + (b * hw + p) * Cp + c);
  }
}

// Same conversion, four consecutive pixels per thread: 16-byte loads from every source plane, Cp * 4 contiguous
// output elements per thread.  For few-channel inputs (the 5-band aerial tiles, pitch 16) the one-group-per-thread
// kernel spends half its threads writing pad zeros and reads 4 bytes per lane: 2.3 TB/s of traffic; this one streams.
template <typename T, int CP>
__global__ void __launch_bounds__(FFA_EW_THREADS)
nchw_to_nhwc_x4_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, long long hw) {
  const long long quads = (long long)B * hw / 4;
  // bf16: the 4 pixels x CP channels of a thread are 128 contiguous bytes, so one store instruction of a wave would put
  // 16 bytes on each of 64 different lines.  The block's 32 KB leave through LDS instead (pitch 9 pieces per thread:
  // conflict-free both ways) as whole 16-byte pieces in memory order, 1 KB per wave instruction.
  constexpr bool STAGE = sizeof(T) == 2 && CP == 16;
  constexpr int NP = 4 * CP * (int)sizeof(T) / 16;  // 16-byte pieces per thread
  __shared__ __align__(16) unsigned char stage[STAGE ? FFA_EW_THREADS * (NP + 1) * 16 : 16];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i0 = blockIdx.x * (long long)blockDim.x; i0 < quads; i0 += stride) {
    const long long i = i0 + threadIdx.x;
    const bool live = i < quads;
    const long long pix = i * 4;
    float v[4][CP];
    if (live) {
      const long long b = pix / hw, p = pix % hw;
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) q = *reinterpret_cast<const float4*>(src + (b * C + c) * hw + p);
        v[0][c] = q.x;
        v[1][c] = q.y;
        v[2][c] = q.z;
        v[3][c] = q.w;
      }
    }
    if constexpr (STAGE) {
      if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int g = 0; g < CP / 8; ++g) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[k][g * 8 + e];
            ffa_store8<T>(reinterpret_cast<T*>(stage + (threadIdx.x * (NP + 1) + k * (CP / 8) + g) * 16), o);
          }
      }
      __syncthreads();
      const long long left = quads - i0;
      const int npc = (int)(left < FFA_EW_THREADS ? left : FFA_EW_THREADS) * NP;
      uint4* out = reinterpret_cast<uint4*>(dst + i0 * 4 * CP);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int idx = threadIdx.x + FFA_EW_THREADS * k;
        if (idx < npc) out[idx] = *reinterpret_cast<const uint4*>(stage + ((idx / NP) * (NP + 1) + idx % NP) * 16);
      }
      __syncthreads();
    } else if (live) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int g = 0; g < CP / 8; ++g) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = v[k][g * 8 + e];
          ffa_store8<T>(dst + (pix + k) * CP + g * 8, o);
        }
    }
  }
}

extern "C" int ffa_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int Cp,
                                hipStream_t stream) {
  FFA_REQUIRE(src && dst && Cp % 8 == 0 && Cp >= C, "nchw_to_nhwc: bad arguments (C=%d Cp=%d)", C, Cp);
  if (Cp == 16 && ((long long)H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const long long quads = (long long)B * H * W / 4;
    if (dtype == FFA_BF16)
      hipLaunchKernelGGL((nchw_to_nhwc_x4_kernel<ffa_bf16, 16>), dim3(ew_grid(quads)), dim3(FFA_EW_THREADS), 0, stream, src,
                         (ffa_bf16*)dst, B, C, (long long)H * W);
    else
      hipLaunchKernelGGL((nchw_to_nhwc_x4_kernel<float, 16>), dim3(ew_grid(quads)), dim3(FFA_EW_THREADS), 0, stream, src,
                         (float*)dst, B, C, (long long)H * W);
    return ffa_check_launch("nchw_to_nhwc");
  }
  const long long items = (long long)B * H * W * (Cp / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<ffa_bf16>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream, src,
                       (ffa_bf16*)dst, B, C, H, W, Cp);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream, src,
                       (float*)dst, B, C, H, W, Cp);
  return ffa_check_launch("nchw_to_nhwc");
}

// Raw NCHW raster tiles (uint8 / uint16 / int16 / float32 samples) -> normalised NHWC compute tensor in one pass:
// dst = (src - mean[c]) / std[c] (the 'custom' normalisation of flair_hub/data/utils_data/norm.py:37-44; 'scaling' is
// mean 0, std 255).  The zonal loop then ships the raster's own sample size over PCIe (1 byte for the aerial
// mosaics) instead of 4 and does no per-tile float work on the host.
template <typename S, typename T>
__global__ void raw_nchw_to_nhwc_kernel(const S* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W,
                                        int Cp, const float* __restrict__ mean, const float* __restrict__ stdv) {
  const int groups = Cp / 8;
  const long long hw = (long long)H * W;
  const long long total = (long long)B * hw * groups;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i % ((long long)B * hw);
    const int g = (int)(i / ((long long)B * hw));
    const long long b = pix / hw, p = pix % hw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = g * 8 + e;
      v[e] = (c < C) ? ((float)src[(b * C + c) * hw + p] - mean[c]) / stdv[c] : 0.f;
    }
    ffa_store8<T>(dst + pix * Cp + g * 8, v);
  }
}

template <typename S>
static int raw_layout_launch(int dtype, const void* src, void* dst, int B, int C, int H, int W, int Cp,
                             const float* mean, const float* stdv, hipStream_t stream) {
  const long long items = (long long)B * H * W * (Cp / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL((raw_nchw_to_nhwc_kernel<S, ffa_bf16>), dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const S*)src, (ffa_bf16*)dst, B, C, H, W, Cp, mean, stdv);
  else
    hipLaunchKernelGGL((raw_nchw_to_nhwc_kernel<S, float>), dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const S*)src, (float*)dst, B, C, H, W, Cp, mean, stdv);
  return ffa_check_launch("raw_nchw_to_nhwc");
}

extern "C" int ffa_u8_nchw_to_nhwc(int dtype, const uint8_t* src, void* dst, int B, int C, int H, int W, int Cp,
                                   const float* mean, const float* stdv, hipStream_t stream) {
  FFA_REQUIRE(src && dst && mean && stdv && Cp % 8 == 0 && Cp >= C, "u8_nchw_to_nhwc: bad arguments (C=%d Cp=%d)", C, Cp);
  return raw_layout_launch<uint8_t>(dtype, src, dst, B, C, H, W, Cp, mean, stdv, stream);
}

// The same for the other sample types rasters come in: src_kind FFA_SRC_U8 (0), FFA_SRC_U16 (1: SPOT, Sentinel
// reflectances), FFA_SRC_I16 (2), FFA_SRC_F32 (3: elevation models).
extern "C" int ffa_raw_nchw_to_nhwc(int dtype, int src_kind, const void* src, void* dst, int B, int C, int H, int W,
                                    int Cp, const float* mean, const float* stdv, hipStream_t stream) {
  FFA_REQUIRE(src && dst && mean && stdv && Cp % 8 == 0 && Cp >= C, "raw_nchw_to_nhwc: bad arguments (C=%d Cp=%d)", C,
              Cp);
  switch (src_kind) {
    case 0: return raw_layout_launch<uint8_t>(dtype, src, dst, B, C, H, W, Cp, mean, stdv, stream);
    case 1: return raw_layout_launch<uint16_t>(dtype, src, dst, B, C, H, W, Cp, mean, stdv, stream);
    case 2: return raw_layout_launch<int16_t>(dtype, src, dst, B, C, H, W, Cp, mean, stdv, stream);
    case 3: return raw_layout_launch<float>(dtype, src, dst, B, C, H, W, Cp, mean, stdv, stream);
  }
  ffa_set_error("raw_nchw_to_nhwc: unknown source sample kind %d", src_kind);
  return FFA_ERR_ARG;
}

extern "C" int ffa_nhwc_to_nchw(int dtype, const void* src, float* dst, int B, int C, int H, int W, int Cp,
                                hipStream_t stream) {
  FFA_REQUIRE(src && dst && Cp >= C, "nhwc_to_nchw: bad arguments");
  const long long items = (long long)B * C * H * W;
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<ffa_bf16>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)src, dst, B, C, H, W, Cp);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_grid(items)), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)src, dst, B, C, H, W, Cp);
  return ffa_check_launch("nhwc_to_nchw");
}

// ------------------------------------------------------------------------------------------------
// per-channel two-value reductions over [N pixels][C]:  thread t owns channel group t % CG and
// pixel lane t / CG; block partials land in ws[block][2][C]

// relu modes of the backward kernels: 0 = none, 1 = mask from the stored forward output y (needed when a residual
// was added before the ReLU), 2 = mask recomputed from x as (x*scale + shift > 0) with the same fma the forward
// pass used -- saves reading y (one third of the reduce pass, one quarter of the apply pass)
struct StatOp {  // sum(x), sum(x^2)
  static constexpr bool kChunk = false;
  __device__ __forceinline__ void init(int, const float*, const float*, const float*, const float*) {}
  template <typename R, typename T>
  __device__ __forceinline__ void load(long long off, R& xr, R&, R&, const T* x, const T*, const T*, int) const {
    xr.load(x + off);
  }
  template <typename R>
  __device__ __forceinline__ void acc(float (&a)[8], float (&b)[8], const R& xr, const R&, const R&, int) const {
    float v[8];
    xr.get(v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] += v[e];
      b[e] += v[e] * v[e];
    }
  }
};

struct BnBwdOp {  // sum(g), sum(g * xhat) with g = dy masked by the ReLU
  static constexpr bool kChunk = true;
  float mean[8], rstd[8], sc[8], sh[8];
  __device__ __forceinline__ void init(int c0, const float* m, const float* r, const float* gamma, const float* beta) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[e] = m[c0 + e];
      rstd[e] = r[c0 + e];
      sc[e] = (gamma ? gamma[c0 + e] : 1.f) * rstd[e];
      sh[e] = (beta ? beta[c0 + e] : 0.f) - mean[e] * sc[e];
    }
  }
  template <typename R, typename T>
  __device__ __forceinline__ void load(long long off, R& xr, R& gr, R& yr, const T* x, const T* dy, const T* y,
                                       int relu) const {
    xr.load(x + off);
    gr.load(dy + off);
    if (relu == 1) yr.load(y + off);
  }
  template <typename R>
  __device__ __forceinline__ void acc(float (&a)[8], float (&b)[8], const R& xr, const R& gr, const R& yr,
                                      int relu) const {
    float xv[8], gv[8];
    xr.get(xv);
    gr.get(gv);
    if (relu == 1) {
      float yv[8];
      yr.get(yv);
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = yv[e] > 0.f ? gv[e] : 0.f;
    } else if (relu == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = (xv[e] * sc[e] + sh[e]) > 0.f ? gv[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] += gv[e];
      b[e] += gv[e] * (xv[e] - mean[e]) * rstd[e];
    }
  }
};

// Streaming loops below keep FFA_EW_UNROLL independent 16-byte loads per operand in flight per thread (issued before
// the first use) and consume them in index order: the sums are the same chains of additions as a plain loop.  With one
// vector per iteration the wave waited out a full HBM round trip per 32 bytes (3.9 TB/s on the reduce at 4 waves/SIMD).
// (FFA_EW_UNROLL is defined at the top of the file: the launch grids depend on it.)
// FFA_EW_CHUNK 1: the U vectors of an iteration are consecutive 4 KB rows of one block (a block streams U * 4 KB
// contiguous bytes per operand per iteration); 0: they are a grid stride apart
#ifndef FFA_EW_CHUNK
#define FFA_EW_CHUNK 1
#endif
template <typename T>
struct EwUnroll {
  static constexpr int U = sizeof(T) == 2 ? FFA_EW_UNROLL : (FFA_EW_UNROLL > 2 ? 2 : FFA_EW_UNROLL);
};

// 8 channels as they sit in memory: loads land here and are widened to f32 only where they are used, so that U
// vectors per operand in flight cost U * 4 registers (bf16), not U * 8
template <typename T>
struct Raw8;
template <>
struct Raw8<ffa_bf16> {
  uint4 u;
  __device__ __forceinline__ void load(const ffa_bf16* p) { u = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float (&v)[8]) const {
    v[0] = __uint_as_float(u.x << 16);
    v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16);
    v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16);
    v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16);
    v[7] = __uint_as_float(u.w & 0xffff0000u);
  }
};
template <>
struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = reinterpret_cast<const float4*>(p)[0];
    b = reinterpret_cast<const float4*>(p)[1];
  }
  __device__ __forceinline__ void get(float (&v)[8]) const {
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
};

template <typename T, typename Op>
__global__ void __launch_bounds__(FFA_EW_THREADS)
channel_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                      const float* __restrict__ mean, const float* __restrict__ rstd,
                      const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ ws,
                      long long npix, int C, int relu) {
  __shared__ float red[FFA_EW_THREADS][17];
  const int CG = C / 8;
  const int PL = FFA_EW_THREADS / CG;
  const int t = threadIdx.x;
  const int cg = t % CG, pl = t / CG;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = b[e] = 0.f;
  if (pl < PL) {
    Op op;
    op.init(cg * 8, mean, rstd, gamma, beta);
    constexpr int U = EwUnroll<T>::U;
    const long long S = (long long)gridDim.x * PL;
    // Forward statistics (StatOp) keep the grid-stride pixel order: with the in-order accumulation below every thread's
    // sum is then the same chain of additions as a plain one-pixel-per-iteration loop, bit for bit.  (E[x^2] - mean^2 in
    // f32 partials carries ~1e-7 * mean^2 / var of rounding; another grouping moves scale / shift by up to 1e-5 on
    // low-variance channels, enough to flip a ReLU mask or two against the reference's evaluation on the 10 x 10 maps
    // of the U-TAE golden: tools/utae_grad_margin.py.)  The backward sums feed no mask and take the contiguous rows.
    // (forward statistics of tensors large enough to saturate the partial rows take the contiguous rows too: the
    // 64-channel 256^2 stem output, 268 MB, read at 2.5 TB/s a grid stride apart)
    const bool CHUNK = FFA_EW_CHUNK && (Op::kChunk || gridDim.x == FFA_MAX_PARTIALS);
    const long long us = CHUNK ? PL : S;  // distance between the U pixels of one iteration
    for (long long p = (long long)blockIdx.x * PL * (CHUNK ? U : 1) + pl; p < npix; p += S * U) {
      Raw8<T> xv[U], gv[U], yv[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long q = p + us * u;
        ok[u] = q < npix;
        op.load((ok[u] ? q : p) * C + cg * 8, xv[u], gv[u], yv[u], x, dy, y, relu);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ok[u]) op.acc(a, b, xv[u], gv[u], yv[u], relu);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[t][e] = a[e];
    red[t][8 + e] = b[e];
  }
  __syncthreads();
  for (int idx = t; idx < 2 * C; idx += FFA_EW_THREADS) {
    const int which = idx / C, c = idx % C;
    const int g = c / 8, e = (c % 8) + which * 8;
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * CG + g][e];
    ws[((long long)blockIdx.x * 2 + which) * C + c] = s;
  }
}

static int reduce_blocks(long long npix, int C) {
  const int PL = FFA_EW_THREADS / (C / 8);
  long long g = (npix + (long long)PL * 8 - 1) / ((long long)PL * 8);  // >= 8 pixels per thread
  if (g > FFA_MAX_PARTIALS) g = FFA_MAX_PARTIALS;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" long long ffa_bn_workspace_bytes(int C) {
  // block partials [FFA_MAX_PARTIALS][2][C] + three per-channel coefficient vectors for the backward apply
  return ((long long)FFA_MAX_PARTIALS * 2 * C + 6LL * C) * (long long)sizeof(float);
}

// Fixed-order sum of the block partials ws[p][which][C]: 256 threads = 32 partial lanes x 8 channels; lane l
// adds partials l, l+32, ... in double, lane 0 then adds the 32 lane sums in order.  (A single thread walking
// 1024 partials is a 1024-deep chain of dependent L2 round trips: 190 us per call, measured.)
#define FFA_FIN_THREADS 256
__device__ __forceinline__ bool reduce_partials(const float* __restrict__ ws, int nparts, int C, int& c, double& s,
                                                double& q) {
  __shared__ double sh[2][32][8];
  const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
  c = blockIdx.x * 8 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    // eight partials' loads in flight per lane, then added in index order: the sum is the same chain of
    // additions as a plain loop, but costs one memory round trip per eight partials instead of per partial
    for (int p0 = lane; p0 < nparts; p0 += 32 * 8) {
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 32 * u;
        const bool ok = p < nparts;
        const long long o = (long long)(ok ? p : p0) * 2 * C + c;
        const float ta = ws[o], tb = ws[o + C];
        va[u] = ok ? ta : 0.f;
        vb[u] = ok ? tb : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += (double)va[u];
        b += (double)vb[u];
      }
    }
  }
  sh[0][lane][cl] = a;
  sh[1][lane][cl] = b;
  __syncthreads();
  if (lane != 0 || c >= C) return false;
  s = 0.0;
  q = 0.0;
  for (int l = 0; l < 32; ++l) {
    s += sh[0][l][cl];
    q += sh[1][l][cl];
  }
  return true;
}

__global__ void __launch_bounds__(FFA_FIN_THREADS)
bn_finalize_kernel(const float* __restrict__ ws, int nparts, double count, int C, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                   float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  int c;
  double s, q;
  if (!reduce_partials(ws, nparts, C, c, s, q)) return;
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f;
  const float bt = beta ? beta[c] : 0.f;
  const float sc = g * rstd;
  scale[c] = sc;
  shift[c] = bt - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// Training-mode statistics of x[N][C] -> scale/shift (= gamma*rstd, beta - mean*gamma*rstd), saved
// mean/rstd for the backward pass, running-stat update (momentum convention of torch.nn.BatchNorm2d).
extern "C" int ffa_bn_stats(int dtype, const void* x, long long npix, int C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* scale,
                            float* shift, float* mean_out, float* rstd_out, void* workspace, long long workspace_bytes,
                            hipStream_t stream) {
  FFA_REQUIRE(x && scale && shift && mean_out && rstd_out && workspace, "bn_stats: null pointer");
  FFA_REQUIRE(C % 8 == 0 && C >= 8 && C <= 8 * FFA_EW_THREADS, "bn_stats: unsupported channel count %d", C);
  FFA_REQUIRE(npix > 0, "bn_stats: empty input");
  if (workspace_bytes < ffa_bn_workspace_bytes(C)) {
    ffa_set_error("bn_stats: workspace too small");
    return FFA_ERR_WORKSPACE;
  }
  const int nb = reduce_blocks(npix, C);
  float* ws = static_cast<float*>(workspace);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL((channel_reduce_kernel<ffa_bf16, StatOp>), dim3(nb), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)x, (const ffa_bf16*)nullptr, (const ffa_bf16*)nullptr, nullptr, nullptr, nullptr, nullptr, ws,
                       npix, C, 0);
  else
    hipLaunchKernelGGL((channel_reduce_kernel<float, StatOp>), dim3(nb), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)x, (const float*)nullptr, (const float*)nullptr, nullptr, nullptr, nullptr, nullptr, ws, npix,
                       C, 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ffa_cdiv(C, 8)), dim3(FFA_FIN_THREADS), 0, stream, ws, nb, (double)npix,
                     C, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean_out, rstd_out);
  return ffa_check_launch("bn_stats");
}

// Folds many partial rows ws[p][2][C] (p < nparts) into R rows dst[r][2][C]: row r is the fixed-order sum of the
// partials p = r, r + R, ...  (the conv epilogue leaves one partial per pixel tile: up to 32768 of them)
__global__ void __launch_bounds__(FFA_FIN_THREADS)
partials_fold_kernel(const float* __restrict__ ws, long long nparts, int C, int R, float* __restrict__ dst) {
  const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;  // 32 lanes x 8 channels
  const int c = blockIdx.x * 8 + cl;
  const int r = blockIdx.y;
  __shared__ float sh[2][32][8];
  float a = 0.f, b = 0.f;
  if (c < C) {
    for (long long p0 = r + (long long)lane * R; p0 < nparts; p0 += 32LL * 4 * R) {
      float va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long p = p0 + 32LL * u * R;
        const bool ok = p < nparts;
        const long long o = (ok ? p : p0) * 2 * C + c;
        const float ta = ws[o], tb = ws[o + C];
        va[u] = ok ? ta : 0.f;
        vb[u] = ok ? tb : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a += va[u];
        b += vb[u];
      }
    }
  }
  sh[0][lane][cl] = a;
  sh[1][lane][cl] = b;
  __syncthreads();
  if (lane == 0 && c < C) {
    float s = 0.f, q = 0.f;
    for (int l = 0; l < 32; ++l) {
      s += sh[0][l][cl];
      q += sh[1][l][cl];
    }
    dst[((long long)r * 2 + 0) * C + c] = s;
    dst[((long long)r * 2 + 1) * C + c] = q;
  }
}

// Training-mode BatchNorm parameters from partial sums somebody else produced (ffa_conv2d_stats): same outputs and
// running-statistics update as ffa_bn_stats, without the pass over the tensor.
extern "C" int ffa_bn_finalize(const float* partials, long long nparts, long long npix, int C, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* mean_out, float* rstd_out, void* workspace,
                               long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(partials && scale && shift && mean_out && rstd_out && workspace, "bn_finalize: null pointer");
  FFA_REQUIRE(C % 8 == 0 && C >= 8 && nparts > 0 && npix > 0, "bn_finalize: bad arguments");
  if (workspace_bytes < ffa_bn_workspace_bytes(C)) {
    ffa_set_error("bn_finalize: workspace too small");
    return FFA_ERR_WORKSPACE;
  }
  const float* src = partials;
  int n = (int)nparts;
  if (nparts > FFA_MAX_PARTIALS) {
    const int R = 64;
    float* ws = static_cast<float*>(workspace);
    hipLaunchKernelGGL(partials_fold_kernel, dim3(ffa_cdiv(C, 8), R), dim3(FFA_FIN_THREADS), 0, stream, partials, nparts,
                       C, R, ws);
    src = ws;
    n = R;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ffa_cdiv(C, 8)), dim3(FFA_FIN_THREADS), 0, stream, src, n, (double)npix,
                     C, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean_out, rstd_out);
  return ffa_check_launch("bn_finalize");
}

__global__ void bn_eval_params_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float eps, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(running_var[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * rstd;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - running_mean[c] * sc;
}

// Eval-mode affine of a BatchNorm2d: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale.
// The scale is folded into the packed conv weights (ffa_pack_conv_weight), the shift becomes the conv bias.
extern "C" int ffa_bn_eval_params(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift,
                                  hipStream_t stream) {
  FFA_REQUIRE(running_mean && running_var && scale && shift, "bn_eval_params: null pointer");
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3(ffa_cdiv(C, 128)), dim3(128), 0, stream, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  return ffa_check_launch("bn_eval_params");
}

template <typename T>
__global__ void __launch_bounds__(FFA_EW_THREADS)
bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                const float* __restrict__ scale, const float* __restrict__ shift, long long nvec, int C, int relu) {
  constexpr int U = EwUnroll<T>::U;
  const int CG = C / 8;
  const long long S = (long long)gridDim.x * blockDim.x;
  // U vectors per iteration: consecutive rows of the block when that keeps the thread's channel group, else a grid
  // stride apart (which always does: ew_grid_c)
  const bool chunk = FFA_EW_CHUNK && (FFA_EW_THREADS % CG == 0);
  const long long us = chunk ? FFA_EW_THREADS : S;
  long long i = blockIdx.x * (long long)blockDim.x * (chunk ? U : 1) + threadIdx.x;
  // the launch makes the grid stride a multiple of C / 8 (ew_grid_c): a thread keeps its channel group for the whole
  // loop and loads the per-channel vectors once
  float sc[8], sh[8];
  ffa_load8<float>(scale + ((int)i % CG) * 8, sc);
  ffa_load8<float>(shift + ((int)i % CG) * 8, sh);
  for (; i < nvec; i += S * U) {
    Raw8<T> xr[U], rr[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = i + us * u;
      ok[u] = q < nvec;
      const long long o = (ok[u] ? q : i) * 8;
      xr[u].load(x + o);
      if (res) rr[u].load(res + o);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = i + us * u;
      float v[8];
      xr[u].get(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
      if (res) {
        float r[8];
        rr[u].get(r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (ok[u]) ffa_store8<T>(y + q * 8, v);
    }
  }
}

// y = relu?( x * scale[c] + shift[c] (+ residual) )
extern "C" int ffa_bn_apply(int dtype, const void* x, const void* residual, void* y, const float* scale,
                            const float* shift, long long npix, int C, int relu, hipStream_t stream) {
  FFA_REQUIRE(x && y && scale && shift && C % 8 == 0, "bn_apply: bad arguments");
  const long long nvec = npix * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(bn_apply_kernel<ffa_bf16>, dim3(ew_grid_c(nvec, C)), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)x, (const ffa_bf16*)residual, (ffa_bf16*)y, scale, shift, nvec, C, relu);
  else
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(ew_grid_c(nvec, C)), dim3(FFA_EW_THREADS), 0, stream, (const float*)x,
                       (const float*)residual, (float*)y, scale, shift, nvec, C, relu);
  return ffa_check_launch("bn_apply");
}

// dgamma / dbeta from the block partials, plus the three per-channel coefficients of the apply pass:
//   dx = gamma*rstd * (g - dbeta/N - xhat*dgamma/N) = kg*g + kx*x + k0
//   kg = gamma*rstd, kx = -kg*rstd*dgamma/N, k0 = -kg*dbeta/N - kx*mean
__global__ void __launch_bounds__(FFA_FIN_THREADS)
bn_bwd_finalize_kernel(const float* __restrict__ ws, int nparts, int C, const float* __restrict__ gamma,
                       const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd, float inv_count,
                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef, int raw_x = 0) {
  int c;
  double s, q;
  if (!reduce_partials(ws, nparts, C, c, s, q)) return;
  if (raw_x) q = (q - (double)mean[c] * s) * (double)rstd[c];  // partials hold sum(g*x): sum(g*xhat) = rstd*(.. - mean*sum g)
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
  if (coef) {
    const float kg = (gamma ? gamma[c] : 1.f) * rstd[c];
    const float kx = -kg * rstd[c] * (float)q * inv_count;
    coef[c] = kg;
    coef[C + c] = kx;
    coef[2 * C + c] = -kg * (float)s * inv_count - kx * mean[c];
    coef[3 * C + c] = kg;  // forward scale = gamma * rstd
    coef[4 * C + c] = (beta ? beta[c] : 0.f) - mean[c] * kg;  // forward shift, same expression as bn_finalize
  }
}

template <typename T>
__global__ void __launch_bounds__(FFA_EW_THREADS)
bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                    const float* __restrict__ coef, T* __restrict__ dx, T* __restrict__ dres, long long nvec, int C,
                    int relu) {
  constexpr int U = EwUnroll<T>::U;
  const int CG = C / 8;
  const long long S = (long long)gridDim.x * blockDim.x;
  // U vectors per iteration: consecutive rows of the block when that keeps the thread's channel group, else a grid
  // stride apart (which always does: ew_grid_c)
  const bool chunk = FFA_EW_CHUNK && (FFA_EW_THREADS % CG == 0);
  const long long us = chunk ? FFA_EW_THREADS : S;
  long long i = blockIdx.x * (long long)blockDim.x * (chunk ? U : 1) + threadIdx.x;
  float kg[8], kx[8], k0[8], sc[8], sh[8];
  auto load_coef = [&](int c0) {
    ffa_load8<float>(coef + c0, kg);
    ffa_load8<float>(coef + C + c0, kx);
    ffa_load8<float>(coef + 2 * C + c0, k0);
    if (relu == 2) {
      ffa_load8<float>(coef + 3 * C + c0, sc);
      ffa_load8<float>(coef + 4 * C + c0, sh);
    }
  };
  load_coef(((int)i % CG) * 8);  // grid stride % (C / 8) == 0, see bn_apply_kernel
  for (; i < nvec; i += S * U) {
    Raw8<T> xr[U], gr[U], yr[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = i + us * u;
      ok[u] = q < nvec;
      const long long o = (ok[u] ? q : i) * 8;
      xr[u].load(x + o);
      gr[u].load(dy + o);
      if (relu == 1) yr[u].load(y + o);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = i + us * u;
      float xv[8], gv[8];
      xr[u].get(xv);
      gr[u].get(gv);
      if (relu == 1) {
        float yv[8];
        yr[u].get(yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = yv[e] > 0.f ? gv[e] : 0.f;
      } else if (relu == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = (xv[e] * sc[e] + sh[e]) > 0.f ? gv[e] : 0.f;
      }
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = kg[e] * gv[e] + kx[e] * xv[e] + k0[e];
      if (ok[u]) {
        if (dres) ffa_store8<T>(dres + q * 8, gv);
        ffa_store8<T>(dx + q * 8, o);
      }
    }
  }
}

// Backward of y = relu?(bn_train(x) (+ residual)).  Inputs: x (pre-norm conv output), dy, y (only read
// when relu), saved mean/rstd.  Outputs: dx, dgamma, dbeta and -- when dres is non-null -- the gradient
// of the residual branch (dy masked by the ReLU).
// stages: bit 0 = reduction + finalize (dgamma, dbeta, apply coefficients into the workspace), bit 1 = apply (dx,
// dres from the coefficients a stage-1 call left in the SAME workspace), bit 2 = the reduction alone, bit 3 = the
// finalize alone (1 == 4 | 8).  ffa_bn_bwd = all; the split exists so that a profiler-free harness can bracket each
// kernel with its own events (bench.py's HBM roofline entry).
extern "C" int ffa_bn_bwd_stages(int dtype, const void* x, const void* dy, const void* y, const float* gamma,
                                 const float* beta, const float* mean, const float* rstd, void* dx, void* dres,
                                 float* dgamma, float* dbeta, long long npix, int C, int relu, void* workspace,
                                 long long workspace_bytes, int stages, hipStream_t stream);

extern "C" int ffa_bn_bwd(int dtype, const void* x, const void* dy, const void* y, const float* gamma,
                          const float* beta, const float* mean, const float* rstd, void* dx, void* dres,
                          float* dgamma, float* dbeta,
                          long long npix, int C, int relu, void* workspace, long long workspace_bytes,
                          hipStream_t stream) {
  return ffa_bn_bwd_stages(dtype, x, dy, y, gamma, beta, mean, rstd, dx, dres, dgamma, dbeta, npix, C, relu, workspace,
                           workspace_bytes, 3, stream);
}

extern "C" int ffa_bn_bwd_stages(int dtype, const void* x, const void* dy, const void* y, const float* gamma,
                                 const float* beta, const float* mean, const float* rstd, void* dx, void* dres,
                                 float* dgamma, float* dbeta, long long npix, int C, int relu, void* workspace,
                                 long long workspace_bytes, int stages, hipStream_t stream) {
  FFA_REQUIRE(stages >= 1 && stages <= 15, "bn_bwd: stages is a mask of 1 (reduce + finalize), 2 (apply), 4 (reduce), 8 (finalize)");
  FFA_REQUIRE(x && dy && dx && mean && rstd && dgamma && dbeta && workspace, "bn_bwd: null pointer");
  FFA_REQUIRE(relu >= 0 && relu <= 2, "bn_bwd: relu mode must be 0, 1 (mask from y) or 2 (mask from x)");
  FFA_REQUIRE(relu != 1 || y, "bn_bwd: relu mode 1 needs the forward output");
  FFA_REQUIRE(C % 8 == 0 && C >= 8 && C <= 8 * FFA_EW_THREADS, "bn_bwd: unsupported channel count %d", C);
  if (workspace_bytes < ffa_bn_workspace_bytes(C)) {
    ffa_set_error("bn_bwd: workspace too small");
    return FFA_ERR_WORKSPACE;
  }
  const int nb = reduce_blocks(npix, C);
  float* ws = static_cast<float*>(workspace);
  const long long nvec = npix * (C / 8);
  const float inv_count = (float)(1.0 / (double)npix);
  float* coef = ws + (long long)FFA_MAX_PARTIALS * 2 * C;
  if (dtype == FFA_BF16) {
    if (stages & 5)
      hipLaunchKernelGGL((channel_reduce_kernel<ffa_bf16, BnBwdOp>), dim3(nb), dim3(FFA_EW_THREADS), 0, stream,
                         (const ffa_bf16*)x, (const ffa_bf16*)dy, (const ffa_bf16*)y, mean, rstd, gamma, beta, ws, npix,
                         C, relu);
    if (stages & 9)
      hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ffa_cdiv(C, 8)), dim3(FFA_FIN_THREADS), 0, stream, ws, nb, C,
                         gamma, beta, mean, rstd, inv_count, dgamma, dbeta, coef);
    if (stages & 2)
      hipLaunchKernelGGL(bn_bwd_apply_kernel<ffa_bf16>, dim3(ew_grid_c(nvec, C)), dim3(FFA_EW_THREADS), 0, stream,
                         (const ffa_bf16*)x, (const ffa_bf16*)dy, (const ffa_bf16*)y, coef, (ffa_bf16*)dx,
                         (ffa_bf16*)dres, nvec, C, relu);
  } else {
    if (stages & 5)
      hipLaunchKernelGGL((channel_reduce_kernel<float, BnBwdOp>), dim3(nb), dim3(FFA_EW_THREADS), 0, stream,
                         (const float*)x, (const float*)dy, (const float*)y, mean, rstd, gamma, beta, ws, npix, C, relu);
    if (stages & 9)
      hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ffa_cdiv(C, 8)), dim3(FFA_FIN_THREADS), 0, stream, ws, nb, C,
                         gamma, beta, mean, rstd, inv_count, dgamma, dbeta, coef);
    if (stages & 2)
      hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_grid_c(nvec, C)), dim3(FFA_EW_THREADS), 0, stream,
                         (const float*)x, (const float*)dy, (const float*)y, coef, (float*)dx, (float*)dres, nvec, C,
                         relu);
  }
  return ffa_check_launch("bn_bwd");
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward as ONE co-resident kernel for tensors whose dy and x fit the register files of the chip
// (bf16, <= ~48 MB per tensor): every thread keeps its share of x and of the masked dy in registers across two
// grid-wide barriers -- pass 1 reads dy, x (and y for the mask-from-output mode) once and leaves per-block partial
// sums, the blocks then add the partials (one output value per block, fixed order), pass 2 turns the registers into dx.
// Three passes over memory instead of five and one launch instead of three (reduce, finalize, apply).
// MEASURED SLOWER and therefore not used by default (flairhip/ops.py FUSED_BN_BWD_COOP): a grid barrier across the
// eight XCDs costs ~20-25 us (device-scope atomics + L2 write-back / invalidate), so a call takes 72-76 us whatever
// the tensor size, against 24-55 us for the three kernels on the tensors that fit.
//
// The barriers are bounded spins on device-scope counters (zero on entry; the last block to leave zeroes them
// again): the grid is one 512-thread block per CU, which is co-resident whenever the launch gets the whole chip; if
// some CUs are busy with another stream's kernels the late blocks simply arrive late.  A barrier that is not
// satisfied within ~2 s gives up (sets sync[3], which the host wrapper of the NEXT call reports) instead of hanging
// the device.

#define FFA_COOP_THREADS 512

__device__ __forceinline__ bool ffa_grid_barrier(unsigned* counter, unsigned nblocks, unsigned* fail) {
  __shared__ int ok_s;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nblocks) {
      __builtin_amdgcn_s_sleep(16);
      if (++spins > (1u << 22)) {
        ok = 0;
        atomicExch(fail, 1u);
        break;
      }
    }
    __threadfence();
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

__device__ __forceinline__ void ffa_unpack8_bf16(const ffa_u32x4& u, float (&v)[8]) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}

template <int ITEMS>
__global__ void __launch_bounds__(FFA_COOP_THREADS, 1)
bn_bwd_coop_kernel(const ffa_bf16* __restrict__ x, const ffa_bf16* __restrict__ dy, const ffa_bf16* __restrict__ y,
                   const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                   const float* __restrict__ rstd, ffa_bf16* __restrict__ dx, ffa_bf16* __restrict__ dres,
                   float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ ws,
                   unsigned* __restrict__ sync, long long nvec, int C, int relu, float inv_count) {
  __shared__ float red[FFA_COOP_THREADS][17];
  __shared__ double dred[FFA_COOP_THREADS / 64];
  const int CG = C / 8;
  const int t = threadIdx.x;
  const long long nthreads = (long long)gridDim.x * FFA_COOP_THREADS;  // a multiple of CG (host-checked)
  const long long tid_g = (long long)blockIdx.x * FFA_COOP_THREADS + t;
  const int cg = (int)(tid_g % CG), c0 = cg * 8;
  float mu[8], rs[8], sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[c0 + e];
    rs[e] = rstd[c0 + e];
    sc[e] = (gamma ? gamma[c0 + e] : 1.f) * rs[e];       // same expressions as BnBwdOp::init
    sh[e] = (beta ? beta[c0 + e] : 0.f) - mu[e] * sc[e];
  }
  // ---- pass 1: registers <- x, masked dy; per-thread sums in item order ----
  ffa_u32x4 xr[ITEMS], gr[ITEMS];
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = b[e] = 0.f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const long long i = tid_g + (long long)k * nthreads;
    xr[k] = ffa_u32x4{0u, 0u, 0u, 0u};
    gr[k] = ffa_u32x4{0u, 0u, 0u, 0u};
    if (i < nvec) {
      xr[k] = *reinterpret_cast<const ffa_u32x4*>(x + i * 8);
      ffa_u32x4 g = *reinterpret_cast<const ffa_u32x4*>(dy + i * 8);
      float xv[8], gv[8];
      ffa_unpack8_bf16(xr[k], xv);
      unsigned keep = 0xffu;  // bit e: element e passes the ReLU
      if (relu == 1) {
        float yv[8];
        ffa_unpack8_bf16(*reinterpret_cast<const ffa_u32x4*>(y + i * 8), yv);
        keep = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) keep |= (yv[e] > 0.f ? 1u : 0u) << e;
      } else if (relu == 2) {
        keep = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) keep |= ((xv[e] * sc[e] + sh[e]) > 0.f ? 1u : 0u) << e;
      }
      uint32_t w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        w[q] &= ((keep >> (2 * q)) & 1u ? 0x0000ffffu : 0u) | ((keep >> (2 * q + 1)) & 1u ? 0xffff0000u : 0u);
      gr[k] = ffa_u32x4{w[0], w[1], w[2], w[3]};
      ffa_unpack8_bf16(gr[k], gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] += gv[e];
        b[e] += gv[e] * (xv[e] - mu[e]) * rs[e];
      }
    }
  }
  // block partials: threads of one channel group are t, t + CG, ... (512 % CG == 0)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[t][e] = a[e];
    red[t][8 + e] = b[e];
  }
  __syncthreads();
  const int PL = FFA_COOP_THREADS / CG;
  for (int idx = t; idx < 2 * C; idx += FFA_COOP_THREADS) {
    const int which = idx / C, c = idx % C;
    const int g = c / 8, e = (c % 8) + which * 8;
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * CG + g][e];
    ws[((long long)blockIdx.x * 2 + which) * C + c] = s;
  }
  const unsigned nb = gridDim.x;
  bool ok = ffa_grid_barrier(sync + 0, nb, sync + 3);
  // ---- totals: value v = which * C + c is summed by block v % nb over the nb block partials, in block order ----
  float* tot = ws + (long long)FFA_MAX_PARTIALS * 2 * C;  // [2][C]
  for (int v = blockIdx.x; v < 2 * C; v += nb) {
    double s = 0.0;
    for (unsigned p = t; p < nb; p += FFA_COOP_THREADS) s += (double)ws[(long long)p * 2 * C + v];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((t & 63) == 0) dred[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
      double r = 0.0;
      for (int wv = 0; wv < FFA_COOP_THREADS / 64; ++wv) r += dred[wv];
      tot[v] = (float)r;
    }
    __syncthreads();
  }
  ok = ffa_grid_barrier(sync + 1, nb, sync + 3) && ok;
  // ---- pass 2: dx from the registers ----
  float kg[8], kx[8], k0[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float s = __hip_atomic_load(tot + c0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float q = __hip_atomic_load(tot + C + c0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    kg[e] = (gamma ? gamma[c0 + e] : 1.f) * rs[e];  // same expressions as bn_bwd_finalize_kernel
    kx[e] = -kg[e] * rs[e] * q * inv_count;
    k0[e] = -kg[e] * s * inv_count - kx[e] * mu[e];
    if (blockIdx.x == 0 && t < CG) {  // thread t of block 0 owns channel group t
      dbeta[c0 + e] = s;
      dgamma[c0 + e] = q;
    }
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const long long i = tid_g + (long long)k * nthreads;
    if (i < nvec) {
      float xv[8], gv[8], o[8];
      ffa_unpack8_bf16(xr[k], xv);
      ffa_unpack8_bf16(gr[k], gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = kg[e] * gv[e] + kx[e] * xv[e] + k0[e];
      ffa_store8<ffa_bf16>(dx + i * 8, o);
      if (dres) *reinterpret_cast<ffa_u32x4*>(dres + i * 8) = gr[k];
    }
  }
  // ---- leave: the last block out re-arms the counters for the next launch ----
  __syncthreads();
  if (t == 0) {
    __threadfence();
    if (atomicAdd(sync + 2, 1u) == nb - 1) {
      sync[0] = 0u;
      sync[1] = 0u;
      sync[2] = 0u;
      __threadfence();
    }
  }
  (void)ok;
}

// One-kernel form of ffa_bn_bwd (bf16 only).  sync: four uint32 in device memory, zero before the first call and
// never touched by the caller afterwards (the kernel re-arms them; sync[3] latches a barrier time-out).  Returns
// FFA_ERR_UNSUPPORTED when the tensor does not fit the register budget of the chip (the caller then uses ffa_bn_bwd).
extern "C" int ffa_bn_bwd_fused(int dtype, const void* x, const void* dy, const void* y, const float* gamma,
                                const float* beta, const float* mean, const float* rstd, void* dx, void* dres,
                                float* dgamma, float* dbeta, long long npix, int C, int relu, void* workspace,
                                long long workspace_bytes, unsigned* sync, hipStream_t stream) {
  FFA_REQUIRE(x && dy && dx && mean && rstd && dgamma && dbeta && workspace && sync, "bn_bwd_fused: null pointer");
  FFA_REQUIRE(relu >= 0 && relu <= 2 && (relu != 1 || y), "bn_bwd_fused: bad relu mode");
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return FFA_ERR_UNSUPPORTED;
    cus = prop.multiProcessorCount;
  }
  const int CG = C / 8;
  if (dtype != FFA_BF16 || C % 8 != 0 || CG < 1 || FFA_COOP_THREADS % CG != 0 || cus < 8 || cus > FFA_MAX_PARTIALS) {
    ffa_set_error("bn_bwd_fused: unsupported dtype / channel count %d", C);
    return FFA_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < ffa_bn_workspace_bytes(C)) {
    ffa_set_error("bn_bwd_fused: workspace too small");
    return FFA_ERR_WORKSPACE;
  }
  const long long nvec = npix * CG;
  const long long nthreads = (long long)cus * FFA_COOP_THREADS;
  const long long items = (nvec + nthreads - 1) / nthreads;
  if (items > 16) {
    ffa_set_error("bn_bwd_fused: %lld vectors per thread exceed the register budget", items);
    return FFA_ERR_UNSUPPORTED;
  }
  const float inv_count = (float)(1.0 / (double)npix);
#define FFA_COOP_LAUNCH(N_)                                                                                      \
  hipLaunchKernelGGL(bn_bwd_coop_kernel<N_>, dim3(cus), dim3(FFA_COOP_THREADS), 0, stream, (const ffa_bf16*)x,   \
                     (const ffa_bf16*)dy, (const ffa_bf16*)y, gamma, beta, mean, rstd, (ffa_bf16*)dx,            \
                     (ffa_bf16*)dres, dgamma, dbeta, (float*)workspace, sync, nvec, C, relu, inv_count)
  // (8 and 24 vectors per thread were tried: hipcc spills both; 16 x 16 B x 2 tensors = 128 data registers fits)
  if (items <= 4) FFA_COOP_LAUNCH(4);
  else FFA_COOP_LAUNCH(16);
#undef FFA_COOP_LAUNCH
  return ffa_check_launch("bn_bwd_fused");
}

// ffa_bn_bwd for y = relu(bn_train(x)) when the two reductions were already taken by the producer of dy
// (ffa_conv2d_bnbwd): partials[nparts][2][C] = per-tile (sum g, sum g*x).  Finalize + apply only.
extern "C" int ffa_bn_bwd_partials(int dtype, const void* x, const void* dy, const float* partials, long long nparts,
                                   const float* gamma, const float* beta, const float* mean, const float* rstd,
                                   void* dx, float* dgamma, float* dbeta, long long npix, int C, void* workspace,
                                   long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(x && dy && partials && dx && mean && rstd && dgamma && dbeta && workspace, "bn_bwd_partials: null pointer");
  FFA_REQUIRE(C % 8 == 0 && C >= 8 && C <= 8 * FFA_EW_THREADS && nparts > 0, "bn_bwd_partials: bad arguments");
  if (workspace_bytes < ffa_bn_workspace_bytes(C)) {
    ffa_set_error("bn_bwd_partials: workspace too small");
    return FFA_ERR_WORKSPACE;
  }
  float* ws = static_cast<float*>(workspace);
  float* coef = ws + (long long)FFA_MAX_PARTIALS * 2 * C;
  const float* src = partials;
  int n = (int)nparts;
  if (nparts > FFA_MAX_PARTIALS) {
    const int R = 64;
    hipLaunchKernelGGL(partials_fold_kernel, dim3(ffa_cdiv(C, 8), R), dim3(FFA_FIN_THREADS), 0, stream, partials, nparts,
                       C, R, ws);
    src = ws;
    n = R;
  }
  const long long nvec = npix * (C / 8);
  const float inv_count = (float)(1.0 / (double)npix);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ffa_cdiv(C, 8)), dim3(FFA_FIN_THREADS), 0, stream, src, n, C, gamma, beta,
                     mean, rstd, inv_count, dgamma, dbeta, coef, 1);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<ffa_bf16>, dim3(ew_grid_c(nvec, C)), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)x, (const ffa_bf16*)dy, (const ffa_bf16*)nullptr, coef, (ffa_bf16*)dx,
                       (ffa_bf16*)nullptr, nvec, C, 2);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_grid_c(nvec, C)), dim3(FFA_EW_THREADS), 0, stream, (const float*)x,
                       (const float*)dy, (const float*)nullptr, coef, (float*)dx, (float*)nullptr, nvec, C, 2);
  return ffa_check_launch("bn_bwd_partials");
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1), floor mode, -inf padding; ties -> first tap in row-major
// window order (the element ATen's CPU kernel keeps: strict '>' while scanning), recorded as a
// window-local index 0..8 so the backward pass routes gradients exactly like the reference.

// One block per output row (forward) / input row (backward), threads over (x, channel group): the index arithmetic is
// one 32-bit divide per element (the flat 64-bit i -> (b, y, x, g) split cost more issue slots than the memory traffic).
template <typename T>
__global__ void __launch_bounds__(FFA_EW_THREADS)
maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int B, int H, int W, int C,
                   int Ho, int Wo) {
  const unsigned CG = C / 8;
  const int oy = blockIdx.x % Ho;
  const long long b = blockIdx.x / Ho;
  const long long row = (long long)blockIdx.x * Wo * CG;
  for (unsigned j = threadIdx.x; j < (unsigned)Wo * CG; j += FFA_EW_THREADS) {
    const int ox = (int)(j / CG), g = (int)(j % CG);
    const long long i = row + j;
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    bool first = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = oy * 2 - 1 + r;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ix = ox * 2 - 1 + s;
        if (ix < 0 || ix >= W) continue;
        float v[8];
        ffa_load8<T>(x + ((b * H + iy) * W + ix) * C + g * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (first || v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = r * 3 + s;
          }
        }
        first = false;
      }
    }
    ffa_store8<T>(y + i * 8, best);
    uint32_t lo = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    uint32_t hi = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + i * 8) = make_uint2(lo, hi);
  }
}

template <typename T>
__global__ void __launch_bounds__(FFA_EW_THREADS)
maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, const T* __restrict__ add,
                   T* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
  // gather form: input pixel (iy, ix) is tap (r, s) of output (oy, ox) when iy = 2*oy - 1 + r
  const unsigned CG = C / 8;
  const int iy = blockIdx.x % H;
  const long long b = blockIdx.x / H;
  const long long row = (long long)blockIdx.x * W * CG;
  for (unsigned j = threadIdx.x; j < (unsigned)W * CG; j += FFA_EW_THREADS) {
    const int ix = (int)(j / CG), g = (int)(j % CG);
    const long long i = row + j;
    // The windows that contain input pixel (iy, ix): rows oy with iy = 2 oy - 1 + r -- one (r = 1) when iy is even, two
    // (r = 0 and r = 2) when it is odd -- and the same for the columns: 1, 2 or 4 candidates.  All their index and
    // gradient loads are issued before the first use (the branchy tap loop serialised up to nine dependent loads).
    const int ny = (iy & 1) ? 2 : 1, nx = (ix & 1) ? 2 : 1;
    const int oy_c[2] = {(iy & 1) ? (iy + 1) >> 1 : iy >> 1, (iy - 1) >> 1};
    const int r_c[2] = {(iy & 1) ? 0 : 1, 2};
    const int ox_c[2] = {(ix & 1) ? (ix + 1) >> 1 : ix >> 1, (ix - 1) >> 1};
    const int s_c[2] = {(ix & 1) ? 0 : 1, 2};
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    Raw8<T> addr, gr[4];
    uint2 kk[4];
    bool ok[4];
    if (add) addr.load(add + i * 8);  // a second gradient of the pooled tensor's input (U-Net skip), summed here
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cy = c >> 1, cx = c & 1;
      ok[c] = cy < ny && cx < nx && oy_c[cy] >= 0 && oy_c[cy] < Ho && ox_c[cx] >= 0 && ox_c[cx] < Wo;
      const long long o = ok[c] ? (((b * Ho + oy_c[cy]) * Wo + ox_c[cx]) * CG + g) * 8 : 0;
      kk[c] = *reinterpret_cast<const uint2*>(idx + o);
      gr[c].load(dy + o);
    }
    if (add) addr.get(acc);
    // candidates in index order = taps ascending: the same order of additions as a loop over the nine taps
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!ok[c]) continue;
      const int tap = r_c[c >> 1] * 3 + s_c[c & 1];
      float gv[8];
      gr[c].get(gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t word = e < 4 ? kk[c].x : kk[c].y;
        const int wq = (word >> (8 * (e & 3))) & 0xff;
        if (wq == tap) acc[e] += gv[e];
      }
    }
    ffa_store8<T>(dx + i * 8, acc);
  }
}

extern "C" int ffa_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* idx, int B, int H, int W, int C,
                                    hipStream_t stream) {
  FFA_REQUIRE(x && y && idx && C % 8 == 0, "maxpool_fwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  FFA_REQUIRE(B > 0 && (long long)B * Ho < (1LL << 31), "maxpool_fwd: too many rows");
  const int rows = B * Ho;
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(maxpool_fwd_kernel<ffa_bf16>, dim3(rows), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)x, (ffa_bf16*)y, idx, B, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(rows), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)x, (float*)y, idx, B, H, W, C, Ho, Wo);
  return ffa_check_launch("maxpool_fwd");
}

extern "C" int ffa_maxpool3x3s2_bwd(int dtype, const void* dy, const uint8_t* idx, const void* add, void* dx, int B,
                                    int H, int W, int C, hipStream_t stream) {
  FFA_REQUIRE(dy && dx && idx && C % 8 == 0, "maxpool_bwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  FFA_REQUIRE(B > 0 && (long long)B * H < (1LL << 31), "maxpool_bwd: too many rows");
  const int rows = B * H;
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(maxpool_bwd_kernel<ffa_bf16>, dim3(rows), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)dy, idx, (const ffa_bf16*)add, (ffa_bf16*)dx, B, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(rows), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)dy, idx, (const float*)add, (float*)dx, B, H, W, C, Ho, Wo);
  return ffa_check_launch("maxpool_bwd");
}

// Per-channel sum and sum of squares of x[N][C] (f32 outputs); used for the head-conv bias gradient.
extern "C" int ffa_channel_sums(int dtype, const void* x, long long npix, int C, float* sum_out, float* sumsq_out,
                                void* workspace, long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(x && sum_out && sumsq_out && workspace, "channel_sums: null pointer");
  FFA_REQUIRE(C % 8 == 0 && C >= 8 && C <= 8 * FFA_EW_THREADS, "channel_sums: unsupported channel count %d", C);
  if (workspace_bytes < ffa_bn_workspace_bytes(C)) {
    ffa_set_error("channel_sums: workspace too small");
    return FFA_ERR_WORKSPACE;
  }
  const int nb = reduce_blocks(npix, C);
  float* ws = static_cast<float*>(workspace);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL((channel_reduce_kernel<ffa_bf16, StatOp>), dim3(nb), dim3(FFA_EW_THREADS), 0, stream,
                       (const ffa_bf16*)x, (const ffa_bf16*)nullptr, (const ffa_bf16*)nullptr, nullptr, nullptr, nullptr, nullptr, ws,
                       npix, C, 0);
  else
    hipLaunchKernelGGL((channel_reduce_kernel<float, StatOp>), dim3(nb), dim3(FFA_EW_THREADS), 0, stream,
                       (const float*)x, (const float*)nullptr, (const float*)nullptr, nullptr, nullptr, nullptr, nullptr, ws, npix,
                       C, 0);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ffa_cdiv(C, 8)), dim3(FFA_FIN_THREADS), 0, stream, ws, nb, C,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f,
                     sumsq_out, sum_out, (float*)nullptr);
  return ffa_check_launch("channel_sums");
}

