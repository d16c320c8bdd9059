// Token GEMM for the transformer layers (nn.Linear of timm's swin_transformer.py: qkv / proj / Mlp.fc1 / Mlp.fc2 /
// PatchMerging.reduction, and PatchEmbed.proj after ffa_space_to_depth):
//
//   out[m][n] = act( sum_k a[m][k] * w[n][k] + bias[n] ) + residual[m][n]          bf16 in / out, f32 accumulate
//
// Both operands are K-contiguous (activations [M][lda], weights in nn.Linear's own [N][K] layout), which is the
// operand order of v_mfma_f32_16x16x32_bf16: a lane's fragment is 16 contiguous bytes of one row.  The kernel forms
// C^T = W A^T (A operand = 16 features x 32 k, B operand = 32 k x 16 tokens) so that a lane ends up with four
// consecutive FEATURES of one token -- one 8-byte piece of an output row; the block's 128 x 128 tile is then turned
// into whole 128-byte row segments through LDS for the bias / GELU / residual epilogue and the store.
//
// Block: 128 tokens x 128 features, 4 waves as 2 x 2 (64 x 64 each = 4 x 4 MFMA tiles, 64 accumulator VGPRs), BK = 64.
// LDS: two images of (128 + 128) rows x 144 bytes (128 data + 16 pad: the 16 rows x 4 k-pieces of a fragment read hit
// 64 distinct 4-bank groups) = 72 KB, two blocks per CU.  Pipeline: chunk c+2 is in flight from HBM into registers and
// chunk c+1 moves registers -> LDS (other image) while chunk c feeds the matrix pipe: ONE barrier per 32 MFMAs per wave.
// Bound: HBM for the thin early stages (K = 128: 1.5 flop per byte moved at stage 1), MFMA for K >= 512.
#include "ffa_common.h"
#include <stdlib.h>

#define FFA_ACT_NONE 0
#define FFA_ACT_GELU 1
#define FFA_ACT_DGELU 2  // out = acc * gelu'(aux): the input gradient of fc2 carried through Mlp's activation
#define FFA_ACT_RELU 3   // 1x1 convolution + folded BatchNorm + ReLU of the UPerNet decoder (evaluation)

struct GemmArgs {
  const ffa_bf16* a;
  const ffa_bf16* w;
  const float* bias;
  const ffa_bf16* residual;
  ffa_bf16* out;
  ffa_bf16* aux;           // GELU: the pre-activation is stored here as well (training); DGELU: read from here
  const float* row_scale;  // optional per-sample factor (DropPath): rows [i * rows_per_scale, (i+1) * rows_per_scale)
  long long lda, ldr, ldc, ldaux;
  int M, K, N, act, nblk_n, rows_per_scale;
};

namespace {
constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int GPITCH = 144;                       // bytes per staged row
constexpr int GIMG = (GBM + GBN) * GPITCH;        // one LDS image
constexpr int GEP = 144;                          // epilogue row pitch (64 features bf16 + pad)
}  // namespace

// nn.GELU() (erf form) for the bf16 epilogues.  gelu(x) = x Phi(x) with Phi(x) - 1/2 = x Q(x^2): Q is a degree-8
// near-minimax fit of erf(x / sqrt 2) / (2x) on |x| <= 4.4 (x is clamped there: Phi(4.4) = 1 - 5e-6), |error| <= 1.7e-5 in
// Phi and <= 7.2e-5 in gelu -- below the bf16 rounding step of any activation above 0.02 in magnitude.  Nine FMAs on
// PAIRS of elements (v_pk_fma_f32), no transcendental: the epilogue of the memory-bound early-stage GEMMs is VALU time
// nothing overlaps (measured on M = 524288, K = 96, N = 384: +105 us of GELU / +192 us of GELU' on 188 / 162 us with
// libm-grade erf by Abramowitz & Stegun 7.1.26 (exp + rcp); the f32 parity kernel keeps erff).
typedef float ffa_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ffa_f32x2 gemm_phi_minus_half2(ffa_f32x2 x) {
  ffa_f32x2 xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -4.4f, 4.4f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -4.4f, 4.4f);
  const ffa_f32x2 t = xc * xc;
  ffa_f32x2 q = {4.234514475e-11f, 4.234514475e-11f};
  q = q * t + ffa_f32x2{-4.329148151e-09f, -4.329148151e-09f};
  q = q * t + ffa_f32x2{1.947642545e-07f, 1.947642545e-07f};
  q = q * t + ffa_f32x2{-5.124695235e-06f, -5.124695235e-06f};
  q = q * t + ffa_f32x2{8.877788787e-05f, 8.877788787e-05f};
  q = q * t + ffa_f32x2{-1.084315358e-03f, -1.084315358e-03f};
  q = q * t + ffa_f32x2{9.749136865e-03f, 9.749136865e-03f};
  q = q * t + ffa_f32x2{-6.626226753e-02f, -6.626226753e-02f};
  q = q * t + ffa_f32x2{3.988730609e-01f, 3.988730609e-01f};
  return xc * q;
}
// v[i] = gelu(v[i]), N even
template <int N>
__device__ __forceinline__ void gemm_gelu_n(float* v) {
#pragma unroll
  for (int e = 0; e < N; e += 2) {
    const ffa_f32x2 x = {v[e], v[e + 1]};
    const ffa_f32x2 r = x * (gemm_phi_minus_half2(x) + ffa_f32x2{0.5f, 0.5f});
    v[e] = r.x;
    v[e + 1] = r.y;
  }
}
// o[i] *= gelu'(u[i]),  gelu'(x) = Phi(x) + x phi(x)
template <int N>
__device__ __forceinline__ void gemm_dgelu_n(float* o, const float* u) {
#pragma unroll
  for (int e = 0; e < N; e += 2) {
    const ffa_f32x2 x = {u[e], u[e + 1]};
    const ffa_f32x2 a = x * x * ffa_f32x2{-0.72134752044448170368f, -0.72134752044448170368f};  // -x^2 / 2 in log2 units
    ffa_f32x2 ex;
    ex.x = __builtin_amdgcn_exp2f(a.x);
    ex.y = __builtin_amdgcn_exp2f(a.y);
    const ffa_f32x2 d = gemm_phi_minus_half2(x) + ffa_f32x2{0.5f, 0.5f} +
                        x * ex * ffa_f32x2{0.39894228040143267794f, 0.39894228040143267794f};
    o[e] *= d.x;
    o[e + 1] *= d.y;
  }
}

// TT = 16-token tiles per wave: 4 -> the 128-token block tile described above; 2 -> 64 tokens per block, twice the blocks,
// for launches that would leave most CUs with one block or none (small batches at the deep stages)
template <int TT>
__global__ void __launch_bounds__(256, 2) gemm_bf16_kernel(GemmArgs g) {
  constexpr int BM = 32 * TT;
  constexpr int IMG = (BM + GBN) * GPITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * IMG];
  // XCD-aware order: the 8 XCDs take consecutive workgroup ids round-robin; give each XCD a contiguous run of tiles so
  // that the feature blocks of one token panel (same A tile) share an L2
  const int nb = gridDim.x;
  int id = blockIdx.x;
  {
    const int per = nb >> 3, rem = nb & 7, xcd = id & 7, loc = id >> 3;
    id = xcd < rem ? xcd * (per + 1) + loc : rem * (per + 1) + (xcd - rem) * per + loc;
  }
  const int bm = id / g.nblk_n, bn = id % g.nblk_n;
  const int m0 = bm * BM, n0 = bn * GBN;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int n = lane & 15, grp = lane >> 4;

  // staging: piece p = tid + 256 j -> row p / 8, 16-byte piece p % 8 of the chunk
  const unsigned char* arow[4];
  const unsigned char* wrow[4];
  int soff[4];
  const int pc = tid & 7;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (tid >> 3) + 32 * j;
    int ar = m0 + (r < BM ? r : BM - 1);
    if (ar > g.M - 1) ar = g.M - 1;
    int wr = n0 + r;
    if (wr > g.N - 1) wr = g.N - 1;
    arow[j] = reinterpret_cast<const unsigned char*>(g.a + (long long)ar * g.lda);
    wrow[j] = reinterpret_cast<const unsigned char*>(g.w + (long long)wr * g.K);
    soff[j] = r * GPITCH + pc * 16;
  }
  ffa_u32x4 ra[4], rw[4];
  auto load_regs = [&](int c) {
    int kb = (c * GBK + pc * 8) * 2;       // byte offset of this thread's piece in a row
    const int kmax = (g.K - 8) * 2;
    if (kb > kmax) kb = kmax;              // past the end of a ragged last chunk: any valid piece, never multiplied
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < TT) ra[j] = *reinterpret_cast<const ffa_u32x4*>(arow[j] + kb);
      rw[j] = *reinterpret_cast<const ffa_u32x4*>(wrow[j] + kb);
    }
  };
  auto store_lds = [&](int buf) {
    unsigned char* img = smem + buf * IMG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < TT) *reinterpret_cast<ffa_u32x4*>(img + soff[j]) = ra[j];
      *reinterpret_cast<ffa_u32x4*>(img + BM * GPITCH + soff[j]) = rw[j];
    }
  };

  ffa_f32x4 acc[4][TT];  // [feature tile][token tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[i][j] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + GBK - 1) / GBK;
  load_regs(0);
  store_lds(0);
  if (nk > 1) load_regs(1);
  __syncthreads();
  const int tok_base = (wm * 16 * TT + n) * GPITCH + grp * 16;
  const int fea_base = BM * GPITCH + (wn * 64 + n) * GPITCH + grp * 16;
  for (int c = 0; c < nk; ++c) {
    if (c + 1 < nk) store_lds((c + 1) & 1);
    if (c + 2 < nk) load_regs(c + 2);
    const unsigned char* img = smem + (c & 1) * IMG;
    const int ksteps = (g.K - c * GBK) >= GBK ? 2 : 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks < ksteps) {
        ffa_bf16x8 fw[4], ft[TT];
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const ffa_bf16x8*>(img + fea_base + i * 16 * GPITCH + ks * 64);
#pragma unroll
        for (int j = 0; j < TT; ++j) ft[j] = *reinterpret_cast<const ffa_bf16x8*>(img + tok_base + j * 16 * GPITCH + ks * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], ft[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias + activation in registers, the wave's 64 x 64 tile through LDS, residual + store by rows
  unsigned char* ep = smem + wave * (16 * TT * GEP);
  const int fcol0 = n0 + wn * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = fcol0 + i * 16 + grp * 4;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && f < g.N) {
      const float4 bv = *reinterpret_cast<const float4*>(g.bias + f);
      b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
    }
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[i][j][e] + b4[e];
        if (g.act == FFA_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
      }
      if (g.act == FFA_ACT_GELU && !g.aux) gemm_gelu_n<4>(v);
      uint2 pk;
      pk.x = ffa_pack_bf16x2(v[0], v[1]);
      pk.y = ffa_pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(ep + (j * 16 + n) * GEP + (i * 16 + grp * 4) * 2) = pk;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  const int fp = lane & 7;
  const int col = fcol0 + fp * 8;
#pragma unroll
  for (int j = 0; j < 2 * TT; ++j) {
    const int t = (lane >> 3) + 8 * j;
    const int row = m0 + wm * 16 * TT + t;
    if (row < g.M && col < g.N) {
      ffa_u32x4 v = *reinterpret_cast<const ffa_u32x4*>(ep + t * GEP + fp * 16);
      if (g.residual || g.aux || g.row_scale) {  // block-uniform
        float o[8];
        ffa_load8<ffa_bf16>(reinterpret_cast<const ffa_bf16*>(&v), o);
        if (g.aux) {
          if (g.act == FFA_ACT_GELU) {  // keep the (bf16-rounded) pre-activation for the backward pass
            *reinterpret_cast<ffa_u32x4*>(g.aux + (long long)row * g.ldaux + col) = v;
            gemm_gelu_n<8>(o);
          } else if (g.act == FFA_ACT_DGELU) {
            float u[8];
            ffa_load8<ffa_bf16>(g.aux + (long long)row * g.ldaux + col, u);
            gemm_dgelu_n<8>(o, u);
          }
        }
        if (g.row_scale) {
          const float sc = g.row_scale[row / g.rows_per_scale];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] *= sc;
        }
        if (g.residual) {
          float r[8];
          ffa_load8<ffa_bf16>(g.residual + (long long)row * g.ldr + col, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += r[e];
        }
        ffa_store8<ffa_bf16>(g.out + (long long)row * g.ldc + col, o);
      } else {
        *reinterpret_cast<ffa_u32x4*>(g.out + (long long)row * g.ldc + col) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same GEMM on a 256 x 256 block tile for the MFMA-bound shapes (K >= 256, N a multiple of 256 or large):
// 8 waves as 2 (tokens) x 4 (features), 128 tokens x 64 features per wave (8 x 4 MFMA tiles, 128 accumulator VGPRs:
// 2.7 MFMAs per ds_read_b128 instead of 2), BK = 32, operands streamed global -> LDS by LDS-DMA
// (global_load_lds_dwordx4 from inline asm: with the builtin hipcc stops counting lgkmcnt) into a
// four-stage ring of (256 + 256) rows x 64 bytes, three tiles ahead, counted vmcnt (never 0 in the loop), one raw
// s_barrier per 32 MFMAs per wave.  The DMA writes linearly (16 rows x 64 bytes per wave instruction), so the
// bank-conflict swizzle is applied on the SOURCE side and again on the read: 16-byte chunk c of row r lives at chunk
// c ^ f(r), f = [0, 3, 2, 1][(r >> 2) & 3] -- the four 16-lane groups of a ds_read_b128 (lanes {0-3, 12-15, 20-27}, ...)
// then touch 16 distinct 16-byte slots of the 256-byte bank row.
namespace {
constexpr int HBM_ = 256, HBN_ = 256, HBK = 32;
constexpr int HROWB = 64;                       // bytes per staged row
constexpr int HSTAGE = (HBM_ + HBN_) * HROWB;   // 32 KB
constexpr int HSTAGES = 4;
}  // namespace

__device__ __forceinline__ int gemm256_swz(int r) { return (4 - ((r >> 2) & 3)) & 3; }

__device__ __forceinline__ void gemm_dma16(const unsigned char* src, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void gemm_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(N) : "memory");
}

__global__ void __launch_bounds__(512, 1) gemm256_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[HSTAGES * HSTAGE];
  const int nb = gridDim.x;
  int id = blockIdx.x;
  {
    const int per = nb >> 3, rem = nb & 7, xcd = id & 7, loc = id >> 3;
    id = xcd < rem ? xcd * (per + 1) + loc : rem * (per + 1) + (xcd - rem) * per + loc;
  }
  const int bm = id / g.nblk_n, bn = id % g.nblk_n;
  const int m0 = bm * HBM_, n0 = bn * HBN_;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int n = lane & 15, grp = lane >> 4;

  // DMA: this wave fills token rows [32 wave, 32 wave + 32) and feature rows [32 wave, +32) of a stage, 16 rows per
  // instruction; lane i writes row i / 4, chunk i % 4 and therefore fetches chunk (i % 4) ^ f(row)
  const unsigned char* src[4];
  unsigned dst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 32 + (j & 1) * 16 + (lane >> 2);   // row inside the 256-row operand tile
    const int c = (lane & 3) ^ gemm256_swz(r);
    if (j < 2) {
      int row = m0 + r;
      if (row > g.M - 1) row = g.M - 1;
      src[j] = reinterpret_cast<const unsigned char*>(g.a + (long long)row * g.lda) + c * 16;
    } else {
      int row = n0 + r;
      if (row > g.N - 1) row = g.N - 1;
      src[j] = reinterpret_cast<const unsigned char*>(g.w + (long long)row * g.K) + c * 16;
    }
    dst[j] = (unsigned)((j < 2 ? 0 : HBM_ * HROWB) + (wave * 32 + (j & 1) * 16) * HROWB);
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  auto issue = [&](int t) {
    const unsigned base = lds0 + (unsigned)((t & (HSTAGES - 1)) * HSTAGE);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      gemm_dma16(src[j] + (long long)t * (HBK * 2),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(base + dst[j])));
  };

  ffa_f32x4 acc[4][8];  // [feature tile][token tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / HBK;
  issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
  // fragment read offsets: row (tile * 16 + n), chunk grp ^ f(row); f depends on the row modulo 16 only
  const int fsw = (grp ^ gemm256_swz(n)) * 16;
  const int tok_off = (wm * 128 + n) * HROWB + fsw;
  const int fea_off = HBM_ * HROWB + (wn * 64 + n) * HROWB + fsw;
  for (int t = 0; t < nk; ++t) {
    // tile t has landed for every wave once each wave's own pieces are retired and the barrier is passed; the
    // barrier also says that everyone is done reading tile t - 1, whose stage tile t + 3 is about to overwrite
    if (t + 2 < nk) gemm_wait_barrier<8>();
    else if (t + 1 < nk) gemm_wait_barrier<4>();
    else gemm_wait_barrier<0>();
    if (t + 3 < nk) issue(t + 3);
    const unsigned char* img = smem + (t & (HSTAGES - 1)) * HSTAGE;
    ffa_bf16x8 fw[4], ft[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const ffa_bf16x8*>(img + fea_off + i * 16 * HROWB);
#pragma unroll
    for (int j = 0; j < 8; ++j) ft[j] = *reinterpret_cast<const ffa_bf16x8*>(img + tok_off + j * 16 * HROWB);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], ft[j], acc[i][j], 0, 0, 0);
  }
  __syncthreads();  // every wave is done with the ring: it becomes the epilogue's staging area

  // ---- epilogue, two halves of 64 tokens per wave: registers -> LDS (bf16) -> whole 128-byte row segments
  unsigned char* ep = smem + wave * (64 * GEP);
  const int fcol0 = n0 + wn * 64;
  const int fp = lane & 7;
  const int col = fcol0 + fp * 8;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = fcol0 + i * 16 + grp * 4;
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      if (g.bias && f < g.N) {
        const float4 bv = *reinterpret_cast<const float4*>(g.bias + f);
        b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][half * 4 + j][e] + b4[e];
          if (g.act == FFA_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
        }
        if (g.act == FFA_ACT_GELU && !g.aux) gemm_gelu_n<4>(v);
        uint2 pk;
        pk.x = ffa_pack_bf16x2(v[0], v[1]);
        pk.y = ffa_pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(ep + (j * 16 + n) * GEP + (i * 16 + grp * 4) * 2) = pk;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tt = (lane >> 3) + 8 * j;
      const int row = m0 + wm * 128 + half * 64 + tt;
      if (row < g.M && col < g.N) {
        ffa_u32x4 v = *reinterpret_cast<const ffa_u32x4*>(ep + tt * GEP + fp * 16);
        if (g.residual || g.aux || g.row_scale) {  // block-uniform
          float o[8];
          ffa_load8<ffa_bf16>(reinterpret_cast<const ffa_bf16*>(&v), o);
          if (g.aux) {
            if (g.act == FFA_ACT_GELU) {
              *reinterpret_cast<ffa_u32x4*>(g.aux + (long long)row * g.ldaux + col) = v;
              gemm_gelu_n<8>(o);
            } else if (g.act == FFA_ACT_DGELU) {
              float u[8];
              ffa_load8<ffa_bf16>(g.aux + (long long)row * g.ldaux + col, u);
              gemm_dgelu_n<8>(o, u);
            }
          }
          if (g.row_scale) {
            const float sc = g.row_scale[row / g.rows_per_scale];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= sc;
          }
          if (g.residual) {
            float r[8];
            ffa_load8<ffa_bf16>(g.residual + (long long)row * g.ldr + col, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += r[e];
          }
          ffa_store8<ffa_bf16>(g.out + (long long)row * g.ldc + col, o);
        } else {
          *reinterpret_cast<ffa_u32x4*>(g.out + (long long)row * g.ldc + col) = v;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// f32 parity mode of the token GEMM: plain FMA, 64 x 64 tile, the same epilogue order (bias -> activation -> row scale
// -> residual).  It exists so that the transformer layers can be TRAINED in f32 and every gradient compared with the
// CPU oracle at f32 tolerance (tests/test_swin_gpu.py); it is not a speed path (libm erff, no MFMA).
struct GemmF32Args {
  const float* a;
  const float* w;
  const float* bias;
  const float* residual;
  float* out;
  float* aux;
  const float* row_scale;
  long long lda, ldr, ldc, ldaux;
  int M, K, N, act, rows_per_scale;
};

__device__ __forceinline__ float gemm_gelu_f32(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gemm_dgelu_f32(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * expf(-0.5f * x * x) * 0.39894228040143267794f;
}

__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmF32Args g) {
  __shared__ float sa[16][68];  // [k][token]
  __shared__ float sw[16][68];  // [k][feature]
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long long m0 = (long long)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    for (int i = threadIdx.x; i < 1024; i += 256) {
      const int r = i >> 4, c = i & 15;
      const int k = k0 + c;
      const long long m = m0 + r;
      const int n = n0 + r;
      sa[c][r] = (m < g.M && k < g.K) ? g.a[m * g.lda + k] : 0.f;
      sw[c][r] = (n < g.N && k < g.K) ? g.w[(long long)n * g.K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[4], wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = sa[kk][ty * 4 + i];
        wv[i] = sw[kk][tx * 4 + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * wv[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
    const float sc = g.row_scale ? g.row_scale[m / g.rows_per_scale] : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= g.N) continue;
      float v = acc[i][j] + (g.bias ? g.bias[n] : 0.f);
      if (g.act == FFA_ACT_GELU) {
        if (g.aux) g.aux[m * g.ldaux + n] = v;
        v = gemm_gelu_f32(v);
      } else if (g.act == FFA_ACT_DGELU) {
        v *= gemm_dgelu_f32(g.aux[m * g.ldaux + n]);
      } else if (g.act == FFA_ACT_RELU) {
        v = fmaxf(v, 0.f);
      }
      v *= sc;
      if (g.residual) v += g.residual[m * g.ldr + n];
      g.out[m * g.ldc + n] = v;
    }
  }
}

// 256-tile kernel when the shape is MFMA-bound and fills the chip; FFA_GEMM_TILE=128|256 overrides (A/B runs)
static bool gemm_use_256(int M, int K, int N) {
  static const char* force = getenv("FFA_GEMM_TILE");
  if (force && force[0] == '1') return false;
  const bool ok = (K % HBK == 0) && K >= 64 && M >= 256 && N >= 128;
  if (force && force[0] == '2') return ok;
  if (!ok || K < 256) return false;
  const long long blocks = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const double waste = (double)(((N + 255) / 256) * 256) / (double)N;
  return blocks >= 240 && waste <= 1.15;  // one 512-thread block per CU: fewer blocks than CUs leave the chip idle
}

extern "C" int ffa_linear_ex(int dtype, const void* a, long long lda, const void* w, const float* bias,
                             const void* residual, long long ldr, void* out, long long ldc, int M, int K, int N, int act,
                             void* aux, long long ldaux, const float* row_scale, int rows_per_scale,
                             hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "linear: dtype %d", dtype);
  FFA_REQUIRE(a && w && out && M > 0 && K > 0 && N > 0, "linear: bad arguments");
  if (dtype == FFA_F32) {
    FFA_REQUIRE(lda >= K && ldc >= N && (!residual || ldr >= N) && (!aux || ldaux >= N), "linear: row pitches must cover the row");
    FFA_REQUIRE(act == FFA_ACT_NONE || act == FFA_ACT_GELU || act == FFA_ACT_RELU || (act == FFA_ACT_DGELU && aux),
                "linear: activation %d", act);
    FFA_REQUIRE(!row_scale || rows_per_scale > 0, "linear: rows_per_scale must be positive");
    GemmF32Args f;
    f.a = (const float*)a; f.w = (const float*)w; f.bias = bias; f.residual = (const float*)residual;
    f.out = (float*)out; f.aux = (float*)aux; f.row_scale = row_scale;
    f.lda = lda; f.ldr = ldr; f.ldc = ldc; f.ldaux = ldaux;
    f.M = M; f.K = K; f.N = N; f.act = act; f.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    FFA_REQUIRE((N + 63) / 64 < 65536, "linear: N too large");
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, stream, f);
    return ffa_check_launch("linear");
  }
  FFA_REQUIRE(K % 32 == 0 && N % 8 == 0, "linear: K = %d must be a multiple of 32 and N = %d of 8", K, N);
  FFA_REQUIRE(lda >= K && lda % 8 == 0 && ldc >= N && ldc % 8 == 0 && (!residual || (ldr >= N && ldr % 8 == 0)) &&
                  (!aux || (ldaux >= N && ldaux % 8 == 0)),
              "linear: row pitches must cover the row and be multiples of 8 elements");
  FFA_REQUIRE(act == FFA_ACT_NONE || act == FFA_ACT_GELU || act == FFA_ACT_RELU || (act == FFA_ACT_DGELU && aux),
              "linear: activation %d", act);
  FFA_REQUIRE(!(aux && (act == FFA_ACT_NONE || act == FFA_ACT_RELU)), "linear: an auxiliary tensor needs the GELU / DGELU epilogue");
  FFA_REQUIRE(!row_scale || rows_per_scale > 0, "linear: rows_per_scale must be positive");
  GemmArgs g;
  g.a = (const ffa_bf16*)a;
  g.w = (const ffa_bf16*)w;
  g.bias = bias;
  g.residual = (const ffa_bf16*)residual;
  g.out = (ffa_bf16*)out;
  g.aux = (ffa_bf16*)aux;
  g.row_scale = row_scale;
  g.lda = lda; g.ldr = ldr; g.ldc = ldc; g.ldaux = ldaux;
  g.M = M; g.K = K; g.N = N; g.act = act;
  g.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  if (gemm_use_256(M, K, N)) {
    g.nblk_n = (N + HBN_ - 1) / HBN_;
    const long long blocks = (long long)((M + HBM_ - 1) / HBM_) * g.nblk_n;
    FFA_REQUIRE(blocks < (1LL << 31), "linear: grid too large");
    hipLaunchKernelGGL(gemm256_bf16_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, g);
    return ffa_check_launch("linear");
  }
  g.nblk_n = (N + GBN - 1) / GBN;
  const long long blocks = (long long)((M + GBM - 1) / GBM) * g.nblk_n;
  FFA_REQUIRE(blocks < (1LL << 31), "linear: grid too large");
  if (blocks < 384) {  // fewer than 1.5 blocks per CU: 64-token tiles double the grid
    const long long blocks64 = (long long)((M + 63) / 64) * g.nblk_n;
    hipLaunchKernelGGL(gemm_bf16_kernel<2>, dim3((unsigned)blocks64), dim3(256), 0, stream, g);
  } else {
    hipLaunchKernelGGL(gemm_bf16_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, g);
  }
  return ffa_check_launch("linear");
}

extern "C" int ffa_linear(int dtype, const void* a, long long lda, const void* w, const float* bias, const void* residual,
                          long long ldr, void* out, long long ldc, int M, int K, int N, int act, hipStream_t stream) {
  return ffa_linear_ex(dtype, a, lda, w, bias, residual, ldr, out, ldc, M, K, N, act, nullptr, 0, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient GEMM of nn.Linear:  dW[n][k] = sum_m dy[m][n] * x[m][k]   (f32 out, [N][K] like nn.Linear.weight)
//
// The contraction runs over the TOKEN index m, the slow dimension of both row-major operands, so the MFMA fragments
// (8 consecutive m per lane) are transposed reads: the [64 m][128 n] / [64 m][128 k] tiles are staged row-major in LDS
// and read with ds_read_b64_tr_b16 (4 tokens x 16 columns per 16-lane group and instruction).  Row pitch 288 bytes and
// the k-slot order  slot 8g + j  <->  token (j < 4 ? 4g + j : 16 + 4g + j - 4)  of a 32-token step put the 8 rows a
// 32-lane half touches on 8 disjoint bank octets (conflict-free).  Block: 128 n x 128 k of dW over a contiguous range of
// 64-token chunks (split over the tokens: [split][N][K] f32 slabs in the workspace, summed in a fixed order by
// gemm_tn_reduce_kernel -- deterministic); 4 waves as 2 x 2, 64 x 64 each; double-buffered LDS, register prefetch.
// Bound: HBM at the thin stages (each operand is read once; the output is tiny), MFMA for the 768-wide ones.
struct GemmTnArgs {
  const ffa_bf16* dy;  // [M][ldy]
  const ffa_bf16* x;   // [M][ldx]
  float* slab;         // [splits][N][K]
  float* bias_slab;    // nullable: [splits][N] column sums of dy (nn.Linear's bias gradient), written by the bk = 0 blocks
  long long ldy, ldx;
  int M, N, K, nblk_n, nblk_k, chunks_per_split;
};

namespace {
constexpr int TBM = 64;            // tokens per chunk
constexpr int TPITCH = 288;        // bytes per staged row (128 columns + 32 pad)
constexpr int TIMG = 2 * TBM * TPITCH;
}  // namespace

__device__ __forceinline__ ffa_s16x4 gemm_read_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) ffa_s16x4*)(const_cast<unsigned char*>(p)));
}

__global__ void __launch_bounds__(256, 2) gemm_tn_bf16_kernel(GemmTnArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TIMG];
  int id = blockIdx.x;
  const int tiles = g.nblk_n * g.nblk_k;
  const int split = id / tiles;
  id -= split * tiles;
  const int bn = id / g.nblk_k, bk = id % g.nblk_k;
  const int n0 = bn * 128, k0 = bk * 128;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wn = wave >> 1, wk = wave & 1;
  const int grp = lane >> 4;

  // staging: a chunk is 64 rows x 16 pieces of 16 bytes per operand = 1024 pieces, 4 per thread and operand
  const int pc = tid & 15;  // 16-byte piece (8 columns) of the 128-column tile row
  int ycol = n0 + pc * 8, xcol = k0 + pc * 8;
  if (ycol > g.N - 8) ycol = g.N - 8;  // past the matrix: any valid piece, its products land outside dW
  if (xcol > g.K - 8) xcol = g.K - 8;
  const long long c0 = (long long)split * g.chunks_per_split;
  long long c1 = c0 + g.chunks_per_split;
  const long long nchunks = ((long long)g.M + TBM - 1) / TBM;
  if (c1 > nchunks) c1 = nchunks;
  ffa_u32x4 ry[4], rx[4];
  // bias gradient: this thread's pieces always cover the same 8 columns of dy (pc is fixed), so their column sums
  // accumulate in registers as the chunks stream through
  const bool want_bias = g.bias_slab != nullptr && bk == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto load_regs = [&](long long c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long m = c * TBM + (tid >> 4) + 16 * j;
      const bool live = m < g.M;
      if (!live) m = g.M - 1;
      ry[j] = *reinterpret_cast<const ffa_u32x4*>(g.dy + m * g.ldy + ycol);
      rx[j] = *reinterpret_cast<const ffa_u32x4*>(g.x + m * g.ldx + xcol);
      if (!live) ry[j] = ffa_u32x4{0u, 0u, 0u, 0u};  // rows past M contribute nothing
    }
  };
  auto store_lds = [&](int buf) {
    unsigned char* img = smem + buf * TIMG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (tid >> 4) + 16 * j;
      *reinterpret_cast<ffa_u32x4*>(img + r * TPITCH + pc * 16) = ry[j];
      *reinterpret_cast<ffa_u32x4*>(img + TBM * TPITCH + r * TPITCH + pc * 16) = rx[j];
      if (want_bias) {  // block-uniform
        float v[8];
        ffa_load8<ffa_bf16>(reinterpret_cast<const ffa_bf16*>(&ry[j]), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += v[e];
      }
    }
  };
  ffa_f32x4 acc[4][4];  // [n tile][k tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};

  if (c0 < c1) {
    load_regs(c0);
    store_lds(0);
    if (c0 + 1 < c1) load_regs(c0 + 1);
  }
  __syncthreads();
  // transposed-read address of this lane inside a 4-token x 16-column block: token (lane / 4) % 4, columns 4 (lane % 4)
  const int tr_off = ((lane >> 2) & 3) * TPITCH + (lane & 3) * 8;
  const int yb = grp * 4 * TPITCH + (wn * 64) * 2 + tr_off;                  // + step * 32 rows, + tile * 32 bytes
  const int xb = TBM * TPITCH + grp * 4 * TPITCH + (wk * 64) * 2 + tr_off;
  int buf = 0;
  for (long long c = c0; c < c1; ++c) {
    if (c + 1 < c1) store_lds(buf ^ 1);
    if (c + 2 < c1) load_regs(c + 2);
    const unsigned char* img = smem + buf * TIMG;
#pragma unroll
    for (int st = 0; st < 2; ++st) {  // two 32-token steps per chunk
      ffa_bf16x8 fy[4], fx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned char* py = img + yb + st * 32 * TPITCH + i * 32;
        const unsigned char* px = img + xb + st * 32 * TPITCH + i * 32;
        const ffa_s16x4 y0 = gemm_read_tr16(py), y1 = gemm_read_tr16(py + 16 * TPITCH);
        const ffa_s16x4 x0 = gemm_read_tr16(px), x1 = gemm_read_tr16(px + 16 * TPITCH);
        ffa_u32x4 vy, vx;
        vy.x = __builtin_bit_cast(ffa_u32x2, y0).x; vy.y = __builtin_bit_cast(ffa_u32x2, y0).y;
        vy.z = __builtin_bit_cast(ffa_u32x2, y1).x; vy.w = __builtin_bit_cast(ffa_u32x2, y1).y;
        vx.x = __builtin_bit_cast(ffa_u32x2, x0).x; vx.y = __builtin_bit_cast(ffa_u32x2, x0).y;
        vx.z = __builtin_bit_cast(ffa_u32x2, x1).x; vx.w = __builtin_bit_cast(ffa_u32x2, x1).y;
        fy[i] = __builtin_bit_cast(ffa_bf16x8, vy);
        fx[i] = __builtin_bit_cast(ffa_bf16x8, vx);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    buf ^= 1;
  }
  if (want_bias) {  // 16 row-threads per column piece -> one sum per column, through LDS (free after the loop)
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + pc * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += red[r * 128 + tid];
      // columns clamped to the matrix edge were loaded twice: only the piece's own columns are written
      if (n0 + tid < g.N) g.bias_slab[(long long)split * g.N + n0 + tid] = v;
    }
  }
  // D[row n = 4 grp + e][col k = lane % 16] -> slab[split][n][k]
  float* out = g.slab + (long long)split * g.N * g.K;
  const int kc = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + kc;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + wn * 64 + i * 16 + grp * 4 + e;
        if (n < g.N && k < g.K) out[(long long)n * g.K + k] = acc[i][j][e];
      }
    }
}

// dw[i] (+)= sum_s slab[s][i]: a block owns 8 consecutive float4 columns; its 32 split-lanes take the slabs
// s = lane, lane + 32, ... (adjacent threads read adjacent 16-byte pieces of one slab: full 128-byte lines), then the
// lanes are combined by xor-shuffles inside each wave and across the four waves through LDS -- a fixed order
__global__ void __launch_bounds__(256) gemm_tn_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                             long long nk, int splits, int accumulate) {
  __shared__ float4 part[4][8];
  const int col = threadIdx.x & 7, sl = threadIdx.x >> 3;  // split lane 0..31; a wave holds 8 of them
  const long long i = (blockIdx.x * 8LL + col) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < nk)
    for (int s = sl; s < splits; s += 32) {
      const float4 v = *reinterpret_cast<const float4*>(slab + (long long)s * nk + i);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    a.x += __shfl_xor(a.x, o, 64);
    a.y += __shfl_xor(a.y, o, 64);
    a.z += __shfl_xor(a.z, o, 64);
    a.w += __shfl_xor(a.w, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 8) part[wave][col] = a;
  __syncthreads();
  if (threadIdx.x < 8 && i < nk) {
    float4 r = accumulate ? *reinterpret_cast<const float4*>(dw + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      r.x += part[w][col].x; r.y += part[w][col].y; r.z += part[w][col].z; r.w += part[w][col].w;
    }
    *reinterpret_cast<float4*>(dw + i) = r;
  }
}

// f32 parity mode of the weight-gradient GEMM: 64 n x 64 k tile per block over a contiguous token range, plain FMA,
// the same [split][N][K] (+ [split][N]) slabs and the same fixed-order reduce as the bf16 kernel
__global__ void __launch_bounds__(256) gemm_tn_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ slab, float* __restrict__ bias_slab,
                                                          long long ldy, long long ldx, int M, int N, int K, int nblk_n,
                                                          int nblk_k, int rows_per_split) {
  __shared__ float sdy[16][68];  // [token][n]
  __shared__ float sx[16][68];   // [token][k]
  int b = blockIdx.x;
  const int bk = b % nblk_k;
  b /= nblk_k;
  const int bn = b % nblk_n;
  const int split = b / nblk_n;
  const int n0 = bn * 64, k0 = bk * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // tx -> k, ty -> n
  const long long mbeg = (long long)split * rows_per_split;
  long long mend = mbeg + rows_per_split;
  if (mend > M) mend = M;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  float bsum = 0.f;
  for (long long m0 = mbeg; m0 < mend; m0 += 16) {
    for (int i = threadIdx.x; i < 1024; i += 256) {
      const int r = i >> 6, c = i & 63;
      const long long m = m0 + r;
      sdy[r][c] = (m < mend && n0 + c < N) ? dy[m * ldy + n0 + c] : 0.f;
      sx[r][c] = (m < mend && k0 + c < K) ? x[m * ldx + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int mm = 0; mm < 16; ++mm) {
      float dv[4], xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dv[i] = sdy[mm][ty * 4 + i];
        xv[i] = sx[mm][tx * 4 + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += dv[i] * xv[j];
    }
    if (bias_slab && bk == 0 && threadIdx.x < 64) {
#pragma unroll
      for (int mm = 0; mm < 16; ++mm) bsum += sdy[mm][threadIdx.x];
    }
    __syncthreads();
  }
  float* out = slab + (long long)split * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + ty * 4 + i, k = k0 + tx * 4 + j;
      if (n < N && k < K) out[(long long)n * K + k] = acc[i][j];
    }
  if (bias_slab && bk == 0 && threadIdx.x < 64 && n0 + (int)threadIdx.x < N)
    bias_slab[(long long)split * N + n0 + threadIdx.x] = bsum;
}

static void gemm_tn_plan_f32(int M, int N, int K, int* nblk_n, int* nblk_k, int* splits, int* rows_per_split) {
  *nblk_n = (N + 63) / 64;
  *nblk_k = (K + 63) / 64;
  const long long tiles = (long long)*nblk_n * *nblk_k;
  const long long chunks = ((long long)M + 63) / 64;
  long long s = (1024 + tiles - 1) / tiles;
  if (s > chunks) s = chunks;
  if (s < 1) s = 1;
  const long long per = (chunks + s - 1) / s;
  *splits = (int)((chunks + per - 1) / per);
  *rows_per_split = (int)(per * 64);
}

static void gemm_tn_plan(int M, int N, int K, int* nblk_n, int* nblk_k, int* splits, int* cps) {
  *nblk_n = (N + 127) / 128;
  *nblk_k = (K + 127) / 128;
  const long long tiles = (long long)*nblk_n * *nblk_k;
  const long long chunks = ((long long)M + TBM - 1) / TBM;
  long long s = (768 + tiles - 1) / tiles;  // about three blocks per CU
  if (s > chunks) s = chunks;
  if (s < 1) s = 1;
  long long per = (chunks + s - 1) / s;
  s = (chunks + per - 1) / per;
  *splits = (int)s;
  *cps = (int)per;
}

extern "C" long long ffa_linear_wgrad_workspace_bytes(int M, int N, int K) {
  int a, b, s, c, sf;
  gemm_tn_plan(M, N, K, &a, &b, &s, &c);
  gemm_tn_plan_f32(M, N, K, &a, &b, &sf, &c);  // the entry point takes no dtype: room for either plan
  if (sf > s) s = sf;
  return (long long)s * ((long long)N * K + N) * (long long)sizeof(float);
}

extern "C" int ffa_linear_wgrad(int dtype, const void* x, long long ldx, const void* dy, long long ldy, float* dw,
                                float* dbias, int M, int K, int N, int accumulate, void* workspace,
                                long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "linear_wgrad: dtype %d", dtype);
  FFA_REQUIRE(x && dy && dw && M > 0 && K >= 8 && N >= 8 && K % 8 == 0 && N % 8 == 0, "linear_wgrad: bad arguments");
  FFA_REQUIRE(ldx >= K && ldx % 8 == 0 && ldy >= N && ldy % 8 == 0, "linear_wgrad: row pitches must cover the rows");
  FFA_REQUIRE(((long long)N * K) % 4 == 0 && N % 4 == 0, "linear_wgrad: N * K and N must be multiples of 4");
  if (!workspace || workspace_bytes < ffa_linear_wgrad_workspace_bytes(M, N, K)) {
    ffa_set_error("linear_wgrad: workspace of %lld bytes needed", ffa_linear_wgrad_workspace_bytes(M, N, K));
    return FFA_ERR_WORKSPACE;
  }
  if (dtype == FFA_F32) {
    int nbn, nbk, splits, rps;
    gemm_tn_plan_f32(M, N, K, &nbn, &nbk, &splits, &rps);
    float* slab = (float*)workspace;
    float* bias_slab = dbias ? slab + (long long)splits * N * K : nullptr;
    hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3((unsigned)((long long)nbn * nbk * splits)), dim3(256), 0, stream,
                       (const float*)dy, (const float*)x, slab, bias_slab, ldy, ldx, M, N, K, nbn, nbk, rps);
    const long long nk = (long long)N * K;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((nk / 4 + 7) / 8)), dim3(256), 0, stream,
                       (const float*)slab, dw, nk, splits, accumulate ? 1 : 0);
    if (dbias)
      hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((N / 4 + 7) / 8)), dim3(256), 0, stream,
                         (const float*)bias_slab, dbias, (long long)N, splits, accumulate ? 1 : 0);
    return ffa_check_launch("linear_wgrad");
  }
  GemmTnArgs g;
  g.dy = (const ffa_bf16*)dy;
  g.x = (const ffa_bf16*)x;
  g.slab = (float*)workspace;
  g.ldy = ldy; g.ldx = ldx;
  g.M = M; g.N = N; g.K = K;
  int splits;
  gemm_tn_plan(M, N, K, &g.nblk_n, &g.nblk_k, &splits, &g.chunks_per_split);
  g.bias_slab = dbias ? g.slab + (long long)splits * N * K : nullptr;
  const long long blocks = (long long)g.nblk_n * g.nblk_k * splits;
  hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g);
  const long long nk = (long long)N * K;
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((nk / 4 + 7) / 8)), dim3(256), 0, stream,
                     (const float*)workspace, dw, nk, splits, accumulate ? 1 : 0);
  if (dbias)
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((N / 4 + 7) / 8)), dim3(256), 0, stream,
                       (const float*)g.bias_slab, dbias, (long long)N, splits, accumulate ? 1 : 0);
  return ffa_check_launch("linear_wgrad");
}
