// Implicit-GEMM convolution on MFMA for gfx950, NHWC, bf16 or f32 operands, f32 accumulate.
//
// Replaces the cuDNN/ATen conv2d reached through segmentation_models_pytorch from
// flair_hub/models/monotemp_model.py:68-92 (encoder/decoder conv stack called at
// flair_hub/models/flair_model.py:376 and :417-419).  The same kernel serves
//   * forward conv (3x3 s1/s2, 1x1 s1/s2, 7x7 s2) with fused bias / residual add / ReLU epilogue
//   * dgrad: a stride-1 conv over dy with flipped+transposed packed weights; for stride-2
//     layers dy is read through a virtual zero-insertion (dil = 2), never materialised.
//
// GEMM orientation: D[co][pixel] = W[co][k] * X[k][pixel], k = (tap, channel).  Weights are the
// MFMA A operand (rows), pixels the B operand (columns), so every lane ends up holding runs of
// consecutive output channels of ONE pixel -> 16-byte NHWC stores without an LDS transpose.
//
// Per block: BCO output channels x (TH x TW) output pixels of one image.  Per chunk (one 32-byte
// k-step of input channels, RG kernel rows): the input halo tile and the weight slab are staged
// global -> registers -> LDS (zero fill for padding / zero-insertion happens in registers), the
// next chunk's global loads are issued before the MFMAs of the current one.
// LDS images:  halo  [IH*IW pixels][32 B + 16 B pad]  (pitch 48 B: conflict-free ds_read_b128
//                     for 16 consecutive pixels, 3 is coprime with the 16 slots of a bank row)
//              weight [BCO rows][TAPS*32 B + 16 B pad] (odd number of 16-B slots per row)
#include "ffa_common.h"

#include <stdlib.h>

#ifndef FFA_CONV_READS_FIRST
#define FFA_CONV_READS_FIRST 0
#endif

// Developer instrumentation (never built by flairhip/build.py): -DFFA_CONV_TRACE=1 makes wave 0 of the first 64
// blocks log s_memtime at every phase boundary into a device buffer read back with ffa_conv_trace_read.
#ifndef FFA_CONV_TRACE
#define FFA_CONV_TRACE 0
#endif
#if FFA_CONV_TRACE
#ifndef FFA_CONV_TRACE_FIRST
#define FFA_CONV_TRACE_FIRST 0
#endif
__device__ long long ffa_conv_trace_buf[64 * 256];
#define FFA_TRACE(slot_)                                                                    \
  if (trace_on) {                                                                           \
    if (trace_n < 256) ffa_conv_trace_buf[(blockIdx.x - FFA_CONV_TRACE_FIRST) * 256 + trace_n] = (long long)(slot_) << 56 | (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFFFLL); \
    ++trace_n;                                                                              \
  }
#else
#define FFA_TRACE(slot_)
#endif

struct ConvArgs {
  const void* in;
  const void* w;
  void* out;
  const float* bias;  // [>= co block coverage] or null
  float* stats;       // [npt][2][Co] per-tile channel sums / sums of squares of the stored output, or null
  const void* in2;    // UP kernels: the skip tensor [B][Hi][Wi][C2]; `in` is then the low-res map [B][Hi/2][Wi/2][C1]
  int c1_bytes;       // UP kernels: bytes of one pixel of `in` (C1 * sizeof(T)); the virtual input has Ci = C1 + C2
  // BatchNorm-backward partials (bnx != null; needs stats): the tensor written is the gradient dy of
  // y = relu(x * bn_sc + bn_sh); stats then receives per tile (sum g, sum g*x) with g = dy masked by y > 0
  const void* bnx;    // the pre-normalisation tensor x, same shape / pitch as out
  const float* bn_sc;
  const float* bn_sh;
  void* out2;         // split epilogue (c1_out > 0): output channels >= c1_out go here, pixel pitch Co - c1_out
  int c1_out;         // split epilogue: output channels < c1_out are 2x2-sum-pooled into `out` [B][Ho/2][Wo/2][c1_out]
  const void* res;    // same layout as out, or null
  int B, Hi, Wi, Ci;  // stored input dims (Ci = channel pitch)
  int Ho, Wo, Co;     // Co = stored output channel pitch
  int pad, dil, relu;
  int nchunks;  // Ci * sizeof(T) / 32
  int tiles_x, tiles_y, npt, ncb;
};

template <typename T>
struct Mma;
template <>
struct Mma<ffa_bf16> {
  static __device__ __forceinline__ void run(const ffa_u32x4& a, const ffa_u32x4& b, ffa_f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ffa_bf16x8, a),
                                                __builtin_bit_cast(ffa_bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  static __device__ __forceinline__ void run(const ffa_u32x4& a, const ffa_u32x4& b, ffa_f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// HK = 32-byte k-steps of channels staged per halo fill.  With HK = 4 every halo pixel is fetched as one whole
// 128-byte line (64 bf16 channels) and then multiplied over four weight sub-chunks; with HK = 1 each line is
// re-requested from L2 four times, 32 bytes at a time -- measured L2-bound on the 128..256-channel layers.
template <int KH, int KW, int STRIDE, int RG, int BCO, int WCO, int WPX, int TH, int TW, int HK>
struct ConvGeom {
  static constexpr int NTHR = 64 * WCO * WPX;
  static constexpr int NPX = TH * TW;
  static constexpr int WAVE_CO = BCO / WCO;
  static constexpr int MT = WAVE_CO / 32;
  static constexpr int WAVE_PX = NPX / WPX;
  static constexpr int NT = WAVE_PX / 32;
  static constexpr bool ONE = (KH == 1 && KW == 1);
  static constexpr int LS = ONE ? 1 : STRIDE;     // LDS pixel step between neighbouring outputs
  static constexpr int GSTEP = ONE ? STRIDE : 1;  // global pixel step between neighbouring halo pixels
  static constexpr int NRG = KH / RG;
  static constexpr int IH = (TH - 1) * LS + RG;
  static constexpr int IW = (TW - 1) * LS + KW;
  static constexpr int PP = HK * 32 + 16;  // 3 / 5 / 9 slots of 16 B: odd -> conflict-free ds_read_b128
  static constexpr int HPP = 2 * HK;       // 16-byte pieces per halo pixel
  static constexpr int TAPS = RG * KW;
  static constexpr int WP = TAPS * 32 + 16;
  static constexpr int HALO_BYTES = IH * IW * PP;
  static constexpr int W_BYTES = BCO * WP;
  static constexpr int LDS_BYTES = HALO_BYTES + W_BYTES;
  static constexpr int W_PIECES = BCO * TAPS * 2;
  static constexpr int H_PIECES = IH * IW * HPP;
  static constexpr int NWP = (W_PIECES + NTHR - 1) / NTHR;
  static constexpr int NHP = (H_PIECES + NTHR - 1) / NTHR;
  static_assert(KH % RG == 0, "row group must divide kernel height");
  static_assert(HK == 1 || RG == KH, "deep halo staging needs all kernel rows in one chunk");
  static_assert(WAVE_CO == 32 || WAVE_CO == 64, "wave co tile");
  static_assert(WAVE_PX % 32 == 0, "wave px tile");
  static_assert((TW & (TW - 1)) == 0, "TW must be a power of two");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// UP = the input is the VIRTUAL tensor cat(nearest_x2(in), in2) of a U-Net decoder block (smp DecoderBlock:
// F.interpolate(scale_factor=2, mode="nearest") + torch.cat): channel groups below C1 are fetched from the low-res
// map at (y/2, x/2), the rest from the skip tensor, so the concatenated tensor is never written or re-read.
template <typename T, int KH, int KW, int STRIDE, int RG, int BCO, int WCO, int WPX, int TH, int TW, int HK,
          bool UP = false>
__global__ void __launch_bounds__(64 * WCO * WPX, 2) conv_igemm_kernel(ConvArgs a) {
  static_assert(!UP || (HK >= 2 && KH == 3 && STRIDE == 1), "the two-source loader lives in the pipelined 3x3 path");
  using G = ConvGeom<KH, KW, STRIDE, RG, BCO, WCO, WPX, TH, TW, HK>;
  constexpr int EB = ElemTraits<T>::kBytes;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];
  unsigned char* sIn = smem;
  unsigned char* sW = smem + G::HALO_BYTES;

  // block -> (pixel tile, co block); blocks b and b+8 share an XCD (observed round-robin), so the
  // co blocks of one pixel tile are kept on one XCD's L2 (speed only, never correctness)
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int j = bid >> 3;
  const int pt = (j / a.ncb) * 8 + xcd;
  const int cb = j % a.ncb;
  if (pt >= a.npt) return;
  const int tx = pt % a.tiles_x;
  const int t2 = pt / a.tiles_x;
  const int ty = t2 % a.tiles_y;
  const int b = t2 / a.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy_base = oy0 * STRIDE - a.pad;
  const int ix_base = ox0 * STRIDE - a.pad;

  const int tid = threadIdx.x;
#if FFA_CONV_TRACE
  // FFA_CONV_TRACE_FIRST: first traced block (a later dispatch round shows the steady state)
  const bool trace_on = (threadIdx.x == 0 && (int)blockIdx.x >= FFA_CONV_TRACE_FIRST && (int)blockIdx.x < FFA_CONV_TRACE_FIRST + 64);
  int trace_n = 0;
#endif
  FFA_TRACE(0)
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wco = wave / WPX;
  const int wpx = wave % WPX;
  const int rho = lane & 31;
  const int half = lane >> 5;

  // per-lane LDS fragment offsets
  int aoff[G::MT];
#pragma unroll
  for (int mt = 0; mt < G::MT; ++mt) {
    // LDS rows are stored in fragment order (row mt*32 + rho of a 64-row group holds output channel
    // 16*(rho>>3) + 8*((rho>>2)&1) + 4*mt + (rho&3), the permutation is applied by ffa_pack_conv_weight), so
    // neighbouring lanes read neighbouring rows: 19 slots per row is odd -> conflict-free ds_read_b128
    aoff[mt] = (wco * G::WAVE_CO + mt * 32 + rho) * G::WP + half * 16;
  }
  int boff[G::NT];
#pragma unroll
  for (int nt = 0; nt < G::NT; ++nt) {
    const int n = wpx * G::WAVE_PX + nt * 32 + rho;
    const int py = n / TW, px = n % TW;
    boff[nt] = ((py * G::LS) * G::IW + px * G::LS) * G::PP + half * 16;
  }

  ffa_f32x16 acc[G::MT][G::NT];
#pragma unroll
  for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const unsigned char* in_b = static_cast<const unsigned char*>(a.in);
  const unsigned char* w_b = static_cast<const unsigned char*>(a.w) +
                             (size_t)cb * a.nchunks * G::NRG * (size_t)(BCO * G::TAPS * 32);
  const int total_chunks = a.nchunks * G::NRG;

  ffa_u32x4 wregA[G::NWP], wregB[G::NWP];  // weight slabs of chunk c+1 / c+2 in flight (prefetch distance 2)
  ffa_u32x4 hreg[G::NHP];

  // halo piece geometry, computed once: byte offset from the input base (32-bit: the host checks the tensor is
  // smaller than 2 GiB), -1 where the piece is padding / zero insertion / past the tile; LDS byte address
  int hoff[G::NHP], hlds[G::NHP];
  int hoff2[UP ? G::NHP : 1];  // UP: offsets into the skip tensor (hoff then addresses the low-res map)
#pragma unroll
  for (int k = 0; k < G::NHP; ++k) {
    const int i = tid + k * G::NTHR;
    const int q = i / G::HPP, hh = i % G::HPP;
    const int hy = q / G::IW, hx = q % G::IW;
    int vy = iy_base + hy * G::GSTEP;
    int vx = ix_base + hx * G::GSTEP;
    bool ok = (i < G::H_PIECES) && vy >= 0 && vx >= 0;
    if (a.dil == 2) {
      ok = ok && (((vy | vx) & 1) == 0);
      vy >>= 1;
      vx >>= 1;
    }
    ok = ok && vy < a.Hi && vx < a.Wi;
    if constexpr (UP) {
      hoff[k] = ok ? (((b * (a.Hi >> 1) + (vy >> 1)) * (a.Wi >> 1) + (vx >> 1)) * a.c1_bytes + hh * 16) : -1;
      hoff2[k] = ok ? (((b * a.Hi + vy) * a.Wi + vx) * (a.Ci * EB - a.c1_bytes) + hh * 16) : -1;
    } else {
      hoff[k] = ok ? (((b * a.Hi + vy) * a.Wi + vx) * a.Ci * EB + hh * 16) : -1;
    }
    hlds[k] = q * G::PP + hh * 16;
  }

#define FFA_LOAD_W(c_, WR)                                                                                   \
  {                                                                                                          \
    const ffa_u32x4* wsrc_ = reinterpret_cast<const ffa_u32x4*>(w_b + (size_t)(c_) * (BCO * G::TAPS * 32)); \
    _Pragma("unroll") for (int k = 0; k < G::NWP; ++k) {                                                     \
      const int i = tid + k * G::NTHR;                                                                       \
      WR[k] = wsrc_[(k + 1 < G::NWP || G::W_PIECES % G::NTHR == 0 || i < G::W_PIECES) ? i : 0];                                \
    }                                                                                                        \
  }
  /* c_ = chunk whose halo is needed: channel bytes (c_/NRG/HK)*HK*32 .., kernel rows (c_ % NRG)*RG ..
     The per-piece byte offset and validity are chunk independent (NRG == 1): they were computed once into
     hoff[] (-1 = zero fill), so a refill costs one scalar add for the base and no vector address math. */
#define FFA_LOAD_H(c_)                                                                                       \
  {                                                                                                          \
    const int cbyte_ = (((c_) / G::NRG) / HK) * (HK * 32);                                                   \
    if constexpr (G::NRG == 1) {                                                                             \
      const unsigned char* base_ = in_b + cbyte_;                                                            \
      _Pragma("unroll") for (int k = 0; k < G::NHP; ++k) {                                                   \
        ffa_u32x4 v = ffa_u32x4{0u, 0u, 0u, 0u};                                                             \
        if (hoff[k] >= 0) v = *reinterpret_cast<const ffa_u32x4*>(base_ + (unsigned)hoff[k]);                \
        hreg[k] = v;                                                                                         \
      }                                                                                                      \
    } else {                                                                                                 \
      const int iy0_ = iy_base + ((c_) % G::NRG) * RG;                                                       \
      _Pragma("unroll") for (int k = 0; k < G::NHP; ++k) {                                                   \
        const int i = tid + k * G::NTHR;                                                                     \
        const int q = i / G::HPP;                                                                            \
        const int hh = i % G::HPP;                                                                           \
        const int hy = q / G::IW, hx = q % G::IW;                                                            \
        int vy = iy0_ + hy * G::GSTEP;                                                                       \
        int vx = ix_base + hx * G::GSTEP;                                                                    \
        bool ok = (i < G::H_PIECES) && vy >= 0 && vx >= 0;                                                   \
        if (a.dil == 2) {                                                                                    \
          ok = ok && (((vy | vx) & 1) == 0);                                                                 \
          vy >>= 1;                                                                                          \
          vx >>= 1;                                                                                          \
        }                                                                                                    \
        ok = ok && vy < a.Hi && vx < a.Wi;                                                                   \
        ffa_u32x4 v = ffa_u32x4{0u, 0u, 0u, 0u};                                                             \
        if (ok) {                                                                                            \
          const size_t off =                                                                                 \
              ((size_t)(b * a.Hi + vy) * a.Wi + vx) * (size_t)(a.Ci * EB) + cbyte_ + hh * 16;                \
          v = *reinterpret_cast<const ffa_u32x4*>(in_b + off);                                               \
        }                                                                                                    \
        hreg[k] = v;                                                                                         \
      }                                                                                                      \
    }                                                                                                        \
  }
#define FFA_STORE_W(WR)                                                                    \
  {                                                                                        \
    _Pragma("unroll") for (int k = 0; k < G::NWP; ++k) {                                   \
      const int i = tid + k * G::NTHR;                                                     \
      if (k + 1 < G::NWP || G::W_PIECES % G::NTHR == 0 || i < G::W_PIECES) {                                 \
        const int row = i / (G::TAPS * 2), col = i % (G::TAPS * 2);                        \
        *reinterpret_cast<ffa_u32x4*>(sW + row * G::WP + col * 16) = WR[k];                \
      }                                                                                    \
    }                                                                                      \
  }
#define FFA_STORE_H()                                                                      \
  {                                                                                        \
    _Pragma("unroll") for (int k = 0; k < G::NHP; ++k) {                                   \
      const int i = tid + k * G::NTHR;                                                     \
      if (k + 1 < G::NHP || G::H_PIECES % G::NTHR == 0 || i < G::H_PIECES)                                   \
        *reinterpret_cast<ffa_u32x4*>(sIn + hlds[k]) = hreg[k];                            \
    }                                                                                      \
  }

  // one chunk of matrix work from the LDS images; `sub` = byte offset of the chunk's k-step inside a halo pixel
  // `issue(tap)` = the global loads this chunk sends on behalf of later chunks, spread over the taps so that the
  // texture path (64 B/clk per CU: a block's 29 KB per chunk is ~450 cycles of it) drains under the matrix work
  // instead of in a burst in front of it; `nv(tap)` = how many VMEM instructions that is (for the pinned order)
  auto compute = [&](int sub, auto issue, auto nv) __attribute__((always_inline)) {
    // fragments are double-buffered in registers: the ds_reads of tap t+1 are issued before the MFMAs of
    // tap t, so their LDS latency hides under the matrix pipe instead of stalling every tap
    ffa_u32x4 af[2][G::MT], bf[2][G::NT];
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) af[0][mt] = *reinterpret_cast<const ffa_u32x4*>(sW + aoff[mt]);
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) bf[0][nt] = *reinterpret_cast<const ffa_u32x4*>(sIn + boff[nt] + sub);
#pragma unroll
    for (int tap = 0; tap < G::TAPS; ++tap) {
      const int cur = tap & 1, nxt = cur ^ 1;
      __builtin_amdgcn_sched_barrier(0);  // one scheduling region per tap: its MFMAs + the reads of the next tap
      issue(tap);
      if (tap + 1 < G::TAPS) {
        const int r1 = (tap + 1) / KW, s1 = (tap + 1) % KW;
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
          af[nxt][mt] = *reinterpret_cast<const ffa_u32x4*>(sW + aoff[mt] + (tap + 1) * 32);
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt)
          bf[nxt][nt] = *reinterpret_cast<const ffa_u32x4*>(sIn + boff[nt] + sub + (r1 * G::IW + s1) * G::PP);
      }
#pragma unroll
      for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt) Mma<T>::run(af[cur][mt], bf[cur][nt], acc[mt][nt]);
      // pin the interleave (hipcc otherwise sinks the reads back to just before their first use):
      // one MFMA group, one ds_read, ... so every read has a full tap of matrix work to land under
      {
        constexpr int NM = G::MT * G::NT;
        const int NR = (tap + 1 < G::TAPS) ? G::MT + G::NT : 0;
        const int NV = nv(tap);
        constexpr int PER = (sizeof(T) == 2) ? 1 : 4;  // MFMAs per Mma<T>::run
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
          if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = NM; i < G::MT + G::NT; ++i)
          if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = NM; i < NM + 4; ++i)
          if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // Halo refills are issued HD chunks before they are stored (deep staging: three chunks of matrix work to land
  // under), weight slabs two chunks ahead (two register sets, loop unrolled by two)
  constexpr int HD = (HK >= 4 && G::NRG == 1) ? 3 : 1;
#define FFA_CHUNK(c_, WCUR, WNXT)                                                                            \
  {                                                                                                          \
    const int cq_ = (c_);                                                                                    \
    if (cq_ + 2 < total_chunks) FFA_LOAD_W(cq_ + 2, WNXT)                                                    \
    if (cq_ + HD < total_chunks && (G::NRG > 1 || ((cq_ + HD) % HK) == 0)) FFA_LOAD_H(cq_ + HD)              \
    FFA_TRACE(2)                                                                                             \
    compute((G::NRG > 1) ? 0 : (cq_ % HK) * 32, no_issue, no_nv);                                            \
    FFA_TRACE(3)                                                                                             \
    __syncthreads();                                                                                         \
    FFA_TRACE(4)                                                                                             \
    if (cq_ + 1 < total_chunks) {                                                                            \
      FFA_STORE_W(WCUR)                                                                                      \
      if (G::NRG > 1 || ((cq_ + 1) % HK) == 0) FFA_STORE_H()                                                 \
      FFA_TRACE(5)                                                                                           \
      __syncthreads();                                                                                       \
      FFA_TRACE(6)                                                                                           \
    }                                                                                                        \
  }

  auto no_issue = [](int) __attribute__((always_inline)) {};
  auto no_nv = [](int) __attribute__((always_inline)) { return 0; };

  FFA_LOAD_W(0, wregA)
  FFA_LOAD_H(0)
  FFA_STORE_W(wregA)
  FFA_STORE_H()
  __syncthreads();
  FFA_TRACE(1)
  if (total_chunks > 1) FFA_LOAD_W(1, wregA)
  if constexpr (HK >= 2 && G::NRG == 1) {
    // Pipelined path (3x3 stride 1, whole groups of HK chunks per halo fill): every global load is issued from
    // inside a tap's scheduling region.  Loads are unconditional -- a chunk index past the end is clamped (the
    // data is never stored), a padding piece reads offset 0 and is zeroed when it is stored -- so the regions hold
    // no branches and the pinned MFMA / ds_read / global_load order survives.
    constexpr int HJ = (HK >= 4) ? 1 : 0;  // chunk of the group during which the next group's halo is requested
    const int last_group = total_chunks - HK;
    for (int c0 = 0; c0 < total_chunks; c0 += HK) {
      const int cn = (c0 + HK <= last_group) ? c0 + HK : last_group;  // next group (clamped)
      const int gbn = (cn / HK) * (HK * 32);  // its first channel byte within a (virtual) input pixel
      const bool nextA = !UP || gbn < a.c1_bytes;
      const unsigned char* hbase = nextA ? in_b + gbn : static_cast<const unsigned char*>(a.in2) + (gbn - a.c1_bytes);
#pragma unroll
      for (int j = 0; j < HK; ++j) {
        const int c = c0 + j;
        const int cw = (c + 2 < total_chunks) ? c + 2 : total_chunks - 1;
        const ffa_u32x4* wsrc = reinterpret_cast<const ffa_u32x4*>(w_b + (size_t)cw * (BCO * G::TAPS * 32));
        auto issue = [&](int tap) __attribute__((always_inline)) {
#pragma unroll
          for (int k = 0; k < G::NWP; ++k) {
            if (k % G::TAPS != tap) continue;
            const int i = tid + k * G::NTHR;
            const ffa_u32x4 v = wsrc[(k + 1 < G::NWP || G::W_PIECES % G::NTHR == 0 || i < G::W_PIECES) ? i : 0];
            if (j & 1) wregA[k] = v;
            else wregB[k] = v;
          }
          if (j == HJ) {
#pragma unroll
            for (int k = 0; k < G::NHP; ++k) {
              if (k % G::TAPS != tap) continue;
              int off = hoff[k];
              if constexpr (UP) off = nextA ? hoff[k] : hoff2[k];
              hreg[k] = *reinterpret_cast<const ffa_u32x4*>(hbase + (unsigned)(off >= 0 ? off : 0));
            }
          }
        };
        auto nv = [&](int tap) __attribute__((always_inline)) {
          int n = 0;
#pragma unroll
          for (int k = 0; k < G::NWP; ++k) n += (k % G::TAPS == tap) ? 1 : 0;
          if (j == HJ) {
#pragma unroll
            for (int k = 0; k < G::NHP; ++k) n += (k % G::TAPS == tap) ? 1 : 0;
          }
          return n;
        };
        FFA_TRACE(2)
        compute(j * 32, issue, nv);
        FFA_TRACE(3)
        __syncthreads();
        FFA_TRACE(4)
        if (c + 1 < total_chunks) {
          if (j & 1) FFA_STORE_W(wregB)
          else FFA_STORE_W(wregA)
          if (j == HK - 1) {
#pragma unroll
            for (int k = 0; k < G::NHP; ++k) {
              const int i = tid + k * G::NTHR;
              if (k + 1 < G::NHP || G::H_PIECES % G::NTHR == 0 || i < G::H_PIECES)
                *reinterpret_cast<ffa_u32x4*>(sIn + hlds[k]) = (hoff[k] >= 0) ? hreg[k] : ffa_u32x4{0u, 0u, 0u, 0u};
            }
          }
          FFA_TRACE(5)
          __syncthreads();
          FFA_TRACE(6)
        }
      }
    }
  } else {
    for (int c = 0; c < total_chunks; c += 2) {
      FFA_CHUNK(c, wregA, wregB)
      if (c + 1 < total_chunks) FFA_CHUNK(c + 1, wregB, wregA)
    }
  }

#undef FFA_CHUNK
#undef FFA_LOAD_W
#undef FFA_LOAD_H
#undef FFA_STORE_W
#undef FFA_STORE_H

  // epilogue: lane (rho, half) owns pixel n = wave px base + nt*32 + rho and, per g, a run of
  // consecutive channels (8 for MT==2, 4 for MT==1)
  T* out = static_cast<T*>(a.out);
  const T* res = static_cast<const T*>(a.res);
  const int co_wave = cb * BCO + wco * G::WAVE_CO;
  int opitch = a.Co, cshift = 0;  // pixel pitch of the tensor written and the channel it starts at
  if (a.c1_out > 0) {
    // Split epilogue = backward of "nearest x2 upsample + concat" fused into the dgrad of a decoder block's first
    // conv: the gradient of the virtual input cat(up(lo), skip) never exists as a tensor.  Channel blocks below
    // c1_out are summed over their 2x2 pixel quads (the adjoint of nearest x2) and written at half resolution,
    // the rest goes to the skip-gradient tensor with its own pitch.
    if (cb * BCO >= a.c1_out) {
      out = static_cast<T*>(a.out2);
      opitch = a.Co - a.c1_out;
      cshift = a.c1_out;
    } else {
      static_assert(G::WAVE_PX == 64, "quad pooling assumes two 32-pixel fragments per wave");
      constexpr int NV = (G::MT == 2) ? 8 : 4;
      const int Hl = a.Ho >> 1, Wl = a.Wo >> 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = (G::MT == 2) ? co_wave + 16 * g + 8 * half : co_wave + 8 * g + 4 * half;
        float q[G::NT][NV];
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt) {
          const int n = wpx * G::WAVE_PX + nt * 32 + rho;
          const bool ok = (oy0 + n / TW) < a.Ho && (ox0 + n % TW) < a.Wo;
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const float v = (G::MT == 2) ? acc[i >> 2][nt][4 * g + (i & 3)] : acc[0][nt][4 * g + i];
            q[nt][i] = ok ? v : 0.f;
          }
        }
        if (TW == 32) {  // fragment nt = tile row 2*wpx + nt, lane = column: vertical partner in registers
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            float t = q[0][i] + q[1][i];
            t += __shfl_xor(t, 1, 64);
            q[0][i] = t;
          }
          const int ly = (oy0 >> 1) + wpx, lx = (ox0 >> 1) + (rho >> 1);
          if ((rho & 1) == 0 && ly < Hl && lx < Wl && c0 < a.c1_out) {
            T* dst = out + ((size_t)(b * Hl + ly) * Wl + lx) * (size_t)a.c1_out + c0;
            if (G::MT == 2) {
              float v8[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) v8[i] = q[0][i & 7];
              ffa_store8<T>(dst, v8);
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) ffa_store_elem<T>(dst + i, q[0][i]);
            }
          }
        } else {  // 16x16 tiles: fragment nt = rows 4*wpx + 2*nt + (rho >> 4), lane & 15 = column
#pragma unroll
          for (int nt = 0; nt < G::NT; ++nt) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
              float t = q[nt][i] + __shfl_xor(q[nt][i], 16, 64);
              t += __shfl_xor(t, 1, 64);
              q[nt][i] = t;
            }
            const int ly = (oy0 >> 1) + 2 * wpx + nt, lx = (ox0 >> 1) + ((rho & 15) >> 1);
            if ((rho & 17) == 0 && ly < Hl && lx < Wl && c0 < a.c1_out) {
              T* dst = out + ((size_t)(b * Hl + ly) * Wl + lx) * (size_t)a.c1_out + c0;
              if (G::MT == 2) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = q[nt][i & 7];
                ffa_store8<T>(dst, v8);
              } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) ffa_store_elem<T>(dst + i, q[nt][i]);
              }
            }
          }
        }
      }
      return;
    }
  }
  // BatchNorm batch statistics of the tensor being written, from the registers that hold it (a.stats != null):
  // per lane NCH channels x (sum, sum of squares) over its pixels, of the values AS STORED (bf16-rounded)
  constexpr int NCH = (G::MT == 2) ? 32 : 16;
  float st[2 * NCH];
  const bool want_stats = a.stats != nullptr;
#pragma unroll
  for (int i = 0; i < 2 * NCH; ++i) st[i] = 0.f;
#pragma unroll
  for (int nt = 0; nt < G::NT; ++nt) {
    const int n = wpx * G::WAVE_PX + nt * 32 + rho;
    const int oy = oy0 + n / TW, ox = ox0 + n % TW;
    if (oy >= a.Ho || ox >= a.Wo) continue;
    const long long pix = ((long long)(b * a.Ho + oy) * a.Wo + ox) * (long long)opitch - cshift;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (G::MT == 2) {
        const int c0 = co_wave + 16 * g + 8 * half;
        if (c0 >= a.Co) continue;
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[0][nt][4 * g + i];
          v[4 + i] = acc[G::MT - 1][nt][4 * g + i];
        }
        if (a.bias) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += a.bias[c0 + i];
        }
        if (res) {
          float rv[8];
          ffa_load8<T>(res + pix + c0, rv);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += rv[i];
        }
        if (a.relu) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        ffa_store8<T>(out + pix + c0, v);
        if (want_stats) {
          if (a.bnx) {
            float xv[8], sc[8], sh[8];
            ffa_load8<T>(static_cast<const T*>(a.bnx) + pix + c0, xv);
            ffa_load8<float>(a.bn_sc + c0, sc);
            ffa_load8<float>(a.bn_sh + c0, sh);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float r = (EB == 2) ? ffa_bf16_bits_to_f32(ffa_f32_to_bf16_bits(v[i])) : v[i];
              const float gg = (xv[i] * sc[i] + sh[i]) > 0.f ? r : 0.f;  // same fma as the forward / apply kernels
              st[g * 8 + i] += gg;
              st[NCH + g * 8 + i] = __builtin_fmaf(gg, xv[i], st[NCH + g * 8 + i]);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float r = (EB == 2) ? ffa_bf16_bits_to_f32(ffa_f32_to_bf16_bits(v[i])) : v[i];
              st[g * 8 + i] += r;
              st[NCH + g * 8 + i] = __builtin_fmaf(r, r, st[NCH + g * 8 + i]);  // explicit: both conv kernels must round alike
            }
          }
        }
      } else {
        const int c0 = co_wave + 8 * g + 4 * half;
        if (c0 >= a.Co) continue;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[0][nt][4 * g + i];
        if (a.bias) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += a.bias[c0 + i];
        }
        if (res) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += ffa_load_elem<T>(res + pix + c0 + i);
        }
        if (a.relu) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (EB == 2) {
          uint2 u;
          u.x = ffa_pack_bf16x2(v[0], v[1]);
          u.y = ffa_pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(out + pix + c0) = u;
        } else {
          *reinterpret_cast<float4*>(out + pix + c0) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (want_stats) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float r = (EB == 2) ? ffa_bf16_bits_to_f32(ffa_f32_to_bf16_bits(v[i])) : v[i];
            if (a.bnx) {
              const float xi = ffa_load_elem<T>(static_cast<const T*>(a.bnx) + pix + c0 + i);
              const float gg = (xi * a.bn_sc[c0 + i] + a.bn_sh[c0 + i]) > 0.f ? r : 0.f;
              st[g * 4 + i] += gg;
              st[NCH + g * 4 + i] = __builtin_fmaf(gg, xi, st[NCH + g * 4 + i]);
            } else {
              st[g * 4 + i] += r;
              st[NCH + g * 4 + i] = __builtin_fmaf(r, r, st[NCH + g * 4 + i]);
            }
          }
        }
      }
    }
  }
  if (want_stats) {
    static_assert(WCO == 1, "the statistics epilogue assumes one co wave group per block");
    // Transposing reduction over the 32 lanes of a half-wave: at every step a lane keeps one half of its values and
    // adds the partner's copy of that half -- 2*NCH - 2 shuffles instead of 5 per value.  Afterwards lane rho holds
    // two entries of [sums | sums of squares]: entry index = (bits of rho, high to low) * 2 + j.
#pragma unroll
    for (int bit = 4; bit >= 0; --bit) {
      const int n = (2 * NCH) >> (4 - bit);  // values a lane still holds (compile-time: the loop is unrolled)
      if (n > 2) {
        const bool up = (rho >> bit) & 1;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          if (j < n / 2) {
            // (the two reads go through an opaque asm: hipcc otherwise rewrites "up ? st[a] : st[b]" into
            // st[up ? a : b], a dynamically indexed register array = a 32-way compare/select chain per value)
            float lo = st[j], hi = st[j + n / 2];
            asm volatile("" : "+v"(lo), "+v"(hi));
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            st[j] = keep + __shfl_xor(send, 1 << bit, 64);
          }
        }
      } else {  // NCH == 16: the last lane bit is a plain butterfly, both lanes of a pair end up with the totals
        st[0] += __shfl_xor(st[0], 1 << bit, 64);
        st[1] += __shfl_xor(st[1], 1 << bit, 64);
      }
    }
    // the pixel waves of the block add up through LDS (free: a barrier followed the last fragment reads)
    float* red = reinterpret_cast<float*>(smem);
    red[(wpx * 64 + lane) * 2 + 0] = st[0];
    red[(wpx * 64 + lane) * 2 + 1] = st[1];
    __syncthreads();
    constexpr int NOUT = (G::MT == 2) ? 128 : 64;
    if (tid < NOUT) {
      const int j = tid & 1;
      const int ln = (G::MT == 2) ? (tid >> 1) : ((tid >> 1) * 2);  // MT == 1: even lanes carry the totals
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WPX; ++w) t += red[(w * 64 + ln) * 2 + j];
      const int r5 = ln & 31, hf = ln >> 5;
      const int which = r5 >> 4;
      int c;
      if (G::MT == 2) {
        const int L = (r5 & 15) * 2 + j;
        c = cb * BCO + 16 * (L >> 3) + 8 * hf + (L & 7);
      } else {
        const int L = ((r5 >> 1) & 7) * 2 + j;
        c = cb * BCO + 8 * (L >> 2) + 4 * hf + (L & 3);
      }
      if (c < a.Co) a.stats[((size_t)pt * 2 + which) * a.Co + c] = t;
    }
  }
  FFA_TRACE(7)
}

// ------------------------------------------------------------------------------------------------
// Persistent variant of the pipelined 3x3 stride-1 path (the default for plain input, dil 1, no split / bnbwd
// epilogue; FFA_CONV_PERSIST=0 disables).  A block walks the virtual block list with stride gridDim.x and carries the software pipeline ACROSS
// tiles: the next tile's halo and first two weight slabs are requested from inside the last chunk group of the
// current tile and stored to LDS before the current tile's epilogue, so the prologue burst (35-40 % of a block's
// life on the short-K layers, tools/conv_trace.py) and the epilogue's stores overlap matrix work instead of
// standing alone.  The statistics epilogue reduces through its own 2 KB of LDS (the halo image already belongs to
// the next tile by then).
template <typename T, int BCO, int TH, int TW, int HK>
__global__ void __launch_bounds__(256, 2) conv3x3_persist_kernel(ConvArgs a) {
  constexpr int KW = 3;
  using G = ConvGeom<3, 3, 1, 3, BCO, 1, 4, TH, TW, HK>;
  constexpr int EB = ElemTraits<T>::kBytes;
  constexpr int WPX = 4;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES + 2048];
  unsigned char* sIn = smem;
  unsigned char* sW = smem + G::HALO_BYTES;
  float* red = reinterpret_cast<float*>(smem + G::LDS_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wpx = tid >> 6;
  const int rho = lane & 31;
  const int half = lane >> 5;
  const int total_vb = ((a.npt + 7) / 8) * 8 * a.ncb;
  const int total_chunks = a.nchunks;
  const size_t slab = (size_t)(BCO * G::TAPS * 32);

  int aoff[G::MT];
#pragma unroll
  for (int mt = 0; mt < G::MT; ++mt) aoff[mt] = (mt * 32 + rho) * G::WP + half * 16;
  int boff[G::NT];
#pragma unroll
  for (int nt = 0; nt < G::NT; ++nt) {
    const int n = wpx * G::WAVE_PX + nt * 32 + rho;
    boff[nt] = ((n / TW) * G::IW + (n % TW)) * G::PP + half * 16;
  }
  // LDS address of halo piece k of this thread: piece tid + k * NTHR is pixel tid / HPP + k * (NTHR / HPP)
  static_assert(G::NTHR % G::HPP == 0, "halo pieces of one thread share the 16-byte slot");
  const int hlds0 = (tid / G::HPP) * G::PP + (tid % G::HPP) * 16;
  constexpr int HLDS_STEP = (G::NTHR / G::HPP) * G::PP;

  struct TileId {
    int pt, cb, b, oy0, ox0;
  };
  // virtual block -> tile (same XCD-aware order as conv_igemm_kernel); pt >= npt marks the padding of the last group
  auto decode = [&](int vb) {
    TileId t;
    const int j = vb >> 3;
    t.pt = (j / a.ncb) * 8 + (vb & 7);
    t.cb = j % a.ncb;
    const int tx = t.pt % a.tiles_x;
    const int t2 = t.pt / a.tiles_x;
    t.b = t2 / a.tiles_y;
    t.oy0 = (t2 % a.tiles_y) * TH;
    t.ox0 = tx * TW;
    return t;
  };
  auto next_valid = [&](int vb) {
    while (vb < total_vb && decode(vb).pt >= a.npt) vb += gridDim.x;
    return vb;
  };
  auto halo_offsets = [&](const TileId& t, int* ho) {
#pragma unroll
    for (int k = 0; k < G::NHP; ++k) {
      const int i = tid + k * G::NTHR;
      const int q = i / G::HPP, hh = i % G::HPP;
      const int vy = t.oy0 - a.pad + q / G::IW;
      const int vx = t.ox0 - a.pad + q % G::IW;
      const bool ok = (i < G::H_PIECES) && vy >= 0 && vx >= 0 && vy < a.Hi && vx < a.Wi;
      ho[k] = ok ? (((t.b * a.Hi + vy) * a.Wi + vx) * a.Ci * EB + hh * 16) : -1;
    }
  };

  int vb = next_valid(blockIdx.x);
  if (vb >= total_vb) return;
  TileId cur = decode(vb);
  int hoff[G::NHP];  // piece offsets of the NEXT halo fill (this tile's next group, then the next tile's first)
  halo_offsets(cur, hoff);
  const unsigned char* in_b = static_cast<const unsigned char*>(a.in);
  const unsigned char* w_all = static_cast<const unsigned char*>(a.w);
  const unsigned char* w_b = w_all + (size_t)cur.cb * total_chunks * slab;

  ffa_u32x4 wregA[G::NWP], wregB[G::NWP];
  ffa_u32x4 hreg[G::NHP];
  ffa_f32x16 acc[G::MT][G::NT];

#define FFA_P_STORE_W(WR)                                                                  \
  {                                                                                        \
    _Pragma("unroll") for (int k = 0; k < G::NWP; ++k) {                                   \
      const int i = tid + k * G::NTHR;                                                     \
      if (k + 1 < G::NWP || G::W_PIECES % G::NTHR == 0 || i < G::W_PIECES) {               \
        const int row = i / (G::TAPS * 2), col = i % (G::TAPS * 2);                        \
        *reinterpret_cast<ffa_u32x4*>(sW + row * G::WP + col * 16) = WR[k];                \
      }                                                                                    \
    }                                                                                      \
  }
#define FFA_P_STORE_H(HO)                                                                  \
  {                                                                                        \
    _Pragma("unroll") for (int k = 0; k < G::NHP; ++k) {                                   \
      const int i = tid + k * G::NTHR;                                                     \
      if (k + 1 < G::NHP || G::H_PIECES % G::NTHR == 0 || i < G::H_PIECES)                 \
        *reinterpret_cast<ffa_u32x4*>(sIn + hlds0 + k * HLDS_STEP) = (HO[k] >= 0) ? hreg[k] : ffa_u32x4{0u, 0u, 0u, 0u}; \
    }                                                                                      \
  }

  // prologue of the block's first tile (the only exposed one)
  {
    const ffa_u32x4* wsrc = reinterpret_cast<const ffa_u32x4*>(w_b);
#pragma unroll
    for (int k = 0; k < G::NWP; ++k) {
      const int i = tid + k * G::NTHR;
      wregA[k] = wsrc[(k + 1 < G::NWP || G::W_PIECES % G::NTHR == 0 || i < G::W_PIECES) ? i : 0];
    }
#pragma unroll
    for (int k = 0; k < G::NHP; ++k) hreg[k] = *reinterpret_cast<const ffa_u32x4*>(in_b + (unsigned)(hoff[k] >= 0 ? hoff[k] : 0));
    FFA_P_STORE_W(wregA)
    FFA_P_STORE_H(hoff)
    __syncthreads();
    const ffa_u32x4* wsrc1 = reinterpret_cast<const ffa_u32x4*>(w_b + slab);
#pragma unroll
    for (int k = 0; k < G::NWP; ++k) {
      const int i = tid + k * G::NTHR;
      wregA[k] = wsrc1[(k + 1 < G::NWP || G::W_PIECES % G::NTHR == 0 || i < G::W_PIECES) ? i : 0];
    }
  }

  constexpr int HJ = (HK >= 4) ? 1 : 0;
  while (true) {
    const int nvb = next_valid(vb + gridDim.x);
    const bool has_next = nvb < total_vb;
    const TileId nxt = decode(has_next ? nvb : vb);
    const unsigned char* w_n = w_all + (size_t)nxt.cb * total_chunks * slab;

#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    for (int c0 = 0; c0 < total_chunks; c0 += HK) {
      const bool last_group = (c0 + HK >= total_chunks);
      const bool cross = last_group && has_next;  // the halo requested during this group belongs to the next tile
      // next halo fill: the following channel group of this tile, or group 0 of the next tile; a block without a
      // next tile re-reads its last group (never stored)
      const unsigned char* hbase = in_b + (last_group ? (has_next ? 0 : c0 * 32) : (c0 + HK) * 32);
      // from here on this tile needs no further halo of its own: hoff switches to the next tile
      if (cross) halo_offsets(nxt, hoff);
#pragma unroll
      for (int j = 0; j < HK; ++j) {
        const int c = c0 + j;
        // weight slab two chunks ahead; past the end of this tile it is chunk 0 / 1 of the next tile's co block
        const unsigned char* wp = (c + 2 < total_chunks) ? w_b + (size_t)(c + 2) * slab
                                  : (has_next ? w_n + (size_t)(c + 2 - total_chunks) * slab
                                              : w_b + (size_t)(total_chunks - 1) * slab);
        const ffa_u32x4* wsrc = reinterpret_cast<const ffa_u32x4*>(wp);
        auto issue = [&](int tap) __attribute__((always_inline)) {
#pragma unroll
          for (int k = 0; k < G::NWP; ++k) {
            if (k % G::TAPS != tap) continue;
            const int i = tid + k * G::NTHR;
            const ffa_u32x4 v = wsrc[(k + 1 < G::NWP || G::W_PIECES % G::NTHR == 0 || i < G::W_PIECES) ? i : 0];
            if (j & 1) wregA[k] = v;
            else wregB[k] = v;
          }
          if (j == HJ) {
#pragma unroll
            for (int k = 0; k < G::NHP; ++k) {
              if (k % G::TAPS != tap) continue;
              hreg[k] = *reinterpret_cast<const ffa_u32x4*>(hbase + (unsigned)(hoff[k] >= 0 ? hoff[k] : 0));
            }
          }
        };
        // ---- one chunk of matrix work (same pinned MFMA / ds_read / global_load interleave as conv_igemm_kernel)
        {
          const int sub = j * 32;
          ffa_u32x4 af[2][G::MT], bf[2][G::NT];
#pragma unroll
          for (int mt = 0; mt < G::MT; ++mt) af[0][mt] = *reinterpret_cast<const ffa_u32x4*>(sW + aoff[mt]);
#pragma unroll
          for (int nt = 0; nt < G::NT; ++nt) bf[0][nt] = *reinterpret_cast<const ffa_u32x4*>(sIn + boff[nt] + sub);
#pragma unroll
          for (int tap = 0; tap < G::TAPS; ++tap) {
            const int cu = tap & 1, nx = cu ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            issue(tap);
            if (tap + 1 < G::TAPS) {
              const int r1 = (tap + 1) / KW, s1 = (tap + 1) % KW;
#pragma unroll
              for (int mt = 0; mt < G::MT; ++mt)
                af[nx][mt] = *reinterpret_cast<const ffa_u32x4*>(sW + aoff[mt] + (tap + 1) * 32);
#pragma unroll
              for (int nt = 0; nt < G::NT; ++nt)
                bf[nx][nt] = *reinterpret_cast<const ffa_u32x4*>(sIn + boff[nt] + sub + (r1 * G::IW + s1) * G::PP);
            }
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < G::NT; ++nt) Mma<T>::run(af[cu][mt], bf[cu][nt], acc[mt][nt]);
            {
              constexpr int NM = G::MT * G::NT;
              const int NR = (tap + 1 < G::TAPS) ? G::MT + G::NT : 0;
              int NV = 0;
#pragma unroll
              for (int k = 0; k < G::NWP; ++k) NV += (k % G::TAPS == tap) ? 1 : 0;
              if (j == HJ) {
#pragma unroll
                for (int k = 0; k < G::NHP; ++k) NV += (k % G::TAPS == tap) ? 1 : 0;
              }
              constexpr int PER = (sizeof(T) == 2) ? 1 : 4;
#pragma unroll
              for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
              }
#pragma unroll
              for (int i = NM; i < G::MT + G::NT; ++i)
                if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
              for (int i = NM; i < NM + 4; ++i)
                if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (c + 1 < total_chunks || has_next) {
          if (j & 1) FFA_P_STORE_W(wregB)
          else FFA_P_STORE_W(wregA)
          if (j == HK - 1) FFA_P_STORE_H(hoff)
          __syncthreads();
        }
      }
    }

    // ---- epilogue of `cur` (its LDS images already hold the next tile) ----
    {
      T* out = static_cast<T*>(a.out);
      const T* res = static_cast<const T*>(a.res);
      const int co_wave = cur.cb * BCO;
      constexpr int NCH = (G::MT == 2) ? 32 : 16;
      float st[2 * NCH];
      const bool want_stats = a.stats != nullptr;
#pragma unroll
      for (int i = 0; i < 2 * NCH; ++i) st[i] = 0.f;
#pragma unroll
      for (int nt = 0; nt < G::NT; ++nt) {
        const int n = wpx * G::WAVE_PX + nt * 32 + rho;
        const int oy = cur.oy0 + n / TW, ox = cur.ox0 + n % TW;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const long long pix = ((long long)(cur.b * a.Ho + oy) * a.Wo + ox) * (long long)a.Co;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (G::MT == 2) {
            const int c0 = co_wave + 16 * g + 8 * half;
            if (c0 >= a.Co) continue;
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[i] = acc[0][nt][4 * g + i];
              v[4 + i] = acc[G::MT - 1][nt][4 * g + i];
            }
            if (a.bias) {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] += a.bias[c0 + i];
            }
            if (res) {
              float rv[8];
              ffa_load8<T>(res + pix + c0, rv);
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] += rv[i];
            }
            if (a.relu) {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            ffa_store8<T>(out + pix + c0, v);
            if (want_stats) {
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float r = (EB == 2) ? ffa_bf16_bits_to_f32(ffa_f32_to_bf16_bits(v[i])) : v[i];
                st[g * 8 + i] += r;
                st[NCH + g * 8 + i] = __builtin_fmaf(r, r, st[NCH + g * 8 + i]);  // explicit: both conv kernels must round alike
              }
            }
          } else {
            const int c0 = co_wave + 8 * g + 4 * half;
            if (c0 >= a.Co) continue;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[0][nt][4 * g + i];
            if (a.bias) {
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] += a.bias[c0 + i];
            }
            if (res) {
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] += ffa_load_elem<T>(res + pix + c0 + i);
            }
            if (a.relu) {
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            if (EB == 2) {
              uint2 u;
              u.x = ffa_pack_bf16x2(v[0], v[1]);
              u.y = ffa_pack_bf16x2(v[2], v[3]);
              *reinterpret_cast<uint2*>(out + pix + c0) = u;
            } else {
              *reinterpret_cast<float4*>(out + pix + c0) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (want_stats) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float r = (EB == 2) ? ffa_bf16_bits_to_f32(ffa_f32_to_bf16_bits(v[i])) : v[i];
                st[g * 4 + i] += r;
                st[NCH + g * 4 + i] = __builtin_fmaf(r, r, st[NCH + g * 4 + i]);
              }
            }
          }
        }
      }
      if (want_stats) {  // same transposing reduction and channel map as conv_igemm_kernel
#pragma unroll
        for (int bit = 4; bit >= 0; --bit) {
          const int n = (2 * NCH) >> (4 - bit);
          if (n > 2) {
            const bool up = (rho >> bit) & 1;
#pragma unroll
            for (int jj = 0; jj < NCH; ++jj) {
              if (jj < n / 2) {
                float lo = st[jj], hi = st[jj + n / 2];
                asm volatile("" : "+v"(lo), "+v"(hi));
                const float keep = up ? hi : lo;
                const float send = up ? lo : hi;
                st[jj] = keep + __shfl_xor(send, 1 << bit, 64);
              }
            }
          } else {
            st[0] += __shfl_xor(st[0], 1 << bit, 64);
            st[1] += __shfl_xor(st[1], 1 << bit, 64);
          }
        }
        red[(wpx * 64 + lane) * 2 + 0] = st[0];
        red[(wpx * 64 + lane) * 2 + 1] = st[1];
        __syncthreads();
        constexpr int NOUT = (G::MT == 2) ? 128 : 64;
        if (tid < NOUT) {
          const int jj = tid & 1;
          const int ln = (G::MT == 2) ? (tid >> 1) : ((tid >> 1) * 2);
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < WPX; ++w) t += red[(w * 64 + ln) * 2 + jj];
          const int r5 = ln & 31, hf = ln >> 5;
          const int which = r5 >> 4;
          int c;
          if (G::MT == 2) {
            const int L = (r5 & 15) * 2 + jj;
            c = cur.cb * BCO + 16 * (L >> 3) + 8 * hf + (L & 7);
          } else {
            const int L = ((r5 >> 1) & 7) * 2 + jj;
            c = cur.cb * BCO + 8 * (L >> 2) + 4 * hf + (L & 3);
          }
          if (c < a.Co) a.stats[((size_t)cur.pt * 2 + which) * a.Co + c] = t;
        }
        __syncthreads();  // red is reused by the next tile
      }
    }
    if (!has_next) break;
    vb = nvb;
    cur = nxt;
    w_b = w_n;
  }
#undef FFA_P_STORE_W
#undef FFA_P_STORE_H
}

// ------------------------------------------------------------------------------------------------
// host side: configuration choice + launch

struct ConvPlan {
  int bco;     // output channels per block (= row count of one packed weight slab)
  int rg;      // kernel rows per chunk
  int th, tw;  // output pixel tile
};

static int conv_rg(int kh) { return kh == 7 ? 1 : kh; }

// Preferred block height in output channels for a layer with `cout` real output channels.  The
// caller packs the weights with this value and hands the same value back to ffa_conv2d.
static int conv_pref_bco(int kh, int kw, int stride, int cout) {
  // 64 rows + deep (full 128-byte line) halo staging beats 128 rows + 32-byte staging, see ConvGeom
  if (kh == 3 && kw == 3 && stride == 1) return cout > 32 ? 64 : 32;
  return cout > 32 ? 64 : 32;
}

static bool conv_supported(int kh, int kw, int stride, int bco) {
  const bool shape = (kh == 3 && kw == 3 && (stride == 1 || stride == 2)) ||
                     (kh == 1 && kw == 1 && (stride == 1 || stride == 2)) || (kh == 7 && kw == 7 && stride == 2);
  if (!shape) return false;
  return bco == 64 || bco == 32;
}

// Grid of the persistent 3x3 stride-1 kernel for a launch that would otherwise take `grid` one-tile blocks, or 0 for
// the one-tile kernel.  On by default for every eligible launch (same-box A/B: 15.79 -> 15.50 ms per training step,
// -4...-9 % on the 64-channel 128^2 / 256^2 layers whose grids need several dispatch rounds; a grid that fits one
// round simply runs one tile per block).  FFA_CONV_PERSIST=0 restores conv_igemm_kernel everywhere;
// FFA_CONV_PERSIST_MIN / FFA_CONV_PERSIST_GRID move the threshold / the grid (the tests force several tiles per
// block on small tensors).  Read per call: a getenv is noise next to a launch.
static int conv_persist_grid(int dil, int c1_out, const void* bnx, int grid) {
  const char* pe = getenv("FFA_CONV_PERSIST");
  if ((pe && pe[0] == '0') || dil != 1 || c1_out != 0 || bnx != nullptr) return 0;
  const char* pm = getenv("FFA_CONV_PERSIST_MIN");
  const char* pg = getenv("FFA_CONV_PERSIST_GRID");
  const int min_grid = pm ? atoi(pm) : 0;
  int pgrid = pg ? atoi(pg) : 512;
  pgrid = pgrid < 8 ? 8 : (pgrid / 8) * 8;  // whole groups of 8: a block keeps its XCD across tiles
  if (grid <= min_grid) return 0;
  return pgrid < grid ? pgrid : grid;
}

template <typename T, int KH, int KW, int STRIDE, int RG, int BCO, int WCO, int WPX, int TH, int TW, int HK,
          bool UP = false>
static int launch_hk(const ConvArgs& a, hipStream_t stream) {
  const int grid = ffa_cdiv(a.npt, 8) * 8 * a.ncb;
  if constexpr (KH == 3 && KW == 3 && STRIDE == 1 && HK >= 2 && !UP && WCO == 1 && WPX == 4) {
    const int pgrid = conv_persist_grid(a.dil, a.c1_out, a.bnx, grid);
    if (pgrid > 0) {
      hipLaunchKernelGGL((conv3x3_persist_kernel<T, BCO, TH, TW, HK>), dim3(pgrid), dim3(256), 0, stream, a);
      return ffa_check_launch("conv3x3_persist");
    }
  }
  hipLaunchKernelGGL((conv_igemm_kernel<T, KH, KW, STRIDE, RG, BCO, WCO, WPX, TH, TW, HK, UP>), dim3(grid),
                     dim3(64 * WCO * WPX), 0, stream, a);
  return ffa_check_launch("conv_igemm");
}

// two-source (nearest x2 + concat) launch: 3x3 stride 1 only; the caller checked that C1 is a whole number of
// HK-chunk groups
template <typename T, int BCO, int TH, int TW>
static int launch_up(const ConvArgs& a, hipStream_t stream) {
  if (a.nchunks % 4 == 0) return launch_hk<T, 3, 3, 1, 3, BCO, 1, 4, TH, TW, 4, true>(a, stream);
  return launch_hk<T, 3, 3, 1, 3, BCO, 1, 4, TH, TW, 2, true>(a, stream);
}

template <typename T, int KH, int KW, int STRIDE, int RG, int BCO, int WCO, int WPX, int TH, int TW>
static int launch_cfg(const ConvArgs& a, hipStream_t stream) {
  // deep halo staging for the 3x3 stride-1 workhorse when whole 64- / 128-byte runs of channels exist
  // (a 128-row block was measured equal at best: its weight slab leaves no LDS for the deep halo at two blocks
  // per CU, so 64 rows is the largest block instantiated)
  if constexpr (KH == 3 && STRIDE == 1) {
    if (a.nchunks % 4 == 0) return launch_hk<T, KH, KW, STRIDE, RG, BCO, WCO, WPX, TH, TW, 4>(a, stream);
    if (a.nchunks % 2 == 0) return launch_hk<T, KH, KW, STRIDE, RG, BCO, WCO, WPX, TH, TW, 2>(a, stream);
  }
  return launch_hk<T, KH, KW, STRIDE, RG, BCO, WCO, WPX, TH, TW, 1>(a, stream);
}

template <typename T, int KH, int KW, int STRIDE, int RG>
static int launch_shape(const ConvArgs& a, const ConvPlan& p, hipStream_t stream) {
  const bool wide = (p.tw == 32);
#define FFA_CONV_CASE(BCO_, WCO_, WPX_)                                                      \
  if (p.bco == BCO_) {                                                                       \
    return wide ? launch_cfg<T, KH, KW, STRIDE, RG, BCO_, WCO_, WPX_, 8, 32>(a, stream)      \
                : launch_cfg<T, KH, KW, STRIDE, RG, BCO_, WCO_, WPX_, 16, 16>(a, stream);    \
  }
  FFA_CONV_CASE(64, 1, 4)
  FFA_CONV_CASE(32, 1, 4)
#undef FFA_CONV_CASE
  ffa_set_error("conv: no kernel for bco=%d", p.bco);
  return FFA_ERR_UNSUPPORTED;
}

template <typename T>
static int launch_dtype(const ConvArgs& a, int kh, int kw, int stride, const ConvPlan& p, hipStream_t stream) {
  if (kh == 3 && kw == 3 && stride == 1) return launch_shape<T, 3, 3, 1, 3>(a, p, stream);
  if (kh == 3 && kw == 3 && stride == 2) return launch_shape<T, 3, 3, 2, 3>(a, p, stream);
  if (kh == 1 && kw == 1 && stride == 1) return launch_shape<T, 1, 1, 1, 1>(a, p, stream);
  if (kh == 1 && kw == 1 && stride == 2) return launch_shape<T, 1, 1, 2, 1>(a, p, stream);
  if (kh == 7 && kw == 7 && stride == 2) return launch_shape<T, 7, 7, 2, 1>(a, p, stream);
  ffa_set_error("conv: unsupported kernel %dx%d stride %d", kh, kw, stride);
  return FFA_ERR_UNSUPPORTED;
}

extern "C" int ffa_conv_block_co(int kh, int kw, int stride, int cout) {
  const int bco = conv_pref_bco(kh, kw, stride, cout);
  return conv_supported(kh, kw, stride, bco) ? bco : FFA_ERR_UNSUPPORTED;
}

// ring kernel entry points (conv3x3_ring.hip)
extern "C" long long ffa_ring_stat_rows(int B, int H, int W, int co_rows);
extern "C" int ffa_ring_conv3x3(int dtype, const void* in, const void* w_ring, const float* bias, const void* residual,
                                void* out, float* stat_partials, const float* pro_scale, const float* pro_shift, int B,
                                int H, int W, int Ci, int Co, int co_rows, int relu, hipStream_t stream);
extern "C" int ffa_ring_pack(int dtype, const float* w_oihw, const float* scale, void* dst, int O, int I, int transpose,
                             int co_rows, int ci_pitch, hipStream_t stream);

// thin kernel entry points (conv3x3_thin.hip)
extern "C" int ffa_stem_eligible(int dtype, int kh, int kw, int stride, int cout, int ci_pitch);
extern "C" long long ffa_stem_stat_rows(int B, int Ho, int Wo);
extern "C" int ffa_stem_pack(const float* w_oihw, const float* scale, void* dst, int O, int I, hipStream_t stream);
extern "C" int ffa_stem_conv7x7(const void* in, const void* w_stem, const float* bias, void* out, float* stat_partials, int B,
                                int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int relu, hipStream_t stream);
extern "C" int ffa_thin_eligible(int dtype, int kh, int kw, int stride, int rows_real, int ci_pitch);
extern "C" long long ffa_thin_stat_rows(int B, int H, int W, int ci_pitch);
extern "C" int ffa_thin_conv3x3(const void* in, const void* w_thin, const float* bias, const void* residual, void* out,
                                float* stat_partials, int B, int H, int W, int Ci, int Co, int co_rows, int relu, int up,
                                int pool, hipStream_t stream);
extern "C" int ffa_thin_pack(const float* w_oihw, const float* scale, void* dst, int O, int I, int transpose, int co_rows,
                             int ci_pitch, hipStream_t stream);
extern "C" int ffa_thin_conv3x3_pro(const void* in, const void* w_thin, const float* bias, const void* residual,
                                    void* out, float* stat_partials, const float* pro_scale, const float* pro_shift, int B,
                                    int H, int W, int Ci, int Co, int co_rows, int relu, int up, int pool,
                                    hipStream_t stream);

// Operand layout + block height for a layer: the value to hand to ffa_pack_conv_weight / ffa_conv2d as `bco`.
// Bits 0..11 = rows per block (the padded row count is a multiple of it); FFA_BCO_RING set = the operand is packed
// for the ring kernels of conv3x3_ring.hip (3x3 stride 1 pad 1, >= 64 rows, whole 64-byte groups of input channels;
// plain ffa_conv2d / ffa_conv2d_stats calls only -- pass allow_ring = 0 for operands used by the two-source,
// split-epilogue or zero-insertion (dil = 2) calls).
//   bf16: conv3x3_ring16_kernel (v_mfma_f32_16x16x32_bf16, LDS-DMA weights AND halo) is the default since round 3 --
//         36 / 32.5 / 34.7 us on the 128 / 256 / 512-channel layers against 40 / 37.3 / 38 for conv3x3_persist_kernel
//         (same box); FFA_RING=0 restores the conv_igemm operands;
//   f32:  conv3x3_ring_kernel only with FFA_RING=1 (it equals the persist kernel).
// allow_ring bit 1 (value 2): the caller can also take FFA_BCO_THIN -- conv3x3_thin_kernel for bf16 layers with at most
// 32 stored input channels and 32 rows (plain, statistics, two-source with C2 = 0 and pooled-split with C2 = 0 calls;
// not the zero-insertion or the BatchNorm-backward-partials forms); rows per block 16 or 32.  FFA_THIN=0 disables.
extern "C" int ffa_conv_plan(int dtype, int kh, int kw, int stride, int cout, int ci_pitch, int allow_ring) {
  const int bco = ffa_conv_block_co(kh, kw, stride, cout);
  if (bco < 0) return bco;
  {
    // the ResNet stem (bit 2 of `allow`; FFA_STEM=0 keeps conv_igemm_kernel<7, 7, 2>)
    const char* t = getenv("FFA_STEM");
    if ((allow_ring & 4) && !(t && t[0] == '0') && ffa_stem_eligible(dtype, kh, kw, stride, cout, ci_pitch))
      return 64 | FFA_BCO_STEM;
  }
  {
    const char* t = getenv("FFA_THIN");
    if ((allow_ring & 2) && !(t && t[0] == '0') && ffa_thin_eligible(dtype, kh, kw, stride, cout, ci_pitch))
      return (cout <= 16 ? 16 : 32) | FFA_BCO_THIN | (ci_pitch == 32 ? FFA_BCO_THIN32 : 0);
  }
  allow_ring &= 1;
  const char* e = getenv("FFA_RING");
  const int eb = (dtype == FFA_BF16) ? 2 : 4;
  const bool on = (dtype == FFA_BF16) ? !(e && e[0] == '0') : (e && e[0] == '1');
  if (allow_ring && on && kh == 3 && kw == 3 && stride == 1 && cout >= 64 && (ci_pitch * eb) % 64 == 0)
    return 64 | FFA_BCO_RING;
  return bco;
}

extern "C" int ffa_conv_row_group(int kh) { return conv_rg(kh); }

static int conv2d_impl(int dtype, const void* in, const void* w_packed, const float* bias, const void* residual,
                       void* out, float* stat_partials, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                       int co_rows, int bco, int kh, int kw, int stride, int pad, int dil, int relu,
                       hipStream_t stream, void* out2 = nullptr, int c1_out = 0, const void* bnx = nullptr,
                       const float* bn_sc = nullptr, const float* bn_sh = nullptr) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "conv: bad dtype %d", dtype);
  FFA_REQUIRE(in && w_packed && out, "conv: null pointer");
  FFA_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "conv: bad dims");
  FFA_REQUIRE(Ci % 16 == 0 && Co % 8 == 0, "conv: channel pitch must be a multiple of 16 (in) / 8 (out), got %d/%d", Ci, Co);
  FFA_REQUIRE(dil == 1 || dil == 2, "conv: dil must be 1 or 2");
  FFA_REQUIRE(dil == 1 || stride == 1, "conv: zero-insertion input needs stride 1");
  FFA_REQUIRE((long long)B * Hi * Wi * Ci * (dtype == FFA_BF16 ? 2 : 4) < (1LL << 31),
              "conv: input tensor must be smaller than 2 GiB (32-bit piece offsets)");
  if (bco & FFA_BCO_STEM) {
    FFA_REQUIRE(dtype == FFA_BF16 && kh == 7 && kw == 7 && stride == 2 && pad == 3 && dil == 1 && c1_out == 0 && bnx == nullptr &&
                    residual == nullptr,
                "conv: a stem-layout operand serves the bf16 7x7 stride-2 pad-3 convolution only (no residual input)");
    return ffa_stem_conv7x7(in, w_packed, bias, out, stat_partials, B, Hi, Wi, Ci, Ho, Wo, Co, relu, stream);
  }
  if (bco & FFA_BCO_THIN) {
    FFA_REQUIRE(dtype == FFA_BF16 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && Hi == Ho && Wi == Wo &&
                    bnx == nullptr,
                "conv: a thin-layout operand serves bf16 3x3 stride-1 pad-1 convolutions only");
    if (c1_out > 0) {  // pooled split epilogue: only the all-low-resolution form (no skip part)
      FFA_REQUIRE(c1_out == Co && out2 == nullptr && !stat_partials, "conv: thin pooled form takes C2 = 0 only");
      return ffa_thin_conv3x3(in, w_packed, nullptr, nullptr, out, nullptr, B, Hi, Wi, Ci, Co, co_rows, 0, 0, 1, stream);
    }
    return ffa_thin_conv3x3(in, w_packed, bias, residual, out, stat_partials, B, Hi, Wi, Ci, Co, co_rows, relu, 0, 0,
                            stream);
  }
  if (bco & FFA_BCO_RING) {
    FFA_REQUIRE(kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && Hi == Ho && Wi == Wo && c1_out == 0 &&
                    bnx == nullptr,
                "conv: a ring-layout operand serves plain 3x3 stride-1 pad-1 convolutions only");
    return ffa_ring_conv3x3(dtype, in, w_packed, bias, residual, out, stat_partials, nullptr, nullptr, B, Hi, Wi, Ci, Co,
                            co_rows, relu, stream);
  }
  if (!conv_supported(kh, kw, stride, bco)) {
    ffa_set_error("conv: unsupported kernel %dx%d stride %d block %d", kh, kw, stride, bco);
    return FFA_ERR_UNSUPPORTED;
  }
  ConvPlan p;
  p.bco = bco;
  p.rg = conv_rg(kh);
  p.tw = (Wo >= 32) ? 32 : 16;
  p.th = (Wo >= 32) ? 8 : 16;
  FFA_REQUIRE(co_rows % p.bco == 0, "conv: packed weight rows %d not a multiple of block %d", co_rows, p.bco);
  ConvArgs a;
  a.in = in;
  a.w = w_packed;
  a.out = out;
  a.bias = bias;
  a.stats = stat_partials;
  a.in2 = nullptr;
  a.c1_bytes = 0;
  a.out2 = out2;
  a.c1_out = c1_out;
  a.bnx = bnx;
  a.bn_sc = bn_sc;
  a.bn_sh = bn_sh;
  a.res = residual;
  a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci;
  a.Ho = Ho; a.Wo = Wo; a.Co = Co;
  a.pad = pad; a.dil = dil; a.relu = relu;
  const int eb = (dtype == FFA_BF16) ? 2 : 4;
  a.nchunks = Ci * eb / 32;
  a.tiles_x = ffa_cdiv(Wo, p.tw);
  a.tiles_y = ffa_cdiv(Ho, p.th);
  a.npt = B * a.tiles_x * a.tiles_y;
  a.ncb = co_rows / p.bco;
  if (dtype == FFA_BF16) return launch_dtype<ffa_bf16>(a, kh, kw, stride, p, stream);
  return launch_dtype<float>(a, kh, kw, stride, p, stream);
}

extern "C" int ffa_conv2d(int dtype, const void* in, const void* w_packed, const float* bias, const void* residual,
                          void* out, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int co_rows, int bco,
                          int kh, int kw, int stride, int pad, int dil, int relu, hipStream_t stream) {
  return conv2d_impl(dtype, in, w_packed, bias, residual, out, nullptr, B, Hi, Wi, Ci, Ho, Wo, Co, co_rows, bco, kh, kw,
                     stride, pad, dil, relu, stream);
}

// 3x3 stride-1 pad-1 convolution of relu(in * pro_scale[c] + pro_shift[c]) -- the training-mode BatchNorm + ReLU of the
// PRODUCING layer evaluated while this layer stages its input ("normalise on load": the normalised tensor is never
// written; smp Conv2dReLU / torchvision BasicBlock chains conv -> BN -> ReLU -> conv).  bf16 operands in the ring16 or
// thin layout (ffa_conv_plan); `in` is the PRE-normalisation tensor [B][H][W][Ci] -- or, up != 0, the low-resolution
// map [B][H/2][W/2][Ci] of the skip-less nearest-x2 form (thin layout only).  Zero padding applies to the normalised
// tensor.  stat_partials may be null.  Bit-identical to ffa_bn_apply followed by ffa_conv2d / ffa_conv2d_stats.
// FFA_ERR_UNSUPPORTED for operands in the conv_igemm layout.
extern "C" int ffa_conv2d_pro(int dtype, const void* in, const void* w_packed, const float* bias, const void* residual,
                              void* out, float* stat_partials, const float* pro_scale, const float* pro_shift, int B,
                              int H, int W, int Ci, int Co, int co_rows, int bco, int relu, int up, hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 && in && w_packed && out && pro_scale && pro_shift, "conv_pro: bad arguments");
  if (bco & FFA_BCO_THIN)
    return ffa_thin_conv3x3_pro(in, w_packed, bias, residual, out, stat_partials, pro_scale, pro_shift, B, H, W, Ci, Co,
                                co_rows, relu, up, 0, stream);
  if ((bco & FFA_BCO_RING) && !up)
    return ffa_ring_conv3x3(dtype, in, w_packed, bias, residual, out, stat_partials, pro_scale, pro_shift, B, H, W, Ci, Co,
                            co_rows, relu, stream);
  ffa_set_error("conv_pro: no prologue kernel for this operand layout (bco 0x%x, up %d)", bco, up);
  return FFA_ERR_UNSUPPORTED;
}

// ffa_conv2d whose output is the gradient dy of y = relu(bn(x)) (a dgrad convolution feeding a BatchNorm
// backward): besides writing dy it leaves, per pixel tile, sum(g) and sum(g * x) with g = dy where
// x * bn_scale + bn_shift > 0 else 0, in stat_partials[rows][2][Co] -- the reduction pass of the BatchNorm backward
// comes out of the conv epilogue (ffa_bn_bwd_partials finishes the job).  bnx has the shape / pitch of out.
extern "C" int ffa_conv2d_bnbwd(int dtype, const void* in, const void* w_packed, const void* residual, void* out,
                                float* stat_partials, const void* bnx, const float* bn_scale, const float* bn_shift,
                                int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int co_rows, int bco, int kh,
                                int kw, int stride, int pad, int dil, hipStream_t stream) {
  FFA_REQUIRE(stat_partials && bnx && bn_scale && bn_shift, "conv_bnbwd: null pointer");
  return conv2d_impl(dtype, in, w_packed, nullptr, residual, out, stat_partials, B, Hi, Wi, Ci, Ho, Wo, Co, co_rows,
                     bco, kh, kw, stride, pad, dil, 0, stream, nullptr, 0, bnx, bn_scale, bn_shift);
}

// Number of partial-statistics rows ffa_conv2d_stats writes for an output of B x Ho x Wo pixels (= pixel tiles).
static long long conv_stat_rows_igemm(int B, int Ho, int Wo) {
  const int tw = (Wo >= 32) ? 32 : 16, th = (Wo >= 32) ? 8 : 16;
  return (long long)B * ffa_cdiv(Wo, tw) * ffa_cdiv(Ho, th);
}
// co_rows / bco: the operand the convolution will run with (ffa_conv_plan); the row count is the number of pixel
// tiles of the kernel that serves it
extern "C" long long ffa_conv_stat_rows(int B, int Ho, int Wo, int co_rows, int bco) {
  if (bco & FFA_BCO_STEM) return ffa_stem_stat_rows(B, Ho, Wo);
  if (bco & FFA_BCO_THIN) return ffa_thin_stat_rows(B, Ho, Wo, (bco & FFA_BCO_THIN32) ? 32 : 16);
  if (bco & FFA_BCO_RING) return ffa_ring_stat_rows(B, Ho, Wo, co_rows);
  return conv_stat_rows_igemm(B, Ho, Wo);
}

// 1 when ffa_conv2d / ffa_conv2d_stats run this convolution on the persistent kernel (conv3x3_persist_kernel: its
// own symbol in a profile), 0 for conv_igemm_kernel.  Ci = input channel pitch, co_rows / bco as for ffa_conv2d.
extern "C" int ffa_conv_is_persistent(int dtype, int B, int Ho, int Wo, int Ci, int co_rows, int bco, int kh, int kw,
                                      int stride, int dil) {
  if (!(kh == 3 && kw == 3 && stride == 1) || bco <= 0 || co_rows % bco != 0) return 0;
  const int nchunks = Ci * (dtype == FFA_BF16 ? 2 : 4) / 32;
  if (nchunks % 2 != 0) return 0;  // HK = 1 instantiations have no pipelined path
  if (bco & (FFA_BCO_RING | FFA_BCO_THIN | FFA_BCO_STEM)) return 0;
  const int npt = (int)conv_stat_rows_igemm(B, Ho, Wo);
  return conv_persist_grid(dil, 0, nullptr, ffa_cdiv(npt, 8) * 8 * (co_rows / bco)) > 0 ? 1 : 0;
}

// ffa_conv2d that also leaves per-tile channel sums / sums of squares of the tensor it writes in
// stat_partials[rows][2][Co] (rows = ffa_conv_stat_rows): the BatchNorm batch statistics come out of the conv
// epilogue's registers instead of a second pass over the output (ffa_bn_finalize turns them into scale / shift).
extern "C" int ffa_conv2d_stats(int dtype, const void* in, const void* w_packed, const float* bias,
                                const void* residual, void* out, float* stat_partials, int B, int Hi, int Wi, int Ci,
                                int Ho, int Wo, int Co, int co_rows, int bco, int kh, int kw, int stride, int pad,
                                int dil, int relu, hipStream_t stream) {
  FFA_REQUIRE(stat_partials, "conv_stats: null statistics buffer");
  return conv2d_impl(dtype, in, w_packed, bias, residual, out, stat_partials, B, Hi, Wi, Ci, Ho, Wo, Co, co_rows, bco,
                     kh, kw, stride, pad, dil, relu, stream);
}

// 3x3 stride-1 pad-1 convolution over the VIRTUAL input cat(nearest_x2(lo), skip) of a U-Net decoder block
// (smp DecoderBlock.forward: F.interpolate(scale_factor=2, mode="nearest"), torch.cat, conv1): lo is
// [B][Hl][Wl][C1], skip [B][2Hl][2Wl][C2] (null when C2 == 0), the packed weight has C1 + C2 input channels in that
// order.  Returns FFA_ERR_UNSUPPORTED when C1 does not cover whole channel groups of the kernel's halo staging
// (the caller then materialises the concat with ffa_upsample_nearest2x_concat_fwd).  bias (eval-mode folded
// BatchNorm shift) and stat_partials may be null; relu applies in the epilogue.
extern "C" int ffa_conv2d_upcat(int dtype, const void* lo, const void* skip, const void* w_packed, const float* bias,
                                void* out, float* stat_partials, int B, int Hl, int Wl, int C1, int C2, int Co,
                                int co_rows, int bco, int relu, hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "conv_upcat: bad dtype %d", dtype);
  FFA_REQUIRE(lo && w_packed && out && (skip || C2 == 0), "conv_upcat: null pointer");
  FFA_REQUIRE(B > 0 && Hl > 0 && Wl > 0 && C1 > 0 && C2 >= 0 && C1 % 16 == 0 && C2 % 16 == 0 && Co % 8 == 0,
              "conv_upcat: bad dims");
  const int eb = (dtype == FFA_BF16) ? 2 : 4;
  const int Hi = 2 * Hl, Wi = 2 * Wl, Ci = C1 + C2;
  if (bco & FFA_BCO_THIN) {
    if (C2 != 0 || dtype != FFA_BF16) {
      ffa_set_error("conv_upcat: a thin-layout operand takes the skip-less form only");
      return FFA_ERR_UNSUPPORTED;
    }
    return ffa_thin_conv3x3(lo, w_packed, bias, nullptr, out, stat_partials, B, Hi, Wi, C1, Co, co_rows, relu, 1, 0, stream);
  }
  FFA_REQUIRE((long long)B * Hi * Wi * (C1 > C2 ? C1 : C2) * eb < (1LL << 31),
              "conv_upcat: source tensors must be smaller than 2 GiB (32-bit piece offsets)");
  if (!conv_supported(3, 3, 1, bco) || co_rows % bco != 0) {
    ffa_set_error("conv_upcat: unsupported block %d / rows %d", bco, co_rows);
    return FFA_ERR_UNSUPPORTED;
  }
  const int nchunks = Ci * eb / 32;
  const int hk = (nchunks % 4 == 0) ? 4 : (nchunks % 2 == 0 ? 2 : 1);
  if (hk == 1 || (C1 * eb) % (hk * 32) != 0) {
    ffa_set_error("conv_upcat: C1 = %d does not cover whole %d-byte channel groups", C1, hk * 32);
    return FFA_ERR_UNSUPPORTED;
  }
  ConvArgs a;
  a.in = lo;
  a.in2 = skip;
  a.c1_bytes = C1 * eb;
  a.out2 = nullptr;
  a.c1_out = 0;
  a.bnx = nullptr;
  a.bn_sc = a.bn_sh = nullptr;
  a.w = w_packed;
  a.out = out;
  a.bias = bias;
  a.stats = stat_partials;
  a.res = nullptr;
  a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci;
  a.Ho = Hi; a.Wo = Wi; a.Co = Co;
  a.pad = 1; a.dil = 1; a.relu = relu;
  a.nchunks = nchunks;
  const int tw = (Wi >= 32) ? 32 : 16, th = (Wi >= 32) ? 8 : 16;
  a.tiles_x = ffa_cdiv(Wi, tw);
  a.tiles_y = ffa_cdiv(Hi, th);
  a.npt = B * a.tiles_x * a.tiles_y;
  a.ncb = co_rows / bco;
#define FFA_UP_CASE(T_)                                                         \
  if (bco == 64) return (tw == 32) ? launch_up<T_, 64, 8, 32>(a, stream) : launch_up<T_, 64, 16, 16>(a, stream); \
  return (tw == 32) ? launch_up<T_, 32, 8, 32>(a, stream) : launch_up<T_, 32, 16, 16>(a, stream);
  if (dtype == FFA_BF16) { FFA_UP_CASE(ffa_bf16) }
  FFA_UP_CASE(float)
#undef FFA_UP_CASE
}

// Input gradient of the convolution ffa_conv2d_upcat computes, delivered as the two gradients the decoder block
// needs: dlo [B][Ho/2][Wo/2][C1] (2x2 sums = adjoint of nearest x2) and dskip [B][Ho][Wo][C2] (null when C2 == 0).
// dy is [B][Ho][Wo][Cdy]; w_packed_t is the transposed (dgrad) operand with co_rows >= C1 + C2 rows.  The gradient
// of the concatenated tensor is never written.  FFA_ERR_UNSUPPORTED unless C1 is a whole number of bco-row blocks.
extern "C" int ffa_conv2d_dgrad_upcat(int dtype, const void* dy, const void* w_packed_t, void* dlo, void* dskip, int B,
                                      int Ho, int Wo, int Cdy, int C1, int C2, int co_rows, int bco,
                                      hipStream_t stream) {
  FFA_REQUIRE(dy && w_packed_t && dlo && (dskip || C2 == 0), "dgrad_upcat: null pointer");
  FFA_REQUIRE(Ho % 2 == 0 && Wo % 2 == 0 && C1 > 0 && C2 >= 0 && C1 % 8 == 0 && C2 % 8 == 0, "dgrad_upcat: bad dims");
  if (bco & FFA_BCO_THIN) {
    if (C2 != 0) {
      ffa_set_error("dgrad_upcat: a thin-layout operand takes the skip-less form only");
      return FFA_ERR_UNSUPPORTED;
    }
    return conv2d_impl(dtype, dy, w_packed_t, nullptr, nullptr, dlo, nullptr, B, Ho, Wo, Cdy, Ho, Wo, C1, co_rows, bco, 3,
                       3, 1, 1, 1, 0, stream, nullptr, C1);
  }
  if (C1 % bco != 0) {
    ffa_set_error("dgrad_upcat: C1 = %d is not a multiple of the %d-row block", C1, bco);
    return FFA_ERR_UNSUPPORTED;
  }
  return conv2d_impl(dtype, dy, w_packed_t, nullptr, nullptr, dlo, nullptr, B, Ho, Wo, Cdy, Ho, Wo, C1 + C2, co_rows,
                     bco, 3, 3, 1, 1, 1, 0, stream, dskip, C1);
}

// ------------------------------------------------------------------------------------------------
// weight packing: OIHW f32 master weights -> per-(co block, chunk, row group) LDS-image slabs
//   dst[cb][cc][rg][row < BCO][tap < RG*KW][e < 32/sizeof(T)]
// src element (row, ch, r, s) is read at src[row*s_row + ch*s_ch + r*KW + s]; flip mirrors the taps
// (dgrad: rows = ci, ch = co, flipped).  scale[row] (optional) folds an eval-mode BN into the weights.

struct PackArgs {
  const float* src;
  void* dst;
  const float* scale;
  long long s_row, s_ch;
  int rows, chs;  // valid rows / channels in src
  int kh, kw, rg, bco, nchunks, ncb, flip;
};

template <typename T>
__global__ void pack_weight_kernel(PackArgs p) {
  constexpr int EPC = ElemTraits<T>::kPerStep;
  const int taps = p.rg * p.kw;
  const int nrg = p.kh / p.rg;
  const long long total = (long long)p.ncb * p.nchunks * nrg * p.bco * taps * EPC;
  T* dst = static_cast<T*>(p.dst);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long t = i;
    const int e = t % EPC; t /= EPC;
    const int tap = t % taps; t /= taps;
    const int row_l = t % p.bco; t /= p.bco;
    const int rg = t % nrg; t /= nrg;
    const int cc = t % p.nchunks; t /= p.nchunks;
    const int cb = (int)t;
    int row_in_block = row_l;
    if (p.bco >= 64) {  // fragment order inside each 64-row wave group (see conv_igemm_kernel)
      const int j = row_l & 63, mt = j >> 5, rho = j & 31;
      row_in_block = (row_l & ~63) + 16 * (rho >> 3) + 8 * ((rho >> 2) & 1) + 4 * mt + (rho & 3);
    }
    const int row = cb * p.bco + row_in_block;
    const int ch = cc * EPC + e;
    int r = rg * p.rg + tap / p.kw;
    int s = tap % p.kw;
    float v = 0.f;
    if (row < p.rows && ch < p.chs) {
      if (p.flip) {
        r = p.kh - 1 - r;
        s = p.kw - 1 - s;
      }
      v = p.src[row * p.s_row + ch * p.s_ch + r * p.kw + s];
      if (p.scale) v *= p.scale[row];
    }
    ffa_store_elem<T>(dst + i, v);
  }
}

extern "C" long long ffa_pack_conv_weight_bytes(int dtype, int co_rows, int ci_pitch, int kh, int kw) {
  const long long eb = (dtype == FFA_BF16) ? 2 : 4;
  return (long long)co_rows * ci_pitch * kh * kw * eb;
}

// co_rows / ci_pitch: padded row count (multiple of the block size from ffa_conv_block_co) and the
// channel pitch of the activation the conv will read.  transpose=1 builds the dgrad operand from the
// same OIHW tensor (rows = input channels, channels = output channels, taps mirrored).
extern "C" int ffa_pack_conv_weight(int dtype, const float* w_oihw, const float* scale, void* dst, int O, int I,
                                    int kh, int kw, int transpose, int co_rows, int ci_pitch, int bco, int rg,
                                    hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "pack: bad dtype");
  FFA_REQUIRE(w_oihw && dst, "pack: null pointer");
  if (bco & FFA_BCO_STEM) {
    FFA_REQUIRE(kh == 7 && kw == 7 && dtype == FFA_BF16 && !transpose, "pack: the stem layout is for the bf16 7x7 forward operand");
    return ffa_stem_pack(w_oihw, scale, dst, O, I, stream);
  }
  if (bco & FFA_BCO_THIN) {
    FFA_REQUIRE(kh == 3 && kw == 3 && dtype == FFA_BF16, "pack: the thin layout is for bf16 3x3 kernels");
    return ffa_thin_pack(w_oihw, scale, dst, O, I, transpose, co_rows, ci_pitch, stream);
  }
  if (bco & FFA_BCO_RING) {
    FFA_REQUIRE(kh == 3 && kw == 3, "pack: the ring layout is for 3x3 kernels");
    return ffa_ring_pack(dtype, w_oihw, scale, dst, O, I, transpose, co_rows, ci_pitch, stream);
  }
  FFA_REQUIRE(bco > 0 && co_rows % bco == 0, "pack: rows %d not a multiple of block %d", co_rows, bco);
  FFA_REQUIRE(rg > 0 && kh % rg == 0, "pack: bad row group");
  const int epc = (dtype == FFA_BF16) ? 16 : 8;
  FFA_REQUIRE(ci_pitch % 16 == 0, "pack: channel pitch must be a multiple of 16");
  PackArgs p;
  p.src = w_oihw;
  p.dst = dst;
  p.scale = scale;
  if (!transpose) {
    p.rows = O; p.chs = I;
    p.s_row = (long long)I * kh * kw;
    p.s_ch = (long long)kh * kw;
    p.flip = 0;
  } else {
    p.rows = I; p.chs = O;
    p.s_row = (long long)kh * kw;
    p.s_ch = (long long)I * kh * kw;
    p.flip = 1;
  }
  FFA_REQUIRE(p.rows <= co_rows && p.chs <= ci_pitch, "pack: padded dims smaller than the tensor");
  p.kh = kh; p.kw = kw; p.rg = rg; p.bco = bco;
  p.nchunks = ci_pitch / epc;
  p.ncb = co_rows / bco;
  const long long total = (long long)co_rows * ci_pitch * kh * kw;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<ffa_bf16>, dim3(grid), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, stream, p);
  return ffa_check_launch("pack_weight");
}

// ------------------------------------------------------------------------------------------------
// batched packing: every conv weight of the network (forward and dgrad operands) in ONE launch per step.
// The descriptor table lives in device memory and is built once (ffa_pack_desc_fill on the host, then copied);
// blockIdx.y selects the descriptor, blockIdx.x strides over its elements.

// one thread = eight consecutive channels of one (row, tap): the index arithmetic (seven divisions by run-time
// values) is paid once per 16-byte store instead of once per element (201 -> ~50 us for the 186 operands of the
// U-Net at batch time)
__device__ __forceinline__ void pack_eight(const PackArgs& p, long long i8, int dtype) {
  const int EPC = (dtype == FFA_BF16) ? 16 : 8;
  const int G8 = EPC / 8;
  const int taps = p.rg * p.kw;
  const int nrg = p.kh / p.rg;
  long long t = i8;
  const int e0 = (int)(t % G8) * 8; t /= G8;
  const int tap = t % taps; t /= taps;
  const int row_l = t % p.bco; t /= p.bco;
  const int rg = t % nrg; t /= nrg;
  const int cc = t % p.nchunks; t /= p.nchunks;
  const int cb = (int)t;
  int row_in_block = row_l;
  if (p.bco >= 64) {
    const int j = row_l & 63, mt = j >> 5, rho = j & 31;
    row_in_block = (row_l & ~63) + 16 * (rho >> 3) + 8 * ((rho >> 2) & 1) + 4 * mt + (rho & 3);
  }
  const int row = cb * p.bco + row_in_block;
  const int ch0 = cc * EPC + e0;
  int r = rg * p.rg + tap / p.kw;
  int sx = tap % p.kw;
  if (p.flip) {
    r = p.kh - 1 - r;
    sx = p.kw - 1 - sx;
  }
  float v[8];
  const bool row_ok = row < p.rows;
  const float sc = (row_ok && p.scale) ? p.scale[row] : 1.f;
  const float* src = p.src + (long long)row * p.s_row + r * p.kw + sx;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = ch0 + j;
    v[j] = (row_ok && ch < p.chs) ? src[(long long)ch * p.s_ch] * sc : 0.f;
  }
  if (dtype == FFA_BF16)
    ffa_store8<ffa_bf16>(static_cast<ffa_bf16*>(p.dst) + i8 * 8, v);
  else
    ffa_store8<float>(static_cast<float*>(p.dst) + i8 * 8, v);
}

__global__ void pack_weight_batched_kernel(const PackArgs* __restrict__ descs, int dtype) {
  const PackArgs p = descs[blockIdx.y];
  const int EPC = (dtype == FFA_BF16) ? 16 : 8;
  const long long total8 = (long long)p.ncb * p.nchunks * (p.kh / p.rg) * p.bco * (p.rg * p.kw) * (EPC / 8);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8;
       i += (long long)gridDim.x * blockDim.x)
    pack_eight(p, i, dtype);
}

extern "C" int ffa_pack_desc_bytes(void) { return (int)sizeof(PackArgs); }

// Fills one HOST descriptor (same arguments as ffa_pack_conv_weight, pointers are device pointers).
extern "C" int ffa_pack_desc_fill(void* host_desc, const float* w_oihw, const float* scale, void* dst, int O, int I,
                                  int kh, int kw, int transpose, int co_rows, int ci_pitch, int bco, int rg,
                                  int dtype) {
  FFA_REQUIRE(host_desc && w_oihw && dst, "pack_desc_fill: null pointer");
  FFA_REQUIRE(bco > 0 && co_rows % bco == 0 && rg > 0 && kh % rg == 0 && ci_pitch % 16 == 0,
              "pack_desc_fill: bad geometry");
  PackArgs p;
  memset(&p, 0, sizeof(p));
  p.src = w_oihw;
  p.dst = dst;
  p.scale = scale;
  if (!transpose) {
    p.rows = O; p.chs = I;
    p.s_row = (long long)I * kh * kw;
    p.s_ch = (long long)kh * kw;
    p.flip = 0;
  } else {
    p.rows = I; p.chs = O;
    p.s_row = (long long)kh * kw;
    p.s_ch = (long long)I * kh * kw;
    p.flip = 1;
  }
  FFA_REQUIRE(p.rows <= co_rows && p.chs <= ci_pitch, "pack_desc_fill: padded dims smaller than the tensor");
  p.kh = kh; p.kw = kw; p.rg = rg; p.bco = bco;
  p.nchunks = ci_pitch / ((dtype == FFA_BF16) ? 16 : 8);
  p.ncb = co_rows / bco;
  memcpy(host_desc, &p, sizeof(p));
  return FFA_OK;
}

// The same descriptor for the column block W[:, col0 : col0 + ncols] of a weight with I_total input channels (one
// modality's share of a FusionHandler 1x1 convolution, flair_hub/models/flair_model.py:470-475): the source strides
// stay those of the whole tensor, so the slices of every stage ride in the batched pack instead of one copy + one
// pack launch each.
extern "C" int ffa_pack_desc_fill_cols(void* host_desc, const float* w_oihw, const float* scale, void* dst, int O,
                                       int I_total, int col0, int ncols, int kh, int kw, int transpose, int co_rows,
                                       int ci_pitch, int bco, int rg, int dtype) {
  FFA_REQUIRE(host_desc && w_oihw && dst, "pack_desc_fill_cols: null pointer");
  FFA_REQUIRE(col0 >= 0 && ncols > 0 && col0 + ncols <= I_total, "pack_desc_fill_cols: column block outside the tensor");
  FFA_REQUIRE(bco > 0 && co_rows % bco == 0 && rg > 0 && kh % rg == 0 && ci_pitch % 16 == 0,
              "pack_desc_fill_cols: bad geometry");
  PackArgs p;
  memset(&p, 0, sizeof(p));
  p.src = w_oihw + (long long)col0 * kh * kw;
  p.dst = dst;
  p.scale = scale;
  if (!transpose) {
    p.rows = O; p.chs = ncols;
    p.s_row = (long long)I_total * kh * kw;
    p.s_ch = (long long)kh * kw;
    p.flip = 0;
  } else {
    p.rows = ncols; p.chs = O;
    p.s_row = (long long)kh * kw;
    p.s_ch = (long long)I_total * kh * kw;
    p.flip = 1;
  }
  FFA_REQUIRE(p.rows <= co_rows && p.chs <= ci_pitch, "pack_desc_fill_cols: padded dims smaller than the block");
  p.kh = kh; p.kw = kw; p.rg = rg; p.bco = bco;
  p.nchunks = ci_pitch / ((dtype == FFA_BF16) ? 16 : 8);
  p.ncb = co_rows / bco;
  memcpy(host_desc, &p, sizeof(p));
  return FFA_OK;
}

extern "C" int ffa_pack_conv_weights_batched(int dtype, const void* descs_device, int n, hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "pack_batched: bad dtype");
  FFA_REQUIRE(descs_device && n > 0 && n <= 65535, "pack_batched: bad descriptor table");
  // 256 blocks per descriptor: the largest operands (2.4 M elements) set the tail, small ones exit at once
  hipLaunchKernelGGL(pack_weight_batched_kernel, dim3(256, n), dim3(256), 0, stream,
                     static_cast<const PackArgs*>(descs_device), dtype);
  return ffa_check_launch("pack_weight_batched");
}

#if FFA_CONV_TRACE
extern "C" int ffa_conv_trace_read(long long* host_dst, int n) {
  if (n > 64 * 256) n = 64 * 256;
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ffa_conv_trace_buf), (size_t)n * sizeof(long long));
}
extern "C" int ffa_conv_trace_clear() {
  static long long zeros[64 * 256];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ffa_conv_trace_buf), zeros, sizeof(zeros));
}
#endif
