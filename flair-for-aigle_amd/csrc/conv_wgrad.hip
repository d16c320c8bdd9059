// Weight gradient of the NHWC convolutions on MFMA (gfx950).
//
//   dW[co][r][s][ci] = sum over (b, oy, ox) of dy[b][oy][ox][co] * x[b][oy*st - pad + r][ox*st - pad + s][ci]
//
// (the backward of the conv2d calls named in conv_igemm.hip).  GEMM view: M = co, N = ci, one
// accumulator tile per tap, K = output pixels.  Both operands have K (pixels) as their slow memory
// axis in NHWC, so the tiles are staged in natural [pixel][32 channels] form and the K-contiguous
// MFMA fragments are produced by the LDS transpose read ds_read_b64_tr_b16 (bf16) or by plain
// ds_read_b32 (f32: one element per lane per v_mfma_f32_32x32x2_f32).  Because the tr read takes a
// per-lane row address, the 9 tap shifts of the input halo cost nothing: one halo tile in LDS
// serves all taps.
//
// Work split: block = (co block, ci block, kernel-row group) x split; a split walks a strided
// subset of the spatial tiles and writes its f32 partial slab; wgrad_reduce_kernel sums the
// slabs in fixed order (deterministic, no float atomics) and writes the OIHW f32 gradient.
#include "ffa_common.h"
#include <hip/hip_ext.h>

#include <stdlib.h>

// Developer instrumentation (never built by flairhip/build.py): -DFFA_WGRAD_TRACE=1 makes wave 0 of the first 64
// blocks log s_memtime at every phase boundary (tools/conv_trace.py).
#ifndef FFA_WGRAD_TRACE
#define FFA_WGRAD_TRACE 0
#endif
#if FFA_WGRAD_TRACE
__device__ long long ffa_wgrad_trace_buf[64 * 256];
#define FFA_WTRACE(slot_)                                                                   \
  if (trace_on) {                                                                           \
    if (trace_n < 256)                                                                      \
      ffa_wgrad_trace_buf[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + trace_n] =          \
          (long long)(slot_) << 56 | (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFFFLL);    \
    ++trace_n;                                                                              \
  }
#else
#define FFA_WTRACE(slot_)
#endif

struct WgradArgs {
  const void* x;
  const void* dy;
  float* slabs;  // [nsplit][CoT][KH*KW][CiT]
  int B, Hi, Wi, Ci;
  int Ho, Wo, Co;
  int pad;
  int tiles_x, tiles_y, npt;
  int ncob, ncib, nsplit;
  // two-source input (x = nearest_x2(lo) ++ skip of a U-Net decoder block): x is lo [B][Hi/2][Wi/2][C1], x2 the
  // skip tensor [B][Hi][Wi][Ci - C1] (null when C1 == Ci); x2 == null && C1 == 0: ordinary single input
  const void* x2;
  int C1;
  int CoT, CiT;
  // "normalise on load": the convolution input was relu(x * pro_sc[c] + pro_sh[c]) (BatchNorm + ReLU of the producing
  // layer folded into the consumer): the staged x pieces are rewritten accordingly, padding stays zero
  const float* pro_sc;
  const float* pro_sh;
  int ci_real;  // real input channels (<= Ci): the stem's 8-channel form needs <= 8
};

// WCO x WCI waves own distinct (co, ci) sub-tiles; WK further waves split the tile's k-steps (pixels) and
// write their own partial slab, which keeps 256-thread blocks (fast staging) for layers with few channels.
template <int KH, int KW, int STRIDE, int RG, int WCO, int WCI, int WK, int TH, int TW, int EB>
struct WgradGeom {
  static constexpr int NTHR = 64 * WCO * WCI * WK;
  static constexpr int NPX = TH * TW;
  static constexpr bool ONE = (KH == 1 && KW == 1);
  static constexpr int LS = ONE ? 1 : STRIDE;
  static constexpr int GSTEP = ONE ? STRIDE : 1;
  static constexpr int NRG = KH / RG;
  static constexpr int IH = (TH - 1) * LS + RG;
  static constexpr int IW = (TW - 1) * LS + KW;
  static constexpr int TAPS = RG * KW;
  static constexpr int ROWB = 32 * EB;         // bytes of one pixel row of a 32-channel plane
  static constexpr int PARTS = ROWB / 16;      // 16-byte pieces per pixel row
  static constexpr int DY_BYTES = WCO * NPX * ROWB;
  static constexpr int IN_BYTES = WCI * IH * IW * ROWB;
  static constexpr int STAGE_BYTES = DY_BYTES + IN_BYTES;
  // WK > 1: the k-split wave groups are summed through LDS before ONE slab leaves the block (a slab costs a
  // write + a read of Co*taps*Ci*4 bytes in HBM; with 512 single-slab blocks that was half of the kernel time)
  static constexpr bool MERGE = (WK > 1) && (WCO * WCI == 4 || KH == 7);  // small-channel blocks keep per-group slabs
  static constexpr int MERGE_BYTES = MERGE ? WCO * WCI * (WK - 1) * TAPS * 16 * 64 * 4 : 0;
  static constexpr int LDS_BYTES = STAGE_BYTES > MERGE_BYTES ? STAGE_BYTES : MERGE_BYTES;
  static constexpr int DY_PIECES = WCO * NPX * PARTS;
  static constexpr int IN_PIECES = WCI * IH * IW * PARTS;
  static constexpr int NDP = (DY_PIECES + NTHR - 1) / NTHR;
  static constexpr int NIP = (IN_PIECES + NTHR - 1) / NTHR;
  static constexpr int KSTEPS = NPX / 16;
  static_assert(TW % 16 == 0, "a 16-pixel k-step must stay inside one tile row");
  static_assert(KSTEPS % WK == 0, "k-steps must divide over the k-split waves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// 4 pixels x 16 channels of bf16, transposed: lane (16-lane group member li) passes the address of
// pixel row (li >> 2), 8-byte segment (li & 3); it receives channel li of the four pixels.
__device__ __forceinline__ ffa_s16x4 lds_read_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) ffa_s16x4*)(const_cast<unsigned char*>(p)));
}

template <typename T, int KH, int KW, int STRIDE, int RG, int WCO, int WCI, int WK, int TH, int TW, bool PRO = false>
__global__ void __launch_bounds__(64 * WCO * WCI * WK, (sizeof(T) == 2 ? 2 : 1)) conv_wgrad_kernel(WgradArgs a) {
  constexpr int EB = ElemTraits<T>::kBytes;
  static_assert(!PRO || (EB == 2 && KH == 3 && STRIDE == 1), "the prologue lives in the bf16 3x3 stride-1 path");
  using G = WgradGeom<KH, KW, STRIDE, RG, WCO, WCI, WK, TH, TW, EB>;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];
  unsigned char* sDy = smem;
  unsigned char* sIn = smem + G::DY_BYTES;

  const int tid = threadIdx.x;
#if FFA_WGRAD_TRACE
  const bool trace_on = (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 64);
  int trace_n = 0;
#endif
  FFA_WTRACE(0)
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wk = wave / (WCO * WCI);
  const int wco = (wave / WCI) % WCO, wci = wave % WCI;

  int tb = blockIdx.x;
  const int rg = tb % G::NRG;
  tb /= G::NRG;
  const int cib = tb % a.ncib;
  const int cob = tb / a.ncib;
  const int split = blockIdx.y;
  const int co0 = cob * 32 * WCO;  // first output channel of the block
  const int ci0 = cib * 32 * WCI;

  ffa_f32x16 acc[G::TAPS];
#pragma unroll
  for (int t = 0; t < G::TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const unsigned char* x_b = static_cast<const unsigned char*>(a.x);
  const unsigned char* dy_b = static_cast<const unsigned char*>(a.dy);
  // Two-source input: every 32-channel plane of this block lies in ONE source (the host checks that C1 is a
  // multiple of the block's channel extent), so the block picks base pointer, pixel pitch, image size and the
  // channel origin once.  srcA = the low-res map, read at (y >> 1, x >> 1).
  const bool srcA = a.C1 > 0 && ci0 < a.C1;
  const bool srcB = a.C1 > 0 && !srcA;
  if (srcB) x_b = static_cast<const unsigned char*>(a.x2);
  const int xC = srcA ? a.C1 : (srcB ? a.Ci - a.C1 : a.Ci);  // channel pitch of the source actually read
  const int xc0 = srcB ? ci0 - a.C1 : ci0;                   // first channel of the block inside that source
  const int xH = srcA ? (a.Hi >> 1) : a.Hi, xW = srcA ? (a.Wi >> 1) : a.Wi;

  // fragment addressing (see file header)
  const int li = lane & 15;
  const int gsel = (lane >> 4) & 1;
  const int khalf = lane >> 5;
  const unsigned char* dyPlane = sDy + wco * (G::NPX * G::ROWB);
  const unsigned char* inPlane = sIn + wci * (G::IH * G::IW * G::ROWB);

  // staging registers: the NEXT tile's global loads are issued before the MFMAs of the current tile
  ffa_u32x4 dreg[G::NDP];
  ffa_u32x4 ireg[G::NIP];

  // Tile-independent piece geometry, computed once: byte offset relative to the tile origin (-1: the piece never
  // exists -- past the piece count or in a pad channel block) and its (row, column) inside the tile / halo, packed
  // as row << 8 | column.  Per tile only the origin (scalar) and the bounds tests remain: the per-piece index
  // arithmetic (divisions, 64-bit multiplies, ~250 cycles per piece) used to keep all eight waves out of the
  // matrix pipe for a quarter of every tile (tools/conv_trace.py).
  constexpr bool PRE = (EB == 2 && KH == 3 && STRIDE == 1);  // elsewhere the piece tables would spill
  int drel[PRE ? G::NDP : 1], irel[PRE ? G::NIP : 1];
  unsigned short dpos[PRE ? G::NDP : 1], ipos[PRE ? G::NIP : 1];
#pragma unroll
  for (int k = 0; k < (PRE ? G::NDP : 0); ++k) {
    const int i = tid + k * G::NTHR;
    const int part = i % G::PARTS;
    const int n = (i / G::PARTS) % G::NPX;
    const int plane = i / (G::PARTS * G::NPX);
    const int c = co0 + plane * 32 + part * (16 / EB);
    const int r = n / TW, col = n % TW;
    dpos[k] = (unsigned short)((r << 8) | col);
    drel[k] = (i < G::DY_PIECES && c < a.Co) ? ((r * a.Wo + col) * a.Co + c) * EB : -1;
  }
#pragma unroll
  for (int k = 0; k < (PRE ? G::NIP : 0); ++k) {
    const int i = tid + k * G::NTHR;
    const int part = i % G::PARTS;
    const int q = (i / G::PARTS) % (G::IH * G::IW);
    const int plane = i / (G::PARTS * G::IH * G::IW);
    const int c = xc0 + plane * 32 + part * (16 / EB);
    const int r = (q / G::IW) * G::GSTEP, col = (q % G::IW) * G::GSTEP;
    const bool exists = i < G::IN_PIECES && c < xC;
    ipos[k] = (unsigned short)((r << 8) | col);
    // low-res source: the halo origin is odd (tile origin - 1), so (origin + r) >> 1 = (origin + 1) / 2 + ((r - 1) >> 1)
    const int rr = srcA ? ((r - 1) >> 1) : r, cc = srcA ? ((col - 1) >> 1) : col;
    // signed offset (negative for the first halo row / column of the low-res source); INT_MIN = never exists
    irel[k] = exists ? ((rr * xW + cc) * xC + c) * EB : (int)0x80000000;
  }
  static_assert(!PRE || (G::IH * G::GSTEP < 256 && G::IW * G::GSTEP < 256), "packed piece position");

#define FFA_WG_LOAD_GENERIC(pt_)                                                                                       \
  {                                                                                                            \
    const int tx_ = (pt_) % a.tiles_x;                                                                         \
    const int t2_ = (pt_) / a.tiles_x;                                                                         \
    const int ty_ = t2_ % a.tiles_y;                                                                           \
    const int b_ = t2_ / a.tiles_y;                                                                            \
    const int oy0_ = ty_ * TH, ox0_ = tx_ * TW;                                                                \
    const int iy0_ = oy0_ * STRIDE - a.pad + rg * RG;                                                          \
    const int ix0_ = ox0_ * STRIDE - a.pad;                                                                    \
    _Pragma("unroll") for (int k = 0; k < G::NDP; ++k) {                                                       \
      const int i = tid + k * G::NTHR;                                                                         \
      const int part = i % G::PARTS;                                                                           \
      const int n = (i / G::PARTS) % G::NPX;                                                                   \
      const int plane = i / (G::PARTS * G::NPX);                                                               \
      const int oy = oy0_ + n / TW, ox = ox0_ + n % TW;                                                        \
      const int c = co0 + plane * 32 + part * (16 / EB);                                                       \
      ffa_u32x4 v = ffa_u32x4{0u, 0u, 0u, 0u};                                                                 \
      if (i < G::DY_PIECES && oy < a.Ho && ox < a.Wo && c < a.Co)                                              \
        v = *reinterpret_cast<const ffa_u32x4*>(dy_b +                                                         \
                                                (((size_t)(b_ * a.Ho + oy) * a.Wo + ox) * a.Co + c) * EB);     \
      dreg[k] = v;                                                                                             \
    }                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < G::NIP; ++k) {                                                       \
      const int i = tid + k * G::NTHR;                                                                         \
      const int part = i % G::PARTS;                                                                           \
      const int q = (i / G::PARTS) % (G::IH * G::IW);                                                          \
      const int plane = i / (G::PARTS * G::IH * G::IW);                                                        \
      const int vy = iy0_ + (q / G::IW) * G::GSTEP;                                                            \
      const int vx = ix0_ + (q % G::IW) * G::GSTEP;                                                            \
      const int c = xc0 + plane * 32 + part * (16 / EB);                                                       \
      ffa_u32x4 v = ffa_u32x4{0u, 0u, 0u, 0u};                                                                 \
      if (i < G::IN_PIECES && vy >= 0 && vx >= 0 && vy < a.Hi && vx < a.Wi && c < xC) {                        \
        const int sy_ = srcA ? (vy >> 1) : vy, sx_ = srcA ? (vx >> 1) : vx;                                    \
        v = *reinterpret_cast<const ffa_u32x4*>(x_b + (((size_t)(b_ * xH + sy_) * xW + sx_) * xC + c) * EB);   \
      }                                                                                                        \
      ireg[k] = v;                                                                                             \
    }                                                                                                          \
  }

#define FFA_WG_LOAD(pt_)                                                                                       \
  if constexpr (!PRE) FFA_WG_LOAD_GENERIC(pt_) else {                                                          \
    const int tx_ = (pt_) % a.tiles_x;                                                                         \
    const int t2_ = (pt_) / a.tiles_x;                                                                         \
    const int ty_ = t2_ % a.tiles_y;                                                                           \
    const int b_ = t2_ / a.tiles_y;                                                                            \
    const int oy0_ = ty_ * TH, ox0_ = tx_ * TW;                                                                \
    const int iy0_ = oy0_ * STRIDE - a.pad + rg * RG;                                                          \
    const int ix0_ = ox0_ * STRIDE - a.pad;                                                                    \
    const unsigned char* dyt_ = dy_b + ((long long)(b_ * a.Ho + oy0_) * a.Wo + ox0_) * (long long)(a.Co * EB); \
    const int by_ = srcA ? ((iy0_ + 1) >> 1) : iy0_, bx_ = srcA ? ((ix0_ + 1) >> 1) : ix0_;                    \
    const unsigned char* xt_ = x_b + ((long long)(b_ * xH + by_) * xW + bx_) * (long long)(xC * EB);           \
    _Pragma("unroll") for (int k = 0; k < G::NDP; ++k) {                                                       \
      ffa_u32x4 v = ffa_u32x4{0u, 0u, 0u, 0u};                                                                 \
      if (drel[k] >= 0 && oy0_ + (dpos[k] >> 8) < a.Ho && ox0_ + (dpos[k] & 0xff) < a.Wo)                      \
        v = *reinterpret_cast<const ffa_u32x4*>(dyt_ + (unsigned)drel[k]);                                     \
      dreg[k] = v;                                                                                             \
    }                                                                                                          \
    ivalid = 0u;                                                                                               \
    _Pragma("unroll") for (int k = 0; k < G::NIP; ++k) {                                                       \
      ffa_u32x4 v = ffa_u32x4{0u, 0u, 0u, 0u};                                                                 \
      if (irel[k] != (int)0x80000000 && (unsigned)(iy0_ + (ipos[k] >> 8)) < (unsigned)a.Hi &&                  \
          (unsigned)(ix0_ + (ipos[k] & 0xff)) < (unsigned)a.Wi) {                                              \
        v = *reinterpret_cast<const ffa_u32x4*>(xt_ + (long long)irel[k]);                                     \
        ivalid |= 1u << k;                                                                                     \
      }                                                                                                        \
      ireg[k] = v;                                                                                             \
    }                                                                                                          \
  }
#define FFA_WG_STORE()                                                                          \
  {                                                                                             \
    _Pragma("unroll") for (int k = 0; k < G::NDP; ++k) {                                        \
      const int i = tid + k * G::NTHR;                                                          \
      if (G::DY_PIECES % G::NTHR == 0 || i < G::DY_PIECES)                                      \
        *reinterpret_cast<ffa_u32x4*>(sDy + (size_t)i * 16) = dreg[k];                          \
    }                                                                                           \
    _Pragma("unroll") for (int k = 0; k < G::NIP; ++k) {                                        \
      const int i = tid + k * G::NTHR;                                                          \
      if (G::IN_PIECES % G::NTHR == 0 || i < G::IN_PIECES) {                                    \
        ffa_u32x4 v = ireg[k];                                                                  \
        if constexpr (PRO) {                                                                    \
          if (svalid & (1u << k)) v = pro_piece(v, i);                                          \
        }                                                                                       \
        *reinterpret_cast<ffa_u32x4*>(sIn + (size_t)i * 16) = v;                                \
      }                                                                                         \
    }                                                                                           \
  }
  // PRO: relu(x * sc + sh) on the 8 channels of piece i (channel = block origin + 32 * plane + 8 * part), same fma /
  // max / rounding as ffa_bn_apply; only pieces that were really loaded (svalid): the zero padding stays zero
  unsigned ivalid = 0u, svalid = 0u;
  auto pro_piece = [&](ffa_u32x4 v, int i) {
    const int part = i % G::PARTS;
    const int plane = i / (G::PARTS * G::IH * G::IW);
    const int c = xc0 + plane * 32 + part * 8;
    float sc[8], sh[8], f[8];
    ffa_load8<float>(a.pro_sc + c, sc);
    ffa_load8<float>(a.pro_sh + c, sh);
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fmaxf(__builtin_fmaf(f[e], sc[e], sh[e]), 0.f);
    v.x = ffa_pack_bf16x2(f[0], f[1]);
    v.y = ffa_pack_bf16x2(f[2], f[3]);
    v.z = ffa_pack_bf16x2(f[4], f[5]);
    v.w = ffa_pack_bf16x2(f[6], f[7]);
    return v;
  };

  int pt = split;
  if (pt < a.npt) FFA_WG_LOAD(pt)
  FFA_WTRACE(1)
  for (; pt < a.npt; pt += a.nsplit) {
    __syncthreads();  // previous tile's fragment reads are done
    FFA_WTRACE(2)
    svalid = ivalid;  // validity of the pieces now in ireg[] (the next FFA_WG_LOAD rewrites ivalid)
    FFA_WG_STORE()    // piece i lives at byte i*16: [plane][pixel][32 ch] is linear in the piece index
    FFA_WTRACE(3)
    __syncthreads();
    FFA_WTRACE(4)
    if (pt + a.nsplit < a.npt) FFA_WG_LOAD(pt + a.nsplit)
    FFA_WTRACE(5)

    // ---- K loop over the tile's pixels, 16 per step; wave wk takes steps wk, wk + WK, ...
#pragma unroll 1
    for (int ks = wk; ks < G::KSTEPS; ks += WK) {
      const int n0 = ks * 16;
      const int py = n0 / TW, px0 = n0 % TW;
      if constexpr (EB == 2) {
        // A fragment: co = lane & 31, k = 8*khalf + j
        const int pa = n0 + 8 * khalf + (li >> 2);
        const unsigned char* ap = dyPlane + pa * G::ROWB + gsel * 32 + (li & 3) * 8;
        const ffa_s16x4 a0 = lds_read_tr16(ap);
        const ffa_s16x4 a1 = lds_read_tr16(ap + 4 * G::ROWB);
        ffa_u32x4 af;
        af.x = __builtin_bit_cast(ffa_u32x2, a0).x;
        af.y = __builtin_bit_cast(ffa_u32x2, a0).y;
        af.z = __builtin_bit_cast(ffa_u32x2, a1).x;
        af.w = __builtin_bit_cast(ffa_u32x2, a1).y;
        const int pxl = px0 + 8 * khalf + (li >> 2);
        const unsigned char* bbase =
            inPlane + ((py * G::LS) * G::IW + pxl * G::LS) * G::ROWB + gsel * 32 + (li & 3) * 8;
        // B fragments double-buffered: the two transpose reads of tap t+1 are issued before the MFMA of tap t
        auto load_b = [&](int tap) {
          const unsigned char* bp = bbase + ((tap / KW) * G::IW + (tap % KW)) * G::ROWB;
          const ffa_s16x4 b0 = lds_read_tr16(bp);
          const ffa_s16x4 b1 = lds_read_tr16(bp + 4 * G::LS * G::ROWB);
          ffa_u32x4 bf;
          bf.x = __builtin_bit_cast(ffa_u32x2, b0).x;
          bf.y = __builtin_bit_cast(ffa_u32x2, b0).y;
          bf.z = __builtin_bit_cast(ffa_u32x2, b1).x;
          bf.w = __builtin_bit_cast(ffa_u32x2, b1).y;
          return bf;
        };
        // prefetch distance 2: with one MFMA per tap a single tap (32 cycles) is shorter than the LDS latency
        ffa_u32x4 bq[3];
        bq[0] = load_b(0);
        if (G::TAPS > 1) bq[1] = load_b(1);
#pragma unroll
        for (int tap = 0; tap < G::TAPS; ++tap) {
          __builtin_amdgcn_sched_barrier(0);  // one scheduling region per tap (else hipcc sinks the prefetch)
          if (tap + 2 < G::TAPS) bq[(tap + 2) % 3] = load_b(tap + 2);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(ffa_bf16x8, af), __builtin_bit_cast(ffa_bf16x8, bq[tap % 3]), acc[tap], 0, 0, 0);
          if (tap + 2 < G::TAPS) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
      } else {
        // f32: 8 MFMAs of k = 2 pixels; lane supplies A[co = lane&31][k = lane>>5], B[k][ci = lane&31]
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int n = n0 + 2 * m + khalf;
          const float av = *reinterpret_cast<const float*>(dyPlane + n * G::ROWB + (lane & 31) * 4);
          const int pxl = px0 + 2 * m + khalf;
          const unsigned char* bbase = inPlane + ((py * G::LS) * G::IW + pxl * G::LS) * G::ROWB + (lane & 31) * 4;
#pragma unroll
          for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int s = 0; s < KW; ++s) {
              const float bv = *reinterpret_cast<const float*>(bbase + (r * G::IW + s) * G::ROWB);
              acc[r * KW + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[r * KW + s], 0, 0, 0);
            }
        }
      }
    }
    FFA_WTRACE(6)
  }
#undef FFA_WG_LOAD
#undef FFA_WG_LOAD_GENERIC
#undef FFA_WG_STORE

  // ---- sum the k-split wave groups through LDS (fixed order: group 0 + group 1 + ...)
  if constexpr (G::MERGE) {
    float* red = reinterpret_cast<float*>(smem);
    const int wq = wco * WCI + wci;
    __syncthreads();  // every wave is done with the staging buffers
    if (wk > 0) {
      float* dst = red + ((size_t)((wk - 1) * WCO * WCI + wq) * G::TAPS * 16) * 64 + lane;
#pragma unroll
      for (int t = 0; t < G::TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * 64] = acc[t][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int g = 0; g < WK - 1; ++g) {
      const float* src = red + ((size_t)(g * WCO * WCI + wq) * G::TAPS * 16) * 64 + lane;
#pragma unroll
      for (int t = 0; t < G::TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += src[(t * 16 + r) * 64];
    }
  }

  // ---- write the partial slab: D[co][ci], lane column = ci, rows co = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int ci = ci0 + wci * 32 + (lane & 31);
  const int co_w = co0 + wco * 32;
  const size_t taps_total = (size_t)KH * KW;
  const size_t slab = G::MERGE ? (size_t)split : (size_t)split * WK + wk;
#pragma unroll
  for (int t = 0; t < G::TAPS; ++t) {
    const int tapg = rg * G::TAPS + t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_w + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      a.slabs[((slab * a.CoT + co) * taps_total + tapg) * a.CiT + ci] = acc[t][r];
    }
  }
  FFA_WTRACE(7)
}

// ------------------------------------------------------------------------------------------------
// Ring variant for the bulk of the network (bf16, 3x3 stride 1, >= 64 channels on both sides): one block per CU,
// one wave per SIMD, up to 512 VGPRs per lane.  The spatial tiles of a split stream through a 3-slot LDS ring
// filled by global_load_lds (LDS-DMA, no staging registers): while tile t is multiplied, tiles t+1 and t+2
// are in flight, tracked with counted vmcnt and ONE raw s_barrier per tile.  A whole k-step's fragments (A + 9
// taps of B) are double-buffered in registers so the transpose reads of step s+1 hide under the MFMAs of step s.
// Out-of-image halo pixels and pad channels: the DMA is issued for every lane (wave-uniform instruction count
// keeps vmcnt exact) from a safe address and the affected 16-byte pieces are overwritten with zeros after the
// wait, before the barrier that publishes the tile.

template <int TH, int TW>
__global__ void __launch_bounds__(256, 1) conv_wgrad_ring_kernel(WgradArgs a) {
  using G = WgradGeom<3, 3, 1, 3, 2, 2, 1, TH, TW, 2>;
  constexpr int NSTAGE = 3;
  constexpr int PIECES = G::DY_PIECES + G::IN_PIECES;
  constexpr int NP = (PIECES + 255) / 256;  // LDS-DMA instructions per wave per tile
  constexpr int STAGE_BYTES = NP * 256 * 16;
  static_assert(G::DY_PIECES % 256 == 0, "dy pieces must fill whole instructions");
  static_assert(NSTAGE * STAGE_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __align__(16) unsigned char smem[NSTAGE * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wco = wave >> 1, wci = wave & 1;
  int tb = blockIdx.x;
  const int cib = tb % a.ncib;
  const int cob = tb / a.ncib;
  const int split = blockIdx.y;
  const int co0 = cob * 64, ci0 = cib * 64;

  ffa_f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const unsigned char* x_b = static_cast<const unsigned char*>(a.x);
  const unsigned char* dy_b = static_cast<const unsigned char*>(a.dy);
  const int li = lane & 15, gsel = (lane >> 4) & 1, khalf = lane >> 5;
  const int ntiles = (a.npt - split + a.nsplit - 1) / a.nsplit;  // tiles split, split + nsplit, ...

  unsigned okmask[NSTAGE];

  // Per-thread piece geometry is tile independent: the byte offset of every piece relative to the tile
  // origin and four "edge" bit sets (piece lies in the halo's first / last row / column) are computed once.
  // The host only selects this kernel when the tiles divide the image and pad == 1, so a piece can be outside
  // the image only through those edges: per tile the invalid set is a handful of scalar selects, and every
  // DMA address is (uniform 32-bit tile offset + per-lane constant) -- no per-piece bounds arithmetic.
  int prel[NP];
  unsigned pstatic_bad = 0, e_top = 0, e_bot = 0, e_left = 0, e_right = 0;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int i = tid + k * 256;
    if (k * 256 < G::DY_PIECES) {
      const int part = i % G::PARTS;
      const int n = (i / G::PARTS) % G::NPX;
      const int plane = i / (G::PARTS * G::NPX);
      const int c = co0 + plane * 32 + part * 8;
      prel[k] = (((n / TW) * a.Wo + (n % TW)) * a.Co + c) * 2;
      pstatic_bad |= (c < a.Co ? 0u : 1u) << k;
    } else {
      const int jj = i - G::DY_PIECES;
      const int part = jj % G::PARTS;
      const int q = (jj / G::PARTS) % (G::IH * G::IW);
      const int plane = jj / (G::PARTS * G::IH * G::IW);
      const int c = ci0 + plane * 32 + part * 8;
      const int r = q / G::IW, cc = q % G::IW;
      prel[k] = ((r * a.Wi + cc) * a.Ci + c) * 2;
      pstatic_bad |= ((jj < G::IN_PIECES && c < a.Ci) ? 0u : 1u) << k;
      e_top |= (r == 0 ? 1u : 0u) << k;
      e_bot |= (r == G::IH - 1 ? 1u : 0u) << k;
      e_left |= (cc == 0 ? 1u : 0u) << k;
      e_right |= (cc == G::IW - 1 ? 1u : 0u) << k;
    }
  }

  int voff[NP];  // byte offsets (from dy / x base) of the tile being issued
  auto prepare = [&](int t, int stage) {
    const int pt = split + t * a.nsplit;
    const int tx = pt % a.tiles_x;
    const int t2 = pt / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    const int b = t2 / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int dy_org = (((b * a.Ho + oy0) * a.Wo + ox0) * a.Co) * 2;
    const int in_org = (((b * a.Hi + oy0 - 1) * a.Wi + ox0 - 1) * a.Ci) * 2;
    unsigned bad = pstatic_bad;
    if (oy0 == 0) bad |= e_top;
    if (oy0 + TH == a.Ho) bad |= e_bot;
    if (ox0 == 0) bad |= e_left;
    if (ox0 + TW == a.Wo) bad |= e_right;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int off = (k * 256 < G::DY_PIECES ? dy_org : in_org) + prel[k];
      voff[k] = ((bad >> k) & 1u) ? 0 : off;
    }
    okmask[stage] = bad;
  };
  auto emit = [&](int k, int stage) {
    const unsigned char* base = (k * 256 < G::DY_PIECES) ? dy_b : x_b;
    const unsigned lbase =
        (unsigned)__builtin_amdgcn_readfirstlane((int)(stage * STAGE_BYTES + (wave * 64 + k * 256) * 16));
    // inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in the kernel hipcc (ROCm 7.2) drains
    // lgkmcnt to 0 in front of every MFMA step, i.e. waits for the transpose reads it has just issued for the NEXT
    // step (found on conv3x3_ring_kernel); the DMA's completion is waited for by hand below anyway
    {
      const unsigned char* src_ = base + (unsigned)voff[k];
      const unsigned dst_ = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + lbase);
      unsigned keep_;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
          : "=&s"(keep_)
          : "v"(src_), "s"(dst_)
          : "memory");
    }
  };
  auto issue_all = [&](int t, int stage) {
    prepare(t, stage);
#pragma unroll
    for (int k = 0; k < NP; ++k) emit(k, stage);
  };

  if (ntiles > 0) issue_all(0, 0);
  if (ntiles > 1) issue_all(1, 1);

#pragma unroll 1
  for (int t = 0; t < ntiles; ++t) {
    const int stage = t % NSTAGE;
    // this wave's DMA of tile t has landed when only the newer tile's NP instructions remain outstanding
    if (t + 1 < ntiles) {
      if (NP == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      else if (NP == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    {
      unsigned mask = stage == 0 ? okmask[0] : (stage == 1 ? okmask[1] : okmask[2]);
      if (mask) {
#pragma unroll
        for (int k = 0; k < NP; ++k)
          if (mask & (1u << k))
            *reinterpret_cast<ffa_u32x4*>(smem + stage * STAGE_BYTES + (size_t)(tid + k * 256) * 16) =
                ffa_u32x4{0u, 0u, 0u, 0u};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = (t + 2 < ntiles);
    const int nstage = (t + 2) % NSTAGE;
    if (more) prepare(t + 2, nstage);  // the DMA instructions themselves are spread over the k-steps below

    const unsigned char* sDy = smem + stage * STAGE_BYTES;
    const unsigned char* sIn = sDy + G::DY_BYTES;
    const unsigned char* dyPlane = sDy + wco * (G::NPX * G::ROWB) + gsel * 32 + (li & 3) * 8;
    const unsigned char* inPlane = sIn + wci * (G::IH * G::IW * G::ROWB) + gsel * 32 + (li & 3) * 8;
    const int lrow = 8 * khalf + (li >> 2);  // pixel of the k-step this lane addresses

    ffa_u32x4 fa[2];
    ffa_u32x4 fb[2][9];
    auto tr2 = [&](const unsigned char* p0) {
      const ffa_s16x4 v0 = lds_read_tr16(p0);
      const ffa_s16x4 v1 = lds_read_tr16(p0 + 4 * G::ROWB);
      ffa_u32x4 f;
      f.x = __builtin_bit_cast(ffa_u32x2, v0).x;
      f.y = __builtin_bit_cast(ffa_u32x2, v0).y;
      f.z = __builtin_bit_cast(ffa_u32x2, v1).x;
      f.w = __builtin_bit_cast(ffa_u32x2, v1).y;
      return f;
    };
#define FFA_RING_LOAD(ks_, buf_)                                                                         \
  {                                                                                                      \
    constexpr int n0_ = (ks_) * 16;                                                                      \
    constexpr int py_ = n0_ / TW, px0_ = n0_ % TW;                                                       \
    fa[buf_] = tr2(dyPlane + (n0_ + lrow) * G::ROWB);                                                    \
    const unsigned char* bb_ = inPlane + (py_ * G::IW + px0_ + lrow) * G::ROWB;                          \
    _Pragma("unroll") for (int tap = 0; tap < 9; ++tap)                                                  \
        fb[buf_][tap] = tr2(bb_ + ((tap / 3) * G::IW + (tap % 3)) * G::ROWB);                            \
  }
#define FFA_RING_MMA(buf_)                                                                                     \
  _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(      \
      __builtin_bit_cast(ffa_bf16x8, fa[buf_]), __builtin_bit_cast(ffa_bf16x8, fb[buf_][tap]), acc[tap], 0, 0, 0);
#define FFA_RING_STEP(ks_)                                                              \
  {                                                                                     \
    __builtin_amdgcn_sched_barrier(0); /* keep each step's reads inside its own region */ \
    if ((ks_) + 1 < G::KSTEPS) FFA_RING_LOAD(((ks_) + 1 < G::KSTEPS ? (ks_) + 1 : 0), ((ks_) + 1) & 1) \
    FFA_RING_MMA((ks_) & 1)                                                             \
    if (more) {                                                                         \
      _Pragma("unroll") for (int k_ = (ks_); k_ < NP; k_ += G::KSTEPS) emit(k_, nstage); \
    }                                                                                   \
    if ((ks_) + 1 < G::KSTEPS) {                                                        \
      _Pragma("unroll") for (int q_ = 0; q_ < 9; ++q_) {                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                              \
      }                                                                                 \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                \
    }                                                                                   \
  }
    FFA_RING_LOAD(0, 0)
    FFA_RING_STEP(0)
    FFA_RING_STEP(1)
    FFA_RING_STEP(2)
    FFA_RING_STEP(3)
    FFA_RING_STEP(4)
    FFA_RING_STEP(5)
    FFA_RING_STEP(6)
    FFA_RING_STEP(7)
    static_assert(G::KSTEPS == 8, "ring kernel is written for 128-pixel tiles");
#undef FFA_RING_LOAD
#undef FFA_RING_MMA
#undef FFA_RING_STEP
  }

  const int ci = ci0 + wci * 32 + (lane & 31);
  const int co_w = co0 + wco * 32;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_w + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      a.slabs[(((size_t)split * a.CoT + co) * 9 + t) * a.CiT + ci] = acc[t][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stem (7x7 stride 2, <= 16 stored input channels) weight gradient, bf16.  The generic kernel pads the 5 input
// channels to a 32-channel plane (84 % of the MFMA columns wasted) and walks the 7 kernel rows as 7 separate
// passes over the halo (5.9 GB staged per step, 0.95 ms).  Here one block stages the dy tile and the FULL
// 21 x 37 halo once per 8 x 16 output tile and its eight waves split the work as 2 (32-channel co tiles) x 4 (kernel
// row pairs); the 32 MFMA columns carry TWO taps x 16 channels: the transpose read takes a per-lane row address, so
// the upper 16 lanes simply address the halo one pixel to the right.  Per (row, tap pair) one accumulator tile:
// 2 rows x 4 pairs = 8 tiles = 128 accumulator registers per wave.

template <int DUMMY>
__global__ void __launch_bounds__(512, 2) stem_wgrad_kernel(WgradArgs a) {
  constexpr int TH = 8, TW = 16, NPX = TH * TW, KS = NPX / 16;
  constexpr int IH = (TH - 1) * 2 + 7, IW = (TW - 1) * 2 + 7;  // 21 x 37
  constexpr int DY_ROWB = 64, IN_ROWB = 32;
  constexpr int DY_BYTES = 2 * NPX * DY_ROWB;   // [2 co planes][128 px][32 co]
  constexpr int IN_BYTES = IH * IW * IN_ROWB;   // [777 px][16 ch]
  constexpr int DY_PIECES = DY_BYTES / 16, IN_PIECES = IN_BYTES / 16;
  constexpr int NDP = (DY_PIECES + 511) / 512, NIP = (IN_PIECES + 511) / 512;
  __shared__ __align__(16) unsigned char smem[DY_BYTES + IN_BYTES];
  unsigned char* sDy = smem;
  unsigned char* sIn = smem + DY_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave & 1, wr = wave >> 1;        // rows 2*wr, 2*wr+1 (wr == 3: row 6 only)
  const int nrows = (wr == 3) ? 1 : 2;
  const int cob = blockIdx.x, split = blockIdx.y;
  const int co0 = cob * 64;
  const int li = lane & 15, gsel = (lane >> 4) & 1, khalf = lane >> 5;

  ffa_f32x16 acc[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][p][e] = 0.f;

  const unsigned char* x_b = static_cast<const unsigned char*>(a.x);
  const unsigned char* dy_b = static_cast<const unsigned char*>(a.dy);
  ffa_u32x4 dreg[NDP], ireg[NIP];

  auto load_tile = [&](int pt) {
    const int tx = pt % a.tiles_x;
    const int t2 = pt / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    const int b = t2 / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * 2 - a.pad, ix0 = ox0 * 2 - a.pad;
#pragma unroll
    for (int k = 0; k < NDP; ++k) {
      const int i = tid + k * 512;
      const int part = i & 3, n = (i >> 2) % NPX, plane = i / (4 * NPX);
      const int oy = oy0 + n / TW, ox = ox0 + n % TW;
      const int c = co0 + plane * 32 + part * 8;
      ffa_u32x4 v = ffa_u32x4{0u, 0u, 0u, 0u};
      if (i < DY_PIECES && oy < a.Ho && ox < a.Wo && c < a.Co)
        v = *reinterpret_cast<const ffa_u32x4*>(dy_b + (((size_t)(b * a.Ho + oy) * a.Wo + ox) * a.Co + c) * 2);
      dreg[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NIP; ++k) {
      const int i = tid + k * 512;
      const int part = i & 1, q = i >> 1;
      const int vy = iy0 + q / IW, vx = ix0 + q % IW;
      ffa_u32x4 v = ffa_u32x4{0u, 0u, 0u, 0u};
      if (i < IN_PIECES && vy >= 0 && vx >= 0 && vy < a.Hi && vx < a.Wi)
        v = *reinterpret_cast<const ffa_u32x4*>(x_b + (((size_t)(b * a.Hi + vy) * a.Wi + vx) * a.Ci + part * 8) * 2);
      ireg[k] = v;
    }
  };
  auto tr2 = [&](const unsigned char* p0, int step) {
    const ffa_s16x4 v0 = lds_read_tr16(p0);
    const ffa_s16x4 v1 = lds_read_tr16(p0 + step);
    ffa_u32x4 f;
    f.x = __builtin_bit_cast(ffa_u32x2, v0).x;
    f.y = __builtin_bit_cast(ffa_u32x2, v0).y;
    f.z = __builtin_bit_cast(ffa_u32x2, v1).x;
    f.w = __builtin_bit_cast(ffa_u32x2, v1).y;
    return f;
  };

  int pt = split;
  if (pt < a.npt) load_tile(pt);
  for (; pt < a.npt; pt += a.nsplit) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NDP; ++k) {
      const int i = tid + k * 512;
      if (i < DY_PIECES) *reinterpret_cast<ffa_u32x4*>(sDy + (size_t)i * 16) = dreg[k];
    }
#pragma unroll
    for (int k = 0; k < NIP; ++k) {
      const int i = tid + k * 512;
      if (i < IN_PIECES) *reinterpret_cast<ffa_u32x4*>(sIn + (size_t)i * 16) = ireg[k];
    }
    __syncthreads();
    if (pt + a.nsplit < a.npt) load_tile(pt + a.nsplit);

    const unsigned char* aBase = sDy + wco * (NPX * DY_ROWB) + gsel * 32 + (li & 3) * 8;
    // the upper 16 lanes (gsel = 1) carry the odd tap of the pair: one halo pixel to the right
    const unsigned char* bBase = sIn + gsel * IN_ROWB + (li & 3) * 8;
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
      const int n0 = ks * 16;
      const int py = n0 / TW, px0 = n0 % TW;
      const int prow = 8 * khalf + (li >> 2);  // output pixel of the k-step this lane addresses
      const ffa_u32x4 af = tr2(aBase + (n0 + prow) * DY_ROWB, 4 * DY_ROWB);
      const unsigned char* bk = bBase + ((py * 2) * IW + (px0 + prow) * 2) * IN_ROWB;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (r < nrows) {
          const int kr = 2 * wr + r;
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const ffa_u32x4 bf = tr2(bk + (kr * IW + 2 * p) * IN_ROWB, 4 * 2 * IN_ROWB);
            acc[r][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ffa_bf16x8, af),
                                                                __builtin_bit_cast(ffa_bf16x8, bf), acc[r][p], 0, 0, 0);
          }
        }
      }
    }
  }

  // slab [split][CoT][49][16]: column = lane & 31 -> tap 2p + (column >> 4), channel column & 15
  const int col = lane & 31;
  const int ci = col & 15, sodd = col >> 4;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r >= nrows) continue;
    const int kr = 2 * wr + r;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int s = 2 * p + sodd;
      if (s >= 7) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        a.slabs[(((size_t)split * a.CoT + co) * 49 + kr * 7 + s) * 16 + ci] = acc[r][p][e];
      }
    }
  }
}

// dw is OIHW [Co][Ci][taps].  256 threads = SL split lanes x 256/SL quads of four consecutive ci: a lane's
// slab reads are 16-B loads coalesced along ci, lane l adds splits l, l+SL, ... (four loads in flight, added in
// index order) and lane 0 then adds the SL lane sums in order -> a fixed summation order for a given
// (nsplit, SL), and no thread walks hundreds of dependent L2 round trips.  SL = 4 serves the few-slab layers
// (many channels), SL = 32 the 256-slab ones (few channels).
template <int SL>
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int nsplit, int CoT, int CiT, int Co,
                    int Ci, int taps, int accumulate) {
  constexpr int NQ = 256 / SL;
  __shared__ float sh[SL][NQ][5];
  const int ql = threadIdx.x % NQ, sl = threadIdx.x / NQ;
  const int Cq = (Ci + 3) / 4;
  const long long total = (long long)Co * taps * Cq;
  const size_t slab = (size_t)CoT * taps * CiT;
  for (long long base = (long long)blockIdx.x * NQ; base < total; base += (long long)gridDim.x * NQ) {
    const long long i = base + ql;
    const bool valid = i < total;
    int ci = 0, tap = 0, co = 0;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      ci = (int)(i % Cq) * 4;
      const long long t2 = i / Cq;
      tap = (int)(t2 % taps);
      co = (int)(t2 / taps);
      const float* src = slabs + ((size_t)co * taps + tap) * CiT + ci;
      for (int k0 = sl; k0 < nsplit; k0 += 4 * SL) {
        ffa_f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * SL;
          const ffa_f32x4 t = *reinterpret_cast<const ffa_f32x4*>(src + (size_t)(k < nsplit ? k : k0) * slab);
          v[u] = k < nsplit ? t : ffa_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] += v[u][e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[sl][ql][e] = s[e];
    __syncthreads();
    if (sl == 0 && valid) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < SL; ++l) t += sh[l][ql][e];
        if (ci + e < Ci) {
          float* dst = dw + ((size_t)co * Ci + ci + e) * taps + tap;
          *dst = accumulate ? (*dst + t) : t;
        }
      }
    }
    __syncthreads();
  }
}

// 3x3 variant: one block = one output channel x NQ quads of input channels, every thread keeps all nine taps of
// its quad (36 sums).  The OIHW gradient [co][ci][tap] is then written as ONE contiguous run of NQ * 36 floats per
// block; wgrad_reduce_kernel's thread-per-(tap, quad) mapping scatters 4-byte stores 36 bytes apart and touches
// every 128-byte line of dw from nine different blocks (56 us for the 768 -> 256 decoder layer, 0.6 TB/s).
// Summation order per element is the same as wgrad_reduce_kernel's (lane l adds slabs l, l + SL, ... ascending,
// then the SL lane sums ascending), so the two kernels agree bit for bit.
template <int SL>
__global__ void __launch_bounds__(256)
wgrad_reduce3x3_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int nsplit, int CoT, int CiT, int Co,
                       int Ci, int accumulate) {
  constexpr int NQ = 256 / SL;
  constexpr int TAPS = 9;
  constexpr int ROW = TAPS * 4 + 1;  // 37 floats per (lane, quad): odd -> conflict-free column walks
  __shared__ float sh[SL * NQ * ROW];
  const int ql = threadIdx.x % NQ, sl = threadIdx.x / NQ;
  const int Cq = (Ci + 3) / 4;
  const int qblocks = (Cq + NQ - 1) / NQ;
  const int co = blockIdx.x / qblocks;
  const int q0 = (blockIdx.x % qblocks) * NQ;
  const int q = q0 + ql;
  const size_t slab = (size_t)CoT * TAPS * CiT;
  float acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
  if (q < Cq) {
    const float* src = slabs + (size_t)co * TAPS * CiT + q * 4;
    for (int k = sl; k < nsplit; k += SL) {
      ffa_f32x4 v[TAPS];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) v[t] = *reinterpret_cast<const ffa_f32x4*>(src + (size_t)k * slab + t * CiT);
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] += v[t][e];
    }
  }
  float* mine = sh + (sl * NQ + ql) * ROW;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) mine[t * 4 + e] = acc[t][e];
  __syncthreads();
  // output run of this block: channels q0 * 4 .. of `co`, [ci][tap] contiguous in dw
  const int ci_lo = q0 * 4;
  const int nci = min(NQ * 4, Ci - ci_lo);
  float* dst = dw + ((size_t)co * Ci + ci_lo) * TAPS;
  for (int o = threadIdx.x; o < nci * TAPS; o += 256) {
    const int cl = o / TAPS, tap = o % TAPS;
    const float* col = sh + (cl >> 2) * ROW + tap * 4 + (cl & 3);
    float t = 0.f;
#pragma unroll 8
    for (int l = 0; l < SL; ++l) t += col[l * NQ * ROW];
    dst[o] = accumulate ? (dst[o] + t) : t;
  }
}

// ------------------------------------------------------------------------------------------------
// Thin layers (bf16, 3x3 stride 1 pad 1, <= 32 stored channels on BOTH sides: the U-Net decoder tail and head; round 3).
// conv_wgrad_kernel pads their 16-channel operands to 32 x 32 MFMA tiles (a 16 x 16 layer issues four times its
// matrix work: 18 MFMA cycles per pixel and SIMD, which is what the HBM rate allows -- the kernel sat at 55-60 % of its
// byte floor).  Here:
//   * v_mfma_f32_16x16x32_bf16: exact 16-channel tiles, K = 32 pixels = one row of the 8 x 32 tile; every wave owns
//     ALL (co, ci, tap) tiles of the layer (36 ... 144 accumulator registers) and two rows of each spatial tile -- no
//     channel split, no operand re-read;
//   * both operands keep their natural [pixel][channels] form in LDS and the K-contiguous fragments come out of
//     ds_read_b64_tr_b16 (two reads of 4 pixels x 16 channels per fragment; the nine tap shifts are address offsets);
//   * x halo + dy tile arrive by LDS-DMA (zero page for the padding / ragged edges, nearest-x2 source for the
//     decoder's upsampled input), one or two tiles ahead of the tile being multiplied; no stores in the loop, so the
//     vmcnt arithmetic is exact;
//   * the four waves are summed through LDS in a fixed order and ONE f32 slab per block goes to the split-K workspace
//     (wgrad_reduce_kernel finishes: deterministic, no float atomics).

__device__ __attribute__((aligned(16))) const unsigned int ffa_wgthin_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void wgthin_dma16(const unsigned char* src, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wgthin_wait_and_meet() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(N) : "memory");
}

template <int CI, int CO>
struct WgThinGeom {
  static constexpr int TH = 8, TW = 32;
  static constexpr int IH = TH + 2, IW = TW + 2;
  static constexpr int XB = CI * 2, YB = CO * 2;  // bytes per pixel
  static constexpr int XP = IH * IW * (XB / 16);  // 16-byte pieces of the x halo
  static constexpr int YP = TH * TW * (YB / 16);  // ... of the dy tile (follows the halo in the slot)
  static constexpr int PIECES = XP + YP;
  static constexpr int NHW = (PIECES + 255) / 256;
  static constexpr int X_BYTES = XP * 16;
  static constexpr int SLOT = PIECES * 16;
  static constexpr int NSLOT = (3 * SLOT <= 78 * 1024) ? 3 : 2;
  static constexpr int MT = CO / 16, NT = CI / 16;
  static constexpr int ACC_BYTES = MT * NT * 9 * 4 * 64 * 4;  // one wave's accumulators
  static constexpr int LDS_BYTES = (NSLOT * SLOT > 2 * ACC_BYTES) ? NSLOT * SLOT : 2 * ACC_BYTES;
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two blocks per CU");
};

template <int CI, int CO, bool UP, bool PRO = false>
__global__ void __launch_bounds__(256, 2) conv3x3_thin_wgrad_kernel(WgradArgs a) {
  using G = WgThinGeom<CI, CO>;
  constexpr int MT = G::MT, NT = G::NT;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;  // fragment row (co) / column (ci); address role: pixel li >> 2, 8-byte segment li & 3
  const int kg = lane >> 4;  // pixels 8*kg .. 8*kg+7 of a 32-pixel k-step

  // ---- pieces of this thread: p = tid + k * 256; p < XP: halo pixel p / (XB/16) of x, else pixel of the dy tile ----
  int pinfo[G::NHW];  // x: (hy << 8 | hx) << 4 | slot;  dy: 1 << 30 | (ry << 8 | rx) << 4 | slot;  -1: no piece
#pragma unroll
  for (int k = 0; k < G::NHW; ++k) {
    const int p = tid + k * 256;
    if (p < G::XP) {
      const int q = p / (G::XB / 16), sl = p % (G::XB / 16);
      pinfo[k] = ((((q / G::IW) << 8) | (q % G::IW)) << 4) | sl;
    } else if (p < G::PIECES) {
      const int j = p - G::XP;
      const int q = j / (G::YB / 16), sl = j % (G::YB / 16);
      pinfo[k] = (1 << 30) | ((((q / G::TW) << 8) | (q % G::TW)) << 4) | sl;
    } else {
      pinfo[k] = -1;
    }
  }
  const unsigned char* x_b = static_cast<const unsigned char*>(a.x);
  const unsigned char* dy_b = static_cast<const unsigned char*>(a.dy);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(ffa_wgthin_zero16);
  const int Hs = UP ? (a.Hi >> 1) : a.Hi, Ws = UP ? (a.Wi >> 1) : a.Wi;
  const bool has_tail = (G::PIECES % 256 == 0) || (wave * 64 + (G::NHW - 1) * 256 < G::PIECES);

  auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
    const int tx = t % a.tiles_x;
    const int t2 = t / a.tiles_x;
    oy0 = (t2 % a.tiles_y) * G::TH;
    b = t2 / a.tiles_y;
    ox0 = tx * G::TW;
  };
  auto issue_tile = [&](int t, int slot) {
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
#pragma unroll
    for (int k = 0; k < G::NHW; ++k) {
      const int info = pinfo[k];
      const int yy = (info >> 12) & 0xff, xx = (info >> 4) & 0xff, sl = info & 15;
      const unsigned char* src = zero;
      if (info >= 0) {
        if (info & (1 << 30)) {
          const int oy = oy0 + yy, ox = ox0 + xx;
          if (oy < a.Ho && ox < a.Wo) src = dy_b + ((size_t)((b * a.Ho + oy) * a.Wo + ox) * G::YB + sl * 16);
        } else {
          const int vy = oy0 - 1 + yy, vx = ox0 - 1 + xx;
          if (vy >= 0 && vx >= 0 && vy < a.Hi && vx < a.Wi) {
            const int sy = UP ? (vy >> 1) : vy, sx = UP ? (vx >> 1) : vx;
            src = x_b + ((size_t)((b * Hs + sy) * Ws + sx) * G::XB + sl * 16);
          }
        }
      }
      const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(
          smem + slot * G::SLOT + (wave * 64 + k * 256) * 16);
      if (k + 1 < G::NHW || G::PIECES % 256 == 0) {
        wgthin_dma16(src, dst);
      } else if (has_tail) {
        if (info >= 0) wgthin_dma16(src, dst);
      }
    }
  };
  // PRO: rewrite this thread's own x pieces of the landed tile as relu(x * sc + sh) (thin forward kernel's recipe:
  // own pieces need only the wave's own vmcnt wait; the padding stays zero), then the block meets
  auto fix_tile = [&](int t, int slot) {
    if constexpr (PRO) {
      int b, oy0, ox0;
      tile_origin(t, b, oy0, ox0);
      float sc[8], sh[8];
      const int c0 = (tid % (G::XB / 16)) * 8;
      ffa_load8<float>(a.pro_sc + c0, sc);
      ffa_load8<float>(a.pro_sh + c0, sh);
#pragma unroll
      for (int k = 0; k < G::NHW; ++k) {
        const int info = pinfo[k];
        const int yy = (info >> 12) & 0xff, xx = (info >> 4) & 0xff;
        const int vy = oy0 - 1 + yy, vx = ox0 - 1 + xx;
        if (info >= 0 && !(info & (1 << 30)) && vy >= 0 && vx >= 0 && vy < a.Hi && vx < a.Wi) {
          ffa_u32x4* ptr = reinterpret_cast<ffa_u32x4*>(smem + slot * G::SLOT + (tid + k * 256) * 16);
          ffa_u32x4 v = *ptr;
          float f[8];
          f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
          f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
          f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
          f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaxf(__builtin_fmaf(f[e], sc[e], sh[e]), 0.f);
          v.x = ffa_pack_bf16x2(f[0], f[1]);
          v.y = ffa_pack_bf16x2(f[2], f[3]);
          v.z = ffa_pack_bf16x2(f[4], f[5]);
          v.w = ffa_pack_bf16x2(f[6], f[7]);
          *ptr = v;
        }
      }
    }
  };
  // wait until this wave's fill of the CURRENT tile has landed: with a three-slot ring one younger fill may stay in
  // flight, with two slots nothing younger exists at this point
  auto wait_tile = [&](int t, int slot) {
    if constexpr (PRO) {
      if constexpr (G::NSLOT == 3) {
        if (has_tail) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G::NHW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G::NHW - 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      fix_tile(t, slot);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if constexpr (G::NSLOT == 3) {
      if (has_tail) wgthin_wait_and_meet<G::NHW>();
      else wgthin_wait_and_meet<G::NHW - 1>();
    } else {
      wgthin_wait_and_meet<0>();
    }
  };

  ffa_f32x4 acc[MT][NT][9];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc[mt][nt][tp] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane fragment bases: pixel 8*kg + (li >> 2) of a k-step, 8-byte segment li & 3 of a 16-channel group; the
  // wave's first tile row is 2 * wave (two rows per wave)
  const int px0 = 8 * kg + (li >> 2);
  const int aBase = G::X_BYTES + ((2 * wave) * G::TW + px0) * G::YB + (li & 3) * 8;
  const int bBase = ((2 * wave) * G::IW + px0) * G::XB + (li & 3) * 8;

  auto tr2 = [&](const unsigned char* p0, int step) {
    const ffa_s16x4 v0 = lds_read_tr16(p0);
    const ffa_s16x4 v1 = lds_read_tr16(p0 + step);
    ffa_u32x4 f;
    f.x = __builtin_bit_cast(ffa_u32x2, v0).x;
    f.y = __builtin_bit_cast(ffa_u32x2, v0).y;
    f.z = __builtin_bit_cast(ffa_u32x2, v1).x;
    f.w = __builtin_bit_cast(ffa_u32x2, v1).y;
    return f;
  };

  int t = blockIdx.x;
  const int stride = gridDim.x;
  if (t < a.npt) {
    issue_tile(t, 0);
    if constexpr (G::NSLOT == 3) issue_tile(t + stride < a.npt ? t + stride : t, 1);
    int slot = 0;
    for (; t < a.npt; t += stride) {
      wait_tile(t, slot);
      {
        const int tn = t + (G::NSLOT - 1) * stride;
        issue_tile(tn < a.npt ? tn : t, (slot + G::NSLOT - 1) % G::NSLOT);
      }
      const unsigned char* sS = smem + slot * G::SLOT;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {  // the wave's two rows = two k-steps of 32 pixels
        ffa_u32x4 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = tr2(sS + aBase + rr * G::TW * G::YB + mt * 32, 4 * G::YB);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const ffa_u32x4 bf = tr2(sS + bBase + ((rr + tp / 3) * G::IW + tp % 3) * G::XB + nt * 32, 4 * G::XB);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[mt][nt][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ffa_bf16x8, af[mt]),
                                                                       __builtin_bit_cast(ffa_bf16x8, bf), acc[mt][nt][tp],
                                                                       0, 0, 0);
          }
        }
      }
      slot = (slot + 1) % G::NSLOT;
    }
  }
  // every DMA (also the fills for tiles that do not exist) has landed before the slots become reduction scratch
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- sum the four waves in a fixed order: (0 + 2) and (1 + 3), then (0 + 1) ----
  float* red = reinterpret_cast<float*>(smem);
  constexpr int NACC = MT * NT * 9 * 4;  // floats per lane
  auto put = [&](int which) {
    float* dst = red + (size_t)which * NACC * 64 + lane;
    int o = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[(o++) * 64] = acc[mt][nt][tp][i];
  };
  auto add = [&](int which) {
    const float* src = red + (size_t)which * NACC * 64 + lane;
    int o = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[mt][nt][tp][i] += src[(o++) * 64];
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave != 0) return;
  add(0);

  // ---- the block's slab [CoT][9][CiT]: D[row = co][col = ci], lane (li = ci, rows 4*kg + i) ----
  float* slab = a.slabs + (size_t)blockIdx.x * a.CoT * 9 * a.CiT;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = mt * 16 + 4 * kg + i, ci = nt * 16 + li;
          slab[((size_t)co * 9 + tp) * a.CiT + ci] = acc[mt][nt][tp][i];
        }
}

// ------------------------------------------------------------------------------------------------
// 64 x 64-channel blocks of the bulk of the network (bf16, 3x3 stride 1 pad 1, Co and Ci multiples of 64; 8 x 32-pixel
// tiles, or 16 x 16 on maps narrower than 32; round 3): conv3x3_thin_wgrad_kernel's recipe -- v_mfma_f32_16x16x32_bf16 on transposed LDS reads, operands by
// LDS-DMA into a second slot while the first is multiplied, no LDS store phase, no staging registers -- at the channel
// decomposition of conv_wgrad_kernel<2, 2, 2>: eight waves = two k groups (rows 0-3 / 4-7 of the 8 x 32 tile) x (2 co x
// 2 ci) sub-tiles of 32 x 32 channels x 9 taps (144 accumulator registers), one block per CU, the k groups merged
// through LDS, one f32 slab per block, wgrad_reduce* unchanged.  conv_wgrad_kernel spent ~20 % of a tile storing its
// register-staged pieces to LDS with every wave of the CU off the matrix pipe.
// LDS images: [pixel][64 channels = 128 B]; the 16-byte slot index of a pixel is XORed with 2 * phi(column), phi = bit 1 |
// bit 3 << 1 of the pixel's column in its image: conflict-free ds_read_b64_tr_b16 for every tap shift (enumerated),
// applied on the DMA source address.

template <int TW_>
struct Wg64Geom {
  static constexpr int TW = TW_, TH = 256 / TW_, IH = TH + 2, IW = TW + 2;
  static constexpr int RS = 32 / TW;  // image rows per 32-pixel k-step: 1 (8 x 32 tiles) or 2 (16 x 16 tiles)
  static constexpr int XP = IH * IW * 8, YP = TH * TW * 8;  // 16-byte pieces
  static constexpr int PIECES = XP + YP;
  static constexpr int NHW = (PIECES + 511) / 512;
  static constexpr int X_BYTES = XP * 16;
  static constexpr int SLOT = PIECES * 16;
  static constexpr int ACC_BYTES = 144 * 64 * 4;  // one wave
  static constexpr int LDS_BYTES = (2 * SLOT > 4 * ACC_BYTES) ? 2 * SLOT : 4 * ACC_BYTES;
  static_assert(TW == 32 || TW == 16, "tile width");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ int wg64_phi(int px) { return ((px >> 1) & 1) | (((px >> 3) & 1) << 1); }

template <int TW>
__global__ void __launch_bounds__(512) conv3x3_wgrad64_kernel(WgradArgs a) {
  using G = Wg64Geom<TW>;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kgp = wave >> 2;             // k group: k-steps (32 pixels: one tile row, or two at TW = 16) 4 * kgp .. + 3
  const int wco = (wave >> 1) & 1, wci = wave & 1;
  const int li = lane & 15, kg = lane >> 4;
  const int cib = blockIdx.x % a.ncib, cob = blockIdx.x / a.ncib;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int split = blockIdx.y;

  // x = one tensor, or nearest_x2(lo) ++ skip (WgradArgs): a 64-channel block lies in ONE of the two sources (C1 is a
  // multiple of 64); from lo, halo pixel (yy, xx) of the tile at (oy0, ox0) -- both multiples of 8 -- is lo pixel
  // (oy0 / 2 - 1 + ((yy - 1) >> 1) + 1, ...): again a launch constant past a block-uniform origin
  const bool two = a.C1 > 0;
  const bool from_lo = two && ci0 < a.C1;
  const size_t xpb = (size_t)(!two ? a.Ci : from_lo ? a.C1 : a.Ci - a.C1) * 2, ypb = (size_t)a.Co * 2;  // bytes per pixel
  const int xW = from_lo ? a.Wi / 2 : a.Wi, xH = from_lo ? a.Hi / 2 : a.Hi;
  // per 16-byte piece of a slot (x halo, then the dy tile), fixed for the launch: offset of its source from the tile's
  // origin pixel in 16-byte units << 13 | dy piece << 12 | no piece << 11 | row << 6 | column; the swizzle is in the offset
  int pinfo[G::NHW];
#pragma unroll
  for (int k = 0; k < G::NHW; ++k) {
    const int p = tid + k * 512;
    if (p < G::XP) {
      const int q = p >> 3, yy = q / G::IW, xx = q % G::IW, sl = (p & 7) ^ (2 * wg64_phi(xx));
      const int py = from_lo ? ((yy - 1) >> 1) + 1 : yy, px = from_lo ? ((xx - 1) >> 1) + 1 : xx;
      pinfo[k] = ((((py * xW + px) * (int)xpb) >> 4) + sl) << 13 | (yy << 6) | xx;
    } else if (p < G::PIECES) {
      const int q = (p - G::XP) >> 3, yy = q / G::TW, xx = q % G::TW, sl = (p & 7) ^ (2 * wg64_phi(xx));
      pinfo[k] = ((((yy * a.Wo + xx) * (int)ypb) >> 4) + sl) << 13 | (1 << 12) | (yy << 6) | xx;
    } else {
      pinfo[k] = 1 << 11;
    }
  }
  const unsigned char* x_b = (two && !from_lo) ? static_cast<const unsigned char*>(a.x2) + (ci0 - a.C1) * 2
                                               : static_cast<const unsigned char*>(a.x) + ci0 * 2;
  const unsigned char* dy_b = static_cast<const unsigned char*>(a.dy) + co0 * 2;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(ffa_wgthin_zero16);
  const bool has_tail = (G::PIECES % 512 == 0) || (wave * 64 + (G::NHW - 1) * 512 < G::PIECES);

  auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
    const int tx = t % a.tiles_x;
    const int t2 = t / a.tiles_x;
    oy0 = (t2 % a.tiles_y) * G::TH;
    b = t2 / a.tiles_y;
    ox0 = tx * G::TW;
  };
  // a piece's source = tile origin (block-uniform, 64-bit) + a per-piece 32-bit offset fixed for the launch; validity is
  // two unsigned compares: ~10 VALU per piece, no branches (the straightforward form cost ~40 instructions a piece
  // = a quarter of a tile's matrix time with every wave of the CU off the pipe)
  auto issue_tile = [&](int t, int slot) {
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
    const int xoy = from_lo ? (oy0 >> 1) - 1 : oy0 - 1, xox = from_lo ? (ox0 >> 1) - 1 : ox0 - 1;
    const unsigned char* xt = x_b + ((long long)(b * xH + xoy) * xW + xox) * (long long)xpb;
    const unsigned char* yt = dy_b + ((long long)(b * a.Ho + oy0) * a.Wo + ox0) * (long long)ypb;
#pragma unroll
    for (int k = 0; k < G::NHW; ++k) {
      const int info = pinfo[k];
      const int yy = (info >> 6) & 31, xx = info & 63;
      // pieces 0 .. XP-1 are x: k < XP / 512 all x, k > XP / 512 all dy, one mixed k
      const bool mixed = (k == G::XP / 512) && (G::XP % 512 != 0);
      const bool is_dy = mixed ? (info & (1 << 12)) != 0 : (k * 512 >= G::XP);
      const int oy = yy + (is_dy ? oy0 : oy0 - 1), ox = xx + (is_dy ? ox0 : ox0 - 1);
      const bool valid = (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
      const unsigned char* src = valid ? (is_dy ? yt : xt) + (size_t)(((unsigned)info >> 13) << 4) : zero;
      const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(
          smem + slot * G::SLOT + (wave * 64 + k * 512) * 16);
      if (k + 1 < G::NHW || G::PIECES % 512 == 0) {
        wgthin_dma16(src, dst);
      } else if (has_tail) {
        if (!(info & (1 << 11))) wgthin_dma16(src, dst);
      }
      __builtin_amdgcn_sched_barrier(0);  // one piece's address registers at a time (144 accumulators are live)
    }
  };

  ffa_f32x4 acc[2][2][9];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc[mt][nt][tp] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane (li, kg) supplies pixel k = 8 * kg + (li >> 2) (+ 4 for the second read) of a 32-pixel
  // k-step, 8-byte segment li & 3 of a 16-channel tile; physical 16-byte slot = (2 * tile + (li >> 1 & 1)) ^ 2 * phi(pixel)
  const int kpx = 8 * kg + (li >> 2);
  const int seg = (li & 1) * 8, hbit = (li >> 1) & 1;
  // dy: a k-step is 32 consecutive pixels of the [TH][TW] image; the lane's column is kpx % TW (+ 4): bits 1 and 3 of it
  // do not depend on the k-step or the + 4
  const int yphi = wg64_phi(kpx % G::TW);
  int aoff[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    aoff[mt] = G::X_BYTES + (kgp * 128 + kpx) * 128 + (((2 * (wco * 2 + mt) + hbit) ^ (2 * yphi)) * 16) + seg;

  // x: the swizzle is a function of the halo COLUMN (a halo row is 34 (18) * 128 B = 17 (9) times all 64 banks: rows do
  // not move banks), so a tap's row is an immediate offset and the lane part is one of 3 (tap column) x 2 (tile) x 2 (read)
  // values; at TW = 16 the lane's pixel sits in row kpx / 16 of the k-step's two rows
  int boff[3][2][2];
#pragma unroll
  for (int sx = 0; sx < 3; ++sx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int hx = kpx % G::TW + sx + 4 * h;
        boff[sx][nt][h] = ((kpx / G::TW) * G::IW + hx) * 128 + (((2 * (wci * 2 + nt) + hbit) ^ (2 * wg64_phi(hx))) * 16) + seg;
      }
  auto tr2 = [&](const unsigned char* p0, const unsigned char* p1) {
    const ffa_s16x4 v0 = lds_read_tr16(p0);
    const ffa_s16x4 v1 = lds_read_tr16(p1);
    ffa_u32x4 f;
    f.x = __builtin_bit_cast(ffa_u32x2, v0).x;
    f.y = __builtin_bit_cast(ffa_u32x2, v0).y;
    f.z = __builtin_bit_cast(ffa_u32x2, v1).x;
    f.w = __builtin_bit_cast(ffa_u32x2, v1).y;
    return f;
  };

  const int ntl = (a.npt - split + a.nsplit - 1) / a.nsplit;  // tiles split, split + nsplit, ...
  if (ntl > 0) {
    issue_tile(split, 0);
    int slot = 0;
    for (int it = 0; it < ntl; ++it) {
      wgthin_wait_and_meet<0>();  // this tile's fill has landed (nothing younger is in flight); the other slot is free
      const bool more = it + 1 < ntl;
      if (more) issue_tile(split + (it + 1) * a.nsplit, slot ^ 1);
      const unsigned char* sS = smem + slot * G::SLOT;
      // the k group's four k-steps of 32 pixels (RS = 1 or 2 image rows each).  Tap (r, s) of k-step j reads the halo rows
      // from RS * j + r on, at column shift s: each (start row, shift) fragment is read ONCE and multiplied with the dy
      // fragments of every k-step it serves -- up to three at RS = 1: 88 instead of 160 LDS reads per wave and tile, which
      // were the kernel's bound (1.1 x the matrix pipe's time); two at RS = 2 (start rows 0 .. 8, the odd ones serve r = 1)
      ffa_u32x4 af[4][2];  // k-step j is read when its first start row comes up and dies 2 start rows later
      const unsigned char* sRow = sS + (G::RS * 4 * kgp) * (G::IW * 128);
#pragma unroll
      for (int hr = 0; hr < G::RS * 3 + 3; ++hr) {
        if (hr % G::RS == 0 && hr / G::RS < 4) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            af[hr / G::RS][mt] = tr2(sS + aoff[mt] + (hr / G::RS) * 32 * 128, sS + aoff[mt] + (hr / G::RS) * 32 * 128 + 4 * 128);
        }
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const ffa_u32x4 bf = tr2(sRow + hr * (G::IW * 128) + boff[sx][nt][0], sRow + hr * (G::IW * 128) + boff[sx][nt][1]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
              if (hr - r < 0 || (hr - r) % G::RS != 0 || (hr - r) / G::RS > 3) continue;
              const int rr = (hr - r) / G::RS;
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
                acc[mt][nt][r * 3 + sx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(ffa_bf16x8, af[rr][mt]), __builtin_bit_cast(ffa_bf16x8, bf), acc[mt][nt][r * 3 + sx], 0, 0,
                    0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      slot ^= 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- k group 1 -> LDS, k group 0 adds and writes the block's slab ----
  float* red = reinterpret_cast<float*>(smem) + (size_t)(wave & 3) * 144 * 64 + lane;
  if (kgp == 1) {
    int o = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
          for (int i = 0; i < 4; ++i) red[(o++) * 64] = acc[mt][nt][tp][i];
  }
  __syncthreads();
  if (kgp == 1) return;
  {
    int o = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[mt][nt][tp][i] += red[(o++) * 64];
  }
  float* slab = a.slabs + (size_t)split * a.CoT * 9 * a.CiT;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = co0 + wco * 32 + mt * 16 + 4 * kg + i, ci = ci0 + wci * 32 + nt * 16 + li;
          slab[((size_t)co * 9 + tp) * a.CiT + ci] = acc[mt][nt][tp][i];
        }
}

// ------------------------------------------------------------------------------------------------
// Stem weight gradient, second form (round 3; <= 8 real input channels; FFA_STEM_WGRAD8=0 restores stem_wgrad_kernel):
// conv7x7_stem_kernel's observation applied to dW = dy^T x -- only the first 8 channels (16 bytes) of an input pixel are
// staged (LDS-DMA gathers one piece per pixel), so the 16 columns of a v_mfma_f32_16x16x32_bf16 tile are TWO adjacent taps
// x 8 channels and, the convolution having stride 2, the 32-byte row of output pixel px and tap pair sp is simply the
// halo bytes of input pixels 2 px + 2 sp, + 1: half the matrix work of the two-taps-x-16-channels form.  Eight waves =
// two k groups (tile rows 0-3 / 4-7) x four tap pairs; a wave keeps all 64 output channels (4 tiles) x 7 kernel rows = 112
// accumulator registers and reads every (halo row, pair) fragment ONCE for the up to four tile rows it serves (halo row
// 2 j + r: rows j, j + 1, ... meet it at r, r - 2, ...).  dy tile and halo by LDS-DMA into a second slot while the first is
// multiplied; k groups merged through LDS; slab layout [split][Co][49][16] as stem_wgrad_kernel's (same reduce).

struct WgStem8Geom {
  static constexpr int TH = 8, TW = 32, IH = 2 * TH + 5, IWC = 72;  // 21 halo rows of 69 (padded to 72) 16-byte pixels
  static constexpr int XP = IH * IWC, YP = TH * TW * 8;             // 16-byte pieces
  static constexpr int PIECES = XP + YP;
  static constexpr int NHW = (PIECES + 511) / 512;
  static constexpr int X_BYTES = XP * 16;
  static constexpr int SLOT = PIECES * 16;
  static constexpr int ACC_BYTES = 112 * 64 * 4;  // one wave
  static constexpr int LDS_BYTES = (2 * SLOT > 4 * ACC_BYTES) ? 2 * SLOT : 4 * ACC_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__global__ void __launch_bounds__(512) stem_wgrad8_kernel(WgradArgs a) {
  using G = WgStem8Geom;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kgp = wave >> 2;  // k group: tile rows 4 * kgp .. + 3
  const int sp = wave & 3;    // tap pair: taps 2 * sp, 2 * sp + 1 of every kernel row
  const int li = lane & 15, kg = lane >> 4;
  const int split = blockIdx.y;

  // per 16-byte piece of a slot (x halo, then the dy tile), fixed for the launch: source offset from the tile's origin pixel
  // in 16-byte units << 14 | dy piece << 13 | no piece << 12 | row << 7 | column
  int pinfo[G::NHW];
#pragma unroll
  for (int k = 0; k < G::NHW; ++k) {
    const int p = tid + k * 512;
    if (p < G::XP) {
      const int hy = p / G::IWC, hx = p % G::IWC;
      pinfo[k] = (hx < 2 * G::TW + 5) ? (((hy * a.Wi + hx) * 2) << 14 | (hy << 7) | hx) : (1 << 12);
    } else if (p < G::PIECES) {
      const int q = (p - G::XP) >> 3, yy = q / G::TW, xx = q % G::TW, sl = (p & 7) ^ (2 * wg64_phi(xx));
      pinfo[k] = ((yy * a.Wo + xx) * 8 + sl) << 14 | (1 << 13) | (yy << 7) | xx;
    } else {
      pinfo[k] = 1 << 12;
    }
  }
  const unsigned char* x_b = static_cast<const unsigned char*>(a.x);
  const unsigned char* dy_b = static_cast<const unsigned char*>(a.dy);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(ffa_wgthin_zero16);
  const bool has_tail = (G::PIECES % 512 == 0) || (wave * 64 + (G::NHW - 1) * 512 < G::PIECES);
  const int tiles_x = (a.Wo + G::TW - 1) / G::TW, tiles_y = (a.Ho + G::TH - 1) / G::TH;
  const int npt = a.B * tiles_x * tiles_y;

  auto issue_tile = [&](int t, int slot) {
    const int tx = t % tiles_x, t2 = t / tiles_x;
    const int oy0 = (t2 % tiles_y) * G::TH, b = t2 / tiles_y, ox0 = tx * G::TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const unsigned char* xt = x_b + ((long long)(b * a.Hi + iy0) * a.Wi + ix0) * 32;  // only dereferenced where valid
    const unsigned char* yt = dy_b + ((long long)(b * a.Ho + oy0) * a.Wo + ox0) * 128;
#pragma unroll
    for (int k = 0; k < G::NHW; ++k) {
      const int info = pinfo[k];
      const int yy = (info >> 7) & 31, xx = info & 127;
      const bool mixed = (k == G::XP / 512) && (G::XP % 512 != 0);
      const bool is_dy = mixed ? (info & (1 << 13)) != 0 : (k * 512 >= G::XP);
      const bool valid = !(info & (1 << 12)) &&
                         (is_dy ? (oy0 + yy < a.Ho && ox0 + xx < a.Wo)
                                : ((unsigned)(iy0 + yy) < (unsigned)a.Hi && (unsigned)(ix0 + xx) < (unsigned)a.Wi));
      const unsigned char* src = valid ? (is_dy ? yt : xt) + (size_t)(((unsigned)info >> 14) << 4) : zero;
      const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(
          smem + slot * G::SLOT + (wave * 64 + k * 512) * 16);
      if (k + 1 < G::NHW || G::PIECES % 512 == 0) {
        wgthin_dma16(src, dst);
      } else if (has_tail) {
        if (tid + k * 512 < G::PIECES) wgthin_dma16(src, dst);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  ffa_f32x4 acc[4][7];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 7; ++r) acc[mt][r] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};

  // lane (li, kg) supplies output pixel kpx = 8 * kg + (li >> 2) (+ 4 for the second read) of a 32-pixel k-step (one tile
  // row), 8-byte segment li & 3 of a 32-byte row: dy -> a 16-channel tile (slot pair 2 * mt, + 1, swizzled by the column),
  // x -> the two taps' 2 x 8 channels at input pixel 2 * kpx + 2 * sp
  const int kpx = 8 * kg + (li >> 2);
  const int seg = (li & 1) * 8, hbit = (li >> 1) & 1;
  const int yphi = wg64_phi(kpx);
  int aoff[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
    aoff[mt] = G::X_BYTES + (kgp * 128 + kpx) * 128 + (((2 * mt + hbit) ^ (2 * yphi)) * 16) + seg;
  const int boff = ((2 * 4 * kgp) * G::IWC + 2 * kpx + 2 * sp) * 16 + (li & 3) * 8;

  auto tr2 = [&](const unsigned char* p0, int step) {
    const ffa_s16x4 v0 = lds_read_tr16(p0);
    const ffa_s16x4 v1 = lds_read_tr16(p0 + step);
    ffa_u32x4 f;
    f.x = __builtin_bit_cast(ffa_u32x2, v0).x;
    f.y = __builtin_bit_cast(ffa_u32x2, v0).y;
    f.z = __builtin_bit_cast(ffa_u32x2, v1).x;
    f.w = __builtin_bit_cast(ffa_u32x2, v1).y;
    return f;
  };

  const int ntl = (npt - split + a.nsplit - 1) / a.nsplit;
  if (ntl > 0) {
    issue_tile(split, 0);
    int slot = 0;
    for (int it = 0; it < ntl; ++it) {
      wgthin_wait_and_meet<0>();
      if (it + 1 < ntl) issue_tile(split + (it + 1) * a.nsplit, slot ^ 1);
      const unsigned char* sS = smem + slot * G::SLOT;
      ffa_u32x4 af[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[j][mt] = tr2(sS + aoff[mt] + j * 32 * 128, 4 * 128);
#pragma unroll
      for (int h = 0; h < 13; ++h) {  // halo row 2 * (4 * kgp) + h serves tile row j at kernel row r = h - 2 j
        const ffa_u32x4 bf = tr2(sS + boff + h * (G::IWC * 16), 4 * 2 * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = h - 2 * j;
          if (r < 0 || r > 6) continue;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[mt][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ffa_bf16x8, af[j][mt]),
                                                                 __builtin_bit_cast(ffa_bf16x8, bf), acc[mt][r], 0, 0, 0);
        }
      }
      slot ^= 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- k group 1 -> LDS, k group 0 adds and writes the block's slab ----
  float* red = reinterpret_cast<float*>(smem) + (size_t)(wave & 3) * 112 * 64 + lane;
  if (kgp == 1) {
    int o = 0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(o++) * 64] = acc[mt][r][i];
  }
  __syncthreads();
  if (kgp == 1) return;
  {
    int o = 0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mt][r][i] += red[(o++) * 64];
  }
  // D[row = co][col]: column li -> tap 2 * sp + (li >> 3), channel li & 7; the eighth tap does not exist
  const int s = 2 * sp + (li >> 3), ch = li & 7;
  if (s < 7) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = mt * 16 + 4 * kg + i;
          a.slabs[(((size_t)split * a.CoT + co) * 49 + r * 7 + s) * 16 + ch] = acc[mt][r][i];
        }
  }
}

// ------------------------------------------------------------------------------------------------

struct WgradPlan {
  int wco, wci, wk, th, tw, rg, nsplit, ncob, ncib, CoT, CiT, npt, tiles_x, tiles_y, ring, nslab, stem, thin;
};

static bool wgrad_plan(int dtype, int kh, int kw, int stride, int Co, int Ci, int B, int Ho, int Wo, WgradPlan* p,
                       bool allow_ring = true, bool allow_thin = true) {
  const bool s1 = (kh == 3 && kw == 3 && stride == 1);
  const bool s2 = (kh == 3 && kw == 3 && stride == 2);
  const bool one = (kh == 1 && kw == 1 && (stride == 1 || stride == 2));
  const bool stem = (kh == 7 && kw == 7 && stride == 2);
  if (!(s1 || s2 || one || stem)) return false;
  const bool f32 = (dtype == FFA_F32);
  p->rg = stem ? 1 : kh;
  p->wk = 1;
  p->thin = 0;
  {
    // conv3x3_thin_wgrad_kernel: bf16, 3x3 stride 1, channel pitches 16 or 32 on both sides (FFA_THIN_WGRAD=0 disables)
    const char* tw_ = getenv("FFA_THIN_WGRAD");
    if (allow_thin && s1 && !f32 && !(tw_ && tw_[0] == '0') && (Co == 16 || Co == 32) && (Ci == 16 || Ci == 32)) {
      p->thin = 1;
      p->th = 8; p->tw = 32;
      p->wco = p->wci = 1;
      p->ncob = p->ncib = 1;
      p->CoT = Co; p->CiT = Ci;
      p->tiles_x = ffa_cdiv(Wo, 32);
      p->tiles_y = ffa_cdiv(Ho, 8);
      p->npt = B * p->tiles_x * p->tiles_y;
      p->ring = 0; p->stem = 0;
      p->nsplit = p->npt < 512 ? p->npt : 512;  // persistent blocks, two per CU; one slab each
      p->nslab = p->nsplit;
      return true;
    }
  }
  if (s1) {
    p->tw = (Wo >= 32) ? 32 : 16;
    p->th = (Wo >= 32) ? 4 : 8;
    p->wco = (Co > 32) ? 2 : 1;
    p->wci = (Ci > 32 && !f32) ? 2 : 1;  // f32 planes are twice as large: one ci plane per block
    if (!f32) {
      p->wk = 4 / (p->wco * p->wci);
      if (p->wk == 1) {  // 64 x 64 channel blocks: eight waves, two k-split groups, 256-pixel tiles, one slab
        p->wk = 2;
        p->th *= 2;
      } else if (p->wk == 4) {
        // thin layers (<= 32 channels on both sides, HBM-bound): 256-pixel tiles = twice the bytes in flight per
        // block, -9 % (512-pixel tiles spill); FFA_WG_THIN_SMALL=1 restores the 128-pixel tiles for A/B runs
        static const bool small_tiles = getenv("FFA_WG_THIN_SMALL") && getenv("FFA_WG_THIN_SMALL")[0] == '1';
        if (!small_tiles) p->th *= 2;
      }
    }
  } else {  // only the shapes the network needs are instantiated for the strided / 1x1 / stem kernels
    p->tw = 16;
    p->th = stem ? 8 : 4;
    p->wco = 2;
    p->wci = (f32 || stem) ? 1 : 2;
    if (stem && !f32) p->wk = 2;
  }
  p->stem = (stem && !f32 && Ci <= 16 && Co % 64 == 0) ? 1 : 0;  // stem_wgrad_kernel
  p->ncob = ffa_cdiv(Co, 32 * p->wco);
  p->ncib = ffa_cdiv(Ci, 32 * p->wci);
  p->CoT = p->ncob * 32 * p->wco;
  p->CiT = p->ncib * 32 * p->wci;
  p->tiles_x = ffa_cdiv(Wo, p->tw);
  p->tiles_y = ffa_cdiv(Ho, p->th);
  p->npt = B * p->tiles_x * p->tiles_y;
  const int tile_blocks = p->ncob * p->ncib * (kh / p->rg);
  // conv_wgrad_ring_kernel (one block per CU, LDS-DMA ring): tiles must divide the image and offsets must fit
  // 31 bits.  Measured on MI355X (tools/bench_kernels.py, round 1): 0.48-0.52 PFLOP/s vs 0.52-0.58 for the
  // register-staged kernel at two blocks per CU -- SQ_WAIT_ANY 40 % of wave cycles, i.e. the DMA of 45 KB per
  // 128-pixel tile does not arrive within two tiles of matrix work.  Kept selectable (FFA_WGRAD_RING=1) for the
  // next round's work on the fill path (full-line piece order, deeper ring); off by default.
  static const bool ring_enabled = getenv("FFA_WGRAD_RING") && getenv("FFA_WGRAD_RING")[0] == '1';
  // the ring kernel walks 128-pixel tiles (4 x 32 / 8 x 16) with ONE k group per block: half the rows of the
  // 256-pixel tiles chosen above for the eight-wave kernel
  const int ring_th = p->th / 2;
  p->ring = (ring_enabled && allow_ring && s1 && !f32 && p->wco == 2 && p->wci == 2 && p->wk == 2 && Ho % ring_th == 0 &&
             Wo % p->tw == 0 && (long long)B * Ho * Wo * (Co > Ci ? Co : Ci) * 2 < (1LL << 31))
                ? 1
                : 0;
  if (p->ring) {
    p->th = ring_th;
    p->wk = 1;
    p->tiles_y = ffa_cdiv(Ho, p->th);
    p->npt = B * p->tiles_x * p->tiles_y;
  }
  const int threads = 64 * p->wco * p->wci * p->wk;
  // a split costs one f32 slab of HBM traffic; 512-thread blocks run one per CU, so their grid is kept at or
  // just below 256 blocks (a 288-block grid would need two rounds)
  int ns = (p->ring || threads >= 512) ? 256 / tile_blocks : ffa_cdiv(512, tile_blocks);
  if (ns > p->npt) ns = p->npt;
  if (ns < 1) ns = 1;
  if (p->stem) {  // one 512-thread block per CU, 64 output channels per block, 16-channel slabs
    p->wk = 1;
    p->CoT = Co;
    p->CiT = 16;
    ns = 256 / (Co / 64);
    if (ns > p->npt) ns = p->npt;
    if (ns < 1) ns = 1;
  }
  p->nsplit = ns;
  const bool merge = p->wk > 1 && (p->wco * p->wci == 4 || stem);  // mirrors WgradGeom::MERGE
  p->nslab = merge ? ns : ns * p->wk;
  return true;
}

extern "C" long long ffa_conv_wgrad_workspace_bytes(int dtype, int kh, int kw, int stride, int Co, int Ci, int B, int Ho,
                                                    int Wo) {
  WgradPlan p, q;
  if (!wgrad_plan(dtype, kh, kw, stride, Co, Ci, B, Ho, Wo, &p)) return FFA_ERR_UNSUPPORTED;
  long long need = (long long)p.nslab * p.CoT * kh * kw * p.CiT * (long long)sizeof(float);
  if (p.thin) {  // a two-source call with a real skip part falls back to conv_wgrad_kernel: size for either
    wgrad_plan(dtype, kh, kw, stride, Co, Ci, B, Ho, Wo, &q, true, false);
    const long long other = (long long)q.nslab * q.CoT * kh * kw * q.CiT * (long long)sizeof(float);
    if (other > need) need = other;
  }
  return need;
}

template <typename T, int KH, int KW, int STRIDE, int RG, int WCO, int WCI, int WK, int TH, int TW, bool PRO = false>
static void launch_wgrad_cfg(const WgradArgs& a, int nrg, hipStream_t stream) {
  dim3 grid(a.ncob * a.ncib * nrg, a.nsplit);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, KH, KW, STRIDE, RG, WCO, WCI, WK, TH, TW, PRO>), grid,
                     dim3(64 * WCO * WCI * WK), 0, stream, a);
}

template <typename T>
static int launch_wgrad(const WgradArgs& a, const WgradPlan& p, int kh, int kw, int stride, hipStream_t stream) {
  constexpr bool F32 = (sizeof(T) == 4);
  const bool wide = (p.tw == 32);
  if constexpr (!F32) {
    if (p.thin) {
      const bool up = a.C1 > 0, pro = a.pro_sc != nullptr;
#define FFA_WGTHIN(CI_, CO_)                                                                                     \
  if (a.Ci == CI_ && a.Co == CO_) {                                                                              \
    if (up && pro) hipLaunchKernelGGL((conv3x3_thin_wgrad_kernel<CI_, CO_, true, true>), dim3(a.nsplit), dim3(256), 0, stream, a);   \
    else if (up) hipLaunchKernelGGL((conv3x3_thin_wgrad_kernel<CI_, CO_, true, false>), dim3(a.nsplit), dim3(256), 0, stream, a);    \
    else if (pro) hipLaunchKernelGGL((conv3x3_thin_wgrad_kernel<CI_, CO_, false, true>), dim3(a.nsplit), dim3(256), 0, stream, a);   \
    else hipLaunchKernelGGL((conv3x3_thin_wgrad_kernel<CI_, CO_, false, false>), dim3(a.nsplit), dim3(256), 0, stream, a);           \
    return ffa_check_launch("conv3x3_thin_wgrad");                                                               \
  }
      FFA_WGTHIN(16, 16)
      FFA_WGTHIN(16, 32)
      FFA_WGTHIN(32, 16)
      FFA_WGTHIN(32, 32)
#undef FFA_WGTHIN
      ffa_set_error("conv_wgrad: no thin kernel for pitches %d / %d", a.Ci, a.Co);
      return FFA_ERR_UNSUPPORTED;
    }
    if (p.ring) {
      dim3 grid(a.ncob * a.ncib, a.nsplit);
      if (wide) hipLaunchKernelGGL((conv_wgrad_ring_kernel<4, 32>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((conv_wgrad_ring_kernel<8, 16>), grid, dim3(256), 0, stream, a);
      return ffa_check_launch("conv_wgrad_ring");
    }
  }
  if (kh == 3 && stride == 1) {
#define FFA_WG_S1(WCO_, WCI_, WK_, THW_, THN_)                                               \
  if (p.wco == WCO_ && p.wci == WCI_ && p.wk == WK_) {                                       \
    if (wide) launch_wgrad_cfg<T, 3, 3, 1, 3, WCO_, WCI_, WK_, THW_, 32>(a, 1, stream);      \
    else launch_wgrad_cfg<T, 3, 3, 1, 3, WCO_, WCI_, WK_, THN_, 16>(a, 1, stream);           \
    return ffa_check_launch("conv_wgrad");                                                   \
  }
    if constexpr (F32) {
      FFA_WG_S1(2, 1, 1, 4, 8)
      FFA_WG_S1(1, 1, 1, 4, 8)
    } else {
      if (a.pro_sc) {  // normalise-on-load: the 64 x 64-channel block configuration only (the layers that use it)
        if (!(p.wco == 2 && p.wci == 2 && p.wk == 2) || a.C1 > 0) {
          ffa_set_error("conv_wgrad: no prologue for this channel configuration");
          return FFA_ERR_UNSUPPORTED;
        }
        if (wide) launch_wgrad_cfg<T, 3, 3, 1, 3, 2, 2, 2, 8, 32, true>(a, 1, stream);
        else launch_wgrad_cfg<T, 3, 3, 1, 3, 2, 2, 2, 16, 16, true>(a, 1, stream);
        return ffa_check_launch("conv_wgrad");
      }
      {
        // conv3x3_wgrad64_kernel: whole 64-channel blocks, 8 x 32 or 16 x 16 tiles (FFA_WGRAD64=0: conv_wgrad_kernel)
        const char* e64 = getenv("FFA_WGRAD64");
        const int ih = wide ? 10 : 18, iw = wide ? 34 : 18;
        if (!(e64 && e64[0] == '0') && p.wco == 2 && p.wci == 2 && p.wk == 2 && p.th == (wide ? 8 : 16) && a.C1 % 64 == 0 &&
            (a.C1 == 0 || (a.Hi % 2 == 0 && a.Wi % 2 == 0)) && a.pad == 1 && a.Co % 64 == 0 && a.Ci % 64 == 0 &&
            ((long long)ih * a.Wi + iw) * 2 * (a.Ci > a.Co ? a.Ci : a.Co) < (1LL << 23)) {
          hipEvent_t ts, te;
          const dim3 grid(a.ncob * a.ncib, a.nsplit);
          const bool timed = ffa_ktime_next(FFA_KT_WGRAD64, &ts, &te);
          if (wide) {
            if (timed) hipExtLaunchKernelGGL(conv3x3_wgrad64_kernel<32>, grid, dim3(512), 0, stream, ts, te, 0, a);
            else hipLaunchKernelGGL(conv3x3_wgrad64_kernel<32>, grid, dim3(512), 0, stream, a);
          } else {
            if (timed) hipExtLaunchKernelGGL(conv3x3_wgrad64_kernel<16>, grid, dim3(512), 0, stream, ts, te, 0, a);
            else hipLaunchKernelGGL(conv3x3_wgrad64_kernel<16>, grid, dim3(512), 0, stream, a);
          }
          return ffa_check_launch("conv3x3_wgrad64");
        }
      }
      FFA_WG_S1(2, 2, 2, 8, 16)
      FFA_WG_S1(2, 1, 2, 4, 8)
      FFA_WG_S1(1, 2, 2, 4, 8)
      if (p.wco == 1 && p.wci == 1 && p.wk == 4 && (p.th == 8 && wide || p.th == 16 && !wide)) {
        if (wide) launch_wgrad_cfg<T, 3, 3, 1, 3, 1, 1, 4, 8, 32>(a, 1, stream);
        else launch_wgrad_cfg<T, 3, 3, 1, 3, 1, 1, 4, 16, 16>(a, 1, stream);
        return ffa_check_launch("conv_wgrad");
      }
      FFA_WG_S1(1, 1, 4, 4, 8)
    }
#undef FFA_WG_S1
  } else if (kh == 3 && stride == 2) {
    launch_wgrad_cfg<T, 3, 3, 2, 3, 2, F32 ? 1 : 2, 1, 4, 16>(a, 1, stream);
    return ffa_check_launch("conv_wgrad");
  } else if (kh == 1 && stride == 2) {
    launch_wgrad_cfg<T, 1, 1, 2, 1, 2, F32 ? 1 : 2, 1, 4, 16>(a, 1, stream);
    return ffa_check_launch("conv_wgrad");
  } else if (kh == 1 && stride == 1) {
    launch_wgrad_cfg<T, 1, 1, 1, 1, 2, F32 ? 1 : 2, 1, 4, 16>(a, 1, stream);
    return ffa_check_launch("conv_wgrad");
  } else if (kh == 7) {
    if constexpr (!F32) {
      if (p.stem) {
        const char* e8 = getenv("FFA_STEM_WGRAD8");
        if (!(e8 && e8[0] == '0') && a.ci_real <= 8 && a.Ci == 16 && a.Co == 64 && a.pad == 3 && a.Wi <= 3000 && a.Wo <= 1500 &&
            (long long)a.B * a.Hi * a.Wi * 32 < (1LL << 31)) {
          hipLaunchKernelGGL(stem_wgrad8_kernel, dim3(1, a.nsplit), dim3(512), 0, stream, a);
          return ffa_check_launch("stem_wgrad8");
        }
        hipLaunchKernelGGL((stem_wgrad_kernel<0>), dim3(a.Co / 64, a.nsplit), dim3(512), 0, stream, a);
        return ffa_check_launch("stem_wgrad");
      }
    }
    launch_wgrad_cfg<T, 7, 7, 2, 1, 2, 1, F32 ? 1 : 2, 8, 16>(a, 7, stream);
    return ffa_check_launch("conv_wgrad");
  }
  ffa_set_error("conv_wgrad: no kernel for %dx%d stride %d waves %dx%dx%d", kh, kw, stride, p.wco, p.wci, p.wk);
  return FFA_ERR_UNSUPPORTED;
}

// x: [B][Hi][Wi][Ci] (the conv input), dy: [B][Ho][Wo][Co]; Ci / Co are channel pitches, the
// gradient is written for the first Co_real x Ci_real entries as OIHW f32 (accumulate != 0 adds to it).
static int wgrad_impl(int dtype, const void* x, const void* x2, int C1, const void* dy, float* dw_oihw, int B, int Hi,
                      int Wi, int Ci, int Ho, int Wo, int Co, int Co_real, int Ci_real, int kh, int kw, int stride,
                      int pad, int accumulate, void* workspace, long long workspace_bytes, hipStream_t stream,
                      const float* pro_scale = nullptr, const float* pro_shift = nullptr) {
  FFA_REQUIRE((pro_scale == nullptr) == (pro_shift == nullptr), "conv_wgrad: prologue needs scale and shift");
  FFA_REQUIRE(!pro_scale || (dtype == FFA_BF16 && kh == 3 && kw == 3 && stride == 1 && pad == 1),
              "conv_wgrad: the prologue is for bf16 3x3 stride-1 pad-1 layers");
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "conv_wgrad: bad dtype");
  FFA_REQUIRE(x && dy && dw_oihw && workspace, "conv_wgrad: null pointer");
  FFA_REQUIRE(Ci % 8 == 0 && Co % 8 == 0, "conv_wgrad: channel pitch must be a multiple of 8");
  FFA_REQUIRE(Co_real <= Co && Ci_real <= Ci, "conv_wgrad: real channels exceed pitch");
  WgradPlan p;
  // the ring kernel's edge masks assume a one-pixel halo, and it has no two-source loader
  // the thin kernel reads one source: plain input, or the skip-less nearest-x2 form (C1 == Ci); pad-1 only
  const bool thin_ok = pad == 1 && (C1 == 0 || (C1 == Ci && x2 == nullptr));
  if (!wgrad_plan(dtype, kh, kw, stride, Co, Ci, B, Ho, Wo, &p, pad == 1 && C1 == 0, thin_ok)) {
    ffa_set_error("conv_wgrad: unsupported kernel %dx%d stride %d", kh, kw, stride);
    return FFA_ERR_UNSUPPORTED;
  }
  if (C1 > 0 && !p.thin) {
    if (C1 % (32 * p.wci) != 0) {
      ffa_set_error("conv_wgrad_upcat: C1 = %d is not a multiple of the block's %d input channels", C1, 32 * p.wci);
      return FFA_ERR_UNSUPPORTED;
    }
  }
  const long long need = (long long)p.nslab * p.CoT * kh * kw * p.CiT * (long long)sizeof(float);
  if (workspace_bytes < need) {
    ffa_set_error("conv_wgrad: workspace too small (%lld < %lld)", workspace_bytes, need);
    return FFA_ERR_WORKSPACE;
  }
  WgradArgs a;
  a.x = x; a.dy = dy; a.slabs = static_cast<float*>(workspace);
  a.x2 = x2; a.C1 = C1;
  a.pro_sc = pro_scale; a.pro_sh = pro_shift;
  a.ci_real = Ci_real;
  a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci;
  a.Ho = Ho; a.Wo = Wo; a.Co = Co;
  a.pad = pad;
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.npt = p.npt;
  a.ncob = p.ncob; a.ncib = p.ncib; a.nsplit = p.nsplit;
  a.CoT = p.CoT; a.CiT = p.CiT;
  int rc = (dtype == FFA_BF16) ? launch_wgrad<ffa_bf16>(a, p, kh, kw, stride, stream)
                               : launch_wgrad<float>(a, p, kh, kw, stride, stream);
  if (rc != FFA_OK) return rc;
  const long long total = (long long)Co_real * kh * kw * ((Ci_real + 3) / 4);
  const char* rv = getenv("FFA_WG_REDUCE_V1");  // A/B switch, read per call (tests flip it)
  const bool reduce_v1 = rv && rv[0] == '1';
  // contiguous OIHW runs per block, where that still fills the chip: one block per (output channel, quad group), so
  // the thin layers (16-64 channels: 16-128 blocks walking 256 slabs each) stay on the thread-per-element kernel
  // (measured +25 us each on the 16 / 32-channel decoder layers, -10...-15 us on the 256 / 512-channel ones)
  const int cq = (Ci_real + 3) / 4;
  const int v2_blocks = Co_real * ffa_cdiv(cq, p.nslab > 8 ? 8 : 64);
  if (kh == 3 && kw == 3 && !reduce_v1 && v2_blocks >= 512) {
    if (p.nslab > 8) {
      hipLaunchKernelGGL(wgrad_reduce3x3_kernel<32>, dim3(v2_blocks), dim3(256), 0, stream,
                         (const float*)workspace, dw_oihw, p.nslab, p.CoT, p.CiT, Co_real, Ci_real, accumulate);
    } else {
      hipLaunchKernelGGL(wgrad_reduce3x3_kernel<4>, dim3(v2_blocks), dim3(256), 0, stream,
                         (const float*)workspace, dw_oihw, p.nslab, p.CoT, p.CiT, Co_real, Ci_real, accumulate);
    }
    return ffa_check_launch("wgrad_reduce3x3");
  }
  if (p.nslab > 8) {
    long long g = (total + 7) / 8;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3((int)g), dim3(256), 0, stream, (const float*)workspace, dw_oihw,
                       p.nslab, p.CoT, p.CiT, Co_real, Ci_real, kh * kw, accumulate);
  } else {
    long long g = (total + 63) / 64;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((int)g), dim3(256), 0, stream, (const float*)workspace, dw_oihw,
                       p.nslab, p.CoT, p.CiT, Co_real, Ci_real, kh * kw, accumulate);
  }
  return ffa_check_launch("wgrad_reduce");
}

extern "C" int ffa_conv_wgrad(int dtype, const void* x, const void* dy, float* dw_oihw, int B, int Hi, int Wi, int Ci,
                              int Ho, int Wo, int Co, int Co_real, int Ci_real, int kh, int kw, int stride, int pad,
                              int accumulate, void* workspace, long long workspace_bytes, hipStream_t stream) {
  return wgrad_impl(dtype, x, nullptr, 0, dy, dw_oihw, B, Hi, Wi, Ci, Ho, Wo, Co, Co_real, Ci_real, kh, kw, stride, pad,
                    accumulate, workspace, workspace_bytes, stream);
}

// ffa_conv_wgrad / ffa_conv_wgrad_upcat (skip-less form: C2 = 0) for a layer whose input was relu(x * pro_scale[c] +
// pro_shift[c]) -- the BatchNorm + ReLU of the producing layer folded into its consumers ("normalise on load"): x is the
// PRE-normalisation tensor, the staged pieces are rewritten in the loader (same fma / max / rounding as ffa_bn_apply,
// zero padding applied after the normalisation).  bf16, 3x3 stride 1 pad 1; up != 0: x is the low-resolution map of the
// nearest-x2 form [B][Hi/2][Wi/2][Ci].  FFA_ERR_UNSUPPORTED for channel configurations without a prologue kernel.
extern "C" int ffa_conv_wgrad_pro(int dtype, const void* x, const void* dy, float* dw_oihw, const float* pro_scale,
                                  const float* pro_shift, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                                  int Co_real, int Ci_real, int up, int accumulate, void* workspace,
                                  long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(pro_scale && pro_shift, "conv_wgrad_pro: null prologue vectors");
  return wgrad_impl(dtype, x, nullptr, up ? Ci : 0, dy, dw_oihw, B, Hi, Wi, Ci, Ho, Wo, Co, Co_real, Ci_real, 3, 3, 1, 1,
                    accumulate, workspace, workspace_bytes, stream, pro_scale, pro_shift);
}

// Weight gradient of the 3x3 stride-1 pad-1 convolution whose input is the virtual cat(nearest_x2(lo), skip)
// (see ffa_conv2d_upcat): lo [B][Hl][Wl][C1], skip [B][2Hl][2Wl][C2] or null; dw is OIHW [Co_real][C1 + C2][3][3].
// Workspace: ffa_conv_wgrad_workspace_bytes(dtype, 3, 3, 1, Co, C1 + C2, B, 2Hl, 2Wl).
extern "C" int ffa_conv_wgrad_upcat(int dtype, const void* lo, const void* skip, const void* dy, float* dw_oihw, int B,
                                    int Hl, int Wl, int C1, int C2, int Co, int Co_real, int accumulate,
                                    void* workspace, long long workspace_bytes, hipStream_t stream) {
  FFA_REQUIRE(lo && (skip || C2 == 0) && C1 > 0 && C2 >= 0, "conv_wgrad_upcat: bad arguments");
  return wgrad_impl(dtype, lo, skip, C1, dy, dw_oihw, B, 2 * Hl, 2 * Wl, C1 + C2, 2 * Hl, 2 * Wl, Co, Co_real, C1 + C2,
                    3, 3, 1, 1, accumulate, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// hardware-layout probes used by tests/test_layout_probes.py: they pin the lane maps this file and
// conv_igemm.hip rely on (MFMA operand / accumulator layout, transpose-read semantics).

__global__ void probe_tr16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst) {
  // src: 64 rows x 64 bf16 (128-B rows).  Each lane reads with row = lane>>2 .. as documented above
  __shared__ __align__(16) unsigned char buf[64 * 128];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) reinterpret_cast<uint16_t*>(buf)[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x;
  const int li = lane & 15;
  const int grp = lane >> 4;
  // group g reads rows 4g..4g+3, columns 16g.. : address of row (li>>2), segment (li&3)
  const unsigned char* p = buf + (4 * grp + (li >> 2)) * 128 + grp * 32 + (li & 3) * 8;
  ffa_s16x4 v = lds_read_tr16(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) dst[lane * 4 + e] = (uint16_t)v[e];
}

extern "C" int ffa_probe_tr16(const uint16_t* src, uint16_t* dst, hipStream_t stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, stream, src, dst);
  return ffa_check_launch("probe_tr16");
}

__global__ void probe_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ D,
                                  int use_f32) {
  // A: [32][16] row-major, B: [16][32] row-major, D: [32][32] row-major, values exactly representable in bf16
  const int lane = threadIdx.x;
  const int rho = lane & 31, half = lane >> 5;
  ffa_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (!use_f32) {
    ffa_bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = (__bf16)A[rho * 16 + 8 * half + j];
      b[j] = (__bf16)Bm[(8 * half + j) * 32 + rho];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // same lane->k map as the bf16 path at 16-byte granularity: k = 4*half + j, second k block +8
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rho * 16 + 4 * half + j], Bm[(4 * half + j) * 32 + rho], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rho * 16 + 8 + 4 * half + j], Bm[(8 + 4 * half + j) * 32 + rho], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    D[row * 32 + rho] = acc[r];
  }
}

extern "C" int ffa_probe_mfma(const float* A, const float* B, float* D, int use_f32, hipStream_t stream) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, stream, A, B, D, use_f32);
  return ffa_check_launch("probe_mfma");
}

#if FFA_WGRAD_TRACE
extern "C" int ffa_wgrad_trace_read(long long* host_dst, int n) {
  if (n > 64 * 256) n = 64 * 256;
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ffa_wgrad_trace_buf), (size_t)n * sizeof(long long));
}
extern "C" int ffa_wgrad_trace_clear() {
  static long long zeros[64 * 256];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ffa_wgrad_trace_buf), zeros, sizeof(zeros));
}
#endif
