// The ResNet stem: 7x7 stride-2 pad-3 convolution of a 5-channel tile into 64 channels (bf16; round 3).
//
// Replaces the generic conv_igemm_kernel<7, 7, 2> for torchvision's `conv1 = Conv2d(in, 64, 7, 2, 3, bias=False)` as smp's
// ResNetEncoder keeps it (reference: flair_hub/models/monotemp_model.py:68-92 -> smp.create_model; oracle/unet_resnet34.py).
// The generic kernel multiplies one 16-channel pixel (the NHWC pitch of the 5-channel input) per 32-byte k-step: K = 49
// taps x 16 = 784, of which 245 are real -- it ran at 0.18 PFLOP/s of useful work, 3.4x its HBM floor.  Here:
//   * only the first 8 channels (16 bytes) of every input pixel are staged (LDS-DMA, per-lane source address: the DMA
//     gathers one 16-byte piece per pixel; pad channels 5..7 hold zeros by the layout contract);
//   * one K = 32 step of v_mfma_f32_16x16x32_bf16 = FOUR horizontally adjacent taps x 8 channels: lane (pixel, kg) reads
//     the 16 bytes of input pixel 2 * ox + 4 * half + kg -- K = 7 rows x 2 halves x 32 = 448 (the eighth tap has zero
//     weights), 43 % fewer matrix instructions;
//   * even and odd input columns live in separate LDS planes, so the 16 lanes of a fragment (stride-2 pixels) read 256
//     consecutive bytes: conflict-free ds_read_b128;
//   * the whole weight operand (14 k-steps x 64 rows x 64 B = 56 KB, ring16's row permutation and slot swizzle) stays in
//     LDS for the launch; persistent blocks of eight waves walk 16 x 32-pixel output tiles, the halo of the next tile
//     arrives while the current one is multiplied;
//   * epilogue as conv3x3_ring16_kernel's: a lane owns 16 consecutive channels of its pixel, optional bias / ReLU,
//     BatchNorm batch statistics of the stored values (per-tile partial rows for ffa_bn_finalize).
#include "ffa_common.h"
#include "ffa_common_host.h"

struct StemArgs {
  const void* in;   // [B][Hi][Wi][16] bf16
  const void* w;    // operand packed by ffa_stem_pack
  void* out;        // [B][Ho][Wo][Co] bf16
  const float* bias;
  float* stats;     // [npt][2][Co] or null
  int B, Hi, Wi, Ho, Wo, Co;
  int relu;
  int tiles_x, tiles_y, npt;
};

struct StemGeom {
  static constexpr int TH = 16, TW = 32, NW = 8, NTHR = 512;
  static constexpr int IH = 2 * TH + 5;           // 37 input rows
  static constexpr int IWH = 36;                  // pixel pairs per row and parity (2 * TW + 5 = 69 columns)
  static constexpr int ROWP = 2 * IWH;            // 16-byte pieces per halo row: [parity][IWH]
  static constexpr int HP = IH * ROWP;            // 2664 pieces
  static constexpr int NHW = (HP + NTHR - 1) / NTHR;
  static constexpr int HBYTES = HP * 16;
  static constexpr int KSTEPS = 14;
  static constexpr int WBYTES = KSTEPS * 4096;    // [k-step][64 rows][64 B]
  static constexpr int WP = WBYTES / 16;
  static constexpr int W_OFF = 0, H_OFF = WBYTES, RED_OFF = WBYTES + 2 * HBYTES;
  static constexpr int RED_BYTES = NW * 64 * 2 * 4;
  static constexpr int LDS_BYTES = RED_OFF + RED_BYTES;
  static_assert(WP % NTHR == 0, "every thread issues the same number of weight pieces");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __attribute__((aligned(16))) const unsigned int ffa_stem_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void stem_dma16(const unsigned char* src, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_base)
      : "memory");
}

__global__ void __launch_bounds__(512) conv7x7_stem_kernel(StemArgs a) {
  using G = StemGeom;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;

  // ---- weights: once per block ----
  {
    const unsigned char* wsrc = static_cast<const unsigned char*>(a.w);
#pragma unroll
    for (int k = 0; k < G::WP / G::NTHR; ++k) {
      const int p = tid + k * G::NTHR;
      stem_dma16(wsrc + (size_t)p * 16,
                 (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + G::W_OFF + (wave * 64 + k * G::NTHR) * 16));
    }
  }

  // ---- halo pieces of this thread, fixed for the launch: piece p = (row hy, parity, pair xi) <- input pixel
  //      (2 * oy0 - 3 + hy, 2 * ox0 - 3 + 2 * xi + parity); source offset from the tile's origin pixel in bytes ----
  int pinfo[G::NHW];  // hy << 8 | hx, -1: no piece
  int poff[G::NHW];
#pragma unroll
  for (int k = 0; k < G::NHW; ++k) {
    const int p = tid + k * G::NTHR;
    const int hy = p / G::ROWP, rem = p % G::ROWP;
    const int hx = 2 * (rem % G::IWH) + rem / G::IWH;
    pinfo[k] = (p < G::HP && hx < 2 * G::TW + 5) ? ((hy << 8) | hx) : -1;
    poff[k] = (hy * a.Wi + hx) * 32;
  }
  const unsigned char* in_b = static_cast<const unsigned char*>(a.in);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(ffa_stem_zero16);

  auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
    const int tx = t % a.tiles_x;
    const int t2 = t / a.tiles_x;
    oy0 = (t2 % a.tiles_y) * G::TH;
    b = t2 / a.tiles_y;
    ox0 = tx * G::TW;
  };
  auto issue_halo = [&](int t, int buf) {
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const unsigned char* base = in_b + ((long long)(b * a.Hi + iy0) * a.Wi + ix0) * 32;  // may lie outside: only valid pieces use it
#pragma unroll
    for (int k = 0; k < G::NHW; ++k) {
      const int info = pinfo[k];
      const int iy = iy0 + (info >> 8), ix = ix0 + (info & 255);
      const bool valid = info >= 0 && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
      const unsigned char* src = valid ? base + poff[k] : zero;
      const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(
          smem + G::H_OFF + buf * G::HBYTES + (wave * 64 + k * G::NTHR) * 16);
      if (k + 1 < G::NHW || G::HP % G::NTHR == 0) {
        stem_dma16(src, dst);
      } else if (wave * 64 + k * G::NTHR < G::HP) {      // wave uniform
        if (tid + k * G::NTHR < G::HP) stem_dma16(src, dst);  // (lanes past the last piece would write into the next buffer)
      }
    }
  };

  // ---- fragment addressing ----
  // A: k-step image [64 rows][64 B]; lane (row = col, kg) reads slot kg of row mt * 16 + col, slot ^= 2 when bit 2 of the row
  const int a0 = G::W_OFF + col * 64 + ((kg ^ (((col >> 2) & 1) << 1)) * 16);
  // B: fragment nt = output row 2 * wave + (nt >> 1), columns (nt & 1) * 16 + col; tap 4 * half + kg of kernel row r reads
  // input column 2 * ocol + 4 * half + kg: parity kg & 1, pair index ocol + 2 * half + (kg >> 1)
  int b0[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int py = 2 * wave + (nt >> 1), ocol = (nt & 1) * 16 + col;
    b0[nt] = G::H_OFF + (((2 * py) * 2 + (kg & 1)) * G::IWH + ocol + (kg >> 1)) * 16;
  }

  ffa_f32x4 acc[4][4];
  int t = blockIdx.x;
  if (t < a.npt) issue_halo(t, 0);
  int buf = 0;
  for (; t < a.npt; t += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // this tile's halo (and the weights) landed; other buffer free
    const int tn = t + gridDim.x;
    if (tn < a.npt) issue_halo(tn, buf ^ 1);
    const unsigned char* sH = smem + buf * G::HBYTES;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};
    ffa_u32x4 fa[2][4], fb[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) fa[0][mt] = *reinterpret_cast<const ffa_u32x4*>(smem + a0 + mt * 1024);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) fb[0][nt] = *reinterpret_cast<const ffa_u32x4*>(sH + b0[nt]);
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < G::KSTEPS) {
        const int r = (ks + 1) >> 1, half = (ks + 1) & 1;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          fa[nxt][mt] = *reinterpret_cast<const ffa_u32x4*>(smem + a0 + (ks + 1) * 4096 + mt * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          fb[nxt][nt] = *reinterpret_cast<const ffa_u32x4*>(sH + b0[nt] + (r * G::ROWP + 2 * half) * 16);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ffa_bf16x8, fa[cur][mt]),
                                                                __builtin_bit_cast(ffa_bf16x8, fb[cur][nt]), acc[mt][nt],
                                                                0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 8 && ks + 1 < G::KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }

    // ---- epilogue: lane (col, kg) owns pixel (2 * wave + (nt >> 1), (nt & 1) * 16 + col) and channels 16 * kg .. + 15 ----
    {
      int b, oy0, ox0;
      tile_origin(t, b, oy0, ox0);
      ffa_bf16* out = static_cast<ffa_bf16*>(a.out);
      float st[32];
      const bool want_stats = a.stats != nullptr;
#pragma unroll
      for (int i = 0; i < 32; ++i) st[i] = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int oy = oy0 + 2 * wave + (nt >> 1), ox = ox0 + (nt & 1) * 16 + col;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const long long pix = ((long long)(b * a.Ho + oy) * a.Wo + ox) * (long long)a.Co;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c0 = 16 * kg + 8 * h;
          float v[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[2 * h][nt][i];
            v[4 + i] = acc[2 * h + 1][nt][i];
          }
          if (a.bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += a.bias[c0 + i];
          }
          if (a.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
          }
          ffa_u32x4 u;
          u.x = ffa_pack_bf16x2(v[0], v[1]);
          u.y = ffa_pack_bf16x2(v[2], v[3]);
          u.z = ffa_pack_bf16x2(v[4], v[5]);
          u.w = ffa_pack_bf16x2(v[6], v[7]);
          *reinterpret_cast<ffa_u32x4*>(out + pix + c0) = u;
          if (want_stats) {
            const float rr[8] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                 __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u),
                                 __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                                 __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              st[h * 8 + i] += rr[i];
              st[16 + h * 8 + i] = __builtin_fmaf(rr[i], rr[i], st[16 + h * 8 + i]);
            }
          }
        }
      }
      if (want_stats) {
        // transposing reduction over the 16 lanes that share kg (conv3x3_ring16_kernel's): afterwards lane col holds
        // entries 2 * col and 2 * col + 1 of [16 sums | 16 sums of squares] of channels 16 * kg ..
#pragma unroll
        for (int bit = 3; bit >= 0; --bit) {
          const int n = 32 >> (3 - bit);
          const bool up = (col >> bit) & 1;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            if (j < n / 2) {
              float lo = st[j], hi = st[j + n / 2];
              asm volatile("" : "+v"(lo), "+v"(hi));
              const float keep = up ? hi : lo;
              const float send = up ? lo : hi;
              st[j] = keep + __shfl_xor(send, 1 << bit, 64);
            }
          }
        }
        float* red = reinterpret_cast<float*>(smem + G::RED_OFF);
        red[(wave * 64 + lane) * 2 + 0] = st[0];
        red[(wave * 64 + lane) * 2 + 1] = st[1];
        __syncthreads();
        if (tid < 128) {
          const int j = tid & 1, ln = tid >> 1;
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < G::NW; ++w) s += red[(w * 64 + ln) * 2 + j];
          const int cl = ln & 15, kq = ln >> 4;
          const int which = cl >> 3;
          const int ch = 16 * kq + (cl & 7) * 2 + j;
          a.stats[((size_t)t * 2 + which) * a.Co + ch] = s;
        }
        // (red is written again only behind the next tile's barrier)
      }
    }
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this block may still be writing its LDS
}

// ---- weight packing: dst[k-step = r * 2 + half][LDS row rho][slot'][8 channels] (bf16), LDS row rho = mt * 16 + 4 * q + i
//      holds output channel 16 * q + 4 * mt + i (so that an accumulator lane ends up with 16 consecutive channels), slot =
//      tap 4 * half + slot of kernel row r (tap 7: zeros), slot' = slot ^ 2 when bit 2 of rho ----
__global__ void stem_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, unsigned short* __restrict__ dst,
                                 int O, int I) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread = one 16-byte slot
  if (i >= StemGeom::KSTEPS * 64 * 4) return;
  const int slotp = i & 3, rho = (i >> 2) & 63, ks = i >> 8;
  const int slot = slotp ^ (((rho >> 2) & 1) << 1);
  const int r = ks >> 1, s = 4 * (ks & 1) + slot;
  const int mt = rho >> 4, q = (rho >> 2) & 3, ii = rho & 3;
  const int co = 16 * q + 4 * mt + ii;
  unsigned short v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float x = 0.f;
    if (co < O && c < I && s < 7) {
      x = w[((size_t)(co * I + c) * 7 + r) * 7 + s];
      if (scale) x *= scale[co];
    }
    v[c] = ffa_f32_to_bf16_bits(x);
  }
  uint4 u;
  u.x = v[0] | ((unsigned)v[1] << 16);
  u.y = v[2] | ((unsigned)v[3] << 16);
  u.z = v[4] | ((unsigned)v[5] << 16);
  u.w = v[6] | ((unsigned)v[7] << 16);
  reinterpret_cast<uint4*>(dst)[i] = u;
}

extern "C" int ffa_stem_eligible(int dtype, int kh, int kw, int stride, int cout, int ci_pitch) {
  return dtype == FFA_BF16 && kh == 7 && kw == 7 && stride == 2 && cout == 64 && ci_pitch == 16;
}

extern "C" long long ffa_stem_pack_bytes(void) { return StemGeom::WBYTES; }

extern "C" long long ffa_stem_stat_rows(int B, int Ho, int Wo) {
  return (long long)B * ffa_cdiv(Wo, StemGeom::TW) * ffa_cdiv(Ho, StemGeom::TH);
}

// w_oihw [O = 64][I <= 8][7][7] f32 -> the kernel's operand; scale (optional, per output channel) folds an eval-mode BatchNorm
extern "C" int ffa_stem_pack(const float* w_oihw, const float* scale, void* dst, int O, int I, hipStream_t stream) {
  FFA_REQUIRE(w_oihw && dst && O > 0 && O <= 64 && I > 0 && I <= 8, "stem pack: 1..64 output and 1..8 input channels, got %d / %d", O, I);
  const int n = StemGeom::KSTEPS * 64 * 4;
  hipLaunchKernelGGL(stem_pack_kernel, dim3(ffa_cdiv(n, 256)), dim3(256), 0, stream, w_oihw, scale,
                     static_cast<unsigned short*>(dst), O, I);
  return ffa_check_launch("stem_pack");
}

// in [B][Hi][Wi][16] bf16 (channels >= 8 are never read; pad channels must be zero), out [B][Ho][Wo][Co] with
// Ho = (Hi - 1) / 2 + 1, Wo likewise and Co = 64; stat_partials [ffa_stem_stat_rows][2][Co] or null
extern "C" int ffa_stem_conv7x7(const void* in, const void* w_stem, const float* bias, void* out, float* stat_partials, int B,
                                int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int relu, hipStream_t stream) {
  FFA_REQUIRE(in && w_stem && out, "stem conv: null pointer");
  FFA_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ci == 16 && Co == 64, "stem conv: input pitch 16, output pitch 64 (got %d / %d)", Ci, Co);
  FFA_REQUIRE(Ho == (Hi - 1) / 2 + 1 && Wo == (Wi - 1) / 2 + 1, "stem conv: output size of a 7x7 stride-2 pad-3 convolution");
  FFA_REQUIRE((long long)B * Hi * Wi * 32 < (1LL << 31) && (long long)StemGeom::IH * Wi * 32 < (1LL << 31),
              "stem conv: input tensor must be smaller than 2 GiB");
  StemArgs a;
  a.in = in; a.w = w_stem; a.out = out; a.bias = bias; a.stats = stat_partials;
  a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.relu = relu;
  a.tiles_x = ffa_cdiv(Wo, StemGeom::TW);
  a.tiles_y = ffa_cdiv(Ho, StemGeom::TH);
  a.npt = B * a.tiles_x * a.tiles_y;
  int cap = 256;
  if (const char* e = getenv("FFA_STEM_GRID")) {
    const int v = atoi(e);
    if (v > 0) cap = v;
  }
  const int grid = a.npt < cap ? a.npt : cap;
  hipLaunchKernelGGL(conv7x7_stem_kernel, dim3(grid), dim3(StemGeom::NTHR), 0, stream, a);
  return ffa_check_launch("conv7x7_stem");
}
