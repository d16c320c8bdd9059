// 3x3 stride-1 pad-1 convolution for THIN layers (<= 32 input channels, <= 32 output rows), bf16, gfx950 -- the
// "thin" kernel (round 3).
//
// The U-Net decoder tail and the segmentation head (smp DecoderBlock 3 / 4 conv2, block 4 conv1, SegmentationHead;
// reached from flair_hub/models/monotemp_model.py:68-92, called at flair_hub/models/flair_model.py:417-419) are 16- and
// 32-channel convolutions on 256^2 / 512^2 maps: 0.4-0.8 GB of activations per launch against 0.04-0.15 TFLOP, i.e.
// HBM-bound by a factor of 2-5 on the MFMA.  On conv_igemm.hip they ran at 1.4-3.7 TB/s because that kernel pads the
// output rows to a 32-row MFMA tile (a 16-channel layer does twice the matrix work), re-reads its weight fragment from
// LDS for every pixel fragment (1.5 ds_read_b128 per MFMA = 75 % of the LDS bandwidth of a CU) and stages through
// registers with two barriers per 32-byte k-step.  Here:
//   * v_mfma_f32_16x16x32_bf16: 16 output rows per tile (no padding for 16-channel layers); K = 32 is one tap x 32
//     channels, or TWO taps x 16 channels for the 16-channel inputs (taps (0,1) (2,3) (4,5) (6,7) (8,-): five k-steps
//     instead of nine, the unpaired half multiplies zero weights);
//   * the whole weight operand lives in REGISTERS for the life of the block (<= 72 VGPRs), loaded once from an image
//     packed in fragment order; only pixel fragments are read from LDS: 1 (16 rows) or 0.5 (32 rows) reads per MFMA;
//   * the input halo tile arrives by LDS-DMA (global_load_lds_dwordx4) with per-lane source addresses -- border pixels
//     read a zero page, the nearest-x2 upsampled decoder input reads lo[y >> 1][x >> 1] (the upsampled tensor never
//     exists) -- into a 3-slot ring, two tiles ahead of the tile being multiplied, tracked with a counted vmcnt and ONE
//     barrier per tile; blocks are persistent;
//   * epilogue from the accumulators: bias / residual / ReLU, BatchNorm batch statistics of the stored values, or the
//     2x2 sum pooling that is the adjoint of nearest-x2 (dgrad of the block-4 conv1: the gradient of the upsampled
//     tensor never exists either).
// Accumulator tile (mt, nt): lane (col = lane & 15, kg = lane >> 4) holds rows 4*kg .. 4*kg+3 of pixel col.  The pack
// puts output channel 8*kg + 4*mt + i (two tiles) or 4*kg + i (one tile) into tile row 4*kg + i, so a lane stores 16
// (or 8) contiguous bytes per pixel.
#include "ffa_common.h"

#include <stdlib.h>

struct ThinArgs {
  const void* in;      // [B][H][W][CI] (UP: [B][H/2][W/2][CI], read at (y >> 1, x >> 1))
  const void* w;       // thin pack: [mt][k-step][lane][16 B]
  void* out;           // [B][H][W][Co]  (POOL: [B][H/2][W/2][Co], 2x2 sums)
  const float* bias;   // [Co] or null
  const void* res;     // same layout as out, or null
  float* stats;        // [tiles][2][Co] or null
  const float* pro_sc; // PRO kernels: the convolution reads relu(in * pro_sc[c] + pro_sh[c]) (BatchNorm + ReLU of the
  const float* pro_sh; //   producing layer, evaluated on the halo in LDS: no normalised tensor in memory)
  int B, H, W;         // output (= virtual input) size
  int Co;              // stored output channel pitch
  int relu;
  int tiles_x, tiles_y, ntiles;
};

__device__ __attribute__((aligned(16))) const unsigned int ffa_thin_zero16[4] = {0u, 0u, 0u, 0u};

// one LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS at lds_base + lane * 16
// (M0 written and restored inside the statement; completion is waited for by hand: see conv3x3_ring.hip)
__device__ __forceinline__ void thin_dma16(const unsigned char* src, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void thin_wait_and_meet() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(N) : "memory");
}

template <int CI, int MT>
struct ThinGeom {
  // 64-byte pixels take 8 x 32 tiles so that three halo slots of two blocks fit a CU's LDS; 32-byte pixels 16 x 32
  static constexpr int TH = (CI == 32) ? 8 : 16;
  static constexpr int TW = 32;
  static constexpr int NTHR = 256;
  static constexpr int NT = TH * TW / 4 / 16;  // 16-pixel fragments per wave (a wave owns TH / 4 rows of the tile)
  static constexpr int IH = TH + 2, IW = TW + 2;
  static constexpr int PIXB = CI * 2;          // bytes of one halo pixel: 32 or 64
  static constexpr int PPX = PIXB / 16;        // 16-byte pieces per pixel
  static constexpr int ROWB = IW * PIXB;
  static constexpr int KS = (CI == 32) ? 9 : 5;
  static constexpr int H_PIECES = IH * IW * PPX;
  static constexpr int NHW = (H_PIECES + NTHR - 1) / NTHR;  // DMA instructions of a wave per tile (the tail one may be short / absent)
  static constexpr int SLOT = IH * ROWB;
  static constexpr int NSLOT = 3;
  static constexpr int RED_OFF = NSLOT * SLOT;  // statistics scratch: 4 waves x 64 lanes x 2 floats
  static constexpr int LDS_BYTES = RED_OFF + 4 * 64 * 2 * 4;
  static_assert(CI == 16 || CI == 32, "input channel pitch");
  static_assert(MT == 1 || MT == 2, "output row tiles");
  static_assert(SLOT % 16 == 0 && 2 * LDS_BYTES <= 160 * 1024, "two blocks per CU");
};

template <int CI, int MT, bool UP, bool POOL, bool STATS, bool PRO = false>
__global__ void __launch_bounds__(256, 2) conv3x3_thin_kernel(ThinArgs a) {
  using G = ThinGeom<CI, MT>;
  using T = ffa_bf16;
  constexpr int NT = G::NT, KS = G::KS;
  constexpr int CPL = 4 * MT;  // channels a lane holds per pixel
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;
  const int kg = lane >> 4;

  // ---- weights: the block's whole operand, in registers ----
  ffa_u32x4 wA[MT][KS];
  {
    const ffa_u32x4* wp = static_cast<const ffa_u32x4*>(a.w);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wA[mt][ks] = wp[(mt * KS + ks) * 64 + lane];
  }

  // ---- halo pieces of this thread: piece p = tid + k * 256 -> halo pixel p / PPX, 16-byte slot p % PPX ----
  int hyx[G::NHW];  // ((hy << 8 | hx) << 4) | source slot, or -1 past the last piece
#pragma unroll
  for (int k = 0; k < G::NHW; ++k) {
    const int p = tid + k * G::NTHR;
    const int q = p / G::PPX, sp = p % G::PPX;
    const int hy = q / G::IW, hx = q % G::IW;
    // 64-byte pixels: the 16-byte slot index is XORed with 2 when bit 2 of the halo column is set (conflict-free
    // fragment reads, as in conv3x3_ring16_kernel); the DMA destination is linear, so the XOR goes on the SOURCE slot
    const int slot = (CI == 32) ? (sp ^ (((hx >> 2) & 1) << 1)) : sp;
    hyx[k] = (p < G::H_PIECES) ? ((((hy << 8) | hx) << 4) | slot) : -1;
  }
  const unsigned char* in_b = static_cast<const unsigned char*>(a.in);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(ffa_thin_zero16);
  const int Hs = UP ? (a.H >> 1) : a.H, Ws = UP ? (a.W >> 1) : a.W;  // stored source size
  // the tail instruction exists only in the waves that own pieces of it (exact vmcnt arithmetic per wave)
  const bool has_tail = (G::H_PIECES % G::NTHR == 0) || (wave * 64 + (G::NHW - 1) * G::NTHR < G::H_PIECES);

  auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
    const int tx = t % a.tiles_x;
    const int t2 = t / a.tiles_x;
    oy0 = (t2 % a.tiles_y) * G::TH;
    b = t2 / a.tiles_y;
    ox0 = tx * G::TW;
  };
  auto issue_halo = [&](int t, int slot) {
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
#pragma unroll
    for (int k = 0; k < G::NHW; ++k) {
      const int hy = (hyx[k] >> 12) & 0xff, hx = (hyx[k] >> 4) & 0xff, sl = hyx[k] & 15;
      const int vy = oy0 - 1 + hy, vx = ox0 - 1 + hx;
      const bool ok = hyx[k] >= 0 && vy >= 0 && vx >= 0 && vy < a.H && vx < a.W;
      const int sy = UP ? (vy >> 1) : vy, sx = UP ? (vx >> 1) : vx;
      const unsigned char* src = ok ? in_b + ((size_t)((b * Hs + sy) * Ws + sx) * G::PIXB + sl * 16) : zero;
      const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(
          smem + slot * G::SLOT + (wave * 64 + k * G::NTHR) * 16);
      if (k + 1 < G::NHW || G::H_PIECES % G::NTHR == 0) {
        thin_dma16(src, dst);
      } else if (has_tail) {  // wave uniform
        if (hyx[k] >= 0) thin_dma16(src, dst);
      }
    }
  };
  // this wave's DMA of the tile two fills ago has landed when at most its newest fill is outstanding
  auto wait_tile = [&]() {
    if (has_tail) thin_wait_and_meet<G::NHW>();
    else thin_wait_and_meet<G::NHW - 1>();
  };
  // PRO ("normalise on load"): once this wave's own pieces of the tile have landed, thread t rewrites in LDS the
  // pieces of LOGICAL 16-byte slot t % PPX of its pixels -- always the same 8 channels, whose scale / shift it keeps in
  // registers for the whole launch, and always pieces this very wave's DMA wrote (a pixel's pieces are moved by PPX
  // neighbouring lanes) -- except the zero padding (the padding of the normalised tensor is zero, not relu(shift)).
  // Same fma / max / rounding as ffa_bn_apply: bit-identical to convolving the materialised tensor.
  float psc[PRO ? 8 : 1], psh[PRO ? 8 : 1];
  const int jj = tid % G::PPX;
  if constexpr (PRO) {
    ffa_load8<float>(a.pro_sc + jj * 8, psc);
    ffa_load8<float>(a.pro_sh + jj * 8, psh);
  }
  auto fix_tile = [&](int t, int slot) {
    if constexpr (PRO) {
      int b, oy0, ox0;
      tile_origin(t, b, oy0, ox0);
      unsigned char* base = smem + slot * G::SLOT;
#pragma unroll
      for (int k = 0; k < G::NHW; ++k) {
        const int hy = (hyx[k] >> 12) & 0xff, hx = (hyx[k] >> 4) & 0xff;
        const int vy = oy0 - 1 + hy, vx = ox0 - 1 + hx;
        if (hyx[k] >= 0 && vy >= 0 && vx >= 0 && vy < a.H && vx < a.W) {
          const int phys = (CI == 32) ? (jj ^ (((hx >> 2) & 1) << 1)) : jj;
          ffa_u32x4* ptr = reinterpret_cast<ffa_u32x4*>(base + (hy * G::IW + hx) * G::PIXB + phys * 16);
          ffa_u32x4 v = *ptr;
          float f[8];
          f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
          f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
          f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
          f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaxf(__builtin_fmaf(f[e], psc[e], psh[e]), 0.f);
          v.x = ffa_pack_bf16x2(f[0], f[1]);
          v.y = ffa_pack_bf16x2(f[2], f[3]);
          v.z = ffa_pack_bf16x2(f[4], f[5]);
          v.w = ffa_pack_bf16x2(f[6], f[7]);
          *ptr = v;
        }
      }
    }
  };

  // ---- per-lane fragment read bases (slot base added per tile) ----
  // fragment nt of wave w: tile row w * (NT / 2) + (nt >> 1), columns 16 * (nt & 1) + col; tap (r, s) adds r rows, s pixels
  const int row0 = wave * (NT / 2);
  int fb[(CI == 32) ? 3 : KS];  // CI == 32: per horizontal tap s (the slot swizzle depends on col + s); CI == 16: per k-step
  if constexpr (CI == 32) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int hx = col + s;
      fb[s] = (row0 * G::IW + hx) * G::PIXB + ((kg ^ (((hx >> 2) & 1) << 1)) * 16);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      int tap = 2 * ks + (kg >> 1);
      if (tap > 8) tap = 8;  // the unpaired half of the last step: any valid address (its weights are zero)
      fb[ks] = ((row0 + tap / 3) * G::IW + col + tap % 3) * G::PIXB + (kg & 1) * 16;
    }
  }

  int t = blockIdx.x;
  if (t >= a.ntiles) return;
  // prologue: two fills in flight (a block without a second / third tile re-reads its last one into a slot nobody
  // reads: every wave's DMA count stays uniform)
  issue_halo(t, 0);
  {
    const int t1 = t + (int)gridDim.x;
    issue_halo(t1 < a.ntiles ? t1 : t, 1);
  }
  int slot = 0;

  for (; t < a.ntiles; t += gridDim.x) {
    if constexpr (PRO) {
      if (has_tail) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G::NHW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G::NHW - 1) : "memory");
      fix_tile(t, slot);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      wait_tile();  // halo of tile t is in LDS for every wave; every wave is done with the slot of the tile before it
    }
    {
      const int t2 = t + 2 * (int)gridDim.x;
      issue_halo(t2 < a.ntiles ? t2 : t, (slot + 2) % 3);
    }
    const unsigned char* sH = smem + slot * G::SLOT;

    ffa_f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      ffa_u32x4 bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int noff = ((nt >> 1) * G::IW + (nt & 1) * 16) * G::PIXB;  // immediate
        if constexpr (CI == 32)
          bf[nt] = *reinterpret_cast<const ffa_u32x4*>(sH + fb[ks % 3] + (ks / 3) * G::ROWB + noff);
        else
          bf[nt] = *reinterpret_cast<const ffa_u32x4*>(sH + fb[ks] + noff);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ffa_bf16x8, wA[mt][ks]),
                                                                __builtin_bit_cast(ffa_bf16x8, bf[nt]), acc[mt][nt], 0, 0, 0);
    }

    // ---- epilogue of tile t: lane (col, kg) owns pixel (row, 16 * (nt & 1) + col) and CPL consecutive channels ----
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
    T* out = static_cast<T*>(a.out);
    const int c0 = CPL * kg;
    float st[STATS ? 2 * CPL : 1];
    if constexpr (STATS) {
#pragma unroll
      for (int i = 0; i < 2 * CPL; ++i) st[i] = 0.f;
    }
    if constexpr (POOL) {
      // 2x2 sums: fragments (nt, nt + 2) hold vertically adjacent rows of the same columns, lanes col ^ 1 the
      // horizontal neighbour; even columns write the pooled pixel
      const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
      for (int np = 0; np < NT; np += 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int nt = np + h;  // rows row0 + (np >> 1), + 1; column half h
          float v[CPL];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float s2 = acc[mt][nt][i] + acc[mt][nt + 2][i];
              s2 += __shfl_xor(s2, 1, 64);
              v[4 * mt + i] = s2;
            }
          const int oy = oy0 + row0 + (np >> 1), ox = ox0 + 16 * h + col;
          if ((col & 1) == 0 && oy < a.H && ox < a.W && c0 < a.Co) {
            T* dst = out + ((size_t)(b * Hp + (oy >> 1)) * Wp + (ox >> 1)) * (size_t)a.Co + c0;
            if constexpr (MT == 2) {
              float v8[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) v8[i] = v[i];
              ffa_store8<T>(dst, v8);
            } else {
              uint2 u;
              u.x = ffa_pack_bf16x2(v[0], v[1]);
              u.y = ffa_pack_bf16x2(v[2], v[3]);
              *reinterpret_cast<uint2*>(dst) = u;
            }
          }
        }
      }
    } else {
      const T* res = static_cast<const T*>(a.res);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy0 + row0 + (nt >> 1), ox = ox0 + 16 * (nt & 1) + col;
        if (oy >= a.H || ox >= a.W || c0 >= a.Co) continue;
        const size_t pix = ((size_t)(b * a.H + oy) * a.W + ox) * (size_t)a.Co + c0;
        float v[CPL];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[4 * mt + i] = acc[mt][nt][i];
        if (a.bias) {
#pragma unroll
          for (int i = 0; i < CPL; ++i) v[i] += a.bias[c0 + i];
        }
        if (res) {
#pragma unroll
          for (int i = 0; i < CPL; ++i) v[i] += ffa_load_elem<T>(res + pix + i);
        }
        if (a.relu) {
#pragma unroll
          for (int i = 0; i < CPL; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if constexpr (MT == 2) {
          float v8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = v[i];
          ffa_store8<T>(out + pix, v8);
        } else {
          uint2 u;
          u.x = ffa_pack_bf16x2(v[0], v[1]);
          u.y = ffa_pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(out + pix) = u;
        }
        if constexpr (STATS) {
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            const float rr = ffa_bf16_bits_to_f32(ffa_f32_to_bf16_bits(v[i]));
            st[i] += rr;
            st[CPL + i] = __builtin_fmaf(rr, rr, st[CPL + i]);  // explicit: all conv kernels round alike
          }
        }
      }
    }
    if constexpr (STATS) {
      // sum over the 16 lanes that share kg (transposing reduction while a lane holds more than one value): afterwards
      // lane col holds entry col >> (4 - log2(2 * CPL)) ... of [sums | sums of squares]; the waves meet through LDS
      constexpr int NV = 2 * CPL;  // 8 or 16 values per lane
#pragma unroll
      for (int bit = 3; bit >= 0; --bit) {
        const int n = NV >> (3 - bit);  // values a lane still holds at this step (16, 8, 4, 2 or 8, 4, 2, 1)
        if (n >= 2) {
          const bool up = (col >> bit) & 1;
#pragma unroll
          for (int j = 0; j < NV / 2; ++j) {
            if (j < n / 2) {
              float lo = st[j], hi = st[j + n / 2];
              asm volatile("" : "+v"(lo), "+v"(hi));
              const float keep = up ? hi : lo;
              const float send = up ? lo : hi;
              st[j] = keep + __shfl_xor(send, 1 << bit, 64);
            }
          }
        } else {  // one value left (NV == 8): plain butterfly over the last lane bit, both lanes end with the total
          st[0] += __shfl_xor(st[0], 1 << bit, 64);
        }
      }
      // entry index of st[0] in [sums (CPL) | squares (CPL)]: NV == 16: col (bits high to low); NV == 8: col >> 1
      float* red = reinterpret_cast<float*>(smem + G::RED_OFF);
      red[wave * 64 + lane] = st[0];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 64) {
        const float tot = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
        const int cl = tid & 15, kq = tid >> 4;
        const int e = (NV == 16) ? cl : (cl >> 1);
        const int which = e / CPL, ch = CPL * kq + e % CPL;
        if ((NV == 16 || (cl & 1) == 0) && ch < a.Co) a.stats[((size_t)t * 2 + which) * a.Co + ch] = tot;
      }
      // red is rewritten only after the next tile's barrier (wait_tile): every wave has left this read by then
    }
    slot = (slot + 1) % 3;
  }
  // trailing DMA (fills for tiles that do not exist) must not outlive the block's LDS allocation
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// host side

template <int CI, int MT>
static int thin_launch(const ThinArgs& a, int up, int pool, hipStream_t stream) {
  // two blocks per CU; FFA_THIN_GRID caps the grid (the tests force several tiles per persistent block on small tensors)
  const char* pg = getenv("FFA_THIN_GRID");
  int cap = pg ? atoi(pg) : 512;
  if (cap < 1) cap = 1;
  const int grid = a.ntiles < cap ? a.ntiles : cap;
  const bool st = a.stats != nullptr, pro = a.pro_sc != nullptr;
#define FFA_THIN_GO(UP_, POOL_, ST_, PRO_)                                                                         \
  hipLaunchKernelGGL((conv3x3_thin_kernel<CI, MT, UP_, POOL_, ST_, PRO_>), dim3(grid), dim3(256), 0, stream, a); \
  return ffa_check_launch("conv3x3_thin");
  if (pool) { FFA_THIN_GO(false, true, false, false) }
  if (pro) {
    if (up) {
      if (st) { FFA_THIN_GO(true, false, true, true) }
      FFA_THIN_GO(true, false, false, true)
    }
    if (st) { FFA_THIN_GO(false, false, true, true) }
    FFA_THIN_GO(false, false, false, true)
  }
  if (up) {
    if (st) { FFA_THIN_GO(true, false, true, false) }
    FFA_THIN_GO(true, false, false, false)
  }
  if (st) { FFA_THIN_GO(false, false, true, false) }
  FFA_THIN_GO(false, false, false, false)
#undef FFA_THIN_GO
}

// 1 when a layer qualifies for the thin kernel: bf16, 3x3 stride 1, at most 32 stored input channels and 32 rows
extern "C" int ffa_thin_eligible(int dtype, int kh, int kw, int stride, int rows_real, int ci_pitch) {
  return dtype == FFA_BF16 && kh == 3 && kw == 3 && stride == 1 && (ci_pitch == 16 || ci_pitch == 32) &&
         rows_real > 0 && rows_real <= 32;
}

extern "C" long long ffa_thin_pack_bytes(int co_rows, int ci_pitch) {
  const int mt = co_rows / 16, ks = (ci_pitch == 32) ? 9 : 5;
  return (long long)mt * ks * 1024;
}

extern "C" long long ffa_thin_stat_rows(int B, int H, int W, int ci_pitch) {
  const int th = (ci_pitch == 32) ? 8 : 16;
  return (long long)B * ffa_cdiv(W, 32) * ffa_cdiv(H, th);
}

// in: [B][H][W][Ci] (up: [B][H/2][W/2][Ci]); out / residual: [B][H][W][Co] (pool: out [B][H/2][W/2][Co]).
// co_rows = 16 or 32 (the packed operand's rows).  up: the input is nearest_x2 of `in`; pool: the output is 2x2
// sum-pooled (no bias / residual / relu / statistics then).
// pro_scale / pro_shift (both or neither, [Ci] f32): the convolution reads relu(in * scale[c] + shift[c]).
extern "C" int ffa_thin_conv3x3_pro(const void* in, const void* w_thin, const float* bias, const void* residual,
                                    void* out, float* stat_partials, const float* pro_scale, const float* pro_shift, int B,
                                    int H, int W, int Ci, int Co, int co_rows, int relu, int up, int pool,
                                    hipStream_t stream);
extern "C" int ffa_thin_conv3x3(const void* in, const void* w_thin, const float* bias, const void* residual, void* out,
                                float* stat_partials, int B, int H, int W, int Ci, int Co, int co_rows, int relu, int up,
                                int pool, hipStream_t stream) {
  return ffa_thin_conv3x3_pro(in, w_thin, bias, residual, out, stat_partials, nullptr, nullptr, B, H, W, Ci, Co, co_rows,
                              relu, up, pool, stream);
}
extern "C" int ffa_thin_conv3x3_pro(const void* in, const void* w_thin, const float* bias, const void* residual,
                                    void* out, float* stat_partials, const float* pro_scale, const float* pro_shift, int B,
                                    int H, int W, int Ci, int Co, int co_rows, int relu, int up, int pool,
                                    hipStream_t stream) {
  FFA_REQUIRE((pro_scale == nullptr) == (pro_shift == nullptr) && !(pool && pro_scale),
              "thin conv: prologue needs scale and shift (and is not part of the pooled form)");
  FFA_REQUIRE(in && w_thin && out, "thin conv: null pointer");
  FFA_REQUIRE(B > 0 && H > 0 && W > 0 && (Ci == 16 || Ci == 32) && (co_rows == 16 || co_rows == 32) && Co % 8 == 0,
              "thin conv: bad dims (Ci %d, Co %d, rows %d)", Ci, Co, co_rows);
  FFA_REQUIRE(!(up || pool) || (H % 2 == 0 && W % 2 == 0), "thin conv: the x2 forms need even sizes");
  FFA_REQUIRE(!pool || (!bias && !residual && !relu && !stat_partials), "thin conv: the pooled form has a plain epilogue");
  FFA_REQUIRE(!up || !residual, "thin conv: no residual input in the upsampled form");
  FFA_REQUIRE((long long)B * H * W * Ci * 2 < (1LL << 40), "thin conv: input too large");
  ThinArgs a;
  a.in = in; a.w = w_thin; a.out = out; a.bias = bias; a.res = residual; a.stats = stat_partials;
  a.pro_sc = pro_scale; a.pro_sh = pro_shift;
  a.B = B; a.H = H; a.W = W; a.Co = Co; a.relu = relu;
  const int th = (Ci == 32) ? 8 : 16;
  a.tiles_x = ffa_cdiv(W, 32);
  a.tiles_y = ffa_cdiv(H, th);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  if (Ci == 32) return (co_rows == 32) ? thin_launch<32, 2>(a, up, pool, stream) : thin_launch<32, 1>(a, up, pool, stream);
  return (co_rows == 32) ? thin_launch<16, 2>(a, up, pool, stream) : thin_launch<16, 1>(a, up, pool, stream);
}

// ------------------------------------------------------------------------------------------------
// weight packing: dst[mt][k-step][lane][8 bf16], lane = (row = lane & 15, kg = lane >> 4)
//   tile row 4*q + i of tile mt  <->  output channel 8*q + 4*mt + i (two tiles) / 4*q + i (one tile)
//   32 channels: k-step = tap, elements = channels 8*kg .. 8*kg+7
//   16 channels: k-step = taps (2*ks, 2*ks + 1); kg >> 1 picks the tap (tap 9 = zeros), kg & 1 the channel half
// src element (row, ch, tap) at src[row * s_row + ch * s_ch + tap]; flip mirrors the taps (dgrad operand).

struct ThinPackArgs {
  const float* src;
  void* dst;
  const float* scale;
  long long s_row, s_ch;
  int rows, chs;  // valid rows / channels in src
  int mt, ci, flip;
};

__device__ __forceinline__ void thin_pack_piece(const ThinPackArgs& p, int idx) {
  const int ks_n = (p.ci == 32) ? 9 : 5;
  const int lane = idx % 64;
  const int ks = (idx / 64) % ks_n;
  const int mt = idx / (64 * ks_n);
  const int r = lane & 15, kg = lane >> 4;
  const int q = r >> 2, i = r & 3;
  const int row = (p.mt == 2) ? (8 * q + 4 * mt + i) : (4 * q + i);
  int tap, ch0;
  if (p.ci == 32) {
    tap = ks;
    ch0 = 8 * kg;
  } else {
    tap = 2 * ks + (kg >> 1);
    ch0 = 8 * (kg & 1);
  }
  float v[8];
  const bool ok = row < p.rows && tap < 9;
  const float sc = (ok && p.scale) ? p.scale[row] : 1.f;
  const int st = p.flip ? 8 - tap : tap;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = ch0 + j;
    v[j] = (ok && ch < p.chs) ? p.src[(long long)row * p.s_row + (long long)ch * p.s_ch + st] * sc : 0.f;
  }
  ffa_store8<ffa_bf16>(static_cast<ffa_bf16*>(p.dst) + (long long)idx * 8, v);
}

__global__ void thin_pack_kernel(ThinPackArgs p) {
  const int total = p.mt * ((p.ci == 32) ? 9 : 5) * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) thin_pack_piece(p, i);
}

__global__ void thin_pack_batched_kernel(const ThinPackArgs* __restrict__ descs) {
  const ThinPackArgs p = descs[blockIdx.y];
  const int total = p.mt * ((p.ci == 32) ? 9 : 5) * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) thin_pack_piece(p, i);
}

static int thin_pack_fill(ThinPackArgs& p, const float* w_oihw, const float* scale, void* dst, int O, int I, int transpose,
                          int co_rows, int ci_pitch) {
  FFA_REQUIRE(w_oihw && dst, "thin pack: null pointer");
  FFA_REQUIRE((co_rows == 16 || co_rows == 32) && (ci_pitch == 16 || ci_pitch == 32), "thin pack: rows %d / pitch %d",
              co_rows, ci_pitch);
  memset(&p, 0, sizeof(p));
  p.src = w_oihw;
  p.dst = dst;
  p.scale = scale;
  if (!transpose) {
    p.rows = O; p.chs = I;
    p.s_row = (long long)I * 9;
    p.s_ch = 9;
    p.flip = 0;
  } else {
    p.rows = I; p.chs = O;
    p.s_row = 9;
    p.s_ch = (long long)I * 9;
    p.flip = 1;
  }
  FFA_REQUIRE(p.rows <= co_rows && p.chs <= ci_pitch, "thin pack: padded dims smaller than the tensor");
  p.mt = co_rows / 16;
  p.ci = ci_pitch;
  return FFA_OK;
}

extern "C" int ffa_thin_pack(const float* w_oihw, const float* scale, void* dst, int O, int I, int transpose, int co_rows,
                             int ci_pitch, hipStream_t stream) {
  ThinPackArgs p;
  const int rc = thin_pack_fill(p, w_oihw, scale, dst, O, I, transpose, co_rows, ci_pitch);
  if (rc != FFA_OK) return rc;
  hipLaunchKernelGGL(thin_pack_kernel, dim3(5), dim3(256), 0, stream, p);
  return ffa_check_launch("thin_pack");
}

extern "C" int ffa_thin_pack_desc_bytes(void) { return (int)sizeof(ThinPackArgs); }

extern "C" int ffa_thin_pack_desc_fill(void* host_desc, const float* w_oihw, const float* scale, void* dst, int O, int I,
                                       int transpose, int co_rows, int ci_pitch) {
  FFA_REQUIRE(host_desc, "thin pack: null descriptor");
  ThinPackArgs p;
  const int rc = thin_pack_fill(p, w_oihw, scale, dst, O, I, transpose, co_rows, ci_pitch);
  if (rc != FFA_OK) return rc;
  memcpy(host_desc, &p, sizeof(p));
  return FFA_OK;
}

extern "C" int ffa_thin_pack_batched(const void* descs_device, int n, hipStream_t stream) {
  FFA_REQUIRE(descs_device && n > 0 && n <= 65535, "thin pack: bad descriptor table");
  hipLaunchKernelGGL(thin_pack_batched_kernel, dim3(5, n), dim3(256), 0, stream,
                     static_cast<const ThinPackArgs*>(descs_device));
  return ffa_check_launch("thin_pack_batched");
}
