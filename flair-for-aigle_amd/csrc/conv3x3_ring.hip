// 3x3 stride-1 pad-1 convolution (forward and stride-1 dgrad) on MFMA for gfx950 -- the "ring" kernel.
//
// Replaces the cuDNN / ATen conv2d of the ResNet-34 BasicBlocks and U-Net decoder blocks reached through
// segmentation_models_pytorch from flair_hub/models/monotemp_model.py:68-92 (called at
// flair_hub/models/flair_model.py:376 and :417-419), for the layers with >= 64 output channels and whole 64-byte
// groups of input channels -- the MFMA-bound part of the step.
//
// What differs from conv_igemm.hip (which stays for strided / 1x1 / 7x7 / thin layers):
//   * the weight operand -- two thirds of the bytes a block stages -- never touches a VGPR: it is packed in global
//     memory as the exact LDS image and streamed by LDS-DMA (global_load_lds_dwordx4) into a 3-slot ring, one slot
//     per kernel row of a 64-byte channel chunk, two phases ahead of its use, tracked with a counted vmcnt;
//   * ONE barrier per phase (a phase = one kernel row x 64 bytes of channels = 6 k-steps of MFMA per wave) instead
//     of two per 32-byte k-step chunk;
//   * the input halo (one third of the bytes) keeps the register path -- zero padding, and optionally a fused
//     BatchNorm + ReLU of the producing layer (relu(x * sc[c] + sh[c]), "normalise on load"), are applied in
//     registers -- and is double-buffered in LDS, requested a whole chunk (three phases) before it is stored;
//   * blocks are persistent: the phase stream runs across pixel tiles, so a tile's epilogue overlaps the loads of
//     the next one.
//
// GEMM orientation as in conv_igemm.hip: D[co][pixel] = W[co][k] * X[k][pixel]; a wave owns 64 output channels x
// NT*32 pixels; lanes end up with runs of 8 consecutive channels of one pixel (16-byte NHWC stores).
//
// LDS images
//   ring slot [wco][tap s][k-step][64 rows in fragment order][32 B], the two 16-byte halves of a row swapped when
//             bit 3 of the row is set  -> conflict-free ds_read_b128, linear (lane-order) DMA destination
//   halo      [k-step plane][halo pixel][32 B], halves swapped when bit 3 of the pixel's x is set; plane stride
//             = 64 mod 128 bytes -> conflict-free ds_read_b128 for 32 pixels of a row and conflict-free ds_write_b128
#include "ffa_common.h"
#include <hip/hip_ext.h>

#include <stdlib.h>

// Developer instrumentation (never built by flairhip/build.py; tools/ring_trace.py builds a second library with
// -DFFA_RING_TRACE=1): wave 0 of every block records its lifetime in shader cycles (s_memtime) and in 100 MHz
// real-time ticks (s_memrealtime: their ratio is the clock the chip really holds under this kernel), and the cycles
// it spent in the prologue, waiting at phase-end synchronisations and in tile epilogues.
#ifndef FFA_RING_TRACE
#define FFA_RING_TRACE 0
#endif
#ifndef FFA_RING_SCHED
#define FFA_RING_SCHED 0  // 1: one fragment read behind every MFMA (measured slower, kept for A/B builds)
#endif
#if FFA_RING_TRACE
__device__ long long ffa_ring_trace_buf[1024 * 8];
#define RT_NOW() ((long long)__builtin_readcyclecounter())
#define RT_ADD(acc_, t0_) acc_ += RT_NOW() - (t0_)
#else
#define RT_NOW() 0ll
#define RT_ADD(acc_, t0_)
#endif

struct Ring3Args {
  const void* in;
  const void* w;
  void* out;
  const float* bias;    // [Co] or null
  float* stats;         // [npt][2][Co] per-tile channel sums / sums of squares of the stored output, or null
  const void* res;      // same layout as out, or null
  const float* pro_sc;  // PRO kernels: the input is relu(in * pro_sc[c] + pro_sh[c]), evaluated while staging
  const float* pro_sh;
  int B, H, W;          // stride 1, pad 1: output H x W
  int Ci, Co;           // stored channel pitches (elements)
  int relu;
  int nchunks;          // Ci * sizeof(T) / 64
  int tiles_x, tiles_y, npt, ncb;
  long long cb64_stride;  // bytes between the operands of consecutive 64-row groups
};

template <typename T>
struct RingMma;
template <>
struct RingMma<ffa_bf16> {
  static constexpr int PER = 1;
  static __device__ __forceinline__ void run(const ffa_u32x4& a, const ffa_u32x4& b, ffa_f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ffa_bf16x8, a), __builtin_bit_cast(ffa_bf16x8, b),
                                                c, 0, 0, 0);
  }
};
template <>
struct RingMma<float> {
  static constexpr int PER = 4;
  static __device__ __forceinline__ void run(const ffa_u32x4& a, const ffa_u32x4& b, ffa_f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

template <int WCO, int WPX, int NT, int TH, int TW>
struct RingGeom {
  static constexpr int MT = 2;
  static constexpr int KS = 2;  // 32-byte k-steps per chunk
  static constexpr int NW = WCO * WPX;
  static constexpr int NTHR = 64 * NW;
  static constexpr int BCO = 64 * WCO;
  static constexpr int NPX = TH * TW;
  static constexpr int IH = TH + 2;
  static constexpr int IW = (TW == 32) ? 34 : 24;  // TW == 16: a fragment spans two rows, IW = 0 mod 8 keeps reads conflict-free
  static constexpr int PLANE_RAW = IH * IW * 32;
  static constexpr int PLANE = PLANE_RAW + ((64 - PLANE_RAW % 128) + 128) % 128;  // = 64 mod 128
  static constexpr int HBUF = KS * PLANE;
  static constexpr int SLAB64 = 3 * KS * 64 * 32;  // one kernel row of one 64-row group: 12 KB
  static constexpr int SLOT = WCO * SLAB64;
  static constexpr int NSLOT = 3;
  static constexpr int NWI = SLOT / 1024 / NW;  // DMA wave-instructions per wave per phase
  static constexpr int H_PIECES = IH * IW * 4;
  static constexpr int NHP = (H_PIECES + NTHR - 1) / NTHR;
  static constexpr int RING_OFF = 0;
  static constexpr int HALO_OFF = NSLOT * SLOT;
  // the statistics epilogue reduces through ring slot 2: a tile's last phase is always kernel row 2, and the next
  // DMA into that slot is only issued at the top of the next tile's first phase
  static constexpr int RED_OFF = RING_OFF + 2 * SLOT;
  static constexpr int RED_BYTES = NW * 64 * 2 * 4;
  static constexpr int LDS_BYTES = HALO_OFF + 2 * HBUF;
  static_assert(RED_BYTES <= SLOT, "statistics scratch must fit a ring slot");
  static_assert(NPX == WPX * NT * 32, "pixel tile must be covered by the pixel waves");
  static_assert(TW == 32 || TW == 16, "tile width");
  static_assert((SLOT / 1024) % NW == 0, "every wave issues the same number of DMA instructions");
  static_assert(PLANE % 128 == 64 && PLANE % 16 == 0, "plane stride");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// phase-end synchronisation: this wave's DMA of the NEXT phase has landed when at most its newest NWI vector-memory
// operations (the DMA of the phase after that, issued at the top of this phase) are outstanding; its halo stores
// are in LDS (lgkmcnt); then the block meets.  One asm statement with a memory clobber: neither the compiler's own
// LDS accesses nor its loads move across it.
// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS at lds_base + lane * 16
// (lds_base wave-uniform).  Inline asm on purpose: with __builtin_amdgcn_global_load_lds in the kernel hipcc
// (ROCm 7.2) stops counting lgkmcnt and drains it to 0 in front of every MFMA step (532 of 789 waits were
// lgkmcnt(0); without the builtin they are counted), which stalls every step on the fragment reads just issued for
// two steps later.  The DMA is invisible to the compiler: its completion is waited for by hand (ring_phase_sync),
// the compiler's own vmcnt waits can only become stricter through the extra entries in the queue.  M0 (the LDS
// destination base) is written and restored inside the statement.
__device__ __forceinline__ void ring_dma16(const unsigned char* src, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_base)
      : "memory");
}

template <int N>
__device__ __forceinline__ void ring_phase_sync() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(N) : "memory");
}
// the same without the LDS wait: fragment reads requested for later steps stay in flight across the barrier
template <int N>
__device__ __forceinline__ void ring_phase_sync_nolgkm() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(N) : "memory");
}
__device__ __forceinline__ void ring_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename T, int WCO, int WPX, int NT, int TH, int TW, int OCC, bool PRO>
__global__ void __launch_bounds__(64 * WCO * WPX, OCC) conv3x3_ring_kernel(Ring3Args a) {
  using G = RingGeom<WCO, WPX, NT, TH, TW>;
  constexpr int EB = ElemTraits<T>::kBytes;
  constexpr int EPF = ElemTraits<T>::kPerFrag;
  constexpr int MT = G::MT, KS = G::KS;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave / WPX;
  const int wpx = wave % WPX;
  const int rho = lane & 31;
  const int half = lane >> 5;
  const int NC = a.nchunks;
  const int PT = NC * 3;  // phases per tile
  const int total_vb = ((a.npt + 7) / 8) * 8 * a.ncb;

  // ---- per-lane LDS read addresses (everything else is an immediate) ----
  const int a0 = G::RING_OFF + wco * G::SLAB64 + rho * 32 + ((half ^ ((rho >> 3) & 1)) * 16);
  int bB[NT][3];  // includes the base of the CURRENT halo buffer
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = wpx * (NT * 32) + nt * 32 + rho;
    const int py = n / TW, px = n % TW;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int hx = px + s;
      bB[nt][s] = G::HALO_OFF + (py * G::IW + hx) * 32 + ((half ^ ((hx >> 3) & 1)) * 16);
    }
  }

  // ---- halo staging geometry (tile independent part) ----
  static_assert(G::NTHR % 4 == 0, "a thread keeps its 16-byte slot of the 64-byte chunk");
  const int jj = tid & 3;  // piece of the 64-byte chunk: k-step jj >> 1, half jj & 1
  int hl[G::NHP];          // LDS byte offset inside a halo buffer
  int hyx[G::NHP];         // (hy << 16) | hx, or -1 past the last piece
#pragma unroll
  for (int k = 0; k < G::NHP; ++k) {
    const int q = (tid >> 2) + k * (G::NTHR / 4);
    const int hy = q / G::IW, hx = q % G::IW;
    hl[k] = (jj >> 1) * G::PLANE + q * 32 + (((jj & 1) ^ ((hx >> 3) & 1)) * 16);
    hyx[k] = (q < G::IH * G::IW) ? ((hy << 16) | hx) : -1;
  }

  struct TileId {
    int pt, cb, b, oy0, ox0;
  };
  auto decode = [&](int vb) {
    TileId t;
    const int j = vb >> 3;
    t.pt = (j / a.ncb) * 8 + (vb & 7);  // the co blocks of a pixel tile share an XCD's L2 (speed only)
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
  const int pix_bytes = a.Ci * EB;
  int hoff[G::NHP];  // byte offset of the piece from the input base (chunk 0), -1 = zero fill
  auto halo_offsets = [&](const TileId& t) {
#pragma unroll
    for (int k = 0; k < G::NHP; ++k) {
      const int hy = hyx[k] >> 16, hx = hyx[k] & 0xffff;
      const int vy = t.oy0 - 1 + hy, vx = t.ox0 - 1 + hx;
      const bool ok = hyx[k] >= 0 && vy >= 0 && vx >= 0 && vy < a.H && vx < a.W;
      hoff[k] = ok ? (((t.b * a.H + vy) * a.W + vx) * pix_bytes + jj * 16) : -1;
    }
  };

  const unsigned char* in_b = static_cast<const unsigned char*>(a.in);
  const unsigned char* w_all = static_cast<const unsigned char*>(a.w);

  ffa_u32x4 hreg[G::NHP];
  float psc[PRO ? EPF : 1], psh[PRO ? EPF : 1];

  // weight DMA of one phase: slab `ph` (chunk * 3 + kernel row) of co block `cb` -> ring slot `slot`
  auto issue_w = [&](int cb, int ph, int slot) {
#pragma unroll
    for (int i = 0; i < G::NWI; ++i) {
      const int ii = wave + i * G::NW;  // 1-KB piece of the slot (wave uniform)
      const int g64 = ii / (G::SLAB64 / 1024), pi = ii % (G::SLAB64 / 1024);
      const unsigned char* src = w_all + (long long)(cb * WCO + g64) * a.cb64_stride + (long long)ph * G::SLAB64 +
                                 pi * 1024 + lane * 16;
      ring_dma16(src, (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + G::RING_OFF + slot * G::SLOT +
                                                                                  ii * 1024));
    }
  };
  // halo loads of one chunk (unconditional: a padding piece reads offset 0 and is zeroed when stored)
  auto load_h = [&](int chunk) {
    const unsigned char* base = in_b + chunk * 64;
#pragma unroll
    for (int k = 0; k < G::NHP; ++k)
      hreg[k] = *reinterpret_cast<const ffa_u32x4*>(base + (unsigned)(hoff[k] >= 0 ? hoff[k] : 0));
    if constexpr (PRO) {
      const int c0 = chunk * (64 / EB) + jj * EPF;
#pragma unroll
      for (int e = 0; e < EPF; ++e) {
        psc[e] = a.pro_sc[c0 + e];
        psh[e] = a.pro_sh[c0 + e];
      }
    }
  };
  auto store_h = [&](int buf) {
    unsigned char* dst = smem + G::HALO_OFF + buf * G::HBUF;
#pragma unroll
    for (int k = 0; k < G::NHP; ++k) {
      if (k + 1 < G::NHP || G::H_PIECES % G::NTHR == 0 || hyx[k] >= 0) {
        ffa_u32x4 v = hreg[k];
        if constexpr (PRO) {
          if constexpr (EB == 2) {
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
          } else {
            v.x = __float_as_uint(fmaxf(__builtin_fmaf(__uint_as_float(v.x), psc[0], psh[0]), 0.f));
            v.y = __float_as_uint(fmaxf(__builtin_fmaf(__uint_as_float(v.y), psc[1], psh[1]), 0.f));
            v.z = __float_as_uint(fmaxf(__builtin_fmaf(__uint_as_float(v.z), psc[2], psh[2]), 0.f));
            v.w = __float_as_uint(fmaxf(__builtin_fmaf(__uint_as_float(v.w), psc[3], psh[3]), 0.f));
          }
        }
        if (hoff[k] < 0) v = ffa_u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<ffa_u32x4*>(dst + hl[k]) = v;
      }
    }
  };

  int vb = next_valid(blockIdx.x);
  if (vb >= total_vb) return;
  TileId cur = decode(vb);
#if FFA_RING_TRACE
  const long long rt_t0 = RT_NOW();
  const long long rt_r0 = (long long)__builtin_amdgcn_s_memrealtime();
  long long rt_sync = 0, rt_epi = 0, rt_pro = 0, rt_phases = 0;
#endif

  // ---- prologue of the block's first tile (the only exposed one) ----
  halo_offsets(cur);
  load_h(0);
  issue_w(cur.cb, 0, 0);
  issue_w(cur.cb, 1, 1);
  store_h(0);
  ring_phase_sync<0>();  // slabs 0 and 1 have landed, halo chunk 0 is stored
  RT_ADD(rt_pro, rt_t0);
  int hb = 0;  // halo buffer of the current chunk
  int hd = G::HBUF;  // byte distance from the current halo buffer to the other one

  // Fragment pipeline: the operands of MFMA step t+2 are requested while step t is multiplied (three register sets),
  // across phase, chunk and tile boundaries -- a wave alone on its SIMD keeps the matrix pipe fed through the LDS
  // latency, so the other block of the CU covers this block's epilogues instead of both crawling.  That needs the
  // next phase's LDS images visible two steps before a phase ends: the phase synchronisation sits in the MIDDLE of
  // a phase (behind step SYNC_AT), the weight DMA of phase p+2 is issued right behind it.
  constexpr int NST = 3 * KS;   // MFMA steps per phase: (tap s, k-step), 6
  constexpr int SYNC_AT = 2;
  static_assert(NST % 3 == 0, "the fragment set of a step is (step % 3) in every phase");
  ffa_u32x4 fa[3][MT], fb[3][NT];
  int bBn[NT];  // fragment base of tap 0 in the OTHER halo buffer (the next chunk's rows, read from kernel row 2)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bBn[nt] = bB[nt][0] + hd;
#pragma unroll
  for (int t = 0; t < 2; ++t) {  // steps 0 and 1 of the first phase
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      fa[t][mt] = *reinterpret_cast<const ffa_u32x4*>(smem + a0 + t * 2048 + mt * 1024);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      fb[t][nt] = *reinterpret_cast<const ffa_u32x4*>(smem + bB[nt][0] + t * G::PLANE);
  }

  ffa_f32x16 acc[MT][NT];

  while (true) {
    const int nvb = next_valid(vb + gridDim.x);
    const bool has_next = nvb < total_vb;
    const TileId nxt = decode(has_next ? nvb : vb);

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    for (int c = 0; c < NC; ++c) {
      // the halo requested during this chunk: the next chunk of this tile, or chunk 0 of the next tile
      const bool last_c = (c + 1 == NC);
      const bool hvalid = !last_c || has_next;
      if (last_c && has_next) halo_offsets(nxt);
      const int hchunk = last_c ? 0 : c + 1;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        // weight slab two phases ahead: this tile's, the next tile's first ones, or (no next tile) a re-read of
        // this tile's last slab, which lands in a slot nobody reads again (keeps every wave's DMA count uniform)
        int wcb_ = cur.cb, wph_ = c * 3 + r + 2;
        if (wph_ >= PT) {
          if (has_next) {
            wcb_ = nxt.cb;
            wph_ -= PT;
          } else {
            wph_ = PT - 1;
          }
        }
        const unsigned char* sA = smem + a0 + r * G::SLOT;
        const unsigned char* sAn = smem + a0 + ((r + 1) % 3) * G::SLOT;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          __builtin_amdgcn_sched_barrier(0);
          // ---- request the fragments of step st + 2 ----
          {
            const int t2 = st + 2, fbuf = t2 % 3;
            if (t2 < NST) {
              const int s1 = t2 / KS, k1 = t2 % KS;
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                fa[fbuf][mt] = *reinterpret_cast<const ffa_u32x4*>(sA + t2 * 2048 + mt * 1024);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                fb[fbuf][nt] =
                    *reinterpret_cast<const ffa_u32x4*>(smem + bB[nt][s1] + k1 * G::PLANE + r * G::IW * 32);
            } else {  // steps 0 / 1 of the next phase: ring slot r + 1, halo row r + 1 (or row 0 of the next chunk)
              const int t3 = t2 - NST;
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                fa[fbuf][mt] = *reinterpret_cast<const ffa_u32x4*>(sAn + t3 * 2048 + mt * 1024);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                if (r < 2)
                  fb[fbuf][nt] = *reinterpret_cast<const ffa_u32x4*>(smem + bB[nt][0] + t3 * G::PLANE +
                                                                      (r + 1) * G::IW * 32);
                else
                  fb[fbuf][nt] = *reinterpret_cast<const ffa_u32x4*>(smem + bBn[nt] + t3 * G::PLANE);
              }
            }
          }
          // ---- multiply step st ----
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) RingMma<T>::run(fa[st % 3][mt], fb[st % 3][nt], acc[mt][nt]);
          {
            constexpr int NM = MT * NT;
#if FFA_RING_SCHED == 0
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM * RingMma<T>::PER, 0);
#else
#pragma unroll
            for (int i = 0; i < NM; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, RingMma<T>::PER, 0);
              if (i < MT + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#endif
          }
          if (st == SYNC_AT) {
            __builtin_amdgcn_sched_barrier(0);
            // next chunk's halo into the other buffer (requested a whole chunk ago), then the phase meets: every
            // wave's DMA of the next phase has landed (issued a phase ago, so vmcnt(0) waits for nothing younger
            // than that), and every wave is done with the previous phase's ring slot -> refill it
            if (r == 2 && hvalid) store_h(hb ^ 1);
#if FFA_RING_TRACE
            const long long rt_s = RT_NOW();
            ++rt_phases;
#endif
            if (r == 2) ring_phase_sync<0>();  // halo stores must be in LDS
            else ring_phase_sync_nolgkm<0>();
            RT_ADD(rt_sync, rt_s);
            issue_w(wcb_, wph_, (r + 2) % 3);
            if (r == 0 && hvalid) load_h(hchunk);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next chunk reads the other halo buffer
      {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int s = 0; s < 3; ++s) bB[nt][s] += hd;
        }
        hb ^= 1;
        hd = -hd;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bBn[nt] = bB[nt][0] + hd;
      }
    }

    // ---- epilogue of `cur`: lane (rho, half) owns pixel n = wave px base + nt*32 + rho and, per g, 8 channels ----
#if FFA_RING_TRACE
    const long long rt_e = RT_NOW();
#endif
    {
      T* out = static_cast<T*>(a.out);
      const T* res = static_cast<const T*>(a.res);
      const int co_wave = cur.cb * G::BCO + wco * 64;
      constexpr int NCH = 32;
      float st[2 * NCH];
      const bool want_stats = a.stats != nullptr;
#pragma unroll
      for (int i = 0; i < 2 * NCH; ++i) st[i] = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = wpx * (NT * 32) + nt * 32 + rho;
        const int oy = cur.oy0 + n / TW, ox = cur.ox0 + n % TW;
        if (oy >= a.H || ox >= a.W) continue;
        const long long pix = ((long long)(cur.b * a.H + oy) * a.W + ox) * (long long)a.Co;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = co_wave + 16 * g + 8 * half;
          if (c0 >= a.Co) continue;
          float v[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[0][nt][4 * g + i];
            v[4 + i] = acc[1][nt][4 * g + i];
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
              const float rr = (EB == 2) ? ffa_bf16_bits_to_f32(ffa_f32_to_bf16_bits(v[i])) : v[i];
              st[g * 8 + i] += rr;
              st[NCH + g * 8 + i] = __builtin_fmaf(rr, rr, st[NCH + g * 8 + i]);  // explicit: all conv kernels round alike
            }
          }
        }
      }
      if (want_stats) {
        // transposing reduction over the 32 lanes of a half-wave (see conv_igemm.hip): afterwards lane rho holds two
        // entries of [sums | sums of squares], entry index = (bits of rho, high to low) * 2 + j
#pragma unroll
        for (int bit = 4; bit >= 0; --bit) {
          const int n = (2 * NCH) >> (4 - bit);
          const bool up = (rho >> bit) & 1;
#pragma unroll
          for (int j = 0; j < NCH; ++j) {
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
        ring_lds_sync();
        if (tid < 128 * WCO) {
          const int wc = tid >> 7, t7 = tid & 127;
          const int j = t7 & 1;
          const int ln = t7 >> 1;
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < WPX; ++w) t += red[((wc * WPX + w) * 64 + ln) * 2 + j];
          const int r5 = ln & 31, hf = ln >> 5;
          const int which = r5 >> 4;
          const int L = (r5 & 15) * 2 + j;
          const int c = cur.cb * G::BCO + wc * 64 + 16 * (L >> 3) + 8 * hf + (L & 7);
          if (c < a.Co) a.stats[((size_t)cur.pt * 2 + which) * a.Co + c] = t;
        }
        ring_lds_sync();  // red is reused by the next tile
      }
    }
    RT_ADD(rt_epi, rt_e);
    if (!has_next) break;
    vb = nvb;
    cur = nxt;
  }
#if FFA_RING_TRACE
  if (tid == 0 && blockIdx.x < 1024) {
    long long* o = ffa_ring_trace_buf + blockIdx.x * 8;
    o[0] = rt_t0;
    o[1] = RT_NOW();
    o[2] = rt_r0;
    o[3] = (long long)__builtin_amdgcn_s_memrealtime();
    o[4] = rt_sync;
    o[5] = rt_epi;
    o[6] = rt_pro;
    o[7] = rt_phases;
  }
#endif
  // the trailing DMA (re-reads of the last slab) must not outlive the block's LDS allocation
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// bf16 on v_mfma_f32_16x16x32_bf16 ("ring16", round 3): the same ring / halo / phase structure with the 16x16 MFMA
// shape, whose K = 32 is one whole 64-byte channel chunk of one tap -- half the accumulator traffic per FLOP of the
// 32x32x16 form, and the shape on which the chip holds the higher clock under load (MI355X_MICROARCH.md, DVFS item 7).
//   * a wave owns 64 output channels x NT*16 pixels: MT = 4 tiles of 16 rows, NT pixel fragments of 16 columns;
//     accumulator tile (mt, nt): lane (col = lane & 15, kg = lane >> 4) holds rows 4*kg .. 4*kg+3 of pixel col.  The
//     pack stores output channel 16*kg + 4*mt + i of a 64-channel group in LDS row mt*16 + 4*kg + i, so a lane ends
//     up with 16 CONSECUTIVE channels of its pixel (two 16-byte NHWC stores);
//   * operand fragments: lane (row or pixel = lane & 15, kg) reads the 16 bytes of channels 8*kg .. 8*kg+7.  Both LDS
//     images are plain 64-byte rows (one chunk of one row / pixel) whose 16-byte slot index is XORed with 2 when bit 2
//     of the row index (weights) or of the halo column (pixels) is set: conflict-free ds_read_b128 for 16 consecutive
//     rows / pixels at ANY alignment (all 16 columns x 3 taps checked by enumeration), conflict-free ds_write_b128,
//     and still the linear destination LDS-DMA needs (the pack applies the XOR in global memory);
//   * a phase (kernel row r of one chunk) = 3 steps (taps) of MT*NT MFMAs; fragments of step t+1 are requested while
//     step t multiplies; the phase synchronisation sits behind step 0, the DMA of phase p+2 right behind it.

// source of the padding pieces of an LDS-DMA halo fill (per-lane source addresses: a border pixel reads zeros)
__device__ __attribute__((aligned(16))) const unsigned int ffa_ring_zero16[4] = {0u, 0u, 0u, 0u};

template <int WCO, int WPX, int NT, int TH, int TW>
struct Ring16Geom {
  static constexpr int MT = 4;
  static constexpr int NW = WCO * WPX;
  static constexpr int NTHR = 64 * NW;
  static constexpr int BCO = 64 * WCO;
  static constexpr int NPX = TH * TW;
  static constexpr int IH = TH + 2;
  static constexpr int IW = TW + 2;
  static constexpr int ROWB = IW * 64;          // bytes of one halo row
  static constexpr int HBUF = IH * ROWB;
  static constexpr int TAPB = 64 * 64;          // one tap of one 64-row group: 4 KB
  static constexpr int SLAB64 = 3 * TAPB;       // one kernel row of one 64-row group: 12 KB
  static constexpr int SLOT = WCO * SLAB64;
  static constexpr int NSLOT = 3;
  static constexpr int NWI = SLOT / 1024 / NW;  // DMA wave-instructions per wave per phase
  static constexpr int H_PIECES = IH * IW * 4;
  static constexpr int NHP = (H_PIECES + NTHR - 1) / NTHR;
  static constexpr int RING_OFF = 0;
  static constexpr int HALO_OFF = NSLOT * SLOT;
  // statistics scratch = the start of ring slot 2 (tap 0 of the tile's last phase: consumed before that phase's
  // synchronisation, and the next DMA into the slot is only issued behind the next tile's first synchronisation)
  static constexpr int RED_OFF = RING_OFF + 2 * SLOT;
  static constexpr int RED_BYTES = NW * 64 * 2 * 4;
  static constexpr int LDS_BYTES = HALO_OFF + 2 * HBUF;
  static_assert(NPX == WPX * NT * 16, "pixel tile must be covered by the pixel waves");
  static_assert(TW == 32 || TW == 16, "tile width");
  static_assert((SLOT / 1024) % NW == 0, "every wave issues the same number of DMA instructions");
  static_assert(RED_BYTES <= TAPB, "statistics scratch must stay inside tap 0 of the slot");
  static_assert(NTHR % 4 == 0, "a thread keeps its 16-byte slot of the 64-byte chunk");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int WCO, int WPX, int NT, int TH, int TW, int OCC, bool PRO>
__global__ void __launch_bounds__(64 * WCO * WPX, OCC) conv3x3_ring16_kernel(Ring3Args a) {
  using G = Ring16Geom<WCO, WPX, NT, TH, TW>;
  using T = ffa_bf16;
  constexpr int MT = G::MT;
  __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave / WPX;
  const int wpx = wave % WPX;
  const int col = lane & 15;  // operand row (weights) / column (pixels) of a fragment; accumulator column
  const int kg = lane >> 4;   // channels 8*kg .. 8*kg+7 of a chunk; accumulator rows 4*kg .. 4*kg+3
  const int NC = a.nchunks;
  const int PT = NC * 3;  // phases per tile
  const int total_vb = ((a.npt + 7) / 8) * 8 * a.ncb;

  // ---- per-lane LDS read addresses (ring slot, tap, tile and kernel row are immediates) ----
  const int a0 = G::RING_OFF + wco * G::SLAB64 + col * 64 + ((kg ^ (((col >> 2) & 1) << 1)) * 16);
  int bB[NT][3];  // includes the base of the CURRENT halo buffer
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = wpx * (NT * 16) + nt * 16 + col;
    const int py = n / TW, px = n % TW;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int hx = px + s;
      bB[nt][s] = G::HALO_OFF + (py * G::IW + hx) * 64 + ((kg ^ (((hx >> 2) & 1) << 1)) * 16);
    }
  }

  // ---- halo staging geometry (tile independent part) ----
  // Halo piece tid + k * NTHR = pixel q = (tid >> 2) + k * NTHR / 4 of the halo tile, 16-byte slot tid & 3.  The halo is
  // filled by LDS-DMA: the destination of a wave instruction is linear (base + lane * 16), so the thread owns PHYSICAL
  // slot tid & 3 of its pixel and fetches the logical slot that belongs there (the XOR goes on the source address).
  const int jj = tid & 3;

  struct TileId {
    int pt, cb, b, oy0, ox0;
  };
  auto decode = [&](int vb) {
    TileId t;
    const int j = vb >> 3;
    t.pt = (j / a.ncb) * 8 + (vb & 7);  // the co blocks of a pixel tile share an XCD's L2 (speed only)
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
  const int pix_bytes = a.Ci * 2;
  int hoff[G::NHP];  // byte offset of the piece from the input base (chunk 0), -1 = zero fill
  auto halo_offsets = [&](const TileId& t) {
#pragma unroll
    for (int k = 0; k < G::NHP; ++k) {
      const int q = (tid >> 2) + k * (G::NTHR / 4);
      const int hy = q / G::IW, hx = q % G::IW;
      const int vy = t.oy0 - 1 + hy, vx = t.ox0 - 1 + hx;
      const bool ok = q < G::IH * G::IW && vy >= 0 && vx >= 0 && vy < a.H && vx < a.W;
      const int slot = jj ^ (((hx >> 2) & 1) << 1);
      hoff[k] = ok ? (((t.b * a.H + vy) * a.W + vx) * pix_bytes + slot * 16) : -1;
    }
  };

  const unsigned char* in_b = static_cast<const unsigned char*>(a.in);
  const unsigned char* w_all = static_cast<const unsigned char*>(a.w);

  auto issue_w = [&](int cb, int ph, int slot) {
#pragma unroll
    for (int i = 0; i < G::NWI; ++i) {
      const int ii = wave + i * G::NW;  // 1-KB piece of the slot (wave uniform)
      const int g64 = ii / (G::SLAB64 / 1024), pi = ii % (G::SLAB64 / 1024);
      const unsigned char* src = w_all + (long long)(cb * WCO + g64) * a.cb64_stride + (long long)ph * G::SLAB64 +
                                 pi * 1024 + lane * 16;
      ring_dma16(src, (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + G::RING_OFF + slot * G::SLOT +
                                                                                  ii * 1024));
    }
  };
  // LDS-DMA halo fill (no prologue): chunk `chunk` of the tile hoff[] describes -> halo buffer `buf`.  The tail
  // instruction runs with the lanes past the last piece masked off (they would write into the other buffer).
  auto dma_h = [&](int chunk, int buf) {
    const unsigned char* base = in_b + chunk * 64;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(ffa_ring_zero16);
#pragma unroll
    for (int k = 0; k < G::NHP; ++k) {
      const unsigned char* src = hoff[k] >= 0 ? base + (unsigned)hoff[k] : zero;
      const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(
          smem + G::HALO_OFF + buf * G::HBUF + (wave * 64 + k * G::NTHR) * 16);
      if (k + 1 < G::NHP || G::H_PIECES % G::NTHR == 0) {
        ring_dma16(src, dst);
      } else if (wave * 64 + k * G::NTHR < G::H_PIECES) {  // wave uniform
        if (tid + k * G::NTHR < G::H_PIECES) ring_dma16(src, dst);
      }
    }
  };
  // PRO ("normalise on load"): the convolution reads relu(in * pro_sc[c] + pro_sh[c]).  The raw halo arrives by DMA;
  // once this wave's own pieces have landed (its vmcnt wait) thread t rewrites, in LDS, the pieces of LOGICAL slot
  // t & 3 of its pixels -- the same 8 channels for every piece (one scale / shift load per chunk), and pieces that this
  // very wave's DMA wrote (pixel q is filled by lanes 4*(q % 64) .. +3) -- skipping the zero padding (hoff < 0: the
  // padding of the normalised tensor is zero, not relu(shift)).  Same fma / max / rounding as ffa_bn_apply.
  auto fix_h = [&](int chunk, int buf) {
    if constexpr (PRO) {
      float psc[8], psh[8];
      const int c0 = chunk * 32 + jj * 8;
      ffa_load8<float>(a.pro_sc + c0, psc);
      ffa_load8<float>(a.pro_sh + c0, psh);
      unsigned char* base = smem + G::HALO_OFF + buf * G::HBUF;
#pragma unroll
      for (int k = 0; k < G::NHP; ++k) {
        const int q = (tid >> 2) + k * (G::NTHR / 4);
        const int hx = q % G::IW;
        if ((k + 1 < G::NHP || G::H_PIECES % G::NTHR == 0 || q < G::IH * G::IW) && hoff[k] >= 0) {
          ffa_u32x4* ptr = reinterpret_cast<ffa_u32x4*>(base + q * 64 + ((jj ^ (((hx >> 2) & 1) << 1)) * 16));
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

  int vb = next_valid(blockIdx.x);
  if (vb >= total_vb) return;
  TileId cur = decode(vb);

  // ---- prologue of the block's first tile (the only exposed one) ----
  halo_offsets(cur);
  dma_h(0, 0);
  issue_w(cur.cb, 0, 0);
  issue_w(cur.cb, 1, 1);
  if constexpr (PRO) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fix_h(0, 0);
  }
  ring_phase_sync<0>();  // slabs 0 and 1 and halo chunk 0 are in LDS (and normalised)
  int hb = 0;
  int hd = G::HBUF;  // byte distance from the current halo buffer to the other one

  ffa_u32x4 fa[3][MT], fb[3][NT];  // fragment sets by step of the phase (two of them live at a time)
  int bBn[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bBn[nt] = bB[nt][0] + hd;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) fa[0][mt] = *reinterpret_cast<const ffa_u32x4*>(smem + a0 + mt * 1024);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) fb[0][nt] = *reinterpret_cast<const ffa_u32x4*>(smem + bB[nt][0]);

  ffa_f32x4 acc[MT][NT];

  while (true) {
    const int nvb = next_valid(vb + gridDim.x);
    const bool has_next = nvb < total_vb;
    const TileId nxt = decode(has_next ? nvb : vb);

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = ffa_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < NC; ++c) {
      const bool last_c = (c + 1 == NC);
      const bool hvalid = !last_c || has_next;
      if (last_c && has_next) halo_offsets(nxt);
      const int hchunk = last_c ? 0 : c + 1;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        int wcb_ = cur.cb, wph_ = c * 3 + r + 2;
        if (wph_ >= PT) {
          if (has_next) {
            wcb_ = nxt.cb;
            wph_ -= PT;
          } else {
            wph_ = PT - 1;
          }
        }
        const unsigned char* sA = smem + a0 + r * G::SLOT;
        const unsigned char* sAn = smem + a0 + ((r + 1) % 3) * G::SLOT;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
          __builtin_amdgcn_sched_barrier(0);
          // ---- request the fragments of the next step ----
          if (st < 2) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              fa[st + 1][mt] = *reinterpret_cast<const ffa_u32x4*>(sA + (st + 1) * G::TAPB + mt * 1024);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              fb[st + 1][nt] = *reinterpret_cast<const ffa_u32x4*>(smem + bB[nt][st + 1] + r * G::ROWB);
          } else {  // step 0 of the next phase: ring slot r + 1, halo row r + 1 (or row 0 of the next chunk)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[0][mt] = *reinterpret_cast<const ffa_u32x4*>(sAn + mt * 1024);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              if (r < 2) fb[0][nt] = *reinterpret_cast<const ffa_u32x4*>(smem + bB[nt][0] + (r + 1) * G::ROWB);
              else fb[0][nt] = *reinterpret_cast<const ffa_u32x4*>(smem + bBn[nt]);
            }
          }
          // ---- multiply step st ----
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ffa_bf16x8, fa[st][mt]),
                                                                    __builtin_bit_cast(ffa_bf16x8, fb[st][nt]),
                                                                    acc[mt][nt], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < MT * NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < MT + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if (st == 0) {
            __builtin_amdgcn_sched_barrier(0);
            // next chunk's halo into the other buffer (requested a whole chunk ago), then the phase meets: every
            // wave's DMA of the next phase has landed (issued a phase ago) and every wave is done with the previous
            // phase's ring slot -> refill it
            // vector-memory queue, oldest first -- at r = 0: W(p+1); at r = 1: W(p+1), then the NHP halo pieces issued
            // behind it (they may stay in flight for another phase); at r = 2: halo pieces, W(p+1): the halo is read
            // from step 2 of this phase on, so everything must have landed
            if (r == 1 && hvalid) {  // (no halo fill behind W(p+1) when this is the block's last chunk: plain wait)
              // exact count per wave: the tail instruction of the halo fill exists only in the waves that own pieces
              // of it (a wave that waited for one operation too few would publish a weight slab still in flight)
              constexpr bool TAIL = G::H_PIECES % G::NTHR != 0;
              if (!TAIL || wave * 64 + (G::NHP - 1) * G::NTHR < G::H_PIECES) ring_phase_sync_nolgkm<G::NHP>();
              else ring_phase_sync_nolgkm<G::NHP - 1>();
            } else if (PRO && r == 2) {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              // the tile whose halo this is: hoff[] already describes it (the next chunk of `cur`, or chunk 0 of `nxt`)
              if (hvalid) fix_h(hchunk, hb ^ 1);
              ring_phase_sync<0>();  // the rewritten pieces must be in LDS before the block meets
            } else {
              ring_phase_sync_nolgkm<0>();
            }
            issue_w(wcb_, wph_, (r + 2) % 3);
            if (r == 0 && hvalid) dma_h(hchunk, hb ^ 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next chunk reads the other halo buffer
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int s = 0; s < 3; ++s) bB[nt][s] += hd;
      }
      hb ^= 1;
      hd = -hd;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bBn[nt] = bB[nt][0] + hd;
    }

    // ---- epilogue of `cur`: lane (col, kg) owns pixel n = wave px base + nt*16 + col and 16 consecutive channels ----
    {
      T* out = static_cast<T*>(a.out);
      const T* res = static_cast<const T*>(a.res);
      const int co_lane = cur.cb * G::BCO + wco * 64 + 16 * kg;
      float st[32];  // [16 sums | 16 sums of squares] of this lane's channels
      const bool want_stats = a.stats != nullptr;
#pragma unroll
      for (int i = 0; i < 32; ++i) st[i] = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = wpx * (NT * 16) + nt * 16 + col;
        const int oy = cur.oy0 + n / TW, ox = cur.ox0 + n % TW;
        if (oy >= a.H || ox >= a.W) continue;
        const long long pix = ((long long)(cur.b * a.H + oy) * a.W + ox) * (long long)a.Co;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c0 = co_lane + 8 * h;
          if (c0 >= a.Co) continue;
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
          // rounded once: the stored words are also what the statistics are taken from (the values BatchNorm will read)
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
              st[16 + h * 8 + i] = __builtin_fmaf(rr[i], rr[i], st[16 + h * 8 + i]);  // explicit: all conv kernels round alike
            }
          }
        }
      }
      if (want_stats) {
        // transposing reduction over the 16 lanes that share kg: afterwards lane col holds entries 2*col and 2*col + 1
        // of [sums | sums of squares]
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
        ring_lds_sync();
        if (tid < 128 * WCO) {
          const int wc = tid >> 7, t7 = tid & 127;
          const int j = t7 & 1;
          const int ln = t7 >> 1;
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < WPX; ++w) t += red[((wc * WPX + w) * 64 + ln) * 2 + j];
          const int cl = ln & 15, kq = ln >> 4;
          const int which = cl >> 3;
          const int ch = cur.cb * G::BCO + wc * 64 + 16 * kq + (cl & 7) * 2 + j;
          if (ch < a.Co) a.stats[((size_t)cur.pt * 2 + which) * a.Co + ch] = t;
        }
        ring_lds_sync();  // red is reused by the next tile
      }
    }
    if (!has_next) break;
    vb = nvb;
    cur = nxt;
  }
  // the trailing DMA (re-reads of the last slab) must not outlive the block's LDS allocation
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// host side

struct RingPlan {
  int cfg;     // 0: none
  int th, tw;  // pixel tile
  int bco;     // output channels per block
};

// Tile configuration of a layer.  cfg 1: 64 co x 8x32 px, 4 waves, two blocks per CU; cfg 2: 64 co x 16x16 px,
// 4 waves; cfg 4: 128 co x 8x32 px, 8 waves.  (cfg 3 / 5 of round 2 -- 64 co x 128 px per wave -- measured slower and
// are gone; their numbers map to 4 / 1.)
static RingPlan ring_plan(int B, int H, int W, int co_rows) {
  RingPlan p;
  p.cfg = 0;
  p.th = p.tw = p.bco = 0;
  const char* force = getenv("FFA_RING_CFG");
  int cfg = force ? atoi(force) : -1;
  if (cfg < 0) {
    if (W < 32) cfg = 2;
    else cfg = 1;
  }
  if (cfg == 3) cfg = 4;
  if (cfg == 5) cfg = 1;
  if (cfg == 4 && (co_rows % 128 != 0 || W < 32)) cfg = (W < 32) ? 2 : 1;
  if (cfg == 1 && W < 32) cfg = 2;
  p.cfg = cfg;
  switch (cfg) {
    case 1: p.th = 8; p.tw = 32; p.bco = 64; break;
    case 2: p.th = 16; p.tw = 16; p.bco = 64; break;
    case 4: p.th = 8; p.tw = 32; p.bco = 128; break;
    default: p.cfg = 0; break;
  }
  return p;
}

template <int WCO, int WPX, int NT, int TH, int TW, int OCC>
static int ring16_launch(const Ring3Args& a, int grid, hipStream_t stream) {
  hipEvent_t ts, te;  // measurement session open (bench.py): the kernel's own duration from launch-attached events
  if (!a.pro_sc && ffa_ktime_next(WCO == 2 ? FFA_KT_RING16_128CO : (TW == 32 ? FFA_KT_RING16_8x32 : FFA_KT_RING16_16x16), &ts, &te)) {
    hipExtLaunchKernelGGL((conv3x3_ring16_kernel<WCO, WPX, NT, TH, TW, OCC, false>), dim3(grid), dim3(64 * WCO * WPX), 0,
                          stream, ts, te, 0, a);
    return ffa_check_launch("conv3x3_ring16");
  }
  if (a.pro_sc)
    hipLaunchKernelGGL((conv3x3_ring16_kernel<WCO, WPX, NT, TH, TW, OCC, true>), dim3(grid), dim3(64 * WCO * WPX), 0,
                       stream, a);
  else
    hipLaunchKernelGGL((conv3x3_ring16_kernel<WCO, WPX, NT, TH, TW, OCC, false>), dim3(grid), dim3(64 * WCO * WPX), 0,
                       stream, a);
  return ffa_check_launch("conv3x3_ring16");
}

template <typename T, int WCO, int WPX, int NT, int TH, int TW, int OCC>
static int ring_launch(const Ring3Args& a, int grid, hipStream_t stream) {
  if (a.pro_sc)
    hipLaunchKernelGGL((conv3x3_ring_kernel<T, WCO, WPX, NT, TH, TW, OCC, true>), dim3(grid), dim3(64 * WCO * WPX), 0,
                       stream, a);
  else
    hipLaunchKernelGGL((conv3x3_ring_kernel<T, WCO, WPX, NT, TH, TW, OCC, false>), dim3(grid), dim3(64 * WCO * WPX), 0,
                       stream, a);
  return ffa_check_launch("conv3x3_ring");
}

// rows of the per-tile statistics a ring launch writes (its pixel tiles), 0 when the ring kernel does not apply
extern "C" long long ffa_ring_stat_rows(int B, int H, int W, int co_rows) {
  const RingPlan p = ring_plan(B, H, W, co_rows);
  if (!p.cfg) return 0;
  return (long long)B * ffa_cdiv(W, p.tw) * ffa_cdiv(H, p.th);
}

// in: [B][H][W][Ci], out / residual: [B][H][W][Co]; w_ring: operand packed by the ring layout of
// ffa_pack_conv_weight for co_rows rows.  pro_scale / pro_shift (both or neither): the convolution reads
// relu(in * pro_scale[c] + pro_shift[c]) instead of in (training-mode BatchNorm + ReLU of the producing layer,
// evaluated while the halo is staged; zero padding applies to the normalised tensor).
extern "C" int ffa_ring_conv3x3(int dtype, const void* in, const void* w_ring, const float* bias, const void* residual, void* out,
                     float* stat_partials, const float* pro_scale, const float* pro_shift, int B, int H, int W, int Ci,
                     int Co, int co_rows, int relu, hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "ring conv: bad dtype %d", dtype);
  FFA_REQUIRE(in && w_ring && out, "ring conv: null pointer");
  FFA_REQUIRE((pro_scale == nullptr) == (pro_shift == nullptr), "ring conv: prologue needs scale and shift");
  const int eb = (dtype == FFA_BF16) ? 2 : 4;
  FFA_REQUIRE(B > 0 && H > 0 && W > 0 && (Ci * eb) % 64 == 0 && Co % 8 == 0 && co_rows % 64 == 0,
              "ring conv: bad dims (Ci %d, Co %d, rows %d)", Ci, Co, co_rows);
  FFA_REQUIRE((long long)B * H * W * Ci * eb < (1LL << 31), "ring conv: input tensor must be smaller than 2 GiB");
  const RingPlan p = ring_plan(B, H, W, co_rows);
  if (!p.cfg) {
    ffa_set_error("ring conv: no configuration for %dx%d, %d rows", H, W, co_rows);
    return FFA_ERR_UNSUPPORTED;
  }
  Ring3Args a;
  a.in = in;
  a.w = w_ring;
  a.out = out;
  a.bias = bias;
  a.stats = stat_partials;
  a.res = residual;
  a.pro_sc = pro_scale;
  a.pro_sh = pro_shift;
  a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.relu = relu;
  a.nchunks = Ci * eb / 64;
  a.tiles_x = ffa_cdiv(W, p.tw);
  a.tiles_y = ffa_cdiv(H, p.th);
  a.npt = B * a.tiles_x * a.tiles_y;
  a.ncb = co_rows / p.bco;
  a.cb64_stride = (long long)a.nchunks * 3 * (3 * 2 * 64 * 32);
  const int total = ffa_cdiv(a.npt, 8) * 8 * a.ncb;
  const char* pg = getenv("FFA_RING_GRID");
  // two 4-wave blocks per CU (cfg 1; cfg 2 on the bf16 kernel), one 8-wave block (cfg 4)
  int cap = pg ? atoi(pg) : ((p.cfg == 1 || (p.cfg == 2 && dtype == FFA_BF16)) ? 512 : 256);
  cap = cap < 8 ? 8 : (cap / 8) * 8;
  const int grid = total < cap ? total : cap;
  if (dtype == FFA_BF16) {  // v_mfma_f32_16x16x32_bf16 kernel (NT counts 16-pixel fragments there)
    switch (p.cfg) {
      case 1: return ring16_launch<1, 4, 4, 8, 32, 2>(a, grid, stream);
      case 2: return ring16_launch<1, 4, 4, 16, 16, 2>(a, grid, stream);
      case 4: return ring16_launch<2, 4, 4, 8, 32, 2>(a, grid, stream);
    }
    return FFA_ERR_UNSUPPORTED;
  }
#define FFA_RING_DISPATCH(T_)                                                        \
  switch (p.cfg) {                                                                   \
    case 1: return ring_launch<T_, 1, 4, 2, 8, 32, 2>(a, grid, stream);              \
    case 2: return ring_launch<T_, 1, 4, 2, 16, 16, 1>(a, grid, stream);             \
    case 4: return ring_launch<T_, 2, 4, 2, 8, 32, 2>(a, grid, stream);              \
  }
  FFA_RING_DISPATCH(float)
#undef FFA_RING_DISPATCH
  return FFA_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// weight packing for the ring layout:
//   dst[cb64][chunk][kernel row r][tap s][k-step][row < 64 (fragment order)][half'][elements of 16 B]
// half' = half ^ bit 3 of the row.  src element (row, ch, r, s) is read at src[row*s_row + ch*s_ch + r*3 + s];
// flip mirrors the taps (dgrad operand: rows = ci, channels = co).

struct RingPackArgs {
  const float* src;
  void* dst;
  const float* scale;
  long long s_row, s_ch;
  int rows, chs;  // valid rows / channels in src
  int nchunks, ncb64, flip;
};

template <typename T>
__device__ __forceinline__ void ring_pack_piece(const RingPackArgs& p, long long i16) {
  // one thread = one 16-byte piece
  constexpr int EPF = ElemTraits<T>::kPerFrag;
  long long t = i16;
  const int hp = (int)(t % 2); t /= 2;
  const int row_l = (int)(t % 64); t /= 64;
  const int ks = (int)(t % 2); t /= 2;
  const int s = (int)(t % 3); t /= 3;
  const int r = (int)(t % 3); t /= 3;
  const int cc = (int)(t % p.nchunks); t /= p.nchunks;
  const int cb = (int)t;
  const int mt = row_l >> 5, rho = row_l & 31;
  const int row = cb * 64 + 16 * (rho >> 3) + 8 * ((rho >> 2) & 1) + 4 * mt + (rho & 3);
  const int hsrc = hp ^ ((row_l >> 3) & 1);
  const int ch0 = cc * (4 * EPF) + ks * (2 * EPF) + hsrc * EPF;
  int rr = r, ss = s;
  if (p.flip) {
    rr = 2 - r;
    ss = 2 - s;
  }
  float v[8];
  const bool row_ok = row < p.rows;
  const float sc = (row_ok && p.scale) ? p.scale[row] : 1.f;
  const float* src = p.src + (long long)row * p.s_row + rr * 3 + ss;
#pragma unroll
  for (int j = 0; j < EPF; ++j) {
    const int ch = ch0 + j;
    v[j] = (row_ok && ch < p.chs) ? src[(long long)ch * p.s_ch] * sc : 0.f;
  }
  if constexpr (EPF == 8) {
    ffa_store8<ffa_bf16>(static_cast<ffa_bf16*>(p.dst) + i16 * 8, v);
  } else {
    *reinterpret_cast<float4*>(static_cast<float*>(p.dst) + i16 * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// bf16 operand of conv3x3_ring16_kernel: dst[cb64][chunk][kernel row r][tap s][row < 64][slot' < 4][8 elements],
// LDS row mt*16 + 4*kg + i holds output channel 16*kg + 4*mt + i of the 64-row group, slot' = slot ^ 2 when bit 2 of
// the row is set (slot = channels 8*slot .. 8*slot+7 of the 32-channel chunk).  Same bytes per (group, chunk) as the
// 32x32 layout, so sizes and strides are shared.
// One thread = one (row, 16-byte slot) of a chunk, all nine taps: it reads 8 channels x 9 contiguous taps of the OIHW
// master (36-byte runs; a thread per piece touched a 32-byte sector per 4 useful bytes: 287 us per step for the
// U-Net's operands) and writes its nine pieces, one into each tap plane.  Threads of a block run over the 64 rows first,
// so the transposed (dgrad) operand -- whose rows are the contiguous axis of the master -- reads whole runs too.
__device__ __forceinline__ void ring16_pack_rowslot(const RingPackArgs& p, long long idx) {
  // lanes follow the contiguous axis of the master: (slot, chunk) = consecutive 288-byte runs of one row for the
  // forward operand [row][ch][tap]; rows = consecutive 36-byte runs for the transposed one [ch][row][tap]
  long long t = idx;
  int row_l, sp, cc;
  if (!p.flip) {
    sp = (int)(t % 4); t /= 4;
    cc = (int)(t % p.nchunks); t /= p.nchunks;
    row_l = (int)(t % 64); t /= 64;
  } else {
    row_l = (int)(t % 64); t /= 64;
    sp = (int)(t % 4); t /= 4;
    cc = (int)(t % p.nchunks); t /= p.nchunks;
  }
  const int cb = (int)t;
  const int mt = row_l >> 4, kg = (row_l >> 2) & 3, i = row_l & 3;
  const int row = cb * 64 + 16 * kg + 4 * mt + i;
  const int slot = sp ^ (((row_l >> 2) & 1) << 1);
  const int ch0 = cc * 32 + slot * 8;
  const bool row_ok = row < p.rows;
  const float sc = (row_ok && p.scale) ? p.scale[row] : 1.f;
  float v[8][9];
  if (!p.flip && row_ok && ch0 + 8 <= p.chs && (p.s_row & 3) == 0 && ((size_t)p.src & 15) == 0) {
    // forward operand: the 8 channels x 9 taps of this thread are 72 CONTIGUOUS floats of the master -> 18 16-byte
    // loads (a 4-byte load per element made every wave instruction touch 64 cache lines: 171 us per step)
    const ffa_f32x4* src4 = reinterpret_cast<const ffa_f32x4*>(p.src + (long long)row * p.s_row + (long long)ch0 * 9);
    float f[72];
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const ffa_f32x4 t4 = src4[q];
      f[4 * q] = t4[0]; f[4 * q + 1] = t4[1]; f[4 * q + 2] = t4[2]; f[4 * q + 3] = t4[3];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) v[j][tp] = f[j * 9 + tp] * sc;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = ch0 + j;
      const bool ok = row_ok && ch < p.chs;
      const float* src = p.src + (long long)(ok ? row : 0) * p.s_row + (long long)(ok ? ch : 0) * p.s_ch;
      // destination tap tp reads source tap tp, or 8 - tp for the mirrored (dgrad) operand: the choice goes into the
      // ADDRESS -- indexing v[][] with a run-time value would send the array to scratch memory
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) v[j][tp] = ok ? src[p.flip ? 8 - tp : tp] * sc : 0.f;
    }
  }
  ffa_bf16* dst = static_cast<ffa_bf16*>(p.dst) + ((((long long)cb * p.nchunks + cc) * 9) * 256 + row_l * 4 + sp) * 8;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {  // destination tap (r, s) = tp
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[j][tp];
    ffa_store8<ffa_bf16>(dst + (long long)tp * 256 * 8, o);
  }
}

__global__ void ring_pack_kernel(RingPackArgs p, int dtype) {
  if (dtype == FFA_BF16) {
    const long long total = (long long)p.ncb64 * p.nchunks * 64 * 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
      ring16_pack_rowslot(p, i);
    return;
  }
  const long long total = (long long)p.ncb64 * p.nchunks * 3 * 3 * 2 * 64 * 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    ring_pack_piece<float>(p, i);
}

__global__ void ring_pack_batched_kernel(const RingPackArgs* __restrict__ descs, int dtype) {
  const RingPackArgs p = descs[blockIdx.y];
  if (dtype == FFA_BF16) {
    const long long total = (long long)p.ncb64 * p.nchunks * 64 * 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
      ring16_pack_rowslot(p, i);
    return;
  }
  const long long total = (long long)p.ncb64 * p.nchunks * 3 * 3 * 2 * 64 * 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    ring_pack_piece<float>(p, i);
}

static int ring_pack_fill(RingPackArgs& p, const float* w_oihw, const float* scale, void* dst, int O, int I,
                          int transpose, int co_rows, int ci_pitch, int dtype) {
  FFA_REQUIRE(w_oihw && dst, "ring pack: null pointer");
  const int eb = (dtype == FFA_BF16) ? 2 : 4;
  FFA_REQUIRE(co_rows % 64 == 0 && (ci_pitch * eb) % 64 == 0, "ring pack: rows %d / pitch %d not whole groups", co_rows,
              ci_pitch);
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
  FFA_REQUIRE(p.rows <= co_rows && p.chs <= ci_pitch, "ring pack: padded dims smaller than the tensor");
  p.nchunks = ci_pitch * eb / 64;
  p.ncb64 = co_rows / 64;
  return FFA_OK;
}

extern "C" int ffa_ring_pack(int dtype, const float* w_oihw, const float* scale, void* dst, int O, int I, int transpose,
                  int co_rows, int ci_pitch, hipStream_t stream) {
  RingPackArgs p;
  const int rc = ring_pack_fill(p, w_oihw, scale, dst, O, I, transpose, co_rows, ci_pitch, dtype);
  if (rc != FFA_OK) return rc;
  const long long total = (long long)p.ncb64 * p.nchunks * 3 * 3 * 2 * 64 * 2;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(ring_pack_kernel, dim3(grid), dim3(256), 0, stream, p, dtype);
  return ffa_check_launch("ring_pack");
}

extern "C" int ffa_ring_pack_desc_bytes(void) { return (int)sizeof(RingPackArgs); }

extern "C" int ffa_ring_pack_desc_fill(void* host_desc, const float* w_oihw, const float* scale, void* dst, int O, int I,
                            int transpose, int co_rows, int ci_pitch, int dtype) {
  FFA_REQUIRE(host_desc, "ring pack: null descriptor");
  RingPackArgs p;
  const int rc = ring_pack_fill(p, w_oihw, scale, dst, O, I, transpose, co_rows, ci_pitch, dtype);
  if (rc != FFA_OK) return rc;
  memcpy(host_desc, &p, sizeof(p));
  return FFA_OK;
}

extern "C" int ffa_ring_pack_batched(int dtype, const void* descs_device, int n, hipStream_t stream) {
  FFA_REQUIRE(dtype == FFA_BF16 || dtype == FFA_F32, "ring pack: bad dtype");
  FFA_REQUIRE(descs_device && n > 0 && n <= 65535, "ring pack: bad descriptor table");
  hipLaunchKernelGGL(ring_pack_batched_kernel, dim3(256, n), dim3(256), 0, stream,
                     static_cast<const RingPackArgs*>(descs_device), dtype);
  return ffa_check_launch("ring_pack_batched");
}

#if FFA_RING_TRACE
extern "C" int ffa_ring_trace_read(long long* host_dst, int n) {
  if (n > 1024 * 8) n = 1024 * 8;
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ffa_ring_trace_buf), (size_t)n * sizeof(long long));
}
#endif
