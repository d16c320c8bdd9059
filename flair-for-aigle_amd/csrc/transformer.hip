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
                                                         const float* __restrict__ beta, long long rows, int C, int Ho,
                                                         int Wo, float eps) {
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
static int layer_norm_launch(const void* x, void* y, const float* gamma, const float* beta, long long rows, int C, int Ho,
                             int Wo, float eps, hipStream_t stream) {
  const int CG = C / 8;
#define FFA_LN_CASE(LPR, GPL)                                                                                        \
  hipLaunchKernelGGL((layer_norm_kernel<T, MERGE, LPR, GPL>), dim3((unsigned)((rows + 256 / LPR - 1) / (256 / LPR))), \
                     dim3(256), 0, stream, (const T*)x, (T*)y, gamma, beta, rows, C, Ho, Wo, eps)
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

extern "C" int ffa_layer_norm(int dtype, const void* x, void* y, const float* gamma, const float* beta, long long rows,
                              int C, float eps, hipStream_t stream) {
  FFA_REQUIRE(x && y && gamma && beta && rows > 0 && C > 0 && C % 8 == 0, "layer_norm: bad arguments");
  if (dtype == FFA_BF16) return layer_norm_launch<ffa_bf16, false>(x, y, gamma, beta, rows, C, 0, 0, eps, stream);
  return layer_norm_launch<float, false>(x, y, gamma, beta, rows, C, 0, 0, eps, stream);
}

extern "C" int ffa_patch_merge_norm(int dtype, const void* x, void* y, const float* gamma, const float* beta, int B,
                                    int H, int W, int C, float eps, hipStream_t stream) {
  FFA_REQUIRE(x && y && gamma && beta && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "patch_merge_norm: bad arguments");
  FFA_REQUIRE(H % 2 == 0 && W % 2 == 0, "patch_merge_norm: odd map %d x %d (the padded variant is not implemented)", H, W);
  const long long rows = (long long)B * (H / 2) * (W / 2);
  if (dtype == FFA_BF16)
    return layer_norm_launch<ffa_bf16, true>(x, y, gamma, beta, rows, 4 * C, H / 2, W / 2, eps, stream);
  return layer_norm_launch<float, true>(x, y, gamma, beta, rows, 4 * C, H / 2, W / 2, eps, stream);
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

template <typename T>
__global__ void adaptive_avg_pool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int S) {
  const int CG = C / 8;
  const long long total = (long long)B * S * S * CG;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % CG);
    long long p = i / CG;
    const int ox = (int)(p % S);
    p /= S;
    const int oy = (int)(p % S);
    const long long b = p / S;
    const int y0 = (oy * H) / S, y1 = ((oy + 1) * H + S - 1) / S;
    const int x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        float v[8];
        ffa_load8<T>(x + ((b * H + yy) * W + xx) * C + g * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    const float inv = 1.0f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    ffa_store8<T>(y + i * 8, acc);
  }
}

extern "C" int ffa_adaptive_avg_pool(int dtype, const void* x, void* y, int B, int H, int W, int C, int S,
                                     hipStream_t stream) {
  FFA_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && S > 0, "adaptive_avg_pool: bad arguments");
  const long long items = (long long)B * S * S * (C / 8);
  if (dtype == FFA_BF16)
    hipLaunchKernelGGL(adaptive_avg_pool_kernel<ffa_bf16>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const ffa_bf16*)x, (ffa_bf16*)y, B, H, W, C, S);
  else
    hipLaunchKernelGGL(adaptive_avg_pool_kernel<float>, dim3(tf_grid(items)), dim3(FFA_TF_THREADS), 0, stream,
                       (const float*)x, (float*)y, B, H, W, C, S);
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
