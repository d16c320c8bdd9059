// AdamW / Adam over every parameter tensor of the model in ONE launch (round 3).
//
// Replaces torch.optim.AdamW(fused=True) -- the optimizer the reference builds at flair_hub/tasks/tasks_module.py:385-389
// (Adam / AdamW, betas, weight decay) -- whose multi-tensor kernel takes four launches of ~64 us for the U-Net's 24.4 M f32
// parameters (2.7 TB/s of the 686 MB it must move: p, g, m, v in, p, m, v out).  Here: up to 72 tensor descriptors
// (p, g, m, v, step, n) ride in the kernel argument together with the first block of every tensor (binary search per
// block) -- no device tables, nothing to copy, trivially capturable --, three launches for the U-Net's 186 tensors; every
// thread keeps four 16-byte vectors per operand in flight (the streaming recipe of the BatchNorm kernels).  Same arithmetic and operation order as
// torch's fused kernel (ATen/native/cuda/fused_adam_utils.cuh: decoupled decay on the parameter, lerp for the first moment,
// bias corrections formed in double, sqrt(v) / sqrt(bc2) + eps), so optimizer state is interchangeable with torch's.
#include "ffa_common.h"
#include "ffa_common_host.h"

struct AdamTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  const float* step;  // device scalar: the step count AFTER this update's increment (torch's capturable convention)
  long long n;
};
constexpr int kAdamMaxTensors = 72;  // 72 x 48 B + 73 x 4 B < the 4 KB kernel-argument limit
constexpr int kAdamChunk = 4096;     // elements per block: 256 threads x 4 vectors of 4 floats
struct AdamArgs {
  AdamTensor t[kAdamMaxTensors];
  int first_block[kAdamMaxTensors + 1];  // first block of tensor i; [count] = grid size
  int count;
};

__global__ void __launch_bounds__(256) adamw_multi_kernel(AdamArgs a, const float* __restrict__ lr_ptr, double beta1,
                                                          double beta2, float eps, float weight_decay, int decoupled,
                                                          int maximize) {
  int lo = 0, hi = a.count;  // the tensor whose block range holds blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= a.first_block[mid]) lo = mid;
    else hi = mid;
  }
  const AdamTensor& t = a.t[lo];
  const long long base = (long long)((int)blockIdx.x - a.first_block[lo]) * kAdamChunk;
  __shared__ float coef[2];
  if (threadIdx.x == 0) {
    const double s = (double)*t.step;
    const double bc1 = 1.0 - pow(beta1, s);
    const double bc2 = 1.0 - pow(beta2, s);
    coef[0] = (float)((double)*lr_ptr / bc1);  // step_size
    coef[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float lr = *lr_ptr;
  const float step_size = coef[0], bc2_sqrt = coef[1];
  const float b1w = (float)(1.0 - beta1), b2 = (float)beta2, b2w = (float)(1.0 - beta2);
  const long long left = t.n - base;
  float* __restrict__ tp = t.p;
  const float* __restrict__ tg = t.g;
  float* __restrict__ tm = t.m;
  float* __restrict__ tv = t.v;
  // no FMA contraction: the vector path and the scalar path (ragged tails, gradients living at an unaligned offset of a
  // data-parallel bucket) must round alike, or a replica's update depends on where its gradient happens to sit
  auto update = [&](float& p, float g, float& m, float& v) {
#pragma clang fp contract(off)
    if (maximize) g = -g;
    if (decoupled) p -= lr * weight_decay * p;            // AdamW
    else if (weight_decay != 0.f) g += weight_decay * p;  // Adam's L2 term
    m = m + b1w * (g - m);                                 // std::lerp(m, g, 1 - beta1) for a weight below 0.5
    v = b2 * v + b2w * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p -= step_size * m / denom;
  };
  if (left >= kAdamChunk && (((size_t)(tp + base) | (size_t)(tg + base) | (size_t)(tm + base) | (size_t)(tv + base)) & 15) == 0) {
    float4 P[4], G[4], M[4], V[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long i = base + (k * 256 + threadIdx.x) * 4;
      P[k] = *reinterpret_cast<const float4*>(tp + i);
      G[k] = *reinterpret_cast<const float4*>(tg + i);
      M[k] = *reinterpret_cast<const float4*>(tm + i);
      V[k] = *reinterpret_cast<const float4*>(tv + i);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      update(P[k].x, G[k].x, M[k].x, V[k].x);
      update(P[k].y, G[k].y, M[k].y, V[k].y);
      update(P[k].z, G[k].z, M[k].z, V[k].z);
      update(P[k].w, G[k].w, M[k].w, V[k].w);
      const long long i = base + (k * 256 + threadIdx.x) * 4;
      *reinterpret_cast<float4*>(tp + i) = P[k];
      *reinterpret_cast<float4*>(tm + i) = M[k];
      *reinterpret_cast<float4*>(tv + i) = V[k];
    }
  } else {  // the ragged last chunk of a tensor, or a tensor that is not 16-byte aligned
    const long long end = left < kAdamChunk ? left : kAdamChunk;
    for (long long j = threadIdx.x; j < end; j += 256) {
      float p = tp[base + j], m = tm[base + j], v = tv[base + j];
      update(p, tg[base + j], m, v);
      tp[base + j] = p;
      tm[base + j] = m;
      tv[base + j] = v;
    }
  }
}

// host arrays of device pointers, one entry per parameter tensor (f32, contiguous); lr a device scalar; step[i] a device
// f32 scalar that already counts this update.  Launches ceil(n / 72) kernels on `stream`; nothing else.
extern "C" int ffa_adamw_multi(int n_tensors, void* const* p, const void* const* g, void* const* m, void* const* v,
                               const void* const* step, const long long* numel, const float* lr, double beta1,
                               double beta2, float eps, float weight_decay, int decoupled, int maximize,
                               hipStream_t stream) {
  FFA_REQUIRE(n_tensors > 0 && p && g && m && v && step && numel && lr, "adamw_multi: bad arguments");
  FFA_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1, "adamw_multi: betas must lie in [0, 1)");
  for (int i0 = 0; i0 < n_tensors; i0 += kAdamMaxTensors) {
    AdamArgs a;
    const int cnt = n_tensors - i0 < kAdamMaxTensors ? n_tensors - i0 : kAdamMaxTensors;
    long long nb = 0;
    for (int j = 0; j < cnt; ++j) {
      const int i = i0 + j;
      FFA_REQUIRE(p[i] && g[i] && m[i] && v[i] && step[i] && numel[i] > 0, "adamw_multi: null or empty tensor %d", i);
      a.t[j].p = (float*)p[i]; a.t[j].g = (const float*)g[i]; a.t[j].m = (float*)m[i]; a.t[j].v = (float*)v[i];
      a.t[j].step = (const float*)step[i]; a.t[j].n = numel[i];
      a.first_block[j] = (int)nb;
      nb += (numel[i] + kAdamChunk - 1) / kAdamChunk;
      FFA_REQUIRE(nb < (1LL << 31), "adamw_multi: too many blocks");
    }
    a.first_block[cnt] = (int)nb;
    a.count = cnt;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a, lr, beta1, beta2, eps, weight_decay,
                       decoupled, maximize);
    const int rc = ffa_check_launch("adamw_multi");
    if (rc) return rc;
  }
  return 0;
}
