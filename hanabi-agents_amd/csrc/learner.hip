// learner.hip — the non-GEMM parts of one learner update, fused for gfx950.
//
// DQNLearning.update_q of the reference (hanabi_agents/rlax_dqn/rlax_rainbow.py:152-217) is, apart from the
// dense layers, a long chain of tiny elementwise / reduce ops on [B, A, K] = [256, 20, 51] tensors; as
// separate launches they cost ~0.5 ms per update on the MI355X (launch-bound). Here they are three kernels:
//
//   replay_gather    indices -> X [2B, L] in the GEMM dtype (rows 0..B-1 = obs_tm1, B..2B-1 = obs_t) plus the
//                    compact action / reward / terminal columns (experience_buffer.py:83-87 gather)
//   c51_loss_grad    IS weights (rlax_rainbow.py:188-189), double-Q action selection on the online logits of
//                    obs_t (:175-176), target distribution from the target net, Cramer/L2 projection onto the
//                    support (rlax.categorical_l2_project), cross-entropy against the online
//                    logits of obs_tm1 (:182-184), and d(mean(td*w))/d(logits_tm1) — one wavefront per sample
//   noisy_adam       gradient of the merged weight W = w + w_mu + w_sigma*eps routed to the three parameters
//                    (g, g, g*eps), optix/torch Adam (eps 3.125e-5 outside the sqrt, rlax_rainbow.py:257) on all
//                    three, and the NEXT effective weight written in the GEMM dtype — so the forward never
//                    re-forms effective weights and the actor re-uses them
// The GEMMs between them stay in hipBLASLt (torch.addmm / torch.mm on MFMA).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "../../include/hanabi_hip.h"
#include "common.hpp"

using hb::fail;

namespace {

template <typename T> __device__ __forceinline__ float ld(const T* p, long long i);
template <> __device__ __forceinline__ float ld<float>(const float* p, long long i) { return p[i]; }
template <> __device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16* p, long long i) { return __bfloat162float(p[i]); }
template <> __device__ __forceinline__ float ld<__half>(const __half* p, long long i) { return __half2float(p[i]); }
template <typename T> __device__ __forceinline__ void st(T* p, long long i, float v);
template <> __device__ __forceinline__ void st<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<__hip_bfloat16>(__hip_bfloat16* p, long long i, float v) { p[i] = __float2bfloat16(v); }
template <> __device__ __forceinline__ void st<__half>(__half* p, long long i, float v) { p[i] = __float2half(v); }

__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------
// n-step assembly at SAMPLE time (spec: hanabi_agents/rainbow/replay_memory.py:316-345). The ring
// holds 1-step transitions of one seat; when every insert appends the same number of rows (n_ins = games, the
// lock-step driver), the seat's next transition of the same game sits n_ins slots further on. Starting at the
// sampled slot the chain follows up to n_step-1 successors, stops at an episode end or at the write pointer,
// and returns R = sum gamma^k r_k, the final obs_t / terminal flag and disc = gamma^m (m = steps taken).
struct NStep {
  const long long* size_wp;  // device: {entries in the ring, next slot to be written}
  long long cap, n_ins;
  int n_step;
  float gamma;
};

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const int8_t* __restrict__ ring_tm1, const int8_t* __restrict__ ring_t,
                                                     const int8_t* __restrict__ ring_act, const float* __restrict__ ring_rew,
                                                     const uint8_t* __restrict__ ring_term, const int64_t* __restrict__ idx,
                                                     int B, int L, T* __restrict__ x, int x_ld, int32_t* __restrict__ act,
                                                     float* __restrict__ rew, float* __restrict__ term,
                                                     float* __restrict__ disc, const NStep ns, int packed_words) {
  __shared__ long long s_slot;
  const int r = blockIdx.x;  // output row 0..2B-1
  const int b = r < B ? r : r - B;
  long long slot = idx[b];
  if (r >= B) {  // obs_t row: walk the n-step chain (one thread, a handful of dependent 1-byte / 4-byte reads)
    if (threadIdx.x == 0) {
      long long j = slot;
      float R = ring_rew[j], g = ns.gamma;
      int m = 1;
      if (ns.n_step > 1) {
        const long long size = ns.size_wp[0], wp = ns.size_wp[1];
        const long long ahead = size >= ns.cap ? ((wp - 1 - slot) % ns.cap + ns.cap) % ns.cap : size - 1 - slot;
        while (m < ns.n_step && !ring_term[j] && static_cast<long long>(m) * ns.n_ins <= ahead) {
          j = (j + ns.n_ins) % ns.cap;
          R += g * ring_rew[j];
          g *= ns.gamma;
          ++m;
        }
      }
      rew[b] = R;
      term[b] = ring_term[j] ? 1.f : 0.f;
      disc[b] = g;
      s_slot = j;
    }
    __syncthreads();
    slot = s_slot;
  } else if (threadIdx.x == 0) {
    act[b] = ring_act[slot];
  }
  T* dst = x + static_cast<long long>(r) * x_ld;  // row stride x_ld >= L: padding columns are left untouched (zero)
  if (packed_words > 0) {  // bit-packed rings: rows of packed_words u32, observation bit j = word j >> 5, bit j & 31
    const uint32_t* src = reinterpret_cast<const uint32_t*>(r < B ? ring_tm1 : ring_t) + slot * packed_words;
    for (int j = threadIdx.x; j < L; j += 256) st<T>(dst, j, static_cast<float>((src[j >> 5] >> (j & 31)) & 1u));
    return;
  }
  const int8_t* src = (r < B ? ring_tm1 : ring_t) + slot * L;
  for (int j = threadIdx.x; j < L; j += 256) st<T>(dst, j, static_cast<float>(src[j]));
}

// ---------------------------------------------------------------------------------------------
// one wavefront per sample; requires A <= 64 and K <= 64
template <typename T>
__global__ __launch_bounds__(256) void c51_kernel(const T* __restrict__ logits_on, const T* __restrict__ logits_t,
                                                  const int32_t* __restrict__ act, const float* __restrict__ rew,
                                                  const float* __restrict__ term, const double* __restrict__ prios,
                                                  const float* __restrict__ beta_dev, const float* __restrict__ disc, int mask_terminal,
                                                  const float* __restrict__ support, int B, int A, int K, int rs,
                                                  float* __restrict__ td_out, float* __restrict__ w_out,
                                                  T* __restrict__ dlogits, float* __restrict__ counter,
                                                  const T* __restrict__ bias_on, const T* __restrict__ bias_t) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (counter && blockIdx.x == 0 && threadIdx.x == 0) *counter += 1.f;  // nothing else touches it during this kernel
  if (b >= B) return;
  const int AK = A * K;
  float* sel = lds + wave * (AK + 64);
  // ---- importance weight: ((1/P).astype(f32)) ** beta, normalised by its batch maximum (rlax_rainbow.py:188-189)
  const float beta = *beta_dev;
  float wmax = 0.f;
  for (int j = lane; j < B; j += 64) wmax = fmaxf(wmax, powf(static_cast<float>(1.0 / prios[j]), beta));
  wmax = wave_max(wmax);
  const float w_b = powf(static_cast<float>(1.0 / prios[b]), beta) / wmax;
  // ---- double-Q selector: q_sel[a] = mean(softmax(online(obs_t))[a] * z) (no legal mask, as the reference)
  const T* row_sel = logits_on + static_cast<long long>(B + b) * rs;  // rows are rs >= A*K elements apart
  // (bias_on / bias_t: output-layer biases not yet added by the caller's GEMM, e.g. a batched GEMM without epilogue)
  for (int e = lane; e < AK; e += 64) sel[e] = ld<T>(row_sel, e) + (bias_on ? ld<T>(bias_on, e) : 0.f);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float q = -INFINITY;
  if (lane < A) {
    const float* r = sel + lane * K;
    float m = r[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, r[k]);
    float s = 0.f, t = 0.f;
    for (int k = 0; k < K; ++k) {
      const float e = __expf(r[k] - m);
      s += e;
      t += e * support[k];
    }
    q = t / s / static_cast<float>(K);
  }
  const float qbest = wave_max(q);
  const unsigned long long hit = __ballot(lane < A && q == qbest);
  const int a_star = __ffsll(static_cast<long long>(hit)) - 1;  // lowest index among ties, like argmax
  // ---- target distribution p = softmax(target(obs_t)[a*]) and its projection onto the support
  const bool atom = lane < K;
  const float lt = atom ? ld<T>(logits_t, static_cast<long long>(b) * rs + a_star * K + lane) +
                              (bias_t ? ld<T>(bias_t, a_star * K + lane) : 0.f)
                        : -INFINITY;
  const float mt = wave_max(lt);
  const float et = atom ? __expf(lt - mt) : 0.f;
  const float p = et / wave_sum(et);
  const float vmin = support[0], vmax = support[K - 1];
  const float delta = (vmax - vmin) / static_cast<float>(K - 1);
  float gamma = disc[b];  // gamma^m of this sample's (n-step) transition
  if (mask_terminal) gamma *= 1.f - term[b];
  const float zi = atom ? support[lane] : 0.f;
  const float tz = fminf(fmaxf(rew[b] + gamma * zi, vmin), vmax);  // clipped r + gamma * z_j held by lane j
  float target = 0.f;
  const float inv_delta = 1.f / delta;
  for (int j = 0; j < K; ++j) {  // j is wave-uniform: the broadcasts are v_readlane, not LDS permutes
    const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), j));
    const float tzj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tz), j));
    target += pj * fminf(fmaxf(1.f - fabsf(tzj - zi) * inv_delta, 0.f), 1.f);
  }
  if (!atom) target = 0.f;
  // ---- cross-entropy against log_softmax(online(obs_tm1)[a_tm1]) and its gradient
  const int a_tm1 = act[b];
  const float l1 = atom ? ld<T>(logits_on, static_cast<long long>(b) * rs + a_tm1 * K + lane) +
                              (bias_on ? ld<T>(bias_on, a_tm1 * K + lane) : 0.f)
                        : -INFINITY;
  const float m1 = wave_max(l1);
  const float e1 = atom ? __expf(l1 - m1) : 0.f;
  const float s1 = wave_sum(e1);
  const float logp = atom ? (l1 - m1 - __logf(s1)) : 0.f;
  const float td = -wave_sum(target * logp);
  const float tsum = wave_sum(target);
  if (lane == 0) {
    td_out[b] = td;
    w_out[b] = w_b;
  }
  T* drow = dlogits + static_cast<long long>(b) * rs;
  for (int e = lane; e < rs; e += 64) st<T>(drow, e, 0.f);  // padding columns included
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (atom) st<T>(drow, a_tm1 * K + lane, (w_b / static_cast<float>(B)) * (e1 / s1 * tsum - target));
}

// ---------------------------------------------------------------------------------------------
// column sums out[j] = sum_i x[i, j] (bias gradients): 64 columns x 16 row groups per workgroup, so a 256-row
// operand is 16 independent loads per thread (one memory round trip); fixed summation order, fp32 accumulation
template <typename T>
__global__ __launch_bounds__(1024) void colsum_kernel(const T* __restrict__ x, int rows, int cols, float* __restrict__ out) {
  __shared__ float part[16][64];
  const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, rg = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols) {
#pragma unroll 16
    for (int i = rg; i < rows; i += 16) s += ld<T>(x, static_cast<long long>(i) * cols + c);
  }
  part[rg][l] = s;
  __syncthreads();
  if (rg == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][l];
    out[c] = t;
  }
}

// ReLU backward fused with the bias gradient: dy[i, j] <- act[i, j] > 0 ? dy[i, j] : 0 (aten::threshold_backward
// with threshold 0 on the post-activation), out[j] = sum_i of the masked values. Same tiling and summation order as
// colsum_kernel.
template <typename T>
__global__ __launch_bounds__(1024) void relu_bwd_colsum_kernel(T* __restrict__ dy, const T* __restrict__ act, int rows, int cols,
                                                               int act_ld, float* __restrict__ out) {
  __shared__ float part[16][64];
  const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, rg = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols) {
#pragma unroll 16
    for (int i = rg; i < rows; i += 16) {
      const long long k = static_cast<long long>(i) * cols + c;
      const float g = ld<T>(act, static_cast<long long>(i) * act_ld + c) > 0.f ? ld<T>(dy, k) : 0.f;
      st<T>(dy, k, g);
      s += g;
    }
  }
  part[rg][l] = s;
  __syncthreads();
  if (rg == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][l];
    out[c] = t;
  }
}

// ---------------------------------------------------------------------------------------------
struct AdamArgs {
  float *w, *w_mu, *w_sigma;
  const float* noise;  // eps of the layer (same shape); bias noise for bias tensors
  const void* grad;    // d loss / d (merged tensor): fp32 / bf16 / f16 (grad_dtype), rows grad_ld elements apart
  float *m_w, *v_w, *m_mu, *v_mu, *m_sg, *v_sg;
  const float* step;   // number of completed Adam steps (device scalar)
  void* eff;           // merged tensor in the GEMM dtype for the next forward
  long long n;
  int cols, eff_ld;    // parameter tensors are [n/cols, cols] row-major; eff rows are eff_ld >= cols elements apart
  int grad_dtype, grad_ld;
  float lr, b1, b2, eps;
  float step_offset;   // this step's number t = *step + step_offset
};

__device__ __forceinline__ float load_grad(const void* g, int dtype, long long i) {
  if (dtype == 1) return __bfloat162float(static_cast<const __hip_bfloat16*>(g)[i]);
  if (dtype == 2) return __half2float(static_cast<const __half*>(g)[i]);
  return static_cast<const float*>(g)[i];
}

__device__ __forceinline__ float adam1(float p, float g, float& m, float& v, float b1, float b2, float bc1, float bc2s,
                                       float lr, float eps) {
  // explicit roundings (no compiler-chosen FMA contraction): the scalar and the 4-wide kernels must agree bit for bit
  m = __fmaf_rn(b1, m, __fmul_rn(1.f - b1, g));
  v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.f - b2, g), g));
  // torch.optim.Adam / optix.adam: eps outside the sqrt
  return __fsub_rn(p, __fdiv_rn(__fmul_rn(lr / bc1, m), __fadd_rn(__fdiv_rn(sqrtf(v), bc2s), eps)));
}
// The step Adam takes for moments (m, v): p_new = p - adam_step(...). w and w_mu of a NoisyLinear always see the same
// gradient, so their moments are equal for ever; a caller may pass the SAME moment arrays for both (m_mu == m_w,
// v_mu == v_w) and the kernels then apply w's step to w_mu without touching the moments a second time.
__device__ __forceinline__ float adam_step(float m, float v, float bc1, float bc2s, float lr, float eps) {
  return __fdiv_rn(__fmul_rn(lr / bc1, m), __fadd_rn(__fdiv_rn(sqrtf(v), bc2s), eps));
}
__device__ __forceinline__ float merged(float w, float mu, float sg, float nz) {  // W = w + w_mu + w_sigma * noise
  return __fadd_rn(__fadd_rn(w, mu), __fmul_rn(sg, nz));
}

template <typename T>
__global__ __launch_bounds__(256) void noisy_adam_kernel(const AdamArgs a) {
  const float t = *a.step + a.step_offset;
  const float bc1 = 1.f - powf(a.b1, t), bc2s = sqrtf(1.f - powf(a.b2, t));
  T* eff = static_cast<T*>(a.eff);
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long row = i / a.cols, col = i - row * a.cols;
    const float g = load_grad(a.grad, a.grad_dtype, row * a.grad_ld + col), nz = a.noise[i];
    float m, v;
    m = a.m_w[i]; v = a.v_w[i];
    const float w = adam1(a.w[i], g, m, v, a.b1, a.b2, bc1, bc2s, a.lr, a.eps);
    a.m_w[i] = m; a.v_w[i] = v; a.w[i] = w;
    float mu;
    if (a.m_mu == a.m_w) {  // shared moments: (m, v) are w's, already advanced
      mu = __fsub_rn(a.w_mu[i], adam_step(m, v, bc1, bc2s, a.lr, a.eps));
    } else {
      m = a.m_mu[i]; v = a.v_mu[i];
      mu = adam1(a.w_mu[i], g, m, v, a.b1, a.b2, bc1, bc2s, a.lr, a.eps);
      a.m_mu[i] = m; a.v_mu[i] = v;
    }
    a.w_mu[i] = mu;
    m = a.m_sg[i]; v = a.v_sg[i];
    const float sg = adam1(a.w_sigma[i], __fmul_rn(g, nz), m, v, a.b1, a.b2, bc1, bc2s, a.lr, a.eps);
    a.m_sg[i] = m; a.v_sg[i] = v; a.w_sigma[i] = sg;
    st<T>(eff, row * a.eff_ld + col, merged(w, mu, sg, nz));
  }
}

// all merged tensors of the network in ONE launch: workgroup b belongs to tensor t where first[t] <= b < first[t+1]
struct AdamMulti {
  AdamArgs t[8];
  int first[9];
  int count;
};

// W consecutive elements per thread (one 16- or 8-byte access per array); every tensor has cols % 4 == 0, n < 2^31 and
// 16-byte aligned rows (checked on the host), so a group never straddles a row
// W consecutive elements per thread. W = 4 (16-byte accesses, 72 registers) is the form that runs; W = 2 (8-byte accesses,
// 44 registers) fits into the 48 registers per SIMD lane the one-kernel actor (csrc/actor_fused.hip) leaves free and so runs
// BESIDE the policy kernel instead of waiting for it — measured slower in the loop (hb_noisy_adam_multi below; DESIGN
// section 5c "Co-residency"), kept for measurements. Same arithmetic per element in every form (adam1's explicit roundings).
template <int W> struct VecW;
template <> struct VecW<4> { typedef float4 type; };
template <> struct VecW<2> { typedef float2 type; };
// (element offsets are 32-bit unsigned — the host checks n < 2^30 — so every access is "scalar base + 32-bit lane offset": one
//  address register per thread instead of a 64-bit pair per array)
template <int W>
__device__ __forceinline__ void ldw(const float* p, uint32_t off, float out[W]) {
  const typename VecW<W>::type v = *reinterpret_cast<const typename VecW<W>::type*>(reinterpret_cast<const char*>(p) + static_cast<uint64_t>(off << 2));
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int k = 0; k < W; ++k) out[k] = f[k];
}
template <int W>
__device__ __forceinline__ void stw(float* p, uint32_t off, const float in[W]) {
  typename VecW<W>::type v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int k = 0; k < W; ++k) f[k] = in[k];
  *reinterpret_cast<typename VecW<W>::type*>(reinterpret_cast<char*>(p) + static_cast<uint64_t>(off << 2)) = v;
}
template <int W>
__device__ __forceinline__ void load_gradw(const void* g, int dtype, uint32_t i, float out[W]) {
  if (dtype == 0) {
    ldw<W>(static_cast<const float*>(g), i, out);
  } else if (dtype == 1) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(static_cast<const char*>(g) + static_cast<uint64_t>(i << 1));
    if constexpr (W == 4) {
      const uint2 v = *reinterpret_cast<const uint2*>(p);
      out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xFFFF0000u);
      out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xFFFF0000u);
    } else {
      const uint32_t v = *p;
      out[0] = __uint_as_float(v << 16); out[1] = __uint_as_float(v & 0xFFFF0000u);
    }
  } else {
    const __half* h = reinterpret_cast<const __half*>(static_cast<const char*>(g) + static_cast<uint64_t>(i << 1));
#pragma unroll
    for (int k = 0; k < W; ++k) out[k] = __half2float(h[k]);
  }
}

template <typename T, int W>
__global__ __launch_bounds__(256) void noisy_adam_multiw_kernel(const AdamMulti m) {
  constexpr int SH = W == 4 ? 2 : 1;
  int ti = 0;
  while (ti + 1 < m.count && static_cast<int>(blockIdx.x) >= m.first[ti + 1]) ++ti;
  const AdamArgs& a = m.t[ti];
  const float t = *a.step + a.step_offset;
  const float bc1 = 1.f - powf(a.b1, t), bc2s = sqrtf(1.f - powf(a.b2, t));
  T* eff = static_cast<T*>(a.eff);
  const int nw = static_cast<int>(a.n >> SH), cw = a.cols >> SH;
  // (no grid-stride loop: the host launches one thread per group, so no address becomes a loop-carried 64-bit pointer)
  const int q = (static_cast<int>(blockIdx.x) - m.first[ti]) * 256 + static_cast<int>(threadIdx.x);
  if (q < nw) {
    const int row = q / cw, col = (q - row * cw) << SH;
    const uint32_t i = static_cast<uint32_t>(q) << SH;
    float g[W], nz[W];
    load_gradw<W>(a.grad, a.grad_dtype, static_cast<uint32_t>(row) * static_cast<uint32_t>(a.grad_ld) + col, g);
    ldw<W>(a.noise, i, nz);
    float res[3][W], lm[W], lv[W];
    auto one = [&](float* p, float* mp, float* vp, int which) {
      float pa[W], ma[W], va[W];
      ldw<W>(p, i, pa); ldw<W>(mp, i, ma); ldw<W>(vp, i, va);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        pa[k] = adam1(pa[k], which == 2 ? __fmul_rn(g[k], nz[k]) : g[k], ma[k], va[k], a.b1, a.b2, bc1, bc2s, a.lr, a.eps);
        res[which][k] = pa[k];
        lm[k] = ma[k];
        lv[k] = va[k];
      }
      stw<W>(p, i, pa); stw<W>(mp, i, ma); stw<W>(vp, i, va);
    };
    one(a.w, a.m_w, a.v_w, 0);

    if (a.m_mu == a.m_w) {  // shared moments: apply w's step to w_mu
      float pv[W];
      ldw<W>(a.w_mu, i, pv);
#pragma unroll
      for (int k = 0; k < W; ++k) res[1][k] = pv[k] = __fsub_rn(pv[k], adam_step(lm[k], lv[k], bc1, bc2s, a.lr, a.eps));
      stw<W>(a.w_mu, i, pv);
    } else {
      one(a.w_mu, a.m_mu, a.v_mu, 1);
    }

    one(a.w_sigma, a.m_sg, a.v_sg, 2);
    const uint32_t e = static_cast<uint32_t>(row) * static_cast<uint32_t>(a.eff_ld) + col;
#pragma unroll
    for (int k = 0; k < W; ++k) st<T>(eff, e + k, merged(res[0][k], res[1][k], res[2][k], nz[k]));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void noisy_adam_multi_kernel(const AdamMulti m) {
  int ti = 0;
  while (ti + 1 < m.count && static_cast<int>(blockIdx.x) >= m.first[ti + 1]) ++ti;
  const AdamArgs& a = m.t[ti];
  const int nb = m.first[ti + 1] - m.first[ti];
  const float t = *a.step + a.step_offset;
  const float bc1 = 1.f - powf(a.b1, t), bc2s = sqrtf(1.f - powf(a.b2, t));
  T* eff = static_cast<T*>(a.eff);
  for (long long i = static_cast<long long>(blockIdx.x - m.first[ti]) * 256 + threadIdx.x; i < a.n;
       i += static_cast<long long>(nb) * 256) {
    const long long row = i / a.cols, col = i - row * a.cols;
    const float g = load_grad(a.grad, a.grad_dtype, row * a.grad_ld + col), nz = a.noise[i];
    float mm, vv;
    mm = a.m_w[i]; vv = a.v_w[i];
    const float w = adam1(a.w[i], g, mm, vv, a.b1, a.b2, bc1, bc2s, a.lr, a.eps);
    a.m_w[i] = mm; a.v_w[i] = vv; a.w[i] = w;
    float mu;
    if (a.m_mu == a.m_w) {
      mu = __fsub_rn(a.w_mu[i], adam_step(mm, vv, bc1, bc2s, a.lr, a.eps));
    } else {
      mm = a.m_mu[i]; vv = a.v_mu[i];
      mu = adam1(a.w_mu[i], g, mm, vv, a.b1, a.b2, bc1, bc2s, a.lr, a.eps);
      a.m_mu[i] = mm; a.v_mu[i] = vv;
    }
    a.w_mu[i] = mu;
    mm = a.m_sg[i]; vv = a.v_sg[i];
    const float sg = adam1(a.w_sigma[i], __fmul_rn(g, nz), mm, vv, a.b1, a.b2, bc1, bc2s, a.lr, a.eps);
    a.m_sg[i] = mm; a.v_sg[i] = vv; a.w_sigma[i] = sg;
    st<T>(eff, row * a.eff_ld + col, merged(w, mu, sg, nz));
  }
}

// ---- Adam that also writes the copies the forward kernels read (round 3) -----------------------------
// hb_noisy_adam_multi_pack: the same optimizer step as noisy_adam_multiw_kernel<T, 4> (same arithmetic per element, same
// row-major `eff`), and in the same pass the 16-bit copies of the new effective weights that used to take two more launches
// per update: the k-contiguous (transposed) copy hb_thin_gemm reads (hb_actor_pack_weights before) and the fragment-major copy
// of the one-kernel actor (hb_actor_fused_pack before), plus the fp32 biases of the latter. A workgroup owns a 32 x 32 tile
// [k0 .. k0 + 31][n0 .. n0 + 31] of a weight tensor: thread t updates 4 consecutive columns of row t / 8 (128-byte row pieces),
// the merged values go through a 2 KB LDS tile, and 128 threads each emit one 16-byte vector of 8 consecutive k of one column —
// the unit both copies are made of. Bias tensors (one row) take the element-wise path and write their fp32 copy.
struct AdamPackOut {
  void* wt; int wt_ld;          // transposed copy [cols][wt_ld], or null
  uint4* frag; int frag_kind;   // 0 none, 1 = W1f [s][32 nt][64 lanes], 2 = W2f [pass][16 s][32 nt][64 lanes] (csrc/actor_fused.hip)
  const int* col_map;           // kind 2 and bias_f32 of the output layer: physical column of logit column j; null = identity
  float* bias_f32;              // bias tensors: fp32 copy of the ROUNDED effective bias (at col_map[j] when given), or null
};
struct AdamPackMulti {
  AdamArgs t[4];
  AdamPackOut o[4];
  int first[5];
  int tiles_n[4];               // column tiles of a tiled (weight) tensor; 0 = element-wise (bias) tensor
  int count;
};

template <typename T>
__global__ __launch_bounds__(256) void noisy_adam_pack_kernel(const AdamPackMulti m) {
  __shared__ uint16_t tile[8][132];
  int ti = 0;
  while (ti + 1 < m.count && static_cast<int>(blockIdx.x) >= m.first[ti + 1]) ++ti;
  const AdamArgs& a = m.t[ti];
  const AdamPackOut& o = m.o[ti];
  const int tid = static_cast<int>(threadIdx.x), tb = static_cast<int>(blockIdx.x) - m.first[ti];
  const float t = *a.step + a.step_offset;
  const float bc1 = 1.f - powf(a.b1, t), bc2s = sqrtf(1.f - powf(a.b2, t));
  T* eff = static_cast<T*>(a.eff);
  const int tn = m.tiles_n[ti];
  const int rows = static_cast<int>(a.n / a.cols);
  int k, n;                      // this thread's row and first column
  if (tn > 0) {
    k = (tb / tn) * 8 + (tid >> 5);
    n = (tb % tn) * 128 + 4 * (tid & 31);
  } else {
    k = 0;
    n = (tb * 256 + tid) * 4;
  }
  const bool valid = k < rows && n < a.cols;
  float mv[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    const uint32_t i = static_cast<uint32_t>(k) * static_cast<uint32_t>(a.cols) + n;
    float g[4], nz[4], res[3][4], lm[4], lv[4];
    load_gradw<4>(a.grad, a.grad_dtype, static_cast<uint32_t>(k) * static_cast<uint32_t>(a.grad_ld) + n, g);
    ldw<4>(a.noise, i, nz);
    auto one = [&](float* p, float* mp, float* vp, int which) {
      float pa[4], ma[4], va[4];
      ldw<4>(p, i, pa); ldw<4>(mp, i, ma); ldw<4>(vp, i, va);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pa[j] = adam1(pa[j], which == 2 ? __fmul_rn(g[j], nz[j]) : g[j], ma[j], va[j], a.b1, a.b2, bc1, bc2s, a.lr, a.eps);
        res[which][j] = pa[j];
        lm[j] = ma[j];
        lv[j] = va[j];
      }
      stw<4>(p, i, pa); stw<4>(mp, i, ma); stw<4>(vp, i, va);
    };
    one(a.w, a.m_w, a.v_w, 0);
    if (a.m_mu == a.m_w) {
      float pv[4];
      ldw<4>(a.w_mu, i, pv);
#pragma unroll
      for (int j = 0; j < 4; ++j) res[1][j] = pv[j] = __fsub_rn(pv[j], adam_step(lm[j], lv[j], bc1, bc2s, a.lr, a.eps));
      stw<4>(a.w_mu, i, pv);
    } else {
      one(a.w_mu, a.m_mu, a.v_mu, 1);
    }
    one(a.w_sigma, a.m_sg, a.v_sg, 2);
    const uint32_t e = static_cast<uint32_t>(k) * static_cast<uint32_t>(a.eff_ld) + n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mv[j] = merged(res[0][j], res[1][j], res[2][j], nz[j]);
      st<T>(eff, e + j, mv[j]);
    }
    if (tn == 0 && o.bias_f32) {   // the rounded value, as the packers read it back from `eff`
#pragma unroll
      for (int j = 0; j < 4; ++j) o.bias_f32[o.col_map ? o.col_map[n + j] : n + j] = ld<T>(eff, e + j);
    }
  }
  if (tn == 0 || (!o.wt && !o.frag_kind)) return;   // (workgroup-uniform)
  {
    T r4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) st<T>(r4, j, mv[j]);   // rows / columns past the tensor: zeros, like the packers' padding
    *reinterpret_cast<uint2*>(&tile[tid >> 5][4 * (tid & 31)]) = *reinterpret_cast<const uint2*>(r4);
  }
  __syncthreads();
  if (tid < 128) {
    const int col = tid;
    const int nn = (tb % tn) * 128 + col, k0 = (tb / tn) * 8;
    if (nn < a.cols) {
      uint16_t v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = tile[i][col];
      const uint4 vec = *reinterpret_cast<const uint4*>(v);
      if (o.wt) *reinterpret_cast<uint4*>(static_cast<uint16_t*>(o.wt) + static_cast<size_t>(nn) * o.wt_ld + k0) = vec;
      const int s = k0 >> 5, ch = (k0 >> 3) & 3;
      if (o.frag_kind == 1) {
        o.frag[(s * 32 + (nn >> 4)) * 64 + ch * 16 + (nn & 15)] = vec;
      } else if (o.frag_kind == 2) {
        const int ph = o.col_map[nn];
        o.frag[(((ph >> 9) * 16 + s) * 32 + ((ph & 511) >> 4)) * 64 + ch * 16 + (ph & 15)] = vec;
      }
    }
  }
}

template <typename T>
void launch_adam(const AdamArgs& a, hipStream_t s) {
  long long blocks = (a.n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((noisy_adam_kernel<T>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
}

}  // namespace

extern "C" {

static int replay_gather_impl(const int8_t* ring_obs_tm1_dev, const int8_t* ring_obs_t_dev, const int8_t* ring_act_dev,
                     const float* ring_rew_dev, const uint8_t* ring_term_dev, const int64_t* idx_dev, int64_t batch,
                     int32_t obs_len, void* x_dev, int32_t x_dtype, int32_t x_ld, int32_t* act_dev, float* rew_dev,
                     float* term_dev, float* disc_dev, int32_t n_step, float gamma, int64_t capacity, int64_t rows_per_insert,
                     const int64_t* size_wp_dev, void* stream, int packed_words) {
  if (!ring_obs_tm1_dev || !ring_obs_t_dev || !ring_act_dev || !ring_rew_dev || !ring_term_dev || !idx_dev || !x_dev ||
      !act_dev || !rew_dev || !term_dev || !disc_dev)
    return fail(HB_ERR_INVALID, "null argument");
  if (batch <= 0) return HB_OK;
  if (x_ld < obs_len) return fail(HB_ERR_INVALID, "x_ld must be >= obs_len");
  if (n_step < 1) return fail(HB_ERR_INVALID, "n_step must be >= 1");
  if (n_step > 1 && (!size_wp_dev || rows_per_insert < 1 || capacity < 1))
    return fail(HB_ERR_INVALID, "n_step > 1 needs size_wp_dev, rows_per_insert and capacity");
  const NStep ns{reinterpret_cast<const long long*>(size_wp_dev), capacity, rows_per_insert, n_step, gamma};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(2 * batch)), block(256);
  const int B = static_cast<int>(batch);
  if (x_dtype == 0)
    hipLaunchKernelGGL((gather_kernel<float>), grid, block, 0, s, ring_obs_tm1_dev, ring_obs_t_dev, ring_act_dev, ring_rew_dev, ring_term_dev, idx_dev, B, obs_len, static_cast<float*>(x_dev), x_ld, act_dev, rew_dev, term_dev, disc_dev, ns, packed_words);
  else if (x_dtype == 1)
    hipLaunchKernelGGL((gather_kernel<__hip_bfloat16>), grid, block, 0, s, ring_obs_tm1_dev, ring_obs_t_dev, ring_act_dev, ring_rew_dev, ring_term_dev, idx_dev, B, obs_len, static_cast<__hip_bfloat16*>(x_dev), x_ld, act_dev, rew_dev, term_dev, disc_dev, ns, packed_words);
  else if (x_dtype == 2)
    hipLaunchKernelGGL((gather_kernel<__half>), grid, block, 0, s, ring_obs_tm1_dev, ring_obs_t_dev, ring_act_dev, ring_rew_dev, ring_term_dev, idx_dev, B, obs_len, static_cast<__half*>(x_dev), x_ld, act_dev, rew_dev, term_dev, disc_dev, ns, packed_words);
  else
    return fail(HB_ERR_INVALID, "x_dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_replay_gather(const int8_t* ring_obs_tm1_dev, const int8_t* ring_obs_t_dev, const int8_t* ring_act_dev,
                     const float* ring_rew_dev, const uint8_t* ring_term_dev, const int64_t* idx_dev, int64_t batch,
                     int32_t obs_len, void* x_dev, int32_t x_dtype, int32_t x_ld, int32_t* act_dev, float* rew_dev,
                     float* term_dev, float* disc_dev, int32_t n_step, float gamma, int64_t capacity, int64_t rows_per_insert,
                     const int64_t* size_wp_dev, void* stream) {
  return replay_gather_impl(ring_obs_tm1_dev, ring_obs_t_dev, ring_act_dev, ring_rew_dev, ring_term_dev, idx_dev, batch, obs_len,
                            x_dev, x_dtype, x_ld, act_dev, rew_dev, term_dev, disc_dev, n_step, gamma, capacity, rows_per_insert,
                            size_wp_dev, stream, 0);
}

int hb_replay_gather_packed(const uint32_t* ring_bits_tm1_dev, const uint32_t* ring_bits_t_dev, const int8_t* ring_act_dev,
                            const float* ring_rew_dev, const uint8_t* ring_term_dev, const int64_t* idx_dev, int64_t batch,
                            int32_t obs_len, void* x_dev, int32_t x_dtype, int32_t x_ld, int32_t* act_dev, float* rew_dev,
                            float* term_dev, float* disc_dev, int32_t n_step, float gamma, int64_t capacity,
                            int64_t rows_per_insert, const int64_t* size_wp_dev, void* stream) {
  if (obs_len < 1) return fail(HB_ERR_INVALID, "obs_len must be positive");
  return replay_gather_impl(reinterpret_cast<const int8_t*>(ring_bits_tm1_dev), reinterpret_cast<const int8_t*>(ring_bits_t_dev),
                            ring_act_dev, ring_rew_dev, ring_term_dev, idx_dev, batch, obs_len, x_dev, x_dtype, x_ld, act_dev,
                            rew_dev, term_dev, disc_dev, n_step, gamma, capacity, rows_per_insert, size_wp_dev, stream,
                            (obs_len + 31) / 32);
}

int hb_c51_loss_grad(const void* logits_online_dev, const void* logits_target_dev, int32_t dtype, const int32_t* act_dev,
                     const float* rew_dev, const float* term_dev, const double* prios_dev, const float* beta_dev,
                     const float* disc_dev, int32_t mask_terminal, const float* support_dev, int64_t batch, int32_t n_actions,
                     int32_t n_atoms, int32_t row_stride, float* td_dev, float* w_dev, void* dlogits_dev, float* update_counter_dev,
                     const void* bias_online_dev, const void* bias_target_dev, void* stream) {
  if (!logits_online_dev || !logits_target_dev || !act_dev || !rew_dev || !term_dev || !prios_dev || !beta_dev || !disc_dev ||
      !support_dev || !td_dev || !w_dev || !dlogits_dev)
    return fail(HB_ERR_INVALID, "null argument");
  if (n_actions < 1 || n_actions > 64 || n_atoms < 2 || n_atoms > 64) return fail(HB_ERR_INVALID, "need n_actions <= 64 and 2 <= n_atoms <= 64");
  if (batch <= 0) return HB_OK;
  if (row_stride < n_actions * n_atoms) return fail(HB_ERR_INVALID, "row_stride must be >= n_actions * n_atoms");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>((batch + 3) / 4)), block(256);
  const size_t lds = 4 * (static_cast<size_t>(n_actions) * n_atoms + 64) * sizeof(float);
  const int B = static_cast<int>(batch);
#define HB_C51(T)                                                                                                       \
  hipLaunchKernelGGL((c51_kernel<T>), grid, block, lds, s, static_cast<const T*>(logits_online_dev),                    \
                     static_cast<const T*>(logits_target_dev), act_dev, rew_dev, term_dev, prios_dev, beta_dev, disc_dev, \
                     mask_terminal, support_dev, B, n_actions, n_atoms, row_stride, td_dev, w_dev, static_cast<T*>(dlogits_dev),       \
                     update_counter_dev, static_cast<const T*>(bias_online_dev), static_cast<const T*>(bias_target_dev))
  if (dtype == 0) HB_C51(float);
  else if (dtype == 1) HB_C51(__hip_bfloat16);
  else if (dtype == 2) HB_C51(__half);
  else return fail(HB_ERR_INVALID, "dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
#undef HB_C51
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_colsum(const void* x_dev, int32_t dtype, int64_t rows, int64_t cols, float* out_dev, void* stream) {
  if (!x_dev || !out_dev) return fail(HB_ERR_INVALID, "null argument");
  if (rows <= 0 || cols <= 0) return HB_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>((cols + 63) / 64)), block(1024);
  const int r = static_cast<int>(rows), c = static_cast<int>(cols);
  if (dtype == 0) hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, s, static_cast<const float*>(x_dev), r, c, out_dev);
  else if (dtype == 1) hipLaunchKernelGGL((colsum_kernel<__hip_bfloat16>), grid, block, 0, s, static_cast<const __hip_bfloat16*>(x_dev), r, c, out_dev);
  else if (dtype == 2) hipLaunchKernelGGL((colsum_kernel<__half>), grid, block, 0, s, static_cast<const __half*>(x_dev), r, c, out_dev);
  else return fail(HB_ERR_INVALID, "dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_noisy_adam(float* w_dev, float* w_mu_dev, float* w_sigma_dev, const float* noise_dev, const float* grad_dev,
                  float* m_w_dev, float* v_w_dev, float* m_mu_dev, float* v_mu_dev, float* m_sigma_dev, float* v_sigma_dev,
                  const float* step_dev, void* eff_dev, int32_t eff_dtype, int64_t n, int32_t cols, int32_t eff_ld, float lr,
                  float beta1, float beta2, float eps, void* stream) {
  if (!w_dev || !w_mu_dev || !w_sigma_dev || !noise_dev || !grad_dev || !m_w_dev || !v_w_dev || !m_mu_dev || !v_mu_dev ||
      !m_sigma_dev || !v_sigma_dev || !step_dev || !eff_dev)
    return fail(HB_ERR_INVALID, "null argument");
  if (n <= 0) return HB_OK;
  if (cols < 1 || eff_ld < cols || n % cols) return fail(HB_ERR_INVALID, "need cols >= 1, eff_ld >= cols and n a multiple of cols");
  AdamArgs a{w_dev, w_mu_dev, w_sigma_dev, noise_dev, grad_dev, m_w_dev, v_w_dev, m_mu_dev, v_mu_dev, m_sigma_dev, v_sigma_dev,
             step_dev, eff_dev, n, cols, eff_ld, 0, cols, lr, beta1, beta2, eps, 1.f};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eff_dtype == 0) launch_adam<float>(a, s);
  else if (eff_dtype == 1) launch_adam<__hip_bfloat16>(a, s);
  else if (eff_dtype == 2) launch_adam<__half>(a, s);
  else return fail(HB_ERR_INVALID, "eff_dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_relu_bwd_colsum(void* dy_dev, const void* act_dev, int64_t act_ld, int32_t dtype, int64_t rows, int64_t cols,
                       float* out_dev, void* stream) {
  if (!dy_dev || !act_dev || !out_dev) return fail(HB_ERR_INVALID, "null argument");
  if (rows <= 0 || cols <= 0) return HB_OK;
  if (act_ld == 0) act_ld = cols;
  if (act_ld < cols) return fail(HB_ERR_INVALID, "act_ld must be 0 or >= cols");
  const int al = static_cast<int>(act_ld);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>((cols + 63) / 64)), block(1024);
  const int r = static_cast<int>(rows), c = static_cast<int>(cols);
  if (dtype == 0) hipLaunchKernelGGL((relu_bwd_colsum_kernel<float>), grid, block, 0, s, static_cast<float*>(dy_dev), static_cast<const float*>(act_dev), r, c, al, out_dev);
  else if (dtype == 1) hipLaunchKernelGGL((relu_bwd_colsum_kernel<__hip_bfloat16>), grid, block, 0, s, static_cast<__hip_bfloat16*>(dy_dev), static_cast<const __hip_bfloat16*>(act_dev), r, c, al, out_dev);
  else if (dtype == 2) hipLaunchKernelGGL((relu_bwd_colsum_kernel<__half>), grid, block, 0, s, static_cast<__half*>(dy_dev), static_cast<const __half*>(act_dev), r, c, al, out_dev);
  else return fail(HB_ERR_INVALID, "dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_noisy_adam_multi(const hb_adam_tensor* tensors, int32_t count, const float* step_dev, float step_offset, int32_t eff_dtype,
                        float lr, float beta1, float beta2, float eps, void* stream) {
  if (!tensors || !step_dev) return fail(HB_ERR_INVALID, "null argument");
  if (count < 1 || count > 8) return fail(HB_ERR_INVALID, "count must be 1..8");
  AdamMulti m{};
  m.count = count;
  int blocks = 0;
  bool vec4 = true;  // several elements per thread when every tensor allows 16-byte accesses
  // Elements per thread: 4 by default; HB_ADAM_WIDTH=2 is the 44-register form that fits beside the one-kernel actor, =1 the
  // scalar one (36 registers). Same arithmetic per element in all three. Measured in the benched loop (round 3, same box,
  // alternating): 4 -> 0.112-0.114 ms per step, 2 -> 0.120, 1 -> 0.117: an Adam that shares the CUs with the policy kernel
  // streams its 55 MB through the L2 the policy kernel's weights live in and itself runs at one workgroup per CU; waiting for
  // the CUs is faster. The narrow forms stay for measurements (profiles/r03/README.md).
  const int width = [] { const char* e = getenv("HB_ADAM_WIDTH"); const int w = e ? atoi(e) : 4; return w == 2 || w == 1 ? w : 4; }();
  if (width == 1) vec4 = false;
  for (int i = 0; i < count; ++i) {
    const hb_adam_tensor& d = tensors[i];
    const int gl = d.grad_ld ? d.grad_ld : d.cols;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    // (32-bit element offsets inside the vector kernels: n < 2^30, rows * row stride < 2^31)
    vec4 = vec4 && d.cols > 0 && d.cols % 4 == 0 && d.n % 4 == 0 && d.n < (1LL << 30) &&
           (d.n / d.cols) * static_cast<long long>(d.eff_ld > gl ? d.eff_ld : gl) < (1LL << 31) && d.eff_ld % 4 == 0 && gl % 4 == 0 && al16(d.w) &&
           al16(d.w_mu) && al16(d.w_sigma) && al16(d.noise) && al16(d.grad) && al16(d.m_w) && al16(d.v_w) && al16(d.m_mu) &&
           al16(d.v_mu) && al16(d.m_sigma) && al16(d.v_sigma);
  }
  for (int i = 0; i < count; ++i) {
    const hb_adam_tensor& d = tensors[i];
    if (!d.w || !d.w_mu || !d.w_sigma || !d.noise || !d.grad || !d.m_w || !d.v_w || !d.m_mu || !d.v_mu || !d.m_sigma ||
        !d.v_sigma || !d.eff)
      return fail(HB_ERR_INVALID, "null pointer in tensor %d", i);
    if (d.n <= 0 || d.cols < 1 || d.eff_ld < d.cols || d.n % d.cols) return fail(HB_ERR_INVALID, "bad shape in tensor %d", i);
    if (d.grad_dtype < 0 || d.grad_dtype > 2 || (d.grad_ld != 0 && d.grad_ld < d.cols))
      return fail(HB_ERR_INVALID, "bad gradient dtype / row stride in tensor %d", i);
    m.t[i] = AdamArgs{d.w, d.w_mu, d.w_sigma, d.noise, d.grad, d.m_w, d.v_w, d.m_mu, d.v_mu, d.m_sigma, d.v_sigma,
                      step_dev, d.eff, d.n, d.cols, d.eff_ld, d.grad_dtype, d.grad_ld ? d.grad_ld : d.cols, lr, beta1, beta2, eps, step_offset};
    m.first[i] = blocks;
    long long nb = ((vec4 ? d.n / width : d.n) + 255) / 256;
    if (!vec4 && nb > 1024) nb = 1024;   // (the vector kernels take one group per thread, the scalar one strides)
    blocks += static_cast<int>(nb);
  }
  m.first[count] = blocks;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec4 && width == 4 && eff_dtype == 0) hipLaunchKernelGGL((noisy_adam_multiw_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, m);
  else if (vec4 && width == 4 && eff_dtype == 1) hipLaunchKernelGGL((noisy_adam_multiw_kernel<__hip_bfloat16, 4>), dim3(blocks), dim3(256), 0, s, m);
  else if (vec4 && width == 4 && eff_dtype == 2) hipLaunchKernelGGL((noisy_adam_multiw_kernel<__half, 4>), dim3(blocks), dim3(256), 0, s, m);
  else if (vec4 && eff_dtype == 0) hipLaunchKernelGGL((noisy_adam_multiw_kernel<float, 2>), dim3(blocks), dim3(256), 0, s, m);
  else if (vec4 && eff_dtype == 1) hipLaunchKernelGGL((noisy_adam_multiw_kernel<__hip_bfloat16, 2>), dim3(blocks), dim3(256), 0, s, m);
  else if (vec4 && eff_dtype == 2) hipLaunchKernelGGL((noisy_adam_multiw_kernel<__half, 2>), dim3(blocks), dim3(256), 0, s, m);
  else if (eff_dtype == 0) hipLaunchKernelGGL((noisy_adam_multi_kernel<float>), dim3(blocks), dim3(256), 0, s, m);
  else if (eff_dtype == 1) hipLaunchKernelGGL((noisy_adam_multi_kernel<__hip_bfloat16>), dim3(blocks), dim3(256), 0, s, m);
  else if (eff_dtype == 2) hipLaunchKernelGGL((noisy_adam_multi_kernel<__half>), dim3(blocks), dim3(256), 0, s, m);
  else return fail(HB_ERR_INVALID, "eff_dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
  HB_HIP(hipGetLastError());
  return HB_OK;
}


int hb_noisy_adam_multi_pack(const hb_adam_tensor* tensors, const hb_adam_pack* packs, int32_t count, const float* step_dev,
                             float step_offset, int32_t eff_dtype, float lr, float beta1, float beta2, float eps, void* stream) {
  if (!tensors || !packs || !step_dev) return fail(HB_ERR_INVALID, "null argument");
  if (count < 1 || count > 4) return fail(HB_ERR_INVALID, "count must be 1..4");
  if (eff_dtype != 1 && eff_dtype != 2) return fail(HB_ERR_INVALID, "eff_dtype must be 1 (bf16) or 2 (f16): the copies are 16-bit");
  AdamPackMulti m{};
  m.count = count;
  int blocks = 0;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  for (int i = 0; i < count; ++i) {
    const hb_adam_tensor& d = tensors[i];
    const hb_adam_pack& k = packs[i];
    if (!d.w || !d.w_mu || !d.w_sigma || !d.noise || !d.grad || !d.m_w || !d.v_w || !d.m_mu || !d.v_mu || !d.m_sigma ||
        !d.v_sigma || !d.eff)
      return fail(HB_ERR_INVALID, "null pointer in tensor %d", i);
    if (d.n <= 0 || d.cols < 1 || d.eff_ld < d.cols || d.n % d.cols) return fail(HB_ERR_INVALID, "bad shape in tensor %d", i);
    const int gl = d.grad_ld ? d.grad_ld : d.cols;
    const long long rows = d.n / d.cols;
    if (d.grad_dtype < 0 || d.grad_dtype > 2 || gl < d.cols) return fail(HB_ERR_INVALID, "bad gradient dtype / row stride in tensor %d", i);
    // 16-byte accesses and 32-bit element offsets, as in the 4-wide kernel
    if (d.cols % 4 || d.n >= (1LL << 30) || rows * static_cast<long long>(d.eff_ld > gl ? d.eff_ld : gl) >= (1LL << 31) || d.eff_ld % 4 ||
        gl % 4 || !al16(d.w) || !al16(d.w_mu) || !al16(d.w_sigma) || !al16(d.noise) || !al16(d.grad) || !al16(d.m_w) || !al16(d.v_w) ||
        !al16(d.m_mu) || !al16(d.v_mu) || !al16(d.m_sigma) || !al16(d.v_sigma) || (reinterpret_cast<uintptr_t>(d.eff) & 7u))
      return fail(HB_ERR_ALIGN, "tensor %d: needs cols, row strides %% 4 == 0 and 16-byte aligned arrays", i);
    const bool tiled = rows > 1;
    if (tiled) {
      if (k.bias_f32) return fail(HB_ERR_INVALID, "tensor %d: bias_f32 is for one-row (bias) tensors", i);
      if (k.frag_kind < 0 || k.frag_kind > 2 || (k.frag_kind != 0) != (k.frag != nullptr)) return fail(HB_ERR_INVALID, "tensor %d: frag / frag_kind", i);
      if (k.frag_kind == 2 && (!k.col_map_dev || rows % 32)) return fail(HB_ERR_INVALID, "tensor %d: the output layer's copy needs col_map and rows %% 32 == 0", i);
      if (k.frag_kind == 1 && d.cols != 512) return fail(HB_ERR_INVALID, "tensor %d: the first layer's fragment copy is for 512 hidden units", i);
      if (k.wt && (k.wt_ld < (rows + 7) / 8 * 8 || k.wt_ld % 8 || !al16(k.wt))) return fail(HB_ERR_INVALID, "tensor %d: wt_ld must cover the rows rounded up to 8 (and be a multiple of 8), wt 16-byte aligned", i);
      if (k.frag && !al16(k.frag)) return fail(HB_ERR_ALIGN, "tensor %d: frag must be 16-byte aligned", i);
    } else if (k.wt || k.frag || k.frag_kind) {
      return fail(HB_ERR_INVALID, "tensor %d: a one-row tensor has no transposed / fragment copy", i);
    }
    m.t[i] = AdamArgs{d.w, d.w_mu, d.w_sigma, d.noise, d.grad, d.m_w, d.v_w, d.m_mu, d.v_mu, d.m_sigma, d.v_sigma,
                      step_dev, d.eff, d.n, d.cols, d.eff_ld, d.grad_dtype, gl, lr, beta1, beta2, eps, step_offset};
    m.o[i] = AdamPackOut{k.wt, k.wt_ld, static_cast<uint4*>(k.frag), k.frag_kind, k.col_map_dev, k.bias_f32};
    m.first[i] = blocks;
    if (tiled) {
      m.tiles_n[i] = (d.cols + 127) / 128;
      blocks += static_cast<int>((rows + 7) / 8) * m.tiles_n[i];
    } else {
      m.tiles_n[i] = 0;
      blocks += static_cast<int>((d.n / 4 + 255) / 256);
    }
  }
  m.first[count] = blocks;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eff_dtype == 1) hipLaunchKernelGGL((noisy_adam_pack_kernel<__hip_bfloat16>), dim3(blocks), dim3(256), 0, s, m);
  else hipLaunchKernelGGL((noisy_adam_pack_kernel<__half>), dim3(blocks), dim3(256), 0, s, m);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

}  // extern "C"
