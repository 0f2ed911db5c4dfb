// learner2.hip — the C51 loss and its backward pass through the output layer, exploiting what the loss looks like.
//
// d mean(td * w) / d logits (hanabi_agents/rlax_dqn/rlax_rainbow.py:172-200) is non-zero only in the K = 51 atoms of the
// action a_tm1 each sample took: of the [B, A*K] = [256, 1020] matrix, 5 % is populated. The first fused learner
// (learner.hip) wrote it out densely and fed it to three dense launches (dW2 GEMM, dH GEMM, column sums) plus a ReLU
// backward. Here:
//
//   hb_c51_loss_sparse   the loss kernel, one wavefront per sample with the selector's softmax expectations spread over all
//                        64 lanes; emits td, the IS weights and the COMPACT gradient dl [B, 64] fp32 (k >= K zero)
//   hb_c51_backward      ONE launch for everything between the loss and the first layer's weight gradient:
//                          dH [B, H]   = relu'(H) * sum_k dl[b, k] * W2[j, a_b*K + k]      (+ db1 = its column sums)
//                          dW2 [H, A*K] slice of action a = sum over the samples that took a of H[b, :]^T dl[b, :]
//                          db2 [A*K]    slice of action a = sum over those samples of dl[b, :]
//                        all sums in a fixed order (sample index ascending): bit-reproducible, no atomics
//
// Against the dense chain this removes two GEMMs, two reduction kernels and their launch gaps from the learner's critical
// path (in the self-play loop: 9.8 + 9.4 + 5.3 + 15.6 us of kernels -> one ~6 us launch) and keeps dLoss/dlogits in fp32
// instead of rounding it to the GEMM dtype.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdint>
#include <type_traits>

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

template <typename T> __device__ __forceinline__ float rounded(float v);  // v after a round trip through T
template <> __device__ __forceinline__ float rounded<float>(float v) { return v; }
template <> __device__ __forceinline__ float rounded<__hip_bfloat16>(float v) { return __bfloat162float(__float2bfloat16(v)); }
template <> __device__ __forceinline__ float rounded<__half>(float v) { return __half2float(__float2half(v)); }

__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int DL_LD = 64;  // row stride of the compact gradient (K <= 64 atoms, zero padded)

// ---------------------------------------------------------------------------------------------------------------------
// one wavefront (= one workgroup) per sample; A <= 64, K <= 64
template <typename T>
__global__ __launch_bounds__(64) void c51_sparse_kernel(const T* __restrict__ logits_on, const T* __restrict__ logits_t,
                                                        const int32_t* __restrict__ act, const float* __restrict__ rew,
                                                        const float* __restrict__ term, const double* __restrict__ prios,
                                                        const float* __restrict__ beta_dev, const float* __restrict__ disc,
                                                        int mask_terminal, const float* __restrict__ support, int B, int A, int K, int rs,
                                                        float* __restrict__ td_out, float* __restrict__ w_out, float* __restrict__ dl,
                                                        float* __restrict__ counter, const T* __restrict__ bias_on,
                                                        const T* __restrict__ bias_t) {
  extern __shared__ float sel[];  // A*K + 64 floats
  const int lane = threadIdx.x, b = blockIdx.x;
  if (counter && b == 0 && lane == 0) *counter += 1.f;  // nothing else touches it during this kernel
  const int AK = A * K;
  // ---- the loads that nothing depends on go first: selector row (online net on obs_t), action, reward, discount
  const T* row_sel = logits_on + static_cast<long long>(B + b) * rs;
  for (int e = lane; e < AK; e += 64) sel[e] = ld<T>(row_sel, e) + (bias_on ? ld<T>(bias_on, e) : 0.f);
  const int a_tm1 = act[b];
  const float r_b = rew[b];
  float gamma = disc[b];
  if (mask_terminal) gamma *= 1.f - term[b];
  // ---- importance weight ((1/P).astype(f32)) ** beta / max over the batch (rlax_rainbow.py:188-189). x -> x ** beta is
  // monotone, so the maximum is attained at the smallest (beta > 0) or largest (beta < 0) probability: two pow calls
  // instead of B per wavefront
  const float beta = *beta_dev;
  float ipmin = INFINITY, ipmax = 0.f;
  for (int j = lane; j < B; j += 64) {
    const float ip = static_cast<float>(1.0 / prios[j]);
    ipmin = fminf(ipmin, ip);
    ipmax = fmaxf(ipmax, ip);
  }
  ipmin = wave_min(ipmin);
  ipmax = wave_max(ipmax);
  const float wmax = fmaxf(powf(ipmin, beta), powf(ipmax, beta));
  const float w_b = powf(static_cast<float>(1.0 / prios[b]), beta) / wmax;
  wave_fence();
  // ---- double-Q selector: q_sel[a] = mean(softmax(online(obs_t))[a] * z) (no legal mask, as the reference). S lanes share
  // one action's K atoms (S = 3 for 20 actions: 17 atoms each instead of 51 in a row)
  const int S = 64 / A >= 1 ? 64 / A : 1;
  const int a_l = lane / S, part = lane - a_l * S;
  const bool on = a_l < A;
  const int per = (K + S - 1) / S, k0 = part * per, k1 = k0 + per < K ? k0 + per : K;
  const float* r = sel + (on ? a_l : 0) * K;
  float m = -INFINITY;
  if (on)
    for (int k = k0; k < k1; ++k) m = fmaxf(m, r[k]);
  float ma = m;
  for (int i = 0; i < S; ++i) ma = fmaxf(ma, __shfl(m, a_l * S + i));  // (lanes of inactive actions read their own garbage: unused)
  float s = 0.f, t = 0.f;
  if (on)
    for (int k = k0; k < k1; ++k) {
      const float e = __expf(r[k] - ma);
      s += e;
      t += e * support[k];
    }
  float sa = 0.f, ta = 0.f;
  for (int i = 0; i < S; ++i) {  // fixed order: part 0, 1, 2
    sa += __shfl(s, a_l * S + i);
    ta += __shfl(t, a_l * S + i);
  }
  const float q = on ? ta / sa / static_cast<float>(K) : -INFINITY;
  const float qbest = wave_max(q);
  const unsigned long long hit = __ballot(on && part == 0 && q == qbest);
  const int a_star = (__ffsll(static_cast<long long>(hit)) - 1) / S;  // lowest action index among ties, like argmax
  // ---- target distribution p = softmax(target(obs_t)[a*]) and its projection onto the support
  const bool atom = lane < K;
  const float lt = atom ? ld<T>(logits_t, static_cast<long long>(b) * rs + a_star * K + lane) +
                              (bias_t ? ld<T>(bias_t, a_star * K + lane) : 0.f)
                        : -INFINITY;
  // (issued here so its latency hides under the projection loop)
  const float l1 = atom ? ld<T>(logits_on, static_cast<long long>(b) * rs + a_tm1 * K + lane) +
                              (bias_on ? ld<T>(bias_on, a_tm1 * K + lane) : 0.f)
                        : -INFINITY;
  const float mt = wave_max(lt);
  const float et = atom ? __expf(lt - mt) : 0.f;
  const float p = et / wave_sum(et);
  const float vmin = support[0], vmax = support[K - 1];
  const float delta = (vmax - vmin) / static_cast<float>(K - 1);
  const float zi = atom ? support[lane] : 0.f;
  const float tz = fminf(fmaxf(r_b + gamma * zi, vmin), vmax);  // clipped r + gamma * z_j held by lane j
  float target = 0.f;
  const float inv_delta = 1.f / delta;
  for (int j = 0; j < K; ++j) {  // j is wave-uniform: the broadcasts are v_readlane, not LDS permutes
    const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), j));
    const float tzj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tz), j));
    target += pj * fminf(fmaxf(1.f - fabsf(tzj - zi) * inv_delta, 0.f), 1.f);
  }
  if (!atom) target = 0.f;
  // ---- cross-entropy against log_softmax(online(obs_tm1)[a_tm1]) and its gradient
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
  dl[static_cast<long long>(b) * DL_LD + lane] = atom ? (w_b / static_cast<float>(B)) * (e1 / s1 * tsum - target) : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
struct BwdArgs {
  const float* dl;        // [B, 64]
  const int32_t* act;     // [B]
  const void* h;          // post-ReLU hidden activations of obs_tm1: [B, h_ld] (GEMM dtype)
  const void* w2;         // effective output-layer weights [H, w2_ld] (GEMM dtype), columns a*K + k
  void* dh;               // out [B, H] (GEMM dtype): dLoss/dhidden masked by the ReLU
  float* db1;             // out [H]
  void* dw2;              // out [H, dw2_ld] (GEMM dtype); columns >= A*K are not written
  float* db2;             // out [A*K]
  int B, H, A, K, h_ld, w2_ld, dw2_ld;
  int n_dh_tiles;         // workgroups [0, n_dh_tiles): dH tiles of JT hidden units; the rest: (action, 64-unit tile) of dW2
};

template <typename T, int JT>
__global__ __launch_bounds__(256) void c51_backward_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* h = static_cast<const T*>(a.h);
  if (static_cast<int>(blockIdx.x) < a.n_dh_tiles) {
    // ---- dH tile: hidden units j0 .. j0 + JT - 1 for every sample. The JT rows of W2 sit in LDS, every action's K atoms
    // in a slot of its own: 64 elements (k >= K zero, where dl is zero too) + padding to SLOT elements, chosen so that the
    // 16-byte reads of lanes that took different actions fall on different banks (72 x 2 B: bank offsets a * 36 mod 64,
    // 68 x 4 B: a * 4 mod 64 — all distinct multiples of 4 for 16 consecutive actions)
    constexpr int SLOT = sizeof(T) == 2 ? 72 : 68;
    constexpr int VEC = 16 / sizeof(T);            // elements per 16-byte read
    const int j0 = static_cast<int>(blockIdx.x) * JT;
    const int ld_w = a.A * SLOT;
    T* w2s = reinterpret_cast<T*>(smem);
    float* red = reinterpret_cast<float*>(smem + static_cast<size_t>(JT) * ld_w * sizeof(T));  // [4][JT]
    const T* w2 = static_cast<const T*>(a.w2);
    for (int e = tid; e < JT * ld_w; e += 256) {
      const int jj = e / ld_w, c = e - jj * ld_w;
      const int ac = c / SLOT, k = c - ac * SLOT;
      float v = 0.f;
      if (k < a.K && j0 + jj < a.H) v = ld<T>(w2, static_cast<long long>(j0 + jj) * a.w2_ld + ac * a.K + k);
      st<T>(w2s, e, v);
    }
    __syncthreads();
    float colsum[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) colsum[jj] = 0.f;
    T* dh = static_cast<T*>(a.dh);
    const int kc = (a.K + VEC - 1) / VEC;          // 16-byte chunks that hold atoms (7 of 8 for 51 bf16 atoms)
    for (int b = tid; b < a.B; b += 256) {         // ascending sample order per thread: fixed summation order
      float dlv[DL_LD];
      const float4* dlp = reinterpret_cast<const float4*>(a.dl + static_cast<long long>(b) * DL_LD);
#pragma unroll
      for (int i = 0; i < DL_LD / 4; ++i) {
        const float4 v = dlp[i];
        dlv[4 * i] = v.x; dlv[4 * i + 1] = v.y; dlv[4 * i + 2] = v.z; dlv[4 * i + 3] = v.w;
      }
      const int c0 = a.act[b] * SLOT;
#pragma unroll
      for (int jj = 0; jj < JT; ++jj) {
        const uint4* wr = reinterpret_cast<const uint4*>(w2s + jj * ld_w + c0);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DL_LD / VEC; ++c) {
          if (c < kc) {                            // wave-uniform
            const uint4 u = wr[c];
            const uint32_t d[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if constexpr (sizeof(T) == 4) {
                acc = fmaf(dlv[c * 4 + i], __uint_as_float(d[i]), acc);
              } else if constexpr (std::is_same<T, __hip_bfloat16>::value) {
                acc = fmaf(dlv[c * 8 + 2 * i], __uint_as_float(d[i] << 16), acc);
                acc = fmaf(dlv[c * 8 + 2 * i + 1], __uint_as_float(d[i] & 0xFFFF0000u), acc);
              } else {
                const __half2 hh = *reinterpret_cast<const __half2*>(&d[i]);
                acc = fmaf(dlv[c * 8 + 2 * i], __low2float(hh), acc);
                acc = fmaf(dlv[c * 8 + 2 * i + 1], __high2float(hh), acc);
              }
            }
          }
        }
        if (j0 + jj < a.H) {
          const long long o = static_cast<long long>(b) * a.H + j0 + jj;
          const float g = ld<T>(h, static_cast<long long>(b) * a.h_ld + j0 + jj) > 0.f ? acc : 0.f;
          st<T>(dh, o, g);
          colsum[jj] += rounded<T>(g);             // the column sum of dH as it is stored
        }
      }
    }
    // db1: threads -> waves (shuffle tree) -> workgroup (fixed order 0..3)
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
      const float v = wave_sum(colsum[jj]);
      if (lane == 0) red[wave * JT + jj] = v;
    }
    __syncthreads();
    if (tid < JT && j0 + tid < a.H) a.db1[j0 + tid] = ((red[tid] + red[JT + tid]) + red[2 * JT + tid]) + red[3 * JT + tid];
    return;
  }
  // ---- dW2 / db2 of one action for 64 hidden units: the samples that took this action, in ascending order
  const int id = static_cast<int>(blockIdx.x) - a.n_dh_tiles;
  const int n_jt = (a.H + 63) / 64;
  const int act_id = id / n_jt, j0 = (id - act_id * n_jt) * 64;
  int* list = reinterpret_cast<int*>(smem);          // up to B sample indices
  __shared__ int wave_cnt[4];
  __shared__ int total;
  int n_list = 0;
  for (int base = 0; base < a.B; base += 256) {      // (B = 256: one round)
    const int b = base + tid;
    const bool mine = b < a.B && a.act[b] == act_id;
    const unsigned long long mk = __ballot(mine);
    if (lane == 0) wave_cnt[wave] = __popcll(mk);
    __syncthreads();
    int off = n_list;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (mine) list[off + __popcll(mk & ((1ull << lane) - 1ull))] = b;
    n_list += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) total = n_list;
  __syncthreads();
  const int n = total;
  // thread -> (hidden unit jj = lane, 16 atoms kq*16 .. +15); lanes of a wave share the atoms (broadcast loads of dl)
  const int jj = lane, kq = wave;
  const int j = j0 + jj;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = 0; s < n; ++s) {
    const int b = list[s];
    const float hv = j < a.H ? static_cast<float>(h[static_cast<long long>(b) * a.h_ld + j]) : 0.f;
    const float4* dlp = reinterpret_cast<const float4*>(a.dl + static_cast<long long>(b) * DL_LD + kq * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = dlp[i];
      acc[4 * i] = fmaf(hv, v.x, acc[4 * i]);
      acc[4 * i + 1] = fmaf(hv, v.y, acc[4 * i + 1]);
      acc[4 * i + 2] = fmaf(hv, v.z, acc[4 * i + 2]);
      acc[4 * i + 3] = fmaf(hv, v.w, acc[4 * i + 3]);
    }
  }
  if (j < a.H) {
    T* out = static_cast<T*>(a.dw2) + static_cast<long long>(j) * a.dw2_ld + act_id * a.K + kq * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (kq * 16 + i < a.K) st<T>(out, i, acc[i]);
  }
  if (j0 == 0 && tid < a.K) {  // db2 slice: one thread per atom
    float sum = 0.f;
    for (int s = 0; s < n; ++s) sum += a.dl[static_cast<long long>(list[s]) * DL_LD + tid];
    a.db2[act_id * a.K + tid] = sum;
  }
}

template <typename T>
int launch_backward(const BwdArgs& a0, hipStream_t s) {
  BwdArgs a = a0;
  const size_t row = static_cast<size_t>(a.A) * (sizeof(T) == 2 ? 72 : 68) * sizeof(T);
  const int n_jt = (a.H + 63) / 64;
  const size_t list_bytes = static_cast<size_t>(a.B) * 4;
  // hidden units per dH tile: 8 (64 workgroups for 512 units; bf16, 20 actions: 23 KB of LDS), fewer if LDS is short
  int jt = 8;
  while (jt > 1 && jt * row + 4 * jt * 4 > 64 * 1024) jt >>= 1;
  a.n_dh_tiles = (a.H + jt - 1) / jt;
  const unsigned blocks = static_cast<unsigned>(a.n_dh_tiles + a.A * n_jt);
  size_t lds = jt * row + 4 * jt * 4;
  if (lds < list_bytes) lds = list_bytes;
  if (lds > 150 * 1024) return fail(HB_ERR_INVALID, "batch / output layer too large for the backward kernel's LDS tile");
#define HB_BWD(JT) hipLaunchKernelGGL((c51_backward_kernel<T, JT>), dim3(blocks), dim3(256), lds, s, a)
  if (jt == 8) HB_BWD(8);
  else if (jt == 4) HB_BWD(4);
  else if (jt == 2) HB_BWD(2);
  else HB_BWD(1);
#undef HB_BWD
  return HB_OK;
}

}  // namespace

extern "C" {

int hb_c51_loss_sparse(const void* logits_online_dev, const void* logits_target_dev, int32_t dtype, const int32_t* act_dev,
                       const float* rew_dev, const float* term_dev, const double* prios_dev, const float* beta_dev,
                       const float* disc_dev, int32_t mask_terminal, const float* support_dev, int64_t batch, int32_t n_actions,
                       int32_t n_atoms, int32_t row_stride, float* td_dev, float* w_dev, float* dl_dev, float* update_counter_dev,
                       const void* bias_online_dev, const void* bias_target_dev, void* stream) {
  if (!logits_online_dev || !logits_target_dev || !act_dev || !rew_dev || !term_dev || !prios_dev || !beta_dev || !disc_dev ||
      !support_dev || !td_dev || !w_dev || !dl_dev)
    return fail(HB_ERR_INVALID, "null argument");
  if (n_actions < 1 || n_actions > 64 || n_atoms < 2 || n_atoms > 64) return fail(HB_ERR_INVALID, "need n_actions <= 64 and 2 <= n_atoms <= 64");
  if (batch <= 0) return HB_OK;
  if (row_stride < n_actions * n_atoms) return fail(HB_ERR_INVALID, "row_stride must be >= n_actions * n_atoms");
  if (reinterpret_cast<uintptr_t>(dl_dev) & 15u) return fail(HB_ERR_ALIGN, "dl_dev must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(batch)), block(64);
  const size_t lds = (static_cast<size_t>(n_actions) * n_atoms + 64) * sizeof(float);
  const int B = static_cast<int>(batch);
#define HB_C51S(T)                                                                                                        \
  hipLaunchKernelGGL((c51_sparse_kernel<T>), grid, block, lds, s, static_cast<const T*>(logits_online_dev),               \
                     static_cast<const T*>(logits_target_dev), act_dev, rew_dev, term_dev, prios_dev, beta_dev, disc_dev, \
                     mask_terminal, support_dev, B, n_actions, n_atoms, row_stride, td_dev, w_dev, dl_dev, update_counter_dev,       \
                     static_cast<const T*>(bias_online_dev), static_cast<const T*>(bias_target_dev))
  if (dtype == 0) HB_C51S(float);
  else if (dtype == 1) HB_C51S(__hip_bfloat16);
  else if (dtype == 2) HB_C51S(__half);
  else return fail(HB_ERR_INVALID, "dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
#undef HB_C51S
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_c51_backward(const float* dl_dev, const int32_t* act_dev, const void* hidden_dev, int32_t hidden_ld, const void* w2_dev,
                    int32_t w2_ld, int32_t dtype, int64_t batch, int32_t hidden, int32_t n_actions, int32_t n_atoms, void* dh_dev,
                    float* db1_dev, void* dw2_dev, int32_t dw2_ld, float* db2_dev, void* stream) {
  if (!dl_dev || !act_dev || !hidden_dev || !w2_dev || !dh_dev || !db1_dev || !dw2_dev || !db2_dev)
    return fail(HB_ERR_INVALID, "null argument");
  if (n_actions < 1 || n_actions > 64 || n_atoms < 2 || n_atoms > 64) return fail(HB_ERR_INVALID, "need n_actions <= 64 and 2 <= n_atoms <= 64");
  if (batch <= 0) return HB_OK;
  if (batch > 16384) return fail(HB_ERR_INVALID, "batch must be <= 16384");
  const int AK = n_actions * n_atoms;
  if (hidden < 1 || hidden_ld < hidden || w2_ld < AK || dw2_ld < AK) return fail(HB_ERR_INVALID, "bad hidden size / row strides");
  if (reinterpret_cast<uintptr_t>(dl_dev) & 15u) return fail(HB_ERR_ALIGN, "dl_dev must be 16-byte aligned");
  BwdArgs a{dl_dev, act_dev, hidden_dev, w2_dev, dh_dev, db1_dev, dw2_dev, db2_dev, static_cast<int>(batch), hidden, n_actions,
            n_atoms, hidden_ld, w2_ld, dw2_ld, 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == 0) rc = launch_backward<float>(a, s);
  else if (dtype == 1) rc = launch_backward<__hip_bfloat16>(a, s);
  else if (dtype == 2) rc = launch_backward<__half>(a, s);
  else return fail(HB_ERR_INVALID, "dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
  if (rc) return rc;
  HB_HIP(hipGetLastError());
  return HB_OK;
}

}  // extern "C"
