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
#include <cstdlib>
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

// Wave-wide reductions on the DPP data path (no LDS round trips: a ds_bpermute-based __shfl_xor tree costs ~800 cycles per
// reduction on a lone wavefront, this ~80). quad_perm swaps, row rotations, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3 (gfx9 DPP); lane 63 ends up with the total, which is broadcast through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
#define HB_WAVE_REDUCE(NAME, OP, IDENT)                                                       \
  __device__ __forceinline__ float NAME(float v) {                                            \
    v = OP(v, dpp_mov<0xb1, 0xf>(v, v));   /* quad_perm:[1,0,3,2] */                          \
    v = OP(v, dpp_mov<0x4e, 0xf>(v, v));   /* quad_perm:[2,3,0,1] */                          \
    v = OP(v, dpp_mov<0x124, 0xf>(v, v));  /* row_ror:4 */                                    \
    v = OP(v, dpp_mov<0x128, 0xf>(v, v));  /* row_ror:8: every lane holds its row's result */ \
    v = OP(v, dpp_mov<0x142, 0xa>(IDENT, v));  /* row_bcast:15 -> rows 1, 3 */                \
    v = OP(v, dpp_mov<0x143, 0xc>(IDENT, v));  /* row_bcast:31 -> rows 2, 3 */                \
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));                  \
  }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
HB_WAVE_REDUCE(wave_max, fmaxf, -INFINITY)
HB_WAVE_REDUCE(wave_min, fminf, INFINITY)
HB_WAVE_REDUCE(wave_sum, op_add, 0.f)
#undef HB_WAVE_REDUCE
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int DL_LD = 64;  // row stride of the compact gradient (K <= 64 atoms, zero padded)

// eight 16-bit floats <-> one 16-byte vector, through registers only (no local arrays that end up in scratch)
template <typename T> __device__ __forceinline__ float half_lo(uint32_t d);
template <typename T> __device__ __forceinline__ float half_hi(uint32_t d);
template <> __device__ __forceinline__ float half_lo<__hip_bfloat16>(uint32_t d) { return __uint_as_float(d << 16); }
template <> __device__ __forceinline__ float half_hi<__hip_bfloat16>(uint32_t d) { return __uint_as_float(d & 0xFFFF0000u); }
template <> __device__ __forceinline__ float half_lo<__half>(uint32_t d) { return __low2float(*reinterpret_cast<const __half2*>(&d)); }
template <> __device__ __forceinline__ float half_hi<__half>(uint32_t d) { return __high2float(*reinterpret_cast<const __half2*>(&d)); }
template <> __device__ __forceinline__ float half_lo<float>(uint32_t d) { return __uint_as_float(d); }
template <> __device__ __forceinline__ float half_hi<float>(uint32_t d) { return 0.f; }
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<__hip_bfloat16>(float lo, float hi) {
  const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
  return static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&a)) | (static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}
template <> __device__ __forceinline__ uint32_t pack2<__half>(float lo, float hi) {
  const __half2 h = __floats2half2_rn(lo, hi);
  return *reinterpret_cast<const uint32_t*>(&h);
}
template <> __device__ __forceinline__ uint32_t pack2<float>(float lo, float) { return __float_as_uint(lo); }

// one logits row (+ its bias) -> LDS as fp32, 16 elements per lane at a time: all 16 (32 with the bias) loads of a round are
// issued before the first LDS write, so they share one memory round trip
template <typename T>
__device__ __forceinline__ void stage_row(const T* __restrict__ row, const T* __restrict__ bias, float* dst, int n, int lane) {
  for (int base = 0; base < n; base += 1024) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = base + lane + 64 * i;
      v[i] = e < n ? ld<T>(row, e) + (bias ? ld<T>(bias, e) : 0.f) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = base + lane + 64 * i;
      if (e < n) dst[e] = v[i];
    }
  }
}

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
  extern __shared__ float sel[];  // A*K floats: the selector's logits; 64: the support
  const int lane = threadIdx.x, b = blockIdx.x;
  if (counter && b == 0 && lane == 0) *counter += 1.f;  // nothing else touches it during this kernel
  const int AK = A * K;
  // ---- every global load of the kernel is issued up front, the per-sample scalars first (the action indexes the last one)
  const int a_tm1 = act[b];
  const float r_b = rew[b];
  float gamma = disc[b];
  if (mask_terminal) gamma *= 1.f - term[b];
  const bool atom = lane < K;
  const T* row_sel = logits_on + static_cast<long long>(B + b) * rs;     // online net on obs_t: the double-Q selector
  stage_row<T>(row_sel, bias_on, sel, AK, lane);
  float* ssup = sel + AK;
  ssup[lane] = lane < K ? support[lane] : 0.f;
  const float l1 = atom ? ld<T>(logits_on, static_cast<long long>(b) * rs + a_tm1 * K + lane) +
                              (bias_on ? ld<T>(bias_on, a_tm1 * K + lane) : 0.f)
                        : -INFINITY;
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
  const float wmax = powf(beta >= 0.f ? ipmax : ipmin, beta);
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
      t += e * ssup[k];
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
  // (staging the whole target row in LDS beside the selector's, to avoid this dependent load, measured slower: 12.8 vs 10.6 us)
  const float lt = atom ? ld<T>(logits_t, static_cast<long long>(b) * rs + a_star * K + lane) +
                              (bias_t ? ld<T>(bias_t, a_star * K + lane) : 0.f)
                        : -INFINITY;
  const float mt = wave_max(lt);
  const float et = atom ? __expf(lt - mt) : 0.f;
  const float p = et / wave_sum(et);
  const float vmin = ssup[0], vmax = ssup[K - 1];
  const float delta = (vmax - vmin) / static_cast<float>(K - 1);
  const float zi = ssup[lane];
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
  int block_offset;       // added to blockIdx.x (0; measurements launch the two halves separately)
};

constexpr int BWD_T = 1024;  // threads per workgroup of the backward kernel (16 wavefronts)

template <typename T, int JT>
__global__ __launch_bounds__(BWD_T) void c51_backward_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int wave_cnt[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* h = static_cast<const T*>(a.h);
  const int blk = static_cast<int>(blockIdx.x) + a.block_offset;
  if (blk < a.n_dh_tiles) {
    // ---- dH tile: hidden units j0 .. j0 + JT - 1 for every sample. The JT rows of W2 sit in LDS, every action's K atoms
    // in a slot of its own: 64 elements (k >= K zero, where dl is zero too) + padding to SLOT elements, so that 16-byte reads
    // of lanes whose samples took different actions spread over the banks (72 x 2 B: bank offsets a * 36 mod 64; 68 x 4 B:
    // a * 4 mod 64). FOUR lanes share one sample: lane part p multiplies the 16-byte chunks p, p + 4, ... of the 64-wide
    // dot product and the four partial sums are combined on the DPP path ((p0 + p1) + (p2 + p3): a fixed order).
    constexpr int SLOT = sizeof(T) == 2 ? 72 : 68;
    constexpr int VEC = 16 / sizeof(T);            // elements per 16-byte read
    constexpr int NCH = DL_LD / VEC;               // chunks of the dot product: 8 (16-bit) or 16 (fp32)
    constexpr int CPP = NCH / 4;                   // chunks per lane part
    const int j0 = blk * JT;
    const int ld_w = a.A * SLOT;
    T* w2s = reinterpret_cast<T*>(smem);
    float* red = reinterpret_cast<float*>(smem + static_cast<size_t>(JT) * ld_w * sizeof(T));  // [16][JT]
    const T* w2 = static_cast<const T*>(a.w2);
    // this lane's sample of the first round: action, its chunks of dl, the JT hidden activations. Requested BEFORE the tile is
    // filled, so the two memory round trips overlap
    const int part = tid & 3;
    constexpr bool VEC_IO = JT * sizeof(T) == 16;
    const bool vec = VEC_IO && j0 + JT <= a.H && ((a.h_ld | a.H) & (JT - 1)) == 0;
    float dlv[CPP][VEC];
    float hin[JT];   // this sample's JT hidden activations (only their sign is used)
    int c0 = 0;
    auto load_sample = [&](int base) {
      const int b = base + (tid >> 2);
      const int bb = b < a.B ? b : a.B - 1;
#pragma unroll
      for (int u = 0; u < CPP; ++u) {
        const float4* dlp = reinterpret_cast<const float4*>(a.dl + static_cast<long long>(bb) * DL_LD + (part + 4 * u) * VEC);
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i) {
          const float4 v = dlp[i];
          dlv[u][4 * i] = v.x; dlv[u][4 * i + 1] = v.y; dlv[u][4 * i + 2] = v.z; dlv[u][4 * i + 3] = v.w;
        }
      }
      c0 = a.act[bb] * SLOT;
      if (vec) {
        const uint4 u = *reinterpret_cast<const uint4*>(h + static_cast<long long>(bb) * a.h_ld + j0);
        const uint32_t d[4] = {u.x, u.y, u.z, u.w};
        if constexpr (sizeof(T) == 4) {            // JT = 4 floats
#pragma unroll
          for (int i = 0; i < JT; ++i) hin[i] = __uint_as_float(d[i & 3]);
        } else {                                   // JT = 8 sixteen-bit values
#pragma unroll
          for (int i = 0; i < JT / 2; ++i) {
            hin[2 * i] = half_lo<T>(d[i & 3]);
            hin[2 * i + 1] = half_hi<T>(d[i & 3]);
          }
        }
      } else {
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) hin[jj] = j0 + jj < a.H ? ld<T>(h, static_cast<long long>(bb) * a.h_ld + j0 + jj) : 0.f;
      }
    };
    load_sample(0);
    // fill: zero the whole tile (slot padding, atoms k >= K), then scatter the JT rows. No integer division (c / K through an
    // exact float reciprocal for c < 4096) and every global load independent of the others, so they are all in flight together
    for (int e = tid; e < JT * ld_w * static_cast<int>(sizeof(T)) / 16; e += BWD_T) reinterpret_cast<uint4*>(smem)[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    {
      const int AK = a.A * a.K;
      const float inv_k = 1.0f / static_cast<float>(a.K);
#pragma unroll
      for (int jj = 0; jj < JT; ++jj) {
        if (j0 + jj < a.H) {
          const T* src = w2 + static_cast<long long>(j0 + jj) * a.w2_ld;
          for (int c = tid; c < AK; c += BWD_T) {
            const int ac = static_cast<int>((static_cast<float>(c) + 0.5f) * inv_k);
            w2s[jj * ld_w + ac * SLOT + (c - ac * a.K)] = src[c];
          }
        }
      }
    }
    __syncthreads();
    const int kc = (a.K + VEC - 1) / VEC;          // 16-byte chunks that hold atoms (7 of 8 for 51 bf16 atoms)
    float colsum[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) colsum[jj] = 0.f;
    T* dh = static_cast<T*>(a.dh);
    for (int base = 0; base < a.B; base += BWD_T / 4) {   // ascending sample order per lane: fixed summation order
      const int b = base + (tid >> 2);
      const bool live = b < a.B;
      if (base > 0) load_sample(base);
      float gout[JT];
#pragma unroll
      for (int jj = 0; jj < JT; ++jj) {
        const uint4* wr = reinterpret_cast<const uint4*>(w2s + jj * ld_w + c0);
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < CPP; ++u) {
          const int c = part + 4 * u;
          if (c < kc) {
            const uint4 q = wr[c];
            const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if constexpr (sizeof(T) == 4) {
                acc = fmaf(dlv[u][i], __uint_as_float(d[i]), acc);
              } else if constexpr (std::is_same<T, __hip_bfloat16>::value) {
                acc = fmaf(dlv[u][2 * i], __uint_as_float(d[i] << 16), acc);
                acc = fmaf(dlv[u][2 * i + 1], __uint_as_float(d[i] & 0xFFFF0000u), acc);
              } else {
                const __half2 hh = *reinterpret_cast<const __half2*>(&d[i]);
                acc = fmaf(dlv[u][2 * i], __low2float(hh), acc);
                acc = fmaf(dlv[u][2 * i + 1], __high2float(hh), acc);
              }
            }
          }
        }
        acc += dpp_mov<0xb1, 0xf>(acc, acc);      // parts (0,1) and (2,3)
        acc += dpp_mov<0x4e, 0xf>(acc, acc);      // ... and the two pairs: every lane of the quad holds the dot product
        const float g = hin[jj] > 0.f ? acc : 0.f;
        gout[jj] = g;
        if (live && part == 0 && j0 + jj < a.H) colsum[jj] += rounded<T>(g);   // the column sum of dH as it is stored
      }
      if (live && part == 0) {
        if (vec) {
          uint32_t d[4] = {0u, 0u, 0u, 0u};
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int i = 0; i < JT; ++i) d[i & 3] = __float_as_uint(gout[i]);
          } else {
#pragma unroll
            for (int i = 0; i < JT / 2; ++i) d[i & 3] = pack2<T>(gout[2 * i], gout[2 * i + 1]);
          }
          *reinterpret_cast<uint4*>(dh + static_cast<long long>(b) * a.H + j0) = make_uint4(d[0], d[1], d[2], d[3]);
        } else {
#pragma unroll
          for (int jj = 0; jj < JT; ++jj)
            if (j0 + jj < a.H) st<T>(dh, static_cast<long long>(b) * a.H + j0 + jj, gout[jj]);
        }
      }
    }
    // db1: lanes -> wavefront (DPP tree) -> workgroup (wavefronts 0..15 in order)
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
      const float v = wave_sum(colsum[jj]);
      if (lane == 0) red[wave * JT + jj] = v;
    }
    __syncthreads();
    if (tid < JT && j0 + tid < a.H) {
      float t = 0.f;
      for (int w = 0; w < BWD_T / 64; ++w) t += red[w * JT + tid];
      a.db1[j0 + tid] = t;
    }
    return;
  }
  // ---- dW2 / db2 of one action for 64 hidden units: the samples that took this action, in ascending order
  const int id = blk - a.n_dh_tiles;
  const int n_jt = (a.H + 63) / 64;
  const int act_id = id / n_jt, j0 = (id - act_id * n_jt) * 64;
  int* list = reinterpret_cast<int*>(smem);          // up to B sample indices
  int n = 0;
  for (int base = 0; base < a.B; base += BWD_T) {
    const int b = base + tid;
    const bool mine = b < a.B && a.act[b] == act_id;
    const unsigned long long mk = __ballot(mine);
    if (lane == 0) wave_cnt[wave] = __popcll(mk);
    __syncthreads();
    int off = n, all = 0;
    for (int w = 0; w < BWD_T / 64; ++w) {
      if (w < wave) off += wave_cnt[w];
      all += wave_cnt[w];
    }
    if (mine) list[off + __popcll(mk & ((1ull << lane) - 1ull))] = b;
    n += all;
    __syncthreads();
  }
  // The samples of this action are split into 8 contiguous ranges (ascending), each handled by a PAIR of wavefronts (one per
  // half of the atoms): thread -> (hidden unit j0 + lane, atom half kh, sample range sq) accumulates 32 atoms over its range;
  // the 8 partial sums of every output are then added in range order through LDS. However many of the batch's samples took
  // this action (a greedy policy concentrates them on one), no thread walks more than n / 8 of them.
  const int kh = wave & 1, sq = wave >> 1;
  const int j = j0 + lane;
  const int per = (n + 7) >> 3;
  const int s_begin = sq * per, s_end = s_begin + per < n ? s_begin + per : n;
  // One memory round trip for everything the accumulation needs: this lane's hidden activations of its (at most 32) samples
  // into registers, and the dl rows of ALL the action's samples into LDS (every thread a few independent 16-byte loads)
  float* dls = reinterpret_cast<float*>(smem) + ((a.B + 3) & ~3);    // [n][64] behind the list, on a 16-byte boundary for ANY batch size (float4 accesses)
  float hreg[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const bool ok = s_begin + u < s_end;
    const int b = list[ok ? s_begin + u : 0];
    hreg[u] = (ok && j < a.H) ? ld<T>(h, static_cast<long long>(b) * a.h_ld + j) : 0.f;
  }
  for (int e = tid; e < n * 16; e += BWD_T)
    reinterpret_cast<float4*>(dls)[e] = reinterpret_cast<const float4*>(a.dl + static_cast<long long>(list[e >> 4]) * DL_LD)[e & 15];
  __syncthreads();
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.f;
  float dsum = 0.f;  // db2: atom kh * 32 + lane (lanes < 32), summed over the range
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    if (s_begin + u < s_end) {                            // wave-uniform
      const float* dlp = dls + (s_begin + u) * DL_LD + kh * 32;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 d = reinterpret_cast<const float4*>(dlp)[i];   // the same address in every lane: a broadcast read
        acc[4 * i] = fmaf(hreg[u], d.x, acc[4 * i]);
        acc[4 * i + 1] = fmaf(hreg[u], d.y, acc[4 * i + 1]);
        acc[4 * i + 2] = fmaf(hreg[u], d.z, acc[4 * i + 2]);
        acc[4 * i + 3] = fmaf(hreg[u], d.w, acc[4 * i + 3]);
      }
      dsum += lane < 32 ? dlp[lane & 31] : 0.f;
    }
  }
  constexpr int RLD = 20;                               // floats per (range, hidden unit) row: 16 atoms + pad (80 B)
  float* red = dls + static_cast<size_t>(a.B) * DL_LD;  // [8][64][RLD] behind the staged dl rows
  float* red2 = red + 8 * 64 * RLD;                     // [8][64]: db2 partials
  T* dw2 = static_cast<T*>(a.dw2);
#pragma unroll
  for (int p = 0; p < 4; ++p) {                         // 16 atoms per pass: 16 p .. 16 p + 15 (atom half p >> 1)
    if (kh == (p >> 1)) {
      float4* dst = reinterpret_cast<float4*>(red + (sq * 64 + lane) * RLD);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dst[i] = make_float4(acc[(p & 1) * 16 + 4 * i], acc[(p & 1) * 16 + 4 * i + 1], acc[(p & 1) * 16 + 4 * i + 2],
                             acc[(p & 1) * 16 + 4 * i + 3]);
    }
    __syncthreads();
    {
      const int i = tid & 15, jj = tid >> 4, k = 16 * p + i;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += red[(r * 64 + jj) * RLD + i];
      if (k < a.K && j0 + jj < a.H) st<T>(dw2, static_cast<long long>(j0 + jj) * a.dw2_ld + act_id * a.K + k, t);
    }
    __syncthreads();
  }
  if (j0 == 0) {
    if (lane < 32) red2[sq * 64 + kh * 32 + lane] = dsum;
    __syncthreads();
    if (tid < a.K) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += red2[r * 64 + tid];
      a.db2[act_id * a.K + tid] = t;
    }
  }
}

template <typename T>
int launch_backward(const BwdArgs& a0, hipStream_t s) {
  BwdArgs a = a0;
  const size_t row = static_cast<size_t>(a.A) * (sizeof(T) == 2 ? 72 : 68) * sizeof(T);
  const int n_jt = (a.H + 63) / 64;
  const size_t list_bytes = static_cast<size_t>((a.B + 3) & ~3) * 4;   // (rounded up: the staged rows behind it are read as float4)
  // hidden units per dH tile: 8 (64 workgroups for 512 units; bf16, 20 actions: 23 KB of LDS), fewer if LDS is short
  int jt = 8;
  while (jt > 1 && jt * row + 16 * jt * 4 > 64 * 1024) jt >>= 1;
  a.n_dh_tiles = (a.H + jt - 1) / jt;
  unsigned blocks = static_cast<unsigned>(a.n_dh_tiles + a.A * n_jt);
  if (const char* part = getenv("HB_BWD_PART")) {  // measurement aid: "1" = only the dH tiles, "2" = only the dW2 / db2 tiles
    if (part[0] == '1') blocks = static_cast<unsigned>(a.n_dh_tiles);
    if (part[0] == '2') { a.block_offset = a.n_dh_tiles; blocks = static_cast<unsigned>(a.A * n_jt); }
  }
  size_t lds = jt * row + 16 * jt * 4;
  if (lds < list_bytes) lds = list_bytes;
  // the dW2 / db2 half: sample list, the staged dl rows of up to B samples, the reduction buffers
  const size_t dw2_bytes = list_bytes + static_cast<size_t>(a.B) * DL_LD * 4 + (8 * 64 * 20 + 8 * 64) * sizeof(float);
  if (lds < dw2_bytes) lds = dw2_bytes;
  if (lds > 150 * 1024) return fail(HB_ERR_INVALID, "batch / output layer too large for the backward kernel's LDS tile");
#define HB_BWD(JT) hipLaunchKernelGGL((c51_backward_kernel<T, JT>), dim3(blocks), dim3(BWD_T), lds, s, a)
  if (jt == 8) HB_BWD(8);
  else if (jt == 4) HB_BWD(4);
  else if (jt == 2) HB_BWD(2);
  else HB_BWD(1);
#undef HB_BWD
  return HB_OK;
}

// ---- scalar double-Q loss of the older agent (hanabi_agents/rlax_dqn/rlax_dqn.py:170-205), compact gradient -------------
// One thread per sample, one workgroup (B <= 256). q values sit in "logits" rows of row stride rs with action a at column
// a * cs (cs = 2: the scalar head stored as a 2-atom head whose second atom is never used, so that hb_c51_backward serves it
// unchanged). td = r + gamma * q_target(s')[argmax_a q_online(s')] * (1 - terminal) - q_online(s)[a_tm1]; IS weight
// ((1/P) ** beta) / max; loss = mean(w * 0.5 * td^2)  =>  dLoss/dq[b, a_tm1] = -w * td / B, written to dl[b, 0]
// (dl rows are 64 floats; columns 1..63 are zeroed).
template <typename T>
__global__ __launch_bounds__(256) void dqn_loss_kernel(const T* __restrict__ q_on, const T* __restrict__ q_t, const int32_t* __restrict__ act,
                                                       const float* __restrict__ rew, const float* __restrict__ term,
                                                       const double* __restrict__ prios, const float* __restrict__ beta_dev,
                                                       const float* __restrict__ disc, int B, int A, int cs, int rs,
                                                       float* __restrict__ td_out, float* __restrict__ w_out, float* __restrict__ dl,
                                                       const T* __restrict__ bias_on, const T* __restrict__ bias_t) {
  __shared__ float red[8];
  const int b = threadIdx.x, lane = b & 63, wave = b >> 6;
  const bool on = b < B;
  const float beta = *beta_dev;
  float ip = on ? static_cast<float>(1.0 / prios[b]) : (beta >= 0.f ? 0.f : INFINITY);
  // max over the batch of (1/P) ** beta: attained at the largest (beta >= 0) or smallest 1/P (monotone map)
  float ext = ip;
  ext = beta >= 0.f ? wave_max(ext) : wave_min(ext);
  if (lane == 0) red[wave] = ext;
  __syncthreads();
  float e2 = red[0];
  for (int i = 1; i < (B + 63) / 64; ++i) e2 = beta >= 0.f ? fmaxf(e2, red[i]) : fminf(e2, red[i]);
  if (!on) return;
  const float w = powf(ip, beta) / powf(e2, beta);
  const int a_tm1 = act[b];
  const float q_tm1 = ld<T>(q_on, static_cast<long long>(b) * rs + a_tm1 * cs) + (bias_on ? ld<T>(bias_on, a_tm1 * cs) : 0.f);
  const T* sel = q_on + static_cast<long long>(B + b) * rs;     // online net on obs_t: the double-Q selector
  int best = 0;
  float bv = -INFINITY;
  for (int a = 0; a < A; ++a) {                                   // first maximum, like torch.argmax / jnp.argmax
    const float v = ld<T>(sel, a * cs) + (bias_on ? ld<T>(bias_on, a * cs) : 0.f);
    if (v > bv) { bv = v; best = a; }
  }
  float qt = ld<T>(q_t, static_cast<long long>(b) * rs + best * cs) + (bias_t ? ld<T>(bias_t, best * cs) : 0.f);
  if (term[b] != 0.f) qt = 0.f;
  const float td = rew[b] + disc[b] * qt - q_tm1;
  td_out[b] = td;
  w_out[b] = w;
  float4* row = reinterpret_cast<float4*>(dl + static_cast<long long>(b) * 64);
  row[0] = make_float4(-w * td / static_cast<float>(B), 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 1; i < 16; ++i) row[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- "thin" GEMM: the learner's forward products in a form that runs BESIDE an actor-GEMM workgroup --------------------
// While hb_actor_hidden / hb_actor_q run, every CU holds one of their workgroups (232 VGPRs x 2 wavefronts per SIMD, 133 KB
// of LDS): 48 VGPRs per SIMD and 27 KB of LDS are left, and a library GEMM launched by the other seat's update waits until
// they retire. This kernel fits in the remainder — ONE wavefront per workgroup, a 32 x 16 output tile, MFMA fragments loaded
// straight from global memory (no LDS), 36 VGPRs — so the update's forward pass proceeds WHILE the policy GEMMs run
// (scripts/thin_probe.py: it finishes inside a running hb_actor_q and lengthens that GEMM by 3 %).
//   out[b][m, n] = act(sum_k x[b][m, k] * wt[b][n, k] + bias[n]),  x and wt k-contiguous bf16, fp32 accumulation in k order
typedef __bf16 tg_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tg_f16x8 __attribute__((ext_vector_type(8)));
typedef float tg_f32x4 __attribute__((ext_vector_type(4)));
struct ThinArgs {
  const unsigned char* x;    // bf16 [batch][M][ldx]
  const unsigned char* wt;   // bf16 [batch][N][ldw]
  const __hip_bfloat16* bias;  // [batch][N] or NULL
  void* out;                 // [batch][M][ldo], bf16 or (relu & 2) fp32
  unsigned ldx, ldw, ldo;    // in elements
  unsigned x_bs, w_bs, o_bs; // batch strides in elements
  int k, relu;             // relu: bit 0 = ReLU, bit 1 = fp32 output (bit 2, fp16 operands, selects the kernel instantiation)
  unsigned n_tiles;        // first column tile (16 columns each) this launch covers
  unsigned nb;             // elements between the bias rows of consecutive batch entries (= n)
};
// F16: x, wt, bias and a 16-bit output are fp16 instead of bf16 (v_mfma_f32_16x16x32_f16: same rate)
template <bool F16>
__device__ __forceinline__ uint32_t tg_pack(float lo, float hi) {
  if constexpr (F16) {
    const _Float16 a = static_cast<_Float16>(lo), b = static_cast<_Float16>(hi);
    uint16_t ua, ub;
    __builtin_memcpy(&ua, &a, 2);
    __builtin_memcpy(&ub, &b, 2);
    return static_cast<uint32_t>(ua) | (static_cast<uint32_t>(ub) << 16);
  } else {
    const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
    return static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&a)) | (static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&b)) << 16);
  }
}
template <bool F16>
__device__ __forceinline__ float tg_bias(const __hip_bfloat16* b, unsigned i) {
  if constexpr (F16) return __half2float(reinterpret_cast<const __half*>(b)[i]);
  else return __bfloat162float(b[i]);
}
template <bool F16>
__device__ __forceinline__ tg_f32x4 tg_mfma(const uint4& wv, const uint4& xv, const tg_f32x4& c) {
  if constexpr (F16) {
    union { uint4 u; tg_f16x8 f; } w, x;
    w.u = wv; x.u = xv;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(w.f, x.f, c, 0, 0, 0);
  } else {
    union { uint4 u; tg_bf16x8 f; } w, x;
    w.u = wv; x.u = xv;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.f, x.f, c, 0, 0, 0);
  }
}
template <bool F16, bool PIPE>
__global__ __launch_bounds__(64) void thin_gemm_kernel(const ThinArgs a, const int batch) {
  // One wavefront per (row tile, column tile); it walks the batch itself, so a launch never has more workgroups than
  // (m / 32) * (n / 16): for the learner's shapes that is 1 024 = one wavefront per SIMD of the chip, which leaves an
  // actor-GEMM workgroup arriving LATER its 464 registers per SIMD as well.
  const int lane = threadIdx.x;
  const unsigned lr = lane & 15, kq = lane >> 4;
  const unsigned ct = blockIdx.y + a.n_tiles;   // (n_tiles: first column tile of this launch)
  const unsigned c = ct * 16u + 4u * kq;
  const unsigned half = 32u * a.ldx;   // byte distance of the second 16-row tile
#pragma unroll 1
  for (int z = 0; z < batch; ++z) {
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (a.bias) {
      const __hip_bfloat16* bz = a.bias + static_cast<size_t>(z) * a.nb;
      b0 = tg_bias<F16>(bz, c); b1 = tg_bias<F16>(bz, c + 1); b2 = tg_bias<F16>(bz, c + 2); b3 = tg_bias<F16>(bz, c + 3);
    }
    const unsigned xo = (z * a.x_bs + (blockIdx.x * 32u + lr) * a.ldx + kq * 8u) * 2u;
    const unsigned wo = (z * a.w_bs + (ct * 16u + lr) * a.ldw + kq * 8u) * 2u;
    tg_f32x4 acc0 = tg_f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    // PIPE (HB_THIN_PIPE=1, measurements only): the operands of K step s + 1 requested before the MFMAs of step s — one wavefront per
    // SIMD has nobody to hide its load latency behind. 12 more registers (48: the most that fits beside an actor workgroup).
    // Measured round 3, same box, alternating: the kernel gains 6 % (14.2 vs 15.1 us in the loop) and the STEP loses 15 % (0.128 vs
    // 0.112 ms): at 48 registers it takes everything an actor workgroup leaves, and the learner's other small kernels (36-47
    // registers) no longer share a SIMD with it. The plain loop (36 registers) is what runs.
    const unsigned kend = static_cast<unsigned>(a.k) * 2u;
    if constexpr (PIPE) {
      uint4 x0 = *reinterpret_cast<const uint4*>(a.x + xo), x1 = *reinterpret_cast<const uint4*>(a.x + xo + half);
      uint4 w0 = *reinterpret_cast<const uint4*>(a.wt + wo);
#pragma unroll 1
      for (unsigned kb = 64u; kb < kend; kb += 64u) {
        const uint4 nx0 = *reinterpret_cast<const uint4*>(a.x + xo + kb), nx1 = *reinterpret_cast<const uint4*>(a.x + xo + kb + half);
        const uint4 nw0 = *reinterpret_cast<const uint4*>(a.wt + wo + kb);
        asm volatile("" ::: "memory");   // (the loads are issued here, not sunk below the MFMAs)
        acc0 = tg_mfma<F16>(w0, x0, acc0);
        acc1 = tg_mfma<F16>(w0, x1, acc1);
        x0 = nx0; x1 = nx1; w0 = nw0;
      }
      acc0 = tg_mfma<F16>(w0, x0, acc0);
      acc1 = tg_mfma<F16>(w0, x1, acc1);
    } else {
#pragma unroll 1
      for (unsigned kb = 0; kb < kend; kb += 64u) {
        const uint4 x0 = *reinterpret_cast<const uint4*>(a.x + xo + kb), x1 = *reinterpret_cast<const uint4*>(a.x + xo + kb + half);
        const uint4 w0 = *reinterpret_cast<const uint4*>(a.wt + wo + kb);
        acc0 = tg_mfma<F16>(w0, x0, acc0);
        acc1 = tg_mfma<F16>(w0, x1, acc1);
      }
    }
    // acc_m[j] = out[row 32 bx + 16 m + (lane & 15)][col 16 by + 4 (lane >> 4) + j]
    float v[8] = {acc0[0] + b0, acc0[1] + b1, acc0[2] + b2, acc0[3] + b3, acc1[0] + b0, acc1[1] + b1, acc1[2] + b2, acc1[3] + b3};
    if (a.relu & 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    const size_t oo = z * static_cast<size_t>(a.o_bs) + static_cast<size_t>(blockIdx.x * 32u + lr) * a.ldo + c;
    if (a.relu & 2) {   // fp32 output: the accumulators as they are (the loss kernel's logits)
      float* o = static_cast<float*>(a.out) + oo;
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(o + 16u * static_cast<size_t>(a.ldo)) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      __hip_bfloat16* o = static_cast<__hip_bfloat16*>(a.out) + oo;
      *reinterpret_cast<uint2*>(o) = make_uint2(tg_pack<F16>(v[0], v[1]), tg_pack<F16>(v[2], v[3]));
      *reinterpret_cast<uint2*>(o + 16u * static_cast<size_t>(a.ldo)) = make_uint2(tg_pack<F16>(v[4], v[5]), tg_pack<F16>(v[6], v[7]));
    }
  }
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

int hb_thin_gemm(const void* x_dev, const void* wt_dev, const void* bias_dev, void* out_dev, int64_t m, int32_t n, int32_t k,
                 int32_t ldx, int32_t ldw, int32_t ldo, int32_t batch, int64_t x_batch_stride, int64_t w_batch_stride,
                 int64_t out_batch_stride, int32_t relu, void* stream) {
  if (!x_dev || !wt_dev || !out_dev) return fail(HB_ERR_INVALID, "null argument");
  if (m <= 0 || n <= 0 || batch <= 0) return HB_OK;
  if (m % 32 || n % 16 || k % 32 || k < 32) return fail(HB_ERR_INVALID, "need m % 32 == 0, n % 16 == 0, k % 32 == 0");
  if (ldx < k || ldw < k || ldo < n || ldx % 8 || ldw % 8 || ldo % 4) return fail(HB_ERR_INVALID, "bad leading dimensions");
  if (relu < 0 || relu > 7) return fail(HB_ERR_INVALID, "relu: bit 0 = ReLU, bit 1 = fp32 output, bit 2 = fp16 (not bf16) operands");
  if ((reinterpret_cast<uintptr_t>(x_dev) & 15u) || (reinterpret_cast<uintptr_t>(wt_dev) & 15u) ||
      (reinterpret_cast<uintptr_t>(out_dev) & ((relu & 2) ? 15u : 7u)))
    return fail(HB_ERR_ALIGN, "x / wt must be 16-byte aligned, out 8-byte (fp32 output: 16-byte) aligned");
  const int64_t xe = (batch - 1) * x_batch_stride + m * static_cast<int64_t>(ldx), we = (batch - 1) * w_batch_stride + n * static_cast<int64_t>(ldw);
  if (xe >= (1LL << 30) || we >= (1LL << 30)) return fail(HB_ERR_INVALID, "operands beyond the kernel's 32-bit offsets");
  ThinArgs a{static_cast<const unsigned char*>(x_dev), static_cast<const unsigned char*>(wt_dev),
             static_cast<const __hip_bfloat16*>(bias_dev), out_dev, static_cast<unsigned>(ldx),
             static_cast<unsigned>(ldw), static_cast<unsigned>(ldo), static_cast<unsigned>(x_batch_stride),
             static_cast<unsigned>(w_batch_stride), static_cast<unsigned>(out_batch_stride), k, relu, 0u, static_cast<unsigned>(n)};
  // at most 1 024 workgroups (= wavefronts) per launch: one per SIMD of the chip, so that an actor-GEMM workgroup arriving
  // while this kernel runs still finds its registers; wider outputs take several launches
  const bool pipe = [] { const char* e = getenv("HB_THIN_PIPE"); return e && e[0] == '1'; }();
  const unsigned tiles = static_cast<unsigned>(n / 16), gx = static_cast<unsigned>(m / 32), cap = gx >= 1024u ? 1u : 1024u / gx;
  for (unsigned t0 = 0; t0 < tiles; t0 += cap) {
    a.n_tiles = t0;
    const dim3 grid(gx, tiles - t0 < cap ? tiles - t0 : cap);
    if (pipe) {
      if (relu & 4) hipLaunchKernelGGL((thin_gemm_kernel<true, true>), grid, dim3(64), 0, static_cast<hipStream_t>(stream), a, batch);
      else hipLaunchKernelGGL((thin_gemm_kernel<false, true>), grid, dim3(64), 0, static_cast<hipStream_t>(stream), a, batch);
    } else {
      if (relu & 4) hipLaunchKernelGGL((thin_gemm_kernel<true, false>), grid, dim3(64), 0, static_cast<hipStream_t>(stream), a, batch);
      else hipLaunchKernelGGL((thin_gemm_kernel<false, false>), grid, dim3(64), 0, static_cast<hipStream_t>(stream), a, batch);
    }
  }
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_dqn_loss_sparse(const void* q_online_dev, const void* q_target_dev, int32_t dtype, const int32_t* act_dev, const float* rew_dev,
                       const float* term_dev, const double* prios_dev, const float* beta_dev, const float* disc_dev, int64_t batch,
                       int32_t n_actions, int32_t col_stride, int32_t row_stride, float* td_dev, float* w_dev, float* dl_dev,
                       const void* bias_online_dev, const void* bias_target_dev, void* stream) {
  if (!q_online_dev || !q_target_dev || !act_dev || !rew_dev || !term_dev || !prios_dev || !beta_dev || !disc_dev || !td_dev ||
      !w_dev || !dl_dev)
    return fail(HB_ERR_INVALID, "null argument");
  if (batch <= 0) return HB_OK;
  if (batch > 256) return fail(HB_ERR_INVALID, "batch must be <= 256");
  if (n_actions < 1 || col_stride < 1 || row_stride < (n_actions - 1) * col_stride + 1) return fail(HB_ERR_INVALID, "bad action layout");
  if (reinterpret_cast<uintptr_t>(dl_dev) & 15u) return fail(HB_ERR_ALIGN, "dl_dev must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int B = static_cast<int>(batch);
#define HB_DQNL(T)                                                                                                           \
  hipLaunchKernelGGL((dqn_loss_kernel<T>), dim3(1), dim3(256), 0, s, static_cast<const T*>(q_online_dev),                      \
                     static_cast<const T*>(q_target_dev), act_dev, rew_dev, term_dev, prios_dev, beta_dev, disc_dev, B, n_actions, \
                     col_stride, row_stride, td_dev, w_dev, dl_dev, static_cast<const T*>(bias_online_dev),                    \
                     static_cast<const T*>(bias_target_dev))
  if (dtype == 0) HB_DQNL(float);
  else if (dtype == 1) HB_DQNL(__hip_bfloat16);
  else if (dtype == 2) HB_DQNL(__half);
  else return fail(HB_ERR_INVALID, "dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
#undef HB_DQNL
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
  if (batch > 256) return fail(HB_ERR_INVALID, "batch must be <= 256 (a thread keeps its share of one action's samples in 32 registers)");
  const int AK = n_actions * n_atoms;
  if (hidden < 1 || hidden_ld < hidden || w2_ld < AK || dw2_ld < AK) return fail(HB_ERR_INVALID, "bad hidden size / row strides");
  if (reinterpret_cast<uintptr_t>(dl_dev) & 15u) return fail(HB_ERR_ALIGN, "dl_dev must be 16-byte aligned");
  BwdArgs a{dl_dev, act_dev, hidden_dev, w2_dev, dh_dev, db1_dev, dw2_dev, db2_dev, static_cast<int>(batch), hidden, n_actions,
            n_atoms, hidden_ld, w2_ld, dw2_ld, 0, 0};
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
