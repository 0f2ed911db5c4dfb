// actor.hip — the actor's forward pass as two hand-written MFMA kernels for gfx950 (bf16 in, fp32 accumulate).
//
// Replaces, for the C51 NoisyMLP with one hidden layer (hanabi_agents/rlax_dqn/rlax_rainbow.py:113-122,141-150 on top
// of noisy_mlp.py:176-185 with the merged weights W = w + w_mu + w_sigma * eps):
//
//   hb_actor_hidden   H = relu(obs @ W1 + b1)              the int8 observation rows are widened to bf16 while they
//                                                          are staged into LDS: no cast kernel, no bf16 copy of the
//                                                          observations in HBM
//   hb_actor_q        q[g, a] = mean_k softmax_k(H @ W2 + b2)[g, a, k] * atoms[k]
//                                                          the [N, A*K] logits (67 MB at 32 768 games) never reach
//                                                          HBM: a 256-column tile holds 5 whole actions (5 * 51 =
//                                                          255), so the C51 expectation is taken from the tile in LDS
//   hb_policy_select  legal mask, arg-max with uniform tie-breaking, epsilon-greedy (same draws as hb_policy_act)
//
// GEMM core (both kernels): 256 x 256 output tile per workgroup of 8 wavefronts (2 x 4; 128 x 64 per wavefront =
// 32 accumulator tiles of mfma_f32_16x16x32_bf16), K in steps of 64. Both operands are k-contiguous (the weights are
// kept transposed, hb_actor_pack_weights), staged global -> registers -> LDS with 16-byte accesses, double-buffered
// (one barrier per K step; the next step's global loads are in flight during the MFMAs). LDS rows are 128 B (64 bf16)
// with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7, which makes the fragment reads (ds_read_b128, 16 rows x
// one chunk per 16-lane group) conflict-free. The MFMA operands are swapped (weights as "A", activations as "B") so
// that a lane ends up with 4 consecutive OUTPUT COLUMNS of one row: the epilogue packs them into one 8-byte LDS write.
// Roofline: MFMA (bf16 dense 2.5 PFLOP/s). LDS fragment traffic is 0.375 KB per MFMA. Measured at 32 768 rows
// (scripts/actor_mfma_probe.py): hidden 33 us, q 43 us (GEMM 37 + C51 epilogue), select ~3 us = 85 us against 124 us for
// hb_obs_cast + hipBLASLt GEMMs + hb_policy_act; ~0.75 PFLOP/s executed. One barrier per K step keeps the MFMA pipe
// ~64 % busy (both wavefronts of a SIMD reach the barrier together); s_setprio around the MFMA block changed nothing, and
// a variant with 128-row tiles, K step 32 and two workgroups per CU (66.5 KB LDS each) measured slower (hidden 34.3 vs
// 33.2 us, q 47.0 vs 43.5 us); so did a two-group schedule (memory segment | barrier | MFMA segment | barrier, the two
// row halves one barrier apart: 39.3 / 46.1 us) — it gives up the ds_read / MFMA interleaving inside each wavefront.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdint>
#include <cstdlib>

#include "../../include/hanabi_hip.h"
#include "common.hpp"
#include "env_kernel.hpp"  // philox4x32_10

using hb::fail;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(1)));

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int ROWB = BK * 2;                      // bytes per LDS row (64 bf16)
constexpr int STAGE = (BM + BN) * ROWB;           // 64 KB: activations then weights
constexpr int OUT_LD = 520;                       // bytes per row of the bf16 output tile (2-way conflicts at most)
constexpr int LDS_BYTES = BM * OUT_LD > 2 * STAGE ? BM * OUT_LD : 2 * STAGE;  // 133 120

struct GemmArgs {
  const void* x;        // activations: int8 [M, x_ld] (MODE 0), bf16 [M, x_ld] (MODE 1), bit-packed rows of x_ld BYTES (MODE 2)
  long long m;
  int x_ld;             // row stride in ELEMENTS
  int k_real;           // valid K of the activations (columns >= k_real read as 0; MODE 2: the row's bit count, 32 * words);
                        // k_pad = multiple of 64 >= k_real
  int k_pad;
  const __hip_bfloat16* wt;  // weights transposed: [n_rows_total][k_pad]
  const float* bias;         // [n_rows_total]
  // MODE 0
  __hip_bfloat16* h;    // [M, h_ld]
  int h_ld;
  // MODE 1
  const float* support;
  float* q;             // [M, n_actions]
  int n_actions, n_atoms, group_actions;
  // MODE 1 with the action selection fused in (hb_actor_q_select): the LAST of a row tile's column-group workgroups to finish
  // (ticket counter per row tile) selects the actions of the tile's rows
  const int8_t* legal;          // [M, n_actions]; NULL: q values only
  int32_t* actions;             // [M]
  unsigned int* tickets;        // [row tiles], zero before the first launch; the selecting workgroup re-arms its counter
  float epsilon;
  unsigned long long seed, draw;
  long long first_gid;
};

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
  return static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&a)) |
         (static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}
// The output layer's logits tile is staged in LDS as fp16 (round 3; bf16 before): same 2 bytes, 8 x finer for |x| < 65 504, and the
// q values follow the fp32 logits 8 x more closely (tolerance.py). fp32 accumulators are clamped into the fp16 range first.
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
  const __half2 h = __floats2half2_rn(fminf(fmaxf(lo, -65000.f), 65000.f), fminf(fmaxf(hi, -65000.f), 65000.f));
  return *reinterpret_cast<const uint32_t*>(&h);
}
__device__ __forceinline__ float f16_lo(uint32_t d) { return __half2float(__ushort_as_half(static_cast<unsigned short>(d & 0xFFFFu))); }
__device__ __forceinline__ float f16_hi(uint32_t d) { return __half2float(__ushort_as_half(static_cast<unsigned short>(d >> 16))); }
// four unsigned bytes of `d` -> four bf16 (exact for 0..255): v_cvt_f32_ubyteN + v_perm_b32
__device__ __forceinline__ void widen4(uint32_t d, uint32_t& lo, uint32_t& hi) {
  const float f0 = static_cast<float>(d & 0xFFu), f1 = static_cast<float>((d >> 8) & 0xFFu);
  const float f2 = static_cast<float>((d >> 16) & 0xFFu), f3 = static_cast<float>(d >> 24);
  lo = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
  hi = __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u);
}

// C51 expectation of one (row, action): mean_k softmax(x)_k * support_k over KK fp16 logits that start LEAD
// elements past the 8-byte aligned LDS address p8 (the arithmetic of policy_kernel, policy.hip). The logits are
// fetched with 8-byte LDS reads (rows are 520 B apart: conflict-free per 32-lane half) instead of KK 2-byte reads,
// which made this epilogue LDS-issue-bound (22 us of a 60 us kernel).
template <int KK, int LEAD>
__device__ __forceinline__ float c51_expectation(const uint2* __restrict__ p8, const float* __restrict__ support) {
  constexpr int NQ = (LEAD + KK + 3) / 4;
  uint32_t d[2 * NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const uint2 v = p8[i];
    d[2 * i] = v.x;
    d[2 * i + 1] = v.y;
  }
  float x[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    constexpr int dummy = 0;
    (void)dummy;
    const int h = k + LEAD;
    x[k] = (h & 1) ? f16_hi(d[h >> 1]) : f16_lo(d[h >> 1]);
  }
  float mx = x[0];
#pragma unroll
  for (int k = 1; k < KK; ++k) mx = fmaxf(mx, x[k]);
  float s = 0.f, t = 0.f;
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    const float e = __expf(x[k] - mx);
    s += e;
    t += e * support[k];
  }
  return t / s / static_cast<float>(KK);
}

// eight observation bits -> eight bf16 0.0 / 1.0 (0x3F80): two bits per dword, spread to bits 0 and 16, times 0x3F80
__device__ __forceinline__ uint32_t bits2_bf16(uint32_t b, int p) {
  const uint32_t t = (b >> (2 * p)) & 3u;
  return __umul24((t | (t << 15)) & 0x00010001u, 0x3F80u);
}

__device__ __forceinline__ void store_q(const GemmArgs& a, long long idx, float v) {
  if (a.legal) __hip_atomic_store(a.q + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by another workgroup of this launch
  else a.q[idx] = v;
}

// The selection rule (policy_kernel, policy.hip) for `ng` consecutive games starting at g0, one lane per game, run by the
// first `cap` of `nthreads` threads of a workgroup: q and legal rows are fetched with coalesced loads into LDS (row stride
// A | 1: conflict-free), then ONE scan per game builds the legal mask, the maximum and its tie set together.
// sq: cap * (A | 1) floats followed by as many bytes. COHERENT: q was written by OTHER workgroups of the running kernel:
// read it with device-scope loads.
template <int AT, bool COHERENT>
__device__ __forceinline__ void select_rows(const float* __restrict__ q, const int8_t* __restrict__ legal, long long g0, int ng, int A_rt,
                                            float epsilon, unsigned long long seed, unsigned long long draw, long long first_gid,
                                            int32_t* __restrict__ actions, float* sq, int cap, int tid, int nthreads) {
  const int A = AT > 0 ? AT : A_rt;
  const int ld = A | 1;
  int8_t* sl = reinterpret_cast<int8_t*>(sq + cap * ld);
  // the Philox draw does not depend on the loads: issue it while they are in flight
  const long long g = g0 + tid;
  const unsigned long long gid = static_cast<unsigned long long>(first_gid + g);
  uint32_t r[4];
  hb::philox4x32_10(static_cast<uint32_t>(draw), static_cast<uint32_t>(draw >> 32), static_cast<uint32_t>(gid),
                    static_cast<uint32_t>(gid >> 32), static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), r);
  for (int e = tid; e < ng * A; e += nthreads) {
    const int gg = e / A, i = e - gg * A;
    sq[gg * ld + i] = COHERENT ? __hip_atomic_load(q + g0 * A + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[g0 * A + e];
    sl[gg * ld + i] = legal[g0 * A + e];
  }
  __syncthreads();
  if (tid >= ng) return;
  const float* qr = sq + tid * ld;
  const int8_t* lr = sl + tid * ld;
  float best = -INFINITY;
  unsigned long long legal_mask = 0, ties = 0;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    const float v = qr[i];
    if (lr[i] != 0) {
      legal_mask |= 1ull << i;
      if (v > best) { best = v; ties = 1ull << i; }
      else if (v == best) ties |= 1ull << i;
    }
  }
  const float u = static_cast<float>(r[0] >> 8) * (1.0f / 16777216.0f);
  unsigned long long pool = (u < epsilon) ? legal_mask : ties;
  if (pool == 0) pool = legal_mask;
  int pick = 0;
  const int c = __popcll(pool);
  if (c > 0) {
    int k = static_cast<int>(__umulhi(r[1], static_cast<uint32_t>(c)));
    while (k-- > 0) pool &= pool - 1;
    pick = __ffsll(static_cast<long long>(pool)) - 1;
  }
  actions[g] = pick;
}

// MODE 0: hidden layer from int8 observations; MODE 1: output layer + C51 expectation; MODE 2: hidden layer from bit-packed
// observations (8x fewer bytes per row; a lane's 8 k-values are ONE byte of the row)
template <int MODE>
__global__ __launch_bounds__(NT) void actor_gemm_kernel(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  // Workgroup -> (row tile, column tile). Workgroups are dealt round-robin over the 8 XCDs in launch order (speed only, never
  // correctness): linear id L runs on the XCD labelled L % 8, in the order L / 8. All column tiles of one row tile are given
  // consecutive slots of ONE label, so that the row tile's activations are fetched into that XCD's L2 once and the other
  // column tiles hit there (hb_actor_q: 4 column groups re-read the same 256 x 512 hidden activations).
  int rt = static_cast<int>(blockIdx.x), ct = static_cast<int>(blockIdx.y);
  if ((gridDim.x & 7u) == 0) {
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, label = lin & 7u, slot = lin >> 3;
    ct = static_cast<int>(slot % gridDim.y);
    rt = static_cast<int>((slot / gridDim.y) * 8u + label);
  }
  const long long row0 = static_cast<long long>(rt) * BM;
  const int col0 = ct * BN;
  const int kt_n = a.k_pad / BK;

  // ---- staging: thread owns 16-byte chunk (r, c) of rows r = (tid >> 3) + 64 i, i = 0..3, for both operands
  const int sc = tid & 7, sr = tid >> 3;
  uint4 xr0, xr1, xr2, xr3, wq0, wq1, wq2, wq3;
#define HB_LOAD_ONE(i, XR, WQ)                                                                                      \
  {                                                                                                                 \
    const int r = sr + 64 * (i);                                                                                    \
    long long row = row0 + r;                                                                                       \
    if (row >= a.m) row = a.m - 1; /* clamp: rows past the end are computed and dropped */                          \
    const int k = kt_ * BK + sc * 8;                                                                                \
    if (MODE == 0) {                                                                                                \
      const uint8_t* p = static_cast<const uint8_t*>(a.x) + row * a.x_ld + k;                                       \
      uint32_t d0 = 0, d1 = 0;                                                                                      \
      if (k + 8 <= a.k_real) {                                                                                      \
        const u32x2_u v = *reinterpret_cast<const u32x2_u*>(p); /* byte-aligned 8-byte load */                      \
        d0 = v.x;                                                                                                   \
        d1 = v.y;                                                                                                   \
      } else {                                                                                                      \
        for (int b = 0; b < 8; ++b)                                                                                 \
          if (k + b < a.k_real) {                                                                                   \
            if (b < 4) d0 |= static_cast<uint32_t>(p[b]) << (8 * b);                                                \
            else d1 |= static_cast<uint32_t>(p[b]) << (8 * (b - 4));                                                \
          }                                                                                                         \
      }                                                                                                             \
      XR = make_uint4(d0, d1, 0u, 0u); /* widened when it is written to LDS, after the MFMAs of this step */         \
    } else if (MODE == 2) {                                                                                         \
      const uint8_t* p = static_cast<const uint8_t*>(a.x) + row * a.x_ld + (k >> 3);                                \
      XR = make_uint4(k < a.k_real ? static_cast<uint32_t>(*p) : 0u, 0u, 0u, 0u);                                    \
    } else {                                                                                                        \
      XR = *reinterpret_cast<const uint4*>(static_cast<const __hip_bfloat16*>(a.x) + row * a.x_ld + k);            \
    }                                                                                                               \
    WQ = *reinterpret_cast<const uint4*>(a.wt + static_cast<long long>(col0 + r) * a.k_pad + k);                    \
  }
#define HB_LOAD_STAGE(KT)          \
  {                                \
    const int kt_ = (KT);          \
    HB_LOAD_ONE(0, xr0, wq0)       \
    HB_LOAD_ONE(1, xr1, wq1)       \
    HB_LOAD_ONE(2, xr2, wq2)       \
    HB_LOAD_ONE(3, xr3, wq3)       \
  }
#define HB_STORE_ONE(i, XR, WQ)                                                  \
  {                                                                              \
    const int r = sr + 64 * (i);                                                 \
    const int off = r * ROWB + ((sc ^ ((r >> 1) & 7)) << 4);                     \
    if (MODE == 0) {                                                             \
      uint32_t e0, e1, e2, e3;                                                   \
      widen4(XR.x, e0, e1);                                                      \
      widen4(XR.y, e2, e3);                                                      \
      *reinterpret_cast<uint4*>(base_ + off) = make_uint4(e0, e1, e2, e3);       \
    } else if (MODE == 2) {                                                      \
      *reinterpret_cast<uint4*>(base_ + off) =                                   \
          make_uint4(bits2_bf16(XR.x, 0), bits2_bf16(XR.x, 1), bits2_bf16(XR.x, 2), bits2_bf16(XR.x, 3)); \
    } else {                                                                     \
      *reinterpret_cast<uint4*>(base_ + off) = XR;                               \
    }                                                                            \
    *reinterpret_cast<uint4*>(base_ + BM * ROWB + off) = WQ;                     \
  }
#define HB_STORE_STAGE(BUF)                        \
  {                                                \
    unsigned char* base_ = lds + (BUF) * STAGE;    \
    HB_STORE_ONE(0, xr0, wq0)                      \
    HB_STORE_ONE(1, xr1, wq1)                      \
    HB_STORE_ONE(2, xr2, wq2)                      \
    HB_STORE_ONE(3, xr3, wq3)                      \
  }

  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: 16-row tile t of an operand region starting at row base: row = base + 16 t + (lane & 15),
  // chunk = 4 kk + (lane >> 4); (row >> 1) & 7 == (lane & 15) >> 1 because base and 16 t are multiples of 16
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = BM * ROWB + (wc * 64 + (lane & 15)) * ROWB;

  // Pipeline: while step kt is multiplied out of LDS buffer kt & 1, step kt+1 (loaded into registers one iteration
  // earlier) is written to the other buffer and the global loads of step kt+2 are issued. One barrier per step: it
  // both publishes buffer (kt+1) & 1 and retires the reads of buffer kt & 1 before that one is overwritten.
  HB_LOAD_STAGE(0)
  HB_STORE_STAGE(0)
  if (kt_n > 1) HB_LOAD_STAGE(1)
  __syncthreads();
  for (int kt = 0; kt < kt_n; ++kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    if (kt + 1 < kt_n) HB_STORE_STAGE((kt + 1) & 1)
    if (kt + 2 < kt_n) HB_LOAD_STAGE(kt + 2)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = ((kk * 4 + fq) ^ fsw) << 4;
      bf16x8 wf[4], xf[8];
#pragma unroll
      for (int n = 0; n < 4; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch);
#pragma unroll
      for (int m = 0; m < 8; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[n][m], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: acc[n][m][j] = out[row wr*128 + 16 m + (lane & 15)][col wc*64 + 16 n + 4 (lane >> 4) + j]
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int cl = wc * 64 + n * 16 + fq * 4;
    const float4 b = *reinterpret_cast<const float4*>(a.bias + col0 + cl);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float v0 = acc[n][m][0] + b.x, v1 = acc[n][m][1] + b.y, v2 = acc[n][m][2] + b.z, v3 = acc[n][m][3] + b.w;
      if (MODE != 1) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
      }
      const int rl = wr * 128 + m * 16 + (lane & 15);
      *reinterpret_cast<uint2*>(lds + rl * OUT_LD + cl * 2) =
          MODE == 1 ? make_uint2(pack_f16(v0, v1), pack_f16(v2, v3)) : make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
    }
  }
#undef HB_LOAD_ONE
#undef HB_LOAD_STAGE
#undef HB_STORE_ONE
#undef HB_STORE_STAGE
  __syncthreads();
  if (MODE != 1) {
    // coalesced copy-out: 32 chunks of 16 bytes per row
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int id = tid + NT * i, r = id >> 5, ch = id & 31;
      if (row0 + r < a.m)
        *reinterpret_cast<uint4*>(a.h + (row0 + r) * a.h_ld + col0 + ch * 8) = *reinterpret_cast<const uint4*>(lds + r * OUT_LD + ch * 16);
    }
  } else {
    // C51 expectation per (row, action) from the fp16 logits tile (the arithmetic of policy_kernel, policy.hip)
    const int K = a.n_atoms;
    const int first_action = ct * a.group_actions;
    int ga = a.n_actions - first_action;
    if (ga > a.group_actions) ga = a.group_actions;
    for (int id = tid; id < BM * ga; id += NT) {
      const int r = id & (BM - 1), al = id >> 8;
      const uint16_t* p = reinterpret_cast<const uint16_t*>(lds + r * OUT_LD) + al * K;
      float qv = 0.f;
#ifndef HB_ACTOR_SKIP_C51  // (diagnostic builds time the GEMM alone)
      if (K == 51) {  // the reference's atom count (params.py:18): unrolled, wide LDS reads; `al` is wave-uniform
        const int start = al * 51 * 2, lead = (start & 7) >> 1;
        const uint2* p8 = reinterpret_cast<const uint2*>(lds + r * OUT_LD + (start & ~7));
        if (lead == 0) qv = c51_expectation<51, 0>(p8, a.support);
        else if (lead == 1) qv = c51_expectation<51, 1>(p8, a.support);
        else if (lead == 2) qv = c51_expectation<51, 2>(p8, a.support);
        else qv = c51_expectation<51, 3>(p8, a.support);
        if (row0 + r < a.m) store_q(a, (row0 + r) * a.n_actions + first_action + al, qv);
        continue;
      }
      float mx = -INFINITY;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, f16_lo(p[k]));
      float s = 0.f, t = 0.f;
      for (int k = 0; k < K; ++k) {
        const float e = __expf(f16_lo(p[k]) - mx);
        s += e;
        t += e * a.support[k];
      }
      qv = t / s / static_cast<float>(K);
#endif
      if (row0 + r < a.m) store_q(a, (row0 + r) * a.n_actions + first_action + al, qv);
    }
    if (a.legal) {
      // Action selection fused in: every workgroup publishes its q values and takes a ticket of its row tile; the one that
      // draws the last ticket sees all n_actions values of the tile's rows and selects their actions with the rule of
      // policy_select_kernel. One launch and one trip of the q values through a kernel boundary less on the acting stream's
      // critical path. Coherence across XCDs (each has its own L2) comes from the accesses themselves: q is written with
      // device-scope stores (write-through) and read back with device-scope loads, the ticket is a device-scope RMW; the
      // ordering needs only "my stores have completed" (s_waitcnt vmcnt(0): the workgroup-scope release fence) before the
      // barrier that precedes the RMW. Device-scope FENCES are deliberately absent: each is an L2 write-back + invalidate,
      // and 4 096 of them per launch made the kernel 4x slower (measured: 174 us for the policy call instead of 75).
      __shared__ int s_last;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.tickets + rt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old + 1u == gridDim.y;
        if (last) __hip_atomic_store(a.tickets + rt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
        s_last = last;
      }
      __syncthreads();
      if (s_last) {   // (workgroup-uniform)
        const int ng = a.m - row0 < BM ? static_cast<int>(a.m - row0) : BM;
        float* sq = reinterpret_cast<float*>(lds);
        if (a.n_actions == 20) select_rows<20, true>(a.q, a.legal, row0, ng, 20, a.epsilon, a.seed, a.draw, a.first_gid, a.actions, sq, BM, tid, NT);
        else if (a.n_actions == 48) select_rows<48, true>(a.q, a.legal, row0, ng, 48, a.epsilon, a.seed, a.draw, a.first_gid, a.actions, sq, BM, tid, NT);
        else select_rows<0, true>(a.q, a.legal, row0, ng, a.n_actions, a.epsilon, a.seed, a.draw, a.first_gid, a.actions, sq, BM, tid, NT);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Second form of the same GEMM (same tile, same fragment reads, same MFMA order: bit-identical results), built so that the
// learner's kernels can run BESIDE it on the same CUs (the self-play loop overlaps one seat's update with the other seat's
// policy forward; with 232 VGPRs x 2 waves per SIMD and 133 KB of LDS the first form leaves room for nothing else, and a
// learner kernel queued behind 256-512 of its workgroups waits 30-50 us for a CU):
//   * weights (and, for the output layer, the hidden activations) go global -> LDS directly (global_load_lds, 16 B per
//     lane, XOR swizzle applied on the SOURCE address): no staging registers, no ds_write pass;
//   * bit-packed observations: a thread fetches ONE dword (32 observation bits of one row) per K step, two steps ahead, and
//     expands it into the next stage's activation tile while the current step multiplies;
//   * the epilogue reuses the main loop's 128 KB of LDS, half a tile at a time.
typedef const __attribute__((address_space(1))) void* hb_gptr_t;
typedef __attribute__((address_space(3))) void* hb_lptr_t;
__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((hb_gptr_t)g, (hb_lptr_t)l, 16, 0, 0);  // LDS address: wave-uniform base + lane * 16
}
constexpr int LDS2_BYTES = 2 * STAGE;  // 131 072

template <int MODE>  // 1: output layer + C51 expectation (x = bf16 hidden activations); 2: hidden layer from bit-packed observations
__global__ __launch_bounds__(NT) void actor_gemm2_kernel(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS2_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  int rt = static_cast<int>(blockIdx.x), ct = static_cast<int>(blockIdx.y);
  if ((gridDim.x & 7u) == 0) {  // XCD-aware tile order (see actor_gemm_kernel)
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, label = lin & 7u, slot = lin >> 3;
    ct = static_cast<int>(slot % gridDim.y);
    rt = static_cast<int>((slot / gridDim.y) * 8u + label);
  }
  const long long row0 = static_cast<long long>(rt) * BM;
  const int col0 = ct * BN;
  const int kt_n = a.k_pad / BK;

  // ---- direct-to-LDS staging: one wave instruction writes 1 KiB = 8 rows x 128 B; lane l owns (row l >> 3, physical chunk
  // l & 7) and fetches the LOGICAL chunk that the swizzled read expects there
  const int lr = lane >> 3, pc = lane & 7;
  const unsigned char* wsrc[4];
  const unsigned char* xsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 32 * wave + 8 * i + lr;
    const int c = pc ^ ((r >> 1) & 7);
    wsrc[i] = reinterpret_cast<const unsigned char*>(a.wt + static_cast<long long>(col0 + r) * a.k_pad) + c * 16;
    if (MODE == 1) {
      long long row = row0 + r;
      if (row >= a.m) row = a.m - 1;
      xsrc[i] = reinterpret_cast<const unsigned char*>(static_cast<const __hip_bfloat16*>(a.x) + row * a.x_ld) + c * 16;
    }
  }
  auto stage = [&](int kt, unsigned char* base) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned char* dst = base + (32 * wave + 8 * i) * ROWB;
      if (MODE == 1) glds16(xsrc[i] + kt * (BK * 2), dst);
      glds16(wsrc[i] + kt * (BK * 2), dst + BM * ROWB);
    }
  };
  // ---- bit-packed observations: thread -> (row tid >> 1, half tid & 1): the 32 bits k = 64 kt + 32 half .. + 31 of its row
  const int br = tid >> 1, bh = tid & 1;
  const uint32_t* bsrc = nullptr;
  if (MODE == 2) {
    long long row = row0 + br;
    if (row >= a.m) row = a.m - 1;
    bsrc = reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(a.x) + row * a.x_ld) + bh;
  }
  const int n_words = a.k_real >> 5;  // MODE 2: words per row
  auto load_bits = [&](int kt) -> uint32_t { return (2 * kt + bh < n_words) ? bsrc[2 * kt] : 0u; };
  auto expand_bits = [&](uint32_t b, unsigned char* base) {
    unsigned char* rowp = base + br * ROWB;
    const int sw = (br >> 1) & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t v = b >> (8 * j);
      *reinterpret_cast<uint4*>(rowp + (((4 * bh + j) ^ sw) << 4)) =
          make_uint4(bits2_bf16(v, 0), bits2_bf16(v, 1), bits2_bf16(v, 2), bits2_bf16(v, 3));
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = BM * ROWB + (wc * 64 + (lane & 15)) * ROWB;

  uint32_t bits_next = 0;
  if (MODE == 2) {
    expand_bits(load_bits(0), lds);
    if (kt_n > 1) bits_next = load_bits(1);
  }
  stage(0, lds);
  for (int kt = 0; kt < kt_n; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // stage kt complete for every wave (LDS-DMA landed, expansions written); the other stage is free
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    unsigned char* nxt = lds + ((kt + 1) & 1) * STAGE;
    uint32_t bits_after = 0;
    if (MODE == 2 && kt + 2 < kt_n) bits_after = load_bits(kt + 2);   // (issued BEFORE the LDS-DMAs: an older load)
    if (kt + 1 < kt_n) stage(kt + 1, nxt);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = ((kk * 4 + fq) ^ fsw) << 4;
      bf16x8 wf[4], xf[8];
#pragma unroll
      for (int n = 0; n < 4; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch);
#pragma unroll
      for (int m = 0; m < 8; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch);
      if (MODE == 2 && kk == 0 && kt + 1 < kt_n) expand_bits(bits_next, nxt);  // VALU + 4 LDS writes beside the MFMAs
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[n][m], 0, 0, 0);
    }
    bits_next = bits_after;
  }
  __syncthreads();

  // ---- epilogue, 128 rows at a time through the (now idle) staging LDS: pass p takes the wavefronts with wr == p
  for (int p = 0; p < 2; ++p) {
    if (wr == p) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int cl = wc * 64 + n * 16 + fq * 4;
        const float4 b = *reinterpret_cast<const float4*>(a.bias + col0 + cl);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          float v0 = acc[n][m][0] + b.x, v1 = acc[n][m][1] + b.y, v2 = acc[n][m][2] + b.z, v3 = acc[n][m][3] + b.w;
          if (MODE != 1) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
          }
          const int rl = m * 16 + (lane & 15);
          *reinterpret_cast<uint2*>(lds + rl * OUT_LD + cl * 2) =
              MODE == 1 ? make_uint2(pack_f16(v0, v1), pack_f16(v2, v3)) : make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
        }
      }
    }
    __syncthreads();
    const long long rbase = row0 + p * 128;
    if (MODE != 1) {
#pragma unroll 4
      for (int i = 0; i < 8; ++i) {  // coalesced copy-out: 32 chunks of 16 bytes per row
        const int id = tid + NT * i, r = id >> 5, ch = id & 31;
        if (rbase + r < a.m)
          *reinterpret_cast<uint4*>(a.h + (rbase + r) * a.h_ld + col0 + ch * 8) = *reinterpret_cast<const uint4*>(lds + r * OUT_LD + ch * 16);
      }
    } else {
      const int K = a.n_atoms;
      const int first_action = ct * a.group_actions;
      int ga = a.n_actions - first_action;
      if (ga > a.group_actions) ga = a.group_actions;
      for (int id = tid; id < 128 * ga; id += NT) {
        const int r = id & 127, al = id >> 7;
        float qv;
        if (K == 51) {  // `al` is wave-uniform
          const int start = al * 51 * 2, lead = (start & 7) >> 1;
          const uint2* p8 = reinterpret_cast<const uint2*>(lds + r * OUT_LD + (start & ~7));
          if (lead == 0) qv = c51_expectation<51, 0>(p8, a.support);
          else if (lead == 1) qv = c51_expectation<51, 1>(p8, a.support);
          else if (lead == 2) qv = c51_expectation<51, 2>(p8, a.support);
          else qv = c51_expectation<51, 3>(p8, a.support);
        } else {
          const uint16_t* pp = reinterpret_cast<const uint16_t*>(lds + r * OUT_LD) + al * K;
          float mx = -INFINITY;
          for (int k = 0; k < K; ++k) mx = fmaxf(mx, f16_lo(pp[k]));
          float s = 0.f, t = 0.f;
          for (int k = 0; k < K; ++k) {
            const float e = __expf(f16_lo(pp[k]) - mx);
            s += e;
            t += e * a.support[k];
          }
          qv = t / s / static_cast<float>(K);
        }
        if (rbase + r < a.m) a.q[(rbase + r) * a.n_actions + first_action + al] = qv;
      }
    }
    __syncthreads();
  }
}

// weights [K][N] (row stride w_ld) -> transposed [n'][k_pad] with n' = (n / group_cols) * 256 + n % group_cols
// (group_cols = 0: n' = n); bias -> fp32 at the same n'. 32 x 32 tiles through LDS; up to 4 matrices per launch
// (workgroup b belongs to job j where first[j] <= b < first[j + 1]). Rows / columns that no input maps to are left
// untouched (the caller zero-initialises the outputs once).
struct PackJobs {
  hb_pack_job j[4];
  int first[5];
  int count;
};
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackJobs jobs) {
  __shared__ __hip_bfloat16 t[32][33];
  int ji = 0;
  while (ji + 1 < jobs.count && static_cast<int>(blockIdx.x) >= jobs.first[ji + 1]) ++ji;
  const hb_pack_job& job = jobs.j[ji];
  const __hip_bfloat16* w = static_cast<const __hip_bfloat16*>(job.w);
  const __hip_bfloat16* bias = static_cast<const __hip_bfloat16*>(job.bias);
  __hip_bfloat16* wt = static_cast<__hip_bfloat16*>(job.wt);
  const int tiles_n = (job.n_cols + 31) / 32;
  const int b = static_cast<int>(blockIdx.x) - jobs.first[ji];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int k0 = (b / tiles_n) * 32, n0 = (b % tiles_n) * 32;
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, n = n0 + tx;
    t[i][tx] = (k < job.k_rows && n < job.n_cols) ? w[static_cast<long long>(k) * job.w_ld + n] : __float2bfloat16(0.f);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, k = k0 + tx;
    if (n < job.n_cols && k < job.k_pad) {
      const int np = job.group_cols ? (n / job.group_cols) * 256 + n % job.group_cols : n;
      wt[static_cast<long long>(np) * job.k_pad + k] = t[tx][i];
    }
  }
  if (k0 == 0 && ty == 0) {
    const int n = n0 + tx;
    if (n < job.n_cols) {
      const int np = job.group_cols ? (n / job.group_cols) * 256 + n % job.group_cols : n;
      job.bias_out[np] = __bfloat162float(bias[n]);
    }
  }
}

// lane per game: the selection rule of policy_kernel (policy.hip) on precomputed q values. A workgroup's SEL_T games are
// contiguous in q and legal: fetched with coalesced loads into LDS (row stride A | 1: conflict-free), then ONE scan per
// game builds the legal mask, the maximum and its tie set together. 128 games per workgroup: 256 workgroups at 32 768 games
// (the 256-game version left half of the CUs idle and ran two dependent scans: 7.8 us alone, 11.6 us inside the loop).
// (select_rows above is the body; hb_actor_q_select runs the same body inside the output-layer GEMM's last workgroups.)
constexpr int SEL_T = 128;
template <int AT>  // AT > 0: compile-time action count (unrolled scan); 0: run-time A
__global__ __launch_bounds__(SEL_T) void policy_select_kernel(const float* __restrict__ q, const int8_t* __restrict__ legal, long long n,
                                                              int A_rt, float epsilon, unsigned long long seed, unsigned long long draw,
                                                              long long first_gid, int32_t* __restrict__ actions) {
  extern __shared__ float sq[];                                    // SEL_T * (A | 1) floats, then as many bytes
  const int A = AT > 0 ? AT : A_rt;
  const long long g0 = static_cast<long long>(blockIdx.x) * SEL_T;
  const int ng = n - g0 < SEL_T ? static_cast<int>(n - g0) : SEL_T;
  select_rows<AT, false>(q, legal, g0, ng, A, epsilon, seed, draw, first_gid, actions, sq, SEL_T, static_cast<int>(threadIdx.x), SEL_T);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// The first (register-staged) GEMM form is the default: measured 28.6 / 40.1 us (hidden / q at 32 768 rows) against 31.6 / 41.9 us
// for the direct-to-LDS form, which HB_ACTOR_FORM=2 selects for A/B measurements; results are bit-identical
bool use_first_form() {
  static const int v = [] { const char* e = getenv("HB_ACTOR_FORM"); return e && e[0] == '2' ? 0 : 1; }();
  return v != 0;
}
}  // namespace

extern "C" {

int hb_actor_pack_weights(const hb_pack_job* jobs, int32_t count, void* stream) {
  if (!jobs) return fail(HB_ERR_INVALID, "null argument");
  if (count < 1 || count > 4) return fail(HB_ERR_INVALID, "count must be 1..4");
  PackJobs p{};
  p.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const hb_pack_job& j = jobs[i];
    if (!j.w || !j.bias || !j.wt || !j.bias_out) return fail(HB_ERR_INVALID, "null pointer in job %d", i);
    if (j.k_rows < 1 || j.n_cols < 1 || j.w_ld < j.n_cols || j.k_pad < j.k_rows || j.k_pad % 64)
      return fail(HB_ERR_INVALID, "job %d: need w_ld >= n_cols, k_pad >= k_rows and k_pad a multiple of 64", i);
    if (j.group_cols < 0 || j.group_cols > 256) return fail(HB_ERR_INVALID, "job %d: group_cols must be 0..256", i);
    p.j[i] = j;
    p.first[i] = blocks;
    blocks += ((j.n_cols + 31) / 32) * ((j.k_pad + 31) / 32);
  }
  p.first[count] = blocks;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_actor_hidden(const int8_t* obs_dev, int64_t n_rows, int32_t obs_len, const void* w1t_dev, int32_t k_pad, const float* b1_dev,
                    int32_t hidden, void* h_dev, void* stream) {
  if (!obs_dev || !w1t_dev || !b1_dev || !h_dev) return fail(HB_ERR_INVALID, "null argument");
  if (n_rows <= 0) return HB_OK;
  if (obs_len < 1 || k_pad < obs_len || k_pad % BK) return fail(HB_ERR_INVALID, "k_pad must be a multiple of 64 and >= obs_len");
  if (hidden < BN || hidden % BN) return fail(HB_ERR_INVALID, "hidden must be a multiple of 256");
  if (!aligned16(w1t_dev) || !aligned16(b1_dev) || !aligned16(h_dev)) return fail(HB_ERR_ALIGN, "w1t / b1 / h must be 16-byte aligned");
  GemmArgs a{};
  a.x = obs_dev; a.m = n_rows; a.x_ld = obs_len; a.k_real = obs_len; a.k_pad = k_pad;
  a.wt = static_cast<const __hip_bfloat16*>(w1t_dev); a.bias = b1_dev;
  a.h = static_cast<__hip_bfloat16*>(h_dev); a.h_ld = hidden;
  const dim3 grid(static_cast<unsigned>((n_rows + BM - 1) / BM), static_cast<unsigned>(hidden / BN));
  hipLaunchKernelGGL((actor_gemm_kernel<0>), grid, dim3(NT), 0, static_cast<hipStream_t>(stream), a);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_actor_hidden_packed(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1t_dev, int32_t k_pad,
                           const float* b1_dev, int32_t hidden, void* h_dev, void* stream) {
  if (!obs_bits_dev || !w1t_dev || !b1_dev || !h_dev) return fail(HB_ERR_INVALID, "null argument");
  if (n_rows <= 0) return HB_OK;
  if (obs_len < 1 || k_pad < obs_len || k_pad % BK) return fail(HB_ERR_INVALID, "k_pad must be a multiple of 64 and >= obs_len");
  if (hidden < BN || hidden % BN) return fail(HB_ERR_INVALID, "hidden must be a multiple of 256");
  if (!aligned16(w1t_dev) || !aligned16(b1_dev) || !aligned16(h_dev)) return fail(HB_ERR_ALIGN, "w1t / b1 / h must be 16-byte aligned");
  const int words = (obs_len + 31) / 32;
  GemmArgs a{};
  a.x = obs_bits_dev; a.m = n_rows; a.x_ld = words * 4; a.k_real = words * 32; a.k_pad = k_pad;
  a.wt = static_cast<const __hip_bfloat16*>(w1t_dev); a.bias = b1_dev;
  a.h = static_cast<__hip_bfloat16*>(h_dev); a.h_ld = hidden;
  const dim3 grid(static_cast<unsigned>((n_rows + BM - 1) / BM), static_cast<unsigned>(hidden / BN));
  if (use_first_form()) hipLaunchKernelGGL((actor_gemm_kernel<2>), grid, dim3(NT), 0, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL((actor_gemm2_kernel<2>), grid, dim3(NT), 0, static_cast<hipStream_t>(stream), a);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

static int launch_q(const void* h_dev, int64_t n_rows, int32_t hidden, const void* w2t_dev, const float* b2_dev, const float* support_dev,
                    int32_t n_actions, int32_t n_atoms, float* q_dev, const int8_t* legal_dev, float epsilon, uint64_t seed, uint64_t draw,
                    int64_t first_game_id, int32_t* actions_dev, uint32_t* tickets_dev, void* stream) {
  if (!h_dev || !w2t_dev || !b2_dev || !support_dev || !q_dev) return fail(HB_ERR_INVALID, "null argument");
  if (n_rows <= 0) return HB_OK;
  if (hidden < BK || hidden % BK) return fail(HB_ERR_INVALID, "hidden must be a multiple of 64");
  if (n_actions < 1 || n_atoms < 2 || n_atoms > 256) return fail(HB_ERR_INVALID, "need n_actions >= 1 and 2 <= n_atoms <= 256");
  if (!aligned16(h_dev) || !aligned16(w2t_dev) || !aligned16(b2_dev)) return fail(HB_ERR_ALIGN, "h / w2t / b2 must be 16-byte aligned");
  GemmArgs a{};
  a.x = h_dev; a.m = n_rows; a.x_ld = hidden; a.k_real = hidden; a.k_pad = hidden;
  a.wt = static_cast<const __hip_bfloat16*>(w2t_dev); a.bias = b2_dev;
  a.support = support_dev; a.q = q_dev; a.n_actions = n_actions; a.n_atoms = n_atoms;
  a.group_actions = 256 / n_atoms;
  const int groups = (n_actions + a.group_actions - 1) / a.group_actions;
  const dim3 grid(static_cast<unsigned>((n_rows + BM - 1) / BM), static_cast<unsigned>(groups));
  if (legal_dev) {   // selection fused into the GEMM's last workgroups (first form only)
    if (!actions_dev || !tickets_dev) return fail(HB_ERR_INVALID, "null argument");
    if (n_actions > 64) return fail(HB_ERR_INVALID, "n_actions must be 1..64");
    a.legal = legal_dev; a.actions = actions_dev; a.tickets = tickets_dev; a.epsilon = epsilon;
    a.seed = seed; a.draw = draw; a.first_gid = first_game_id;
  }
  if (use_first_form()) hipLaunchKernelGGL((actor_gemm_kernel<1>), grid, dim3(NT), 0, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL((actor_gemm2_kernel<1>), grid, dim3(NT), 0, static_cast<hipStream_t>(stream), a);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_actor_q(const void* h_dev, int64_t n_rows, int32_t hidden, const void* w2t_dev, const float* b2_dev, const float* support_dev,
               int32_t n_actions, int32_t n_atoms, float* q_dev, void* stream) {
  return launch_q(h_dev, n_rows, hidden, w2t_dev, b2_dev, support_dev, n_actions, n_atoms, q_dev, nullptr, 0.f, 0, 0, 0, nullptr, nullptr,
                  stream);
}

int hb_policy_select(const float* q_dev, const int8_t* legal_dev, int64_t n_games, int32_t n_actions, float epsilon, uint64_t seed,
                     uint64_t draw, int64_t first_game_id, int32_t* actions_dev, void* stream) {
  if (!q_dev || !legal_dev || !actions_dev) return fail(HB_ERR_INVALID, "null argument");
  if (n_actions < 1 || n_actions > 64) return fail(HB_ERR_INVALID, "n_actions must be 1..64");
  if (n_games <= 0) return HB_OK;
  const dim3 grid(static_cast<unsigned>((n_games + SEL_T - 1) / SEL_T)), block(SEL_T);
  const size_t lds = static_cast<size_t>(SEL_T) * (n_actions | 1) * 5;
#define HB_SELECT(AT)                                                                                                        \
  hipLaunchKernelGGL((policy_select_kernel<AT>), grid, block, lds, static_cast<hipStream_t>(stream), q_dev, legal_dev,         \
                     static_cast<long long>(n_games), n_actions, epsilon, static_cast<unsigned long long>(seed),               \
                     static_cast<unsigned long long>(draw), static_cast<long long>(first_game_id), actions_dev)
  if (n_actions == 20) HB_SELECT(20);        // 2-player full Hanabi
  else if (n_actions == 48) HB_SELECT(48);   // 5-player full Hanabi
  else HB_SELECT(0);
#undef HB_SELECT
  HB_HIP(hipGetLastError());
  return HB_OK;
}


// One policy call behind ONE entry point: a host binding pays its per-call cost (ctypes: ~5 us) once instead of three times.
// obs_is_packed != 0: obs_dev holds bit rows (hb_actor_hidden_packed), else int8 rows. tickets_dev != NULL (one zero-initialised
// uint32 per 256-row tile, i.e. ceil(n_rows / 256)): the selection runs inside the output-layer GEMM (two launches); NULL:
// hb_policy_select as a third launch. Same actions either way.
int hb_actor_q_select(const void* h_dev, int64_t n_rows, int32_t hidden, const void* w2t_dev, const float* b2_dev,
                      const float* support_dev, int32_t n_actions, int32_t n_atoms, float* q_dev, const int8_t* legal_dev, float epsilon,
                      uint64_t seed, uint64_t draw, int64_t first_game_id, int32_t* actions_dev, uint32_t* tickets_dev, void* stream) {
  if (!legal_dev || !actions_dev) return fail(HB_ERR_INVALID, "null argument");
  if (tickets_dev && use_first_form())
    return launch_q(h_dev, n_rows, hidden, w2t_dev, b2_dev, support_dev, n_actions, n_atoms, q_dev, legal_dev, epsilon, seed, draw,
                    first_game_id, actions_dev, tickets_dev, stream);
  const int rc = hb_actor_q(h_dev, n_rows, hidden, w2t_dev, b2_dev, support_dev, n_actions, n_atoms, q_dev, stream);
  if (rc != HB_OK) return rc;
  return hb_policy_select(q_dev, legal_dev, n_rows, n_actions, epsilon, seed, draw, first_game_id, actions_dev, stream);
}

int hb_actor_act(const void* obs_dev, int32_t obs_is_packed, const int8_t* legal_dev, int64_t n_rows, int32_t obs_len, const void* w1t_dev,
                 int32_t k_pad, const float* b1_dev, int32_t hidden, void* h_dev, const void* w2t_dev, const float* b2_dev,
                 const float* support_dev, int32_t n_actions, int32_t n_atoms, float* q_dev, float epsilon, uint64_t seed, uint64_t draw,
                 int64_t first_game_id, int32_t* actions_dev, uint32_t* tickets_dev, void* stream) {
  const int rc = obs_is_packed ? hb_actor_hidden_packed(static_cast<const uint32_t*>(obs_dev), n_rows, obs_len, w1t_dev, k_pad, b1_dev,
                                                        hidden, h_dev, stream)
                               : hb_actor_hidden(static_cast<const int8_t*>(obs_dev), n_rows, obs_len, w1t_dev, k_pad, b1_dev, hidden,
                                                 h_dev, stream);
  if (rc != HB_OK) return rc;
  return hb_actor_q_select(h_dev, n_rows, hidden, w2t_dev, b2_dev, support_dev, n_actions, n_atoms, q_dev, legal_dev, epsilon, seed, draw,
                           first_game_id, actions_dev, tickets_dev, stream);
}

}  // extern "C"
