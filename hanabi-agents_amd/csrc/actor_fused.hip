// actor_fused.hip — the actor's whole forward pass as ONE kernel for gfx950: bit-packed observations -> q values, with the
// hidden activations never leaving the CU (round 3; replaces the hb_actor_hidden_packed + hb_actor_q pair of actor.hip on the
// benched path: no 33 MB write + re-read of H, no store tail, no second prologue, no launch gap).
//
//   q[g, a] = mean_k softmax_k(relu(obs[g] @ W1 + b1) @ W2 + b2)[a, k] * support[k]
//   (hanabi_agents/rlax_dqn/rlax_rainbow.py:113-122 on top of noisy_mlp.py:176-185 with the merged W = w + w_mu + w_sigma * eps)
//
// One workgroup of 8 wavefronts owns 128 rows (32 768 games -> 256 workgroups, one per CU). Per wavefront, per layer, the
// output tile is 128 rows x 64 columns (32 accumulator tiles of v_mfma_f32_16x16x32_bf16, 128 accumulator registers):
//   * WEIGHTS never touch LDS: they are kept in HBM/L2 in "fragment-major" order (hb_actor_fused_pack), so that the A operand
//     of one MFMA for all 64 lanes is one contiguous 1 KiB global_load_dwordx4; every wavefront owns different columns, so no
//     weight byte is fetched twice by a CU; loads run one K step (32) ahead of their use.
//   * layer 1 ACTIVATIONS are observation BITS: the lane's 8 k-values of one row are ONE byte of the bit row, and the B operand
//     (8 bf16 of 0.0 / 1.0) is ONE ds_read_b128 from a 256-entry table replicated per 16-byte bank slot (64 KiB, conflict-free:
//     every lane of a 16-lane read group uses its own slot). No staging passes, no barriers in the K loop.
//   * H = relu(.) is written to LDS once as bf16 (128 KiB, XOR-swizzled rows of 1 KiB: conflict-free ds_read_b128) — it overlays
//     the table, which is dead by then — and is the B operand of layer 2.
//   * layer 2 runs in passes of 512 physical columns (64 per wavefront). The C51 expectation is taken FROM THE ACCUMULATORS in
//     fp32 (the logits are never rounded: actor.hip rounded them to bf16 in LDS). The physical column order is chosen by the
//     packer so that the registers of a lane are action-uniform: per (wavefront, pass) "slot", 51 of the 64 columns are one
//     whole action (12 registers x 4 lane groups + 3 lanes of a 13th register) and the other 13 are one QUARTER of an "extra"
//     action whose four quarters sit in four neighbouring wavefronts of the same pass; quarters are combined through (max, sum,
//     weighted sum) triples in LDS after the pass's only barrier. Capacity 10 actions per pass: Hanabi-Full 2 players (20 actions)
//     = 2 passes = 1 024 physical columns for 1 020 logits.
// Roofline: MFMA (bf16 dense 2.5 PFLOP/s); executed 58.0 GFLOP at 32 768 x 658 -> 512 -> 1 020.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "../../include/hanabi_hip.h"
#include "common.hpp"
#include "env_kernel.hpp"  // philox4x32_10

using hb::fail;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FM = 128;                 // rows per workgroup
constexpr int FH = 512;                 // hidden units (the reference's layers=[512], params.py:13)
constexpr int FNT = 512;                // threads
constexpr int FK = 51;                  // atoms (params.py:18)
constexpr int S2 = FH / 32;             // K steps of layer 2
constexpr int LUT_BYTES = 65536;        // 256 byte values x 16 slots x 16 B
constexpr int H_BYTES = FM * FH * 2;    // 131 072
constexpr int FRAG_FLOATS = 3 * 8 * FM; // one buffer of quarter-action triples: [max | sum | weighted sum][wave][row] fp32 (12 KiB)
// 143 360 B = 112 of the CU's 128 LDS allocation granules (1 280 B): 20 KiB stay free for the learner's small kernels, which run beside
// this kernel on another stream (tree fill: 8 KiB; sample + gather; transposer). With two triple buffers (155 648 B) the tree
// fill's 8 192 B did NOT fit (7 680 B were left after rounding) and waited for this kernel to retire.
constexpr int LDS_TOTAL = H_BYTES + FRAG_FLOATS * 4;
constexpr float NEG_BIG = -1e30f;

struct FusedArgs {
  const uint32_t* obs;     // [m][words]
  long long m;
  int words;               // u32 per bit row
  int s1;                  // K steps (of 32) of layer 1: k_pad / 32, even
  const uint4* w1f;        // [s1][32 n-tiles][64 lanes] 16 B
  const float* b1;         // [512]
  const uint4* w2f;        // [n_pass][16][32 n-tiles][64 lanes] 16 B
  const float* b2;         // [512 * n_pass] physical column order, NEG_BIG on unused columns
  const float* support;    // [51]
  float* q;                // [m][n_actions]
  int n_actions, n_pass;
  // hb_actor_fused_act: the epsilon-greedy selection of hb_policy_select (csrc/actor.hip, select_rows: same rule, same Philox
  // draws, identical actions) as the workgroup's last act, on the q rows it has just written
  const int8_t* legal;     // [m][n_actions], or NULL: q values only
  int32_t* actions;        // [m]
  float epsilon;
  unsigned long long seed, draw;
  long long first_gid;
  unsigned long long* stamps;   // diagnostic builds only (-DHB_STAMPS): 16 u64 per wavefront
};

// In-kernel phase stamps (cdna_hip_programming.md section 7): compiled in only with -DHB_STAMPS, into the separate diagnostic
// library (make stamps); the shipped kernel contains none of this.
#ifdef HB_STAMPS
#define HB_FSTAMP(slot)                                                                              \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    unsigned long long hb_t_;                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t_)::"memory");                    \
    if (a.stamps && lane == 0) a.stamps[(static_cast<long long>(blockIdx.x) * 8 + wave) * 16 + (slot)] = hb_t_; \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)
#define HB_FSTAMP_REAL(slot)                                                                         \
  do {                                                                                               \
    unsigned long long hb_t_;                                                                        \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t_)::"memory");                \
    if (a.stamps && lane == 0) a.stamps[(static_cast<long long>(blockIdx.x) * 8 + wave) * 16 + (slot)] = hb_t_; \
  } while (0)
#else
#define HB_FSTAMP(slot) do {} while (0)
#define HB_FSTAMP_REAL(slot) do {} while (0)
#endif

// The kernel exists for two operand types of the same MFMA rate: bf16 (F16 = false; v_mfma_f32_16x16x32_bf16) and fp16 (F16 = true;
// v_mfma_f32_16x16x32_f16 — the reference's own network dtype, rlax_rainbow.py:250-251). Fragments travel as raw 16 bytes; only
// the table of 0.0 / 1.0, the rounding of H and the MFMA itself know the type.
template <bool F16>
__device__ __forceinline__ uint32_t pack16(float lo, float hi) {
  if constexpr (F16) {
    const _Float16 a = static_cast<_Float16>(lo), b = static_cast<_Float16>(hi);   // round to nearest even
    uint16_t ua, ub;
    __builtin_memcpy(&ua, &a, 2);
    __builtin_memcpy(&ub, &b, 2);
    return static_cast<uint32_t>(ua) | (static_cast<uint32_t>(ub) << 16);
  } else {
    const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
    return static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&a)) |
           (static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&b)) << 16);
  }
}
// two observation bits -> two 16-bit floats 0.0 / 1.0 (bf16 0x3F80, fp16 0x3C00)
template <bool F16>
__device__ __forceinline__ uint32_t bits2_one(uint32_t b, int p) {
  const uint32_t t = (b >> (2 * p)) & 3u;
  return __umul24((t | (t << 15)) & 0x00010001u, F16 ? 0x3C00u : 0x3F80u);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4& wv, const uint4& xv, const f32x4& c) {
  if constexpr (F16) {
    union { uint4 u; f16x8 f; } w, x;
    w.u = wv; x.u = xv;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(w.f, x.f, c, 0, 0, 0);
  } else {
    union { uint4 u; bf16x8 f; } w, x;
    w.u = wv; x.u = xv;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.f, x.f, c, 0, 0, 0);
  }
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// Reductions over the four 16-lane groups of a wavefront (lanes l, l ^ 16, l ^ 32, l ^ 48) use gfx950's row swaps, VALU only:
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second, v_permlane32_swap the upper
// half of the first with the lower half of the second; called with the same value twice they return {x[row & ~1], x[row | 1]}
// and {x[lower], x[upper]} (HB_SWAP16 / HB_SWAP32 in the epilogue). Every lane ends with the full result.
// v_max_f32 / v_max3_f32 as they are: fmaxf() makes hipcc quiet possible signalling NaNs first (one v_max x, x, x per operand)
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

template <bool F16>
__global__ __launch_bounds__(FNT) void actor_fused_kernel(const FusedArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_TOTAL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: everything derived from it lives in scalar registers)
  const int r = lane & 15, qd = lane >> 4;
  const long long row0 = static_cast<long long>(blockIdx.x) * FM;
  const int s1 = a.s1, G = (s1 + 3) >> 2;
  // Table and bit rows sit at the TOP of the LDS (over the quarter-triple buffers and the last rows of H), so that after layer 1
  // a wavefront can write the first m_early row tiles of H while other wavefronts are still reading them.
  unsigned char* lutb = lds + (LDS_TOTAL - LUT_BYTES);
  const int bits_off = LDS_TOTAL - LUT_BYTES - 2048 * G;
  unsigned char* bits = lds + bits_off;   // [8 m][G][64 lanes] u32: byte i of the word = byte (4 (4 g + i) + q) of row 16 m + r
  const int m_early = bits_off >> 14;     // row tiles (16 rows x 1 KiB) below the bit rows

  HB_FSTAMP_REAL(14);
  HB_FSTAMP(0);
  // ---- first weights in flight before anything else
  const uint4* w1p = a.w1f + (4 * wave) * 64 + lane;   // step s: + s * 2048; n-tile n of this wavefront: + n * 64
  uint4 wa[4], wb[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) wa[n] = w1p[n * 64];

  // ---- observation bits, byte-transposed so that a lane finds the bytes of four consecutive K steps in one dword; the first
  // item's loads (HBM latency) are in flight while the table is built
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  auto load_item = [&](int it) {
    const int rr = it & (FM - 1), g = it >> 7;
    long long row = row0 + rr;
    if (row >= a.m) row = a.m - 1;   // rows past the end are computed and dropped
    const uint32_t* src = a.obs + row * a.words + 4 * g;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (4 * g + i < a.words) ? src[i] : 0u;
  };
  if (tid < FM * G) load_item(tid);
  // ---- table: entry (v, slot) at v * 256 + slot * 16
  for (int e = tid; e < 4096; e += FNT) {
    const uint32_t v = static_cast<uint32_t>(e) >> 4;
    *reinterpret_cast<uint4*>(lutb + e * 16) = make_uint4(bits2_one<F16>(v, 0), bits2_one<F16>(v, 1), bits2_one<F16>(v, 2), bits2_one<F16>(v, 3));
  }
  for (int it = tid; it < FM * G; it += FNT) {
    const int rr = it & (FM - 1), g = it >> 7;
    if (it != tid) load_item(it);
    uint32_t* dst = reinterpret_cast<uint32_t*>(bits) + (static_cast<size_t>((rr >> 4) * G + g) * 64 + (rr & 15));
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const uint32_t o = ((w[0] >> (8 * qq)) & 0xFFu) | (((w[1] >> (8 * qq)) & 0xFFu) << 8) | (((w[2] >> (8 * qq)) & 0xFFu) << 16) |
                         (((w[3] >> (8 * qq)) & 0xFFu) << 24);
      dst[qq * 16] = o;
    }
  }
  HB_FSTAMP(1);
  __syncthreads();
  HB_FSTAMP(2);

  f32x4 acc[4][8];   // accumulators start at the bias of their 4 columns
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float4 b = *reinterpret_cast<const float4*>(a.b1 + 64 * wave + 16 * n + 4 * qd);
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{b.x, b.y, b.z, b.w};
  }

  // =========================== layer 1: H[128 rows][64 wave + 0..63] over K = 32 s1
  {
    const unsigned char* lut = lutb + r * 16;
    const uint32_t* bw = reinterpret_cast<const uint32_t*>(bits) + lane;
#define HB_L1_STEP(I, WCUR, WNXT, HAVE_NEXT)                                                                   \
  {                                                                                                            \
    const int s_ = 4 * g + (I);                                                                                \
    if (HAVE_NEXT) {                                                                                           \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) WNXT[n] = w1p[(s_ + 1) * 2048 + n * 64];                    \
    }                                                                                                          \
    asm volatile("" ::: "memory"); /* the loads are ISSUED here, a whole step before their use */               \
    uint4 xf[8];                                                                                               \
    _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                                            \
      const uint32_t v = (xw[m] >> (8 * (I))) & 0xFFu;                                                         \
      xf[m] = *reinterpret_cast<const uint4*>(lut + v * 256u);                                                 \
    }                                                                                                          \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                            \
      _Pragma("unroll") for (int m = 0; m < 8; ++m)                                                            \
        acc[n][m] = mfma16<F16>(WCUR[n], xf[m], acc[n][m]);                                                    \
    }                                                                                                          \
  }
    for (int g = 0; g < G; ++g) {
      uint32_t xw[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) xw[m] = bw[(m * G + g) * 64];
      HB_L1_STEP(0, wa, wb, true)                       // (s1 is even: step 4 g + 1 exists)
      const bool more = 4 * g + 2 < s1;
      HB_L1_STEP(1, wb, wa, more)
      if (more) {
        HB_L1_STEP(2, wa, wb, true)
        HB_L1_STEP(3, wb, wa, (4 * g + 4 < s1))
      }
    }
#undef HB_L1_STEP
  }
  HB_FSTAMP(3);

  // first weights of layer 2 in flight across the barrier and the H write
  const uint4* w2p = a.w2f + (4 * wave) * 64 + lane;   // pass p: + p * S2 * 2048; step s: + s * 2048
#pragma unroll
  for (int n = 0; n < 4; ++n) wa[n] = w2p[n * 64];

  // ---- H = relu(acc) as bf16: acc[n][m][j] = row 16 m + r, unit 64 wave + 16 n + 4 q + j; 8-byte LDS writes. Row tiles below the
  // bit rows first; the others once every wavefront has finished reading the table and the bit rows.
#define HB_H_WRITE(COND)                                                                                        \
  _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                                               \
    if (COND) {                                                                                                 \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                           \
        const int chunk = 8 * wave + 2 * n + (qd >> 1);                                                         \
        const float v0 = fmaxf(acc[n][m][0], 0.f), v1 = fmaxf(acc[n][m][1], 0.f);                               \
        const float v2 = fmaxf(acc[n][m][2], 0.f), v3 = fmaxf(acc[n][m][3], 0.f);                               \
        *reinterpret_cast<uint2*>(lds + (16 * m + r) * 1024 + ((chunk ^ r) << 4) + (qd & 1) * 8) =              \
            make_uint2(pack16<F16>(v0, v1), pack16<F16>(v2, v3));                                               \
      }                                                                                                         \
    }                                                                                                           \
  }
  HB_H_WRITE(m < m_early)
  HB_FSTAMP(4);
  __syncthreads();
  HB_H_WRITE(m >= m_early)
#undef HB_H_WRITE

  HB_FSTAMP(5);
  __syncthreads();
  HB_FSTAMP(6);

  // =========================== layer 2 in passes of 512 physical columns + the C51 expectation from the accumulators
  const int n_pass = a.n_pass, A = a.n_actions;
  const int full_cap = 8 * n_pass;
  constexpr float LOG2E = 1.44269504088896340736f;
  const unsigned char* hrow = lds + r * 1024;

  for (int p = 0; p < n_pass; ++p) {
    const int slot = 8 * p + wave;
    {   // accumulators start at the bias (physical column order; NEG_BIG on unused columns, whose weights are zero)
      const float* bp = a.b2 + static_cast<long long>(slot) * 64 + 4 * qd;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float4 b = *reinterpret_cast<const float4*>(bp + 16 * n);
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{b.x, b.y, b.z, b.w};
      }
    }
    const uint4* wp = w2p + static_cast<long long>(p) * S2 * 2048;
// The B operands (rows of H) are read four row tiles at a time and the second four only once the first four's MFMAs have been
// issued (sched_barrier): 16 fragment registers instead of 32. The LDS latency this exposes is covered by the SIMD's other
// wavefront; what it buys is a kernel of 232 registers, i.e. 48 free per SIMD lane beside its two wavefronts: the learner's
// forward kernels (36-47 registers) then run on the same CUs while this kernel holds them (DESIGN: co-residency).
#define HB_L2_HALF(S, WCUR, M0)                                                                                \
  {                                                                                                            \
    uint4 xf[4];                                                                                               \
    const int ch = ((4 * (S) + qd) ^ r) << 4;                                                                  \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) xf[m] = *reinterpret_cast<const uint4*>(hrow + ((M0) + m) * 16384 + ch); \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                            \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                            \
        acc[n][(M0) + m] = mfma16<F16>(WCUR[n], xf[m], acc[n][(M0) + m]);                                      \
    }                                                                                                          \
  }
#define HB_L2_STEP(S, WCUR, WNXT, HAVE_NEXT)                                                                   \
  {                                                                                                            \
    if (HAVE_NEXT) {                                                                                           \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) WNXT[n] = wp[((S) + 1) * 2048 + n * 64];                    \
    }                                                                                                          \
    asm volatile("" ::: "memory");                                                                             \
    HB_L2_HALF(S, WCUR, 0)                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    HB_L2_HALF(S, WCUR, 4)                                                                                     \
  }
#pragma unroll 1
    for (int s = 0; s < S2; s += 2) {
      HB_L2_STEP(s, wa, wb, true)
      HB_L2_STEP(s + 1, wb, wa, (s + 2 < S2))
    }
#undef HB_L2_STEP
#undef HB_L2_HALF
    if (p < 2) HB_FSTAMP(7 + 3 * p);
    // next pass's first weights in flight during the epilogue
    if (p + 1 < n_pass) {
#pragma unroll
      for (int n = 0; n < 4; ++n) wa[n] = wp[S2 * 2048 + n * 64];
    }

    // ---- epilogue of slot (p, wave): register (n, j) of lane group q is physical column 16 n + 4 q + j of the slot
    // (loaded here, per pass, rather than kept across the K loop: 18 registers the loop does not have to carry)
    // per-lane support values, in the register pairs the packed sums use. Whole action: register (n, j), n < 3, holds atom
    // 4 (4 n + j) + q; register (3, 0) holds atom 48 + q in lanes q < 3. Quarter f: registers (3, 1..3) hold atom 13 f + 4 (j - 1) + q,
    // register (3, 0) of lanes q == 3 atom 13 f + 12.
    auto sup_at = [&](int k) { return a.support[k < FK ? k : FK - 1]; };
    f32x2 supF[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) supF[i] = f32x2{sup_at(4 * (2 * i) + qd), sup_at(4 * (2 * i + 1) + qd)};
    const float supF12 = sup_at(48 + qd);
    const int fq = wave & 3;   // which quarter of its extra action this wavefront holds
    const f32x2 supZa = f32x2{sup_at(13 * fq + qd), sup_at(13 * fq + 4 + qd)};
    const f32x2 supZb = f32x2{sup_at(13 * fq + 8 + qd), sup_at(13 * fq + 12)};
    const bool full_ok = slot < A && slot < full_cap;
    float* fb = reinterpret_cast<float*>(lds + H_BYTES) + wave * FM;   // component c: + c * 8 * FM
    // Written STAGE BY STAGE over the 8 row tiles (m is the inner loop everywhere): a wavefront issues in order, and one row
    // tile's softmax is a single dependent chain (max -> swaps -> exp -> sums -> swaps), so consecutive instructions must come
    // from different row tiles to keep the vector pipe busy (row tile after row tile: 9.6 k cycles per pass; see DESIGN).
    const f32x2 L2 = f32x2{LOG2E, LOG2E};
    // (two halves of four row tiles: the temporaries of all eight at once cost 16 registers more than the kernel may use if the
    //  learner's small kernels are to run beside it: 232 registers x 2 wavefronts leave 48 per SIMD lane)
    hb::static_for<2>([&](auto MH) {
    constexpr int M0 = 4 * decltype(MH)::value;
    float qF[4], zm[4], zs[4], zt[4], vf12[4], vz3[4], nmx[4], nmz[4];
#define HB_M for (int m = 0; m < 4; ++m)
#define HB_SWAP16(X, OP) { const u32x2 t_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(X), __float_as_uint(X), false, false); \
                           X = OP(__uint_as_float(t_.x), __uint_as_float(t_.y)); }
#define HB_SWAP32(X, OP) { const u32x2 t_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(X), __float_as_uint(X), false, false); \
                           X = OP(__uint_as_float(t_.x), __uint_as_float(t_.y)); }
#define HB_ADD(A_, B_) ((A_) + (B_))
    // register (3, 0) is shared: lanes q < 3 -> atoms 48..50 of the whole action, lanes q == 3 -> the quarter's 13th position
#pragma unroll
    HB_M { vf12[m] = qd < 3 ? acc[3][M0 + m][0] : NEG_BIG; vz3[m] = qd == 3 ? acc[3][M0 + m][0] : NEG_BIG; }
#pragma unroll
    HB_M nmx[m] = vmax3(acc[0][M0 + m][0], acc[0][M0 + m][1], acc[0][M0 + m][2]);
#pragma unroll
    HB_M nmx[m] = vmax3(nmx[m], acc[0][M0 + m][3], acc[1][M0 + m][0]);
#pragma unroll
    HB_M nmx[m] = vmax3(nmx[m], acc[1][M0 + m][1], acc[1][M0 + m][2]);
#pragma unroll
    HB_M nmx[m] = vmax3(nmx[m], acc[1][M0 + m][3], acc[2][M0 + m][0]);
#pragma unroll
    HB_M nmx[m] = vmax3(nmx[m], acc[2][M0 + m][1], acc[2][M0 + m][2]);
#pragma unroll
    HB_M nmx[m] = vmax3(nmx[m], acc[2][M0 + m][3], vf12[m]);
#pragma unroll
    HB_M nmz[m] = vmax(vmax3(acc[3][M0 + m][1], acc[3][M0 + m][2], acc[3][M0 + m][3]), vz3[m]);
#pragma unroll
    HB_M HB_SWAP16(nmx[m], vmax)
#pragma unroll
    HB_M HB_SWAP16(nmz[m], vmax)
#pragma unroll
    HB_M HB_SWAP32(nmx[m], vmax)
#pragma unroll
    HB_M HB_SWAP32(nmz[m], vmax)
#pragma unroll
    HB_M { nmx[m] *= -LOG2E; nmz[m] *= -LOG2E; }
    f32x2 s2[4], t2[4];
#define HB_PAIR_FIRST(N, LO, SUP)                                                                   \
  _Pragma("unroll") HB_M {                                                                          \
    const f32x2 arg = f32x2{acc[N][M0 + m][LO], acc[N][M0 + m][LO + 1]} * L2 + f32x2{nmx[m], nmx[m]};         \
    const f32x2 e = f32x2{exp2_fast(arg.x), exp2_fast(arg.y)};                                      \
    s2[m] = e;                                                                                      \
    t2[m] = e * SUP;                                                                                \
  }
#define HB_PAIR(N, LO, SUP)                                                                         \
  _Pragma("unroll") HB_M {                                                                          \
    const f32x2 arg = f32x2{acc[N][M0 + m][LO], acc[N][M0 + m][LO + 1]} * L2 + f32x2{nmx[m], nmx[m]};         \
    const f32x2 e = f32x2{exp2_fast(arg.x), exp2_fast(arg.y)};                                      \
    s2[m] += e;                                                                                     \
    t2[m] += e * SUP;                                                                               \
  }
    HB_PAIR_FIRST(0, 0, supF[0])
    HB_PAIR(0, 2, supF[1])
    HB_PAIR(1, 0, supF[2])
    HB_PAIR(1, 2, supF[3])
    HB_PAIR(2, 0, supF[4])
    HB_PAIR(2, 2, supF[5])
#undef HB_PAIR
#undef HB_PAIR_FIRST
    float sF[4], tF[4];
#pragma unroll
    HB_M {
      const float e12 = exp2_fast(vf12[m] * LOG2E + nmx[m]);
      sF[m] = s2[m].x + s2[m].y + e12;
      tF[m] = t2[m].x + t2[m].y + e12 * supF12;
    }
    // the quarter: registers (3, 1), (3, 2) | (3, 3), shared register
#pragma unroll
    HB_M {
      const f32x2 arg = f32x2{acc[3][M0 + m][1], acc[3][M0 + m][2]} * L2 + f32x2{nmz[m], nmz[m]};
      const f32x2 e = f32x2{exp2_fast(arg.x), exp2_fast(arg.y)};
      s2[m] = e;
      t2[m] = e * supZa;
    }
#pragma unroll
    HB_M {
      const f32x2 arg = f32x2{acc[3][M0 + m][3], vz3[m]} * L2 + f32x2{nmz[m], nmz[m]};
      const f32x2 e = f32x2{exp2_fast(arg.x), exp2_fast(arg.y)};
      s2[m] += e;
      t2[m] += e * supZb;
    }
#pragma unroll
    HB_M { zs[m] = s2[m].x + s2[m].y; zt[m] = t2[m].x + t2[m].y; }
#pragma unroll
    HB_M HB_SWAP16(sF[m], HB_ADD)
#pragma unroll
    HB_M HB_SWAP16(tF[m], HB_ADD)
#pragma unroll
    HB_M HB_SWAP16(zs[m], HB_ADD)
#pragma unroll
    HB_M HB_SWAP16(zt[m], HB_ADD)
#pragma unroll
    HB_M HB_SWAP32(sF[m], HB_ADD)
#pragma unroll
    HB_M HB_SWAP32(tF[m], HB_ADD)
#pragma unroll
    HB_M HB_SWAP32(zs[m], HB_ADD)
#pragma unroll
    HB_M HB_SWAP32(zt[m], HB_ADD)
#pragma unroll
    HB_M { zm[m] = -nmz[m]; qF[m] = tF[m] * __builtin_amdgcn_rcpf(sF[m]) * (1.0f / FK); }   // (zm: the maximum in log2 units: the merge uses exp2)
#undef HB_M
#undef HB_SWAP16
#undef HB_SWAP32
#undef HB_ADD
    if (qd == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const long long row = row0 + 16 * (M0 + m) + r;
        if (full_ok && row < a.m) a.q[row * A + slot] = qF[m];
        fb[16 * (M0 + m) + r] = zm[m];
        fb[8 * FM + 16 * (M0 + m) + r] = zs[m];
        fb[16 * FM + 16 * (M0 + m) + r] = zt[m];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    });
    if (p < 2) HB_FSTAMP(8 + 3 * p);
    __syncthreads();
    // ---- the pass's two extra actions: combine their four quarters (wavefronts 4 g .. 4 g + 3). ONE buffer: a second barrier
    // (below) separates these reads from the next pass's writes.
    if (tid < 2 * FM) {
      const int rr = tid & (FM - 1), g = tid >> 7;
      const int ea = full_cap + 2 * p + g;
      const long long row = row0 + rr;
      if (ea < A && row < a.m) {
        const float* f = reinterpret_cast<const float*>(lds + H_BYTES) + (4 * g) * FM + rr;
        const float m0 = f[0], m1 = f[FM], m2 = f[2 * FM], m3 = f[3 * FM];
        const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float e0 = exp2_fast(m0 - M), e1 = exp2_fast(m1 - M), e2 = exp2_fast(m2 - M), e3 = exp2_fast(m3 - M);
        const float* fs = f + 8 * FM;
        const float* ft = f + 16 * FM;
        const float S = fs[0] * e0 + fs[FM] * e1 + fs[2 * FM] * e2 + fs[3 * FM] * e3;
        const float T = ft[0] * e0 + ft[FM] * e1 + ft[2 * FM] * e2 + ft[3 * FM] * e3;
        a.q[row * A + ea] = T * __builtin_amdgcn_rcpf(S) * (1.0f / FK);
      }
    }
    if (p + 1 < n_pass) __syncthreads();
    if (p < 2) HB_FSTAMP(9 + 3 * p);
  }
  if (a.legal) {
    // ---- action selection for the workgroup's 128 rows, one lane per row. The q values were stored by all eight wavefronts:
    // every wavefront drains its stores (they are complete once L2 has them), the barrier orders that before the reads, and the
    // reads are device-scope loads (served by L2, never by this CU's L1).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < FM && row0 + tid < a.m) {
      const long long g = row0 + tid;
      const unsigned long long gid = static_cast<unsigned long long>(a.first_gid + g);
      uint32_t rnd[4];
      hb::philox4x32_10(static_cast<uint32_t>(a.draw), static_cast<uint32_t>(a.draw >> 32), static_cast<uint32_t>(gid),
                        static_cast<uint32_t>(gid >> 32), static_cast<uint32_t>(a.seed), static_cast<uint32_t>(a.seed >> 32), rnd);
      const float* qr = a.q + g * A;
      const int8_t* lr = a.legal + g * A;
      float best = -INFINITY;
      unsigned long long legal_mask = 0, ties = 0;
      for (int i = 0; i < A; ++i) {
        const float v = __hip_atomic_load(qr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lr[i] != 0) {
          legal_mask |= 1ull << i;
          if (v > best) { best = v; ties = 1ull << i; }
          else if (v == best) ties |= 1ull << i;
        }
      }
      const float u = static_cast<float>(rnd[0] >> 8) * (1.0f / 16777216.0f);
      unsigned long long pool = (u < a.epsilon) ? legal_mask : ties;
      if (pool == 0) pool = legal_mask;
      int pick = 0;
      const int c = __popcll(pool);
      if (c > 0) {
        int k = static_cast<int>(__umulhi(rnd[1], static_cast<uint32_t>(c)));
        while (k-- > 0) pool &= pool - 1;
        pick = __ffsll(static_cast<long long>(pool)) - 1;
      }
      a.actions[g] = pick;
    }
  }
  HB_FSTAMP_REAL(15);
}

// ---- packer: effective weights (bf16, row-major, possibly padded GEMM operands) -> the fragment-major copies
struct PackArgs {
  const __hip_bfloat16* w1; int w1_ld;
  const __hip_bfloat16* b1;
  const __hip_bfloat16* w2; int w2_ld;
  const __hip_bfloat16* b2;
  int obs_len, s1, n_actions, n_pass;
  uint4* w1f; float* b1f; uint4* w2f; float* b2f;
  int chunks1, chunks2;   // 16-byte chunks of w1f / w2f
  int f16;                // the 16-bit elements are fp16 (only the biases are converted: the weights are copied as they are)
  // optional (hb_actor_fused_pack_thin): the k-contiguous (transposed) copies hb_thin_gemm reads, [512][w1t_ld] and
  // [n_actions * 51][w2t_ld] — every thread already holds 8 consecutive k of one column, so they cost one more 16-byte store
  uint16_t* w1t; int w1t_ld;
  uint16_t* w2t; int w2t_ld;
};
__device__ __forceinline__ float bias_to_float(const __hip_bfloat16* b, int i, int f16) {
  return f16 ? __half2float(reinterpret_cast<const __half*>(b)[i]) : __bfloat162float(b[i]);
}
// physical column -> logit index (action * 51 + atom) or -1
__host__ __device__ __forceinline__ int fused_logit(int col, int n_actions, int n_pass) {
  const int slot = col >> 6, c = col & 63, n = c >> 4, q = (c >> 2) & 3, j = c & 3;
  const int r16 = 4 * n + j;
  const int full_cap = 8 * n_pass;
  int action, atom;
  if (r16 < 12 || (r16 == 12 && q < 3)) {
    action = slot < full_cap ? slot : n_actions;
    if (action >= n_actions) return -1;
    atom = r16 < 12 ? 4 * r16 + q : 48 + q;
  } else {
    const int p = slot >> 3, w = slot & 7, f = w & 3;
    action = full_cap + 2 * p + (w >> 2);
    atom = r16 == 12 ? 13 * f + 12 : 13 * f + 4 * (r16 - 13) + q;
  }
  if (action >= n_actions || atom >= FK) return -1;
  return action * FK + atom;
}
__global__ __launch_bounds__(256) void actor_fused_pack_kernel(const PackArgs a) {
  const int id = static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x);
  const __hip_bfloat16 zero = __float2bfloat16(0.f);
  if (id < a.chunks1) {
    const int lane = id & 63, nt = (id >> 6) & 31, s = id >> 11;
    const int unit = 16 * nt + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    __hip_bfloat16 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (k0 + i < a.obs_len) ? a.w1[static_cast<long long>(k0 + i) * a.w1_ld + unit] : zero;
    a.w1f[id] = *reinterpret_cast<const uint4*>(v);
    if (a.w1t) *reinterpret_cast<uint4*>(a.w1t + static_cast<size_t>(unit) * a.w1t_ld + k0) = *reinterpret_cast<const uint4*>(v);
    if (id < FH) a.b1f[id] = bias_to_float(a.b1, id, a.f16);
  } else if (id < a.chunks1 + a.chunks2) {
    const int id2 = id - a.chunks1;
    const int lane = id2 & 63, ntp = (id2 >> 6) & 31, s = (id2 >> 11) % S2, pass = (id2 >> 11) / S2;
    const int col = 16 * (32 * pass + ntp) + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    const int lg = fused_logit(col, a.n_actions, a.n_pass);
    __hip_bfloat16 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lg >= 0 ? a.w2[static_cast<long long>(k0 + i) * a.w2_ld + lg] : zero;
    a.w2f[id2] = *reinterpret_cast<const uint4*>(v);
    if (a.w2t && lg >= 0) *reinterpret_cast<uint4*>(a.w2t + static_cast<size_t>(lg) * a.w2t_ld + k0) = *reinterpret_cast<const uint4*>(v);
    if (s == 0 && (lane >> 4) == 0) a.b2f[col] = lg >= 0 ? bias_to_float(a.b2, lg, a.f16) : NEG_BIG;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
int passes_for(int n_actions) { return (n_actions + 9) / 10; }

}  // namespace

extern "C" {

int hb_actor_fused_supported(int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms) {
  return obs_len >= 1 && obs_len <= 4096 && hidden == FH && n_atoms == FK && n_actions >= 1 && n_actions <= 80 ? 1 : 0;   // (<= 32 passes)
}

int hb_actor_fused_columns(int32_t n_actions, int32_t* phys_of_logit_host) {
  if (!phys_of_logit_host) return fail(HB_ERR_INVALID, "null argument");
  if (n_actions < 1 || n_actions > 80) return fail(HB_ERR_INVALID, "n_actions must be 1..80");
  const int np = passes_for(n_actions);
  for (int j = 0; j < n_actions * FK; ++j) phys_of_logit_host[j] = -1;
  for (int col = 0; col < 512 * np; ++col) {
    const int lg = fused_logit(col, n_actions, np);
    if (lg >= 0) phys_of_logit_host[lg] = col;
  }
  for (int j = 0; j < n_actions * FK; ++j)
    if (phys_of_logit_host[j] < 0) return fail(HB_ERR_INVALID, "internal: logit column %d has no physical column", j);
  return HB_OK;
}

int hb_actor_fused_sizes(int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms, int64_t* w1f_bytes, int64_t* w2f_bytes,
                         int32_t* b2f_floats) {
  if (!hb_actor_fused_supported(obs_len, hidden, n_actions, n_atoms)) return fail(HB_ERR_INVALID, "shape not covered by the fused actor kernel");
  const int s1 = 2 * ((obs_len + 63) / 64), np = passes_for(n_actions);
  if (w1f_bytes) *w1f_bytes = static_cast<int64_t>(32) * s1 * 64 * 16;
  if (w2f_bytes) *w2f_bytes = static_cast<int64_t>(32) * np * S2 * 64 * 16;
  if (b2f_floats) *b2f_floats = 512 * np;
  return HB_OK;
}

int hb_actor_fused_pack_dt(const void* w1_dev, int32_t w1_ld, const void* b1_dev, const void* w2_dev, int32_t w2_ld, const void* b2_dev,
                           int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms, void* w1f_dev, float* b1f_dev,
                           void* w2f_dev, float* b2f_dev, int32_t dtype, void* stream) {
  return hb_actor_fused_pack_thin(w1_dev, w1_ld, b1_dev, w2_dev, w2_ld, b2_dev, obs_len, hidden, n_actions, n_atoms, w1f_dev, b1f_dev,
                                  w2f_dev, b2f_dev, nullptr, 0, nullptr, 0, dtype, stream);
}

int hb_actor_fused_pack_thin(const void* w1_dev, int32_t w1_ld, const void* b1_dev, const void* w2_dev, int32_t w2_ld, const void* b2_dev,
                             int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms, void* w1f_dev, float* b1f_dev,
                             void* w2f_dev, float* b2f_dev, void* w1t_dev, int32_t w1t_ld, void* w2t_dev, int32_t w2t_ld,
                             int32_t dtype, void* stream) {
  if (dtype != 1 && dtype != 2) return fail(HB_ERR_INVALID, "dtype must be 1 (bf16) or 2 (f16)");
  if (!w1_dev || !b1_dev || !w2_dev || !b2_dev || !w1f_dev || !b1f_dev || !w2f_dev || !b2f_dev) return fail(HB_ERR_INVALID, "null argument");
  if (!hb_actor_fused_supported(obs_len, hidden, n_actions, n_atoms)) return fail(HB_ERR_INVALID, "shape not covered by the fused actor kernel");
  if (w1_ld < hidden || w2_ld < n_actions * n_atoms) return fail(HB_ERR_INVALID, "row strides shorter than the rows");
  if (!aligned16(w1f_dev) || !aligned16(w2f_dev)) return fail(HB_ERR_ALIGN, "w1f / w2f must be 16-byte aligned");
  PackArgs p{};
  p.w1 = static_cast<const __hip_bfloat16*>(w1_dev); p.w1_ld = w1_ld; p.b1 = static_cast<const __hip_bfloat16*>(b1_dev);
  p.w2 = static_cast<const __hip_bfloat16*>(w2_dev); p.w2_ld = w2_ld; p.b2 = static_cast<const __hip_bfloat16*>(b2_dev);
  p.obs_len = obs_len; p.s1 = 2 * ((obs_len + 63) / 64); p.n_actions = n_actions; p.n_pass = passes_for(n_actions);
  p.w1f = static_cast<uint4*>(w1f_dev); p.b1f = b1f_dev; p.w2f = static_cast<uint4*>(w2f_dev); p.b2f = b2f_dev;
  p.chunks1 = 32 * p.s1 * 64; p.chunks2 = 32 * p.n_pass * S2 * 64;
  p.f16 = dtype == 2 ? 1 : 0;
  if (w1t_dev && (w1t_ld < 32 * p.s1 || w1t_ld % 8 || !aligned16(w1t_dev))) return fail(HB_ERR_INVALID, "w1t: row stride must cover the padded K (a multiple of 8), 16-byte aligned");
  if (w2t_dev && (w2t_ld < hidden || w2t_ld % 8 || !aligned16(w2t_dev))) return fail(HB_ERR_INVALID, "w2t: row stride must cover the hidden units (a multiple of 8), 16-byte aligned");
  p.w1t = static_cast<uint16_t*>(w1t_dev); p.w1t_ld = w1t_ld; p.w2t = static_cast<uint16_t*>(w2t_dev); p.w2t_ld = w2t_ld;
  const int blocks = (p.chunks1 + p.chunks2 + 255) / 256;
  hipLaunchKernelGGL(actor_fused_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_actor_fused_pack(const void* w1_dev, int32_t w1_ld, const void* b1_dev, const void* w2_dev, int32_t w2_ld, const void* b2_dev,
                        int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms, void* w1f_dev, float* b1f_dev, void* w2f_dev,
                        float* b2f_dev, void* stream) {
  return hb_actor_fused_pack_dt(w1_dev, w1_ld, b1_dev, w2_dev, w2_ld, b2_dev, obs_len, hidden, n_actions, n_atoms, w1f_dev, b1f_dev,
                                w2f_dev, b2f_dev, 1, stream);
}

struct FusedSelect {
  const int8_t* legal;
  int32_t* actions;
  float epsilon;
  uint64_t seed, draw;
  int64_t first_gid;
};
static int fused_launch(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev, const float* b1f_dev,
                        const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden, int32_t n_actions,
                        int32_t n_atoms, float* q_dev, unsigned long long* stamps_dev, const FusedSelect* sel, int32_t dtype, void* stream);

int hb_actor_fused_q_dt(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev, const float* b1f_dev,
                        const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden, int32_t n_actions,
                        int32_t n_atoms, float* q_dev, int32_t dtype, void* stream) {
  return fused_launch(obs_bits_dev, n_rows, obs_len, w1f_dev, b1f_dev, w2f_dev, b2f_dev, support_dev, hidden, n_actions, n_atoms, q_dev,
                      nullptr, nullptr, dtype, stream);
}

int hb_actor_fused_act_dt(const uint32_t* obs_bits_dev, const int8_t* legal_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev,
                          const float* b1f_dev, const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden,
                          int32_t n_actions, int32_t n_atoms, float* q_dev, float epsilon, uint64_t seed, uint64_t draw,
                          int64_t first_game_id, int32_t* actions_dev, int32_t dtype, void* stream) {
  if (!legal_dev || !actions_dev) return fail(HB_ERR_INVALID, "null argument");
  if (n_actions > 64) return fail(HB_ERR_INVALID, "n_actions must be 1..64 for the fused selection");
  const FusedSelect sel{legal_dev, actions_dev, epsilon, seed, draw, first_game_id};
  return fused_launch(obs_bits_dev, n_rows, obs_len, w1f_dev, b1f_dev, w2f_dev, b2f_dev, support_dev, hidden, n_actions, n_atoms, q_dev,
                      nullptr, &sel, dtype, stream);
}

int hb_actor_fused_q(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev, const float* b1f_dev,
                     const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden, int32_t n_actions, int32_t n_atoms,
                     float* q_dev, void* stream) {
  return fused_launch(obs_bits_dev, n_rows, obs_len, w1f_dev, b1f_dev, w2f_dev, b2f_dev, support_dev, hidden, n_actions, n_atoms, q_dev,
                      nullptr, nullptr, 1, stream);
}

int hb_actor_fused_act(const uint32_t* obs_bits_dev, const int8_t* legal_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev,
                       const float* b1f_dev, const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden,
                       int32_t n_actions, int32_t n_atoms, float* q_dev, float epsilon, uint64_t seed, uint64_t draw,
                       int64_t first_game_id, int32_t* actions_dev, void* stream) {
  return hb_actor_fused_act_dt(obs_bits_dev, legal_dev, n_rows, obs_len, w1f_dev, b1f_dev, w2f_dev, b2f_dev, support_dev, hidden,
                               n_actions, n_atoms, q_dev, epsilon, seed, draw, first_game_id, actions_dev, 1, stream);
}

#ifdef HB_STAMPS
// diagnostic library only: the same launch with per-wavefront phase stamps (16 u64 per wavefront, 8 wavefronts per 128 rows)
int hb_actor_fused_q_stamped(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev, const float* b1f_dev,
                             const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden, int32_t n_actions,
                             int32_t n_atoms, float* q_dev, unsigned long long* stamps_dev, void* stream) {
  return fused_launch(obs_bits_dev, n_rows, obs_len, w1f_dev, b1f_dev, w2f_dev, b2f_dev, support_dev, hidden, n_actions, n_atoms, q_dev,
                      stamps_dev, nullptr, 1, stream);
}
#endif

static int fused_launch(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev, const float* b1f_dev,
                        const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden, int32_t n_actions,
                        int32_t n_atoms, float* q_dev, unsigned long long* stamps_dev, const FusedSelect* sel, int32_t dtype, void* stream) {
  if (dtype != 1 && dtype != 2) return fail(HB_ERR_INVALID, "dtype must be 1 (bf16) or 2 (f16)");
  if (!obs_bits_dev || !w1f_dev || !b1f_dev || !w2f_dev || !b2f_dev || !support_dev || !q_dev) return fail(HB_ERR_INVALID, "null argument");
  if (!hb_actor_fused_supported(obs_len, hidden, n_actions, n_atoms)) return fail(HB_ERR_INVALID, "shape not covered by the fused actor kernel");
  if (n_rows <= 0) return HB_OK;
  if (!aligned16(w1f_dev) || !aligned16(w2f_dev) || !aligned16(b1f_dev) || !aligned16(b2f_dev))
    return fail(HB_ERR_ALIGN, "w1f / w2f / b1f / b2f must be 16-byte aligned");
  FusedArgs a{};
  a.obs = obs_bits_dev; a.m = n_rows; a.words = (obs_len + 31) / 32; a.s1 = 2 * ((obs_len + 63) / 64);
  a.w1f = static_cast<const uint4*>(w1f_dev); a.b1 = b1f_dev; a.w2f = static_cast<const uint4*>(w2f_dev); a.b2 = b2f_dev;
  a.support = support_dev; a.q = q_dev; a.n_actions = n_actions; a.n_pass = passes_for(n_actions);
  a.stamps = stamps_dev;
  if (sel) {
    a.legal = sel->legal; a.actions = sel->actions; a.epsilon = sel->epsilon; a.seed = sel->seed; a.draw = sel->draw;
    a.first_gid = sel->first_gid;
  }
  const dim3 grid(static_cast<unsigned>((n_rows + FM - 1) / FM));
  if (dtype == 2) hipLaunchKernelGGL(actor_fused_kernel<true>, grid, dim3(FNT), 0, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(actor_fused_kernel<false>, grid, dim3(FNT), 0, static_cast<hipStream_t>(stream), a);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

}  // extern "C"
