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

#include <cstdint>

#include "../../include/hanabi_hip.h"
#include "common.hpp"

using hb::fail;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FM = 128;                 // rows per workgroup
constexpr int FH = 512;                 // hidden units (the reference's layers=[512], params.py:13)
constexpr int FNT = 512;                // threads
constexpr int FK = 51;                  // atoms (params.py:18)
constexpr int S2 = FH / 32;             // K steps of layer 2
constexpr int LUT_BYTES = 65536;        // 256 byte values x 16 slots x 16 B
constexpr int H_BYTES = FM * FH * 2;    // 131 072
constexpr int FRAG_BYTES = 8 * FM * 16; // one buffer of quarter-action triples: [wave][row] float4
constexpr int LDS_TOTAL = H_BYTES + 2 * FRAG_BYTES;   // 163 840 = all of a CU's LDS
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
};

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
  return static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&a)) |
         (static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}
// two observation bits -> two bf16 0.0 / 1.0 (0x3F80)
__device__ __forceinline__ uint32_t bits2_bf16(uint32_t b, int p) {
  const uint32_t t = (b >> (2 * p)) & 3u;
  return __umul24((t | (t << 15)) & 0x00010001u, 0x3F80u);
}
__device__ __forceinline__ bf16x8 as_frag(const uint4& v) {
  union { uint4 u; bf16x8 f; } c;
  c.u = v;
  return c.f;
}
__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

__global__ __launch_bounds__(FNT) void actor_fused_kernel(const FusedArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_TOTAL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, qd = lane >> 4;
  const long long row0 = static_cast<long long>(blockIdx.x) * FM;
  const int s1 = a.s1, G = (s1 + 3) >> 2;
  unsigned char* bits = lds + LUT_BYTES;   // [8 m][G][64 lanes] u32: byte i of the word = byte (4 (4 g + i) + q) of row 16 m + r

  // ---- first weights in flight before anything else
  const uint4* w1p = a.w1f + (4 * wave) * 64 + lane;   // step s: + s * 2048; n-tile n of this wavefront: + n * 64
  uint4 wa[4], wb[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) wa[n] = w1p[n * 64];

  // ---- table: entry (v, slot) at v * 256 + slot * 16
  for (int e = tid; e < 4096; e += FNT) {
    const uint32_t v = static_cast<uint32_t>(e) >> 4;
    *reinterpret_cast<uint4*>(lds + e * 16) = make_uint4(bits2_bf16(v, 0), bits2_bf16(v, 1), bits2_bf16(v, 2), bits2_bf16(v, 3));
  }
  // ---- observation bits, byte-transposed so that a lane finds the bytes of four consecutive K steps in one dword
  for (int it = tid; it < FM * G; it += FNT) {
    const int rr = it & (FM - 1), g = it >> 7;
    long long row = row0 + rr;
    if (row >= a.m) row = a.m - 1;   // rows past the end are computed and dropped
    const uint32_t* src = a.obs + row * a.words + 4 * g;
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (4 * g + i < a.words) ? src[i] : 0u;
    uint32_t* dst = reinterpret_cast<uint32_t*>(bits) + (static_cast<size_t>((rr >> 4) * G + g) * 64 + (rr & 15));
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const uint32_t o = ((w[0] >> (8 * qq)) & 0xFFu) | (((w[1] >> (8 * qq)) & 0xFFu) << 8) | (((w[2] >> (8 * qq)) & 0xFFu) << 16) |
                         (((w[3] >> (8 * qq)) & 0xFFu) << 24);
      dst[qq * 16] = o;
    }
  }
  __syncthreads();

  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // =========================== layer 1: H[128 rows][64 wave + 0..63] over K = 32 s1
  {
    const unsigned char* lut = lds + r * 16;
    const uint32_t* bw = reinterpret_cast<const uint32_t*>(bits) + lane;
#define HB_L1_STEP(I, WCUR, WNXT, HAVE_NEXT)                                                                   \
  {                                                                                                            \
    const int s_ = 4 * g + (I);                                                                                \
    if (HAVE_NEXT) {                                                                                           \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) WNXT[n] = w1p[(s_ + 1) * 2048 + n * 64];                    \
    }                                                                                                          \
    asm volatile("" ::: "memory"); /* the loads are ISSUED here, a whole step before their use */               \
    bf16x8 xf[8];                                                                                              \
    _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                                            \
      const uint32_t v = (xw[m] >> (8 * (I))) & 0xFFu;                                                         \
      xf[m] = *reinterpret_cast<const bf16x8*>(lut + v * 256u);                                                \
    }                                                                                                          \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                            \
      const bf16x8 wf = as_frag(WCUR[n]);                                                                      \
      _Pragma("unroll") for (int m = 0; m < 8; ++m)                                                            \
        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[m], acc[n][m], 0, 0, 0);                    \
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

  // first weights of layer 2 in flight across the barrier and the H write
  const uint4* w2p = a.w2f + (4 * wave) * 64 + lane;   // pass p: + p * S2 * 2048; step s: + s * 2048
#pragma unroll
  for (int n = 0; n < 4; ++n) wa[n] = w2p[n * 64];

  __syncthreads();   // every wavefront has finished reading the table and the bit rows: H may overlay them

  // ---- H = relu(acc + b1) as bf16: acc[n][m][j] = row 16 m + r, unit 64 wave + 16 n + 4 q + j; 8-byte LDS writes
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float4 b = *reinterpret_cast<const float4*>(a.b1 + 64 * wave + 16 * n + 4 * qd);
    const int chunk = 8 * wave + 2 * n + (qd >> 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float v0 = fmaxf(acc[n][m][0] + b.x, 0.f), v1 = fmaxf(acc[n][m][1] + b.y, 0.f);
      const float v2 = fmaxf(acc[n][m][2] + b.z, 0.f), v3 = fmaxf(acc[n][m][3] + b.w, 0.f);
      *reinterpret_cast<uint2*>(lds + (16 * m + r) * 1024 + ((chunk ^ r) << 4) + (qd & 1) * 8) =
          make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
    }
  }
  __syncthreads();

  // =========================== layer 2 in passes of 512 physical columns + the C51 expectation from the accumulators
  const int n_pass = a.n_pass, A = a.n_actions;
  const int full_cap = 8 * n_pass;
  // per-lane support values: whole action: atom 4 i + q (i < 12), atom 48 + q (the 13th register, lanes q < 3)
  float supF[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    const int k = 4 * i + qd;
    supF[i] = a.support[k < FK ? k : FK - 1];
  }
  const int fq = wave & 3;   // which quarter of its extra action this wavefront holds
  float supZ[4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int k = 13 * fq + 4 * i + qd;
    supZ[i] = a.support[k < FK ? k : FK - 1];
  }
  supZ[3] = a.support[13 * fq + 12 < FK ? 13 * fq + 12 : FK - 1];
  constexpr float LOG2E = 1.44269504088896340736f;
  const unsigned char* hrow = lds + r * 1024;

  for (int p = 0; p < n_pass; ++p) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4* wp = w2p + static_cast<long long>(p) * S2 * 2048;
#define HB_L2_STEP(S, WCUR, WNXT, HAVE_NEXT)                                                                   \
  {                                                                                                            \
    if (HAVE_NEXT) {                                                                                           \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) WNXT[n] = wp[((S) + 1) * 2048 + n * 64];                    \
    }                                                                                                          \
    asm volatile("" ::: "memory");                                                                             \
    bf16x8 xf[8];                                                                                              \
    const int ch = ((4 * (S) + qd) ^ r) << 4;                                                                  \
    _Pragma("unroll") for (int m = 0; m < 8; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(hrow + m * 16384 + ch); \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                            \
      const bf16x8 wf = as_frag(WCUR[n]);                                                                      \
      _Pragma("unroll") for (int m = 0; m < 8; ++m)                                                            \
        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[m], acc[n][m], 0, 0, 0);                    \
    }                                                                                                          \
  }
#pragma unroll 1
    for (int s = 0; s < S2; s += 2) {
      HB_L2_STEP(s, wa, wb, true)
      HB_L2_STEP(s + 1, wb, wa, (s + 2 < S2))
    }
#undef HB_L2_STEP
    // next pass's first weights in flight during the epilogue
    if (p + 1 < n_pass) {
#pragma unroll
      for (int n = 0; n < 4; ++n) wa[n] = wp[S2 * 2048 + n * 64];
    }

    // ---- epilogue of slot (p, wave): register (n, j) of lane group q is physical column 16 n + 4 q + j of the slot
    const int slot = 8 * p + wave;
    const float* bp = a.b2 + static_cast<long long>(slot) * 64 + 4 * qd;
    float4 bias[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) bias[n] = *reinterpret_cast<const float4*>(bp + 16 * n);
    const bool full_ok = slot < A && slot < full_cap;
    float4* fb = reinterpret_cast<float4*>(lds + H_BYTES + (p & 1) * FRAG_BYTES) + wave * FM;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float x[16];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        x[4 * n + 0] = (acc[n][m][0] + bias[n].x) * LOG2E;
        x[4 * n + 1] = (acc[n][m][1] + bias[n].y) * LOG2E;
        x[4 * n + 2] = (acc[n][m][2] + bias[n].z) * LOG2E;
        x[4 * n + 3] = (acc[n][m][3] + bias[n].w) * LOG2E;
      }
      // register 12 is shared: lanes q < 3 -> atoms 48..50 of the whole action, lanes q == 3 -> the quarter's 13th position
      const float xf12 = qd < 3 ? x[12] : NEG_BIG * LOG2E;
      const float xz3 = qd == 3 ? x[12] : NEG_BIG * LOG2E;
      // whole action
      float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
#pragma unroll
      for (int i = 4; i < 12; ++i) mx = fmaxf(mx, x[i]);
      mx = fmaxf(mx, xf12);
      float mz = fmaxf(fmaxf(x[13], x[14]), fmaxf(x[15], xz3));
      mx = fmaxf(mx, xor16(mx));
      mz = fmaxf(mz, xor16(mz));
      mx = fmaxf(mx, xor32(mx));
      mz = fmaxf(mz, xor32(mz));
      float sF = 0.f, tF = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float e = exp2_fast(x[i] - mx);
        sF += e;
        tF += e * supF[i];
      }
      {
        const float e = exp2_fast(xf12 - mx);
        sF += e;
        tF += e * supF[12];
      }
      float sZ = 0.f, tZ = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float e = exp2_fast(x[13 + i] - mz);
        sZ += e;
        tZ += e * supZ[i];
      }
      {
        const float e = exp2_fast(xz3 - mz);
        sZ += e;
        tZ += e * supZ[3];
      }
      sF += xor16(sF); tF += xor16(tF); sZ += xor16(sZ); tZ += xor16(tZ);
      sF += xor32(sF); tF += xor32(tF); sZ += xor32(sZ); tZ += xor32(tZ);
      if (qd == 0) {
        const long long row = row0 + 16 * m + r;
        if (full_ok && row < a.m) a.q[row * A + slot] = tF / sF * (1.0f / FK);
        fb[16 * m + r] = make_float4(mz, sZ, tZ, 0.f);   // (mz is in log2 units: exp2 below)
      }
    }
    __syncthreads();
    // ---- the pass's two extra actions: combine their four quarters (wavefronts 4 g .. 4 g + 3)
    if (tid < 2 * FM) {
      const int rr = tid & (FM - 1), g = tid >> 7;
      const int ea = full_cap + 2 * p + g;
      const long long row = row0 + rr;
      if (ea < A && row < a.m) {
        const float4* f4 = reinterpret_cast<const float4*>(lds + H_BYTES + (p & 1) * FRAG_BYTES) + (4 * g) * FM + rr;
        const float4 v0 = f4[0], v1 = f4[FM], v2 = f4[2 * FM], v3 = f4[3 * FM];
        const float M = fmaxf(fmaxf(v0.x, v1.x), fmaxf(v2.x, v3.x));
        const float e0 = exp2_fast(v0.x - M), e1 = exp2_fast(v1.x - M), e2 = exp2_fast(v2.x - M), e3 = exp2_fast(v3.x - M);
        const float S = v0.y * e0 + v1.y * e1 + v2.y * e2 + v3.y * e3;
        const float T = v0.z * e0 + v1.z * e1 + v2.z * e2 + v3.z * e3;
        a.q[row * A + ea] = T / S * (1.0f / FK);
      }
    }
  }
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
};
// physical column -> logit index (action * 51 + atom) or -1
__device__ __forceinline__ int fused_logit(int col, int n_actions, int n_pass) {
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
    if (id < FH) a.b1f[id] = __bfloat162float(a.b1[id]);
  } else if (id < a.chunks1 + a.chunks2) {
    const int id2 = id - a.chunks1;
    const int lane = id2 & 63, ntp = (id2 >> 6) & 31, s = (id2 >> 11) % S2, pass = (id2 >> 11) / S2;
    const int col = 16 * (32 * pass + ntp) + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    const int lg = fused_logit(col, a.n_actions, a.n_pass);
    __hip_bfloat16 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lg >= 0 ? a.w2[static_cast<long long>(k0 + i) * a.w2_ld + lg] : zero;
    a.w2f[id2] = *reinterpret_cast<const uint4*>(v);
    if (s == 0 && (lane >> 4) == 0) a.b2f[col] = lg >= 0 ? __bfloat162float(a.b2[lg]) : NEG_BIG;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
int passes_for(int n_actions) { return (n_actions + 9) / 10; }

}  // namespace

extern "C" {

int hb_actor_fused_supported(int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms) {
  return obs_len >= 1 && obs_len <= 4096 && hidden == FH && n_atoms == FK && n_actions >= 1 && n_actions <= 80 ? 1 : 0;
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

int hb_actor_fused_pack(const void* w1_dev, int32_t w1_ld, const void* b1_dev, const void* w2_dev, int32_t w2_ld, const void* b2_dev,
                        int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms, void* w1f_dev, float* b1f_dev, void* w2f_dev,
                        float* b2f_dev, void* stream) {
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
  const int blocks = (p.chunks1 + p.chunks2 + 255) / 256;
  hipLaunchKernelGGL(actor_fused_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_actor_fused_q(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev, const float* b1f_dev,
                     const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden, int32_t n_actions, int32_t n_atoms,
                     float* q_dev, void* stream) {
  if (!obs_bits_dev || !w1f_dev || !b1f_dev || !w2f_dev || !b2f_dev || !support_dev || !q_dev) return fail(HB_ERR_INVALID, "null argument");
  if (!hb_actor_fused_supported(obs_len, hidden, n_actions, n_atoms)) return fail(HB_ERR_INVALID, "shape not covered by the fused actor kernel");
  if (n_rows <= 0) return HB_OK;
  if (!aligned16(w1f_dev) || !aligned16(w2f_dev) || !aligned16(b1f_dev) || !aligned16(b2f_dev))
    return fail(HB_ERR_ALIGN, "w1f / w2f / b1f / b2f must be 16-byte aligned");
  FusedArgs a{};
  a.obs = obs_bits_dev; a.m = n_rows; a.words = (obs_len + 31) / 32; a.s1 = 2 * ((obs_len + 63) / 64);
  a.w1f = static_cast<const uint4*>(w1f_dev); a.b1 = b1f_dev; a.w2f = static_cast<const uint4*>(w2f_dev); a.b2 = b2f_dev;
  a.support = support_dev; a.q = q_dev; a.n_actions = n_actions; a.n_pass = passes_for(n_actions);
  const dim3 grid(static_cast<unsigned>((n_rows + FM - 1) / FM));
  hipLaunchKernelGGL(actor_fused_kernel, grid, dim3(FNT), 0, static_cast<hipStream_t>(stream), a);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

}  // extern "C"
