// gemm_probe.hip — diagnostic only (never shipped, never loaded by the package): variants of the actor's GEMM core
// side by side, with optional in-kernel stamps, to find out what the main loop of csrc/actor.hip spends its time on.
//   out[M, N] bf16 = x[M, K] (bf16, k-contiguous) @ wt[N, K]^T (bf16, k-contiguous), fp32 accumulate, no bias
// Variant 0: the shipped structure (global -> registers -> LDS, double-buffered, one barrier per K step).
// Variant 1: global_load_lds (16 B) straight into the swizzled LDS image, two stages, one barrier per K step.
// Variant 2: variant 1 with the MFMA block split into four quadrants, each quadrant's fragments read while the
//            previous quadrant multiplies (software-pipelined fragments), s_setprio around the MFMAs.
// Variant 3: 8 waves as two groups half a K step apart (group 1 runs one barrier behind): one group's MFMA segment
//            beside the other group's LDS-read / load-issue segment (the guide's 8-phase ping-pong, coarse form).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int ROWB = BK * 2;
constexpr int OPB = BM * ROWB;            // 32 KB per operand tile
constexpr int STAGE = 2 * OPB;            // 64 KB
constexpr int OUT_LD = 520;
constexpr int LDS_BYTES = BM * OUT_LD > 2 * STAGE ? BM * OUT_LD : 2 * STAGE;

struct Args {
  const __hip_bfloat16* x;
  const __hip_bfloat16* wt;
  __hip_bfloat16* out;
  unsigned long long* stamps;  // [blocks * 8 waves][64] or null
  long long m;
  int k, n;
};

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
  return static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&a)) | (static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}

#define STAMP(slot)                                                                                  \
  do {                                                                                               \
    if (STAMPS) {                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      unsigned long long t_;                                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      if (a.stamps && lane == 0 && (slot) < 64)                                                      \
        a.stamps[((static_cast<long long>(blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * 64 + (slot)] = t_; \
    }                                                                                                \
  } while (0)

// epilogue shared by all variants: acc[n][m][j] = out[row wr*128 + 16 m + (lane & 15)][col wc*64 + 16 n + 4 (lane >> 4) + j]
__device__ __forceinline__ void write_out(const Args& a, unsigned char* lds, f32x4 (&acc)[4][8], int tid, long long row0, int col0) {
  const int lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3, fq = lane >> 4;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int cl = wc * 64 + n * 16 + fq * 4;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int rl = wr * 128 + m * 16 + (lane & 15);
      *reinterpret_cast<uint2*>(lds + rl * OUT_LD + cl * 2) =
          make_uint2(pack_bf16(acc[n][m][0], acc[n][m][1]), pack_bf16(acc[n][m][2], acc[n][m][3]));
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int id = tid + NT * i, r = id >> 5, ch = id & 31;
    if (row0 + r < a.m)
      *reinterpret_cast<uint4*>(a.out + (row0 + r) * a.n + col0 + ch * 8) = *reinterpret_cast<const uint4*>(lds + r * OUT_LD + ch * 16);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template <bool STAMPS>
__global__ __launch_bounds__(NT) void gemm_v0(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const long long row0 = static_cast<long long>(blockIdx.x) * BM;
  const int col0 = static_cast<int>(blockIdx.y) * BN;
  const int kt_n = a.k / BK;
  const int sc = tid & 7, sr = tid >> 3;
  uint4 xr0, xr1, xr2, xr3, wq0, wq1, wq2, wq3;
  STAMP(0);
#define P_LOAD_ONE(i, XR, WQ)                                                                   \
  {                                                                                             \
    const int r = sr + 64 * (i);                                                                \
    long long row = row0 + r;                                                                   \
    if (row >= a.m) row = a.m - 1;                                                              \
    const int k = kt_ * BK + sc * 8;                                                            \
    XR = *reinterpret_cast<const uint4*>(a.x + row * a.k + k);                                  \
    WQ = *reinterpret_cast<const uint4*>(a.wt + static_cast<long long>(col0 + r) * a.k + k);    \
  }
#define P_LOAD_STAGE(KT) { const int kt_ = (KT); P_LOAD_ONE(0, xr0, wq0) P_LOAD_ONE(1, xr1, wq1) P_LOAD_ONE(2, xr2, wq2) P_LOAD_ONE(3, xr3, wq3) }
#define P_STORE_ONE(i, XR, WQ)                                              \
  {                                                                         \
    const int r = sr + 64 * (i);                                            \
    const int off = r * ROWB + ((sc ^ ((r >> 1) & 7)) << 4);                \
    *reinterpret_cast<uint4*>(base_ + off) = XR;                            \
    *reinterpret_cast<uint4*>(base_ + OPB + off) = WQ;                      \
  }
#define P_STORE_STAGE(BUF) { unsigned char* base_ = lds + (BUF) * STAGE; P_STORE_ONE(0, xr0, wq0) P_STORE_ONE(1, xr1, wq1) P_STORE_ONE(2, xr2, wq2) P_STORE_ONE(3, xr3, wq3) }
  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = OPB + (wc * 64 + (lane & 15)) * ROWB;
  P_LOAD_STAGE(0)
  P_STORE_STAGE(0)
  if (kt_n > 1) P_LOAD_STAGE(1)
  __syncthreads();
  STAMP(1);
  for (int kt = 0; kt < kt_n; ++kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    if (kt + 1 < kt_n) P_STORE_STAGE((kt + 1) & 1)
    STAMP(2 + 4 * kt);
    if (kt + 2 < kt_n) P_LOAD_STAGE(kt + 2)
    STAMP(3 + 4 * kt);
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
    STAMP(4 + 4 * kt);
    __syncthreads();
    STAMP(5 + 4 * kt);
  }
  write_out(a, lds, acc, tid, row0, col0);
  STAMP(63);
}

// ---------------------------------------------------------------------------------------------------------------------
// direct-to-LDS staging: a wave instruction writes 1 KiB = 8 rows x 128 B, lane l -> (row l >> 3, physical chunk l & 7);
// the XOR swizzle is applied on the SOURCE address (the lane fetches the logical chunk that belongs at its slot)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);  // C-style casts: generic -> global / LDS address space
}

// stage K tile kt of both operands into `base` (X at base, W at base + OPB): 8 glds per lane
__device__ __forceinline__ void stage_tile(const Args& a, unsigned char* base, int kt, int wave, int lane, long long row0, int col0) {
  const int lr = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 32 * wave + 8 * i + lr;
    const int c = pc ^ ((r >> 1) & 7);
    long long row = row0 + r;
    if (row >= a.m) row = a.m - 1;
    unsigned char* dst = base + (32 * wave + 8 * i) * ROWB;  // wave-uniform; the hardware adds lane * 16
    glds16(a.x + row * a.k + kt * BK + c * 8, dst);
    glds16(a.wt + static_cast<long long>(col0 + r) * a.k + kt * BK + c * 8, dst + OPB);
  }
}

template <bool STAMPS>
__global__ __launch_bounds__(NT) void gemm_v1(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const long long row0 = static_cast<long long>(blockIdx.x) * BM;
  const int col0 = static_cast<int>(blockIdx.y) * BN;
  const int kt_n = a.k / BK;
  STAMP(0);
  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = OPB + (wc * 64 + (lane & 15)) * ROWB;
  stage_tile(a, lds, 0, wave, lane, row0, col0);
  STAMP(1);
  for (int kt = 0; kt < kt_n; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for every wave; everyone is done reading the other buffer
    STAMP(2 + 4 * kt);
    if (kt + 1 < kt_n) stage_tile(a, lds + ((kt + 1) & 1) * STAGE, kt + 1, wave, lane, row0, col0);
    STAMP(3 + 4 * kt);
    const unsigned char* cur = lds + (kt & 1) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = ((kk * 4 + fq) ^ fsw) << 4;
      bf16x8 wf[4], xf[8];
#pragma unroll
      for (int n = 0; n < 4; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch);
#pragma unroll
      for (int m = 0; m < 8; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[n][m], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    STAMP(4 + 4 * kt);
  }
  __syncthreads();
  write_out(a, lds, acc, tid, row0, col0);
  STAMP(63);
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 3: two groups of four waves (wr = 0 / 1) half a step apart. Per K tile each group has a READ segment (all 24
// fragment reads of the tile into registers + the glds issue of its share of tile kt + 2) and an MFMA segment (64 MFMAs from
// registers). Group 1 runs one barrier behind, so on every SIMD one wave multiplies while its partner reads / issues loads.
// LDS: three stages of 48 KB?  no: 2 x 64 KB do not allow tile kt + 2 while kt + 1 is unread; the fragments of tile kt are in
// REGISTERS during the MFMA segment, so buffer kt & 1 is free again as soon as both groups finished READ(kt).
template <bool STAMPS>
__global__ __launch_bounds__(NT) void gemm_v3(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const long long row0 = static_cast<long long>(blockIdx.x) * BM;
  const int col0 = static_cast<int>(blockIdx.y) * BN;
  const int kt_n = a.k / BK;
  STAMP(0);
  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = OPB + (wc * 64 + (lane & 15)) * ROWB;
  bf16x8 wf[2][4], xf[2][8];
  auto read_frags = [&](const unsigned char* cur) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = ((kk * 4 + fq) ^ fsw) << 4;
#pragma unroll
      for (int n = 0; n < 4; ++n) wf[kk][n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch);
#pragma unroll
      for (int m = 0; m < 8; ++m) xf[kk][m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch);
    }
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][n], xf[kk][m], acc[n][m], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // Schedule in half-steps h = 0, 1, 2, ... separated by ONE workgroup barrier each:
  //   group 0: READ(t) in half-step 2t,     MFMA(t) in half-step 2t + 1
  //   group 1: READ(t) in half-step 2t + 1, MFMA(t) in half-step 2t + 2
  // Staging: tile t + 1 is issued by ALL waves at the start of half-step 2t + 1... but buffer (t + 1) & 1 = (t - 1) & 1 was
  // last read by group 1 in half-step 2(t - 1) + 1 = 2t - 1 < 2t + 1: free. It must have landed before group 0 reads it in
  // half-step 2t + 2: every wave waits vmcnt(0) at the end of half-step 2t + 1 (before that barrier).
  stage_tile(a, lds, 0, wave, lane, row0, col0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  STAMP(1);
  const int halves = 2 * kt_n + 1;
  for (int h = 0; h < halves; ++h) {
    const int t_read = wr == 0 ? ((h & 1) == 0 ? h / 2 : -1) : ((h & 1) == 1 ? h / 2 : -1);
    const int t_mfma = wr == 0 ? ((h & 1) == 1 ? h / 2 : -1) : ((h & 1) == 0 ? h / 2 - 1 : -1);
    if ((h & 1) == 1 && h / 2 + 1 < kt_n) stage_tile(a, lds + ((h / 2 + 1) & 1) * STAGE, h / 2 + 1, wave, lane, row0, col0);
    if (t_read >= 0 && t_read < kt_n) read_frags(lds + (t_read & 1) * STAGE);
    if (t_mfma >= 0 && t_mfma < kt_n) mfmas();
    if (h < 62) STAMP(2 + h);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  write_out(a, lds, acc, tid, row0, col0);
  STAMP(63);
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 4: register staging (loads two tiles ahead, as variant 0) + two wave groups half an iteration apart: group 0
// (waves 0-3, rows 0-127) stores / loads in the first half of an iteration and multiplies in the second; group 1 (waves 4-7,
// rows 128-255, the SIMD partners of group 0) multiplies the PREVIOUS tile in the first half and stores / loads in the second.
// One barrier after each half. LDS: W in a ring of 3 tiles (96 KB: a tile is read by group 0 one half after it is complete
// and by group 1 another half later), X per group double-buffered (4 x 16 KB): 160 KB.
constexpr int W_TILE = BN * ROWB;            // 32 KB
constexpr int XH_TILE = 128 * ROWB;          // 16 KB: one group's 128 rows
constexpr int LDS4_BYTES = 3 * W_TILE + 4 * XH_TILE;   // 163 840
template <bool STAMPS>
__global__ __launch_bounds__(NT) void gemm_v4(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS4_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const long long row0 = static_cast<long long>(blockIdx.x) * BM;
  const int col0 = static_cast<int>(blockIdx.y) * BN;
  const int kt_n = a.k / BK;
  const int t = tid & 255, sc = t & 7, srl = t >> 3;   // within the group: chunk, row 0..31 (+ 32 i)
  uint4 xr0, xr1, xr2, xr3, wq0, wq1, wq2, wq3;
  STAMP(0);
#define Q_LOAD_ONE(i, XR, WQ)                                                                                  \
  {                                                                                                            \
    const int r = wr * 128 + srl + 32 * (i);                                                                   \
    long long row = row0 + r;                                                                                  \
    if (row >= a.m) row = a.m - 1;                                                                             \
    const int k = kt_ * BK + sc * 8;                                                                           \
    XR = *reinterpret_cast<const uint4*>(a.x + row * a.k + k);                                                 \
    WQ = *reinterpret_cast<const uint4*>(a.wt + static_cast<long long>(col0 + r) * a.k + k);                   \
  }
#define Q_LOAD_STAGE(KT) { const int kt_ = (KT); Q_LOAD_ONE(0, xr0, wq0) Q_LOAD_ONE(1, xr1, wq1) Q_LOAD_ONE(2, xr2, wq2) Q_LOAD_ONE(3, xr3, wq3) }
#define Q_STORE_ONE(i, XR, WQ)                                                                  \
  {                                                                                             \
    const int rl = srl + 32 * (i);               /* row within the group's 128 */                \
    const int r = wr * 128 + rl;                 /* row within the W tile */                      \
    *reinterpret_cast<uint4*>(xb_ + rl * ROWB + ((sc ^ ((rl >> 1) & 7)) << 4)) = XR;            \
    *reinterpret_cast<uint4*>(wb_ + r * ROWB + ((sc ^ ((r >> 1) & 7)) << 4)) = WQ;              \
  }
#define Q_STORE_STAGE(KT)                                                                       \
  {                                                                                             \
    unsigned char* wb_ = lds + ((KT) % 3) * W_TILE;                                             \
    unsigned char* xb_ = lds + 3 * W_TILE + (wr * 2 + ((KT) & 1)) * XH_TILE;                     \
    Q_STORE_ONE(0, xr0, wq0) Q_STORE_ONE(1, xr1, wq1) Q_STORE_ONE(2, xr2, wq2) Q_STORE_ONE(3, xr3, wq3) \
  }
  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (lane & 15) * ROWB;                       // within the group's X buffer
  const int w_off = (wc * 64 + (lane & 15)) * ROWB;           // within a W tile
#define Q_MFMA(KT)                                                                                                   \
  {                                                                                                                  \
    const unsigned char* wcur = lds + ((KT) % 3) * W_TILE;                                                           \
    const unsigned char* xcur = lds + 3 * W_TILE + (wr * 2 + ((KT) & 1)) * XH_TILE;                                   \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                               \
      const int ch = ((kk * 4 + fq) ^ fsw) << 4;                                                                     \
      bf16x8 wf[4], xf[8];                                                                                           \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(wcur + w_off + n * 16 * ROWB + ch); \
      _Pragma("unroll") for (int m = 0; m < 8; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(xcur + x_off + m * 16 * ROWB + ch); \
      _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                  \
        _Pragma("unroll") for (int m = 0; m < 8; ++m)                                                                \
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[n][m], 0, 0, 0);                     \
    }                                                                                                                \
  }
  Q_LOAD_STAGE(0)
  Q_STORE_STAGE(0)
  if (kt_n > 1) Q_LOAD_STAGE(1)
  __syncthreads();
  STAMP(1);
  for (int kt = 0; kt < kt_n; ++kt) {
    // ---- first half: group 0 stores tile kt + 1 and loads tile kt + 2; group 1 multiplies tile kt - 1
    if (wr == 0) {
      if (kt + 1 < kt_n) Q_STORE_STAGE(kt + 1)
      if (kt + 2 < kt_n) Q_LOAD_STAGE(kt + 2)
    } else if (kt > 0) {
      Q_MFMA(kt - 1)
    }
    STAMP(2 + 4 * kt);
    __syncthreads();
    STAMP(3 + 4 * kt);
    // ---- second half: group 0 multiplies tile kt; group 1 stores tile kt + 1 and loads tile kt + 2
    if (wr == 0) {
      Q_MFMA(kt)
    } else {
      if (kt + 1 < kt_n) Q_STORE_STAGE(kt + 1)
      if (kt + 2 < kt_n) Q_LOAD_STAGE(kt + 2)
    }
    STAMP(4 + 4 * kt);
    __syncthreads();
    STAMP(5 + 4 * kt);
  }
  if (wr == 1) Q_MFMA(kt_n - 1)
  __syncthreads();
  write_out(a, lds, acc, tid, row0, col0);
  STAMP(63);
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 5: variant 0's data movement, but INSIDE the MFMA stream: the 8 LDS writes of the staged tile are interleaved with
// the first K-half's 32 MFMAs and the 8 global loads of the tile after next with the second half's (sched_group_barrier), so
// that the wave never sits in a store-only or load-issue-only phase while its SIMD's matrix pipe idles. Steady-state
// iterations are branch-free (the last two are peeled).
template <bool STAMPS>
__global__ __launch_bounds__(NT) void gemm_v5(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const long long row0 = static_cast<long long>(blockIdx.x) * BM;
  const int col0 = static_cast<int>(blockIdx.y) * BN;
  const int kt_n = a.k / BK;
  const int sc = tid & 7, sr = tid >> 3;
  uint4 xr0, xr1, xr2, xr3, wq0, wq1, wq2, wq3;
  STAMP(0);
  // per-thread source pointers (rows clamped once)
  const __hip_bfloat16* xp[4];
  const __hip_bfloat16* wp[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = sr + 64 * i;
    long long row = row0 + r;
    if (row >= a.m) row = a.m - 1;
    xp[i] = a.x + row * a.k + sc * 8;
    wp[i] = a.wt + static_cast<long long>(col0 + r) * a.k + sc * 8;
    soff[i] = r * ROWB + ((sc ^ ((r >> 1) & 7)) << 4);
  }
#define R_LOAD(KT)                                                                                  \
  {                                                                                                 \
    const int ko = (KT) * BK;                                                                       \
    xr0 = *reinterpret_cast<const uint4*>(xp[0] + ko); wq0 = *reinterpret_cast<const uint4*>(wp[0] + ko); \
    xr1 = *reinterpret_cast<const uint4*>(xp[1] + ko); wq1 = *reinterpret_cast<const uint4*>(wp[1] + ko); \
    xr2 = *reinterpret_cast<const uint4*>(xp[2] + ko); wq2 = *reinterpret_cast<const uint4*>(wp[2] + ko); \
    xr3 = *reinterpret_cast<const uint4*>(xp[3] + ko); wq3 = *reinterpret_cast<const uint4*>(wp[3] + ko); \
  }
#define R_STORE(BUF)                                                                                \
  {                                                                                                 \
    unsigned char* base_ = lds + (BUF) * STAGE;                                                     \
    *reinterpret_cast<uint4*>(base_ + soff[0]) = xr0; *reinterpret_cast<uint4*>(base_ + OPB + soff[0]) = wq0; \
    *reinterpret_cast<uint4*>(base_ + soff[1]) = xr1; *reinterpret_cast<uint4*>(base_ + OPB + soff[1]) = wq1; \
    *reinterpret_cast<uint4*>(base_ + soff[2]) = xr2; *reinterpret_cast<uint4*>(base_ + OPB + soff[2]) = wq2; \
    *reinterpret_cast<uint4*>(base_ + soff[3]) = xr3; *reinterpret_cast<uint4*>(base_ + OPB + soff[3]) = wq3; \
  }
  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = OPB + (wc * 64 + (lane & 15)) * ROWB;
#define R_FRAGS(KK)                                                                                                   \
  const int ch##KK = ((KK * 4 + fq) ^ fsw) << 4;                                                                       \
  bf16x8 wf##KK[4], xf##KK[8];                                                                                         \
  _Pragma("unroll") for (int n = 0; n < 4; ++n) wf##KK[n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch##KK); \
  _Pragma("unroll") for (int m = 0; m < 8; ++m) xf##KK[m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch##KK);
#define R_MFMA(KK)                                                                                                     \
  _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                        \
    _Pragma("unroll") for (int m = 0; m < 8; ++m)                                                                      \
      acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf##KK[n], xf##KK[m], acc[n][m], 0, 0, 0);
  R_LOAD(0)
  R_STORE(0)
  if (kt_n > 1) R_LOAD(1)
  __syncthreads();
  STAMP(1);
  int kt = 0;
  for (; kt + 2 < kt_n; ++kt) {   // steady state: store tile kt + 1, load tile kt + 2, multiply tile kt
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    {
      R_FRAGS(0)
      R_STORE((kt + 1) & 1)
      R_MFMA(0)
      // 12 fragment reads first, then 4 MFMAs per LDS write
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    {
      R_FRAGS(1)
      R_LOAD(kt + 2)
      R_MFMA(1)
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __syncthreads();
  }
  for (; kt < kt_n; ++kt) {       // the last two tiles: nothing left to load
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    if (kt + 1 < kt_n) R_STORE((kt + 1) & 1)
    { R_FRAGS(0) R_MFMA(0) }
    { R_FRAGS(1) R_MFMA(1) }
    __syncthreads();
  }
  write_out(a, lds, acc, tid, row0, col0);
  STAMP(63);
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 6: variant 5's schedule with the activations given as BITS (a.x = uint32 [M, k / 32], bit j of word w = element
// 32 w + j, values 0 / 1): a thread fetches one dword per K step (two steps ahead) and expands it into four 16-byte LDS writes;
// the weights go through registers as before. 5 global loads and 8 LDS writes per thread and step, all inside the MFMA stream.
__device__ __forceinline__ uint32_t bits2_bf16(uint32_t b, int p) {
  const uint32_t t = (b >> (2 * p)) & 3u;
  return __umul24((t | (t << 15)) & 0x00010001u, 0x3F80u);
}
template <bool STAMPS>
__global__ __launch_bounds__(NT) void gemm_v6(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const long long row0 = static_cast<long long>(blockIdx.x) * BM;
  const int col0 = static_cast<int>(blockIdx.y) * BN;
  const int kt_n = a.k / BK;
  const int sc = tid & 7, sr = tid >> 3;
  uint4 wq0, wq1, wq2, wq3;
  uint32_t bits_a, bits_b;   // the words of tiles kt + 1 and kt + 2
  STAMP(0);
  const __hip_bfloat16* wp[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = sr + 64 * i;
    wp[i] = a.wt + static_cast<long long>(col0 + r) * a.k + sc * 8;
    soff[i] = r * ROWB + ((sc ^ ((r >> 1) & 7)) << 4);
  }
  const int br = tid >> 1, bh = tid & 1;
  long long brow = row0 + br;
  if (brow >= a.m) brow = a.m - 1;
  const uint32_t* bp = reinterpret_cast<const uint32_t*>(a.x) + brow * (a.k / 32) + bh;
  const int xoff = br * ROWB, xsw = (br >> 1) & 7;
#define S_LOADW(KT)                                                                                 \
  {                                                                                                 \
    const int ko = (KT) * BK;                                                                       \
    wq0 = *reinterpret_cast<const uint4*>(wp[0] + ko); wq1 = *reinterpret_cast<const uint4*>(wp[1] + ko); \
    wq2 = *reinterpret_cast<const uint4*>(wp[2] + ko); wq3 = *reinterpret_cast<const uint4*>(wp[3] + ko); \
  }
#define S_STORE(BUF, BITS)                                                                          \
  {                                                                                                 \
    unsigned char* base_ = lds + (BUF) * STAGE;                                                     \
    *reinterpret_cast<uint4*>(base_ + OPB + soff[0]) = wq0; *reinterpret_cast<uint4*>(base_ + OPB + soff[1]) = wq1; \
    *reinterpret_cast<uint4*>(base_ + OPB + soff[2]) = wq2; *reinterpret_cast<uint4*>(base_ + OPB + soff[3]) = wq3; \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
      const uint32_t v_ = (BITS) >> (8 * j);                                                        \
      *reinterpret_cast<uint4*>(base_ + xoff + (((4 * bh + j) ^ xsw) << 4)) =                       \
          make_uint4(bits2_bf16(v_, 0), bits2_bf16(v_, 1), bits2_bf16(v_, 2), bits2_bf16(v_, 3));   \
    }                                                                                               \
  }
  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = OPB + (wc * 64 + (lane & 15)) * ROWB;
  bits_a = bp[0];
  S_LOADW(0)
  S_STORE(0, bits_a)
  bits_a = kt_n > 1 ? bp[2] : 0u;
  if (kt_n > 1) S_LOADW(1)
  __syncthreads();
  STAMP(1);
  int kt = 0;
  for (; kt + 2 < kt_n; ++kt) {   // steady state: store tile kt + 1 (W regs + bits_a), load tile kt + 2, multiply tile kt
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    {
      R_FRAGS(0)
      bits_b = bp[2 * (kt + 2)];
      S_STORE((kt + 1) & 1, bits_a)
      R_MFMA(0)
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // the expansion's VALU
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    {
      R_FRAGS(1)
      S_LOADW(kt + 2)
      R_MFMA(1)
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    bits_a = bits_b;
    __syncthreads();
  }
  for (; kt < kt_n; ++kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    if (kt + 1 < kt_n) S_STORE((kt + 1) & 1, bits_a)
    { R_FRAGS(0) R_MFMA(0) }
    { R_FRAGS(1) R_MFMA(1) }
    __syncthreads();
  }
  write_out(a, lds, acc, tid, row0, col0);
  STAMP(63);
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 7: FOUR wavefronts per workgroup (one per SIMD), each owning a 128 x 128 quarter of the 256 x 256 tile: 256
// accumulator registers (AGPRs) per lane. Per K step a wavefront reads (128 + 128) x 64 bf16 of fragments for 128 MFMAs,
// i.e. 1.5x the FLOPs per LDS byte of the 128 x 64 wave tile of variants 0-6 (whose LDS reads, 197 KB per K step per
// workgroup, take about as long as its MFMAs). Operands go global -> LDS directly (variant 1's loader, 16 glds per lane per
// K step); the fragments of the next 32-wide K slice are requested before the current slice multiplies.
constexpr int NT7 = 256;
__device__ __forceinline__ void stage_tile7(const Args& a, unsigned char* base, int kt, int wave, int lane, long long row0, int col0) {
  const int lr = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 64 * wave + 8 * i + lr;
    const int c = pc ^ ((r >> 1) & 7);
    long long row = row0 + r;
    if (row >= a.m) row = a.m - 1;
    unsigned char* dst = base + (64 * wave + 8 * i) * ROWB;
    glds16(a.x + row * a.k + kt * BK + c * 8, dst);
    glds16(a.wt + static_cast<long long>(col0 + r) * a.k + kt * BK + c * 8, dst + OPB);
  }
}

template <bool STAMPS>
__global__ __launch_bounds__(NT7) void gemm_v7(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const long long row0 = static_cast<long long>(blockIdx.x) * BM;
  const int col0 = static_cast<int>(blockIdx.y) * BN;
  const int kt_n = a.k / BK;
  STAMP(0);
  f32x4 acc[8][8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = OPB + (wc * 128 + (lane & 15)) * ROWB;
  stage_tile7(a, lds, 0, wave, lane, row0, col0);
  STAMP(1);
  for (int kt = 0; kt < kt_n; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(2 + 4 * kt);
    if (kt + 1 < kt_n) stage_tile7(a, lds + ((kt + 1) & 1) * STAGE, kt + 1, wave, lane, row0, col0);
    STAMP(3 + 4 * kt);
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    bf16x8 wf[2][8], xf[2][8];
    {
      const int ch = ((0 * 4 + fq) ^ fsw) << 4;
#pragma unroll
      for (int n = 0; n < 8; ++n) wf[0][n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch);
#pragma unroll
      for (int m = 0; m < 8; ++m) xf[0][m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk == 0) {
        const int ch = ((1 * 4 + fq) ^ fsw) << 4;
#pragma unroll
        for (int n = 0; n < 8; ++n) wf[1][n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch);
#pragma unroll
        for (int m = 0; m < 8; ++m) xf[1][m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch);
      }
#pragma unroll
      for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][n], xf[kk][m], acc[n][m], 0, 0, 0);
    }
    STAMP(4 + 4 * kt);
  }
  __syncthreads();
  // epilogue: acc[n][m][j] = out[row wr*128 + 16 m + (lane & 15)][col wc*128 + 16 n + 4 (lane >> 4) + j]
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int cl = wc * 128 + n * 16 + fq * 4;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int rl = wr * 128 + m * 16 + (lane & 15);
      *reinterpret_cast<uint2*>(lds + rl * OUT_LD + cl * 2) =
          make_uint2(pack_bf16(acc[n][m][0], acc[n][m][1]), pack_bf16(acc[n][m][2], acc[n][m][3]));
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    const int id = tid + NT7 * i, r = id >> 5, ch = id & 31;
    if (row0 + r < a.m)
      *reinterpret_cast<uint4*>(a.out + (row0 + r) * a.n + col0 + ch * 8) = *reinterpret_cast<const uint4*>(lds + r * OUT_LD + ch * 16);
  }
  STAMP(63);
}

// variant 8: variant 7 with the main loop's issue order pinned (sched_group_barrier): see the comment in the loop
template <bool STAMPS>
__global__ __launch_bounds__(NT7) void gemm_v8(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const long long row0 = static_cast<long long>(blockIdx.x) * BM;
  const int col0 = static_cast<int>(blockIdx.y) * BN;
  const int kt_n = a.k / BK;
  STAMP(0);
  f32x4 acc[8][8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = OPB + (wc * 128 + (lane & 15)) * ROWB;
  stage_tile7(a, lds, 0, wave, lane, row0, col0);
  STAMP(1);
  for (int kt = 0; kt < kt_n; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(2 + 4 * kt);
    // (unconditional: the last iteration re-stages the last tile into the idle buffer, so the loop body stays one basic block
    // and the schedule below applies to all of it)
    stage_tile7(a, lds + ((kt + 1) & 1) * STAGE, kt + 1 < kt_n ? kt + 1 : kt_n - 1, wave, lane, row0, col0);
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    bf16x8 wf[2][8], xf[2][8];
    {
      const int ch = ((0 * 4 + fq) ^ fsw) << 4;
#pragma unroll
      for (int n = 0; n < 8; ++n) wf[0][n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch);
#pragma unroll
      for (int m = 0; m < 8; ++m) xf[0][m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch);
    }
    {
      const int ch = ((1 * 4 + fq) ^ fsw) << 4;
#pragma unroll
      for (int n = 0; n < 8; ++n) wf[1][n] = *reinterpret_cast<const bf16x8*>(cur + w_off + n * 16 * ROWB + ch);
#pragma unroll
      for (int m = 0; m < 8; ++m) xf[1][m] = *reinterpret_cast<const bf16x8*>(cur + x_off + m * 16 * ROWB + ch);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][n], xf[kk][m], acc[n][m], 0, 0, 0);
    // schedule: the 16 direct-to-LDS loads of the next tile, the 16 fragment reads of the first K slice, then the second
    // slice's 16 reads spread one per 4 MFMAs of the first slice, then the second slice's 64 MFMAs
    __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 64, 0);
    STAMP(4 + 4 * kt);
  }
  __syncthreads();
  // epilogue: acc[n][m][j] = out[row wr*128 + 16 m + (lane & 15)][col wc*128 + 16 n + 4 (lane >> 4) + j]
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int cl = wc * 128 + n * 16 + fq * 4;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int rl = wr * 128 + m * 16 + (lane & 15);
      *reinterpret_cast<uint2*>(lds + rl * OUT_LD + cl * 2) =
          make_uint2(pack_bf16(acc[n][m][0], acc[n][m][1]), pack_bf16(acc[n][m][2], acc[n][m][3]));
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    const int id = tid + NT7 * i, r = id >> 5, ch = id & 31;
    if (row0 + r < a.m)
      *reinterpret_cast<uint4*>(a.out + (row0 + r) * a.n + col0 + ch * 8) = *reinterpret_cast<const uint4*>(lds + r * OUT_LD + ch * 16);
  }
  STAMP(63);
}

// ---------------------------------------------------------------------------------------------------------------------
// "thin" GEMM (co-residency study, scripts/thin_probe.py): ONE wavefront per workgroup, a 32 x 32 output tile, operands
// fetched straight from global memory into MFMA fragments (no LDS), few registers — a kernel that fits in what an actor-GEMM
// workgroup leaves free on its CU (48 VGPRs per SIMD, 27 KB of LDS) and can therefore run WHILE that GEMM runs.
//   out[M, N] bf16 = relu?(x[M, K] @ wt[N, K]^T), both operands k-contiguous, K % 32 == 0, M % 32 == 0, N % 32 == 0
__global__ __launch_bounds__(64) void gemm_thin(const Args a) {
  // 32 rows x 16 columns per wavefront: 8 accumulator registers, three operand fragments, 32-bit offsets
  const int lane = threadIdx.x;
  const int lr = lane & 15, kq = lane >> 4;
  const unsigned xo = (static_cast<unsigned>(blockIdx.x) * 32u + lr) * static_cast<unsigned>(a.k) * 2u + kq * 16u;
  const unsigned wo = (static_cast<unsigned>(blockIdx.y) * 16u + lr) * static_cast<unsigned>(a.k) * 2u + kq * 16u;
  const unsigned half = 32u * static_cast<unsigned>(a.k);   // byte distance of the second 16-row tile
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.wt);
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll 1
  for (unsigned kb = 0; kb < static_cast<unsigned>(a.k) * 2u; kb += 64u) {
    const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xb + xo + kb), x1 = *reinterpret_cast<const bf16x8*>(xb + xo + kb + half);
    const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wb + wo + kb);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x1, acc1, 0, 0, 0);
  }
  // acc_m[j] = out[row 32 bx + 16 m + (lane & 15)][col 16 by + 4 (lane >> 4) + j]
  const long long r0 = static_cast<long long>(blockIdx.x) * 32;
  const int c0 = static_cast<int>(blockIdx.y) * 16;
  *reinterpret_cast<uint2*>(a.out + (r0 + lr) * a.n + c0 + 4 * kq) = make_uint2(pack_bf16(acc0[0], acc0[1]), pack_bf16(acc0[2], acc0[3]));
  *reinterpret_cast<uint2*>(a.out + (r0 + 16 + lr) * a.n + c0 + 4 * kq) = make_uint2(pack_bf16(acc1[0], acc1[1]), pack_bf16(acc1[2], acc1[3]));
}

}  // namespace

extern "C" int probe_thin_gemm(const void* x, long long m, int k, const void* wt, int n, void* out, void* stream) {
  if (k % 32 || n % 16 || m % 32) return -1;
  Args a{static_cast<const __hip_bfloat16*>(x), static_cast<const __hip_bfloat16*>(wt), static_cast<__hip_bfloat16*>(out), nullptr, m, k, n};
  hipLaunchKernelGGL(gemm_thin, dim3(static_cast<unsigned>(m / 32), static_cast<unsigned>(n / 16)), dim3(64), 0,
                     static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int probe_gemm(int variant, int stamped, const void* x, long long m, int k, const void* wt, int n, void* out,
                          unsigned long long* stamps, void* stream) {
  if (k % BK || n % BN) return -1;
  Args a{static_cast<const __hip_bfloat16*>(x), static_cast<const __hip_bfloat16*>(wt), static_cast<__hip_bfloat16*>(out), stamps, m, k, n};
  const dim3 grid(static_cast<unsigned>((m + BM - 1) / BM), static_cast<unsigned>(n / BN));
  hipStream_t s = static_cast<hipStream_t>(stream);
#define LAUNCH(KN)                                                                          \
  do {                                                                                      \
    if (stamped) hipLaunchKernelGGL((KN<true>), grid, dim3(NT), 0, s, a);                   \
    else hipLaunchKernelGGL((KN<false>), grid, dim3(NT), 0, s, a);                          \
  } while (0)
  if (variant == 0) LAUNCH(gemm_v0);
  else if (variant == 1) LAUNCH(gemm_v1);
  else if (variant == 3) LAUNCH(gemm_v3);
  else if (variant == 4) LAUNCH(gemm_v4);
  else if (variant == 5) LAUNCH(gemm_v5);
  else if (variant == 6) LAUNCH(gemm_v6);
  else if (variant == 7) {
    if (stamped) hipLaunchKernelGGL((gemm_v7<true>), grid, dim3(NT7), 0, s, a);
    else hipLaunchKernelGGL((gemm_v7<false>), grid, dim3(NT7), 0, s, a);
  }
  else if (variant == 8) {
    if (stamped) hipLaunchKernelGGL((gemm_v8<true>), grid, dim3(NT7), 0, s, a);
    else hipLaunchKernelGGL((gemm_v8<false>), grid, dim3(NT7), 0, s, a);
  }
  else return -2;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
