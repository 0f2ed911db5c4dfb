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
  uint4 xr[4], wq[4];
  STAMP(0);
  auto load_stage = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 64 * i;
      long long row = row0 + r;
      if (row >= a.m) row = a.m - 1;
      const int k = kt * BK + sc * 8;
      xr[i] = *reinterpret_cast<const uint4*>(a.x + row * a.k + k);
      wq[i] = *reinterpret_cast<const uint4*>(a.wt + static_cast<long long>(col0 + r) * a.k + k);
    }
  };
  auto store_stage = [&](int buf) {
    unsigned char* base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 64 * i;
      const int off = r * ROWB + ((sc ^ ((r >> 1) & 7)) << 4);
      *reinterpret_cast<uint4*>(base + off) = xr[i];
      *reinterpret_cast<uint4*>(base + OPB + off) = wq[i];
    }
  };
  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane & 15) >> 1, fq = lane >> 4;
  const int x_off = (wr * 128 + (lane & 15)) * ROWB;
  const int w_off = OPB + (wc * 64 + (lane & 15)) * ROWB;
  load_stage(0);
  store_stage(0);
  if (kt_n > 1) load_stage(1);
  __syncthreads();
  STAMP(1);
  for (int kt = 0; kt < kt_n; ++kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    if (kt + 1 < kt_n) store_stage((kt + 1) & 1);
    STAMP(2 + 4 * kt);
    if (kt + 2 < kt_n) load_stage(kt + 2);
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

}  // namespace

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
  else return -2;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
