// policy.hip — fused action selection for the actor: one pass over the C51 logits.
//
// Replaces the tail of DQNPolicy.policy / eval_policy (hanabi_agents/rlax_dqn/rlax_rainbow.py:93-150):
//   probs = softmax(logits, -1); q = mean(probs * atoms, -1)        rlax_rainbow.py:117-118,144-145 (mean, App. C-3)
//   q = where(legal, q, -inf)                                       rlax_rainbow.py:119,147
//   legal epsilon-greedy / greedy sample with uniform tie-breaking  rlax_rainbow.py:34-71,122,150
// A dozen elementwise / scan / reduce launches over the [N, A, K] logits (133 MB at 32 768 games) become one
// HBM-bound kernel that reads the logits once.
//
// Mapping: a wavefront owns GPW = 64 / A consecutive games; it stages their A*K logits through LDS with
// coalesced 4-element loads, then lane (game, action) reduces its K atoms from LDS (row stride K is odd ->
// conflict-free), and the first lane of each game picks the move. Randomness: Philox4x32-10 keyed by
// (seed; draw, global game id): word 0 decides explore-vs-greedy, word 1 picks the k-th candidate, so the
// distribution is exactly (1-eps) * uniform(argmax ties) + eps * uniform(legal) (C-14 without float cumsums).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "../../include/hanabi_hip.h"
#include "common.hpp"
#include <type_traits>

#include "env_kernel.hpp"  // philox4x32_10

using hb::fail;

namespace {

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <>
__device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));  // byte-aligned 16-byte access: one dwordx4 instruction on gfx950


// A wavefront owns gpw = 64 / A consecutive games. Their logits (gpw rows of A*K elements, rows `ld` apart) are
// fetched with fully coalesced 4-element loads — ALL of a lane's loads are issued before the first LDS write, so
// the HBM latency is paid once — and parked in LDS in their 16-bit form (6 KB per wave for 3 x 20 x 51 bf16:
// 6 workgroups per CU). Lane (game, action) then reduces its K atoms from LDS (softmax, expectation).
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void policy_kernel(const T* __restrict__ logits, const int8_t* __restrict__ legal,
                                                     const float* __restrict__ support, long long n, int A, int K, int ld,
                                                     float epsilon, unsigned long long seed, unsigned long long draw,
                                                     long long first_gid, int32_t* __restrict__ actions,
                                                     float* __restrict__ q_out) {
  extern __shared__ unsigned char lds_raw[];
  __shared__ float qlds[4][64];
  typedef typename std::conditional<sizeof(T) == 4, uint4, uint2>::type chunk_t;  // 4 elements
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gpw = 64 / A;
  const int AK = A * K;
  const int per_wave = (gpw * AK + 3) & ~3;
  T* buf = reinterpret_cast<T*>(lds_raw) + wave * per_wave;
  const long long g0 = (static_cast<long long>(blockIdx.x) * 4 + wave) * gpw;
  if (g0 >= n) return;
  const long long left = n - g0;
  const int ng = left < gpw ? static_cast<int>(left) : gpw;
  const int elems = ng * AK;
  const T* src = logits + g0 * ld;
  if ((AK & 3) == 0 && (ld & 3) == 0) {
    chunk_t r[16];  // up to 4096 elements per wavefront
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = (u * 64 + lane) * 4;
      if (e < elems) {
        const int gq = e / AK;
        r[u] = *reinterpret_cast<const chunk_t*>(src + static_cast<long long>(gq) * ld + (e - gq * AK));
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = (u * 64 + lane) * 4;
      if (e < elems) *reinterpret_cast<chunk_t*>(buf + e) = r[u];
    }
  } else {
    for (int e = lane; e < elems; e += 64) {
      const int gq = e / AK;
      buf[e] = src[static_cast<long long>(gq) * ld + (e - gq * AK)];
    }
  }
  hb::wave_sync();
  const int gw = lane / A, a = lane - gw * A;
  const bool on = gw < ng;
  float q = -INFINITY;
  bool is_legal = false;
  if (on) {
    const T* row = buf + (gw * A + a) * K;
    float v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = k < K ? to_f32<T>(row[k]) : -INFINITY;
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) m = fmaxf(m, v[k]);
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        const float e = __expf(v[k] - m);
        s += e;
        t += e * support[k];
      }
    const float qv = t / s / static_cast<float>(K);
    if (q_out) q_out[(g0 + gw) * A + a] = qv;
    is_legal = legal[(g0 + gw) * A + a] != 0;
    q = is_legal ? qv : -INFINITY;
  }
  float* qbuf = qlds[wave];
  qbuf[lane] = q;
  hb::wave_sync();
  // per-game arg-max and candidate sets without touching memory again: every lane scans its game's A values
  // in LDS, wave ballots give the legal / tie bit sets, the game's first lane shifts its A bits out
  float best = -INFINITY;
  if (on)
    for (int i = 0; i < A; ++i) best = fmaxf(best, qbuf[gw * A + i]);
  const unsigned long long legal_wave = __ballot(on && is_legal);
  const unsigned long long ties_wave = __ballot(on && is_legal && q == best);
  if (on && a == 0) {
    const unsigned long long mask_a = (1ull << A) - 1ull;
    const unsigned long long legal_mask = (legal_wave >> (gw * A)) & mask_a;
    const unsigned long long ties = (ties_wave >> (gw * A)) & mask_a;
    const unsigned long long gid = static_cast<unsigned long long>(first_gid + g0 + gw);
    uint32_t r[4];
    hb::philox4x32_10(static_cast<uint32_t>(draw), static_cast<uint32_t>(draw >> 32), static_cast<uint32_t>(gid),
                      static_cast<uint32_t>(gid >> 32), static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), r);
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
    actions[g0 + gw] = pick;
  }
}

template <typename T>
int launch_policy(const void* logits, const int8_t* legal, const float* support, int64_t n, int A, int K, int ld, float eps,
                  uint64_t seed, uint64_t draw, int64_t first_gid, int32_t* actions, float* q_out, hipStream_t s) {
  const int gpw = 64 / A;
  const long long per_block = 4LL * gpw;
  const unsigned blocks = static_cast<unsigned>((n + per_block - 1) / per_block);
  const size_t lds = 4 * static_cast<size_t>((gpw * A * K + 3) & ~3) * sizeof(T);
  hipLaunchKernelGGL((policy_kernel<T, 64>), dim3(blocks), dim3(256), lds, s, static_cast<const T*>(logits), legal, support,
                     static_cast<long long>(n), A, K, ld, eps, static_cast<unsigned long long>(seed),
                     static_cast<unsigned long long>(draw), static_cast<long long>(first_gid), actions, q_out);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

// ---------------------------------------------------------------------------------------------
// replay insert: the acting seat's N transitions -> ring slots (start + i) mod cap, one launch
// ---------------------------------------------------------------------------------------------
// copies `bytes` (16 bytes per lane, tail bytewise) to up to two destinations; rows are obs_len bytes apart, so
// neither side is more than byte-aligned after a ring wrap: unaligned dwordx4 accesses (one instruction on gfx950)
__device__ __forceinline__ void copy2(const int8_t* __restrict__ src, int8_t* __restrict__ d0, int8_t* __restrict__ d1,
                                      long long bytes, long long tid, long long nthreads) {
  const long long vec = bytes >> 4;
  for (long long i = tid; i < vec; i += nthreads) {
    const u32x4 v = *reinterpret_cast<const u32x4_u*>(src + (i << 4));
    if (d0) *reinterpret_cast<u32x4_u*>(d0 + (i << 4)) = v;
    if (d1) *reinterpret_cast<u32x4_u*>(d1 + (i << 4)) = v;
  }
  for (long long b = (vec << 4) + tid; b < bytes; b += nthreads) {
    const int8_t v = src[b];
    if (d0) d0[b] = v;
    if (d1) d1[b] = v;
  }
}

struct InsertArgs {
  int8_t* last_obs;          // [n, L] in/out
  const int8_t* obs;         // [n, L]
  const int8_t* legal;       // [n, A]
  const int32_t* actions;    // [n]
  const float* rewards;      // [n]
  const int8_t* step_type;   // [n]
  int8_t* ring_obs_tm1;      // [cap, L]
  int8_t* ring_obs_t;        // [cap, L]
  int8_t* ring_act;          // [cap]
  int8_t* ring_lms;          // [cap, A]
  float* ring_rew;           // [cap]
  uint8_t* ring_term;        // [cap] (bool)
  long long n, cap, start;
  int L, A;
};

// One pass over the batch: every 16-byte chunk of last_obs / obs is read once and written to its three
// destinations by the same thread (ring.obs_tm1 <- last_obs, ring.obs_t <- obs, last_obs <- obs), so the
// read-before-overwrite of last_obs needs no second launch. The ring range may wrap: segment 0 = rows before the
// wrap, segment 1 = the rest.
__global__ __launch_bounds__(256) void replay_insert_kernel(const InsertArgs a) {
  const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long nthreads = static_cast<long long>(gridDim.x) * blockDim.x;
  const long long first = a.cap - a.start < a.n ? a.cap - a.start : a.n;  // rows before the wrap
  for (int seg = 0; seg < 2; ++seg) {
    const long long row0 = seg == 0 ? 0 : first;
    const long long rows = seg == 0 ? first : a.n - first;
    const long long slot0 = seg == 0 ? a.start : 0;
    if (rows <= 0) continue;
    const long long bytes = rows * a.L, vec = bytes >> 4;
    int8_t* lo = a.last_obs + row0 * a.L;
    const int8_t* ob = a.obs + row0 * a.L;
    int8_t* r1 = a.ring_obs_tm1 + slot0 * a.L;
    int8_t* r2 = a.ring_obs_t + slot0 * a.L;
    for (long long i = tid; i < vec; i += nthreads) {
      const u32x4 old = *reinterpret_cast<const u32x4_u*>(lo + (i << 4));
      const u32x4 cur = *reinterpret_cast<const u32x4_u*>(ob + (i << 4));
      *reinterpret_cast<u32x4_u*>(r1 + (i << 4)) = old;
      *reinterpret_cast<u32x4_u*>(r2 + (i << 4)) = cur;
      *reinterpret_cast<u32x4_u*>(lo + (i << 4)) = cur;
    }
    for (long long b = (vec << 4) + tid; b < bytes; b += nthreads) {
      const int8_t old = lo[b], cur = ob[b];
      r1[b] = old;
      r2[b] = cur;
      lo[b] = cur;
    }
    copy2(a.legal + row0 * a.A, a.ring_lms + slot0 * a.A, nullptr, rows * a.A, tid, nthreads);
    for (long long i = tid; i < rows; i += nthreads) {
      a.ring_act[slot0 + i] = static_cast<int8_t>(a.actions[row0 + i]);
      a.ring_rew[slot0 + i] = a.rewards[row0 + i];
      a.ring_term[slot0 + i] = a.step_type[row0 + i] == 2;
    }
  }
}

// int8 0/1 observations [rows, cols] -> bf16 / f16 GEMM operand with row stride out_ld >= cols (padding columns are
// never written: the caller zeroes them once). Lane -> (row, 16-column chunk): one unaligned 16-byte load, two
// 16-byte stores; the cols % 16 tail is done element-wise by the same lanes.
template <typename T>
__global__ __launch_bounds__(256) void obs_cast_kernel(const int8_t* __restrict__ in, T* __restrict__ out, long long rows,
                                                       int cols, int out_ld) {
  const int cpr = cols >> 4, tail = cols & 15;
  const long long chunks = rows * cpr;
  const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long nthreads = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long c = tid; c < chunks; c += nthreads) {
    const long long r = c / cpr;
    const int l = static_cast<int>(c - r * cpr);
    const u32x4 raw = *reinterpret_cast<const u32x4_u*>(in + r * cols + 16 * l);
    const int8_t* b = reinterpret_cast<const int8_t*>(&raw);
    T o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = static_cast<T>(static_cast<float>(b[j]));
    u32x4_u* dst = reinterpret_cast<u32x4_u*>(out + r * out_ld + 16 * l);
    dst[0] = *reinterpret_cast<const u32x4*>(o);
    dst[1] = *reinterpret_cast<const u32x4*>(o + 8);
  }
  if (tail)
    for (long long e = tid; e < rows * tail; e += nthreads) {
      const long long r = e / tail;
      const int j = cpr * 16 + static_cast<int>(e - r * tail);
      out[r * out_ld + j] = static_cast<T>(static_cast<float>(in[r * cols + j]));
    }
}

// ---- bit-packed observations <-> the reference's int8 0/1 layout -------------------------------------------------------
// packed row = ceil(L / 32) u32 words, observation bit i = word i >> 5, bit i & 31; pad bits are zero.
// pack: a wavefront takes 64 consecutive bytes of one row (coalesced 1-byte loads) and ballots them into two words.
__global__ __launch_bounds__(256) void obs_pack_kernel(const int8_t* __restrict__ obs, uint32_t* __restrict__ bits, long long rows,
                                                       int L, int W) {
  const int lane = threadIdx.x & 63;
  const int cpr = (L + 63) >> 6;                                            // 64-byte chunks per row
  const long long chunk = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (chunk >= rows * cpr) return;
  const long long r = chunk / cpr;
  const int c = static_cast<int>(chunk - r * cpr), col = c * 64 + lane;
  const unsigned long long m = __ballot(col < L && obs[r * L + col] != 0);
  if (lane == 0) {
    bits[r * W + 2 * c] = static_cast<uint32_t>(m);
    if (2 * c + 1 < W) bits[r * W + 2 * c + 1] = static_cast<uint32_t>(m >> 32);
  }
}
// unpack: thread -> (row, 16-column chunk): 16 bits -> one byte-aligned 16-byte store (4 bits -> 4 bytes per multiply,
// the expansion of hb::expand_rows in the env kernel); the L % 16 tail bytewise
__global__ __launch_bounds__(256) void obs_unpack_kernel(const uint32_t* __restrict__ bits, int8_t* __restrict__ obs, long long rows,
                                                         int L, int W) {
  const int cpr = (L + 15) >> 4;
  const long long id = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (id >= rows * cpr) return;
  const long long r = id / cpr;
  const int c = static_cast<int>(id - r * cpr);
  const uint32_t v = (bits[r * W + (c >> 1)] >> ((c & 1) * 16)) & 0xFFFFu;
  int8_t* dst = obs + r * L + 16 * c;
  if (16 * c + 16 <= L) {
    u32x4 o;
    o.x = hb::spread4(v);
    o.y = hb::spread4(v >> 4);
    o.z = hb::spread4(v >> 8);
    o.w = hb::spread4(v >> 12);
    *reinterpret_cast<u32x4_u*>(dst) = o;
  } else {
    for (int j = 0; 16 * c + j < L; ++j) dst[j] = static_cast<int8_t>((v >> j) & 1u);
  }
}

}  // namespace

extern "C" {

int hb_obs_pack(const int8_t* obs_dev, uint32_t* bits_dev, int64_t rows, int32_t obs_len, void* stream) {
  if (!obs_dev || !bits_dev) return fail(HB_ERR_INVALID, "null argument");
  if (rows <= 0) return HB_OK;
  if (obs_len < 1) return fail(HB_ERR_INVALID, "obs_len must be positive");
  const int W = (obs_len + 31) / 32;
  const long long waves = rows * ((obs_len + 63) / 64);
  hipLaunchKernelGGL(obs_pack_kernel, dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     obs_dev, bits_dev, static_cast<long long>(rows), obs_len, W);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_obs_unpack(const uint32_t* bits_dev, int8_t* obs_dev, int64_t rows, int32_t obs_len, void* stream) {
  if (!obs_dev || !bits_dev) return fail(HB_ERR_INVALID, "null argument");
  if (rows <= 0) return HB_OK;
  if (obs_len < 1) return fail(HB_ERR_INVALID, "obs_len must be positive");
  const int W = (obs_len + 31) / 32;
  const long long threads = rows * ((obs_len + 15) / 16);
  hipLaunchKernelGGL(obs_unpack_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), bits_dev, obs_dev, static_cast<long long>(rows), obs_len, W);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_obs_cast(const int8_t* obs_dev, void* out_dev, int32_t out_dtype, int64_t rows, int32_t cols, int32_t out_ld,
                void* stream) {
  if (!obs_dev || !out_dev) return fail(HB_ERR_INVALID, "null argument");
  if (rows <= 0 || cols <= 0) return HB_OK;
  if (out_ld < cols) return fail(HB_ERR_INVALID, "out_ld must be >= cols");
  long long blocks = ((rows * (cols >> 4)) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (out_dtype == 1)
    hipLaunchKernelGGL((obs_cast_kernel<__hip_bfloat16>), dim3(blocks), dim3(256), 0, s, obs_dev, static_cast<__hip_bfloat16*>(out_dev), static_cast<long long>(rows), cols, out_ld);
  else if (out_dtype == 2)
    hipLaunchKernelGGL((obs_cast_kernel<__half>), dim3(blocks), dim3(256), 0, s, obs_dev, static_cast<__half*>(out_dev), static_cast<long long>(rows), cols, out_ld);
  else
    return fail(HB_ERR_INVALID, "out_dtype must be 1 (bf16) or 2 (f16)");
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_policy_act(const void* logits_dev, int32_t logits_dtype, const int8_t* legal_dev, const float* support_dev,
                  int64_t n_games, int32_t n_actions, int32_t n_atoms, int32_t row_stride, float epsilon, uint64_t seed,
                  uint64_t draw, int64_t first_game_id, int32_t* actions_dev, float* q_dev, void* stream) {
  if (!logits_dev || !legal_dev || !support_dev || !actions_dev) return fail(HB_ERR_INVALID, "null argument");
  if (n_actions < 1 || n_actions > 64) return fail(HB_ERR_INVALID, "n_actions must be 1..64");
  if (n_atoms < 1 || n_atoms > 64) return fail(HB_ERR_INVALID, "n_atoms must be 1..64 (atoms are held in registers)");
  if ((64 / n_actions) * n_actions * n_atoms > 4096) return fail(HB_ERR_INVALID, "n_actions * n_atoms too large for the staging buffer");
  if (reinterpret_cast<uintptr_t>(logits_dev) & 15) return fail(HB_ERR_ALIGN, "logits_dev must be 16-byte aligned");
  if (n_games <= 0) return HB_OK;
  if (row_stride < n_actions * n_atoms) return fail(HB_ERR_INVALID, "row_stride must be >= n_actions * n_atoms");
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (logits_dtype) {
    case 0: return launch_policy<float>(logits_dev, legal_dev, support_dev, n_games, n_actions, n_atoms, row_stride, epsilon, seed, draw, first_game_id, actions_dev, q_dev, s);
    case 1: return launch_policy<__hip_bfloat16>(logits_dev, legal_dev, support_dev, n_games, n_actions, n_atoms, row_stride, epsilon, seed, draw, first_game_id, actions_dev, q_dev, s);
    case 2: return launch_policy<__half>(logits_dev, legal_dev, support_dev, n_games, n_actions, n_atoms, row_stride, epsilon, seed, draw, first_game_id, actions_dev, q_dev, s);
  }
  return fail(HB_ERR_INVALID, "logits_dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
}

int hb_replay_insert(int8_t* last_obs_dev, const int8_t* obs_dev, const int8_t* legal_dev, const int32_t* actions_dev,
                     const float* rewards_dev, const int8_t* step_type_dev, int8_t* ring_obs_tm1_dev,
                     int8_t* ring_obs_t_dev, int8_t* ring_act_dev, int8_t* ring_lms_dev, float* ring_rew_dev,
                     uint8_t* ring_term_dev, int64_t n, int32_t obs_len, int32_t n_actions, int64_t capacity,
                     int64_t start, void* stream) {
  if (!last_obs_dev || !obs_dev || !legal_dev || !actions_dev || !rewards_dev || !step_type_dev || !ring_obs_tm1_dev ||
      !ring_obs_t_dev || !ring_act_dev || !ring_lms_dev || !ring_rew_dev || !ring_term_dev)
    return fail(HB_ERR_INVALID, "null argument");
  if (n <= 0) return HB_OK;
  if (n > capacity || start < 0 || start >= capacity) return fail(HB_ERR_INVALID, "bad ring range");
  InsertArgs a{last_obs_dev, obs_dev, legal_dev, actions_dev, rewards_dev, step_type_dev, ring_obs_tm1_dev, ring_obs_t_dev,
               ring_act_dev, ring_lms_dev, ring_rew_dev, ring_term_dev, n, capacity, start, obs_len, n_actions};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long vec = (n * obs_len) >> 4;
  unsigned blocks = static_cast<unsigned>((vec + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(replay_insert_kernel, dim3(blocks), dim3(256), 0, s, a);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

}  // extern "C"
