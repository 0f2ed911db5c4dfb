// sum_tree.hip — GPU-resident sum tree for prioritized replay on gfx950.
//
// Replaces sum_tree.SumTreef, the reference's only native component
// (sum_tree/sum_tree/include/sum_tree.h:22-130, bound in sum_tree/sum_tree/src/sum_tree_py.cc:9-22;
// callers hanabi_agents/rlax_dqn/priority_buffer.py:17,30-32,41-42,52).
//
// Layout: one flat fp32 heap in HBM, node 1 = root, children of n = 2n, 2n+1, leaves =
// [cap, 2cap). cap = 2^19 -> 4 MiB, resident in one XCD's L2 / the Infinity Cache. Instead of
// the reference's pointer-linked nodes with a mutex each and a running float difference per
// ancestor (sum_tree.h:80-90), every internal node is always exactly fl(left + right), so the
// result does not depend on update order.
//
// Kernels (all latency-bound: a handful of dependent L2 round trips):
//   update_chunks  n <= 1024 on a deep tree (the learner's B=256 priority update at cap 2^19): one workgroup per
//                  touched 1024-leaf subtree applies its entries and rebuilds it through LDS; then rebuild_top.
//   update_small   n <= 1024 on a shallow tree: ONE workgroup de-duplicates
//                  (last occurrence wins, the sequential order of sum_tree.h:38-44), writes the
//                  leaves and re-sums their ancestors level by level. Optionally fuses
//                  p = (|td|+1e-10)^alpha and the running max/min (priority_buffer.py:48-52).
//   fill_chunks    ring insert of a contiguous (wrapping) leaf range at one value
//                  (priority_buffer.py:29-32): each workgroup owns an aligned 1024-leaf subtree,
//                  rewrites its leaves and rebuilds its 10 levels through LDS.
//   rebuild_top    one workgroup re-sums the levels above the 1024-leaf subtrees.
//   sample         one WAVEFRONT per query: each round the 63 left-children of a 6-level
//                  subtree are fetched by 63 lanes in one load and the 6 binary decisions
//                  (strict <, subtract-on-right: sum_tree.h:92-105) are replayed from registers
//                  with v_readlane, so cap = 2^19 costs 4 dependent loads instead of 19.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <new>

#include "../../include/hanabi_hip.h"
#include "common.hpp"
#include "env_kernel.hpp"  // hb::philox4x32_10

using hb::fail;

struct hb_tree {
  long long cap;
  int depth;          // log2(cap)
  int chunk;          // leaves per rebuild workgroup = min(cap, 1024)
  float* nodes;       // 2*cap
  unsigned* stamp;    // cap, lazily allocated (large updates with possible duplicates)
  unsigned long long* errors;
  float* scratch;     // large per_update: transformed priorities
  long long scratch_n;
  bool single_wg;     // measurements only (env HB_TREE_UPDATE_PATH=single at creation): force update_small
  bool fused_pow;     // the transform inside update_chunks (one launch, 72 registers); env HB_TREE_UPDATE_PATH=two at creation:
                      // per_transform + update_chunks<false> (44 / 22 registers: both fit beside the one-kernel actor; same step time)
  float* small_scratch;   // SMALL_MAX transformed priorities of a small hb_per_update
  bool lazy_top;      // hb_tree_set_lazy_top: writers leave the levels above the 1024-leaf subtrees stale, readers re-sum them
  bool top_stale;     // lazy mode was switched off: the stored top levels are made valid by the next reader or writer, on ITS stream
};

namespace {

constexpr int SMALL_MAX = 1024;

// ---------------------------------------------------------------------------------------------
// update_small: <<<1, 1024>>>
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void update_small_kernel(float* __restrict__ nodes, long long cap, int depth,
                                                           const int64_t* __restrict__ idx, const float* __restrict__ val,
                                                           int n, int per_mode, double alpha, float* max_prio,
                                                           float* min_prio, unsigned long long* errors) {
  __shared__ __attribute__((aligned(16))) int s_idx[SMALL_MAX + 4];  // leaf indices fit 32 bits (cap <= 2^23)
  __shared__ float s_red[2][16];
  const int i = threadIdx.x;
  long long my = -1;
  float v = 0.f;
  if (i < n) {
    my = idx[i];
    v = val[i];
    if (per_mode) {
      // (priorities + 1e-10) ** alpha on float32 data (priority_buffer.py:49): float add, the power in
      // double rounded once to float (numpy powf may differ from this by 1 ulp)
      const float x = fabsf(v) + 1e-10f;
      v = static_cast<float>(pow(static_cast<double>(x), alpha));
    }
    if (my < 0 || my >= cap) {
      atomicAdd(errors, 1ull);
      my = -1;
    }
  }
  s_idx[i] = static_cast<int>(my);
  if (per_mode) {  // running max / min over ALL transformed priorities (priority_buffer.py:50-51)
    float mx = i < n ? v : -INFINITY, mn = i < n ? v : INFINITY;
    for (int o = 32; o > 0; o >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o));
      mn = fminf(mn, __shfl_xor(mn, o));
    }
    if ((i & 63) == 0) { s_red[0][i >> 6] = mx; s_red[1][i >> 6] = mn; }
  }
  __syncthreads();
  if (per_mode && i == 0) {
    float mx = -INFINITY, mn = INFINITY;
    for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) { mx = fmaxf(mx, s_red[0][w]); mn = fminf(mn, s_red[1][w]); }
    if (max_prio && mx > *max_prio) *max_prio = mx;
    if (min_prio && mn < *min_prio) *min_prio = mn;
  }
  // last occurrence of an index wins: branch-free scan, four indices per LDS read (a data-dependent early exit
  // serialises the reads: 13 us at n = 256; this form is ~1 us)
  bool winner = my >= 0;
  {
    const int me = static_cast<int>(my);
    const int4* v = reinterpret_cast<const int4*>(s_idx);
    for (int j = (i + 1) & ~3; j < n; j += 4) {
      const int4 q = v[j >> 2];
      const bool hit = (j > i && q.x == me) || (j + 1 > i && j + 1 < n && q.y == me) ||
                       (j + 2 > i && j + 2 < n && q.z == me) || (j + 3 > i && j + 3 < n && q.w == me);
      winner = winner && !hit;
    }
  }
  if (winner) nodes[cap + my] = v;
  __syncthreads();
  // re-sum the ancestors bottom-up; threads sharing an ancestor write the same value
  long long node = my >= 0 ? (cap + my) : 0;
  for (int d = 0; d < depth; ++d) {
    node >>= 1;
    if (my >= 0) nodes[node] = nodes[2 * node] + nodes[2 * node + 1];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// chunk rebuild: workgroup b owns the aligned subtree of `chunk` leaves number chunk_id(b)
// ---------------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(256) void chunk_kernel(float* __restrict__ nodes, long long cap, int chunk, int first_chunk,
                                                    long long start, long long n, const float* __restrict__ value_dev) {
  __shared__ float s[2048];  // heap of the subtree: s[1] root, leaves s[chunk .. 2chunk)
  const long long nchunks = cap / chunk;
  const long long c = (first_chunk + static_cast<long long>(blockIdx.x)) % nchunks;
  const long long leaf0 = c * chunk;
  float fillv = 0.f;
  if (FILL) fillv = *value_dev;
  for (int j = threadIdx.x; j < chunk; j += 256) {
    const long long leaf = leaf0 + j;
    float v;
    if (FILL) {
      long long rel = leaf - start;
      if (rel < 0) rel += cap;
      const bool in = rel < n;
      v = in ? fillv : nodes[cap + leaf];
      if (in) nodes[cap + leaf] = v;
    } else {
      v = nodes[cap + leaf];
    }
    s[chunk + j] = v;
  }
  __syncthreads();
  // subtree root is global node (cap + leaf0) / chunk; local node m at level width w maps to global base + offset
  for (int w = chunk >> 1; w >= 1; w >>= 1) {
    for (int j = threadIdx.x; j < w; j += 256) {
      const float v = s[2 * (w + j)] + s[2 * (w + j) + 1];
      s[w + j] = v;
      nodes[(cap + leaf0) / (chunk / w) + j] = v;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// update_chunks: small update (n <= 1024) of a deep tree. Workgroup i looks at entry i: if no earlier entry falls
// into the same 1024-leaf subtree it OWNS that subtree — applies every entry that falls into it (last occurrence
// wins) and rebuilds the subtree's 10 levels through LDS, one load round and one store round instead of a dependent
// HBM round trip per level. rebuild_top then re-sums the levels above. Workgroup n (the extra one) keeps the
// running max / min priority and counts out-of-range indices. Result == update_small (every inner node is
// fl(left + right) of its children either way).
// ---------------------------------------------------------------------------------------------
template <bool PER>   // PER: the priority transform (a double-precision pow) runs in here; false: `val` is written as it is
__global__ __launch_bounds__(256) void update_chunks_kernel(float* __restrict__ nodes, long long cap, int chunk,
                                                            const int64_t* __restrict__ idx, const float* __restrict__ val,
                                                            int n, double alpha, float* max_prio,
                                                            float* min_prio, unsigned long long* errors) {
  constexpr int per_mode = PER ? 1 : 0;
  // 20 480 bytes of LDS in all (s, s_idx, s_chunk, s_stamp; the statistics workgroup's scratch aliases s): what the one-kernel
  // actor leaves free on a CU (csrc/actor_fused.hip holds 143 360 B and 2 x 232 registers per SIMD lane: 20 480 B and 48
  // registers are left). With the pow in here the kernel needs 72 registers; PER = false needs 22 and, behind per_transform_kernel
  // (44), runs BESIDE the next policy kernel (HB_TREE_UPDATE_PATH=two). Measured in the benched loop: the same step time either
  // way (0.112-0.114 ms), so the one-launch form stays the default (DESIGN section 5c "Co-residency").
  __shared__ float s[2048];
  __shared__ __attribute__((aligned(16))) int s_idx[SMALL_MAX];
  float (*s_red)[4] = reinterpret_cast<float (*)[4]>(s);
  const int tid = threadIdx.x;
  auto transformed = [&](float v) {
    // (priorities + 1e-10) ** alpha on float32 data (priority_buffer.py:49): float add, the power in double
    // rounded once to float
    return per_mode ? static_cast<float>(pow(static_cast<double>(fabsf(v) + 1e-10f), alpha)) : v;
  };
  if (static_cast<int>(blockIdx.x) == n) {
    float mx = -INFINITY, mn = INFINITY;
    int bad = 0;
    for (int j = tid; j < n; j += 256) {
      const long long k = idx[j];
      if (k < 0 || k >= cap) ++bad;
      if (per_mode) {  // over ALL transformed priorities, also those of ignored indices (priority_buffer.py:50-51)
        const float p = transformed(val[j]);
        mx = fmaxf(mx, p);
        mn = fminf(mn, p);
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o));
      mn = fminf(mn, __shfl_xor(mn, o));
      bad += __shfl_xor(bad, o);
    }
    if ((tid & 63) == 0) {
      s_red[0][tid >> 6] = mx;
      s_red[1][tid >> 6] = mn;
      if (bad) atomicAdd(errors, static_cast<unsigned long long>(bad));
    }
    __syncthreads();
    if (per_mode && tid == 0) {
      for (int w = 1; w < 4; ++w) { mx = fmaxf(mx, s_red[0][w]); mn = fminf(mn, s_red[1][w]); }
      if (max_prio && mx > *max_prio) *max_prio = mx;
      if (min_prio && mn < *min_prio) *min_prio = mn;
    }
    return;
  }
  // one round trip for everything this workgroup may need: all indices (as chunk-local ids later) and values
  __shared__ int s_chunk[SMALL_MAX];
  __shared__ int s_stamp[1024];
  long long my_idx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = tid + 256 * r;
    my_idx[r] = j < n ? idx[j] : -1;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = tid + 256 * r;
    const bool ok = my_idx[r] >= 0 && my_idx[r] < cap;
    if (j < n) {
      s_chunk[j] = ok ? static_cast<int>(my_idx[r] / chunk) : -1;
      s_idx[j] = ok ? static_cast<int>(my_idx[r] % chunk) : -1;
    }
  }
  for (int j = tid; j < chunk; j += 256) s_stamp[j] = 0;
  if (tid == 0) reinterpret_cast<volatile int*>(s)[0] = 0;
  __syncthreads();
  const int my_chunk = __builtin_amdgcn_readfirstlane(s_chunk[blockIdx.x]);   // workgroup-uniform: scalar registers
  if (my_chunk < 0) return;
  // do I own this subtree (no earlier entry falls into it)? and which entry wrote each of its leaves last?
  int earlier = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = tid + 256 * r;
    if (j < n && s_chunk[j] == my_chunk) {
      if (j < static_cast<int>(blockIdx.x)) earlier = 1;
      atomicMax(&s_stamp[s_idx[j]], j + 1);  // last occurrence wins (the sequential order of sum_tree.h:38-44)
    }
  }
  const long long leaf0 = static_cast<long long>(my_chunk) * chunk;
  float keep[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) keep[r] = tid + 256 * r < chunk ? nodes[cap + leaf0 + tid + 256 * r] : 0.f;
  // (a flag in LDS instead of __syncthreads_or, which reserves LDS of its own; word 0 of s is never a tree node)
  if (earlier) reinterpret_cast<volatile int*>(s)[0] = 1;
  __syncthreads();
  if (reinterpret_cast<volatile int*>(s)[0]) return;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (tid + 256 * r < chunk) s[chunk + tid + 256 * r] = keep[r];
  __syncthreads();
#pragma unroll 1   // (one instance of the double-precision pow: the kernel stays within 56 registers, see the note above)
  for (int j = tid; j < n; j += 256) {
    if (s_chunk[j] == my_chunk && s_stamp[s_idx[j]] == j + 1) {
      const float v = transformed(val[j]);
      s[chunk + s_idx[j]] = v;
      nodes[cap + leaf0 + s_idx[j]] = v;
    }
  }
  __syncthreads();
  for (int w = chunk >> 1; w >= 1; w >>= 1) {
    for (int j = tid; j < w; j += 256) {
      const float v = s[2 * (w + j)] + s[2 * (w + j) + 1];
      s[w + j] = v;
      nodes[(cap + leaf0) / (chunk / w) + j] = v;
    }
    __syncthreads();
  }
}

// levels above the chunk roots: <<<1, 1024>>>, ntop = cap / chunk chunk roots living at nodes [ntop, 2 ntop)
__global__ __launch_bounds__(1024) void rebuild_top_kernel(float* __restrict__ nodes, int ntop) {
  extern __shared__ float s[];  // 2 * ntop
  for (int j = threadIdx.x; j < ntop; j += blockDim.x) s[ntop + j] = nodes[ntop + j];
  __syncthreads();
  for (int w = ntop >> 1; w >= 1; w >>= 1) {
    for (int j = threadIdx.x; j < w; j += blockDim.x) {
      const float v = s[2 * (w + j)] + s[2 * (w + j) + 1];
      s[w + j] = v;
      nodes[w + j] = v;
    }
    __syncthreads();
  }
}

// large arbitrary updates: last-occurrence-wins through a stamp array
__global__ void stamp_kernel(unsigned* __restrict__ stamp, long long cap, const int64_t* __restrict__ idx, long long n,
                             unsigned long long* errors) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long k = idx[i];
  if (k < 0 || k >= cap) { atomicAdd(errors, 1ull); return; }
  atomicMax(&stamp[k], static_cast<unsigned>(i + 1));
}
__global__ void scatter_kernel(float* __restrict__ nodes, unsigned* __restrict__ stamp, long long cap,
                               const int64_t* __restrict__ idx, const float* __restrict__ val, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long k = idx[i];
  if (k < 0 || k >= cap) return;
  if (stamp[k] == static_cast<unsigned>(i + 1)) {
    nodes[cap + k] = val[i];
    stamp[k] = 0;
  }
}
__global__ void per_transform_kernel(const float* __restrict__ td, float* __restrict__ out, long long n, double alpha,
                                     float* max_prio, float* min_prio) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p = static_cast<float>(pow(static_cast<double>(fabsf(td[i]) + 1e-10f), alpha));
  out[i] = p;
  // p > 0, so the IEEE bit patterns order like the values
  if (max_prio) atomicMax(reinterpret_cast<unsigned*>(max_prio), __float_as_uint(p));
  if (min_prio) atomicMin(reinterpret_cast<unsigned*>(min_prio), __float_as_uint(p));
}

// ---------------------------------------------------------------------------------------------
// sample: one wavefront per query
// ---------------------------------------------------------------------------------------------
// `top` (optional): an LDS copy of nodes [0, top_n) — the levels above the 1024-leaf subtrees, re-summed by the caller when
// the tree keeps them lazily (hb_tree_set_lazy_top); nodes at or beyond top_n are read from memory
__device__ __forceinline__ long long wave_descend(const float* __restrict__ nodes, long long cap, int depth, float query,
                                                  int lane, float* leaf_val, const float* top = nullptr, int top_n = 0) {
  long long node = 1;
  int remaining = depth;
  while (remaining > 0) {
    const int k = remaining < 6 ? remaining : 6;
    // lane t (1 <= t < 2^k) stands for the subtree node with heap index t (root = 1) and fetches the value
    // of that node's LEFT child: exactly the operand of the comparison made at that node.
    float left = 0.f;
    if (lane >= 1 && lane < (1 << k)) {
      const int d = 31 - __clz(lane);
      const long long parent = (node << d) + (lane - (1 << d));
      left = 2 * parent < top_n ? top[2 * parent] : nodes[2 * parent];
    }
    int t = 1;  // wave-uniform: kept in SGPRs so the broadcast is a v_readlane, not a ds_bpermute
    for (int s = 0; s < k; ++s) {
      const float l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(left), __builtin_amdgcn_readfirstlane(t)));
      if (query < l) {
        t = 2 * t;
      } else {
        query -= l;
        t = 2 * t + 1;
      }
    }
    node = (node << k) + (t - (1 << k));
    remaining -= k;
  }
  if (leaf_val) *leaf_val = nodes[node];
  return node - cap;
}

__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ nodes, long long cap, int depth,
                                                     const float* __restrict__ q, int64_t* __restrict__ idx,
                                                     float* __restrict__ val, long long n) {
  const int lane = threadIdx.x & 63;
  const long long i = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float query = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q[i] * nodes[1])));  // sum_tree.h:47
  float leaf;
  const long long k = wave_descend(nodes, cap, depth, query, lane, &leaf);
  if (lane == 0) {
    idx[i] = k;
    if (val) val[i] = leaf;
  }
}

__global__ __launch_bounds__(256) void per_sample_kernel(const float* __restrict__ nodes, long long cap, int depth,
                                                         const double* __restrict__ u, long long batch, int unit,
                                                         unsigned long long seed, const float* __restrict__ counter,
                                                         int64_t* __restrict__ idx, double* __restrict__ prob) {
  const int lane = threadIdx.x & 63;
  const long long i = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (i >= batch) return;
  // numpy.linspace(1/B, 1, B)[i] - u[i] in float64, then the float conversion pybind applies (priority_buffer.py:37-41)
  const double start = 1.0 / static_cast<double>(batch);
  const double step = batch > 1 ? (1.0 - start) / static_cast<double>(batch - 1) : 0.0;
  const double lin = (batch > 1 && i == batch - 1) ? 1.0 : static_cast<double>(i) * step + start;
  const float total = nodes[1];
  double ui;
  if (unit == 2) {  // drawn here: 53 random bits of Philox(seed; query, counter) -> [0, 1), scaled to the stratum width
    const unsigned long long c = static_cast<unsigned long long>(*counter);
    uint32_t r[4];
    hb::philox4x32_10(static_cast<uint32_t>(i), static_cast<uint32_t>(static_cast<unsigned long long>(i) >> 32),
                      static_cast<uint32_t>(c), static_cast<uint32_t>(c >> 32), static_cast<uint32_t>(seed),
                      static_cast<uint32_t>(seed >> 32), r);
    const unsigned long long bits = (static_cast<unsigned long long>(r[0] >> 5) << 26) | (r[1] >> 6);
    ui = static_cast<double>(bits) * 0x1.0p-53 / static_cast<double>(batch);
  } else {
    ui = unit ? u[i] / static_cast<double>(batch) : u[i];  // unit: u in [0,1) is scaled to the stratum width here
  }
  const float query = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(static_cast<float>(lin - ui) * total)));
  float leaf;
  const long long k = wave_descend(nodes, cap, depth, query, lane, &leaf);
  if (lane == 0) {
    idx[i] = k;
    prob[i] = (static_cast<double>(leaf) + 1e-10) / static_cast<double>(total);  // priority_buffer.py:42
  }
}

// ---------------------------------------------------------------------------------------------
// per_sample (uniforms drawn in the kernel) + the replay gather of learner.hip in ONE launch: workgroup i descends the tree for
// stratum i with its first wavefront, then all four wavefronts expand the two observation rows of the sampled transition
// (obs_tm1 -> x[i], obs_t of the n-step chain's last transition -> x[B + i]) into the GEMM operand. Same arithmetic as
// per_sample_kernel (unit == 2) followed by gather_kernel; saves a launch and a dependent kernel boundary on the learner's
// critical path.
struct SampleGatherArgs {
  const float* nodes; long long cap; int depth;
  unsigned long long seed; const float* counter; int batch;
  int64_t* idx; double* prob;
  const void* ring_tm1; const void* ring_t; const int8_t* ring_act; const float* ring_rew; const uint8_t* ring_term;
  int L, packed_words; void* x; int x_dtype, x_ld;
  int32_t* act; float* rew; float* term; float* disc;
  const long long* size_wp; long long ring_cap, n_ins; int n_step; float gamma;
  int ntop;   // > 0: the tree keeps its top levels lazily; every workgroup re-sums them from the ntop subtree roots in LDS
};
template <typename T> __device__ __forceinline__ void sg_store(T* p, long long i, float v);
template <> __device__ __forceinline__ void sg_store<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void sg_store<__hip_bfloat16>(__hip_bfloat16* p, long long i, float v) { p[i] = __float2bfloat16(v); }
template <> __device__ __forceinline__ void sg_store<__half>(__half* p, long long i, float v) { p[i] = __float2half(v); }

template <typename T>
__global__ __launch_bounds__(256) void per_sample_gather_kernel(const SampleGatherArgs a) {
  __shared__ long long s_slot[2];
  extern __shared__ float s_top[];  // 2 * ntop floats in lazy-top mode
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long i = blockIdx.x;
  const int B = a.batch;
  if (a.ntop > 0) {
    // the same pairwise sums rebuild_top_kernel would have stored (every inner node = fl(left + right)): identical bits
    for (int j = threadIdx.x; j < a.ntop; j += 256) s_top[a.ntop + j] = a.nodes[a.ntop + j];
    __syncthreads();
    for (int w = a.ntop >> 1; w >= 1; w >>= 1) {
      for (int j = threadIdx.x; j < w; j += 256) s_top[w + j] = s_top[2 * (w + j)] + s_top[2 * (w + j) + 1];
      __syncthreads();
    }
  }
  if (wave == 0) {
    const double start = 1.0 / static_cast<double>(B);
    const double step = B > 1 ? (1.0 - start) / static_cast<double>(B - 1) : 0.0;
    const double lin = (B > 1 && i == B - 1) ? 1.0 : static_cast<double>(i) * step + start;
    const float total = a.ntop > 0 ? s_top[1] : a.nodes[1];
    const unsigned long long c = static_cast<unsigned long long>(*a.counter);
    uint32_t r[4];
    hb::philox4x32_10(static_cast<uint32_t>(i), static_cast<uint32_t>(static_cast<unsigned long long>(i) >> 32),
                      static_cast<uint32_t>(c), static_cast<uint32_t>(c >> 32), static_cast<uint32_t>(a.seed),
                      static_cast<uint32_t>(a.seed >> 32), r);
    const unsigned long long bits = (static_cast<unsigned long long>(r[0] >> 5) << 26) | (r[1] >> 6);
    const double ui = static_cast<double>(bits) * 0x1.0p-53 / static_cast<double>(B);
    const float query = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(static_cast<float>(lin - ui) * total)));
    float leaf;
    const long long k = wave_descend(a.nodes, a.cap, a.depth, query, lane, &leaf, s_top, 2 * a.ntop);
    if (lane == 0) {
      a.idx[i] = k;
      a.prob[i] = (static_cast<double>(leaf) + 1e-10) / static_cast<double>(total);
      // the n-step chain of this sample (gather_kernel, learner.hip): a handful of dependent 1-byte / 4-byte reads
      long long j = k;
      float R = a.ring_rew[j], g = a.gamma;
      int m = 1;
      if (a.n_step > 1) {
        const long long size = a.size_wp[0], wp = a.size_wp[1];
        const long long ahead = size >= a.ring_cap ? ((wp - 1 - k) % a.ring_cap + a.ring_cap) % a.ring_cap : size - 1 - k;
        while (m < a.n_step && !a.ring_term[j] && static_cast<long long>(m) * a.n_ins <= ahead) {
          j = (j + a.n_ins) % a.ring_cap;
          R += g * a.ring_rew[j];
          g *= a.gamma;
          ++m;
        }
      }
      a.rew[i] = R;
      a.term[i] = a.ring_term[j] ? 1.f : 0.f;
      a.disc[i] = g;
      a.act[i] = a.ring_act[k];
      s_slot[0] = k;
      s_slot[1] = j;
    }
  }
  __syncthreads();
  T* x = static_cast<T*>(a.x);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const long long slot = s_slot[half];
    T* dst = x + (static_cast<long long>(half) * B + i) * a.x_ld;
    if (a.packed_words > 0) {
      const uint32_t* src = static_cast<const uint32_t*>(half == 0 ? a.ring_tm1 : a.ring_t) + slot * a.packed_words;
      for (int jj = threadIdx.x; jj < a.L; jj += 256) sg_store<T>(dst, jj, static_cast<float>((src[jj >> 5] >> (jj & 31)) & 1u));
    } else {
      const int8_t* src = static_cast<const int8_t*>(half == 0 ? a.ring_tm1 : a.ring_t) + slot * a.L;
      for (int jj = threadIdx.x; jj < a.L; jj += 256) sg_store<T>(dst, jj, static_cast<float>(src[jj]));
    }
  }
}

__global__ void get_kernel(const float* __restrict__ nodes, long long cap, const int64_t* __restrict__ idx,
                           float* __restrict__ val, long long n, unsigned long long* errors) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long k = idx[i];
  if (k < 0 || k >= cap) {
    atomicAdd(errors, 1ull);
    val[i] = 0.f;
  } else {
    val[i] = nodes[cap + k];
  }
}

void launch_top(hb_tree* t, hipStream_t s) {
  const int ntop = static_cast<int>(t->cap / t->chunk);
  if (ntop > 1)
    hipLaunchKernelGGL(rebuild_top_kernel, dim3(1), dim3(ntop >= 2048 ? 1024 : (ntop / 2 < 64 ? 64 : ntop / 2)),
                       2 * ntop * sizeof(float), s, t->nodes, ntop);
}
// lazy mode: the levels above the subtree roots are only as fresh as the last reader made them
void freshen_top(hb_tree* t, hipStream_t s) {
  if (t->lazy_top || t->top_stale) launch_top(t, s);
  if (!t->lazy_top) t->top_stale = false;
}

int rebuild(hb_tree* t, int first_chunk, int nchunks_touched, hipStream_t s) {
  hipLaunchKernelGGL((chunk_kernel<false>), dim3(nchunks_touched), dim3(256), 0, s, t->nodes, t->cap, t->chunk,
                     first_chunk, 0LL, 0LL, static_cast<const float*>(nullptr));
  const int ntop = static_cast<int>(t->cap / t->chunk);
  if (ntop > 1)
    hipLaunchKernelGGL(rebuild_top_kernel, dim3(1), dim3(ntop >= 2048 ? 1024 : (ntop / 2 < 64 ? 64 : ntop / 2)),
                       2 * ntop * sizeof(float), s, t->nodes, ntop);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

}  // namespace

extern "C" {

int hb_tree_create(int64_t capacity, hb_tree** out) {
  if (!out) return fail(HB_ERR_INVALID, "null out");
  *out = nullptr;
  if (capacity < 1 || capacity > (1LL << 23)) return fail(HB_ERR_INVALID, "capacity must be in [1, 2^23]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(HB_ERR_NO_DEVICE, "no HIP device available");
  long long cap = 1;
  int depth = 0;
  while (cap < capacity) { cap <<= 1; ++depth; }  // depth = ceil(log2(capacity)) (sum_tree.h:28)
  hb_tree* t = new (std::nothrow) hb_tree();
  if (!t) return fail(HB_ERR_NOMEM, "out of host memory");
  t->cap = cap;
  t->depth = depth;
  t->chunk = cap < 1024 ? static_cast<int>(cap) : 1024;
  t->stamp = nullptr;
  t->scratch = nullptr;
  t->scratch_n = 0;
  t->lazy_top = false;
  t->top_stale = false;
  {
    const char* path = getenv("HB_TREE_UPDATE_PATH");
    t->single_wg = path && path[0] == 's';
    t->fused_pow = !(path && path[0] == 't');
  }
  t->small_scratch = nullptr;
  HB_HIP_OR(hipMalloc(reinterpret_cast<void**>(&t->nodes), 2 * cap * sizeof(float)), delete t);
  HB_HIP_OR(hipMalloc(reinterpret_cast<void**>(&t->errors), 8), { (void)hipFree(t->nodes); delete t; });
  HB_HIP_OR(hipMemset(t->nodes, 0, 2 * cap * sizeof(float)), hb_tree_destroy(t));
  HB_HIP_OR(hipMemset(t->errors, 0, 8), hb_tree_destroy(t));
  HB_HIP_OR(hipMalloc(reinterpret_cast<void**>(&t->small_scratch), SMALL_MAX * sizeof(float)), hb_tree_destroy(t));
  *out = t;
  return HB_OK;
}

int hb_tree_destroy(hb_tree* t) {
  if (!t) return HB_OK;
  if (t->nodes) (void)hipFree(t->nodes);
  if (t->errors) (void)hipFree(t->errors);
  if (t->stamp) (void)hipFree(t->stamp);
  if (t->scratch) (void)hipFree(t->scratch);
  if (t->small_scratch) (void)hipFree(t->small_scratch);
  delete t;
  return HB_OK;
}

int64_t hb_tree_capacity(const hb_tree* t) { return t ? t->cap : 0; }
float* hb_tree_nodes(hb_tree* t) { return t ? t->nodes : nullptr; }

static int update_impl(hb_tree* t, const int64_t* idx, const float* val, int64_t n, int per_mode, double alpha,
                       float* max_prio, float* min_prio, hipStream_t s) {
  if (n <= 0) return HB_OK;
  // (measured at cap 2^19, scripts/tree_probe.py: one workgroup 7.9 / 12.6 / 20.0 / 68 us at n = 32 / 128 / 256 / 1024;
  //  per-subtree workgroups 11-12 us flat)
  if (n >= 96 && n <= SMALL_MAX && t->cap >= 4 * static_cast<long long>(t->chunk) && !t->single_wg) {
    if (per_mode && t->fused_pow) {
      hipLaunchKernelGGL(update_chunks_kernel<true>, dim3(static_cast<unsigned>(n + 1)), dim3(256), 0, s, t->nodes, t->cap, t->chunk,
                         idx, val, static_cast<int>(n), alpha, max_prio, min_prio, t->errors);
    } else {
      const float* v = val;
      if (per_mode) {   // two small launches that both fit beside the policy kernel instead of one that does not
        hipLaunchKernelGGL(per_transform_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, val, t->small_scratch,
                           static_cast<long long>(n), alpha, max_prio, min_prio);
        v = t->small_scratch;
      }
      hipLaunchKernelGGL(update_chunks_kernel<false>, dim3(static_cast<unsigned>(n + 1)), dim3(256), 0, s, t->nodes, t->cap, t->chunk,
                         idx, v, static_cast<int>(n), 0.0, static_cast<float*>(nullptr), static_cast<float*>(nullptr), t->errors);
    }
    if (!t->lazy_top) launch_top(t, s);
    HB_HIP(hipGetLastError());
    return HB_OK;
  }
  if (n <= SMALL_MAX) {
    const int threads = n <= 256 ? 256 : (n <= 512 ? 512 : 1024);
    hipLaunchKernelGGL(update_small_kernel, dim3(1), dim3(threads), 0, s, t->nodes, t->cap, t->depth, idx, val,
                       static_cast<int>(n), per_mode, alpha, max_prio, min_prio, t->errors);
    HB_HIP(hipGetLastError());
    return HB_OK;
  }
  if (!t->stamp) {
    HB_HIP(hipMalloc(reinterpret_cast<void**>(&t->stamp), t->cap * sizeof(unsigned)));
    HB_HIP(hipMemsetAsync(t->stamp, 0, t->cap * sizeof(unsigned), s));
  }
  const float* v = val;
  if (per_mode) {
    if (t->scratch_n < n) {
      if (t->scratch) HB_HIP(hipFree(t->scratch));
      t->scratch = nullptr;
      HB_HIP(hipMalloc(reinterpret_cast<void**>(&t->scratch), n * sizeof(float)));
      t->scratch_n = n;
    }
    hipLaunchKernelGGL(per_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, s, val, t->scratch, n, alpha, max_prio,
                       min_prio);
    v = t->scratch;
  }
  const unsigned blocks = static_cast<unsigned>((n + 255) / 256);
  hipLaunchKernelGGL(stamp_kernel, dim3(blocks), dim3(256), 0, s, t->stamp, t->cap, idx, n, t->errors);
  hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, s, t->nodes, t->stamp, t->cap, idx, v, n);
  return rebuild(t, 0, static_cast<int>(t->cap / t->chunk), s);
}

int hb_tree_update(hb_tree* t, const int64_t* idx_dev, const float* val_dev, int64_t n, void* stream) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  if (n > 0 && (!idx_dev || !val_dev)) return fail(HB_ERR_INVALID, "null argument");
  return update_impl(t, idx_dev, val_dev, n, 0, 0.0, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int hb_per_update(hb_tree* t, const int64_t* idx_dev, const float* td_dev, int64_t n, double alpha, float* max_prio_dev,
                  float* min_prio_dev, void* stream) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  if (n > 0 && (!idx_dev || !td_dev)) return fail(HB_ERR_INVALID, "null argument");
  // array float32 ** python float: the exponent is taken at float32 precision (priority_buffer.py:49)
  const double a = static_cast<double>(static_cast<float>(alpha));
  return update_impl(t, idx_dev, td_dev, n, 1, a, max_prio_dev, min_prio_dev, static_cast<hipStream_t>(stream));
}

int hb_tree_fill_range(hb_tree* t, int64_t start, int64_t n, const float* value_dev, void* stream) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  if (n <= 0) return HB_OK;
  if (!value_dev) return fail(HB_ERR_INVALID, "null value_dev");
  if (start < 0 || start >= t->cap) return fail(HB_ERR_INVALID, "start out of range");
  if (n > t->cap) n = t->cap;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long nchunks = t->cap / t->chunk;
  long long touched = ((start % t->chunk) + n + t->chunk - 1) / t->chunk;
  if (touched > nchunks) touched = nchunks;
  hipLaunchKernelGGL((chunk_kernel<true>), dim3(static_cast<unsigned>(touched)), dim3(256), 0, s, t->nodes, t->cap,
                     t->chunk, static_cast<int>(start / t->chunk), static_cast<long long>(start),
                     static_cast<long long>(n), value_dev);
  if (!t->lazy_top) launch_top(t, s);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_tree_sample(hb_tree* t, const float* quantile_dev, int64_t* idx_dev, float* val_dev, int64_t n, void* stream) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  if (n <= 0) return HB_OK;
  if (!quantile_dev || !idx_dev) return fail(HB_ERR_INVALID, "null argument");
  freshen_top(t, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(sample_kernel, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), t->nodes, t->cap, t->depth, quantile_dev, idx_dev, val_dev,
                     static_cast<long long>(n));
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_per_sample(hb_tree* t, const double* u_dev, int64_t batch, int32_t unit_uniforms, int64_t* idx_dev, double* prob_dev,
                  void* stream) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  if (batch <= 0) return HB_OK;
  if (!u_dev || !idx_dev || !prob_dev) return fail(HB_ERR_INVALID, "null argument");
  freshen_top(t, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(per_sample_kernel, dim3(static_cast<unsigned>((batch + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), t->nodes, t->cap, t->depth, u_dev, static_cast<long long>(batch),
                     unit_uniforms != 0 ? 1 : 0, 0ull, static_cast<const float*>(nullptr), idx_dev, prob_dev);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_per_sample_philox(hb_tree* t, uint64_t seed, const float* counter_dev, int64_t batch, int64_t* idx_dev,
                         double* prob_dev, void* stream) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  if (batch <= 0) return HB_OK;
  if (!counter_dev || !idx_dev || !prob_dev) return fail(HB_ERR_INVALID, "null argument");
  freshen_top(t, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(per_sample_kernel, dim3(static_cast<unsigned>((batch + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), t->nodes, t->cap, t->depth, static_cast<const double*>(nullptr),
                     static_cast<long long>(batch), 2, static_cast<unsigned long long>(seed), counter_dev, idx_dev, prob_dev);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_per_sample_gather(hb_tree* t, uint64_t seed, const float* counter_dev, int64_t batch, int64_t* idx_dev, double* prob_dev,
                         const void* ring_obs_tm1_dev, const void* ring_obs_t_dev, const int8_t* ring_act_dev,
                         const float* ring_rew_dev, const uint8_t* ring_term_dev, int32_t obs_len, int32_t packed, void* x_dev,
                         int32_t x_dtype, int32_t x_ld, int32_t* act_dev, float* rew_dev, float* term_dev, float* disc_dev,
                         int32_t n_step, float gamma, int64_t capacity, int64_t rows_per_insert, const int64_t* size_wp_dev,
                         void* stream) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  if (batch <= 0) return HB_OK;
  if (!counter_dev || !idx_dev || !prob_dev || !ring_obs_tm1_dev || !ring_obs_t_dev || !ring_act_dev || !ring_rew_dev ||
      !ring_term_dev || !x_dev || !act_dev || !rew_dev || !term_dev || !disc_dev)
    return fail(HB_ERR_INVALID, "null argument");
  if (obs_len < 1 || x_ld < obs_len) return fail(HB_ERR_INVALID, "x_ld must be >= obs_len >= 1");
  if (n_step < 1) return fail(HB_ERR_INVALID, "n_step must be >= 1");
  if (n_step > 1 && (!size_wp_dev || rows_per_insert < 1 || capacity < 1))
    return fail(HB_ERR_INVALID, "n_step > 1 needs size_wp_dev, rows_per_insert and capacity");
  SampleGatherArgs a{t->nodes, t->cap, t->depth, static_cast<unsigned long long>(seed), counter_dev, static_cast<int>(batch),
                     idx_dev, prob_dev, ring_obs_tm1_dev, ring_obs_t_dev, ring_act_dev, ring_rew_dev, ring_term_dev, obs_len,
                     packed ? (obs_len + 31) / 32 : 0, x_dev, x_dtype, x_ld, act_dev, rew_dev, term_dev, disc_dev,
                     reinterpret_cast<const long long*>(size_wp_dev), capacity, rows_per_insert, n_step, gamma, 0};
  const int ntop = static_cast<int>(t->cap / t->chunk);
  if (t->lazy_top && ntop > 1) a.ntop = ntop;
  const size_t lds = static_cast<size_t>(2 * a.ntop) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(batch)), block(256);
  if (x_dtype == 0) hipLaunchKernelGGL((per_sample_gather_kernel<float>), grid, block, lds, s, a);
  else if (x_dtype == 1) hipLaunchKernelGGL((per_sample_gather_kernel<__hip_bfloat16>), grid, block, lds, s, a);
  else if (x_dtype == 2) hipLaunchKernelGGL((per_sample_gather_kernel<__half>), grid, block, lds, s, a);
  else return fail(HB_ERR_INVALID, "x_dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_tree_get(hb_tree* t, const int64_t* idx_dev, float* val_dev, int64_t n, void* stream) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  if (n <= 0) return HB_OK;
  if (!idx_dev || !val_dev) return fail(HB_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(get_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), t->nodes, t->cap, idx_dev, val_dev, static_cast<long long>(n),
                     t->errors);
  HB_HIP(hipGetLastError());
  return HB_OK;
}

int hb_tree_total(hb_tree* t, float* total_dev, void* stream) {
  if (!t || !total_dev) return fail(HB_ERR_INVALID, "null argument");
  freshen_top(t, static_cast<hipStream_t>(stream));
  HB_HIP(hipMemcpyAsync(total_dev, t->nodes + 1, sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return HB_OK;
}

int hb_tree_set_lazy_top(hb_tree* t, int32_t on) {
  if (!t) return fail(HB_ERR_INVALID, "null tree");
  // Leaving lazy mode: the stored top levels are stale. They are NOT re-summed here — this call has no stream, and the tree's
  // writers run on the caller's (non-blocking) streams, so a launch on the NULL stream would not be ordered behind them — but
  // by the next reader or writer, on the stream that call is given (every writer of the non-lazy mode ends with the re-sum).
  if (t->lazy_top && !on) t->top_stale = true;
  t->lazy_top = on != 0;
  return HB_OK;
}

int hb_tree_export_nodes(hb_tree* t, float* nodes_dev, void* stream) {
  if (!t || !nodes_dev) return fail(HB_ERR_INVALID, "null argument");
  freshen_top(t, static_cast<hipStream_t>(stream));
  HB_HIP(hipMemcpyAsync(nodes_dev, t->nodes, 2 * t->cap * sizeof(float), hipMemcpyDeviceToDevice,
                        static_cast<hipStream_t>(stream)));
  return HB_OK;
}

int hb_tree_import_nodes(hb_tree* t, const float* nodes_dev, void* stream) {
  if (!t || !nodes_dev) return fail(HB_ERR_INVALID, "null argument");
  HB_HIP(hipMemcpyAsync(t->nodes, nodes_dev, 2 * t->cap * sizeof(float), hipMemcpyDeviceToDevice,
                        static_cast<hipStream_t>(stream)));
  return HB_OK;
}

int hb_tree_error_count(hb_tree* t, int64_t* out) {
  if (!t || !out) return fail(HB_ERR_INVALID, "null argument");
  unsigned long long v = 0;
  HB_HIP(hipMemcpy(&v, t->errors, 8, hipMemcpyDeviceToHost));
  *out = static_cast<int64_t>(v);
  return HB_OK;
}

}  // extern "C"
