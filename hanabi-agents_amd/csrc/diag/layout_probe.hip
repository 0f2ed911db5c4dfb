// layout_probe.hip — diagnostic only (never shipped, never loaded by the package): the game-state transport of the env kernel
// in the two layouts BASELINE.json's north_star words could mean, side by side on the same data.
//   AoS (shipped, csrc/env_kernel.hpp phases 1 and 3): one 128-byte row per game; a wave copies its G rows HBM -> LDS with
//        coalesced 16-byte loads, each game's lane reads its row from LDS (odd stride, conflict-free), and the row goes back
//        the same way.
//   SoA: 32 word planes of [n_games]; the game's lane loads word w from plane[w][game] (a wave load covers G * 4 contiguous
//        bytes), no LDS staging.
// Both kernels do the same token arithmetic on the 32 words (so that nothing is optimised away) and move the same 256 bytes
// per game. scripts/layout_probe.py times them at the G the env kernel runs (16 games per wave at 32 768 games: the rules are
// lane-per-game and register-heavy, a wave cannot hold 64 games' encoder state) and at G = 64.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int SW = 32, SWP = 33;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void token_work(uint32_t (&w)[SW], uint32_t act) {
  // a dependent pass over the row, as the rules make one: every word changes, nothing can be dropped
  uint32_t carry = act;
#pragma unroll
  for (int i = 0; i < SW; ++i) {
    w[i] = w[i] + carry;
    carry = (w[i] >> 7) ^ static_cast<uint32_t>(i);
  }
}

template <int G>
__global__ __launch_bounds__(256) void aos_kernel(uint32_t* __restrict__ state, const int32_t* __restrict__ actions, long long n) {
  __shared__ uint32_t lds[4 * G * SWP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* const srow = lds + wave * G * SWP;
  const long long g0 = (static_cast<long long>(blockIdx.x) * 4 + wave) * G;
  const long long left = n - g0;
  const int nvalid = left <= 0 ? 0 : (left < G ? static_cast<int>(left) : G);
  const bool active = lane < nvalid && lane < G;
  const uint32_t act = active ? static_cast<uint32_t>(actions[g0 + lane]) : 0u;
  constexpr int Q = SW / 4, ROUNDS = (G * Q + 63) / 64;
  uint4* const rows = reinterpret_cast<uint4*>(state + g0 * SW);
  uint4 v[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = lane + 64 * r;
    v[r] = e < nvalid * Q ? rows[e] : make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = lane + 64 * r;
    if (e < nvalid * Q) {
      uint32_t* d = srow + (e / Q) * SWP + 4 * (e % Q);
      d[0] = v[r].x; d[1] = v[r].y; d[2] = v[r].z; d[3] = v[r].w;
    }
  }
  wave_sync();
  if (active) {
    uint32_t w[SW];
    uint32_t* const row = srow + lane * SWP;
#pragma unroll
    for (int i = 0; i < SW; ++i) w[i] = row[i];
    token_work(w, act);
#pragma unroll
    for (int i = 0; i < SW; ++i) row[i] = w[i];
  }
  wave_sync();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = lane + 64 * r;
    if (e < nvalid * Q) {
      const uint32_t* s = srow + (e / Q) * SWP + 4 * (e % Q);
      rows[e] = make_uint4(s[0], s[1], s[2], s[3]);
    }
  }
}

template <int G>
__global__ __launch_bounds__(256) void soa_kernel(uint32_t* __restrict__ planes, const int32_t* __restrict__ actions, long long n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long g = (static_cast<long long>(blockIdx.x) * 4 + wave) * G + lane;
  if (lane >= G || g >= n) return;
  const uint32_t act = static_cast<uint32_t>(actions[g]);
  uint32_t w[SW];
#pragma unroll
  for (int i = 0; i < SW; ++i) w[i] = planes[static_cast<long long>(i) * n + g];
  token_work(w, act);
#pragma unroll
  for (int i = 0; i < SW; ++i) planes[static_cast<long long>(i) * n + g] = w[i];
}

template <int G>
int launch(int soa, uint32_t* state, const int32_t* actions, long long n, hipStream_t s) {
  const unsigned blocks = static_cast<unsigned>((n + 4 * G - 1) / (4 * G));
  if (soa) hipLaunchKernelGGL(soa_kernel<G>, dim3(blocks), dim3(256), 0, s, state, actions, n);
  else hipLaunchKernelGGL(aos_kernel<G>, dim3(blocks), dim3(256), 0, s, state, actions, n);
  return static_cast<int>(hipGetLastError());
}

}  // namespace

extern "C" int probe_layout(int soa, int games_per_wave, void* state, const void* actions, long long n, void* stream) {
  uint32_t* st = static_cast<uint32_t*>(state);
  const int32_t* ac = static_cast<const int32_t*>(actions);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 0 || !state || !actions) return -1;
  switch (games_per_wave) {
    case 16: return launch<16>(soa, st, ac, n, s);
    case 32: return launch<32>(soa, st, ac, n, s);
    case 64: return launch<64>(soa, st, ac, n, s);
    default: return -1;
  }
}
