// env_kernel.hpp — fused Hanabi env step + legal-move mask + canonical observation encoder
// for gfx950 (MI355X). One kernel launch advances N independent games by one move.
//
// Replaces, for the reference's parallel-env call path (external hanabi_learning_environment;
// spec SURVEY.md App. A; call sites hanabi_agents/rule_based/ruleset.py:4,
// hanabi_agents/rainbow/run_experiment.py:266-273,308): HanabiState::ApplyMove + deal +
// terminal test + reward (A.5), MoveIsLegal over all uids (A.4) and
// CanonicalObservationEncoder::Encode (A.6).
//
// Mapping:
//   * a 64-lane wavefront owns G consecutive games (G = 16/32/64); a 256-thread workgroup is
//     four independent wavefronts. Game state is an AoS row of SW u32 (128 B or 192 B) in HBM;
//     the wave copies its G rows HBM -> LDS with 16-byte coalesced loads (row stride SW+1 words
//     so that lane-per-game LDS accesses are bank-conflict free).
//   * lane g < G runs the rules for its game on the LDS row (scalar fields in VGPRs, hands /
//     knowledge / deck addressed dynamically in LDS) and builds the observation as a BIT-PACKED
//     vector (OBS_LEN bits = 21 words for 2 players) with compile-time field offsets.
//   * a finished game is re-dealt by the whole wavefront: lane j draws the Philox key of deck
//     position j, ranks it against the other keys with v_readlane broadcasts, and drops card j
//     at its rank (sort-by-random-key shuffle; the tie-break by position makes it a deterministic function).
//   * the wave then expands bits -> int8 cooperatively: every lane turns 16 bits into one
//     16-byte store, so the [N, OBS_LEN] int8 output is written as full 1-KiB wave stores.
// Everything is integer/bit work; the kernel is HBM-bound (943 B per env-step
// for 2-player full Hanabi).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <utility>

namespace hb {

enum { MODE_STEP = 0, MODE_OBSERVE = 1, MODE_RESET = 2 };
constexpr int NEXT_DECK_BYTES = 64;  // one pre-shuffled deck per game, padded to a 64-byte row
enum { MV_PLAY = 0, MV_DISCARD = 1, MV_RCOLOR = 2, MV_RRANK = 3 };

struct EnvArgs {
  uint32_t* state;
  const int32_t* actions;
  const uint8_t* mask;
  const uint8_t* decks;
  uint8_t* next_deck;  // [n, 64]: the deck game g will be dealt at its next (re)deal, produced by refill_kernel
  uint8_t* refill;     // [n]: 1 = next_deck[g] was consumed (or is stale) and must be regenerated
  int8_t* obs;          // [n, OBS_LEN] int8 0/1 (the reference's layout), or NULL when only the packed form is wanted
  uint32_t* obs_bits;   // [n, NW] u32, bit i of the observation = word i >> 5, bit i & 31 (pad bits zero), or NULL
  int8_t* legal;
  float* reward;
  int8_t* terminal;
  float* agent_reward;
  int8_t* agent_step_type;
  int8_t* score;
  unsigned long long* illegal;
  unsigned long long* stats;  // per-wavefront running totals [n/16 + 4][2]: episodes finished, sum of final scores
  long long n;
  unsigned long long seed;
  long long first_gid;
  int flags;
  int mode;
  int start_player;
  // MODE_STEP with the agent's action selection fused in (hb_env_step_select*): actions == NULL and
  //   action[g] = eps-greedy(sel_q[g, :], sel_legal[g, :]; Philox(sel_draw, sel_first_gid + g; sel_seed))
  // by the rule of hb_policy_select (csrc/actor.hip: select_rows); the chosen uid is also written to sel_actions[g]
  const float* sel_q;          // [n, A] expected returns
  const int8_t* sel_legal;     // [n, A] the acting seat's legal mask (the previous step's `legal` output)
  int32_t* sel_actions;        // [n] out
  float sel_eps;
  unsigned long long sel_seed, sel_draw;
  long long sel_first_gid;
  unsigned long long* stamps;  // diagnostic builds only (-DHB_STAMPS): 12 u64 per wavefront
  hipEvent_t ev_start, ev_stop;  // host-side only: optional per-dispatch timing events
};

// In-kernel phase stamps (cdna_hip_programming.md §7): compiled in only with -DHB_STAMPS, into a
// separate diagnostic library; the shipped kernel contains none of this.
#ifdef HB_STAMPS
#define HB_STAMP(slot)                                                                               \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    unsigned long long hb_t_;                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t_)::"memory");                    \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    if (a.stamps && lane == 0) a.stamps[(static_cast<long long>(blockIdx.x) * 4 + wave) * 12 + (slot)] = hb_t_; \
  } while (0)
#define HB_STAMP_REAL(slot)                                                                          \
  do {                                                                                               \
    if (a.stamps && lane == 0)                                                                       \
      a.stamps[(static_cast<long long>(blockIdx.x) * 4 + wave) * 12 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define HB_STAMP(slot) do { } while (0)
#define HB_STAMP_REAL(slot) do { } while (0)
#endif

// ---- compile-time description of one game configuration ----
template <int P_, int C_, int R_, int H_, int INFO_, int LIFE_>
struct Cfg {
  static constexpr int P = P_, C = C_, R = R_, H = H_, INFO = INFO_, LIFE = LIFE_;
  static constexpr int BITS = C * R;
  static constexpr int copies(int r) { return r == 0 ? 3 : (r == R - 1 ? 1 : 2); }
  static constexpr int cum(int r) {
    int s = 0;
    for (int i = 0; i < r; ++i) s += copies(i);
    return s;
  }
  static constexpr int CPC = cum(R);  // cards per colour
  static constexpr int D = C * CPC;   // deck size
  static constexpr int A = 2 * H + (P - 1) * (C + R);
  // observation sections
  static constexpr int FLAGS_OFF = (P - 1) * H * BITS;
  static constexpr int BOARD_OFF = FLAGS_OFF + P;
  static constexpr int DECK_T = D - P * H;
  static constexpr int FW_OFF = BOARD_OFF + DECK_T;
  static constexpr int INFO_OFF = FW_OFF + BITS;
  static constexpr int LIFE_OFF = INFO_OFF + INFO;
  static constexpr int DISC_OFF = LIFE_OFF + LIFE;
  static constexpr int LA_OFF = DISC_OFF + D;
  static constexpr int LA_LEN = P + 4 + P + C + R + H + H + BITS + 2;
  static constexpr int KN_OFF = LA_OFF + LA_LEN;
  static constexpr int KN_SLOT = BITS + C + R;
  static constexpr int OBS_LEN = KN_OFF + P * H * KN_SLOT;
  static constexpr int NW = (OBS_LEN + 31) / 32;
  static constexpr int NWP = (NW + 1) | 1;  // >= NW+1 zero pad word, odd LDS row stride
  static constexpr int LW = 3;              // legal bits: 2 words + zero pad
  // state row
  static constexpr int SW = P <= 3 ? 32 : 48;
  static constexpr int SWP = SW + 1;
  static constexpr int W_HANDS = 10, W_KNOW = 10 + P, W_DECK = 10 + 3 * P;
  static constexpr uint32_t ALL_PLAUSIBLE = ((1u << C) - 1u) | (((1u << R) - 1u) << 5);
  static_assert(W_DECK * 4 + D <= SW * 4, "state row too small");
  static_assert(BITS <= 25 && D <= 50 && A <= 64 && H <= 5 && P <= 5 && INFO <= 15 && LIFE <= 7, "limits");
};

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// Colour-plausibility bits c -> bit c*R. Bit c has to move up by (R-1)*c: multiply by sum_k 2^((R-1)k) and keep,
// of copy k, only bit k. The even and the odd colours go separately so that neighbouring copies (2(R-1) apart)
// never overlap and no addition carries.
template <int C, int R>
__device__ __forceinline__ uint32_t spread_colors(uint32_t cp) {
  if constexpr (2 * (R - 1) >= C && C * R <= 24 + 1) {
    constexpr auto sum = [](int first, int step) constexpr {
      uint32_t v = 0;
      for (int k = first; k < C; k += 2) v |= 1u << (step * k);
      return v;
    };
    constexpr uint32_t EIN = sum(0, 1), OIN = sum(1, 1), EMUL = sum(0, R - 1), OMUL = sum(1, R - 1), ESEL = sum(0, R),
                       OSEL = sum(1, R);
    uint32_t v = __umul24(cp & EIN, EMUL) & ESEL;
    if constexpr (C > 1) v |= __umul24(cp & OIN, OMUL) & OSEL;
    return v;
  } else {
    uint32_t v = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) v |= ((cp >> c) & 1u) << (c * R);
    return v;
  }
}

// bit-packed observation under construction; every offset is a compile-time constant so the
// words stay in VGPRs
template <int NW>
struct BitAcc {
  uint32_t w[NW];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = 0;
  }
  template <int OFF, int LEN>
  __device__ __forceinline__ void put(uint32_t v) {  // v < 2^LEN, LEN <= 32
    static_assert(LEN >= 0 && LEN <= 32 && OFF >= 0 && (OFF + LEN + 31) / 32 <= NW, "put range");
    if constexpr (LEN > 0) {  // (a zero-length section exists: 5-player very-small has no deck thermometer)
      constexpr int wi = OFF >> 5, sh = OFF & 31;
      w[wi] |= v << sh;
      if constexpr (sh + LEN > 32) w[wi + 1] |= v >> (32 - sh);
    }
  }
  template <int OFF, int LEN>
  __device__ __forceinline__ void put64(uint64_t v) {  // v < 2^LEN, LEN <= 64
    static_assert(LEN >= 0 && LEN <= 64, "put64 range");
    if constexpr (LEN <= 32) {
      put<OFF, LEN>(static_cast<uint32_t>(v));
    } else {
      put<OFF, 32>(static_cast<uint32_t>(v));
      put<OFF + 32, LEN - 32>(static_cast<uint32_t>(v >> 32));
    }
  }
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 4 bits -> 4 bytes of 0/1 (bit i lands in byte i)
__device__ __forceinline__ uint32_t spread4(uint32_t b) { return ((b & 15u) * 0x00204081u) & 0x01010101u; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));  // lowers to one global_store_dwordx4 on gfx950

// Wave-cooperative expansion of nvalid bit-packed rows (L bits each, `row_stride` words apart in LDS)
// into the row-major [nvalid, L] int8 0/1 matrix at `out`.
// The chunk grid is [row][L/16]: lane -> (row g, chunk l) reads its 16 bits with ONE ds_read_u16 and
// writes ONE 16-byte store at out + g*L + 16*l. Rows are L bytes apart, so the stores are only
// byte-aligned, but a wave still covers contiguous spans of each row; no chunk ever straddles rows.
// The L%16 trailing bytes of every row are written by a second, short pass.
template <int L>
__device__ __forceinline__ void expand_rows(const uint32_t* bits, int row_stride, int nvalid, int8_t* out, int lane) {
  constexpr int CPR = L / 16, TAIL = L % 16;
  if constexpr (CPR > 0) {
    const int chunks = nvalid * CPR;
    // lane -> chunks lane, lane+64, ...: (row g, chunk l) advance incrementally (no division per chunk): +64
    // chunks = +Q rows and +Rm chunks with one carry
    constexpr int Q = 64 / CPR, Rm = 64 % CPR;
    int g = lane / CPR, l = lane - g * CPR;
    for (int c0 = lane; c0 < chunks; c0 += 256) {
      uint32_t v[4];
      int off[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        off[u] = g * L + 16 * l;
        v[u] = (c0 + 64 * u) < chunks ? reinterpret_cast<const uint16_t*>(bits + g * row_stride)[l] : 0u;
        g += Q;
        l += Rm;
        if (l >= CPR) { l -= CPR; ++g; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (c0 + 64 * u < chunks) {
          u32x4 o;
          o.x = spread4(v[u]);
          o.y = spread4(v[u] >> 4);
          o.z = spread4(v[u] >> 8);
          o.w = spread4(v[u] >> 12);
          __builtin_nontemporal_store(o, reinterpret_cast<u32x4_unaligned*>(out + off[u]));
        }
      }
    }
  }
  if constexpr (TAIL > 0) {
    const int n = nvalid * TAIL;
    for (int e = lane; e < n; e += 64) {
      const int g = e / TAIL, j = e - g * TAIL, k = CPR * 16 + j;
      out[g * L + k] = static_cast<int8_t>((bits[g * row_stride + (k >> 5)] >> (k & 31)) & 1u);
    }
  }
}

// The selection rule of hb_policy_select for ONE game, in the lane that owns it (same arithmetic, same draws: bit-identical
// actions): q row and legal row fetched as 16-byte / 4-byte vectors when A allows it.
template <int A>
__device__ __forceinline__ int select_action(const EnvArgs& a, long long gi) {
  float qv[A];
  uint32_t lw[(A + 3) / 4];
  const float* qr = a.sel_q + gi * A;
  const int8_t* lr = a.sel_legal + gi * A;
  if constexpr (A % 4 == 0) {
#pragma unroll
    for (int i = 0; i < A / 4; ++i) {
      const float4 v = reinterpret_cast<const float4*>(qr)[i];
      qv[4 * i] = v.x; qv[4 * i + 1] = v.y; qv[4 * i + 2] = v.z; qv[4 * i + 3] = v.w;
      lw[i] = reinterpret_cast<const uint32_t*>(lr)[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < (A + 3) / 4; ++i) lw[i] = 0;
#pragma unroll
    for (int i = 0; i < A; ++i) {
      qv[i] = qr[i];
      lw[i >> 2] |= static_cast<uint32_t>(static_cast<uint8_t>(lr[i])) << (8 * (i & 3));
    }
  }
  const unsigned long long gid = static_cast<unsigned long long>(a.sel_first_gid + gi);
  uint32_t r[4];
  philox4x32_10(static_cast<uint32_t>(a.sel_draw), static_cast<uint32_t>(a.sel_draw >> 32), static_cast<uint32_t>(gid),
                static_cast<uint32_t>(gid >> 32), static_cast<uint32_t>(a.sel_seed), static_cast<uint32_t>(a.sel_seed >> 32), r);
  float best = -INFINITY;
  unsigned long long legal_mask = 0, ties = 0;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    const float v = qv[i];
    if ((lw[i >> 2] >> (8 * (i & 3))) & 0xFFu) {
      legal_mask |= 1ull << i;
      if (v > best) { best = v; ties = 1ull << i; }
      else if (v == best) ties |= 1ull << i;
    }
  }
  const float u = static_cast<float>(r[0] >> 8) * (1.0f / 16777216.0f);
  unsigned long long pool = (u < a.sel_eps) ? legal_mask : ties;
  if (pool == 0) pool = legal_mask;
  int pick = 0;
  const int c = __popcll(pool);
  if (c > 0) {
    int k = static_cast<int>(__umulhi(r[1], static_cast<uint32_t>(c)));
    while (k-- > 0) pool &= pool - 1;
    pick = __ffsll(static_cast<long long>(pool)) - 1;
  }
  a.sel_actions[gi] = pick;
  return pick;
}

// Wavefronts of a workgroup never share data here, so phases are ordered with a WAVE-level
// barrier only: LDS operations of one wavefront execute in issue order, the fences keep the
// compiler from moving LDS accesses across the phase boundary, and no s_barrier couples a wave
// to a slower neighbour (one that is re-dealing games).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class K, int G>
__global__ __launch_bounds__(256) void env_kernel(const EnvArgs a) {
  constexpr int P = K::P, C = K::C, R = K::R, H = K::H;
  // (Round 3 measured an encoder on all 64 lanes — 64 / G lanes per game, the knowledge slots and the other hands dealt out
  //  over them and OR-ed into the bit row with ds_or_b32 — against this lane-per-game one: bit-identical and SLOWER, 10.5 vs
  //  9.7 us at G = 16 and 10.8 vs 9.2 us at G = 32 (32 768 games): the LDS atomics and the two extra phases cost more than the
  //  ~25 % of the encoder's instructions they save. DESIGN section 4.)
  constexpr int LPG = 1;
  constexpr int PER_WAVE = G * (K::SWP + K::NWP + K::LW);
  __shared__ uint32_t lds[4 * PER_WAVE];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint32_t* const srow = lds + wave * PER_WAVE;
  uint32_t* const obits = srow + G * K::SWP;
  uint32_t* const lbits = obits + G * K::NWP;
  const long long g0 = (static_cast<long long>(blockIdx.x) * 4 + wave) * G;
  const long long left = a.n - g0;
  const int nvalid = left <= 0 ? 0 : (left < G ? static_cast<int>(left) : G);
  const int mode = a.mode;
  HB_STAMP_REAL(10);
  HB_STAMP(0);

  // the per-game inputs are requested first so their HBM latency hides under the state copy
  const bool active = lane < nvalid && lane < G;
  const long long gi = g0 + lane;
  int uid_in = 0;
  bool mask_in = true;
  // This game's pre-shuffled next deck (64 B), used only on a re-deal. Round 3: fetched only by games that CAN end with this
  // move (one life left, deck empty, or one card short of a perfect score: ~1.5 % of the games re-deal in a step, rounds 1-2
  // fetched the row for all of them: 2.1 MB of 14.8 MB per launch) — the request goes out as soon as the state row is in LDS
  // and is in flight while the rules run; a masked reset knows at once.
  uint4 nd[4] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  const uint4* const ndp = reinterpret_cast<const uint4*>(a.next_deck + gi * NEXT_DECK_BYTES);
  if (active) {
    if (mode == MODE_STEP) uid_in = a.sel_q ? select_action<K::A>(a, gi) : a.actions[gi];
    if (mode == MODE_RESET && a.mask) mask_in = a.mask[gi] != 0;
    if ((mode == MODE_RESET && mask_in) || (mode == MODE_STEP && (a.flags & 0x100))) {   // (0x100: measurement aid, fetch always)
      nd[0] = ndp[0]; nd[1] = ndp[1]; nd[2] = ndp[2]; nd[3] = ndp[3];
    }
  }

  // ---- phase 1: state rows HBM -> LDS (coalesced 16-byte loads) ---------------------------
  {
    // every 16-byte piece of the wave's G rows is requested before the first one is written to LDS: ONE memory round trip
    // for the whole copy (a run-time-bounded loop waits for each piece in turn: G / 8 round trips)
    constexpr int Q = K::SW / 4;
    constexpr int ROUNDS = (G * Q + 63) / 64;
    const uint4* src = reinterpret_cast<const uint4*>(a.state + g0 * K::SW);
    uint4 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int e = lane + 64 * r;
      v[r] = e < nvalid * Q ? src[e] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int e = lane + 64 * r;
      if (e < nvalid * Q) {
        const int g = e / Q, q = e - g * Q;
        uint32_t* d = srow + g * K::SWP + 4 * q;
        d[0] = v[r].x; d[1] = v[r].y; d[2] = v[r].z; d[3] = v[r].w;
      }
    }
  }
  HB_STAMP(1);
  wave_sync();
  HB_STAMP(2);

  // ---- phase 2a: rules, one lane per game ---------------------------------------------------
  uint32_t* const row = srow + lane * K::SWP;
  uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w6 = 0;
  uint64_t accw = 0, disc = 0;
  float out_reward = 0.f;
  int out_term = 0, out_score = 0;
  bool illegal = false, need_reset = false, keep_seats = true;
  int reset_start = 0;

  auto hand_n = [&](int p) -> int { return (w1 >> (15 + 3 * p)) & 7; };
  auto fw = [&](int c) -> int { return (w1 >> (3 * c)) & 7; };
  auto score_now = [&]() -> int {
    if (((w0 >> 10) & 7) == 0) return 0;
    int s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += fw(c);
    return s;
  };

  if (active) {
    w0 = row[0]; w1 = row[1]; w2 = row[2]; w3 = row[3]; w6 = row[6];
    accw = (static_cast<uint64_t>(row[5]) << 32) | row[4];
    disc = (static_cast<uint64_t>(row[9]) << 32) | row[8];
    if (mode == MODE_RESET) {
      need_reset = mask_in;
      keep_seats = false;
      reset_start = a.start_player;
    } else if (mode == MODE_STEP && ((w0 >> 19) & 3) == 0) {
      {   // can this move end the game? (lives run out / the last firework card / the last turn after the deck ran dry)
        int fsum = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) fsum += fw(c);
        if (!(a.flags & 0x100) && (((w0 >> 10) & 7) <= 1 || (w0 & 63) == 0 || fsum >= C * R - 1)) {
          nd[0] = ndp[0]; nd[1] = ndp[1]; nd[2] = ndp[2]; nd[3] = ndp[3];
        }
      }
      // The four move types are evaluated WITHOUT divergent branches: every effect is computed under a 0/1
      // predicate and selected in. (The branchy form spent ~60 % of this phase's instructions on exec-mask
      // bookkeeping and register copies at the joins.)
      const int deck_size = w0 & 63, info = (w0 >> 6) & 15, life = (w0 >> 10) & 7;
      const int s = (w0 >> 13) & 7;
      const int turns = (w0 >> 16) & 7, moves = (w0 >> 21) & 255;
      const bool in_range = uid_in >= 0 && uid_in < K::A;
      const int uid = in_range ? uid_in : 0;  // an out-of-range uid is rejected below; keep every derived index in bounds
      const bool is_discard = uid < H, is_play = uid >= H && uid < 2 * H;
      const bool is_rc = uid >= 2 * H && uid < 2 * H + (P - 1) * C, is_rr = uid >= 2 * H + (P - 1) * C;
      const bool card_move = is_discard || is_play, is_hint = !card_move;
      const int type = is_play ? MV_PLAY : (is_discard ? MV_DISCARD : (is_rc ? MV_RCOLOR : MV_RRANK));
      const int xc = uid - 2 * H, xr = uid - 2 * H - (P - 1) * C;
      const int ci = is_discard ? uid : (is_play ? uid - H : 0);
      const int toff = is_rc ? 1 + xc / C : (is_rr ? 1 + xr / R : 0);
      const int hcol = is_rc ? xc % C : 0, hrank = is_rr ? xr % R : 0;
      int t = s + toff;
      if (t >= P) t -= P;
      const int n_s = hand_n(s), n_t = hand_n(t);
      const uint32_t hs = row[K::W_HANDS + s], ht = row[K::W_HANDS + t];
      const uint64_t kn_s = (static_cast<uint64_t>(row[K::W_KNOW + 2 * s + 1]) << 32) | row[K::W_KNOW + 2 * s];
      const uint64_t kn_t = (static_cast<uint64_t>(row[K::W_KNOW + 2 * t + 1]) << 32) | row[K::W_KNOW + 2 * t];
      const uint8_t* deckb = reinterpret_cast<const uint8_t*>(row + K::W_DECK);
      const uint32_t card_new = deckb[deck_size > 0 ? K::D - deck_size : 0];
      // cards of the target hand that the hint touches
      uint32_t match = 0;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const int card = (ht >> (5 * i)) & 31;
        const bool m = is_rc ? (card / R == hcol) : (card % R == hrank);
        match |= static_cast<uint32_t>(is_hint && i < n_t && m) << i;
      }
      const bool ok = in_range && (is_discard ? (info < K::INFO && ci < n_s) : (is_play ? ci < n_s : (info > 0 && match != 0)));
      illegal = !ok;
      // ---- card moves: discard / play slot ci of the mover's hand
      const int card = (hs >> (5 * ci)) & 31;
      const int c_col = card / R, c_rank = card % R;
      const bool success = is_play && fw(c_col < C ? c_col : 0) == c_rank && c_col < C;
      const bool fail = is_play && !success;
      const bool stack_done = success && c_rank == R - 1;
      const int la_info = ((is_discard || stack_done) && info < K::INFO) ? 1 : 0;
      const bool to_discard = is_discard || fail;
      const uint32_t lowm = (1u << (5 * ci)) - 1u;
      uint32_t hs_new = (hs & lowm) | ((hs >> (5 * (ci + 1))) << (5 * ci)) | (31u << 20);
      const uint64_t klow = (1ull << (12 * ci)) - 1ull;
      uint64_t kn_s_new = (kn_s & klow) | ((kn_s >> (12 * (ci + 1))) << (12 * ci));
      const bool draws = deck_size > 0;  // replacement deal (A.5 step 6): only the mover's hand can be short
      const int slot_new = n_s > 0 ? n_s - 1 : 0;
      if (draws) {
        hs_new = (hs_new & ~(31u << (5 * slot_new))) | (card_new << (5 * slot_new));
        kn_s_new |= static_cast<uint64_t>(K::ALL_PLAUSIBLE) << (12 * slot_new);
      }
      // ---- hints: knowledge of every card in the target hand
      uint64_t kn_t_new = kn_t;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const bool m = (match >> i) & 1u;
        uint64_t k = (kn_t >> (12 * i)) & 0xFFFull;
        const uint64_t k_rc = m ? ((k & ~0x1Full) | (1ull << hcol) | (1ull << 10)) : (k & ~(1ull << hcol));
        const uint64_t k_rr = m ? ((k & ~(0x1Full << 5)) | (1ull << (5 + hrank)) | (1ull << 11)) : (k & ~(1ull << (5 + hrank)));
        k = is_rc ? k_rc : k_rr;
        if (i < n_t) kn_t_new = (kn_t_new & ~(0xFFFull << (12 * i))) | (k << (12 * i));
      }
      if (ok) {
        // one pair of LDS stores for whichever hand changed (the mover's cards, or the target's knowledge)
        const int pw = card_move ? s : t;
        const uint64_t kn_w = card_move ? kn_s_new : kn_t_new;
        if (card_move) row[K::W_HANDS + s] = hs_new;
        row[K::W_KNOW + 2 * pw] = static_cast<uint32_t>(kn_w);
        row[K::W_KNOW + 2 * pw + 1] = static_cast<uint32_t>(kn_w >> 32);
        // per-seat bookkeeping: this seat now has an open transition
        w3 = (w3 | (1u << s)) & ~(1u << (5 + s));
        accw &= ~(0xFFull << (8 * s));
        const int before = score_now();
        const int n_s_after = card_move ? (draws ? n_s : n_s - 1) : n_s;
        w1 = (w1 & ~(7u << (15 + 3 * s))) | (static_cast<uint32_t>(n_s_after) << (15 + 3 * s));
        if (success) w1 += 1u << (3 * c_col);
        {  // discard pile = the observation's discard section itself (one thermometer per card identity): the
           // next bit of this card's field is (field << 1 | 1), clipped to the field's width
          const int dpos = (c_col * K::CPC + (c_rank ? 2 * c_rank + 1 : 0)) & 63;  // cum(rank) = 2*rank + 1 past rank 0
          const uint32_t dmask = c_rank == 0 ? 7u : (c_rank == R - 1 ? 1u : 3u);
          const uint32_t dfield = static_cast<uint32_t>(disc >> dpos) & dmask;
          const uint32_t dnew = to_discard ? (((dfield << 1) | 1u) & dmask) : 0u;
          disc |= static_cast<uint64_t>(dnew) << dpos;
        }
        const int info2 = info + (card_move ? la_info : -1);
        const int life2 = life - (fail ? 1 : 0);
        const int deck2 = deck_size - ((card_move && draws) ? 1 : 0);
        const int turns2 = turns - (deck_size == 0 ? 1 : 0);
        const int la_color = card_move ? c_col : hcol, la_rank = card_move ? c_rank : hrank;
        w2 = 1u | (static_cast<uint32_t>(s) << 1) | (static_cast<uint32_t>(type) << 4) | (static_cast<uint32_t>(ci) << 6) |
             (static_cast<uint32_t>(toff) << 9) | (static_cast<uint32_t>(la_color) << 12) |
             (static_cast<uint32_t>(la_rank) << 15) | (static_cast<uint32_t>(success ? 1 : 0) << 18) |
             (static_cast<uint32_t>(card_move ? la_info : 0) << 19) | (match << 20);
        int cur = s + 1;
        if (cur >= P) cur = 0;
        w0 = static_cast<uint32_t>(deck2) | (static_cast<uint32_t>(info2) << 6) | (static_cast<uint32_t>(life2) << 10) |
             (static_cast<uint32_t>(cur) << 13) | (static_cast<uint32_t>(turns2 & 7) << 16) |
             (static_cast<uint32_t>((moves + 1) & 255) << 21);
        const int after = score_now();
        int status = 0;
        if (life2 < 1) status = 1;
        else if (after >= C * R) status = 2;
        else if (turns2 <= 0) status = 3;
        w0 |= static_cast<uint32_t>(status) << 19;
        int r = after - before;
        if ((a.flags & 4) && r < 0) r = 0;
        out_reward = static_cast<float>(r);
        out_term = status != 0;
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const bool live = ((w3 >> q) & 1u) && !((w3 >> (5 + q)) & 1u);
          const int8_t v = static_cast<int8_t>(static_cast<int8_t>(accw >> (8 * q)) + (live ? r : 0));
          accw = (accw & ~(0xFFull << (8 * q))) | (static_cast<uint64_t>(static_cast<uint8_t>(v)) << (8 * q));
          if (live && out_term) w3 |= 1u << (5 + q);
        }
        if (out_term && (a.flags & 1)) {
          need_reset = true;
          reset_start = (a.flags & 2) ? cur : 0;
        }
      }
    }
    out_score = score_now();
  }

  HB_STAMP(3);
  // ---- phase 2b: a finished / masked game takes its pre-shuffled next deck (lane-local copy into the LDS row);
  // the shuffle itself happens off this kernel's critical path in refill_kernel -----------------------------
  if (active && need_reset) {
    const uint32_t* ndw = reinterpret_cast<const uint32_t*>(nd);
#pragma unroll
    for (int i = 0; i < 13; ++i) row[K::W_DECK + i] = ndw[i];
    a.refill[gi] = 1;
  }
  HB_STAMP(4);
  wave_sync();
  HB_STAMP(5);

  // ---- phase 2c: deal fresh games, encode, publish fields -----------------------------------
  if (active) {
    if (need_reset) {
      const uint8_t* deckb = reinterpret_cast<const uint8_t*>(row + K::W_DECK);
      w0 = static_cast<uint32_t>(K::D - P * H) | (static_cast<uint32_t>(K::INFO) << 6) | (static_cast<uint32_t>(K::LIFE) << 10) |
           (static_cast<uint32_t>(reset_start) << 13) | (static_cast<uint32_t>(P) << 16);
      w1 = 0;
      w2 = 0;
      if (!keep_seats) { w3 = 0; accw = 0; }
      w6 += 1;
      disc = 0;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        uint32_t hc = 0;
        uint64_t kn = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          if (i < H) {
            hc |= static_cast<uint32_t>(deckb[p * H + i]) << (5 * i);
            kn |= static_cast<uint64_t>(K::ALL_PLAUSIBLE) << (12 * i);
          } else {
            hc |= 31u << (5 * i);
          }
        }
        row[K::W_HANDS + p] = hc;
        row[K::W_KNOW + 2 * p] = static_cast<uint32_t>(kn);
        row[K::W_KNOW + 2 * p + 1] = static_cast<uint32_t>(kn >> 32);
        w1 |= static_cast<uint32_t>(H) << (15 + 3 * p);
      }
    }
    if (mode != MODE_OBSERVE) {
      row[0] = w0; row[1] = w1; row[2] = w2; row[3] = w3;
      row[4] = static_cast<uint32_t>(accw); row[5] = static_cast<uint32_t>(accw >> 32);
      row[6] = w6; row[7] = 0;
      row[8] = static_cast<uint32_t>(disc); row[9] = static_cast<uint32_t>(disc >> 32);
    }

    if (mode != MODE_RESET) {
      const int o = (w0 >> 13) & 7;
      if constexpr (LPG == 1) {
        const int deck_size = w0 & 63, info = (w0 >> 6) & 15, life = (w0 >> 10) & 7;
        BitAcc<K::NW> acc;
        acc.clear();
        uint64_t legal = 0;
        const int n_o = hand_n(o);
        // 1. other players' hands + "hand is short" flags; hint legality falls out of the same pass
        static_for<P>([&](auto REL) {
          constexpr int rel = decltype(REL)::value;
          int p = o + rel;
          if (p >= P) p -= P;
          const int n_p = hand_n(p);
          acc.template put<K::FLAGS_OFF + rel, 1>(n_p < H ? 1u : 0u);
          if constexpr (rel >= 1) {
            const uint32_t hc = row[K::W_HANDS + p];
            uint32_t cmask = 0, rmask = 0;
            static_for<H>([&](auto I) {
              constexpr int i = decltype(I)::value;
              const int card = (hc >> (5 * i)) & 31;
              const bool have = i < n_p;
              acc.template put<((rel - 1) * H + i) * K::BITS, K::BITS>(have ? (1u << card) : 0u);
              if (have) { cmask |= 1u << (card / R); rmask |= 1u << (card % R); }
            });
            if (info > 0) {
              legal |= static_cast<uint64_t>(cmask) << (2 * H + (rel - 1) * C);
              legal |= static_cast<uint64_t>(rmask) << (2 * H + (P - 1) * C + (rel - 1) * R);
            }
          }
        });
        const uint32_t own = (1u << n_o) - 1u;
        if (info < K::INFO) legal |= own;
        legal |= static_cast<uint64_t>(own) << H;
        // 2. board
        acc.template put64<K::BOARD_OFF, K::DECK_T>((1ull << deck_size) - 1ull);
        static_for<C>([&](auto CI) {
          constexpr int c = decltype(CI)::value;
          acc.template put<K::FW_OFF + c * R, R>((1u << fw(c)) >> 1);
        });
        acc.template put<K::INFO_OFF, K::INFO>((1u << info) - 1u);
        acc.template put<K::LIFE_OFF, K::LIFE>((1u << life) - 1u);
        // 3. discards: one thermometer per card identity,
        acc.template put64<K::DISC_OFF, K::D>(disc);  // kept in this very form in the state row
        // 4. most recent move, observer-relative
        {
          const uint32_t valid = w2 & 1u;
          const int la_player = (w2 >> 1) & 7, la_type = (w2 >> 4) & 3, la_ci = (w2 >> 6) & 7, la_toff = (w2 >> 9) & 7;
          const int la_color = (w2 >> 12) & 7, la_rank = (w2 >> 15) & 7;
          const uint32_t la_scored = (w2 >> 18) & 1u, la_info = (w2 >> 19) & 1u, la_mask = (w2 >> 20) & 31u;
          int actor = la_player - o;
          if (actor < 0) actor += P;
          int target = actor + la_toff;
          if (target >= P) target -= P;
          const uint32_t reveal = valid & static_cast<uint32_t>(la_type >= MV_RCOLOR);
          const uint32_t cardmv = valid & static_cast<uint32_t>(la_type <= MV_DISCARD);
          const uint32_t is_rc = valid & static_cast<uint32_t>(la_type == MV_RCOLOR);
          const uint32_t is_rr = valid & static_cast<uint32_t>(la_type == MV_RRANK);
          const uint32_t is_play = valid & static_cast<uint32_t>(la_type == MV_PLAY);
          constexpr int o1 = K::LA_OFF, o2 = o1 + P, o3 = o2 + 4, o4 = o3 + P, o5 = o4 + C, o6 = o5 + R, o7 = o6 + H,
                        o8 = o7 + H, o9 = o8 + K::BITS;
          acc.template put<o1, P>(valid << actor);
          acc.template put<o2, 4>(valid << la_type);
          acc.template put<o3, P>(reveal << target);
          acc.template put<o4, C>(is_rc << la_color);
          acc.template put<o5, R>(is_rr << la_rank);
          acc.template put<o6, H>(reveal ? la_mask : 0u);
          acc.template put<o7, H>(cardmv << la_ci);
          acc.template put<o8, K::BITS>(cardmv << (la_color * R + la_rank));
          acc.template put<o9, 2>((is_play & la_scored) | ((is_play & la_info) << 1));
        }
        // 5. card knowledge, observer first
        static_for<P>([&](auto REL) {
          constexpr int rel = decltype(REL)::value;
          int p = o + rel;
          if (p >= P) p -= P;
          const int n_p = hand_n(p);
          const uint64_t kn = (static_cast<uint64_t>(row[K::W_KNOW + 2 * p + 1]) << 32) | row[K::W_KNOW + 2 * p];
          static_for<H>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const uint32_t k = static_cast<uint32_t>(kn >> (12 * i)) & 0xFFFu;
            const uint32_t cp = k & 31u, rp = (k >> 5) & 31u;
            const uint32_t plaus = __umul24(spread_colors<C, R>(cp), rp);  // disjoint R-bit fields: no carries
            const uint32_t ch = (k >> 10) & 1u, rh = (k >> 11) & 1u;
            uint64_t v = plaus | (static_cast<uint64_t>(ch ? cp : 0u) << K::BITS) |
                         (static_cast<uint64_t>(rh ? rp : 0u) << (K::BITS + C));
            if (i >= n_p) v = 0;
            acc.template put64<K::KN_OFF + (rel * H + i) * K::KN_SLOT, K::KN_SLOT>(v);
          });
        });
        uint32_t* ob = obits + lane * K::NWP;
  #pragma unroll
        for (int i = 0; i < K::NW; ++i) ob[i] = acc.w[i];
  #pragma unroll
        for (int i = K::NW; i < K::NWP; ++i) ob[i] = 0;
        uint32_t* lb = lbits + lane * K::LW;
        lb[0] = static_cast<uint32_t>(legal);
        lb[1] = static_cast<uint32_t>(legal >> 32);
        lb[2] = 0;
      }
      // per-game scalars (coalesced: consecutive lanes, consecutive addresses)
      const uint32_t pend = (w3 >> o) & 1u, tsin = (w3 >> (5 + o)) & 1u;
      if (a.agent_reward) a.agent_reward[gi] = pend ? static_cast<float>(static_cast<int8_t>(accw >> (8 * o))) : 0.f;
      if (a.agent_step_type) a.agent_step_type[gi] = static_cast<int8_t>(!pend ? 0 : (tsin ? 2 : 1));
      if (mode == MODE_STEP) {
        if (a.reward) a.reward[gi] = out_reward;
        if (a.terminal) a.terminal[gi] = static_cast<int8_t>(out_term);
        if (a.score) a.score[gi] = static_cast<int8_t>(out_score);
      }
    }
  }
  if (mode == MODE_STEP) {
    const unsigned long long bad = __ballot(illegal);
    if (bad && lane == 0) atomicAdd(a.illegal, static_cast<unsigned long long>(__popcll(bad)));
    // running episode statistics: every wavefront owns one slot, so the no-return atomics never contend (thousands
    // of atomics on ONE address cost ~12 ns each and showed up as +6 us on the kernel; a plain load-add-store
    // would put a dependent HBM round trip at the very end of the wave)
    const unsigned long long ended = __ballot(out_term != 0);
    if (ended) {
      int sc = out_term ? out_score : 0;
      for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
      if (lane == 0) {
        unsigned long long* slot = a.stats + 2 * (static_cast<long long>(blockIdx.x) * 4 + wave);
        atomicAdd(slot, static_cast<unsigned long long>(__popcll(ended)));
        atomicAdd(slot + 1, static_cast<unsigned long long>(sc));
      }
    }
  }
  HB_STAMP(6);
  wave_sync();
  HB_STAMP(7);

  // ---- phase 3: LDS -> HBM, full-width coalesced stores -------------------------------------
  if (mode != MODE_OBSERVE) {
    constexpr int Q = K::SW / 4;
    uint4* dst = reinterpret_cast<uint4*>(a.state + g0 * K::SW);
    for (int e = lane; e < nvalid * Q; e += 64) {
      const int g = e / Q, q = e - g * Q;
      const uint32_t* s = srow + g * K::SWP + 4 * q;
      uint4 v;
      v.x = s[0]; v.y = s[1]; v.z = s[2]; v.w = s[3];
      dst[e] = v;
    }
  }
  HB_STAMP(8);
  if (mode != MODE_RESET) {
    if (a.obs_bits) {
      // packed form: the wave's nvalid x NW words are one contiguous span of HBM (rows are NW words, no padding)
      uint32_t* dst = a.obs_bits + g0 * K::NW;
      for (int e = lane; e < nvalid * K::NW; e += 64) {
        const int g = e / K::NW, i = e - g * K::NW;
        dst[e] = obits[g * K::NWP + i];
      }
    }
    if (a.obs) expand_rows<K::OBS_LEN>(obits, K::NWP, nvalid, a.obs + g0 * K::OBS_LEN, lane);
    expand_rows<K::A>(lbits, K::LW, nvalid, a.legal + g0 * K::A, lane);
  }
  HB_STAMP(9);
  HB_STAMP_REAL(11);
}

// Deck pool refill: regenerates next_deck[g] for every game whose flag is set. One wavefront scans 64 games
// and shuffles the flagged ones one at a time with all lanes: lane j draws Philox4x32-10(j, episode, game id;
// seed), key = 26 random bits | j (all distinct); its rank among the D keys comes from D v_readlane
// broadcasts + compare + add-carry; card j of the canonical deck lands at position rank. `episode` is the
// game's deal counter (state word 6), so deck(g, e) is a pure function of (seed, global game id, e).
template <class K>
__global__ __launch_bounds__(256) void refill_kernel(const EnvArgs a) {
  constexpr int GR = 16;  // games scanned per wavefront: few enough that a wave rarely shuffles more than one deck
  const int lane = threadIdx.x & 63;
  const long long g0 = (static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * GR;
  if (g0 >= a.n) return;
  const long long gi = g0 + lane;
  const bool mine = lane < GR && gi < a.n && a.refill[gi] != 0;
  unsigned long long todo = __ballot(mine);
  while (todo) {
    const int src = __ffsll(static_cast<long long>(todo)) - 1;
    todo &= todo - 1;
    const long long g = g0 + src;
    uint8_t* dst = a.next_deck + g * NEXT_DECK_BYTES;
    if (a.decks) {
      if (lane < K::D) dst[lane] = a.decks[g * K::D + lane];
    } else {
      const uint32_t episode = a.state[g * K::SW + 6];
      const unsigned long long gid = static_cast<unsigned long long>(a.first_gid + g);
      uint32_t out[4];
      philox4x32_10(static_cast<uint32_t>(lane), episode, static_cast<uint32_t>(gid), static_cast<uint32_t>(gid >> 32),
                    static_cast<uint32_t>(a.seed), static_cast<uint32_t>(a.seed >> 32), out);
      const uint32_t key = (out[0] & ~63u) | static_cast<uint32_t>(lane);
      int rank = 0;
      static_for<K::D>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const uint32_t ki = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(key), i));
        rank += ki < key ? 1 : 0;
      });
      if (lane < K::D) {
        const int col = lane / K::CPC, k = lane - col * K::CPC;
        const int rk = k < 3 ? 0 : 1 + (k - 3) / 2;
        dst[rank] = static_cast<uint8_t>(col * K::R + rk);
      }
    }
  }
  if (mine) a.refill[gi] = 0;
}

template <class K>
void launch_refill(const EnvArgs& a, hipStream_t stream) {
  const unsigned blocks = static_cast<unsigned>((a.n + 63) / 64);  // 4 wavefronts x 16 games per workgroup
  if (blocks == 0) return;
  hipLaunchKernelGGL((refill_kernel<K>), dim3(blocks), dim3(256), 0, stream, a);
}

template <class K, int G>
void launch_env(const EnvArgs& a, hipStream_t stream) {
  const long long per_block = 4LL * G;
  const unsigned blocks = static_cast<unsigned>((a.n + per_block - 1) / per_block);
  if (blocks == 0) return;
  if (a.ev_start && a.ev_stop)
    hipExtLaunchKernelGGL((env_kernel<K, G>), dim3(blocks), dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, a);
  else
    hipLaunchKernelGGL((env_kernel<K, G>), dim3(blocks), dim3(256), 0, stream, a);
}

using LaunchFn = void (*)(const EnvArgs&, hipStream_t);
struct EnvVariant {
  int P, C, R, H, INFO, LIFE;
  int obs_len, n_actions, deck, state_words;
  LaunchFn g8, g16, g32, g64, refill;
};

template <class K>
constexpr EnvVariant make_variant() {
  return EnvVariant{K::P, K::C, K::R, K::H, K::INFO, K::LIFE, K::OBS_LEN, K::A, K::D, K::SW,
                    &launch_env<K, 8>, &launch_env<K, 16>, &launch_env<K, 32>, &launch_env<K, 64>, &launch_refill<K>};
}

}  // namespace hb
