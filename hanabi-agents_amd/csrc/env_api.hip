// env_api.hip — host side of the C-ABI for the vectorised Hanabi env (include/hanabi_hip.h).
// Owns the HBM state rows, validates arguments, picks the compiled kernel variant and
// enqueues ONE kernel per call on the caller's stream. No CPU fallback exists: without a
// device every constructor fails with HB_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hanabi_hip.h"
#include "common.hpp"
#include "env_kernel.hpp"

namespace hb {
const EnvVariant* variants_full(int* n);
const EnvVariant* variants_small(int* n);
const EnvVariant* variants_vsmall(int* n);

thread_local std::string g_error;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

static const EnvVariant* find_variant(const hb_config* c) {
  const EnvVariant* (*tables[])(int*) = {variants_full, variants_small, variants_vsmall};
  for (auto get : tables) {
    int n = 0;
    const EnvVariant* v = get(&n);
    for (int i = 0; i < n; ++i)
      if (v[i].P == c->players && v[i].C == c->colors && v[i].R == c->ranks && v[i].H == c->hand_size &&
          v[i].INFO == c->max_info && v[i].LIFE == c->max_life)
        return &v[i];
  }
  return nullptr;
}

// pure-host size formulas, valid for any configuration in range
static int copies_of(const hb_config* c, int r) { return r == 0 ? 3 : (r == c->ranks - 1 ? 1 : 2); }
static int deck_of(const hb_config* c) {
  int per = 0;
  for (int r = 0; r < c->ranks; ++r) per += copies_of(c, r);
  return per * c->colors;
}
}  // namespace hb

struct hb_env {
  hb_config cfg;
  const hb::EnvVariant* var;
  long long n;
  unsigned long long seed;
  long long first_gid;
  uint32_t* state;
  unsigned long long* illegal;  // [0] illegal uids, then per-wavefront {episodes finished, sum of final scores}
  long long stat_slots;
  uint8_t* next_deck;  // [n, 64] deck pool: the deck each game is dealt at its next (re)deal
  uint8_t* refill;     // [n] flags: pool entry consumed / stale
  const uint8_t* decks;
  int gpw;
  int device;
  hipEvent_t ev_start, ev_stop;
  // the deck-pool refill runs on a private stream so it overlaps whatever the caller enqueues after a step
  hipStream_t side;
  hipEvent_t ev_step_done, ev_refill_done;
  bool refill_in_flight;
  bool async_refill;
  // A game cannot end in fewer than max_life moves (the quickest ending is max_life misplays in a row), so a pool deck
  // consumed by a re-deal at step t is not needed again before step t + max_life: the refill launch runs every
  // `refill_period` = min(max_life, 3) steps instead of after every step (Hanabi-Full: 1 launch in 3; -Small: every step)
  int refill_period, since_refill;
};

using hb::fail;

extern "C" {

const char* hb_last_error(void) { return hb::g_error.c_str(); }
int hb_abi_version(void) { return HB_ABI_VERSION; }

int hb_config_validate(const hb_config* c) {
  if (!c) return fail(HB_ERR_INVALID, "null config");
  if (c->players < 2 || c->players > 5) return fail(HB_ERR_INVALID, "players must be 2..5 (got %d)", c->players);
  if (c->colors < 1 || c->colors > 5) return fail(HB_ERR_INVALID, "colors must be 1..5 (got %d)", c->colors);
  if (c->ranks < 1 || c->ranks > 5) return fail(HB_ERR_INVALID, "ranks must be 1..5 (got %d)", c->ranks);
  if (c->hand_size < 1 || c->hand_size > 5) return fail(HB_ERR_INVALID, "hand_size must be 1..5 (got %d)", c->hand_size);
  if (c->max_info < 1 || c->max_info > 15) return fail(HB_ERR_INVALID, "max_info must be 1..15");
  if (c->max_life < 1 || c->max_life > 7) return fail(HB_ERR_INVALID, "max_life must be 1..7");
  if (c->players * c->hand_size > hb::deck_of(c)) return fail(HB_ERR_INVALID, "deck too small for the hands");
  return HB_OK;
}
int hb_num_actions(const hb_config* c) { return 2 * c->hand_size + (c->players - 1) * (c->colors + c->ranks); }
int hb_deck_size(const hb_config* c) { return hb::deck_of(c); }
int hb_obs_len(const hb_config* c) {
  const int bits = c->colors * c->ranks, P = c->players, H = c->hand_size, D = hb::deck_of(c);
  return (P - 1) * H * bits + P + (D - P * H + bits + c->max_info + c->max_life) + D +
         (P + 4 + P + c->colors + c->ranks + H + H + bits + 2) + P * H * (bits + c->colors + c->ranks);
}
int hb_state_words(const hb_config* c) { return c->players <= 3 ? 32 : 48; }
int hb_obs_words(const hb_config* c) { return (hb_obs_len(c) + 31) / 32; }

int hb_env_create(const hb_config* cfg, int64_t n_games, uint64_t seed, int64_t first_game_id, hb_env** out) {
  if (!out) return fail(HB_ERR_INVALID, "null out");
  *out = nullptr;
  if (int rc = hb_config_validate(cfg)) return rc;
  if (n_games <= 0) return fail(HB_ERR_INVALID, "n_games must be positive");
  const hb::EnvVariant* var = hb::find_variant(cfg);
  if (!var)
    return fail(HB_ERR_INVALID,
                "no compiled kernel for players=%d colors=%d ranks=%d hand=%d info=%d life=%d "
                "(built: Hanabi-Full / -Small / -Very-Small, 2..5 players)",
                cfg->players, cfg->colors, cfg->ranks, cfg->hand_size, cfg->max_info, cfg->max_life);
  if (var->obs_len != hb_obs_len(cfg) || var->n_actions != hb_num_actions(cfg) || var->deck != hb_deck_size(cfg))
    return fail(HB_ERR_INVALID, "internal: kernel sizes disagree with host formulas");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(HB_ERR_NO_DEVICE, "no HIP device available");
  hb_env* e = new (std::nothrow) hb_env();
  if (!e) return fail(HB_ERR_NOMEM, "out of host memory");
  e->cfg = *cfg;
  e->var = var;
  e->n = n_games;
  e->seed = seed;
  e->first_gid = first_game_id;
  e->decks = nullptr;
  e->gpw = 0;  // automatic
  e->ev_start = e->ev_stop = nullptr;
  e->side = nullptr;
  e->ev_step_done = e->ev_refill_done = nullptr;
  e->refill_in_flight = false;
  e->async_refill = false;
  e->refill_period = cfg->max_life < 3 ? cfg->max_life : 3;
  e->since_refill = 0;
  HB_HIP_OR(hipGetDevice(&e->device), delete e);
  const size_t bytes = static_cast<size_t>(n_games) * var->state_words * 4;
  HB_HIP_OR(hipMalloc(reinterpret_cast<void**>(&e->state), bytes), delete e);
  e->stat_slots = n_games / 8 + 8;  // one slot per wavefront at the smallest games-per-wave setting
  HB_HIP_OR(hipMalloc(reinterpret_cast<void**>(&e->illegal), 8 + 16 * e->stat_slots), { (void)hipFree(e->state); delete e; });
  HB_HIP_OR(hipMemset(e->state, 0, bytes), { hb_env_destroy(e); });
  HB_HIP_OR(hipMemset(e->illegal, 0, 8 + 16 * e->stat_slots), { hb_env_destroy(e); });
  HB_HIP_OR(hipMalloc(reinterpret_cast<void**>(&e->next_deck), static_cast<size_t>(n_games) * hb::NEXT_DECK_BYTES), { hb_env_destroy(e); });
  HB_HIP_OR(hipMalloc(reinterpret_cast<void**>(&e->refill), static_cast<size_t>(n_games)), { hb_env_destroy(e); });
  HB_HIP_OR(hipMemset(e->next_deck, 0, static_cast<size_t>(n_games) * hb::NEXT_DECK_BYTES), { hb_env_destroy(e); });
  HB_HIP_OR(hipMemset(e->refill, 1, static_cast<size_t>(n_games)), { hb_env_destroy(e); });  // nothing generated yet
  HB_HIP_OR(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking), { hb_env_destroy(e); });
  HB_HIP_OR(hipEventCreateWithFlags(&e->ev_step_done, hipEventDisableTiming), { hb_env_destroy(e); });
  HB_HIP_OR(hipEventCreateWithFlags(&e->ev_refill_done, hipEventDisableTiming), { hb_env_destroy(e); });
  *out = e;
  return HB_OK;
}

int hb_env_destroy(hb_env* e) {
  if (!e) return HB_OK;
  if (e->state) (void)hipFree(e->state);
  if (e->illegal) (void)hipFree(e->illegal);
  if (e->next_deck) (void)hipFree(e->next_deck);
  if (e->refill) (void)hipFree(e->refill);
  if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
  if (e->ev_step_done) (void)hipEventDestroy(e->ev_step_done);
  if (e->ev_refill_done) (void)hipEventDestroy(e->ev_refill_done);
  delete e;
  return HB_OK;
}

int64_t hb_env_num_games(const hb_env* e) { return e ? e->n : 0; }

int hb_env_set_decks(hb_env* e, const uint8_t* decks_dev) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  e->decks = decks_dev;
  e->refill_in_flight = false;
  HB_HIP(hipDeviceSynchronize());
  HB_HIP(hipMemset(e->refill, 1, static_cast<size_t>(e->n)));  // every pooled deck is stale now
  return HB_OK;
}

int hb_env_set_games_per_wave(hb_env* e, int32_t g) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if (g != 0 && g != 8 && g != 16 && g != 32 && g != 64)
    return fail(HB_ERR_INVALID, "games per wave must be 0 (automatic), 8, 16, 32 or 64");
  e->gpw = g;
  return HB_OK;
}

int hb_env_set_profile_events(hb_env* e, void* start_event, void* stop_event) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if ((start_event == nullptr) != (stop_event == nullptr)) return fail(HB_ERR_INVALID, "pass both events or neither");
  e->ev_start = static_cast<hipEvent_t>(start_event);
  e->ev_stop = static_cast<hipEvent_t>(stop_event);
  return HB_OK;
}

static void fill_common(hb_env* e, hb::EnvArgs& a) {
  a.state = e->state;
  a.decks = e->decks;
  a.next_deck = e->next_deck;
  a.refill = e->refill;
  a.illegal = e->illegal;
  a.stats = e->illegal + 1;
  a.n = e->n;
  a.seed = e->seed;
  a.first_gid = e->first_gid;
  a.flags = e->cfg.flags;
}

// the caller's stream must not touch the pool while an asynchronous refill is still running
static int join_refill(hb_env* e, void* stream) {
  if (e->refill_in_flight) {
    HB_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), e->ev_refill_done, 0));
    e->refill_in_flight = false;
  }
  return HB_OK;
}

// regenerate consumed / stale deck-pool entries (cheap when nothing is flagged), in order on `stream`
static int refill(hb_env* e, void* stream) {
  e->since_refill = 0;
  hb::EnvArgs a{};
  fill_common(e, a);
  e->var->refill(a, static_cast<hipStream_t>(stream));
  HB_HIP(hipGetLastError());
  return HB_OK;
}

// same, but forked onto the env's private stream behind everything enqueued on `stream` so far; the next call
// that needs the pool joins it (join_refill)
static int refill_async(hb_env* e, void* stream) {
  if (++e->since_refill < e->refill_period) return HB_OK;  // no game can need a second deck yet
  if (!e->async_refill) return refill(e, stream);
  HB_HIP(hipEventRecord(e->ev_step_done, static_cast<hipStream_t>(stream)));
  HB_HIP(hipStreamWaitEvent(e->side, e->ev_step_done, 0));
  if (int rc = refill(e, e->side)) return rc;
  HB_HIP(hipEventRecord(e->ev_refill_done, e->side));
  e->refill_in_flight = true;
  return HB_OK;
}

static int launch(hb_env* e, hb::EnvArgs& a, void* stream) {
  a.state = e->state;
  a.decks = e->decks;
  a.next_deck = e->next_deck;
  a.refill = e->refill;
  a.illegal = e->illegal;
  a.stats = e->illegal + 1;
  a.n = e->n;
  a.seed = e->seed;
  a.first_gid = e->first_gid;
  a.flags = e->cfg.flags;
  {   // measurement aid: HB_ENV_DECK_ALWAYS=1 restores rounds 1-2's unconditional fetch of the deck-pool row
    static const int always = [] { const char* v = getenv("HB_ENV_DECK_ALWAYS"); return v && v[0] == '1' ? 0x100 : 0; }();
    a.flags |= always;
  }
  a.ev_start = e->ev_start;
  a.ev_stop = e->ev_stop;
  // automatic: 32 games per wavefront when only the bit-packed rows leave the kernel and the batch is >= 32 768 games (one
  // wavefront per SIMD there, every phase wider), 16 otherwise (int8 rows: their stores want two wavefronts per SIMD; small
  // batches: more wavefronts). Measured r02, packed: 262 144 games 26.9 us (32) vs 36.2 us (16); 32 768 games 9.2 vs 9.7 us alone
  // and, with the selection fused in, 14.8 vs 17.5 us inside the training loop (0.142 vs 0.144 ms per step). int8 rows:
  // 13.3 (16) vs 14.2 us (32). Results do not depend on it.
  const int gpw = e->gpw ? e->gpw : ((a.obs_bits && !a.obs && e->n >= 32768) ? 32 : 16);
  hb::LaunchFn fn = gpw == 8 ? e->var->g8 : (gpw == 16 ? e->var->g16 : (gpw == 32 ? e->var->g32 : e->var->g64));
  fn(a, static_cast<hipStream_t>(stream));
  HB_HIP(hipGetLastError());
  return HB_OK;
}

static int check_out(const void* obs, const void* legal) {
  if (!obs || !legal) return fail(HB_ERR_INVALID, "obs_dev and legal_dev are required");
  if ((reinterpret_cast<uintptr_t>(obs) & 15) || (reinterpret_cast<uintptr_t>(legal) & 15))
    return fail(HB_ERR_ALIGN, "obs_dev / legal_dev must be 16-byte aligned");
  return HB_OK;
}
static int check_out_packed(const void* bits, const void* obs, const void* legal) {
  if (!bits || !legal) return fail(HB_ERR_INVALID, "obs_bits_dev and legal_dev are required");
  if ((reinterpret_cast<uintptr_t>(bits) & 15) || (reinterpret_cast<uintptr_t>(legal) & 15) || (reinterpret_cast<uintptr_t>(obs) & 15))
    return fail(HB_ERR_ALIGN, "obs_bits_dev / obs_dev / legal_dev must be 16-byte aligned");
  return HB_OK;
}

int hb_env_reset(hb_env* e, const uint8_t* mask_dev, int32_t start_player, void* stream) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if (start_player < 0 || start_player >= e->cfg.players) return fail(HB_ERR_INVALID, "start_player out of range");
  hb::EnvArgs a{};
  a.mode = hb::MODE_RESET;
  a.mask = mask_dev;
  a.start_player = start_player;
  if (int rc = join_refill(e, stream)) return rc;
  if (int rc = refill(e, stream)) return rc;  // the pool must hold deck(game, current deal counter)
  if (int rc = launch(e, a, stream)) return rc;
  return refill(e, stream);
}

int hb_env_observe(hb_env* e, int8_t* obs_dev, int8_t* legal_dev, float* agent_reward_dev, int8_t* agent_step_type_dev,
                   void* stream) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if (int rc = check_out(obs_dev, legal_dev)) return rc;
  hb::EnvArgs a{};
  a.mode = hb::MODE_OBSERVE;
  a.obs = obs_dev;
  a.legal = legal_dev;
  a.agent_reward = agent_reward_dev;
  a.agent_step_type = agent_step_type_dev;
  return launch(e, a, stream);
}

int hb_env_step(hb_env* e, const int32_t* actions_dev, int8_t* obs_dev, int8_t* legal_dev, float* reward_dev,
                int8_t* terminal_dev, float* agent_reward_dev, int8_t* agent_step_type_dev, int8_t* score_dev,
                void* stream) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if (!actions_dev) return fail(HB_ERR_INVALID, "actions_dev is required");
  if (int rc = check_out(obs_dev, legal_dev)) return rc;
  hb::EnvArgs a{};
  a.mode = hb::MODE_STEP;
  a.actions = actions_dev;
  a.obs = obs_dev;
  a.legal = legal_dev;
  a.reward = reward_dev;
  a.terminal = terminal_dev;
  a.agent_reward = agent_reward_dev;
  a.agent_step_type = agent_step_type_dev;
  a.score = score_dev;
  if (int rc = join_refill(e, stream)) return rc;
  if (int rc = launch(e, a, stream)) return rc;
  return refill_async(e, stream);  // re-shuffle the pool entries this step consumed, off the caller's stream
}

int hb_env_observe_packed(hb_env* e, uint32_t* obs_bits_dev, int8_t* obs_dev, int8_t* legal_dev, float* agent_reward_dev,
                          int8_t* agent_step_type_dev, void* stream) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if (int rc = check_out_packed(obs_bits_dev, obs_dev, legal_dev)) return rc;
  hb::EnvArgs a{};
  a.mode = hb::MODE_OBSERVE;
  a.obs = obs_dev;
  a.obs_bits = obs_bits_dev;
  a.legal = legal_dev;
  a.agent_reward = agent_reward_dev;
  a.agent_step_type = agent_step_type_dev;
  return launch(e, a, stream);
}

int hb_env_step_packed(hb_env* e, const int32_t* actions_dev, uint32_t* obs_bits_dev, int8_t* obs_dev, int8_t* legal_dev,
                       float* reward_dev, int8_t* terminal_dev, float* agent_reward_dev, int8_t* agent_step_type_dev,
                       int8_t* score_dev, void* stream) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if (!actions_dev) return fail(HB_ERR_INVALID, "actions_dev is required");
  if (int rc = check_out_packed(obs_bits_dev, obs_dev, legal_dev)) return rc;
  hb::EnvArgs a{};
  a.mode = hb::MODE_STEP;
  a.actions = actions_dev;
  a.obs = obs_dev;
  a.obs_bits = obs_bits_dev;
  a.legal = legal_dev;
  a.reward = reward_dev;
  a.terminal = terminal_dev;
  a.agent_reward = agent_reward_dev;
  a.agent_step_type = agent_step_type_dev;
  a.score = score_dev;
  if (int rc = join_refill(e, stream)) return rc;
  if (int rc = launch(e, a, stream)) return rc;
  return refill_async(e, stream);
}

// hb_env_step_packed with the acting agent's epsilon-greedy selection fused into the kernel: the lane that owns game g picks
// its move from q_dev[g, :] and sel_legal_dev[g, :] by the rule of hb_policy_select (same Philox draws: identical actions),
// writes it to actions_out_dev[g] and applies it. One launch and one round trip of the actions less per env step.
int hb_env_step_select_packed(hb_env* e, const float* q_dev, const int8_t* sel_legal_dev, float epsilon, uint64_t seed, uint64_t draw,
                              int64_t first_game_id, int32_t* actions_out_dev, uint32_t* obs_bits_dev, int8_t* obs_dev,
                              int8_t* legal_dev, float* reward_dev, int8_t* terminal_dev, float* agent_reward_dev,
                              int8_t* agent_step_type_dev, int8_t* score_dev, void* stream) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if (!q_dev || !sel_legal_dev || !actions_out_dev) return fail(HB_ERR_INVALID, "q_dev, sel_legal_dev and actions_out_dev are required");
  if ((reinterpret_cast<uintptr_t>(q_dev) & 15) || (reinterpret_cast<uintptr_t>(sel_legal_dev) & 3))
    return fail(HB_ERR_ALIGN, "q_dev must be 16-byte and sel_legal_dev 4-byte aligned");
  if (e->var->n_actions > 64) return fail(HB_ERR_INVALID, "the fused selection handles at most 64 actions");
  if (int rc = check_out_packed(obs_bits_dev, obs_dev, legal_dev)) return rc;
  hb::EnvArgs a{};
  a.mode = hb::MODE_STEP;
  a.sel_q = q_dev;
  a.sel_legal = sel_legal_dev;
  a.sel_actions = actions_out_dev;
  a.sel_eps = epsilon;
  a.sel_seed = seed;
  a.sel_draw = draw;
  a.sel_first_gid = first_game_id;
  a.obs = obs_dev;
  a.obs_bits = obs_bits_dev;
  a.legal = legal_dev;
  a.reward = reward_dev;
  a.terminal = terminal_dev;
  a.agent_reward = agent_reward_dev;
  a.agent_step_type = agent_step_type_dev;
  a.score = score_dev;
  if (int rc = join_refill(e, stream)) return rc;
  if (int rc = launch(e, a, stream)) return rc;
  return refill_async(e, stream);
}

#ifdef HB_STAMPS
// diagnostic library only: step once with per-wavefront phase stamps written to stamps_dev
int hb_env_step_stamped(hb_env* e, const int32_t* actions_dev, int8_t* obs_dev, int8_t* legal_dev,
                        unsigned long long* stamps_dev, void* stream) {
  hb::EnvArgs a{};
  a.mode = hb::MODE_STEP;
  a.actions = actions_dev;
  a.obs = obs_dev;
  a.legal = legal_dev;
  a.stamps = stamps_dev;
  if (int rc = join_refill(e, stream)) return rc;
  if (int rc = launch(e, a, stream)) return rc;
  return refill_async(e, stream);
}
#endif

int hb_env_illegal_count(hb_env* e, int64_t* out) {
  if (!e || !out) return fail(HB_ERR_INVALID, "null argument");
  unsigned long long v = 0;
  HB_HIP(hipMemcpy(&v, e->illegal, 8, hipMemcpyDeviceToHost));
  *out = static_cast<int64_t>(v);
  return HB_OK;
}

int hb_env_stats(hb_env* e, int64_t* episodes, int64_t* score_sum) {
  if (!e || !episodes || !score_sum) return fail(HB_ERR_INVALID, "null argument");
  std::vector<unsigned long long> v(static_cast<size_t>(2 * e->stat_slots));
  HB_HIP(hipMemcpy(v.data(), e->illegal + 1, 16 * e->stat_slots, hipMemcpyDeviceToHost));
  unsigned long long ep = 0, sc = 0;
  for (long long i = 0; i < e->stat_slots; ++i) { ep += v[2 * i]; sc += v[2 * i + 1]; }
  *episodes = static_cast<int64_t>(ep);
  *score_sum = static_cast<int64_t>(sc);
  return HB_OK;
}

int hb_env_set_refill_period(hb_env* e, int32_t steps) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  if (steps < 1 || steps > e->cfg.max_life) return fail(HB_ERR_INVALID, "refill period must be 1..max_life (%d)", e->cfg.max_life);
  e->refill_period = steps;
  return HB_OK;
}

int hb_env_set_async_refill(hb_env* e, int32_t on) {
  if (!e) return fail(HB_ERR_INVALID, "null env");
  HB_HIP(hipDeviceSynchronize());
  e->refill_in_flight = false;
  e->async_refill = on != 0;
  return HB_OK;
}

int hb_env_export_state(hb_env* e, uint32_t* rows_dev, void* stream) {
  if (!e || !rows_dev) return fail(HB_ERR_INVALID, "null argument");
  HB_HIP(hipMemcpyAsync(rows_dev, e->state, static_cast<size_t>(e->n) * e->var->state_words * 4, hipMemcpyDeviceToDevice,
                        static_cast<hipStream_t>(stream)));
  return HB_OK;
}
const uint32_t* hb_env_state(hb_env* e) { return e ? e->state : nullptr; }

int hb_env_import_state(hb_env* e, const uint32_t* rows_dev, void* stream) {
  if (!e || !rows_dev) return fail(HB_ERR_INVALID, "null argument");
  HB_HIP(hipMemcpyAsync(e->state, rows_dev, static_cast<size_t>(e->n) * e->var->state_words * 4, hipMemcpyDeviceToDevice,
                        static_cast<hipStream_t>(stream)));
  // deal counters may have changed: every pooled deck is stale
  if (int rc = join_refill(e, stream)) return rc;
  HB_HIP(hipMemsetAsync(e->refill, 1, static_cast<size_t>(e->n), static_cast<hipStream_t>(stream)));
  return refill(e, stream);
}

// ---- uniform-random legal policy (bench / tests) ------------------------------------------
__global__ void random_legal_kernel(const int8_t* __restrict__ legal, long long n, int n_actions, unsigned long long seed,
                                    unsigned long long draw, long long first_gid, int32_t* __restrict__ actions) {
  const long long g = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const unsigned long long gid = static_cast<unsigned long long>(first_gid + g);
  uint32_t out[4];
  hb::philox4x32_10(static_cast<uint32_t>(draw), static_cast<uint32_t>(draw >> 32), static_cast<uint32_t>(gid),
                    static_cast<uint32_t>(gid >> 32), static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), out);
  const int8_t* row = legal + g * n_actions;
  unsigned long long m = 0;
  for (int i = 0; i < n_actions; ++i) m |= static_cast<unsigned long long>(row[i] != 0) << i;
  const int n_legal = __popcll(m);
  int pick = 0;
  if (n_legal > 0) {
    int k = static_cast<int>(__umulhi(out[0], static_cast<uint32_t>(n_legal)));
    while (k-- > 0) m &= m - 1;  // drop the k lowest set bits
    pick = __ffsll(static_cast<long long>(m)) - 1;
  }
  actions[g] = pick;
}

int hb_random_legal_actions(const int8_t* legal_dev, int64_t n_games, int32_t n_actions, uint64_t seed, uint64_t draw,
                            int64_t first_game_id, int32_t* actions_dev, void* stream) {
  if (!legal_dev || !actions_dev) return fail(HB_ERR_INVALID, "null argument");
  if (n_actions < 1 || n_actions > 64) return fail(HB_ERR_INVALID, "n_actions must be 1..64");
  if (n_games <= 0) return HB_OK;
  const unsigned blocks = static_cast<unsigned>((n_games + 255) / 256);
  hipLaunchKernelGGL(random_legal_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), legal_dev,
                     static_cast<long long>(n_games), n_actions, static_cast<unsigned long long>(seed),
                     static_cast<unsigned long long>(draw), static_cast<long long>(first_game_id), actions_dev);
  HB_HIP(hipGetLastError());
  return HB_OK;
}


// ---- stream ordering helpers for host drivers -------------------------------------------------------------------------
// The self-play driver orders its acting and learner streams with events a few times per step; these wrappers cost
// ~1 us from ctypes where the torch.cuda.Event methods cost ~8 us each (they look the current stream up first).
int hb_event_create(void** event_out) {
  if (!event_out) return fail(HB_ERR_INVALID, "null event_out");
  hipEvent_t ev = nullptr;
  HB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  *event_out = ev;
  return HB_OK;
}
int hb_event_destroy(void* event) {
  if (event) HB_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return HB_OK;
}
int hb_event_record(void* event, void* stream) {
  if (!event) return fail(HB_ERR_INVALID, "null event");
  HB_HIP(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
  return HB_OK;
}
int hb_stream_wait_event(void* stream, void* event) {
  if (!event) return fail(HB_ERR_INVALID, "null event");
  HB_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
  return HB_OK;
}

}  // extern "C"
