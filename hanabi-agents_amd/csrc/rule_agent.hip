// rule_agent.hip — the reference's rule-based partners (hanabi_agents/rule_based/ruleset.py, rule_based.py) for all
// games at once: one lane per game walks the caller's priority list of rules over the game's packed state row
//  and emits the first move a rule produces, or a random legal move (rule_based.py:13-25).
//
// A rule only reads what the player to act may see — its OWN cards enter only through their knowledge bits
// (plausible colours / ranks, hinted flags); the other hands, fireworks, discard pile, tokens and deck size are
// public. Everything is integer / bit work on ~130 bytes per game; the two probability rules divide two small
// integers in double precision so that thresholds compare exactly as the Python code (and the CPU oracle) does.
#include <hip/hip_runtime.h>

#include "../../include/hanabi_hip.h"
#include "common.hpp"
#include "env_kernel.hpp"

namespace {

struct RuleArgs {
  const uint32_t* rows;
  long long n, first_gid;
  int P, C, R, H, INFO, SW, CPC;
  int n_rules;
  hb_rule rules[HB_MAX_RULES];
  unsigned long long seed, draw;
  int32_t* actions;
  int32_t* fired;
};

// decoded view of one game from the seat of the player to act
struct Seat {
  const uint32_t* row;
  int P, C, R, H, INFO;
  int info, life, deck, s, n_own;
  uint32_t w1;       // fireworks 3 bits per colour, hand sizes 3 bits per player
  uint32_t maxfw;    // 3 bits per colour: first rank that can no longer be reached (get_max_fireworks, ruleset.py:174-196)
  int minfw;         // min(fireworks)
  uint64_t seen;     // 2 bits per card identity: copies visible to this seat (get_visible_cards, ruleset.py:60-77)
  uint64_t kn;       // own knowledge, 12 bits per slot
  uint32_t rnd[4];

  __device__ int fw(int c) const { return static_cast<int>((w1 >> (3 * c)) & 7u); }
  __device__ int mfw(int c) const { return static_cast<int>((maxfw >> (3 * c)) & 7u); }
  __device__ int hand_n(int p) const { return static_cast<int>((w1 >> (15 + 3 * p)) & 7u); }
  __device__ int seat(int off) const {
    const int p = s + off;
    return p >= P ? p - P : p;
  }
  __device__ uint32_t cards(int p) const { return row[10 + p]; }
  __device__ uint64_t know(int p) const {
    return (static_cast<uint64_t>(row[10 + P + 2 * p + 1]) << 32) | row[10 + P + 2 * p];
  }
  __device__ int copies(int r) const { return r == 0 ? 3 : (r == R - 1 ? 1 : 2); }
  __device__ int unseen(int c, int r) const { return copies(r) - static_cast<int>((seen >> (2 * (c * R + r))) & 3u); }
  __device__ bool playable(int c, int r) const { return r == fw(c); }                    // ruleset.py:32
  __device__ bool useless(int c, int r) const { return r < fw(c) || r >= mfw(c); }       // ruleset.py:35-42
  // move uids: discard i, play i, reveal colour (offset, c), reveal rank (offset, r)
  __device__ int uid_discard(int i) const { return i; }
  __device__ int uid_play(int i) const { return H + i; }
  __device__ int uid_color(int off, int c) const { return 2 * H + (off - 1) * C + c; }
  __device__ int uid_rank(int off, int r) const { return 2 * H + (P - 1) * C + (off - 1) * R + r; }
};

struct Slot {
  uint32_t cp, rp;
  bool ch, rh;
  int color, rank;  // meaningful when hinted: the single plausible value
};
__device__ Slot slot_of(uint64_t kn, int i) {
  const uint32_t k = static_cast<uint32_t>(kn >> (12 * i)) & 0xFFFu;
  Slot s;
  s.cp = k & 31u;
  s.rp = (k >> 5) & 31u;
  s.ch = (k >> 10) & 1u;
  s.rh = (k >> 11) & 1u;
  s.color = s.cp ? __ffs(static_cast<int>(s.cp)) - 1 : 0;
  s.rank = s.rp ? __ffs(static_cast<int>(s.rp)) - 1 : 0;
  return s;
}

// get_card_playability / get_probability_useless (ruleset.py:82-106, 127-150) for own slot i: over the plausible
// identities, copies not visible anywhere, and those among them that are playable (or useless)
template <bool USELESS>
__device__ double slot_probability(const Seat& v, int i) {
  const Slot k = slot_of(v.kn, i);
  int total = 0, hit = 0;
  for (int c = 0; c < v.C; ++c) {
    if (!((k.cp >> c) & 1u)) continue;
    for (int r = 0; r < v.R; ++r) {
      if (!((k.rp >> r) & 1u)) continue;
      const int left = v.unseen(c, r);
      total += left;
      if (USELESS ? v.useless(c, r) : v.playable(c, r)) hit += left;
    }
  }
  return total > 0 ? static_cast<double>(hit) / static_cast<double>(total) : 0.0;
}
template <bool USELESS>
__device__ int argmax_probability(const Seat& v, double* best) {  // np.argmax: first maximum
  int idx = 0;
  double b = -1.0;
  for (int i = 0; i < v.n_own; ++i) {
    const double p = slot_probability<USELESS>(v, i);
    if (p > b) { b = p; idx = i; }
  }
  *best = b;
  return idx;
}

__device__ int rule_play_probably_safe(const Seat& v, double threshold, bool need_lives) {  // ruleset.py:617-635
  if (v.n_own == 0) return -1;
  double p;
  const int idx = argmax_probability<false>(v, &p);
  if ((!need_lives || v.life > 1) && p >= threshold) return v.uid_play(idx);
  return -1;
}

__device__ int rule_osawa_discard(const Seat& v) {  // ruleset.py:220-280
  if (v.info == v.INFO) return -1;
  for (int i = 0; i < v.n_own; ++i) {
    const Slot k = slot_of(v.kn, i);
    if (k.ch && v.fw(k.color) == v.R) return v.uid_discard(i);
    if (k.ch && k.rh && (k.rank < v.fw(k.color) || k.rank >= v.mfw(k.color))) return v.uid_discard(i);
    if (k.rh && k.rank < v.minfw) return v.uid_discard(i);
  }
  for (int i = 0; i < v.n_own; ++i) {
    const Slot k = slot_of(v.kn, i);
    bool eventually = false;
    for (int c = 0; c < v.C; ++c)
      if ((k.cp >> c) & 1u)
        for (int r = 0; r < v.R; ++r)
          if (((k.rp >> r) & 1u) && r < v.mfw(c)) eventually = true;
    if (!eventually) return v.uid_discard(i);
  }
  return -1;
}

__device__ int rule_play_safe_card(const Seat& v) {  // ruleset.py:350-379
  for (int i = 0; i < v.n_own; ++i) {
    const Slot k = slot_of(v.kn, i);
    bool all = true;
    for (int c = 0; c < v.C; ++c)
      if ((k.cp >> c) & 1u)
        for (int r = 0; r < v.R; ++r)
          if (((k.rp >> r) & 1u) && !v.playable(c, r)) all = false;
    if (all) return v.uid_play(i);
  }
  return -1;
}

__device__ int rule_play_if_certain(const Seat& v) {  // ruleset.py:383-409
  for (int i = 0; i < v.n_own; ++i) {
    const Slot k = slot_of(v.kn, i);
    if (k.ch && k.rh && k.rank == v.fw(k.color)) return v.uid_play(i);
  }
  return -1;
}

__device__ int rule_tell_unknown(const Seat& v) {  // ruleset.py:285-308
  if (v.info <= 0 || v.P < 2) return -1;
  const int p = v.seat(1);
  const uint32_t hc = v.cards(p);
  const uint64_t kn = v.know(p);
  for (int i = 0; i < v.hand_n(p); ++i) {
    const Slot k = slot_of(kn, i);
    const int card = (hc >> (5 * i)) & 31;
    if (!k.ch) return v.uid_color(1, card / v.R);
    if (!k.rh) return v.uid_rank(1, card % v.R);
  }
  return -1;
}

__device__ int rule_tell_randomly(const Seat& v) {  // ruleset.py:314-346
  if (v.info <= 0) return -1;
  const int p = v.seat(1);
  const int n = v.hand_n(p);
  if (n == 0) return -1;
  const int i = static_cast<int>(__umulhi(v.rnd[0], static_cast<uint32_t>(n)));
  const int card = (v.cards(p) >> (5 * i)) & 31;
  return (v.rnd[1] & 1u) == 0 ? v.uid_rank(1, card % v.R) : v.uid_color(1, card / v.R);
}

__device__ int rule_tell_playable_card_outer(const Seat& v) {  // ruleset.py:413-451
  if (v.info <= 0) return -1;
  for (int off = 1; off < v.P; ++off) {
    const int p = v.seat(off);
    const uint32_t hc = v.cards(p);
    const uint64_t kn = v.know(p);
    for (int i = 0; i < v.hand_n(p); ++i) {
      const Slot k = slot_of(kn, i);
      const int card = (hc >> (5 * i)) & 31, c = card / v.R, r = card % v.R;
      if (!v.playable(c, r)) continue;
      if (!k.rh) return v.uid_rank(off, r);
      if (!k.ch) return v.uid_color(off, c);
    }
  }
  return -1;
}

__device__ int rule_tell_dispensable(const Seat& v, int min_info) {  // ruleset.py:454-514
  if (!(v.info < min_info) || v.info <= 0) return -1;
  for (int off = 1; off < v.P; ++off) {
    const int p = v.seat(off);
    const uint32_t hc = v.cards(p);
    const uint64_t kn = v.know(p);
    for (int i = 0; i < v.hand_n(p); ++i) {
      const Slot k = slot_of(kn, i);
      const int card = (hc >> (5 * i)) & 31, c = card / v.R, r = card % v.R;
      if (!k.ch && v.fw(c) == v.R) return v.uid_color(off, c);
      if (!k.rh && r < v.minfw) return v.uid_rank(off, r);
      if (r < v.fw(c)) {
        if (!k.ch && k.rh) return v.uid_color(off, c);
        if (k.ch && !k.rh) return v.uid_rank(off, r);
      }
    }
  }
  return -1;
}

__device__ int rule_tell_anyone_useless_card(const Seat& v) {  // ruleset.py:522-535
  if (v.info <= 1) return -1;
  for (int off = 1; off < v.P; ++off) {
    const int p = v.seat(off);
    const uint32_t hc = v.cards(p);
    const uint64_t kn = v.know(p);
    for (int i = 0; i < v.hand_n(p); ++i) {
      const Slot k = slot_of(kn, i);
      const int card = (hc >> (5 * i)) & 31, c = card / v.R, r = card % v.R;
      if (!v.useless(c, r)) continue;
      if (!k.ch) return v.uid_color(off, c);
      if (!k.rh) return v.uid_rank(off, r);
    }
  }
  return -1;
}

__device__ int rule_tell_playable_card(const Seat& v) {  // ruleset.py:570-594
  if (v.info <= 0) return -1;
  for (int off = 1; off < v.P; ++off) {
    const int p = v.seat(off);
    const uint32_t hc = v.cards(p);
    for (int i = 0; i < v.hand_n(p); ++i) {
      const int card = (hc >> (5 * i)) & 31, c = card / v.R, r = card % v.R;
      if (v.playable(c, r)) return (v.rnd[1] & 1u) == 0 ? v.uid_rank(off, r) : v.uid_color(off, c);
    }
  }
  return -1;
}

__device__ int rule_legal_random(const Seat& v) {  // ruleset.py:598-604 over the legal moves of SURVEY App. A.4
  const uint64_t own = (1ull << v.n_own) - 1ull;
  uint64_t m = own << v.H;
  if (v.info < v.INFO) m |= own;
  if (v.info > 0)
    for (int off = 1; off < v.P; ++off) {
      const int p = v.seat(off);
      const uint32_t hc = v.cards(p);
      for (int i = 0; i < v.hand_n(p); ++i) {
        const int card = (hc >> (5 * i)) & 31;
        m |= 1ull << v.uid_color(off, card / v.R);
        m |= 1ull << v.uid_rank(off, card % v.R);
      }
    }
  const int n_legal = __popcll(m);
  if (n_legal == 0) return 0;
  int k = static_cast<int>(__umulhi(v.rnd[3], static_cast<uint32_t>(n_legal)));
  while (k-- > 0) m &= m - 1;
  return __ffsll(static_cast<long long>(m)) - 1;
}

__global__ void __launch_bounds__(128) rule_kernel(RuleArgs a) {
  const long long g = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= a.n) return;
  Seat v;
  v.row = a.rows + g * a.SW;
  v.P = a.P; v.C = a.C; v.R = a.R; v.H = a.H; v.INFO = a.INFO;
  const uint32_t w0 = v.row[0];
  v.w1 = v.row[1];
  v.deck = w0 & 63;
  v.info = (w0 >> 6) & 15;
  v.life = (w0 >> 10) & 7;
  v.s = (w0 >> 13) & 7;
  v.n_own = v.hand_n(v.s);
  v.kn = v.know(v.s);
  // discard pile: one thermometer per card identity (state words 8-9)
  const uint64_t disc = (static_cast<uint64_t>(v.row[9]) << 32) | v.row[8];
  v.seen = 0;
  v.maxfw = 0;
  v.minfw = 7;
  for (int c = 0; c < a.C; ++c) {
    int reach = a.R;
    for (int r = 0; r < a.R; ++r) {
      const int pos = c * a.CPC + (r ? 2 * r + 1 : 0), w = v.copies(r);
      const int gone = __popc(static_cast<uint32_t>(disc >> pos) & ((1u << w) - 1u));
      if (gone >= w && r < reach) reach = r;
      const int vis = gone + (v.fw(c) > r ? 1 : 0);
      v.seen |= static_cast<uint64_t>(vis) << (2 * (c * a.R + r));
    }
    v.maxfw |= static_cast<uint32_t>(reach) << (3 * c);
    v.minfw = min(v.minfw, v.fw(c));
  }
  for (int off = 1; off < a.P; ++off) {
    const int p = v.seat(off);
    const uint32_t hc = v.cards(p);
    for (int i = 0; i < v.hand_n(p); ++i) v.seen += 1ull << (2 * ((hc >> (5 * i)) & 31u));
  }
  const unsigned long long gid = static_cast<unsigned long long>(a.first_gid + g);
  hb::philox4x32_10(static_cast<uint32_t>(a.draw), static_cast<uint32_t>(a.draw >> 32), static_cast<uint32_t>(gid),
                    static_cast<uint32_t>(gid >> 32), static_cast<uint32_t>(a.seed), static_cast<uint32_t>(a.seed >> 32), v.rnd);

  int act = -1, fired = a.n_rules;
  for (int q = 0; q < a.n_rules && act < 0; ++q) {
    const hb_rule rl = a.rules[q];
    int m = -1;
    switch (rl.kind) {
      case HB_RULE_LEGAL_RANDOM: m = rule_legal_random(v); break;
      case HB_RULE_DISCARD_OLDEST_FIRST: m = (v.info < v.INFO && v.n_own > 0) ? v.uid_discard(0) : -1; break;
      case HB_RULE_OSAWA_DISCARD: m = rule_osawa_discard(v); break;
      case HB_RULE_TELL_UNKNOWN: m = rule_tell_unknown(v); break;
      case HB_RULE_TELL_RANDOMLY: m = rule_tell_randomly(v); break;
      case HB_RULE_PLAY_SAFE_CARD: m = rule_play_safe_card(v); break;
      case HB_RULE_PLAY_IF_CERTAIN: m = rule_play_if_certain(v); break;
      case HB_RULE_TELL_PLAYABLE_CARD_OUTER: m = rule_tell_playable_card_outer(v); break;
      case HB_RULE_TELL_DISPENSABLE: m = rule_tell_dispensable(v, rl.arg); break;
      case HB_RULE_DISCARD_RANDOMLY:
        m = (v.info < v.INFO && v.n_own > 0) ? v.uid_discard(static_cast<int>(__umulhi(v.rnd[2], static_cast<uint32_t>(v.n_own)))) : -1;
        break;
      case HB_RULE_PLAY_PROBABLY_SAFE: m = rule_play_probably_safe(v, static_cast<double>(rl.threshold), rl.arg != 0); break;
      case HB_RULE_DISCARD_PROBABLY_USELESS:
        if (v.info < v.INFO && v.n_own > 0) {
          double p;
          const int idx = argmax_probability<true>(v, &p);
          if (p >= static_cast<double>(rl.threshold)) m = v.uid_discard(idx);
        }
        break;
      case HB_RULE_HAIL_MARY: m = (v.deck == 0 && v.life > 1) ? rule_play_probably_safe(v, 0.0, false) : -1; break;
      case HB_RULE_TELL_ANYONE_USELESS_CARD: m = rule_tell_anyone_useless_card(v); break;
      case HB_RULE_TELL_PLAYABLE_CARD: m = rule_tell_playable_card(v); break;
      default: break;  // HB_RULE_TELL_MOST_INFORMATION: the reference function ends without a return (ruleset.py:539-562)
    }
    if (m >= 0) {
      act = m;
      fired = q;
    }
  }
  if (act < 0) act = rule_legal_random(v);  // rule_based.py:24
  a.actions[g] = act;
  if (a.fired) a.fired[g] = fired;
}

}  // namespace

extern "C" int hb_rule_act(const hb_config* cfg, const uint32_t* state_rows_dev, int64_t n_games, int64_t first_game_id,
                           const hb_rule* rules, int32_t n_rules, uint64_t seed, uint64_t draw, int32_t* actions_dev,
                           int32_t* fired_dev, void* stream) {
  if (!cfg || !state_rows_dev || !actions_dev) return hb::fail(HB_ERR_INVALID, "null argument");
  if (int rc = hb_config_validate(cfg)) return rc;
  if (n_rules < 0 || n_rules > HB_MAX_RULES) return hb::fail(HB_ERR_INVALID, "n_rules must be 0..%d", HB_MAX_RULES);
  if (n_rules > 0 && !rules) return hb::fail(HB_ERR_INVALID, "null rules");
  for (int i = 0; i < n_rules; ++i)
    if (rules[i].kind < 0 || rules[i].kind >= HB_RULE_KINDS) return hb::fail(HB_ERR_INVALID, "rule %d: unknown kind %d", i, rules[i].kind);
  if (n_games <= 0) return HB_OK;
  RuleArgs a{};
  a.rows = state_rows_dev;
  a.n = n_games;
  a.first_gid = first_game_id;
  a.P = cfg->players; a.C = cfg->colors; a.R = cfg->ranks; a.H = cfg->hand_size; a.INFO = cfg->max_info;
  a.SW = hb_state_words(cfg);
  a.CPC = hb_deck_size(cfg) / cfg->colors;
  a.n_rules = n_rules;
  for (int i = 0; i < n_rules; ++i) a.rules[i] = rules[i];
  a.seed = seed;
  a.draw = draw;
  a.actions = actions_dev;
  a.fired = fired_dev;
  const unsigned blocks = static_cast<unsigned>((n_games + 127) / 128);
  hipLaunchKernelGGL(rule_kernel, dim3(blocks), dim3(128), 0, static_cast<hipStream_t>(stream), a);
  HB_HIP(hipGetLastError());
  return HB_OK;
}
