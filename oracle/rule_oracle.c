/*
 * rule_oracle.c — CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's rule-based agents:
 *   hanabi_agents/rule_based/ruleset.py      helper functions :14-196, class Ruleset :202-655
 *   hanabi_agents/rule_based/rule_based.py   RulebasedAgent.get_move :13-25 (first rule that returns a move,
 *                                            else a random legal move)
 * One game at a time over the natural game_t arrays, following the Python control flow literally (card lists,
 * nested loops), unlike the bit-packed HIP kernel it checks (hanabi-agents_amd/csrc/rule_agent.hip).
 *
 * Parity status: the rules act on `hanabi_learning_environment` observation objects, a package that is absent
 * from /root/reference and from this image, and the reference has no tests or fixtures for them  =>  "parity
 * unpinned" against the running reference. Pinned by hand-worked situations in tests/test_rule_agents.py.
 * Deliberate readings of ruleset.py, each marked READING below:
 *   - `observation.information_tokens < 8` / `== 8` and `fireworks[...] == 5` hard-code Hanabi-Full; the
 *     configuration's max_info / ranks are used (identical for Hanabi-Full).
 *   - a knowledge entry's colour / rank is "known" iff it was hinted (HLE CardKnowledge semantics); where the
 *     Python indexes `fireworks[card.color]` with an un-hinted colour (osawa_discard :230-231) the check is
 *     skipped instead of wrapping to the last colour.
 *   - tell_unknown :293-307 takes the colour / rank to reveal from the card itself (as its commented dict
 *     version did), not from the still-unknown knowledge entry.
 *   - Python's `random` is replaced by Philox(seed; draw, game): word 0 card index, word 1 coin, word 2 discard
 *     index, word 3 legal-move index.
 *   - an empty denominator in get_card_playability (unreachable for consistent states) gives 0, not an exception.
 */
#include <string.h>

#include "game_state.h"

typedef struct {
  int color, rank;
} card_t;

typedef struct {
  const hb_config* cfg;
  const game_t* g;
  int self;         /* absolute seat of the observer (= player to act) */
  uint32_t rnd[4];
} view_t;

static int seat_at(const view_t* v, int offset) { return (v->self + offset) % v->cfg->players; }
static card_t card_of(const view_t* v, int index) {
  card_t c = {index / v->cfg->ranks, index % v->cfg->ranks};
  return c;
}
static uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* move uids, SURVEY App. A.2 */
static int mv_discard(const view_t* v, int i) { (void)v; return i; }
static int mv_play(const view_t* v, int i) { return v->cfg->hand_size + i; }
static int mv_color(const view_t* v, int off, int color) { return 2 * v->cfg->hand_size + (off - 1) * v->cfg->colors + color; }
static int mv_rank(const view_t* v, int off, int rank) {
  return 2 * v->cfg->hand_size + (v->cfg->players - 1) * v->cfg->colors + (off - 1) * v->cfg->ranks + rank;
}

/* ruleset.py:14-32 for a fully specified card */
static int playable_card(const view_t* v, card_t c) { return c.rank == v->g->fireworks[c.color]; }
/* ruleset.py:35-42 */
static int useless_card(const view_t* v, card_t c, const int* max_fireworks) {
  if (c.rank < v->g->fireworks[c.color]) return 1;
  if (c.rank >= max_fireworks[c.color]) return 1;
  return 0;
}
/* ruleset.py:46-57 */
static int get_plausible_cards(const view_t* v, int player_offset, int hand_index, card_t* out) {
  const knowledge_t* k = &v->g->know[seat_at(v, player_offset)][hand_index];
  int n = 0;
  for (int color = 0; color < v->cfg->colors; ++color)
    for (int rank = 0; rank < v->cfg->ranks; ++rank)
      if (k->color_plausible[color] && k->rank_plausible[rank]) {
        out[n].color = color;
        out[n].rank = rank;
        ++n;
      }
  return n;
}
/* ruleset.py:60-77: the other players' cards, the discard pile, the cards already on the fireworks */
static int get_visible_cards(const view_t* v, int player_offset, card_t* out) {
  int n = 0;
  for (int other = 1; other < v->cfg->players; ++other) {
    if (other == player_offset) continue;
    const int p = seat_at(v, other);
    for (int i = 0; i < v->g->hand_n[p]; ++i) out[n++] = card_of(v, v->g->hand[p][i]);
  }
  for (int id = 0; id < v->cfg->colors * v->cfg->ranks; ++id)
    for (int k = 0; k < v->g->discard_count[id]; ++k) out[n++] = card_of(v, id);
  for (int color = 0; color < v->cfg->colors; ++color)
    for (int rank = 0; rank < v->g->fireworks[color]; ++rank) {
      out[n].color = color;
      out[n].rank = rank;
      ++n;
    }
  return n;
}
/* ruleset.py:174-196 */
static void get_max_fireworks(const view_t* v, int* max_fireworks) {
  for (int color = 0; color < v->cfg->colors; ++color) max_fireworks[color] = v->cfg->ranks; /* READING: 5 */
  for (int id = 0; id < v->cfg->colors * v->cfg->ranks; ++id) {
    const card_t c = card_of(v, id);
    if (v->g->discard_count[id] > 0 && v->g->discard_count[id] >= copies_of_rank(v->cfg, c.rank))
      if (max_fireworks[c.color] >= c.rank) max_fireworks[c.color] = c.rank;
  }
}
/* ruleset.py:82-106 (playability) and :127-150 (uselessness) of the own hand */
static void hand_probabilities(const view_t* v, int useless, double* out) {
  card_t visible[3 * MAXDECK], plausible[MAXC * MAXR];
  int max_fireworks[MAXC];
  const int n_visible = get_visible_cards(v, 0, visible);
  get_max_fireworks(v, max_fireworks);
  for (int hand_index = 0; hand_index < v->g->hand_n[v->self]; ++hand_index) {
    int total = 0, hits = 0;
    const int n_pl = get_plausible_cards(v, 0, hand_index, plausible);
    for (int j = 0; j < n_pl; ++j) {
      int num_in_deck = copies_of_rank(v->cfg, plausible[j].rank);
      for (int k = 0; k < n_visible; ++k)
        if (visible[k].color == plausible[j].color && visible[k].rank == plausible[j].rank) --num_in_deck;
      total += num_in_deck;
      if (useless ? useless_card(v, plausible[j], max_fireworks) : playable_card(v, plausible[j])) hits += num_in_deck;
    }
    out[hand_index] = total > 0 ? (double)hits / (double)total : 0.0;
  }
}
static int argmax(const double* x, int n) { /* np.argmax: first maximum */
  int best = 0;
  for (int i = 1; i < n; ++i)
    if (x[i] > x[best]) best = i;
  return best;
}
static int min_fireworks(const view_t* v) {
  int m = v->g->fireworks[0];
  for (int c = 1; c < v->cfg->colors; ++c)
    if (v->g->fireworks[c] < m) m = v->g->fireworks[c];
  return m;
}

/* ---- the rules (class Ruleset); each returns a move uid or -1 for Python's None --------------------------- */
static int discard_oldest_first(const view_t* v) { /* :206-216 */
  if (v->g->info < v->cfg->max_info && v->g->hand_n[v->self] > 0) return mv_discard(v, 0);
  return -1;
}
static int osawa_discard(const view_t* v) { /* :220-280 */
  if (v->g->info == v->cfg->max_info) return -1;
  int max_fireworks[MAXC];
  get_max_fireworks(v, max_fireworks);
  const int n = v->g->hand_n[v->self];
  for (int i = 0; i < n; ++i) {
    const knowledge_t* k = &v->g->know[v->self][i];
    if (k->color_hinted >= 0) /* READING */
      if (v->g->fireworks[k->color_hinted] == v->cfg->ranks) return mv_discard(v, i);
    if (k->color_hinted >= 0 && k->rank_hinted >= 0)
      if (k->rank_hinted < v->g->fireworks[k->color_hinted] || k->rank_hinted >= max_fireworks[k->color_hinted])
        return mv_discard(v, i);
    if (k->rank_hinted >= 0)
      if (k->rank_hinted < min_fireworks(v)) return mv_discard(v, i);
  }
  for (int i = 0; i < n; ++i) {
    card_t plausible[MAXC * MAXR];
    const int n_pl = get_plausible_cards(v, 0, i, plausible);
    int eventually_playable = 0;
    for (int j = 0; j < n_pl; ++j)
      if (plausible[j].rank < max_fireworks[plausible[j].color]) {
        eventually_playable = 1;
        break;
      }
    if (!eventually_playable) return mv_discard(v, i);
  }
  return -1;
}
static int tell_unknown(const view_t* v) { /* :285-308 */
  if (v->g->info > 0) {
    const int p = seat_at(v, 1);
    for (int i = 0; i < v->g->hand_n[p]; ++i) {
      const knowledge_t* k = &v->g->know[p][i];
      const card_t c = card_of(v, v->g->hand[p][i]); /* READING */
      if (k->color_hinted < 0) return mv_color(v, 1, c.color);
      if (k->rank_hinted < 0) return mv_rank(v, 1, c.rank);
    }
  }
  return -1;
}
static int tell_randomly(const view_t* v) { /* :314-346 */
  if (v->g->info > 0) {
    const int p = seat_at(v, 1);
    if (v->g->hand_n[p] == 0) return -1;
    const int i = (int)mulhi32(v->rnd[0], (uint32_t)v->g->hand_n[p]);
    const card_t c = card_of(v, v->g->hand[p][i]);
    if ((v->rnd[1] & 1u) == 0) return mv_rank(v, 1, c.rank);
    return mv_color(v, 1, c.color);
  }
  return -1;
}
static int play_safe_card(const view_t* v) { /* :350-379 */
  for (int i = 0; i < v->g->hand_n[v->self]; ++i) {
    card_t plausible[MAXC * MAXR];
    const int n_pl = get_plausible_cards(v, 0, i, plausible);
    int definitely_playable = 1;
    for (int j = 0; j < n_pl; ++j)
      if (!playable_card(v, plausible[j])) {
        definitely_playable = 0;
        break;
      }
    if (definitely_playable) return mv_play(v, i);
  }
  return -1;
}
static int play_if_certain(const view_t* v) { /* :383-409 */
  for (int i = 0; i < v->g->hand_n[v->self]; ++i) {
    const knowledge_t* k = &v->g->know[v->self][i];
    if (k->color_hinted >= 0 && k->rank_hinted >= 0)
      if (k->rank_hinted == v->g->fireworks[k->color_hinted]) return mv_play(v, i);
  }
  return -1;
}
static int tell_playable_card_outer(const view_t* v) { /* :413-451 */
  if (v->g->info > 0)
    for (int off = 1; off < v->cfg->players; ++off) {
      const int p = seat_at(v, off);
      for (int i = 0; i < v->g->hand_n[p]; ++i) {
        const card_t c = card_of(v, v->g->hand[p][i]);
        const knowledge_t* k = &v->g->know[p][i];
        const int card_playable = playable_card(v, c);
        if (card_playable && k->rank_hinted < 0) return mv_rank(v, off, c.rank);
        else if (card_playable && k->color_hinted < 0) return mv_color(v, off, c.color);
      }
    }
  return -1;
}
static int tell_dispensable(const view_t* v, int min_information_tokens) { /* :454-514 */
  if (v->g->info < min_information_tokens && v->g->info > 0)
    for (int off = 1; off < v->cfg->players; ++off) {
      const int p = seat_at(v, off);
      for (int i = 0; i < v->g->hand_n[p]; ++i) {
        const card_t c = card_of(v, v->g->hand[p][i]);
        const knowledge_t* k = &v->g->know[p][i];
        if (k->color_hinted < 0 && v->g->fireworks[c.color] == v->cfg->ranks) return mv_color(v, off, c.color);
        if (k->rank_hinted < 0 && c.rank < min_fireworks(v)) return mv_rank(v, off, c.rank);
        if (c.rank < v->g->fireworks[c.color]) {
          if (k->color_hinted < 0 && k->rank_hinted >= 0) return mv_color(v, off, c.color);
          if (k->color_hinted >= 0 && k->rank_hinted < 0) return mv_rank(v, off, c.rank);
        }
      }
    }
  return -1;
}
static int tell_anyone_useless_card(const view_t* v) { /* :522-535 */
  if (v->g->info > 1) {
    int max_fireworks[MAXC];
    get_max_fireworks(v, max_fireworks);
    for (int off = 1; off < v->cfg->players; ++off) {
      const int p = seat_at(v, off);
      for (int i = 0; i < v->g->hand_n[p]; ++i) {
        const card_t c = card_of(v, v->g->hand[p][i]);
        const knowledge_t* k = &v->g->know[p][i];
        if (useless_card(v, c, max_fireworks)) {
          if (k->color_hinted < 0) return mv_color(v, off, c.color);
          if (k->rank_hinted < 0) return mv_rank(v, off, c.rank);
        }
      }
    }
  }
  return -1;
}
static int tell_playable_card(const view_t* v) { /* :570-594 */
  if (v->g->info > 0)
    for (int off = 1; off < v->cfg->players; ++off) {
      const int p = seat_at(v, off);
      for (int i = 0; i < v->g->hand_n[p]; ++i) {
        const card_t c = card_of(v, v->g->hand[p][i]);
        if (playable_card(v, c)) return (v->rnd[1] & 1u) == 0 ? mv_rank(v, off, c.rank) : mv_color(v, off, c.color);
      }
    }
  return -1;
}
static int legal_random(const view_t* v) { /* :598-604; legal moves in uid order */
  int legal[64], n = 0;
  const hb_config* c = v->cfg;
  const game_t* g = v->g;
  for (int i = 0; i < c->hand_size; ++i)
    if (i < g->hand_n[v->self] && g->info < c->max_info) legal[n++] = mv_discard(v, i);
  for (int i = 0; i < c->hand_size; ++i)
    if (i < g->hand_n[v->self]) legal[n++] = mv_play(v, i);
  for (int off = 1; off < c->players; ++off)
    for (int color = 0; color < c->colors; ++color) {
      int any = 0;
      const int p = seat_at(v, off);
      for (int i = 0; i < g->hand_n[p]; ++i) any |= card_of(v, g->hand[p][i]).color == color;
      if (any && g->info > 0) legal[n++] = mv_color(v, off, color);
    }
  for (int off = 1; off < c->players; ++off)
    for (int rank = 0; rank < c->ranks; ++rank) {
      int any = 0;
      const int p = seat_at(v, off);
      for (int i = 0; i < g->hand_n[p]; ++i) any |= card_of(v, g->hand[p][i]).rank == rank;
      if (any && g->info > 0) legal[n++] = mv_rank(v, off, rank);
    }
  if (n == 0) return 0;
  return legal[mulhi32(v->rnd[3], (uint32_t)n)];
}
static int discard_randomly(const view_t* v) { /* :607-614 */
  const int n = v->g->hand_n[v->self];
  if (v->g->info < v->cfg->max_info && n > 0) return mv_discard(v, (int)mulhi32(v->rnd[2], (uint32_t)n));
  return -1;
}
static int play_probably_safe(const view_t* v, double threshold, int require_extra_lives) { /* :617-635 */
  const int n = v->g->hand_n[v->self];
  if (n == 0) return -1;
  double playability[MAXH];
  hand_probabilities(v, 0, playability);
  const int card_index = argmax(playability, n);
  if (!require_extra_lives || v->g->life > 1)
    if (playability[card_index] >= threshold) return mv_play(v, card_index);
  return -1;
}
static int discard_probably_useless(const view_t* v, double threshold) { /* :638-650 */
  const int n = v->g->hand_n[v->self];
  if (v->g->info < v->cfg->max_info && n > 0) {
    double probability_useless[MAXH];
    hand_probabilities(v, 1, probability_useless);
    const int card_index = argmax(probability_useless, n);
    if (probability_useless[card_index] >= threshold) return mv_discard(v, card_index);
  }
  return -1;
}
static int hail_mary(const view_t* v) { /* :653-655 */
  const int deck_size = deck_size_of(v->cfg) - v->g->deck_pos;
  if (deck_size == 0 && v->g->life > 1) return play_probably_safe(v, 0.0, 0);
  return -1;
}

static int apply_rule(const view_t* v, const hb_rule* r) {
  switch (r->kind) {
    case HB_RULE_LEGAL_RANDOM: return legal_random(v);
    case HB_RULE_DISCARD_OLDEST_FIRST: return discard_oldest_first(v);
    case HB_RULE_OSAWA_DISCARD: return osawa_discard(v);
    case HB_RULE_TELL_UNKNOWN: return tell_unknown(v);
    case HB_RULE_TELL_RANDOMLY: return tell_randomly(v);
    case HB_RULE_PLAY_SAFE_CARD: return play_safe_card(v);
    case HB_RULE_PLAY_IF_CERTAIN: return play_if_certain(v);
    case HB_RULE_TELL_PLAYABLE_CARD_OUTER: return tell_playable_card_outer(v); /* = tell_anyone_useful_card :518-519 */
    case HB_RULE_TELL_DISPENSABLE: return tell_dispensable(v, r->arg);
    case HB_RULE_DISCARD_RANDOMLY: return discard_randomly(v);
    case HB_RULE_PLAY_PROBABLY_SAFE: return play_probably_safe(v, (double)r->threshold, r->arg != 0);
    case HB_RULE_DISCARD_PROBABLY_USELESS: return discard_probably_useless(v, (double)r->threshold);
    case HB_RULE_HAIL_MARY: return hail_mary(v);
    case HB_RULE_TELL_ANYONE_USELESS_CARD: return tell_anyone_useless_card(v);
    case HB_RULE_TELL_PLAYABLE_CARD: return tell_playable_card(v);
    default: return -1; /* tell_most_information :539-562 computes best_action and falls off the end: None */
  }
}

/* RulebasedAgent.get_move (rule_based.py:13-25) for every game of the env */
void orc_rule_act(const orc_env* env, const hb_rule* rules, int32_t n_rules, uint64_t seed, uint64_t draw,
                  int32_t* actions, int32_t* fired) {
  for (int64_t gi = 0; gi < env->n; ++gi) {
    view_t v;
    v.cfg = &env->cfg;
    v.g = &env->games[gi];
    v.self = v.g->cur_player;
    const uint64_t gid = (uint64_t)(env->first_game_id + gi);
    const uint32_t ctr[4] = {(uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)gid, (uint32_t)(gid >> 32)};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32(ctr, key, v.rnd);
    int action = -1, which = n_rules;
    for (int q = 0; q < n_rules; ++q) {
      action = apply_rule(&v, &rules[q]);
      if (action >= 0) {
        which = q;
        break;
      }
    }
    if (action < 0) action = legal_random(&v);
    actions[gi] = action;
    if (fired) fired[gi] = which;
  }
}
