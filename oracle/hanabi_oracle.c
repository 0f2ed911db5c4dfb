/*
 * hanabi_oracle.c — plain-C CPU restatement of the hot path. TEST INFRASTRUCTURE ONLY
 * (see hanabi_oracle.h for who may load it and for the parity status of each part).
 *
 * Written for readability, one game at a time, with ordinary arrays — deliberately NOT
 * the packed/bit-parallel formulation the HIP kernels use, so that a differential test
 * between the two has real power.
 *
 * Sources restated (the env is absent from /root/reference; citations are to SURVEY.md
 * Appendix A, which records the upstream hanabi_learning_environment algorithm, plus the
 * in-reference facts that corroborate it):
 *   A.1 game parameters      copies per rank [3,2,2,2,1]: hanabi_agents/rule_based/ruleset.py:9
 *   A.2 move uid order       move ctor (type, card_index, target_offset, color, rank): ruleset.py:208-214
 *   A.4 legality             discard illegal at max info tokens: ruleset.py:207,221
 *   A.5 apply move, deal, terminal order, score/reward (lenient switch: rainbow/run_experiment.py:41,310)
 *   A.6 canonical encoder    knowledge API color_plausible/rank_plausible/..._hinted: ruleset.py:52,240
 * Sum tree: sum_tree/sum_tree/include/sum_tree.h (line numbers at each function).
 * Prioritized-replay arithmetic: hanabi_agents/rlax_dqn/priority_buffer.py:36-52.
 */
#include "hanabi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "game_state.h"

/* ======================================================================================
 * sizes (App. A.1, A.2, A.6)
 * ==================================================================================== */
static int num_actions_of(const hb_config* c) {
  return 2 * c->hand_size + (c->players - 1) * c->colors + (c->players - 1) * c->ranks;
}
static int bits_per_card(const hb_config* c) { return c->colors * c->ranks; }
static int hands_len(const hb_config* c) {
  return (c->players - 1) * c->hand_size * bits_per_card(c) + c->players;
}
static int board_len(const hb_config* c) {
  return deck_size_of(c) - c->players * c->hand_size + bits_per_card(c) + c->max_info + c->max_life;
}
static int discards_len(const hb_config* c) { return deck_size_of(c); }
static int last_action_len(const hb_config* c) {
  return c->players + 4 + c->players + c->colors + c->ranks + c->hand_size + c->hand_size + bits_per_card(c) + 2;
}
static int knowledge_len(const hb_config* c) {
  return c->players * c->hand_size * (bits_per_card(c) + c->colors + c->ranks);
}
static int obs_len_of(const hb_config* c) {
  return hands_len(c) + board_len(c) + discards_len(c) + last_action_len(c) + knowledge_len(c);
}
static int state_words_of(const hb_config* c) { return c->players <= 3 ? 32 : 48; }

/* ======================================================================================
 * Philox4x32-10
 * ==================================================================================== */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* canonical (unshuffled) deck: colour-major, rank-minor, copies adjacent (App. A.1) */
static void canonical_deck(const hb_config* c, uint8_t* deck) {
  int n = 0;
  for (int col = 0; col < c->colors; ++col)
    for (int r = 0; r < c->ranks; ++r)
      for (int k = 0; k < copies_of_rank(c, r); ++k) deck[n++] = (uint8_t)(col * c->ranks + r);
}

void orc_shuffled_deck(const hb_config* cfg, uint64_t seed, uint64_t game_id, uint32_t episode, uint8_t* deck) {
  uint8_t canon[MAXDECK];
  uint32_t keys[MAXDECK];
  int D = deck_size_of(cfg);
  canonical_deck(cfg, canon);
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int j = 0; j < D; ++j) {
    uint32_t ctr[4] = {(uint32_t)j, episode, (uint32_t)game_id, (uint32_t)(game_id >> 32)};
    uint32_t out[4];
    orc_philox4x32(ctr, key, out);
    keys[j] = (out[0] & ~63u) | (uint32_t)j; /* 26 random bits, ties broken by position: keys are distinct */
  }
  /* position of card j = number of cards whose key sorts before its key */
  for (int j = 0; j < D; ++j) {
    int rank = 0;
    for (int i = 0; i < D; ++i)
      if (keys[i] < keys[j]) ++rank;
    deck[rank] = canon[j];
  }
}

/* ======================================================================================
 * game state (App. A.3), natural representation
 * ==================================================================================== */
static int score_of(const hb_config* c, const game_t* g) {
  if (g->life <= 0) return 0;
  int s = 0;
  for (int col = 0; col < c->colors; ++col) s += g->fireworks[col];
  return s;
}

static void deal_one(const hb_config* c, game_t* g, int player) {
  int card = g->deck[g->deck_pos++];
  int slot = g->hand_n[player]++;
  g->hand[player][slot] = card;
  knowledge_t* k = &g->know[player][slot];
  for (int i = 0; i < MAXC; ++i) k->color_plausible[i] = i < c->colors;
  for (int i = 0; i < MAXR; ++i) k->rank_plausible[i] = i < c->ranks;
  k->color_hinted = -1;
  k->rank_hinted = -1;
}

/* first player with a short hand while the deck is non-empty, else -1 (App. A.5 step 6) */
static int player_to_deal(const hb_config* c, const game_t* g) {
  if (g->deck_pos >= deck_size_of(c)) return -1;
  for (int p = 0; p < c->players; ++p)
    if (g->hand_n[p] < c->hand_size) return p;
  return -1;
}

static void deal_game(const orc_env* env, int64_t gi, int start_player, int keep_seats) {
  const hb_config* c = &env->cfg;
  game_t* g = &env->games[gi];
  int pending[MAXP], term_since[MAXP], acc[MAXP];
  memcpy(pending, g->pending, sizeof pending);
  memcpy(term_since, g->term_since, sizeof term_since);
  memcpy(acc, g->acc, sizeof acc);
  uint32_t episode = g->episode;
  memset(g, 0, sizeof *g);
  if (keep_seats) {
    memcpy(g->pending, pending, sizeof pending);
    memcpy(g->term_since, term_since, sizeof term_since);
    memcpy(g->acc, acc, sizeof acc);
  }
  int D = deck_size_of(c);
  if (env->decks)
    memcpy(g->deck, env->decks + gi * D, (size_t)D);
  else
    orc_shuffled_deck(c, env->seed, (uint64_t)(env->first_game_id + gi), episode, g->deck);
  g->episode = episode + 1;
  g->info = c->max_info;
  g->life = c->max_life;
  g->turns_to_play = c->players;
  g->status = ST_RUNNING;
  for (int p; (p = player_to_deal(c, g)) >= 0;) deal_one(c, g, p);
  g->cur_player = start_player;
}

/* ======================================================================================
 * moves (App. A.2, A.4)
 * ==================================================================================== */
typedef struct {
  int type, card_index, target_offset, color, rank;
} move_t;

static int decode_move(const hb_config* c, int uid, move_t* m) {
  int H = c->hand_size, P = c->players;
  m->card_index = m->target_offset = m->color = m->rank = -1;
  if (uid < 0) return 0;
  if (uid < H) { m->type = MOVE_DISCARD; m->card_index = uid; return 1; }
  uid -= H;
  if (uid < H) { m->type = MOVE_PLAY; m->card_index = uid; return 1; }
  uid -= H;
  if (uid < (P - 1) * c->colors) {
    m->type = MOVE_REVEAL_COLOR; m->target_offset = 1 + uid / c->colors; m->color = uid % c->colors; return 1;
  }
  uid -= (P - 1) * c->colors;
  if (uid < (P - 1) * c->ranks) {
    m->type = MOVE_REVEAL_RANK; m->target_offset = 1 + uid / c->ranks; m->rank = uid % c->ranks; return 1;
  }
  return 0;
}

static int move_is_legal(const hb_config* c, const game_t* g, const move_t* m) {
  int p = g->cur_player;
  switch (m->type) {
    case MOVE_DISCARD:
      if (g->info >= c->max_info) return 0;
      return m->card_index < g->hand_n[p];
    case MOVE_PLAY:
      return m->card_index < g->hand_n[p];
    case MOVE_REVEAL_COLOR:
    case MOVE_REVEAL_RANK: {
      if (g->info <= 0) return 0;
      if (m->target_offset < 1 || m->target_offset >= c->players) return 0;
      int t = (p + m->target_offset) % c->players;
      for (int i = 0; i < g->hand_n[t]; ++i) {
        int card = g->hand[t][i];
        if (m->type == MOVE_REVEAL_COLOR ? card / c->ranks == m->color : card % c->ranks == m->rank) return 1;
      }
      return 0;
    }
  }
  return 0;
}

static void remove_from_hand(game_t* g, int p, int slot) {
  for (int i = slot; i + 1 < g->hand_n[p]; ++i) {
    g->hand[p][i] = g->hand[p][i + 1];
    g->know[p][i] = g->know[p][i + 1];
  }
  g->hand_n[p]--;
}

static int increment_info(const hb_config* c, game_t* g) {
  if (g->info < c->max_info) { g->info++; return 1; }
  return 0;
}

/* App. A.5. Returns 0 if the uid is illegal (state untouched). */
static int apply_move(const orc_env* env, int64_t gi, int uid, float* reward, int* terminal, int* final_score) {
  const hb_config* c = &env->cfg;
  game_t* g = &env->games[gi];
  move_t m;
  *reward = 0.f; *terminal = 0; *final_score = score_of(c, g);
  if (g->status != ST_RUNNING) return 1; /* finished game without auto-reset: moves are ignored */
  if (!decode_move(c, uid, &m) || !move_is_legal(c, g, &m)) return 0;
  int s = g->cur_player;
  g->pending[s] = 1; g->acc[s] = 0; g->term_since[s] = 0;
  int before = score_of(c, g);
  if (g->deck_pos >= deck_size_of(c)) g->turns_to_play--; /* step 1 */
  g->la_valid = 1; g->la_player = s; g->la_type = m.type;
  g->la_card_index = m.card_index < 0 ? 0 : m.card_index;
  g->la_target_offset = m.target_offset < 0 ? 0 : m.target_offset;
  g->la_color = 0; g->la_rank = 0; g->la_scored = 0; g->la_info_token = 0; g->la_reveal_mask = 0;
  switch (m.type) {
    case MOVE_DISCARD: { /* step 2 */
      int card = g->hand[s][m.card_index];
      g->la_info_token = increment_info(c, g);
      g->la_color = card / c->ranks; g->la_rank = card % c->ranks;
      g->discard_count[card]++;
      remove_from_hand(g, s, m.card_index);
      break;
    }
    case MOVE_PLAY: { /* step 3 */
      int card = g->hand[s][m.card_index];
      int col = card / c->ranks, r = card % c->ranks;
      g->la_color = col; g->la_rank = r;
      if (g->fireworks[col] == r) {
        g->fireworks[col]++;
        g->la_scored = 1;
        if (g->fireworks[col] == c->ranks) g->la_info_token = increment_info(c, g);
      } else {
        g->life--;
        g->discard_count[card]++;
      }
      remove_from_hand(g, s, m.card_index);
      break;
    }
    case MOVE_REVEAL_COLOR:
    case MOVE_REVEAL_RANK: { /* step 4 */
      g->info--;
      int t = (s + m.target_offset) % c->players;
      int mask = 0;
      for (int i = 0; i < g->hand_n[t]; ++i) {
        int card = g->hand[t][i];
        knowledge_t* k = &g->know[t][i];
        if (m.type == MOVE_REVEAL_COLOR) {
          if (card / c->ranks == m.color) {
            mask |= 1 << i;
            k->color_hinted = m.color;
            for (int x = 0; x < MAXC; ++x) k->color_plausible[x] = x == m.color;
          } else {
            k->color_plausible[m.color] = 0;
          }
        } else {
          if (card % c->ranks == m.rank) {
            mask |= 1 << i;
            k->rank_hinted = m.rank;
            for (int x = 0; x < MAXR; ++x) k->rank_plausible[x] = x == m.rank;
          } else {
            k->rank_plausible[m.rank] = 0;
          }
        }
      }
      g->la_reveal_mask = mask;
      if (m.type == MOVE_REVEAL_COLOR) g->la_color = m.color; else g->la_rank = m.rank;
      break;
    }
  }
  g->moves++;
  for (int p; (p = player_to_deal(c, g)) >= 0;) deal_one(c, g, p); /* step 6 */
  g->cur_player = (s + 1) % c->players;
  /* step 7: terminal tests in upstream order */
  if (g->life < 1) g->status = ST_OUT_OF_LIFE;
  else if (score_of(c, g) >= c->colors * c->ranks) g->status = ST_COMPLETED;
  else if (g->turns_to_play <= 0) g->status = ST_OUT_OF_CARDS;
  int after = score_of(c, g); /* step 8 */
  int r = after - before;
  if ((c->flags & HB_FLAG_LENIENT_REWARD) && r < 0) r = 0;
  *reward = (float)r;
  *terminal = g->status != ST_RUNNING;
  *final_score = after;
  for (int q = 0; q < c->players; ++q)
    if (g->pending[q] && !g->term_since[q]) {
      g->acc[q] += r;
      if (*terminal) g->term_since[q] = 1;
    }
  if (*terminal && (c->flags & HB_FLAG_AUTO_RESET))
    deal_game(env, gi, (c->flags & HB_FLAG_RESET_START_NEXT) ? (s + 1) % c->players : 0, 1);
  return 1;
}

/* ======================================================================================
 * legal mask + canonical encoder (App. A.4, A.6) for observer = seat to act
 * ==================================================================================== */
static void legal_mask(const hb_config* c, const game_t* g, int8_t* legal) {
  int A = num_actions_of(c);
  for (int uid = 0; uid < A; ++uid) {
    move_t m;
    decode_move(c, uid, &m);
    legal[uid] = (int8_t)move_is_legal(c, g, &m);
  }
}

static void encode(const hb_config* c, const game_t* g, int8_t* obs) {
  int P = c->players, H = c->hand_size, C = c->colors, R = c->ranks, bits = C * R;
  int o = g->cur_player;
  memset(obs, 0, (size_t)obs_len_of(c));
  int off = 0;
  /* 1. hands of the other players, then "hand is short" flags for every player incl. observer */
  for (int rel = 1; rel < P; ++rel) {
    int p = (o + rel) % P;
    for (int i = 0; i < g->hand_n[p]; ++i) obs[off + i * bits + g->hand[p][i]] = 1;
    off += H * bits;
  }
  for (int rel = 0; rel < P; ++rel)
    if (g->hand_n[(o + rel) % P] < H) obs[off + rel] = 1;
  off += P;
  /* 2. board: deck thermometer, fireworks one-hot per colour, info and life thermometers */
  int D = deck_size_of(c);
  for (int i = 0; i < D - g->deck_pos; ++i) obs[off + i] = 1;
  off += D - P * H;
  for (int col = 0; col < C; ++col) {
    if (g->fireworks[col] > 0) obs[off + g->fireworks[col] - 1] = 1;
    off += R;
  }
  for (int i = 0; i < g->info; ++i) obs[off + i] = 1;
  off += c->max_info;
  for (int i = 0; i < g->life; ++i) obs[off + i] = 1;
  off += c->max_life;
  /* 3. discards: thermometer per card identity, length = its copy count */
  for (int col = 0; col < C; ++col)
    for (int r = 0; r < R; ++r) {
      for (int i = 0; i < g->discard_count[col * R + r]; ++i) obs[off + i] = 1;
      off += copies_of_rank(c, r);
    }
  /* 4. most recent non-deal move, observer-relative */
  if (g->la_valid) {
    int actor = (g->la_player - o + P) % P;
    int reveal = g->la_type == MOVE_REVEAL_COLOR || g->la_type == MOVE_REVEAL_RANK;
    int card_move = g->la_type == MOVE_PLAY || g->la_type == MOVE_DISCARD;
    obs[off + actor] = 1;
    off += P;
    obs[off + g->la_type] = 1; /* [play, discard, reveal colour, reveal rank] */
    off += 4;
    if (reveal) obs[off + (actor + g->la_target_offset) % P] = 1;
    off += P;
    if (g->la_type == MOVE_REVEAL_COLOR) obs[off + g->la_color] = 1;
    off += C;
    if (g->la_type == MOVE_REVEAL_RANK) obs[off + g->la_rank] = 1;
    off += R;
    if (reveal)
      for (int i = 0; i < H; ++i)
        if (g->la_reveal_mask & (1 << i)) obs[off + i] = 1;
    off += H;
    if (card_move) obs[off + g->la_card_index] = 1;
    off += H;
    if (card_move) obs[off + g->la_color * R + g->la_rank] = 1;
    off += bits;
    if (g->la_type == MOVE_PLAY) {
      if (g->la_scored) obs[off] = 1;
      if (g->la_info_token) obs[off + 1] = 1;
    }
    off += 2;
  } else {
    off += last_action_len(c);
  }
  /* 5. card knowledge for every player, observer first */
  for (int rel = 0; rel < P; ++rel) {
    int p = (o + rel) % P;
    for (int i = 0; i < g->hand_n[p]; ++i) {
      const knowledge_t* k = &g->know[p][i];
      for (int col = 0; col < C; ++col)
        if (k->color_plausible[col])
          for (int r = 0; r < R; ++r)
            if (k->rank_plausible[r]) obs[off + col * R + r] = 1;
      off += bits;
      if (k->color_hinted >= 0) obs[off + k->color_hinted] = 1;
      off += C;
      if (k->rank_hinted >= 0) obs[off + k->rank_hinted] = 1;
      off += R;
    }
    off += (H - g->hand_n[p]) * (bits + C + R);
  }
}

static void emit(const orc_env* env, int64_t gi, int8_t* obs, int8_t* legal, float* agent_reward,
                 int8_t* agent_step_type) {
  const hb_config* c = &env->cfg;
  const game_t* g = &env->games[gi];
  if (obs) encode(c, g, obs + gi * obs_len_of(c));
  if (legal) legal_mask(c, g, legal + gi * num_actions_of(c));
  int o = g->cur_player;
  if (agent_reward) agent_reward[gi] = g->pending[o] ? (float)g->acc[o] : 0.f;
  if (agent_step_type)
    agent_step_type[gi] = (int8_t)(!g->pending[o] ? HB_STEP_FIRST : g->term_since[o] ? HB_STEP_LAST : HB_STEP_MID);
}

/* ======================================================================================
 * public env API
 * ==================================================================================== */
orc_env* orc_env_create(const hb_config* cfg, int64_t n_games, uint64_t seed, int64_t first_game_id) {
  orc_env* e = (orc_env*)calloc(1, sizeof *e);
  e->cfg = *cfg;
  e->n = n_games;
  e->seed = seed;
  e->first_game_id = first_game_id;
  e->games = (game_t*)calloc((size_t)n_games, sizeof(game_t));
  e->threads = 1;
  return e;
}
void orc_env_destroy(orc_env* e) {
  if (!e) return;
  free(e->games);
  free(e->decks);
  free(e);
}
void orc_env_set_threads(orc_env* e, int n) { e->threads = n < 1 ? 1 : n; }
void orc_env_set_decks(orc_env* e, const uint8_t* decks) {
  free(e->decks);
  e->decks = NULL;
  if (decks) {
    size_t bytes = (size_t)e->n * (size_t)deck_size_of(&e->cfg);
    e->decks = (uint8_t*)malloc(bytes);
    memcpy(e->decks, decks, bytes);
  }
}
void orc_env_reset(orc_env* e, const uint8_t* mask, int32_t start_player) {
  for (int64_t g = 0; g < e->n; ++g)
    if (!mask || mask[g]) deal_game(e, g, start_player, 0);
}
void orc_env_observe(orc_env* e, int8_t* obs, int8_t* legal, float* agent_reward, int8_t* agent_step_type) {
  for (int64_t g = 0; g < e->n; ++g) emit(e, g, obs, legal, agent_reward, agent_step_type);
}
void orc_env_step(orc_env* e, const int32_t* actions, int8_t* obs, int8_t* legal, float* reward, int8_t* terminal,
                  float* agent_reward, int8_t* agent_step_type, int8_t* score) {
  int64_t illegal = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(e->threads) reduction(+ : illegal) schedule(static)
#endif
  for (int64_t g = 0; g < e->n; ++g) {
    float r;
    int term, fs;
    if (!apply_move(e, g, actions[g], &r, &term, &fs)) ++illegal;
    if (reward) reward[g] = r;
    if (terminal) terminal[g] = (int8_t)term;
    if (score) score[g] = (int8_t)fs;
    emit(e, g, obs, legal, agent_reward, agent_step_type);
  }
  e->illegal += illegal;
}
int64_t orc_env_illegal_count(const orc_env* e) { return e->illegal; }

/* Packed state rows shared with the HIP path (layout: DESIGN.md §3). */
void orc_env_export_state(const orc_env* e, uint32_t* rows) {
  const hb_config* c = &e->cfg;
  int W = state_words_of(c), P = c->players, D = deck_size_of(c);
  for (int64_t gi = 0; gi < e->n; ++gi) {
    const game_t* g = &e->games[gi];
    uint32_t* w = rows + gi * W;
    memset(w, 0, (size_t)W * 4);
    w[0] = (uint32_t)(D - g->deck_pos) | (uint32_t)g->info << 6 | (uint32_t)g->life << 10 |
           (uint32_t)g->cur_player << 13 | (uint32_t)(g->turns_to_play & 7) << 16 | (uint32_t)g->status << 19 |
           (uint32_t)(g->moves & 255) << 21;
    for (int col = 0; col < c->colors; ++col) w[1] |= (uint32_t)g->fireworks[col] << (3 * col);
    for (int p = 0; p < P; ++p) w[1] |= (uint32_t)g->hand_n[p] << (15 + 3 * p);
    if (g->la_valid)
      w[2] = 1u | (uint32_t)g->la_player << 1 | (uint32_t)g->la_type << 4 | (uint32_t)g->la_card_index << 6 |
             (uint32_t)g->la_target_offset << 9 | (uint32_t)g->la_color << 12 | (uint32_t)g->la_rank << 15 |
             (uint32_t)g->la_scored << 18 | (uint32_t)g->la_info_token << 19 | (uint32_t)g->la_reveal_mask << 20;
    for (int p = 0; p < P; ++p) {
      w[3] |= (uint32_t)g->pending[p] << p | (uint32_t)g->term_since[p] << (5 + p);
      w[4 + p / 4] |= (uint32_t)(uint8_t)(int8_t)g->acc[p] << (8 * (p % 4));
    }
    w[6] = g->episode;
    { /* discard pile in the form the encoder emits it: one thermometer per card identity */
      int pos = 0;
      for (int i = 0; i < c->colors * c->ranks; ++i) {
        uint64_t v = (((uint64_t)1 << g->discard_count[i]) - 1) << pos;
        w[8] |= (uint32_t)v;
        w[9] |= (uint32_t)(v >> 32);
        pos += copies_of_rank(c, i % c->ranks);
      }
    }
    for (int p = 0; p < P; ++p) {
      uint32_t cards = 0;
      uint64_t kn = 0;
      for (int i = 0; i < MAXH; ++i) {
        if (i < g->hand_n[p]) {
          const knowledge_t* k = &g->know[p][i];
          uint64_t v = 0;
          for (int x = 0; x < c->colors; ++x) v |= (uint64_t)(k->color_plausible[x] != 0) << x;
          for (int x = 0; x < c->ranks; ++x) v |= (uint64_t)(k->rank_plausible[x] != 0) << (5 + x);
          v |= (uint64_t)(k->color_hinted >= 0) << 10 | (uint64_t)(k->rank_hinted >= 0) << 11;
          kn |= v << (12 * i);
          cards |= (uint32_t)g->hand[p][i] << (5 * i);
        } else {
          cards |= 31u << (5 * i);
        }
      }
      w[10 + p] = cards;
      w[10 + P + 2 * p] = (uint32_t)kn;
      w[10 + P + 2 * p + 1] = (uint32_t)(kn >> 32);
    }
    uint8_t* deck = (uint8_t*)(w + 10 + 3 * P);
    memcpy(deck, g->deck, (size_t)D);
  }
}

void orc_random_legal_actions(const int8_t* legal, int64_t n_games, int32_t n_actions, uint64_t seed,
                              uint64_t draw, int64_t first_game_id, int32_t* actions) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int64_t g = 0; g < n_games; ++g) {
    uint64_t gid = (uint64_t)(first_game_id + g);
    uint32_t ctr[4] = {(uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)gid, (uint32_t)(gid >> 32)};
    uint32_t out[4];
    orc_philox4x32(ctr, key, out);
    const int8_t* row = legal + g * n_actions;
    int n_legal = 0;
    for (int a = 0; a < n_actions; ++a) n_legal += row[a] != 0;
    int pick = 0;
    if (n_legal > 0) {
      int k = (int)(((uint64_t)out[0] * (uint64_t)n_legal) >> 32);
      for (int a = 0; a < n_actions; ++a)
        if (row[a] && k-- == 0) { pick = a; break; }
    }
    actions[g] = pick;
  }
}

/* ======================================================================================
 * sum tree: flat heap, root at node 1, leaves at [cap, 2cap)
 * ==================================================================================== */
struct orc_tree {
  int64_t cap;
  float* nodes;
};

orc_tree* orc_tree_create(int64_t capacity) {
  /* capacity rounds up to a power of two: depth = ceil(log2(capacity)) (sum_tree.h:25-30) */
  int64_t cap = 1;
  while (cap < capacity) cap <<= 1;
  orc_tree* t = (orc_tree*)calloc(1, sizeof *t);
  t->cap = cap;
  t->nodes = (float*)calloc((size_t)(2 * cap), sizeof(float));
  return t;
}
void orc_tree_destroy(orc_tree* t) {
  if (!t) return;
  free(t->nodes);
  free(t);
}
int64_t orc_tree_capacity(const orc_tree* t) { return t->cap; } /* sum_tree.h:76 */
const float* orc_tree_nodes(const orc_tree* t) { return t->nodes; }
float orc_tree_total(const orc_tree* t) { return t->nodes[1]; } /* sum_tree.h:74 */

static void set_leaf(orc_tree* t, int64_t idx, float value) {
  /* sum_tree.h:32-35,80-90 sets the leaf and walks to the root. The reference adds the
   * float difference to each ancestor; here every ancestor is recomputed as the exact
   * float sum of its two children, identical whenever the sums are
   * exactly representable. */
  int64_t node = t->cap + idx;
  t->nodes[node] = value;
  for (node >>= 1; node >= 1; node >>= 1) t->nodes[node] = t->nodes[2 * node] + t->nodes[2 * node + 1];
}
void orc_tree_update(orc_tree* t, const int64_t* idx, const float* val, int64_t n) { /* sum_tree.h:38-44 */
  for (int64_t i = 0; i < n; ++i)
    if (idx[i] >= 0 && idx[i] < t->cap) set_leaf(t, idx[i], val[i]);
}
void orc_tree_fill_range(orc_tree* t, int64_t start, int64_t n, float value) { /* priority_buffer.py:29-32 */
  for (int64_t i = 0; i < n; ++i) set_leaf(t, (start + i) % t->cap, value);
}
static int64_t descend(const orc_tree* t, float quantile, float* leaf) { /* sum_tree.h:46-50,92-105 */
  float query = quantile * t->nodes[1];
  int64_t node = 1;
  while (node < t->cap) {
    float left = t->nodes[2 * node];
    if (query < left) {
      node = 2 * node;
    } else {
      query -= left;
      node = 2 * node + 1;
    }
  }
  if (leaf) *leaf = t->nodes[node];
  return node - t->cap;
}
void orc_tree_sample(const orc_tree* t, const float* q, int64_t* idx, float* val, int64_t n) {
  for (int64_t i = 0; i < n; ++i) idx[i] = descend(t, q[i], val ? &val[i] : NULL);
}
void orc_tree_get(const orc_tree* t, const int64_t* idx, float* val, int64_t n) { /* sum_tree.h:61-72 */
  for (int64_t i = 0; i < n; ++i) val[i] = (idx[i] >= 0 && idx[i] < t->cap) ? t->nodes[t->cap + idx[i]] : 0.f;
}

/* priority_buffer.py:36-42: keys = linspace(1/B, 1, B) - U[0,1/B) in float64, handed to the
 * float tree (pybind converts each key to float), prios = (leaf + 1e-10) / total in float64. */
void orc_per_sample(const orc_tree* t, const double* u, int64_t B, int64_t* idx, double* prob) {
  double start = 1.0 / (double)B;
  double step = B > 1 ? (1.0 - start) / (double)(B - 1) : 0.0;
  double total = (double)t->nodes[1];
  for (int64_t i = 0; i < B; ++i) {
    double lin = (B > 1 && i == B - 1) ? 1.0 : (double)i * step + start;
    float leaf;
    idx[i] = descend(t, (float)(lin - u[i]), &leaf);
    prob[i] = ((double)leaf + 1e-10) / total;
  }
}
/* priority_buffer.py:48-52: (|td| + 1e-10) ** alpha on a float32 array (float32 arithmetic),
 * running max / min, then update_values. The power is taken in double and rounded once. */
void orc_per_update(orc_tree* t, const int64_t* idx, const float* td, int64_t n, double alpha, float* max_prio,
                    float* min_prio) {
  double a = (double)(float)alpha;
  for (int64_t i = 0; i < n; ++i) {
    float x = fabsf(td[i]) + 1e-10f;
    float p = (float)pow((double)x, a);
    if (max_prio && p > *max_prio) *max_prio = p;
    if (min_prio && p < *min_prio) *min_prio = p;
    if (idx[i] >= 0 && idx[i] < t->cap) set_leaf(t, idx[i], p);
  }
}

/* size helpers exported for the Python side of the tests */
int orc_num_actions(const hb_config* c) { return num_actions_of(c); }
int orc_obs_len(const hb_config* c) { return obs_len_of(c); }
int orc_deck_size(const hb_config* c) { return deck_size_of(c); }
int orc_state_words(const hb_config* c) { return state_words_of(c); }
