/*
 * game_state.h — internal to oracle/ (TEST INFRASTRUCTURE ONLY): the natural, unpacked game state shared by
 * hanabi_oracle.c (env, encoder) and rule_oracle.c (rule-based partners). See hanabi_oracle.h.
 */
#ifndef ORACLE_GAME_STATE_H
#define ORACLE_GAME_STATE_H

#include <stdint.h>
#include "hanabi_oracle.h"

#define MAXP 5
#define MAXH 5
#define MAXC 5
#define MAXR 5
#define MAXDECK 50

/* copies per rank [3,2,2,2,1]: hanabi_agents/rule_based/ruleset.py:9 */
static inline int copies_of_rank(const hb_config* c, int rank) {
  if (rank == 0) return 3;
  if (rank == c->ranks - 1) return 1;
  return 2;
}
static inline int cards_per_color(const hb_config* c) {
  int n = 0;
  for (int r = 0; r < c->ranks; ++r) n += copies_of_rank(c, r);
  return n;
}
static inline int deck_size_of(const hb_config* c) { return c->colors * cards_per_color(c); }

enum { MOVE_PLAY = 0, MOVE_DISCARD = 1, MOVE_REVEAL_COLOR = 2, MOVE_REVEAL_RANK = 3 };
enum { ST_RUNNING = 0, ST_OUT_OF_LIFE = 1, ST_COMPLETED = 2, ST_OUT_OF_CARDS = 3 };

typedef struct {
  int color_plausible[MAXC];
  int rank_plausible[MAXR];
  int color_hinted; /* -1 = no explicit colour hint */
  int rank_hinted;
} knowledge_t;

typedef struct {
  uint8_t deck[MAXDECK];
  int deck_pos; /* cards dealt so far; deck size = D - deck_pos */
  int hand_n[MAXP];
  int hand[MAXP][MAXH]; /* card index colour*ranks+rank, oldest first */
  knowledge_t know[MAXP][MAXH];
  int discard_count[MAXC * MAXR];
  int fireworks[MAXC];
  int info, life, cur_player, turns_to_play, status, moves;
  /* most recent non-deal move (App. A.5 step 5) */
  int la_valid, la_player, la_type, la_card_index, la_target_offset, la_color, la_rank, la_scored,
      la_info_token, la_reveal_mask;
  /* per-seat driver bookkeeping: */
  int pending[MAXP];    /* seat has made a move whose transition is still open */
  int term_since[MAXP]; /* an episode ended since that move */
  int acc[MAXP];        /* reward accumulated since that move, frozen at episode end */
  uint32_t episode;     /* deals done so far; the next deal uses this as its Philox episode index */
} game_t;

struct orc_env {
  hb_config cfg;
  int64_t n;
  uint64_t seed;
  int64_t first_game_id;
  game_t* games;
  uint8_t* decks; /* explicit decks or NULL */
  int64_t illegal;
  int threads;
};

#endif
