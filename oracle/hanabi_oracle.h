/*
 * hanabi_oracle.h — CPU restatement (TEST INFRASTRUCTURE ONLY) of the hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library. The product (hanabi-agents_amd/) never links, imports or calls it.
 *
 * Parity status
 *   - env step / legal mask / canonical encoder: the algorithm lives in the third-party
 *     package `hanabi_learning_environment` (pyhanabi_pybind fork; NOT in /root/reference,
 *     no version pinned anywhere: setup.py:13-21, .gitmodules:1-3). This file restates
 *     the published HLE algorithm. The reference holds no golden
 *     vectors for it  =>  "parity unpinned" against upstream; pinned only by the
 *     hand-worked known-answer tests in tests/test_oracle_env.py and by the in-reference
 *     corroborating facts listed in SURVEY.md §8(c).
 *   - sum tree: pinned against the reference's own gtest vectors
 *     (sum_tree/sum_tree/src/test_sum_tree.cc:45-117) and against the reference header
 *     compiled into oracle/_ref (see oracle/Makefile).
 */
#ifndef HANABI_ORACLE_H
#define HANABI_ORACLE_H

#include <stdint.h>
#include "../include/hanabi_hip.h" /* hb_config, HB_FLAG_*, HB_STEP_* */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Philox4x32-10 (Salmon et al., SC'11), shared definition of all randomness ------ */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* deck of (seed, global game id, episode): cards sorted by the key (philox(j, episode, game)[0] & ~63) | j */
void orc_shuffled_deck(const hb_config* cfg, uint64_t seed, uint64_t game_id, uint32_t episode, uint8_t* deck);

/* ---- env ----------------------------------------------------------------------------- */
typedef struct orc_env orc_env;

orc_env* orc_env_create(const hb_config* cfg, int64_t n_games, uint64_t seed, int64_t first_game_id);
void orc_env_destroy(orc_env* env);
void orc_env_set_decks(orc_env* env, const uint8_t* decks); /* [n, deck_size] or NULL (copied) */
void orc_env_reset(orc_env* env, const uint8_t* mask, int32_t start_player);
void orc_env_observe(orc_env* env, int8_t* obs, int8_t* legal, float* agent_reward, int8_t* agent_step_type);
void orc_env_step(orc_env* env, const int32_t* actions, int8_t* obs, int8_t* legal, float* reward,
                  int8_t* terminal, float* agent_reward, int8_t* agent_step_type, int8_t* score);
int64_t orc_env_illegal_count(const orc_env* env);
void orc_env_export_state(const orc_env* env, uint32_t* rows); /* same packed rows as hb_env_export_state */
void orc_env_set_threads(orc_env* env, int n_threads);          /* OpenMP threads for step (cpu_baseline) */

void orc_random_legal_actions(const int8_t* legal, int64_t n_games, int32_t n_actions, uint64_t seed,
                              uint64_t draw, int64_t first_game_id, int32_t* actions);

/* ---- rule-based partners (rule_oracle.c; hanabi_agents/rule_based/ruleset.py, rule_based.py:13-25) -------- */
void orc_rule_act(const orc_env* env, const hb_rule* rules, int32_t n_rules, uint64_t seed, uint64_t draw,
                  int32_t* actions, int32_t* fired /* optional: index of the rule that fired, n_rules = fallback */);

/* ---- sum tree ------------------------------------------------------------------------- */
typedef struct orc_tree orc_tree;
orc_tree* orc_tree_create(int64_t capacity);
void orc_tree_destroy(orc_tree* t);
int64_t orc_tree_capacity(const orc_tree* t);
const float* orc_tree_nodes(const orc_tree* t); /* 2*cap floats, root at 1 */
void orc_tree_update(orc_tree* t, const int64_t* idx, const float* val, int64_t n);
void orc_tree_fill_range(orc_tree* t, int64_t start, int64_t n, float value);
void orc_tree_sample(const orc_tree* t, const float* quantile, int64_t* idx, float* val, int64_t n);
void orc_tree_get(const orc_tree* t, const int64_t* idx, float* val, int64_t n);
float orc_tree_total(const orc_tree* t);
void orc_per_sample(const orc_tree* t, const double* u, int64_t batch, int64_t* idx, double* prob);
void orc_per_update(orc_tree* t, const int64_t* idx, const float* td, int64_t n, double alpha, float* max_prio,
                    float* min_prio);

#ifdef __cplusplus
}
#endif
#endif
