/*
 * hanabi_hip.h — C-ABI of the MI355X-native Hanabi self-play hot path.
 *
 * One shared library (libhanabi_hip.so, built from hanabi-agents_amd/csrc/) exports
 * exactly the entry points declared here. They are what a binding of the reference
 * would call in place of
 *   - the external C++ hanabi_learning_environment parallel env + canonical encoder
 *     (reference call sites: hanabi_agents/rule_based/ruleset.py:4,
 *      hanabi_agents/rule_based/rule_based.py:2,28,
 *      hanabi_agents/rainbow/run_experiment.py:36,127-128,266-273,308), and
 *   - the pybind11 sum_tree.SumTreef module
 *     (sum_tree/sum_tree/src/sum_tree_py.cc:9-22, used by
 *      hanabi_agents/rlax_dqn/priority_buffer.py:17,30-32,41-42,52).
 *
 * Conventions
 *   - Every pointer argument named *_dev is a DEVICE pointer (HBM). Sizes are element
 *     counts. No torch / C++ types cross this boundary.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). All work is
 *     enqueued on it; no entry point synchronises unless its comment says so.
 *   - Every function returns HB_OK (0) or a negative HB_ERR_* code and never throws;
 *     hb_last_error() returns a thread-local message for the last failure.
 *   - Handles are owned by the caller and are bound to the HIP device that was current
 *     at creation. One handle per GPU; not re-entrant.
 *   - There is NO CPU fallback: on a machine without a gfx950 device hb_env_create /
 *     hb_tree_create fail with HB_ERR_NO_DEVICE.
 */
#ifndef HANABI_HIP_H
#define HANABI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HB_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------------------ */
#define HB_OK 0
#define HB_ERR_INVALID (-1)   /* bad argument / unsupported configuration */
#define HB_ERR_NO_DEVICE (-2) /* no HIP device, or kernel image not loadable */
#define HB_ERR_HIP (-3)       /* a HIP runtime call failed */
#define HB_ERR_ALIGN (-4)     /* an output pointer is not 16-byte aligned */
#define HB_ERR_NOMEM (-5)

const char* hb_last_error(void);
int hb_abi_version(void);

/* ---- game configuration ------------------------------------------------------------- */
/* Mirrors the HanabiGame parameters the reference passes through rl_env.make
 * (hanabi_agents/rainbow/run_experiment.py:119-128: 'Hanabi-Full', 'Hanabi-Small', ...).
 * Limits: 2<=players<=5, 1<=colors<=5, 1<=ranks<=5, 1<=hand_size<=5,
 * 1<=max_info<=15, 1<=max_life<=7, players*hand_size <= deck size. */
typedef struct hb_config {
  int32_t players;
  int32_t colors;
  int32_t ranks;
  int32_t hand_size;
  int32_t max_info;
  int32_t max_life;
  int32_t flags; /* HB_FLAG_* */
} hb_config;

#define HB_FLAG_AUTO_RESET 1u       /* a game that ends is re-dealt inside the same step */
#define HB_FLAG_RESET_START_NEXT 2u /* auto-reset: new game starts with the seat after the one that just moved
                                       (keeps N lock-stepped games on one acting seat); otherwise seat 0 */
#define HB_FLAG_LENIENT_REWARD 4u   /* clamp negative rewards to 0 (run_experiment.py:41,310 LENIENT_SCORE) */

/* step types, as dm_env / hanabi_agents/rlax_dqn/rlax_rainbow.py:292-308 use them */
#define HB_STEP_FIRST 0
#define HB_STEP_MID 1
#define HB_STEP_LAST 2

/* Derived sizes (pure host functions, usable without a GPU). */
int hb_config_validate(const hb_config* cfg);
int hb_num_actions(const hb_config* cfg);   /* 2*hand + (P-1)*(colors+ranks): 20 / 48 / 11 */
int hb_obs_len(const hb_config* cfg);       /* canonical encoding length: 658 / 1280 / 171 */
int hb_deck_size(const hb_config* cfg);     /* 50 / 20 */
int hb_state_words(const hb_config* cfg);   /* u32 words per game-state row: 32 (P<=3) or 48 */
int hb_obs_words(const hb_config* cfg);     /* u32 words of a bit-packed observation row: ceil(obs_len / 32) = 21 / 40 / 6 */

/* ---- vectorised environment ----------------------------------------------------------
 * N independent games. Game state lives in HBM as one row of hb_state_words() u32 per
 * game (layout documented in DESIGN.md §3 and mirrored by oracle/hanabi_oracle.c).
 *
 * Replaces (external HLE, spec in SURVEY.md App. A): HanabiState::ApplyMove + deal +
 * terminal test + reward, HanabiState::MoveIsLegal over all uids, and
 * CanonicalObservationEncoder::Encode, fused into one kernel launch per step.          */
typedef struct hb_env hb_env;

/* first_game_id: global id of local game 0 (rank * n_games when sharding over GPUs);
 * decks of game g, episode e are a pure function of (seed, first_game_id + g, e).      */
int hb_env_create(const hb_config* cfg, int64_t n_games, uint64_t seed, int64_t first_game_id, hb_env** out);
int hb_env_destroy(hb_env* env);
int64_t hb_env_num_games(const hb_env* env);

/* Explicit-deck mode (parity / known-answer tests): decks_dev is [n_games, deck_size]
 * uint8 card indices (color*ranks+rank); decks_dev[g][0] is the first card dealt. Each
 * (re)deal of game g then uses that order instead of the Philox shuffle. NULL turns it
 * off. The buffer is borrowed and must outlive its use.                                */
int hb_env_set_decks(hb_env* env, const uint8_t* decks_dev);

/* (Re)deal games. mask_dev: [n_games] uint8, nonzero = reset this game, NULL = all.
 * start_player: seat that moves first (0..P-1). Clears the per-seat bookkeeping of the
 * games it resets and bumps their episode counter.                                     */
int hb_env_reset(hb_env* env, const uint8_t* mask_dev, int32_t start_player, void* stream);

/* Encode the current observation + legal-move mask of the seat to act, without changing
 * any state. obs_dev [n, obs_len] int8 0/1, legal_dev [n, n_actions] int8 0/1 (both
 * 16-byte aligned); agent_reward_dev [n] float / agent_step_type_dev [n] int8 may be
 * NULL.                                                                                */
int hb_env_observe(hb_env* env, int8_t* obs_dev, int8_t* legal_dev, float* agent_reward_dev,
                   int8_t* agent_step_type_dev, void* stream);

/* One move per game: actions_dev[g] is the move uid (App. A.2 order: discard, play,
 * reveal colour, reveal rank) chosen by game g's current seat. Then, for the seat that
 * acts next (after an auto-reset: the first seat of the fresh game):
 *   obs_dev / legal_dev          its canonical observation and legal mask   (required)
 *   reward_dev     [n] float     score delta of THIS move                   (nullable)
 *   terminal_dev   [n] int8      1 iff this move ended the game             (nullable)
 *   agent_reward_dev [n] float   rewards accumulated for that seat since its own last
 *                                move, frozen at the end of its episode     (nullable)
 *   agent_step_type_dev [n] int8 HB_STEP_FIRST (seat has no move pending), HB_STEP_LAST
 *                                (its episode ended since its last move), else MID
 *   score_dev      [n] int8      terminal: final score of the finished game (0 if all
 *                                lives lost); otherwise the running score   (nullable)
 * An illegal uid leaves the game untouched, re-emits its observation and increments the
 * counter read by hb_env_illegal_count().                                              */
int hb_env_step(hb_env* env, const int32_t* actions_dev, int8_t* obs_dev, int8_t* legal_dev,
                float* reward_dev, int8_t* terminal_dev, float* agent_reward_dev,
                int8_t* agent_step_type_dev, int8_t* score_dev, void* stream);

/* Bit-packed form of the same observation. The encoder builds every observation as hb_obs_words() u32 of bits before it
 * expands them to the reference's int8 layout; these entry points hand that form out directly:
 *   obs_bits_dev [n, hb_obs_words()] u32 (16-byte aligned; required): observation element i of game g is
 *                bit (i & 31) of word i >> 5 of row g; the pad bits of the last word are zero;
 *   obs_dev      the int8 [n, obs_len] form as well, or NULL to skip it (658 -> 84 bytes written per game for
 *                2-player full Hanabi). hb_obs_unpack of obs_bits_dev gives exactly what obs_dev would have held.
 * Everything else is as in hb_env_observe / hb_env_step. Consumers: hb_replay_insert (rows are plain bytes: pass
 * obs_len = 4 * hb_obs_words()), hb_replay_gather_packed, hb_actor_hidden_packed.                                   */
int hb_env_observe_packed(hb_env* env, uint32_t* obs_bits_dev, int8_t* obs_dev, int8_t* legal_dev, float* agent_reward_dev,
                          int8_t* agent_step_type_dev, void* stream);
int hb_env_step_packed(hb_env* env, const int32_t* actions_dev, uint32_t* obs_bits_dev, int8_t* obs_dev, int8_t* legal_dev,
                       float* reward_dev, int8_t* terminal_dev, float* agent_reward_dev, int8_t* agent_step_type_dev,
                       int8_t* score_dev, void* stream);

/* hb_env_step_packed with the acting agent's epsilon-greedy selection (DQNPolicy.policy's last step, rlax_rainbow.py:113-122)
 * fused into the env kernel: game g's move is chosen from q_dev[g, :] (fp32 [N, A], 16-byte aligned) and sel_legal_dev[g, :]
 * (int8 [N, A]: the previous step's legal output; it may be the same buffer as legal_dev) by the rule and Philox draws of
 * hb_policy_select — identical actions — written to actions_out_dev[g] and applied. At most 64 actions.                    */
int hb_env_step_select_packed(hb_env* env, const float* q_dev, const int8_t* sel_legal_dev, float epsilon, uint64_t seed,
                              uint64_t draw, int64_t first_game_id, int32_t* actions_out_dev, uint32_t* obs_bits_dev,
                              int8_t* obs_dev, int8_t* legal_dev, float* reward_dev, int8_t* terminal_dev,
                              float* agent_reward_dev, int8_t* agent_step_type_dev, int8_t* score_dev, void* stream);
/* int8 0/1 rows [rows, obs_len] <-> packed rows [rows, ceil(obs_len / 32)] u32 (any nonzero byte packs to 1).        */
int hb_obs_pack(const int8_t* obs_dev, uint32_t* bits_dev, int64_t rows, int32_t obs_len, void* stream);
int hb_obs_unpack(const uint32_t* bits_dev, int8_t* obs_dev, int64_t rows, int32_t obs_len, void* stream);

/* Number of illegal uids seen since creation (synchronises the stream).                */
int hb_env_illegal_count(hb_env* env, int64_t* out);

/* Running totals since creation: episodes finished and the sum of their final scores
 * (accumulated inside hb_env_step; synchronises the stream).                             */
int hb_env_stats(hb_env* env, int64_t* episodes, int64_t* score_sum);

/* Raw state rows for differential tests: rows_dev is [n_games, hb_state_words()] u32.  */
int hb_env_export_state(hb_env* env, uint32_t* rows_dev, void* stream);
int hb_env_import_state(hb_env* env, const uint32_t* rows_dev, void* stream);

/* Uniform-random legal policy used by bench/tests (Philox(seed, game, draw)):
 * actions_dev[g] = the k-th set entry of legal_dev[g], k uniform. No legal move -> 0.  */
int hb_random_legal_actions(const int8_t* legal_dev, int64_t n_games, int32_t n_actions, uint64_t seed,
                            uint64_t draw, int64_t first_game_id, int32_t* actions_dev, void* stream);

/* ---- rule-based partners (hanabi_agents/rule_based/) ------------------------------------
 * The reference's RulebasedAgent (rule_based.py:13-25) walks a priority list of rules over the rich observation
 * object of ONE game and takes the first move a rule returns, falling back to a random legal move. Here the same
 * walk runs for all games in one launch, straight from the env's state rows (a rule only ever reads what the
 * acting player may see: its own card knowledge, the other hands, fireworks, discards, tokens, deck size).
 * Rule semantics follow ruleset.py (line numbers at each kind); randomness is Philox(seed; draw, game) instead of
 * Python's `random` (word 0: card index, word 1: rank-or-colour coin, word 2: discard index, word 3: legal move).
 * Where ruleset.py hard-codes the full game's 8 information tokens / 5 ranks, the configuration's max_info /
 * ranks are used.                                                                                   */
enum {
  HB_RULE_LEGAL_RANDOM = 0,              /* ruleset.py:598-604                                          */
  HB_RULE_DISCARD_OLDEST_FIRST = 1,      /* :206-216                                                    */
  HB_RULE_OSAWA_DISCARD = 2,             /* :220-280                                                    */
  HB_RULE_TELL_UNKNOWN = 3,              /* :285-308  next player only, colour before rank              */
  HB_RULE_TELL_RANDOMLY = 4,             /* :314-346  next player only                                  */
  HB_RULE_PLAY_SAFE_CARD = 5,            /* :350-379  every plausible identity is playable              */
  HB_RULE_PLAY_IF_CERTAIN = 6,           /* :383-409  colour and rank hinted and playable               */
  HB_RULE_TELL_PLAYABLE_CARD_OUTER = 7,  /* :413-451  (= tell_anyone_useful_card :518-519), rank first  */
  HB_RULE_TELL_DISPENSABLE = 8,          /* :454-514  arg = min_information_tokens                      */
  HB_RULE_DISCARD_RANDOMLY = 9,          /* :607-614                                                    */
  HB_RULE_PLAY_PROBABLY_SAFE = 10,       /* :617-635  threshold; arg != 0: needs more than one life     */
  HB_RULE_DISCARD_PROBABLY_USELESS = 11, /* :638-650  threshold                                         */
  HB_RULE_HAIL_MARY = 12,                /* :653-655  empty deck and > 1 life: play the likeliest card  */
  HB_RULE_TELL_ANYONE_USELESS_CARD = 13, /* :522-535  needs > 1 information token                       */
  HB_RULE_TELL_PLAYABLE_CARD = 14,       /* :570-594  rank or colour by coin                            */
  HB_RULE_TELL_MOST_INFORMATION = 15,    /* :539-562  never returns a move in the reference (no return) */
  HB_RULE_KINDS = 16
};
#define HB_MAX_RULES 16
typedef struct hb_rule {
  int32_t kind;
  int32_t arg;
  float threshold;
} hb_rule;

/* Borrowed device pointer to the env's state rows [n_games, hb_state_words()] (layout: DESIGN.md §3). */
const uint32_t* hb_env_state(hb_env* env);

/* actions_dev[g] = move uid chosen by the first rule that fires for the player to act in game g;
 * fired_dev[g] (optional) = index of that rule, n_rules when none fired and the random-legal fallback
 * of rule_based.py:24 was taken. rules is a HOST array of n_rules <= HB_MAX_RULES entries.            */
int hb_rule_act(const hb_config* cfg, const uint32_t* state_rows_dev, int64_t n_games, int64_t first_game_id,
                const hb_rule* rules, int32_t n_rules, uint64_t seed, uint64_t draw, int32_t* actions_dev,
                int32_t* fired_dev, void* stream);

/* Tuning knob for measurements: games handled per 64-lane wavefront (8, 16, 32 or 64); 0 (the default) picks 16,
 * or 32 for a packed-only step over >= 32768 games. The results do not depend on it.  */
int hb_env_set_games_per_wave(hb_env* env, int32_t g);

/* Deck-pool refill placement. Every game keeps its NEXT deck pre-shuffled in HBM, so a re-deal inside
 * hb_env_step is a 52-byte copy; the shuffles that replace consumed decks run in a second small kernel.
 * on == 0 (default): it runs in order on the caller's stream right after the step kernel (graph-capturable).
 * on != 0: it is forked onto a stream owned by the env and joined by the next call that needs the pool, so
 * it overlaps the caller's following work — worthwhile only when the caller is not launch-bound (the fork /
 * join costs four extra runtime calls per step; measured slower in bench.py). Synchronises the device.    */
int hb_env_set_async_refill(hb_env* env, int32_t on);
/* How often that second kernel runs: every `steps` calls of hb_env_step (1..max_life). A game cannot end in fewer than
 * max_life moves, so a pooled deck consumed at step t is not needed again before step t + max_life; the default is
 * min(max_life, 3). Results do not depend on it (deck(g, e) is a pure function of seed, game id and deal counter).       */
int hb_env_set_refill_period(hb_env* env, int32_t steps);

/* Measurement hook: when both are non-NULL (hipEvent_t handles), every following env kernel
 * launch records its own start/stop into them (hipExtLaunchKernelGGL), i.e. the dispatch's
 * begin/end timestamps rather than a pair of stream markers. NULL, NULL turns it off.      */
int hb_env_set_profile_events(hb_env* env, void* start_event, void* stop_event);

/* Stream-ordering helpers for host drivers (hipEvent_t / hipStream_t as void*): the self-play loop orders its acting
 * and learner streams with a handful of events per step. No reference counterpart (the reference is single-threaded
 * host code); they exist so that a binding needs no HIP runtime binding of its own. Events are created with timing off. */
int hb_event_create(void** event_out);
int hb_event_destroy(void* event);
int hb_event_record(void* event, void* stream);
int hb_stream_wait_event(void* stream, void* event);

/* ---- GPU-resident sum tree ------------------------------------------------------------
 * Replaces sum_tree.SumTreef (sum_tree/sum_tree/include/sum_tree.h:22-130 through
 * sum_tree/sum_tree/src/sum_tree_py.cc:9-22). Flat fp32 heap array in HBM: node 1 is the
 * root, leaves are nodes [cap, 2*cap); every internal node is exactly fl(left + right). */
typedef struct hb_tree hb_tree;

int hb_tree_create(int64_t capacity, hb_tree** out); /* capacity rounds up to a power of two (sum_tree.h:28) */
int hb_tree_destroy(hb_tree* t);
int64_t hb_tree_capacity(const hb_tree* t);          /* get_capacity() */
/* device pointer to the 2*cap floats of the heap (zero-copy consumers). While lazy top levels are on (hb_tree_set_lazy_top), and
 * after they are switched off until the next hb_tree_* call that is given a stream, nodes[1 .. cap / 1024) — the levels above
 * the 1024-leaf subtrees, the root included — may be stale: read the total with hb_tree_total, never as nodes[1].           */
float* hb_tree_nodes(hb_tree* t);
/* copy of the whole heap into nodes_dev[2*cap] (differential tests) */
int hb_tree_export_nodes(hb_tree* t, float* nodes_dev, void* stream);
/* the inverse: overwrite the whole heap with nodes_dev[2*cap] as exported above (checkpoint resume: inner
 * nodes keep the exact fp32 values the incremental updates had produced)                          */
int hb_tree_import_nodes(hb_tree* t, const float* nodes_dev, void* stream);

/* Lazy top levels. Writers (hb_tree_update / hb_per_update with 96..1024 entries, hb_tree_fill_range) rebuild the 1024-leaf
 * subtrees they touch and then re-sum the levels above them in a second, one-workgroup launch. on != 0 drops that launch:
 * the levels above the subtree roots go stale and every READER makes them fresh first — hb_per_sample_gather inside its own
 * workgroups (in LDS, the same pairwise sums: identical bits), the other readers (hb_tree_sample, hb_per_sample[_philox],
 * hb_tree_total, hb_tree_export_nodes) by running the re-sum launch themselves. Results never depend on the mode.
 * on == 0 launches nothing (it has no stream): the levels are re-summed by the next reader or writer, on that call's stream. */
int hb_tree_set_lazy_top(hb_tree* t, int32_t on);

/* update_values(indices, values) (sum_tree.h:38-44). Duplicate indices inside one call:
 * the LAST occurrence wins (the sequential order of the reference loop). n<=0 is a no-op.
 * Out-of-range indices are ignored and counted in hb_tree_error_count().                */
int hb_tree_update(hb_tree* t, const int64_t* idx_dev, const float* val_dev, int64_t n, void* stream);

/* Ring insert used by PriorityBuffer.add_transitions (priority_buffer.py:29-32): leaves
 * (start + i) mod cap for i in [0, n) take *value_dev (a device scalar, so the running
 * max priority never visits the host).                                                  */
int hb_tree_fill_range(hb_tree* t, int64_t start, int64_t n, const float* value_dev, void* stream);

/* get_indices + get_values (sum_tree.h:46-72): for each quantile q in [0,1]:
 * descend from the root with query = q * total, going left iff query < left (strict),
 * else query -= left and right (sum_tree.h:92-105). idx_dev[i] = leaf index,
 * val_dev[i] = its value (nullable).                                                    */
int hb_tree_sample(hb_tree* t, const float* quantile_dev, int64_t* idx_dev, float* val_dev, int64_t n,
                   void* stream);

/* get_values(indices) (sum_tree.h:65-72). Out-of-range -> 0 and counted as an error.    */
int hb_tree_get(hb_tree* t, const int64_t* idx_dev, float* val_dev, int64_t n, void* stream);

/* get_total_val() into a device scalar.                                                 */
int hb_tree_total(hb_tree* t, float* total_dev, void* stream);

int hb_tree_error_count(hb_tree* t, int64_t* out); /* synchronises the stream */

/* ---- fused prioritized-replay helpers (priority_buffer.py:36-52) ----------------------
 * hb_per_sample: keys_i = float(linspace(1/B, 1, B)_i - u_i) with u_i in [0, 1/B) given
 * as doubles (priority_buffer.py:37-40), tree descent, and
 * prob_i = (leaf_i + 1e-10) / total in double (priority_buffer.py:42). unit_uniforms != 0:
 * u_i is in [0, 1) and is divided by B inside the kernel (saves the caller a launch).    */
int hb_per_sample(hb_tree* t, const double* u_dev, int64_t batch, int32_t unit_uniforms, int64_t* idx_dev,
                  double* prob_dev, void* stream);
/* The same with the uniforms drawn inside the kernel (no generator launch, nothing for a captured graph to
 * re-seed): u_i = bits53(Philox4x32-10(key = seed; counter = (i, uint64(*counter_dev)))) * 2^-53 / B with
 * bits53 = (word0 >> 5) << 26 | word1 >> 6. counter_dev is any device scalar that changes between calls (the
 * learner passes its optimizer step count).                                                         */
int hb_per_sample_philox(hb_tree* t, uint64_t seed, const float* counter_dev, int64_t batch, int64_t* idx_dev,
                         double* prob_dev, void* stream);
/* hb_per_sample_philox and hb_replay_gather(_packed) (below) in ONE launch: workgroup i descends the tree for stratum i, then
 * expands the sampled transition's two observation rows into x_dev. Outputs and arithmetic are exactly those of the two
 * separate calls; `packed` != 0: the rings hold bit-packed rows (hb_replay_gather_packed), else int8 rows.              */
int hb_per_sample_gather(hb_tree* t, uint64_t seed, const float* counter_dev, int64_t batch, int64_t* idx_dev, double* prob_dev,
                         const void* ring_obs_tm1_dev, const void* ring_obs_t_dev, const int8_t* ring_act_dev,
                         const float* ring_rew_dev, const uint8_t* ring_term_dev, int32_t obs_len, int32_t packed, void* x_dev,
                         int32_t x_dtype, int32_t x_ld, int32_t* act_dev, float* rew_dev, float* term_dev, float* disc_dev,
                         int32_t n_step, float gamma, int64_t capacity, int64_t rows_per_insert, const int64_t* size_wp_dev,
                         void* stream);
/* hb_per_update: p_i = (|td_i| + 1e-10)^alpha (double pow, rounded to float as the
 * pybind float conversion does), max/min priority tracked in device scalars
 * (priority_buffer.py:48-52), then hb_tree_update.                                      */
int hb_per_update(hb_tree* t, const int64_t* idx_dev, const float* td_dev, int64_t n, double alpha,
                  float* max_prio_dev, float* min_prio_dev, void* stream);

/* ---- fused actor tail (hanabi_agents/rlax_dqn/rlax_rainbow.py:113-122,141-150) ---------
 * One pass over the C51 logits [n, n_actions * n_atoms] (logits_dtype 0 = f32, 1 = bf16,
 * 2 = f16): q = mean(softmax(logits) * support, -1), illegal moves -> -inf, then the legal
 * epsilon-greedy sample (epsilon = 0: greedy with uniform tie-breaking). Randomness is
 * Philox4x32-10(seed; draw, first_game_id + g): word 0 -> explore decision, word 1 -> which
 * candidate. support_dev [n_atoms] f32; actions_dev [n] int32; q_dev [n, n_actions] f32 or
 * NULL (receives the UNMASKED q). row_stride >= n_actions*n_atoms: elements between logits
 * rows (the second GEMM's N is padded 1020 -> 1024 because hipBLASLt is 1.4x faster there).  */
int hb_policy_act(const void* logits_dev, int32_t logits_dtype, const int8_t* legal_dev, const float* support_dev,
                  int64_t n_games, int32_t n_actions, int32_t n_atoms, int32_t row_stride, float epsilon, uint64_t seed,
                  uint64_t draw, int64_t first_game_id, int32_t* actions_dev, float* q_dev, void* stream);

/* int8 0/1 observation matrix [rows, cols] -> 16-bit float GEMM operand (out_dtype 1 = bf16,
 * 2 = f16) whose rows are out_ld >= cols elements apart (K padded 658 -> 704: the first GEMM
 * is 1.4x faster); padding columns are not written — zero them once.                       */
int hb_obs_cast(const int8_t* obs_dev, void* out_dev, int32_t out_dtype, int64_t rows, int32_t cols, int32_t out_ld,
                void* stream);

/* ---- fused replay insert (hanabi_agents/rlax_dqn/rlax_rainbow.py:297-308 +
 *      experience_buffer.py:26-81 for a batch without FIRST rows) --------------------------
 * Row i of the batch goes to ring slot (start + i) mod capacity:
 *   ring_obs_tm1 <- last_obs, ring_obs_t <- obs, ring_lms <- legal, ring_act <- actions,
 *   ring_rew <- rewards, ring_term <- (step_type == 2); then last_obs <- obs.
 * The caller advances its ring pointer (experience_buffer.py:58-61,79-81).                */
int hb_replay_insert(int8_t* last_obs_dev, const int8_t* obs_dev, const int8_t* legal_dev, const int32_t* actions_dev,
                     const float* rewards_dev, const int8_t* step_type_dev, int8_t* ring_obs_tm1_dev,
                     int8_t* ring_obs_t_dev, int8_t* ring_act_dev, int8_t* ring_lms_dev, float* ring_rew_dev,
                     uint8_t* ring_term_dev, int64_t n, int32_t obs_len, int32_t n_actions, int64_t capacity,
                     int64_t start, void* stream);

/* ---- fused learner pieces (hanabi_agents/rlax_dqn/rlax_rainbow.py:152-217) ---------------
 * dtype codes: 0 = f32, 1 = bf16, 2 = f16.
 *
 * Precision contract. dtype selects the type of the GEMM operands only (observations, effective weights, hidden
 * activations; the reference runs its whole network in f16, rlax_rainbow.py:250-251); master weights, Adam moments,
 * softmax / projection / cross-entropy and every accumulation are f32, and since round 3 the bf16 path's logits are never
 * rounded (one-kernel actor: fp32 accumulators; learner: fp32 out of hb_thin_gemm). Against the f32 path on the same batch, weights and
 * sampling probabilities (tests/test_dtype_parity.py asserts these on the MI355X; hanabi_agents/rlax_dqn/tolerance.py):
 *                                                        bf16                      f16
 *   per-sample td (where the double-Q selection is     |d| <= 0.012 + 0.004 |td|  |d| <= 0.002 + 0.0005 |td|
 *     unambiguous in f32: top-2 gap > 4e-3 / 6e-4)
 *   loss mean(td * w)                                   2e-3 relative              1e-4 relative
 *   IS weights                                          1e-6 absolute              1e-6 absolute
 *   merged gradients dW1, db1, dW2, db2 (rel. L2)       0.05                       0.03
 *   actor q = mean_k softmax * atoms (|q| <= 0.49)      0.012 absolute             0.003 absolute
 *   chosen move = f32 arg-max where the f32 top-2 gap   > 0.024                    > 0.006
 * Both types run on the same kernels at the same speed (hb_actor_fused_*_dt, hb_thin_gemm bit 2, hb_chain_run): f16 is the
 * reference's own choice and 8 x finer; bf16 (the default) has f32's exponent range, i.e. needs no loss scaling whatever the
 * gradients do (f16 measured: dLoss/dlogits ~ 1e-5 is subnormal there, the hidden layer's gradient loses bits: 0.02 rel. L2).
 *
 * hb_replay_gather: batch gather experience_buffer.py:83-87 straight into the GEMM operand:
 *   x_dev [2*batch, x_ld >= obs_len] (rows 0..B-1 = obs_tm1[idx], B..2B-1 = obs_t[idx]) in x_dtype,
 *   act_dev [B] int32, rew_dev [B] f32, term_dev [B] f32 (0/1), disc_dev [B] f32 = gamma^m.
 * n_step > 1 assembles n-step transitions at sample time (spec: hanabi_agents/rainbow/replay_memory.py:316-345):
 * the seat's next transition of the same game is rows_per_insert slots further on; the chain stops at an episode
 * end or at the ring's write pointer (size_wp_dev = {entries, next slot to write}, device int64[2]);
 * rew = sum gamma^k r_k, obs_t / term from the last step, disc = gamma^m. n_step = 1 is the reference.      */
int hb_replay_gather(const int8_t* ring_obs_tm1_dev, const int8_t* ring_obs_t_dev, const int8_t* ring_act_dev,
                     const float* ring_rew_dev, const uint8_t* ring_term_dev, const int64_t* idx_dev, int64_t batch,
                     int32_t obs_len, void* x_dev, int32_t x_dtype, int32_t x_ld, int32_t* act_dev, float* rew_dev,
                     float* term_dev, float* disc_dev, int32_t n_step, float gamma, int64_t capacity, int64_t rows_per_insert,
                     const int64_t* size_wp_dev, void* stream);

/* The same gather from bit-packed rings (rows of ceil(obs_len / 32) u32, see hb_env_step_packed): x_dev receives the
 * identical 0 / 1 operand.                                                                                              */
int hb_replay_gather_packed(const uint32_t* ring_bits_tm1_dev, const uint32_t* ring_bits_t_dev, const int8_t* ring_act_dev,
                            const float* ring_rew_dev, const uint8_t* ring_term_dev, const int64_t* idx_dev, int64_t batch,
                            int32_t obs_len, void* x_dev, int32_t x_dtype, int32_t x_ld, int32_t* act_dev, float* rew_dev,
                            float* term_dev, float* disc_dev, int32_t n_step, float gamma, int64_t capacity,
                            int64_t rows_per_insert, const int64_t* size_wp_dev, void* stream);

/* hb_c51_loss_grad: rlax_rainbow.py:172-200 on precomputed logits.
 *   logits_online_dev [2B, row_stride >= A*K]: rows 0..B-1 = online(obs_tm1), rows B..2B-1 = online(obs_t);
 *   logits_target_dev [B, A*K] = target(obs_t); support_dev [K] uniform atoms.
 *   Outputs: td_dev [B] (cross-entropy "TD", its |.| is the new priority), w_dev [B] (IS
 *   weights (1/P)^beta / max), dlogits_dev [B, A*K] = d mean(td * w) / d online(obs_tm1).
 *   disc_dev [B]: per-sample discount (gamma for 1-step, gamma^m for n-step transitions).
 *   mask_terminal != 0 multiplies the discount by (1 - term) (off = the reference, App. C-5).
 *   update_counter_dev: NULL, or a device float that this launch increments by one: a per-update counter for the
 *   caller (the optimizer's step number, see hb_noisy_adam_multi's step_offset) without a launch of its own.
 *   bias_online_dev / bias_target_dev: NULL, or the output layers' biases [A*K] (same dtype as the logits) when the
 *   caller's GEMM has not added them (one batched GEMM for both networks has no bias epilogue).               */
int hb_c51_loss_grad(const void* logits_online_dev, const void* logits_target_dev, int32_t dtype, const int32_t* act_dev,
                     const float* rew_dev, const float* term_dev, const double* prios_dev, const float* beta_dev,
                     const float* disc_dev, int32_t mask_terminal, const float* support_dev, int64_t batch, int32_t n_actions,
                     int32_t n_atoms, int32_t row_stride, float* td_dev, float* w_dev, void* dlogits_dev,
                     float* update_counter_dev, const void* bias_online_dev, const void* bias_target_dev, void* stream);

/* The same loss with the gradient in its natural, compact form (csrc/learner2.hip): dLoss/dlogits is non-zero only in the
 * n_atoms atoms of the action each sample took, so instead of a dense [B, A*K] matrix this writes
 *   dl_dev [B, 64] fp32 (16-byte aligned): dl[b, k] = d mean(td * w) / d online(obs_tm1)[b, act[b], k], zero for k >= n_atoms.
 * Everything else (inputs, td_dev, w_dev, update_counter_dev, biases) as in hb_c51_loss_grad; the selector's softmax
 * expectations are spread over all 64 lanes of the sample's wavefront and the batch maximum of the IS weights comes from
 * the extreme probabilities (x -> x^beta is monotone) instead of B pow calls per sample.                                      */
int hb_c51_loss_sparse(const void* logits_online_dev, const void* logits_target_dev, int32_t dtype, const int32_t* act_dev,
                       const float* rew_dev, const float* term_dev, const double* prios_dev, const float* beta_dev,
                       const float* disc_dev, int32_t mask_terminal, const float* support_dev, int64_t batch, int32_t n_actions,
                       int32_t n_atoms, int32_t row_stride, float* td_dev, float* w_dev, float* dl_dev, float* update_counter_dev,
                       const void* bias_online_dev, const void* bias_target_dev, void* stream);

/* hb_thin_gemm: out[b][m, n] = act(sum_k x[b][m, k] * wt[b][n, k] + bias[b][n]) in bf16 with fp32 accumulation, both operands
 * k-contiguous (wt is the TRANSPOSED weight matrix, as hb_actor_pack_weights writes it), as a kernel small enough (one
 * wavefront per workgroup, 36 VGPRs, no LDS) to run on the CUs WHILE hb_actor_hidden / hb_actor_q hold them: the dense layers
 * of DQNLearning.update_q's forward pass (rlax_rainbow.py:172-185 over noisy_mlp.py:176-185) then proceed during the other
 * seat's policy forward instead of queueing behind it. m % 32 == 0, n % 16 == 0, k % 32 == 0; batch >= 1 with element strides;
 * bias (bf16 [batch][n], may be NULL) and ReLU are applied before the bf16 rounding, like a library GEMM epilogue.
 * relu: bit 0 = ReLU; bit 1 = write fp32 instead of bf16 (ldo / out_batch_stride then count floats, out_dev 16-byte aligned):
 * the learner's logits are the accumulators + bias as they are (round 3), so the loss sees no bf16 rounding of the logits;
 * bit 2 = x, wt, bias and a 16-bit output are fp16 instead of bf16 (v_mfma_f32_16x16x32_f16, the same rate).                  */
int hb_thin_gemm(const void* x_dev, const void* wt_dev, const void* bias_dev, void* out_dev, int64_t m, int32_t n, int32_t k,
                 int32_t ldx, int32_t ldw, int32_t ldo, int32_t batch, int64_t x_batch_stride, int64_t w_batch_stride,
                 int64_t out_batch_stride, int32_t relu, void* stream);

/* hb_dqn_loss_sparse: the scalar double-Q loss of the older agent (hanabi_agents/rlax_dqn/rlax_dqn.py:170-205: td = r + g * (1 -
 * terminal) * q_target(s')[argmax q_online(s')] - q_online(s)[a], loss = mean(w_IS * 0.5 * td^2)) on precomputed q values, in the
 * same compact-gradient form as hb_c51_loss_sparse: q_online_dev [2B, row_stride] (rows 0..B-1: obs_tm1, B..2B-1: obs_t),
 * q_target_dev [B, row_stride] (obs_t), action a at column a * col_stride (col_stride 2 stores the scalar head as a 2-atom head
 * whose second atom is unused, so that hb_c51_backward(n_atoms = 2) computes its backward pass); biases in `dtype` at the same
 * columns (may be NULL). Outputs td [B], IS weights [B], dl [B, 64] with dl[b, 0] = dLoss/dq[b, act[b]]. batch <= 256.       */
int hb_dqn_loss_sparse(const void* q_online_dev, const void* q_target_dev, int32_t dtype, const int32_t* act_dev, const float* rew_dev,
                       const float* term_dev, const double* prios_dev, const float* beta_dev, const float* disc_dev, int64_t batch,
                       int32_t n_actions, int32_t col_stride, int32_t row_stride, float* td_dev, float* w_dev, float* dl_dev,
                       const void* bias_online_dev, const void* bias_target_dev, void* stream);

/* hb_c51_backward: everything between that loss and the first layer's weight gradient, in ONE launch (the `jax.grad` of
 * rlax_rainbow.py:203-213 through the output layer and the ReLU of noisy_mlp.py:176-185), from the compact gradient:
 *   dh_dev  [B, hidden]  (dtype)  = (hidden_dev[b, j] > 0) * sum_k dl[b, k] * w2[j, act[b] * n_atoms + k]
 *   db1_dev [hidden]     fp32     = column sums of dh_dev as stored
 *   dw2_dev [hidden, dw2_ld] (dtype): columns a * n_atoms + k = sum over the samples with act == a of hidden[b, j] * dl[b, k]
 *                                  (zero for an action nobody took; columns >= A * n_atoms are not written)
 *   db2_dev [A * n_atoms] fp32    = sum over those samples of dl[b, k]
 * hidden_dev [B, hidden_ld] are the post-ReLU activations of obs_tm1, w2_dev [hidden, w2_ld] the effective output-layer
 * weights, both in `dtype` (0 f32, 1 bf16, 2 f16); batch <= 256. Every sum runs over samples in ascending order: results are
 * bit-reproducible (no atomics). Replaces two dense GEMMs (dW2, dH), hb_colsum and hb_relu_bwd_colsum.                    */
int hb_c51_backward(const float* dl_dev, const int32_t* act_dev, const void* hidden_dev, int32_t hidden_ld, const void* w2_dev,
                    int32_t w2_ld, int32_t dtype, int64_t batch, int32_t hidden, int32_t n_actions, int32_t n_atoms, void* dh_dev,
                    float* db1_dev, void* dw2_dev, int32_t dw2_ld, float* db2_dev, void* stream);

/* hb_colsum: out_dev[j] = sum_i x[i, j] with fp32 accumulation in a fixed order (bias gradients:
 * the column sums of dLoss/dlogits and of dLoss/dhidden). x_dev [rows, cols] contiguous.        */
int hb_colsum(const void* x_dev, int32_t dtype, int64_t rows, int64_t cols, float* out_dev, void* stream);

/* hb_noisy_adam: one Adam step (optix.adam form, rlax_rainbow.py:257) on the three parameters
 * behind one merged NoisyLinear tensor W = w + w_mu + w_sigma * noise (noisy_mlp.py:61-91), given
 * grad_dev = dLoss/dW (f32); writes the new merged tensor to eff_dev in eff_dtype. step_dev holds
 * the number of steps already taken (the caller increments it once per update). The parameter tensors
 * are [n/cols, cols] row-major; eff_dev rows are eff_ld >= cols elements apart (padded GEMM operand).      */
int hb_noisy_adam(float* w_dev, float* w_mu_dev, float* w_sigma_dev, const float* noise_dev, const float* grad_dev,
                  float* m_w_dev, float* v_w_dev, float* m_mu_dev, float* v_mu_dev, float* m_sigma_dev, float* v_sigma_dev,
                  const float* step_dev, void* eff_dev, int32_t eff_dtype, int64_t n, int32_t cols, int32_t eff_ld, float lr,
                  float beta1, float beta2, float eps, void* stream);

/* The same for up to 8 merged tensors (all layers' weights and biases) in ONE launch. w and w_mu see the same
 * gradient at every step, so their moments stay equal: m_mu == m_w and v_mu == v_w may be passed as the SAME arrays, and
 * w's step is then applied to w_mu without a second pass over the moments.                     */
typedef struct hb_adam_tensor {
  float *w, *w_mu, *w_sigma;
  const float* noise;
  const void* grad;      /* gradient of the merged tensor, dtype grad_dtype (0 f32, 1 bf16, 2 f16): e.g. the GEMM output itself */
  float *m_w, *v_w, *m_mu, *v_mu, *m_sigma, *v_sigma;
  void* eff;
  int64_t n;
  int32_t cols, eff_ld;
  int32_t grad_dtype, grad_ld; /* grad rows are grad_ld >= cols elements apart (0: contiguous)                       */
} hb_adam_tensor;
/* This step's number (Adam's bias correction) is t = *step_dev + step_offset: 1 when step_dev counts completed steps
 * (hb_noisy_adam's convention), 0 when it already counts the running one (see hb_c51_loss_grad's counter).      */
int hb_noisy_adam_multi(const hb_adam_tensor* tensors /* host array of device pointers */, int32_t count,
                        const float* step_dev, float step_offset, int32_t eff_dtype, float lr, float beta1, float beta2,
                        float eps, void* stream);

/* hb_noisy_adam_multi_pack (round 3): the same step for the (up to 4) merged tensors of the one-hidden-layer network, which ALSO
 * writes, in the same pass over the new effective weights, the copies the forward kernels read — what hb_actor_pack_weights (the
 * k-contiguous copies of hb_thin_gemm) and hb_actor_fused_pack (the fragment-major copies of the one-kernel actor) otherwise
 * produce in two more launches per update. eff_dtype 1 (bf16) or 2 (f16). packs[i] describes tensor i's extra outputs:
 *   weight tensors (more than one row): wt = transposed copy [cols][wt_ld] (wt_ld >= rows rounded up to 8), or NULL;
 *     frag + frag_kind = the one-kernel actor's copy: 1 = first layer (W1f; cols must be 512), 2 = output layer (W2f; needs
 *     col_map_dev = hb_actor_fused_columns' physical column of every logit column), 0 / NULL = none;
 *   bias tensors (one row): bias_f32 = fp32 copy of the ROUNDED effective bias, at col_map_dev[j] when that is given, or NULL.
 * Same arithmetic per element as hb_noisy_adam_multi; the copies equal what the two packers make of `eff` bit for bit
 * (tests/test_hip_policy.py).                                                                                                 */
typedef struct hb_adam_pack {
  void* wt;
  void* frag;
  const int32_t* col_map_dev;
  float* bias_f32;
  int32_t wt_ld, frag_kind;
} hb_adam_pack;
int hb_noisy_adam_multi_pack(const hb_adam_tensor* tensors, const hb_adam_pack* packs, int32_t count, const float* step_dev,
                             float step_offset, int32_t eff_dtype, float lr, float beta1, float beta2, float eps, void* stream);

/* ReLU backward fused with the bias gradient (the `jax.grad` of relu + the bias add, rlax_rainbow.py:203-206 through
 * noisy_mlp.py:176-185): dy_dev [rows, cols] is masked in place where act_dev (the post-activation) is <= 0, and
 * out_dev[j] = sum over rows of the masked values (fp32, fixed order).                                          */
int hb_relu_bwd_colsum(void* dy_dev, const void* act_dev, int64_t act_ld /* row stride of act_dev, 0 = cols */, int32_t dtype,
                       int64_t rows, int64_t cols, float* out_dev, void* stream);

/* ---- actor forward on MFMA (csrc/actor.hip): rlax_rainbow.py:113-122,141-150 over noisy_mlp.py:176-185 ------------
 * For the C51 network with ONE hidden layer and bf16 effective weights W = w + w_mu + w_sigma * eps:
 *   hb_actor_pack_weights  the GEMM kernels want both operands k-contiguous: per job, W [k_rows, n_cols] (row stride w_ld,
 *                          bf16) -> wt [n', k_pad] bf16 with n' = (n / group_cols) * 256 + n % group_cols (group_cols = 0:
 *                          n' = n) and bias (bf16 [n_cols]) -> bias_out (fp32, at n'). Entries no input maps to are
 *                          not written: zero-initialise both outputs once. For the output layer use group_cols =
 *                          (256 / n_atoms) * n_atoms (255 for 51 atoms), so that every 256-column tile holds whole actions.
 *   hb_actor_hidden        h_dev [n_rows, hidden] bf16 = relu(obs @ W1 + b1); obs_dev int8 [n_rows, obs_len] with entries in
 *                          0..127 (the canonical encoder emits 0 / 1), widened while staged; w1t_dev [hidden, k_pad],
 *                          k_pad a multiple of 64 >= obs_len, hidden a multiple of 256.
 *   hb_actor_q             q_dev [n_rows, n_actions] fp32 = mean_k(softmax_k(h @ W2 + b2) * support_k) (rlax_rainbow.py:
 *                          117-118); w2t_dev [ceil(n_actions / (256 / n_atoms)) * 256, hidden] packed as above. The logits
 *                          are rounded to bf16 before the softmax, like the output of a bf16 GEMM.
 *   hb_policy_select       the selection half of hb_policy_act on precomputed q (same Philox draws, same tie rule).   */
typedef struct hb_pack_job {
  const void* w;      /* bf16 [k_rows, n_cols], row stride w_ld */
  const void* bias;   /* bf16 [n_cols] */
  void* wt;           /* bf16 [n', k_pad] */
  float* bias_out;    /* fp32 [n'] */
  int32_t k_rows, n_cols, w_ld, group_cols, k_pad;
} hb_pack_job;
int hb_actor_pack_weights(const hb_pack_job* jobs /* host array of device pointers */, int32_t count /* 1..4: one launch */,
                          void* stream);
int hb_actor_hidden(const int8_t* obs_dev, int64_t n_rows, int32_t obs_len, const void* w1t_dev, int32_t k_pad,
                    const float* b1_dev, int32_t hidden, void* h_dev, void* stream);
/* hb_actor_hidden on bit-packed observation rows [n_rows, ceil(obs_len / 32)] u32 (hb_env_step_packed): bit-identical h_dev */
int hb_actor_hidden_packed(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1t_dev, int32_t k_pad,
                           const float* b1_dev, int32_t hidden, void* h_dev, void* stream);
int hb_actor_q(const void* h_dev, int64_t n_rows, int32_t hidden, const void* w2t_dev, const float* b2_dev,
               const float* support_dev, int32_t n_actions, int32_t n_atoms, float* q_dev, void* stream);
int hb_policy_select(const float* q_dev, const int8_t* legal_dev, int64_t n_games, int32_t n_actions, float epsilon,
                     uint64_t seed, uint64_t draw, int64_t first_game_id, int32_t* actions_dev, void* stream);

/* hb_actor_q with the action selection fused in: the last of a 256-row tile's column-group workgroups to finish (ticket
 * counter per tile) applies hb_policy_select's rule to the tile's rows. tickets_dev: ceil(n_rows / 256) uint32, zero before the
 * first call (the kernel re-arms them); NULL: hb_actor_q followed by hb_policy_select (two launches). Identical actions.      */
int hb_actor_q_select(const void* h_dev, int64_t n_rows, int32_t hidden, const void* w2t_dev, const float* b2_dev,
                      const float* support_dev, int32_t n_actions, int32_t n_atoms, float* q_dev, const int8_t* legal_dev,
                      float epsilon, uint64_t seed, uint64_t draw, int64_t first_game_id, int32_t* actions_dev,
                      uint32_t* tickets_dev, void* stream);

/* hb_actor_hidden[_packed] + hb_actor_q_select behind one call (one binding round trip per policy call);
 * obs_is_packed != 0: obs_dev is the bit-row form. Same argument rules as the functions it wraps.                            */
int hb_actor_act(const void* obs_dev, int32_t obs_is_packed, const int8_t* legal_dev, int64_t n_rows, int32_t obs_len,
                 const void* w1t_dev, int32_t k_pad, const float* b1_dev, int32_t hidden, void* h_dev, const void* w2t_dev,
                 const float* b2_dev, const float* support_dev, int32_t n_actions, int32_t n_atoms, float* q_dev, float epsilon,
                 uint64_t seed, uint64_t draw, int64_t first_game_id, int32_t* actions_dev, uint32_t* tickets_dev, void* stream);

/* ---- The whole policy forward as ONE kernel (csrc/actor_fused.hip, round 3): bit-packed observation rows -> q values, the
 * hidden activations never leave the CU and the logits are never rounded (fp32 accumulators -> C51 expectation).
 * Replaces DQNPolicy's forward (hanabi_agents/rlax_dqn/rlax_rainbow.py:113-122 over noisy_mlp.py:176-185) for the reference
 * topology layers=[512], n_atoms=51 (params.py:13,18); other shapes use hb_actor_hidden[_packed] + hb_actor_q.
 *   hb_actor_fused_supported  1 when the shape is covered (hidden == 512, n_atoms == 51, n_actions <= 80, obs_len <= 4096)
 *   hb_actor_fused_sizes      bytes of the two fragment-major weight copies and floats of the physical-order output bias
 *   hb_actor_fused_pack       effective weights (bf16 row-major W1 [>= obs_len, 512] / W2 [512, >= A * 51], bf16 biases) -> the
 *                             copies the kernel streams: one launch; call after every weight change
 *   hb_actor_fused_q          q [n_rows, n_actions] fp32 from obs_bits [n_rows, ceil(obs_len / 32)] u32
 *   hb_actor_fused_act        the same launch followed, inside the kernel, by hb_policy_select's rule on the rows each workgroup
 *                             has just written (same Philox draws, same tie rule: identical actions); n_actions <= 64           */
int hb_actor_fused_supported(int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms);
int hb_actor_fused_sizes(int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms, int64_t* w1f_bytes,
                         int64_t* w2f_bytes, int32_t* b2f_floats);
int hb_actor_fused_pack(const void* w1_dev, int32_t w1_ld, const void* b1_dev, const void* w2_dev, int32_t w2_ld,
                        const void* b2_dev, int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms, void* w1f_dev,
                        float* b1f_dev, void* w2f_dev, float* b2f_dev, void* stream);
int hb_actor_fused_q(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev, const float* b1f_dev,
                     const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden, int32_t n_actions,
                     int32_t n_atoms, float* q_dev, void* stream);
int hb_actor_fused_act(const uint32_t* obs_bits_dev, const int8_t* legal_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev,
                       const float* b1f_dev, const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden,
                       int32_t n_actions, int32_t n_atoms, float* q_dev, float epsilon, uint64_t seed, uint64_t draw,
                       int64_t first_game_id, int32_t* actions_dev, void* stream);
/* Physical column (0 .. 512 * ceil(n_actions / 10) - 1) of every logit column action * 51 + atom in the one-kernel actor's
 * output-layer copy / bias: n_actions * 51 ints into HOST memory (for hb_noisy_adam_multi_pack's col_map_dev: upload it).   */
int hb_actor_fused_columns(int32_t n_actions, int32_t* phys_of_logit_host);
/* The same three with the operand type of the 16-bit weights / hidden activations as an argument: dtype 1 = bf16 (what the
 * functions above use), 2 = fp16 — the reference's own network dtype (rlax_rainbow.py:250-251), same MFMA rate
 * (v_mfma_f32_16x16x32_f16), 8 x finer rounding of weights and hidden activations (Precision contract above). Copies packed
 * with one dtype must be run with the same one.                                                                               */
int hb_actor_fused_pack_dt(const void* w1_dev, int32_t w1_ld, const void* b1_dev, const void* w2_dev, int32_t w2_ld,
                           const void* b2_dev, int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms, void* w1f_dev,
                           float* b1f_dev, void* w2f_dev, float* b2f_dev, int32_t dtype, void* stream);
/* hb_actor_fused_pack_dt that ALSO writes the k-contiguous (transposed) copies hb_thin_gemm reads — w1t_dev [hidden][w1t_ld >=
 * obs_len rounded up to 64], w2t_dev [n_actions * n_atoms][w2t_ld >= hidden], 16-bit, either may be NULL — so that an update
 * needs ONE pack launch instead of hb_actor_pack_weights + hb_actor_fused_pack (round 3: 5.5 us and a hand-over per update).   */
int hb_actor_fused_pack_thin(const void* w1_dev, int32_t w1_ld, const void* b1_dev, const void* w2_dev, int32_t w2_ld,
                             const void* b2_dev, int32_t obs_len, int32_t hidden, int32_t n_actions, int32_t n_atoms,
                             void* w1f_dev, float* b1f_dev, void* w2f_dev, float* b2f_dev, void* w1t_dev, int32_t w1t_ld,
                             void* w2t_dev, int32_t w2t_ld, int32_t dtype, void* stream);
int hb_actor_fused_q_dt(const uint32_t* obs_bits_dev, int64_t n_rows, int32_t obs_len, const void* w1f_dev, const float* b1f_dev,
                        const void* w2f_dev, const float* b2f_dev, const float* support_dev, int32_t hidden, int32_t n_actions,
                        int32_t n_atoms, float* q_dev, int32_t dtype, void* stream);
int hb_actor_fused_act_dt(const uint32_t* obs_bits_dev, const int8_t* legal_dev, int64_t n_rows, int32_t obs_len,
                          const void* w1f_dev, const float* b1f_dev, const void* w2f_dev, const float* b2f_dev,
                          const float* support_dev, int32_t hidden, int32_t n_actions, int32_t n_atoms, float* q_dev, float epsilon,
                          uint64_t seed, uint64_t draw, int64_t first_game_id, int32_t* actions_dev, int32_t dtype, void* stream);

/* ---- One host call per step: hb_chain_run (csrc/chain.hip, round 3) ------------------------------------------------------
 * The session that drives DQNAgent (rlax_rainbow.py:277-339: explore / add_experience / update once per env step) issues ~25
 * launches, event waits and event records per step. A host fills an array of hb_cmd ONCE with every pointer, size and stream of
 * one step and then replays it with one call per step, passing what changes (ring position, draw counter, epsilon ...) in two
 * small variable arrays. Each command forwards to the function of the same name above with
 *   p[] / i[] / f[]  its fixed arguments in declaration order (pointers / integers / floating point),
 *   var, fvar        index into vars_i / vars_f of its per-run arguments (-1: none),
 *   cond             index into vars_i of a run / skip switch (-1: always run), stream the hipStream_t it is issued on.
 * Per-run arguments: HB_CMD_REPLAY_INSERT vars_i[var] = start; HB_CMD_ACTOR_FUSED_ACT vars_i[var] = draw, vars_f[fvar] = epsilon
 * (i[5] = seed, i[6] = first_game_id, i[7] = operand dtype: 0 / 1 = bf16, 2 = f16; HB_CMD_ACTOR_FUSED_PACK: i[6] likewise); HB_CMD_TREE_FILL_RANGE vars_i[var], vars_i[var + 1] = start, n (n = 0: skipped);
 * HB_CMD_GRAPH_LAUNCH p[0] = a hipGraphExec_t of the caller (the captured learner update); events are hb_event_create handles.
 * Nothing is computed here: results are those of the individual calls.                                                        */
enum {
  HB_CMD_WAIT_EVENT = 1,       /* hb_stream_wait_event(stream, p[0]) */
  HB_CMD_RECORD_EVENT = 2,     /* hb_event_record(p[0], stream) */
  HB_CMD_REPLAY_INSERT = 3,    /* p[0..11], i[0..3] = n, obs_len (bytes per row), n_actions, capacity */
  HB_CMD_ACTOR_FUSED_ACT = 4,  /* p = obs_bits, legal, w1f, b1f, w2f, b2f, support, q, actions; i = n_rows, obs_len, hidden, A, atoms, seed, first_gid */
  HB_CMD_ENV_STEP_PACKED = 5,  /* p = env, actions, obs_bits, obs, legal, reward, terminal, agent_reward, agent_step_type, score */
  HB_CMD_TREE_FILL_RANGE = 6,  /* p = tree, value */
  HB_CMD_PER_SAMPLE_GATHER = 7,/* p = tree, counter, idx, prob, obs_tm1, obs_t, act, rew, term, x, act_out, rew_out, term_out, disc_out, size_wp;
                                  i = seed, batch, obs_len, packed, x_dtype, x_ld, n_step, capacity, rows_per_insert; f[0] = gamma */
  HB_CMD_GRAPH_LAUNCH = 8,     /* p = hipGraphExec_t */
  HB_CMD_PER_UPDATE = 9,       /* p = tree, idx, td, max_prio, min_prio; i[0] = n; f[0] = alpha */
  HB_CMD_ACTOR_FUSED_PACK = 10,/* p = w1, b1, w2, b2, w1f, b1f, w2f, b2f; i = w1_ld, w2_ld, obs_len, hidden, A, atoms */
  HB_CMD_ACTOR_PACK_WEIGHTS = 11 /* p[0] = hb_pack_job array (host memory), i[0] = count */
};
typedef struct hb_cmd {
  int32_t op, var, fvar, cond;
  void* stream;
  void* p[16];
  int64_t i[10];
  double f[2];
} hb_cmd;
int hb_chain_run(const hb_cmd* cmds, int32_t count, const int64_t* vars_i, const double* vars_f);

#ifdef __cplusplus
}
#endif
#endif /* HANABI_HIP_H */
