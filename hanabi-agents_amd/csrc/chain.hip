// chain.hip — hb_chain_run: a fixed list of this library's own launches issued by ONE host call.
//
// The self-play loop issues ~25 host calls per env step (event waits / records, replay insert, policy kernel, env kernel, tree
// fill, sample + gather, the update's graph, priority write-back). Through a Python binding each costs 3-8 us of interpreter and
// argument-conversion time, and since round 3's one-kernel actor the loop is bound by exactly that (host enqueue 0.117 ms per
// step against ~0.1 ms of GPU work). A host program in the reference's position (the external session that drives DQNAgent,
// hanabi_agents/rlax_dqn/rlax_rainbow.py:277-339) fills an array of hb_cmd ONCE — every pointer, size and stream of one step —
// and then replays it with hb_chain_run, passing only what changes from step to step (ring position, draw counter, epsilon ...)
// as small variable arrays. Nothing here computes anything: every command forwards to the C-ABI function of the same name, so
// results are those of the individual calls by construction (tests/test_selfplay.py runs both forms against each other).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/hanabi_hip.h"
#include "common.hpp"

using hb::fail;

extern "C" {

int hb_chain_run(const hb_cmd* cmds, int32_t count, const int64_t* vars_i, const double* vars_f) {
  if (!cmds || count < 0) return fail(HB_ERR_INVALID, "null argument");
  for (int32_t k = 0; k < count; ++k) {
    const hb_cmd& c = cmds[k];
    // a command with cond >= 0 runs only when vars_i[cond] != 0
    if (c.cond >= 0 && (!vars_i || vars_i[c.cond] == 0)) continue;
    const int64_t* vi = c.var >= 0 ? vars_i + c.var : nullptr;
    const double* vf = c.fvar >= 0 ? vars_f + c.fvar : nullptr;
    if ((c.var >= 0 && !vars_i) || (c.fvar >= 0 && !vars_f)) return fail(HB_ERR_INVALID, "command %d needs a variable array", k);
    int rc = HB_OK;
    switch (c.op) {
      case HB_CMD_WAIT_EVENT:
        rc = hb_stream_wait_event(c.stream, c.p[0]);
        break;
      case HB_CMD_RECORD_EVENT:
        rc = hb_event_record(c.p[0], c.stream);
        break;
      case HB_CMD_REPLAY_INSERT:   // vars_i[var] = ring position of the first inserted row
        if (!vi) return fail(HB_ERR_INVALID, "command %d: hb_replay_insert needs its start variable", k);
        rc = hb_replay_insert(static_cast<int8_t*>(c.p[0]), static_cast<const int8_t*>(c.p[1]), static_cast<const int8_t*>(c.p[2]),
                              static_cast<const int32_t*>(c.p[3]), static_cast<const float*>(c.p[4]),
                              static_cast<const int8_t*>(c.p[5]), static_cast<int8_t*>(c.p[6]), static_cast<int8_t*>(c.p[7]),
                              static_cast<int8_t*>(c.p[8]), static_cast<int8_t*>(c.p[9]), static_cast<float*>(c.p[10]),
                              static_cast<uint8_t*>(c.p[11]), c.i[0], static_cast<int32_t>(c.i[1]), static_cast<int32_t>(c.i[2]),
                              c.i[3], vi[0], c.stream);
        break;
      case HB_CMD_ACTOR_FUSED_ACT:   // vars_i[var] = draw counter, vars_f[fvar] = epsilon
        if (!vi || !vf) return fail(HB_ERR_INVALID, "command %d: hb_actor_fused_act needs its draw and epsilon variables", k);
        // i[7]: operand dtype of the weight copies (0 or 1 = bf16, 2 = f16): hb_actor_fused_act_dt
        rc = hb_actor_fused_act_dt(static_cast<const uint32_t*>(c.p[0]), static_cast<const int8_t*>(c.p[1]), c.i[0],
                                   static_cast<int32_t>(c.i[1]), c.p[2], static_cast<const float*>(c.p[3]), c.p[4],
                                   static_cast<const float*>(c.p[5]), static_cast<const float*>(c.p[6]), static_cast<int32_t>(c.i[2]),
                                   static_cast<int32_t>(c.i[3]), static_cast<int32_t>(c.i[4]), static_cast<float*>(c.p[7]),
                                   static_cast<float>(vf[0]), static_cast<uint64_t>(c.i[5]), static_cast<uint64_t>(vi[0]), c.i[6],
                                   static_cast<int32_t*>(c.p[8]), c.i[7] == 2 ? 2 : 1, c.stream);
        break;
      case HB_CMD_ENV_STEP_PACKED:
        rc = hb_env_step_packed(static_cast<hb_env*>(c.p[0]), static_cast<const int32_t*>(c.p[1]), static_cast<uint32_t*>(c.p[2]),
                                static_cast<int8_t*>(c.p[3]), static_cast<int8_t*>(c.p[4]), static_cast<float*>(c.p[5]),
                                static_cast<int8_t*>(c.p[6]), static_cast<float*>(c.p[7]), static_cast<int8_t*>(c.p[8]),
                                static_cast<int8_t*>(c.p[9]), c.stream);
        break;
      case HB_CMD_TREE_FILL_RANGE:   // vars_i[var], vars_i[var + 1] = first leaf, number of leaves (0: nothing to do)
        if (!vi) return fail(HB_ERR_INVALID, "command %d: hb_tree_fill_range needs its range variables", k);
        if (vi[1] > 0) rc = hb_tree_fill_range(static_cast<hb_tree*>(c.p[0]), vi[0], vi[1], static_cast<const float*>(c.p[1]), c.stream);
        break;
      case HB_CMD_PER_SAMPLE_GATHER:
        rc = hb_per_sample_gather(static_cast<hb_tree*>(c.p[0]), static_cast<uint64_t>(c.i[0]), static_cast<const float*>(c.p[1]), c.i[1],
                                  static_cast<int64_t*>(c.p[2]), static_cast<double*>(c.p[3]), c.p[4], c.p[5],
                                  static_cast<const int8_t*>(c.p[6]), static_cast<const float*>(c.p[7]),
                                  static_cast<const uint8_t*>(c.p[8]), static_cast<int32_t>(c.i[2]), static_cast<int32_t>(c.i[3]), c.p[9],
                                  static_cast<int32_t>(c.i[4]), static_cast<int32_t>(c.i[5]), static_cast<int32_t*>(c.p[10]),
                                  static_cast<float*>(c.p[11]), static_cast<float*>(c.p[12]), static_cast<float*>(c.p[13]),
                                  static_cast<int32_t>(c.i[6]), static_cast<float>(c.f[0]), c.i[7], c.i[8],
                                  static_cast<const int64_t*>(c.p[14]), c.stream);
        break;
      case HB_CMD_GRAPH_LAUNCH:   // p[0]: a hipGraphExec_t the caller instantiated (e.g. the captured learner update)
        if (!c.p[0]) return fail(HB_ERR_INVALID, "command %d: null graph", k);
        HB_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(c.p[0]), static_cast<hipStream_t>(c.stream)));
        break;
      case HB_CMD_PER_UPDATE:
        rc = hb_per_update(static_cast<hb_tree*>(c.p[0]), static_cast<const int64_t*>(c.p[1]), static_cast<const float*>(c.p[2]), c.i[0],
                           c.f[0], static_cast<float*>(c.p[3]), static_cast<float*>(c.p[4]), c.stream);
        break;
      case HB_CMD_ACTOR_FUSED_PACK:
        // i[6]: dtype of the 16-bit operands (0 or 1 = bf16, 2 = f16): hb_actor_fused_pack_dt
        // p[8], i[7] / p[9], i[8]: the thin GEMMs' transposed copies and their row strides (NULL: none): hb_actor_fused_pack_thin
        rc = hb_actor_fused_pack_thin(c.p[0], static_cast<int32_t>(c.i[0]), c.p[1], c.p[2], static_cast<int32_t>(c.i[1]), c.p[3],
                                      static_cast<int32_t>(c.i[2]), static_cast<int32_t>(c.i[3]), static_cast<int32_t>(c.i[4]),
                                      static_cast<int32_t>(c.i[5]), c.p[4], static_cast<float*>(c.p[5]), c.p[6], static_cast<float*>(c.p[7]),
                                      c.p[8], static_cast<int32_t>(c.i[7]), c.p[9], static_cast<int32_t>(c.i[8]), c.i[6] == 2 ? 2 : 1, c.stream);
        break;
      case HB_CMD_ACTOR_PACK_WEIGHTS:   // p[0]: the caller's hb_pack_job array (host memory that outlives the chain), i[0]: count
        rc = hb_actor_pack_weights(static_cast<const hb_pack_job*>(c.p[0]), static_cast<int32_t>(c.i[0]), c.stream);
        break;
      default:
        return fail(HB_ERR_INVALID, "command %d: unknown op %d", k, c.op);
    }
    if (rc != HB_OK) return rc;
  }
  return HB_OK;
}

}  // extern "C"
