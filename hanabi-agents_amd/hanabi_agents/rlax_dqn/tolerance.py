"""Stated tolerances of the reduced-precision Q-net arithmetic against the fp32 path.

`north_star`: "Q-loss within a stated fp32 tolerance". The reference itself runs its network in
float16 (parameters take the dtype of the fp16 init input, hanabi_agents/rlax_dqn/rlax_rainbow.py:
250-251, noisy_mlp.py:55-84); here the master weights, Adam moments, the softmax / projection /
cross-entropy and every accumulation are fp32 and only the GEMM operands (observations, effective
weights, hidden activations, weight gradients as the GEMMs emit them) are rounded to `compute_dtype`.
Since round 3 the LOGITS are no longer rounded on the benched bf16 path: the one-kernel actor takes
the C51 expectation from its fp32 accumulators, the two-kernel actor stages them as fp16, and the
learner's forward (hb_thin_gemm) hands the loss fp32 logits; what is left is the rounding of the
weights and of the hidden activations themselves (bf16: 8 significant bits), which no kernel can
remove without changing the operand type. The numbers below bound the effect; they are what
tests/test_dtype_parity.py asserts on the MI355X for the 2-player (658 -> 512 -> 20x51) and
5-player (1280 -> 512 -> 48x51) nets, `bench.py` quotes them in its JSON line ("tolerance") and
include/hanabi_hip.h / DESIGN.md §6 repeat them.

All "rel_l2" figures are ||x_lp - x_fp32||_2 / ||x_fp32||_2 over the whole tensor.
"""

TOLERANCE = {
    "bfloat16": {
        # learner, one update on the same batch / weights / sampling probabilities (FusedLearner vs DQNLearning.loss fp32)
        "td_abs": 0.012,           # |td_lp - td_fp32| <= td_abs + td_rel * |td_fp32| per sample (td = C51 cross-entropy, ~4.4) on
        "selection_gap": 4e-3,     # (fresh-net q values differ by ~1e-2: 86 % of the samples are unambiguous at this gap)
        "td_rel": 0.004,           # every sample whose double-Q selection is unambiguous in fp32 (top-2 q gap > selection_gap);
                                   # a near-tie may select the other action, which swaps that sample's whole target
                                   # (measured worst, r03: 0.0089 / 0.0056 (2 / 5 players) on clear samples, 0.32 on a flipped one)
        "loss_rel": 2e-3,          # mean(td * w_IS)                                              (measured 2.8e-4)
        "is_weight_abs": 1e-6,     # IS weights never see the GEMM dtype                          (measured 0)
        "grad_rel_l2": 0.05,       # dW1, db1, dW2, db2 (merged tensors), each, flipped samples included (measured <= 0.036)
        "weights_after_5_steps_rel_l2_of_delta": 0.12,   # ||dw_lp - dw_fp32|| / ||dw_fp32||, dw = w_after - w_before (Adam's
                                                         # normalised step amplifies sign flips of near-zero gradients; 0.070)
        "weights_after_5_steps_max_abs": 0.0101,         # <= 2 * lr * steps: no element can be further apart than that
        # actor: q = mean_k softmax(logits) * atoms (|q| <= 0.49), MFMA kernels vs DQNPolicy.q_values fp32, output layer scaled x4
        # so that the logits are as large as a trained net's. Round 2 (logits rounded to bf16 in LDS): bound 0.03, measured 0.017.
        "q_abs": 0.012,            # (measured r03: max 0.0097 / 0.0085 (2 / 5 players), mean 0.0007, both actor forms; against the fp32
                                   #  forward of the SAME bf16 weights the one-kernel actor is within 3e-4: tests/test_actor_fused.py)
        "argmax_gap": 0.024,       # = 2 q_abs: wherever the fp32 top-2 gap over legal moves exceeds this, the chosen move is the fp32 arg-max
    },
    # the reference's own network dtype (rlax_rainbow.py:250-251). Round 3: a first-class path on the same kernels as bf16 —
    # hb_actor_fused_*_dt (v_mfma_f32_16x16x32_f16, the same MFMA rate), hb_thin_gemm with fp16 operands, one host call per step;
    # `bench.py --compute-dtype float16` and the default line's "fp16_operands" key time it (same step time as bf16)
    "float16": {
        "td_abs": 0.002,           # (measured worst 0.0010, r03 on the thin forward: fp32 logits)
        "selection_gap": 6e-4,
        "td_rel": 0.0005,
        "loss_rel": 1e-4,          # (measured 1.8e-6)
        "is_weight_abs": 1e-6,
        "grad_rel_l2": 0.03,       # (measured: output layer 4e-4, hidden layer 0.020 — dLoss/dlogits ~ 1e-5 is subnormal in
                                   # fp16 and no loss scaling is applied, so dH loses bits; bf16 has the range, not the bits)
        "weights_after_5_steps_rel_l2_of_delta": 0.12,
        "weights_after_5_steps_max_abs": 0.0101,
        "q_abs": 0.003,            # (measured r03: one-kernel actor 0.0011 / 0.0010 (2 / 5 players); cast + library GEMMs 0.0019 / 0.0020)
        "argmax_gap": 0.006,       # = 2 q_abs: 94 % / 91 % of the rows of a fresh net lie above it (bf16: 83 % / 75 % above 0.024)
    },
}
