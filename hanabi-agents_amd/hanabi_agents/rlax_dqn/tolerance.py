"""Stated tolerances of the reduced-precision Q-net arithmetic against the fp32 path.

`north_star`: "Q-loss within a stated fp32 tolerance". The reference itself runs its network in
float16 (parameters take the dtype of the fp16 init input, hanabi_agents/rlax_dqn/rlax_rainbow.py:
250-251, noisy_mlp.py:55-84); here the master weights, Adam moments, the softmax / projection /
cross-entropy and every accumulation are fp32 and only the GEMM operands (observations, effective
weights, hidden activations, logits, dlogits, weight gradients as the GEMMs emit them) are rounded
to `compute_dtype`. The numbers below bound the effect of that rounding; they are what
tests/test_dtype_parity.py asserts on the MI355X for the 2-player (658 -> 512 -> 20x51) and
5-player (1280 -> 512 -> 48x51) nets, `bench.py` quotes them in its JSON line ("tolerance") and
include/hanabi_hip.h / DESIGN.md §6 repeat them.

All "rel_l2" figures are ||x_lp - x_fp32||_2 / ||x_fp32||_2 over the whole tensor.
"""

TOLERANCE = {
    "bfloat16": {
        # learner, one update on the same batch / weights / sampling probabilities (FusedLearner vs DQNLearning.loss fp32)
        "td_abs": 0.03,            # |td_lp - td_fp32| <= td_abs + td_rel * |td_fp32| per sample (td = C51 cross-entropy, ~3.9)
        "td_rel": 0.01,
        "loss_rel": 5e-3,          # mean(td * w_IS)
        "is_weight_abs": 1e-6,     # IS weights never see the GEMM dtype
        "grad_rel_l2": 0.03,       # dW1, db1, dW2, db2 (merged tensors), each
        "weights_after_5_steps_rel_l2_of_delta": 0.25,   # ||dw_lp - dw_fp32|| / ||dw_fp32||, dw = w_after - w_before (Adam's
                                                         # normalised step amplifies sign flips of near-zero gradients)
        "weights_after_5_steps_max_abs": 0.0101,         # <= 2 * lr * steps: no element can be further apart than that
        # actor: q = mean_k softmax(logits) * atoms (|q| <= 0.49), MFMA kernels vs DQNPolicy.q_values fp32
        "q_abs": 2e-3,
        "argmax_gap": 4e-3,        # wherever the fp32 top-2 gap over legal moves exceeds this, the chosen move is the fp32 arg-max
    },
    "float16": {
        "td_abs": 0.005,
        "td_rel": 0.002,
        "loss_rel": 1e-3,
        "is_weight_abs": 1e-6,
        "grad_rel_l2": 0.005,
        "weights_after_5_steps_rel_l2_of_delta": 0.08,
        "weights_after_5_steps_max_abs": 0.0101,
        "q_abs": 3e-4,
        "argmax_gap": 6e-4,
    },
}
