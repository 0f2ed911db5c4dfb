"""Agent hyper-parameters.

Same field names, order and defaults as the reference's gin-configurable NamedTuple
(hanabi_agents/rlax_dqn/params.py:8-20), so `RlaxRainbowParams()` / `params._replace(...)` written
for the reference keep working. gin is not required; when it is installed the class is registered
with it exactly like the original. Fields after `beta_is` are additions of this implementation and
default to the reference's behaviour.
"""
from typing import Callable, List, NamedTuple, Union

Schedule = Union[Callable[[int], float], float]


def _const(v):
    return lambda train_step: v


class RlaxRainbowParams(NamedTuple):
    train_batch_size: int = 256
    target_update_period: int = 500
    discount: float = 0.99
    epsilon: Schedule = _const(0.1)          # float or callable(train_step)
    learning_rate: float = 0.001
    layers: List[int] = [512]
    use_double_q: bool = True                # accepted and ignored, as in the reference
    use_priority: bool = True
    experience_buffer_size: int = 2 ** 19
    seed: int = 1234
    n_atoms: int = 51
    atom_vmax: int = 25
    beta_is: Schedule = _const(0.4)          # float or callable(train_step)
    # ---- additions (defaults reproduce the reference) ---------------------------------------------
    mask_terminal: bool = False              # True: no bootstrap through episode ends (reference ignores terminal_t, C-5)
    resample_noise: bool = False             # True: fresh NoisyLinear noise every call (reference noise is frozen, C-2)
    compute_dtype: str = "float32"           # "float32" | "bfloat16" | "float16": GEMM input dtype, fp32 accumulate/master
    distributional: bool = True              # False: scalar double-DQN head (rlax_dqn.py:170-205 spec, BASELINE config 2)
    n_step: int = 1                          # >1: n-step returns assembled at sample time (rainbow/replay_memory.py:316-345 spec);
                                             #     needs inserts of a constant row count (the lock-step self-play driver)
    packed_obs: bool = False                 # True: last_obs and the observation rings hold bit-packed rows (bitpack.py; 84 B instead of
                                             #     658 B per observation) and the actor / learner kernels unpack while staging;
                                             #     inputs and outputs of the agent API accept / return either form
    global_is_max: bool = False              # data-parallel only: normalise the IS weights by the max over ALL ranks' batches (one
                                             #     extra 4-byte all-reduce(max) per update), i.e. exactly the reference's w /= max(w)
                                             #     over the global batch (rlax_rainbow.py:188-189); False: per-rank max
    actor_lag: int = 0                       # 0: the actor always uses the newest weights (the reference; it waits for its own update).
                                             # 1: asynchronous actor: double-buffered actor weights, the policy acts on
                                             #     the weights of the update BEFORE the most recent one, so acting never waits for the
                                             #     learner; the one-update staleness is the only difference (tests/test_async_actor.py)


try:  # optional: same registration the reference performs (params.py:4)
    import gin

    RlaxRainbowParams = gin.configurable(RlaxRainbowParams)
except Exception:  # gin is not installed in this image
    pass
