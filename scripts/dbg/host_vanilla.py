"""cProfile of the host side of a vanilla (config 2) self-play step."""
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '2')
import cProfile, pstats, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import hanabi_hip
from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
from hanabi_hip.selfplay import SelfPlaySession
n = 4096
flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=n, seed=1234, packed=True)
params = RlaxRainbowParams(compute_dtype="bfloat16", mask_terminal=True, packed_obs=True)._replace(distributional=False, use_priority=False)
agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=1234 + 17 * s), device="cuda") for s in (0, 1)]
sess = SelfPlaySession(env, agents)
for _ in range(80):
    sess.step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(400):
    sess.step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host ms/step", host / 400 * 1e3, "total ms/step", (time.perf_counter() - t0) / 400 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(400):
    sess.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
