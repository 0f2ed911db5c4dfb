"""One agent's update() alone on the GPU (no actor, no env running beside it): ms per update, and — under
`rocprofv3 --kernel-trace --stats` — the standalone duration of every kernel in the learner chain."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hanabi-agents_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hanabi_hip  # noqa: E402
from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams  # noqa: E402
from hanabi_hip.selfplay import SelfPlaySession  # noqa: E402

players = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", players, flags), n_games=n, seed=1234, packed=True)
params = RlaxRainbowParams(compute_dtype="bfloat16", mask_terminal=True, packed_obs=True)
agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=1234 + 17 * s), device="cuda")
          for s in range(players)]
sess = SelfPlaySession(env, agents)
sess.run(4 * players)
torch.cuda.synchronize()
a = agents[0]
for _ in range(20):
    a.update()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(reps):
    a.update()
e1.record()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{players}p: update() alone: {e0.elapsed_time(e1) / reps * 1e3:.1f} us on the GPU, host enqueue {host / reps * 1e6:.1f} us per call; "
      f"loss {float(a.last_loss):.4f}")
