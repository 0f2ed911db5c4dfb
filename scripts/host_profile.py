"""cProfile of the host side of the self-play step (where do the ~0.15 ms of enqueue time go?)."""
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '2')
import cProfile, pstats, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import hanabi_hip
from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
from hanabi_hip.selfplay import SelfPlaySession

n = 32768
flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=n, seed=1234, packed=True)
lag = int(sys.argv[1]) if len(sys.argv) > 1 else 0
spa = bool(int(sys.argv[2])) if len(sys.argv) > 2 else None
params = RlaxRainbowParams(compute_dtype="bfloat16", mask_terminal=True, packed_obs=True, actor_lag=lag)
agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=1234 + 17 * s), device="cuda") for s in (0, 1)]
if os.environ.get("HB_FORCE_COLLECTIVE") == "1":   # the multi-rank update path on a one-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for a in agents:
        a.force_collective = True
sess = SelfPlaySession(env, agents, stream_per_agent=spa)
for _ in range(60):
    sess.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(400):
    sess.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
