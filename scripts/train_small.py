"""Learning sanity check: self-play on Hanabi-Small; the mean episode score should rise above the random level."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import hanabi_hip
from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
from hanabi_hip.selfplay import SelfPlaySession
game = sys.argv[1] if len(sys.argv) > 1 else "Hanabi-Small"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
ups = int(sys.argv[4]) if len(sys.argv) > 4 else 4
flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
lag = int(os.environ.get("HB_ACTOR_LAG", "0"))   # 1: asynchronous actor (RlaxRainbowParams.actor_lag)
packed = bool(int(os.environ.get("HB_PACKED", "1")))
players = int(os.environ.get("HB_PLAYERS", "2"))
env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config(game, players, flags), n_games=n, seed=int(os.environ.get("HB_SEED", "1")), packed=packed)
params = RlaxRainbowParams(compute_dtype=os.environ.get("HB_DTYPE", "bfloat16"),   # (float16: the reference's own network dtype)
                           mask_terminal=True, experience_buffer_size=2**18, learning_rate=2.5e-4,
                           epsilon=lambda ts: max(0.02, 1.0 - ts / 3000.0), target_update_period=200, atom_vmax=10,
                           packed_obs=packed, actor_lag=lag)
agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda") for s in range(1, players + 1)]
sess = SelfPlaySession(env, agents, updates_per_step=ups, fuse_select=bool(int(os.environ.get('HB_FUSE_SELECT', '1'))),
                       split_update=bool(int(os.environ.get('HB_SPLIT', '1'))))
t0 = time.time(); last = env.stats()
for k in range(steps // 500):
    sess.run(500)
    ep, sc = env.stats()
    print(f"step {sess.t:6d}  episodes {ep-last[0]:7d}  mean score {(sc-last[1])/max(1,ep-last[0]):.3f}  loss {float(agents[0].last_loss):.3f}  "
          f"eps {params.epsilon(agents[0].train_step):.2f}  {time.time()-t0:.0f}s", flush=True)
    last = (ep, sc)
