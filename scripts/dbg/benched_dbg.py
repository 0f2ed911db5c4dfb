import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd")); sys.path.insert(0, ROOT)
os.environ["HB_ACTOR_FUSED_MIN_ROWS"] = sys.argv[2] if len(sys.argv) > 2 else "0"
import hanabi_hip
from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
from hanabi_hip.selfplay import SelfPlaySession
native = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, steps = 2048, 26
flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=n, seed=3, packed=True)
params = RlaxRainbowParams(train_batch_size=256, experience_buffer_size=n * 16, layers=[512], mask_terminal=True,
                           compute_dtype="bfloat16", packed_obs=True, actor_lag=0, target_update_period=5)
agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda") for s in (1, 2)]
sess = SelfPlaySession(env, agents, native_chain=bool(native))
acts = [[], []]
for t in range(steps):
    seat = t % 2
    ns = sess.native_steps
    st = agents[seat].experience.oldest_entry
    sess.step()
    print("t", t, "seat", seat, "native" if sess.native_steps > ns else "python", "start", st, "train_step", agents[seat].train_step)
    torch.cuda.synchronize()
    acts[seat].append(sess.last_actions[seat].cpu().numpy().copy())
sess.flush(); torch.cuda.synchronize()
print("native steps", sess.native_steps, "grad", sess.grad_steps)
for seat in (0, 1):
    buf = agents[seat].experience
    ring = buf._act_tm1_buf[:buf.size, 0].cpu().numpy().reshape(-1, n)
    for j in range(ring.shape[0]):
        bad = np.nonzero(ring[j] != acts[seat][j])[0]
        nxt = (ring[j][bad] == acts[seat][j + 1][bad]).mean() if len(bad) and j + 1 < len(acts[seat]) else -1
        eqs = {f"{ss}{jj:+d}": round(float((ring[j] == acts[ss][j + jj]).mean()), 3) for ss in (0, 1) for jj in (-1, 0, 1) if 0 <= j + jj < len(acts[ss])}
        print(f"seat {seat} transition {j}: {eqs}")
        print(f"seat {seat} transition {j}: {len(bad)} mismatches; equal to NEXT turn's action: {nxt}; first idx {bad[:6]}")
