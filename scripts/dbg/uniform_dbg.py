import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
n, obs_len, n_act = 512, 658, 20
params = RlaxRainbowParams(distributional=False, use_priority=False, train_batch_size=64, experience_buffer_size=512, target_update_period=4, compute_dtype="float32")
agents = [DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=g) for g in (True, False)]
g = torch.Generator(device="cuda").manual_seed(2)
o1 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
o2 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")
act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
seen = [[], []]
for i, a in enumerate(agents):
    a.add_experience_first((None, (o1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
    a.add_experience((None, (o2, legal)), act, rew, torch.ones(n, dtype=torch.int8, device="cuda"))
    orig = a.experience.sample_indices_dev
    def wrap(b, orig=orig, i=i):
        r = orig(b)
        seen[i].append(r)
        return r
    a.experience.sample_indices_dev = wrap
for step in range(3):
    for a in agents:
        a.update()
torch.cuda.synchronize()
print("calls", len(seen[0]), len(seen[1]))
print("graph agent idx tensors (last = the captured one, holds the last replay's draw):", [t[:6].tolist() for t in seen[0]])
print("eager agent idx:", [t[:6].tolist() for t in seen[1]])
fv = agents[0]._fv
print("fv?", fv is not None, "graph1", agents[0]._graph1 is not None)
w = [torch.cat([p.detach().reshape(-1) for p in a.online.parameters()]) for a in agents]
print("max w diff", float((w[0]-w[1]).abs().max()))
