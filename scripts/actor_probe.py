import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import hanabi_hip
from hanabi_hip import ops
from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
n = 32768
env = hanabi_hip.HanabiEnv(n_games=n, seed=1)
agent = DQNAgent(ObservationSpec((n, 658)), ActionSpec(20), RlaxRainbowParams(compute_dtype="bfloat16", experience_buffer_size=65536), device="cuda")
def tm(fn, reps=30):
    for _ in range(5): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
eff = agent._effective_weights()
(w1,b1),(w2,b2) = eff
x = torch.zeros(n, w1.shape[0], dtype=torch.bfloat16, device="cuda")
print("cast", tm(lambda: ops.obs_cast(env.obs, torch.bfloat16, out=x)))
print("gemm1+relu", tm(lambda: torch._addmm_activation(b1, x, w1, use_gelu=False)), w1.shape)
h = torch._addmm_activation(b1, x, w1, use_gelu=False)
print("gemm1 addmm", tm(lambda: torch.addmm(b1, x, w1)))
print("gemm1 mm", tm(lambda: torch.mm(x, w1)))
w1t = w1.t().contiguous()
print("gemm1 linear(W^T)", tm(lambda: torch.nn.functional.linear(x, w1t, b1)))
print("gemm2", tm(lambda: torch.addmm(b2, h, w2)), w2.shape)
w2t = w2.t().contiguous()
print("gemm2 linear(W^T)", tm(lambda: torch.nn.functional.linear(h, w2t, b2)))
lg = torch.addmm(b2, h, w2)
print("policy", tm(lambda: ops.policy_act(lg, env.legal, agent.atoms[0].contiguous(), 0.1, 1, 1)))
print("explore total", tm(lambda: agent.explore((None,(env.obs, env.legal)))))
xr = torch.randn_like(x)
print("gemm1 random x", tm(lambda: torch._addmm_activation(b1, xr, w1, use_gelu=False)))
