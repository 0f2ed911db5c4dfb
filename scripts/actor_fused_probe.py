"""hb_actor_fused_q (csrc/actor_fused.hip) vs the two-kernel form (hb_actor_hidden_packed + hb_actor_q) and vs an fp32 torch
forward from the same bf16 weights: numerics and time on one MI355X. Usage: actor_fused_probe.py [rows] [players]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
from hanabi_hip import _capi as K, ops
from hanabi_agents.rlax_dqn import bitpack

torch.manual_seed(0)
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
players = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L, A = {2: (658, 20), 3: (783, 30), 4: (908, 38), 5: (1280, 48)}.get(players, (171, 11))
H, NA = 512, 51
Kp, Np = (L + 63) // 64 * 64, (A * NA + 63) // 64 * 64
obs = (torch.rand(N, L, device=dev) < 0.35).to(torch.int8)
bits = bitpack.pack(obs)
w1 = torch.zeros(Kp, H, device=dev, dtype=torch.bfloat16); w1[:L] = (torch.randn(L, H, device=dev) * 0.04).bfloat16()
b1 = (torch.randn(H, device=dev) * 0.05).bfloat16()
w2 = torch.zeros(H, Np, device=dev, dtype=torch.bfloat16); w2[:, :A * NA] = (torch.randn(H, A * NA, device=dev) * 0.2).bfloat16()
b2 = torch.zeros(Np, device=dev, dtype=torch.bfloat16); b2[:A * NA] = (torch.randn(A * NA, device=dev) * 0.5).bfloat16()
support = torch.linspace(-25, 25, NA, device=dev)

os.environ["HB_ACTOR_FUSED"] = "1"
act = ops.ActorMFMA(L, H, A, NA, Kp, dev)
act.fused_min_rows = 0
assert act.fused, "shape not covered"
act.pack(w1, b1, w2, b2)
q_f = act.q_values(bits, support).clone()
act.fused = False
act._q_call = None
q_2 = act.q_values(bits, support).clone()
torch.cuda.synchronize()
# fp32 reference with the same bf16 weights and the hidden activations rounded to bf16 (what both kernels feed layer 2)
h = torch.relu(obs.float() @ w1[:L].float() + b1.float()).bfloat16().float()
lg = (h @ w2[:, :A * NA].float() + b2[:A * NA].float()).view(N, A, NA)
q_ref = (torch.softmax(lg, -1) * support).sum(-1) / NA
print(f"shape N={N} L={L} A={A}: |q| max {q_ref.abs().max().item():.3f}")
print("fused  vs fp32 logits: max |diff|", (q_f - q_ref).abs().max().item())
print("2-kern vs fp32 logits: max |diff|", (q_2 - q_ref).abs().max().item())
top2 = q_ref.topk(2, -1).values
gap = (top2[:, 0] - top2[:, 1])
print("argmax agreement fused / 2-kernel:", (q_f.argmax(-1) == q_ref.argmax(-1)).float().mean().item(), (q_2.argmax(-1) == q_ref.argmax(-1)).float().mean().item(),
      " median top-2 gap", gap.median().item())
# ragged row count
n2 = N - 77
act.fused = True; act._q_call = None; act.h = None
q_r = act.q_values(bits[:n2].contiguous(), support).clone()
print("ragged rows equal:", torch.equal(q_r, q_f[:n2]))
act._q_call = None; act.h = None

def t1(f, reps=50):
    for _ in range(5): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for rep in range(3):
    act.fused = True; act._q_call = None
    tf = t1(lambda: act.q_values(bits, support))
    act.fused = False; act._q_call = None
    t2 = t1(lambda: act.q_values(bits, support))
    print(f"round {rep}: fused {tf:.1f} us   two kernels {t2:.1f} us")
flop = 2.0 * N * (Kp * H + H * 512 * ((A + 9) // 10))
print(f"executed {flop / 1e9:.1f} GFLOP -> fused {flop / tf / 1e9:.3f} PFLOP/s")
tp = t1(lambda: act.pack(w1, b1, w2, b2))
print(f"pack (both forms) {tp:.1f} us")
