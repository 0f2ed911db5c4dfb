"""Hand-written MFMA actor kernels (csrc/actor.hip) vs the library path (hb_obs_cast + hipBLASLt GEMMs + hb_policy_act):
numerics and time, one MI355X."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
from hanabi_hip import _capi as K, ops

torch.manual_seed(0)
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L, H, A, NA = 658, 512, 20, 51
Kp, Np = 704, 1024
obs = (torch.rand(N, L, device=dev) < 0.35).to(torch.int8)
legal = (torch.rand(N, A, device=dev) < 0.6).to(torch.int8); legal[:, 0] = 1
w1 = torch.zeros(Kp, H, device=dev, dtype=torch.bfloat16); w1[:L] = (torch.randn(L, H, device=dev) * 0.04).bfloat16()
b1 = (torch.randn(H, device=dev) * 0.05).bfloat16()
w2 = torch.zeros(H, Np, device=dev, dtype=torch.bfloat16); w2[:, :A * NA] = (torch.randn(H, A * NA, device=dev) * 0.05).bfloat16()
b2 = torch.zeros(Np, device=dev, dtype=torch.bfloat16); b2[:A * NA] = (torch.randn(A * NA, device=dev) * 0.05).bfloat16()
support = torch.linspace(-25, 25, NA, device=dev)
lib, s = K.lib(), K.current_stream()

# packed operands
w1t = torch.zeros(H, Kp, device=dev, dtype=torch.bfloat16); b1p = torch.zeros(H, device=dev)
groups = (A + 4) // 5
w2t = torch.zeros(groups * 256, H, device=dev, dtype=torch.bfloat16); b2p = torch.zeros(groups * 256, device=dev)
jobs = (K.HbPackJob * 2)()
for j, (w, b, wt, bo, kr, nc, ld, grp, kp) in enumerate(((w1, b1, w1t, b1p, L, H, H, 0, Kp), (w2, b2, w2t, b2p, H, A * NA, Np, 255, H))):
    jobs[j].w, jobs[j].bias, jobs[j].wt, jobs[j].bias_out = w.data_ptr(), b.data_ptr(), wt.data_ptr(), bo.data_ptr()
    jobs[j].k_rows, jobs[j].n_cols, jobs[j].w_ld, jobs[j].group_cols, jobs[j].k_pad = kr, nc, ld, grp, kp
K.check(lib.hb_actor_pack_weights(jobs, 2, s))
assert torch.equal(w1t[:, :L], w1[:L].t()) and torch.equal(b1p, b1.float())
n = torch.arange(A * NA, device=dev); npr = n // 255 * 256 + n % 255
assert torch.equal(w2t[npr], w2[:, :A * NA].t()) and torch.equal(b2p[npr], b2[:A * NA].float())

x = torch.zeros(N, Kp, device=dev, dtype=torch.bfloat16)
hbuf = torch.empty(N, H, device=dev, dtype=torch.bfloat16)
q = torch.empty(N, A, device=dev)
act = torch.empty(N, dtype=torch.int32, device=dev)

def lib_path(draw=1):
    xx = ops.obs_cast(obs, torch.bfloat16, out=x)
    h = torch._addmm_activation(b1, xx, w1, use_gelu=False)
    lg = torch.addmm(b2, h, w2)
    return h, lg, ops.policy_act(lg, legal, support, 0.1, 77, draw, 0)

def mfma_path(draw=1):
    K.check(lib.hb_actor_hidden(K.dptr(obs), N, L, K.dptr(w1t), Kp, K.dptr(b1p), H, K.dptr(hbuf), s))
    K.check(lib.hb_actor_q(K.dptr(hbuf), N, H, K.dptr(w2t), K.dptr(b2p), K.dptr(support), A, NA, K.dptr(q), s))
    K.check(lib.hb_policy_select(K.dptr(q), K.dptr(legal), N, A, 0.1, 77, draw, 0, K.dptr(act), s))
    return hbuf, q, act

h_ref, lg, a_ref = lib_path()
h_new, q_new, a_new = mfma_path()
torch.cuda.synchronize()
dh = (h_new.float() - h_ref.float()).abs()
print("hidden: max |diff|", dh.max().item(), "mismatching entries", (dh > 0).float().mean().item())
p = torch.softmax(lg[:, :A * NA].float().view(N, A, NA), -1)
q_ref = (p * support).sum(-1) / NA
print("q: max |diff| vs library logits", (q_new - q_ref).abs().max().item(), "scale", q_ref.abs().max().item())
print("actions equal:", (a_new == a_ref).float().mean().item())
# exact check of the epilogue + select given the kernel's own hidden: recompute logits from h_new in fp32
lg2 = (h_new.float() @ w2.float()[:, :A * NA] + b2.float()[:A * NA]).bfloat16().float().view(N, A, NA)
q2 = (torch.softmax(lg2, -1) * support).sum(-1) / NA
print("q vs fp32 recompute from the same hidden: max |diff|", (q_new - q2).abs().max().item())

def tm(fn, reps=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i + 2)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print(f"library path {tm(lib_path):.1f} us   MFMA path {tm(mfma_path):.1f} us")
def t1(f, reps=30):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("  hb_actor_hidden", f"{t1(lambda: lib.hb_actor_hidden(K.dptr(obs), N, L, K.dptr(w1t), Kp, K.dptr(b1p), H, K.dptr(hbuf), s)):.1f} us",
      " hb_actor_q", f"{t1(lambda: lib.hb_actor_q(K.dptr(hbuf), N, H, K.dptr(w2t), K.dptr(b2p), K.dptr(support), A, NA, K.dptr(q), s)):.1f} us",
      " hb_policy_select", f"{t1(lambda: lib.hb_policy_select(K.dptr(q), K.dptr(legal), N, A, 0.1, 77, 1, 0, K.dptr(act), s)):.1f} us")
print("  lib: cast", f"{t1(lambda: ops.obs_cast(obs, torch.bfloat16, out=x)):.1f}", "gemm1", f"{t1(lambda: torch._addmm_activation(b1, x, w1, use_gelu=False)):.1f}",
      "gemm2", f"{t1(lambda: torch.addmm(b2, h_ref, w2)):.1f}", "policy", f"{t1(lambda: ops.policy_act(lg, legal, support, 0.1, 77, 1, 0)):.1f} us")
