"""hb_c51_loss_sparse / hb_c51_backward alone: time per launch for the whole kernel and for its two halves."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import torch
from hanabi_hip import _capi as K

B, H, A, KK, Np = 256, 512, 20, 51, 1024
g = torch.Generator(device="cuda").manual_seed(0)
dl = torch.zeros(B, 64, device="cuda"); dl[:, :KK] = torch.randn(B, KK, device="cuda", generator=g) * 1e-3
act = torch.randint(0, A, (B,), device="cuda", generator=g, dtype=torch.int32)
hcat = torch.relu(torch.randn(2 * B, 2 * H, device="cuda", generator=g)).to(torch.bfloat16)
w2 = (torch.randn(H, Np, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
dh = torch.zeros(B, H, dtype=torch.bfloat16, device="cuda"); db1 = torch.zeros(H, device="cuda")
dw2 = torch.zeros(H, Np, dtype=torch.bfloat16, device="cuda"); db2 = torch.zeros(Np, device="cuda")
L, s = K.lib(), K.current_stream()
def run():
    K.check(L.hb_c51_backward(K.dptr(dl), K.dptr(act), K.dptr(hcat), 2 * H, K.dptr(w2), Np, 1, B, H, A, KK, K.dptr(dh), K.dptr(db1),
                              K.dptr(dw2), Np, K.dptr(db2), s))
for part in ("", "1", "2", "skew", "skew2"):
    if part.startswith("skew"):
        act[:220] = 5                     # most of the batch took one action (what a greedy policy produces)
        os.environ.pop("HB_BWD_PART", None)
        if part == "skew2": os.environ["HB_BWD_PART"] = "2"
    elif part: os.environ["HB_BWD_PART"] = part
    for _ in range(10): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(200): run()
    b.record(); torch.cuda.synchronize()
    print(f"hb_c51_backward part '{part or 'all'}': {a.elapsed_time(b) / 200 * 1e3:.1f} us per launch (back to back)")

os.environ.pop("HB_BWD_PART", None)
logits_on = (torch.randn(2 * B, Np, device="cuda", generator=g)).to(torch.bfloat16)
logits_t = (torch.randn(2 * B, Np, device="cuda", generator=g)).to(torch.bfloat16)[B:]
rew = torch.zeros(B, device="cuda"); term = torch.zeros(B, device="cuda"); disc = torch.full((B,), 0.99, device="cuda")
prios = (torch.rand(B, device="cuda", generator=g, dtype=torch.float64) + 0.1) / B
beta = torch.tensor(0.4, device="cuda"); support = torch.linspace(-25, 25, KK, device="cuda")
td = torch.zeros(B, device="cuda"); w = torch.zeros(B, device="cuda"); step = torch.zeros((), device="cuda")
bias = torch.zeros(Np, dtype=torch.bfloat16, device="cuda")
dlog = torch.zeros(B, Np, dtype=torch.bfloat16, device="cuda")
def sparse():
    K.check(L.hb_c51_loss_sparse(K.dptr(logits_on), K.dptr(logits_t), 1, K.dptr(act), K.dptr(rew), K.dptr(term), K.dptr(prios), K.dptr(beta),
                                 K.dptr(disc), 1, K.dptr(support), B, A, KK, Np, K.dptr(td), K.dptr(w), K.dptr(dl), K.dptr(step),
                                 K.dptr(bias), K.dptr(bias), s))
def dense():
    K.check(L.hb_c51_loss_grad(K.dptr(logits_on), K.dptr(logits_t), 1, K.dptr(act), K.dptr(rew), K.dptr(term), K.dptr(prios), K.dptr(beta),
                               K.dptr(disc), 1, K.dptr(support), B, A, KK, Np, K.dptr(td), K.dptr(w), K.dptr(dlog), K.dptr(step),
                               K.dptr(bias), K.dptr(bias), s))
for name, fn in (("hb_c51_loss_sparse", sparse), ("hb_c51_loss_grad (dense)", dense)):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(200): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 200 * 1e3:.1f} us per launch (back to back)")
