"""Does a small learner kernel run BESIDE a resident actor-GEMM workgroup (registers / LDS permitting), or does it queue
until the GEMM's workgroups retire? Stream A: hb_actor_q (232 VGPRs x 8 waves, 133 KB LDS per CU). Stream B (high priority):
hb_c51_loss_sparse (40 VGPRs, 4 KB LDS) launched while A is running. Reports B's completion time alone and beside A."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import ctypes as C
import torch
from hanabi_hip import _capi as K

dev = "cuda"
N, H, A, KK, Np, B = 32768, 512, 20, 51, 1024, 256
g = torch.Generator(device=dev).manual_seed(0)
h = torch.relu(torch.randn(N, H, device=dev, generator=g)).to(torch.bfloat16)
w2t = (torch.randn(4 * 256, H, device=dev, generator=g) * 0.05).to(torch.bfloat16)
b2 = torch.zeros(4 * 256, device=dev)
support = torch.linspace(-25, 25, KK, device=dev)
q = torch.empty(N, A, device=dev)
logits_on = torch.randn(2 * B, Np, device=dev, generator=g).to(torch.bfloat16)
logits_t = torch.randn(B, Np, device=dev, generator=g).to(torch.bfloat16)
act = torch.randint(0, A, (B,), device=dev, generator=g, dtype=torch.int32)
rew = torch.zeros(B, device=dev); term = torch.zeros(B, device=dev); disc = torch.full((B,), 0.99, device=dev)
prios = (torch.rand(B, device=dev, generator=g, dtype=torch.float64) + 0.1) / B
beta = torch.tensor(0.4, device=dev); td = torch.zeros(B, device=dev); w = torch.zeros(B, device=dev); step = torch.zeros((), device=dev)
dl = torch.zeros(B, 64, device=dev); bias = torch.zeros(Np, dtype=torch.bfloat16, device=dev)
L = K.lib()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)

def gemm(stream):
    K.check(L.hb_actor_q(K.dptr(h), N, H, K.dptr(w2t), K.dptr(b2), K.dptr(support), A, KK, K.dptr(q), C.c_void_p(stream.cuda_stream)))
def small(stream):
    K.check(L.hb_c51_loss_sparse(K.dptr(logits_on), K.dptr(logits_t), 1, K.dptr(act), K.dptr(rew), K.dptr(term), K.dptr(prios), K.dptr(beta),
                                 K.dptr(disc), 1, K.dptr(support), B, A, KK, Np, K.dptr(td), K.dptr(w), K.dptr(dl), K.dptr(step),
                                 K.dptr(bias), K.dptr(bias), C.c_void_p(stream.cuda_stream)))
for _ in range(5):
    gemm(sa); small(sb)
torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
res = {"alone": [], "beside": [], "gemm_alone": [], "gemm_beside": []}
for it in range(30):
    # alone
    e0, e1 = ev(), ev()
    e0.record(sb); small(sb); e1.record(sb); torch.cuda.synchronize()
    res["alone"].append(e0.elapsed_time(e1) * 1e3)
    g0, g1 = ev(), ev()
    g0.record(sa); gemm(sa); g1.record(sa); torch.cuda.synchronize()
    res["gemm_alone"].append(g0.elapsed_time(g1) * 1e3)
    # beside: start the GEMM, then (from the host, ~10 us later) the small kernel on the other stream
    g0, g1, e0, e1 = ev(), ev(), ev(), ev()
    g0.record(sa); gemm(sa); g1.record(sa)
    e0.record(sb); small(sb); e1.record(sb)
    torch.cuda.synchronize()
    res["beside"].append((g0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3))
    res["gemm_beside"].append(g0.elapsed_time(g1) * 1e3)
med = lambda x: sorted(x)[len(x) // 2]
print(f"small kernel alone: {med(res['alone']):.1f} us; GEMM alone: {med(res['gemm_alone']):.1f} us")
print(f"beside: small kernel finished {med([a for a, _ in res['beside']]):.1f} us after the GEMM STARTED (its own span {med([b for _, b in res['beside']]):.1f} us); "
      f"GEMM took {med(res['gemm_beside']):.1f} us")
