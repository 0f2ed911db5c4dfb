"""Can a register-lean GEMM run WHILE an actor GEMM holds every CU? (what would let a seat's update progress
during the other seat's policy forward.) Stream A: hb_actor_q over 32 768 rows (232 VGPRs x 8 waves and 133 KB of LDS per CU:
48 VGPRs per SIMD and 27 KB are left). Stream B (high priority): the "thin" GEMM of csrc/diag/gemm_probe.hip (one wavefront per
workgroup, 32 x 16 tile, no LDS, 36 VGPRs) on the learner's first-layer shape [512 x 704] x [704 x 1024], and hipBLASLt's
kernel for the same product. Reports each alone and launched beside the running actor GEMM.
Build first: make -C hanabi-agents_amd/csrc probe"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
from hanabi_hip import _capi as K  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "hanabi-agents_amd", "csrc", "diag", "libgemm_probe.so"))
lib.probe_thin_gemm.restype = C.c_int
lib.probe_thin_gemm.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
# the actor GEMM of stream A
N, H, A, KK = 32768, 512, 20, 51
h = torch.relu(torch.randn(N, H, device=dev, generator=g)).to(torch.bfloat16)
w2t = (torch.randn(4 * 256, H, device=dev, generator=g) * 0.05).to(torch.bfloat16)
b2 = torch.zeros(4 * 256, device=dev)
support = torch.linspace(-25, 25, KK, device=dev)
q = torch.empty(N, A, device=dev)
# the learner's first layer
M, Kd, Nn = 512, 704, 1024
x = (torch.rand(M, Kd, device=dev, generator=g) < 0.3).to(torch.bfloat16)
w = (torch.randn(Kd, Nn, device=dev, generator=g) * 0.05).to(torch.bfloat16)
wt = w.t().contiguous()
out = torch.empty(M, Nn, dtype=torch.bfloat16, device=dev)
L = K.lib()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)


def gemm(stream):
    K.check(L.hb_actor_q(K.dptr(h), N, H, K.dptr(w2t), K.dptr(b2), K.dptr(support), A, KK, K.dptr(q), C.c_void_p(stream.cuda_stream)))


def thin(stream):
    assert lib.probe_thin_gemm(x.data_ptr(), M, Kd, wt.data_ptr(), Nn, out.data_ptr(), stream.cuda_stream) == 0


def library(stream):
    with torch.cuda.stream(stream):
        torch.mm(x, w)


thin(sb)
torch.cuda.synchronize()
ref = x.float() @ w.float()
err = (out.float() - ref).abs().max().item()
print(f"thin GEMM max |err| vs fp32: {err:.4f} (scale {ref.abs().max().item():.2f})")
ev = lambda: torch.cuda.Event(enable_timing=True)
med = lambda v: sorted(v)[len(v) // 2]
for name, small in (("thin (36 VGPRs, no LDS)", thin), ("hipBLASLt", library)):
    for _ in range(5):
        gemm(sa); small(sb)
    torch.cuda.synchronize()
    alone, galone, span, after, gbeside = [], [], [], [], []
    for it in range(30):
        e0, e1 = ev(), ev()
        e0.record(sb); small(sb); e1.record(sb); torch.cuda.synchronize()
        alone.append(e0.elapsed_time(e1) * 1e3)
        g0, g1 = ev(), ev()
        g0.record(sa); gemm(sa); g1.record(sa); torch.cuda.synchronize()
        galone.append(g0.elapsed_time(g1) * 1e3)
        g0, g1, e0, e1 = ev(), ev(), ev(), ev()
        g0.record(sa); gemm(sa); g1.record(sa)
        e0.record(sb); small(sb); e1.record(sb)
        torch.cuda.synchronize()
        after.append(g0.elapsed_time(e1) * 1e3); span.append(e0.elapsed_time(e1) * 1e3); gbeside.append(g0.elapsed_time(g1) * 1e3)
    print(f"{name}: alone {med(alone):.1f} us; beside the actor GEMM (alone {med(galone):.1f} us): finished {med(after):.1f} us after the GEMM "
          f"started (own span {med(span):.1f} us), the GEMM then took {med(gbeside):.1f} us")
