"""Where does the fused actor kernel spend its cycles? Runs the -DHB_STAMPS diagnostic build (make -C hanabi-agents_amd/csrc
stamps) and prints per-phase cycles per wavefront (median / max over wavefronts). Read the SHARES, not the absolute time of
this build (cdna_hip_programming.md section 7). Usage: actor_fused_stamps.py [rows] [players]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HANABI_HIP_LIB"] = os.path.join(ROOT, "hanabi-agents_amd", "csrc", "diag", "libhanabi_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
from hanabi_hip import _capi as K, ops  # noqa: E402
from hanabi_agents.rlax_dqn import bitpack  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
players = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L, A = {2: (658, 20), 3: (783, 30), 4: (908, 38), 5: (1280, 48)}[players]
H, NA, dev = 512, 51, "cuda"
Kp, Np = (L + 63) // 64 * 64, (A * NA + 63) // 64 * 64
torch.manual_seed(0)
bits = bitpack.pack((torch.rand(N, L, device=dev) < 0.35).to(torch.int8))
w1 = torch.zeros(Kp, H, device=dev, dtype=torch.bfloat16); w1[:L] = (torch.randn(L, H, device=dev) * 0.04).bfloat16()
b1 = (torch.randn(H, device=dev) * 0.05).bfloat16()
w2 = torch.zeros(H, Np, device=dev, dtype=torch.bfloat16); w2[:, :A * NA] = (torch.randn(H, A * NA, device=dev) * 0.2).bfloat16()
b2 = torch.zeros(Np, device=dev, dtype=torch.bfloat16); b2[:A * NA] = (torch.randn(A * NA, device=dev) * 0.5).bfloat16()
support = torch.linspace(-25, 25, NA, device=dev)
act = ops.ActorMFMA(L, H, A, NA, Kp, dev)
act.fused_min_rows = 0
act.pack(w1, b1, w2, b2)
lib = K.lib()
lib.hb_actor_fused_q_stamped.restype = C.c_int
lib.hb_actor_fused_q_stamped.argtypes = [C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32] * 3 + [C.c_void_p] * 3
nw = (N + 127) // 128 * 8
stamps = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
q = torch.empty(N, A, device=dev)
f = act._fset_ptrs[0]
for _ in range(200):   # (clock settles under load)
    act.q_values(bits, support)
names = ["prologue (table, bits)", "barrier", "layer 1 loop", "H write, early rows", "barrier + late rows", "barrier", "pass 0 loop", "pass 0 epilogue",
         "barrier + merge", "pass 1 loop", "pass 1 epilogue", "barrier + merge"]
rows = []
for _ in range(10):
    K.check(lib.hb_actor_fused_q_stamped(bits.data_ptr(), N, L, f[0], f[1], f[2], f[3], support.data_ptr(), H, A, NA, q.data_ptr(),
                                         stamps.data_ptr(), K.current_stream()))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nw, 16).astype(np.int64)
    d = np.diff(s[:, :13], axis=1)
    tot = s[:, 12] - s[:, 0]
    real = s[:, 15] - s[:, 14]
    rows.append((np.median(d, 0), d.max(0), np.median(tot), tot.max(), (s[:, 15].max() - s[:, 14].min()) * 10, np.median(tot / np.maximum(real, 1)) * 100))
med = np.median([r[0] for r in rows], 0)
mx = np.median([r[1] for r in rows], 0)
print(f"N={N} P={players}: per-wave cycles median {np.median([r[2] for r in rows]):.0f} max {np.median([r[3] for r in rows]):.0f}; "
      f"kernel span {np.median([r[4] for r in rows]):.0f} ns; clock ~{np.median([r[5] for r in rows]):.0f} MHz")
tot = med.sum()
for nm, m, x in zip(names, med, mx):
    print(f"  {nm:24s} median {m:8.0f} ({100 * m / tot:4.1f} %)  max {x:8.0f}")
mf1 = (Kp // 32) * 32 * 16
print(f"MFMA floor per wavefront (16 cycles each, partner wave on the same SIMD doubles it): layer 1 {mf1}, layer 2 per pass {16 * 32 * 16}")
