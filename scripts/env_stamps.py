"""Where does one env step spend its cycles? Runs the -DHB_STAMPS diagnostic build
(make -C hanabi-agents_amd/csrc stamps) and prints per-phase shares (median over wavefronts).
Read the SHARES, not the absolute time of this build (cdna_hip_programming.md §7)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HANABI_HIP_LIB"] = os.path.join(ROOT, "hanabi-agents_amd", "csrc", "diag", "libhanabi_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import hanabi_hip  # noqa: E402
from hanabi_hip import _capi as K  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
players = int(sys.argv[3]) if len(sys.argv) > 3 else 2
flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", players, flags), n_games=n, seed=1234, games_per_wave=g)
L = K.lib()
L.hb_env_step_stamped.restype = C.c_int
L.hb_env_step_stamped.argtypes = [C.c_void_p] * 6
act = torch.empty(n, dtype=torch.int32, device="cuda")
nw = (n + g - 1) // g
nw = ((nw + 3) // 4) * 4
stamps = torch.zeros(nw * 12, dtype=torch.int64, device="cuda")
for t in range(100):
    env.random_legal_actions(4321, t, out=act)
    env.step(act)
names = ["load", "bar1", "rules", "reset", "bar2", "encode", "bar3", "state_wb", "expand"]
acc = []
for t in range(100, 120):
    env.random_legal_actions(4321, t, out=act)
    K.check(L.hb_env_step_stamped(env.h, K.dptr(act), K.dptr(env.obs), K.dptr(env.legal), K.dptr(stamps), K.current_stream()))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nw, 12).astype(np.int64)
    d = np.diff(s[:, :10], axis=1)
    tot = s[:, 9] - s[:, 0]
    real = (s[:, 11] - s[:, 10])  # 100 MHz ticks
    span_real = s[:, 11].max() - s[:, 10].min()
    acc.append((np.median(d, axis=0), np.max(d, axis=0), np.median(tot), tot.max(), span_real, np.median(tot / np.maximum(real, 1)) * 100))
med = np.median([a[0] for a in acc], axis=0)
mx = np.median([a[1] for a in acc], axis=0)
print(f"G={g} N={n} P={players}: per-wave cycles median {np.median([a[2] for a in acc]):.0f} max {np.median([a[3] for a in acc]):.0f}; "
      f"kernel span {np.median([a[4] for a in acc]) * 10:.0f} ns; clock ~{np.median([a[5] for a in acc]):.0f} MHz")
for nm, m, x in zip(names, med, mx):
    print(f"  {nm:9s} median {m:8.0f}  max {x:8.0f}")
