"""AoS-through-LDS vs SoA game-state transport (csrc/diag/layout_probe.hip) on one MI355X: same data, same result, time per
launch. Build first: make -C hanabi-agents_amd/csrc probe.   python scripts/layout_probe.py [n_games] > profiles/rNN/...json"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "hanabi-agents_amd", "csrc", "diag", "liblayout_probe.so"))
lib.probe_layout.restype = C.c_int
lib.probe_layout.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = 400
g = torch.Generator(device="cuda").manual_seed(1)
rows0 = torch.randint(0, 2 ** 31 - 1, (n, 32), device="cuda", dtype=torch.int32, generator=g)
actions = torch.randint(0, 20, (n,), device="cuda", dtype=torch.int32, generator=g)
s = torch.cuda.current_stream().cuda_stream
out = {"n_games": n, "bytes_per_launch": n * 256, "reps": reps, "results": []}
want = None
for gpw in (16, 32, 64):
    for soa in (0, 1):
        st = (rows0.t().contiguous() if soa else rows0.clone())
        assert lib.probe_layout(soa, gpw, st.data_ptr(), actions.data_ptr(), n, s) == 0
        torch.cuda.synchronize()
        got = st.t().contiguous() if soa else st
        if want is None:
            want = got.clone()
        assert torch.equal(got, want), (soa, gpw)
        for _ in range(20):
            lib.probe_layout(soa, gpw, st.data_ptr(), actions.data_ptr(), n, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(reps):
                lib.probe_layout(soa, gpw, st.data_ptr(), actions.data_ptr(), n, s)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        out["results"].append({"layout": "soa" if soa else "aos_lds", "games_per_wave": gpw, "us_per_launch": round(best, 3),
                               "GBps": round(n * 256 / best / 1e3, 1)})
        print(out["results"][-1], file=sys.stderr)
print(json.dumps(out, indent=1))
