"""GEMM-core variants of csrc/diag/gemm_probe.hip on one MI355X: numerics vs torch, time, in-kernel stamp shares.
Build first: make -C hanabi-agents_amd/csrc probe"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "hanabi-agents_amd", "csrc", "diag", "libgemm_probe.so"))
lib.probe_gemm.restype = C.c_int
lib.probe_gemm.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,3".split(","))]
shapes = [(32768, 704, 512), (32768, 512, 1024)]
torch.manual_seed(0)
for (M, Kd, N) in shapes:
    x = (torch.rand(M, Kd, device="cuda") < 0.3).to(torch.bfloat16) if Kd == 704 else torch.relu(torch.randn(M, Kd, device="cuda")).to(torch.bfloat16)
    wt = (torch.randn(N, Kd, device="cuda") * 0.05).to(torch.bfloat16)
    ref = (x[:4096].float() @ wt.float().t())
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    nblk = ((M + 255) // 256) * (N // 256)
    stamps = torch.zeros(nblk * 8 * 64, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    flop = 2.0 * M * Kd * N
    # variant 6 takes the activations as bits (the hidden layer's real input): only meaningful for the 0/1 operand
    xb = (x != 0).to(torch.int64).view(M, Kd // 32, 32)
    xbits = (xb << torch.arange(32, device="cuda")).sum(-1)
    xbits = torch.where(xbits >= 2 ** 31, xbits - 2 ** 32, xbits).to(torch.int32).contiguous()
    for v in variants:
        if v == 6 and Kd != 704:
            continue
        out.zero_()
        xin = xbits if v == 6 else x
        rc = lib.probe_gemm(v, 0, xin.data_ptr(), M, Kd, wt.data_ptr(), N, out.data_ptr(), None, s)
        torch.cuda.synchronize()
        assert rc == 0, rc
        err = (out[:4096].float() - ref).abs().max().item()
        scale = ref.abs().max().item()
        for _ in range(5):
            lib.probe_gemm(v, 0, xin.data_ptr(), M, Kd, wt.data_ptr(), N, out.data_ptr(), None, s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            lib.probe_gemm(v, 0, xin.data_ptr(), M, Kd, wt.data_ptr(), N, out.data_ptr(), None, s)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 30 * 1e3
        print(f"shape {M}x{Kd}x{N} variant {v}: {us:7.1f} us  {flop / us / 1e6:7.1f} TF/s  max|err| {err:.4f} (scale {scale:.2f})", flush=True)
        # stamps
        stamps.zero_()
        lib.probe_gemm(v, 1, xin.data_ptr(), M, Kd, wt.data_ptr(), N, out.data_ptr(), stamps.data_ptr(), s)
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().reshape(nblk * 8, 64).astype(np.int64)
        t0 = st[:, 0]
        tot = st[:, 63] - t0
        used = [j for j in range(64) if (st[:, j] != 0).all()]
        rel = np.median(st[:, used] - t0[:, None], axis=0)
        print(f"   stamped: wave lifetime median {np.median(tot):.0f} cycles, max {tot.max():.0f}; slots {used[:3]}..{used[-3:]}")
        print("   median cycles since start at each stamp:", " ".join(f"{int(r)}" for r in rel))
        d = np.diff(rel)
        print("   deltas:", " ".join(f"{int(r)}" for r in d))
