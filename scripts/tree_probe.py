import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import hanabi_hip
os.environ["HB_TREE_UPDATE_PATH"] = sys.argv[1] if len(sys.argv) > 1 else "chunks"   # "single": the one-workgroup path
t = hanabi_hip.SumTree(2**19)
print("small-update path:", os.environ["HB_TREE_UPDATE_PATH"])
mx = torch.tensor([0.6], device="cuda"); mn = mx.clone()
t.fill_range_dev(0, 300000, mx)
def tm(fn, reps=50):
    for _ in range(5): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
for n in (32, 128, 256, 300, 1024):
    idx = torch.randint(0, 300000, (n,), device="cuda")
    td = torch.randn(n, device="cuda")
    print(n, "per_update", f"{tm(lambda: t.per_update_dev(idx, td, 0.6, mx, mn)):.1f} us",
          "update", f"{tm(lambda: t.update_dev(idx, td.abs())):.1f} us")
u = torch.rand(256, dtype=torch.float64, device="cuda")/256
print("per_sample", f"{tm(lambda: t.per_sample_dev(u)):.1f} us")
print("fill_range 32768", f"{tm(lambda: t.fill_range_dev(1000, 32768, mx)):.1f} us")
