"""Per-queue timeline of a rocprofv3 --kernel-trace run: for every HIP queue, busy time, idle gaps and the average in-loop
duration of each kernel inside a steady-state window (the middle half of the trace). Usage: timeline.py <dir with *_kernel_trace.csv>"""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
t.sort()
envs = [x for x in t if "env_kernel" in x[3]]
# the self-play loop's env launches come first (set-up, warm-up, timed region), stand-alone measurements after them:
# take launches a..b of the env kernel (default 150..300) as the steady-state window
a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (150, 300)
lo, hi = envs[a][0], envs[b][0]
win = [x for x in t if lo <= x[0] < hi]
span = (max(x[1] for x in win) - min(x[0] for x in win)) / 1e3
print(f"{f}: {len(t)} dispatches, window {span:.0f} us with {len(win)} dispatches")
byq = defaultdict(list)
for x in win:
    byq[x[2]].append(x)
nenv = sum(1 for x in win if "env_kernel" in x[3])
print(f"env steps in window: {nenv} -> {span / max(nenv, 1):.1f} us per step")
for q, xs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in xs) / 1e3
    gaps = [(xs[i + 1][0] - xs[i][1]) / 1e3 for i in range(len(xs) - 1)]
    print(f"\nqueue {q}: {len(xs)} dispatches, busy {busy:.0f} us = {busy / span * 100:.0f} % of the window, per step {busy / max(nenv, 1):.1f} us; "
          f"gaps: median {sorted(gaps)[len(gaps) // 2] if gaps else 0:.1f} us")
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, _, n in xs:
        k = re.sub(r"\(anonymous namespace\)::", "", n)
        k = re.sub(r"^void ", "", k).split("(")[0][:60]
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
    for k, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {k:60s} x{c:4d} avg {tot / c:7.1f} us   per step {tot / max(nenv, 1):6.1f} us")
