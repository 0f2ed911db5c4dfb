"""Dispatch-by-dispatch listing of a steady-state window of a rocprofv3 --kernel-trace run (start offset, duration, queue, kernel):
what overlaps what in the self-play loop. Usage: timeline_gantt.py <dir> [first env launch] [number of env launches]"""
import csv, glob, os, re, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
t = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows)
envs = [x for x in t if "env_kernel" in x[3]]
a = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lo, hi = envs[a][1], envs[a + n][1]
qs = sorted({x[2] for x in t if lo <= x[0] < hi})
print(f"window: {n} env steps = {(hi - lo) / 1e3:.1f} us; queues {qs}")
for s, e, q, name in t:
    if lo <= s < hi:
        k = re.sub(r"\(anonymous namespace\)::", "", name)
        k = re.sub(r"^void ", "", k).split("(")[0][:44]
        col = qs.index(q)
        print(f"{(s - lo) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  " + "    " * col + f"q{q} {k}")
