"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: python scripts/pmc_summary.py <dir> [kernel substring ...]"""
import csv, glob, json, os, sys
d = sys.argv[1]
want = sys.argv[2:] or ["actor_gemm_kernel"]
out = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        key = next((w for w in want if w in name), None)
        if key is None:
            continue
        kk = key + ("<0> hidden (int8 rows)" if "ILi0E" in name or "<0>" in name else "<1> q" if "ILi1E" in name or "<1>" in name else
                    "<2> hidden (bit rows)" if "ILi2E" in name or "<2>" in name else "")
        e = out.setdefault(kk, {})
        c = e.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
res = {k: {c: v[0] / v[1] for c, v in e.items()} | {"dispatches": max(v[1] for v in e.values())} for k, e in out.items()}
print(json.dumps(res, indent=1))
