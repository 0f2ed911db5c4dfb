"""HBM traffic of one kernel from two separate rocprofv3 passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE), corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE counts 64-B requests of 128-B reads (x2 for wide coalesced reads), both are in KiB.
usage: pmc_traffic.py <fetch_dir> <write_dir> <kernel substring> <algorithmic bytes per launch> [note]"""
import csv, glob, json, os, sys

def collect(d, counter, sub):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r.get("Kernel_Name", "") and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    vals = vals[len(vals) // 5:]   # skip the warm-up launches
    return {"counter": counter, "dispatches": len(vals), "mean_KiB": sum(vals) / max(1, len(vals)), "min_KiB": min(vals), "max_KiB": max(vals)}

fd, wd, sub, alg = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
f, w = collect(fd, "FETCH_SIZE", sub), collect(wd, "WRITE_SIZE", sub)
fx2, wb = f["mean_KiB"] * 1024 * 2, w["mean_KiB"] * 1024
print(json.dumps({"kernel": sub, "fetch": f, "write": w,
                  "per_launch_bytes": {"fetch_corrected_x2": fx2, "write": wb, "total": fx2 + wb, "algorithmic": alg,
                                       "note": sys.argv[5] if len(sys.argv) > 5 else ""}}, indent=1))
