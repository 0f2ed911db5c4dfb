import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "absent")
from oracle import oracle_py as O
flags = O.FLAG_AUTO_RESET | O.FLAG_RESET_START_NEXT
for threads in (1, 2, 4, 8, 16, 32):
    env = O.OracleEnv(O.make_config("Hanabi-Full", 2, flags), 4096, seed=1234, threads=threads)
    legal = env.observe()["legal"]
    t0 = time.perf_counter()
    for t in range(40):
        act = O.random_legal_actions(legal, 4321, t)
        legal = env.step(act)["legal"]
    print(threads, "threads:", 4096*40/(time.perf_counter()-t0)/1e6, "M env-steps/s")
