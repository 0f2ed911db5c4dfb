# NOTE: HB_INSERT_ON_LEARNER was a temporary switch in hanabi_hip/selfplay.py (the replay insert on the learner stream inside the
# hb_chain_run command list); measured, not kept, the switch is no longer in the tree: results in profiles/r03/ab_update_tail.txt.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
run() {
  label=$1; shift
  env "$@" > $O/ins_$label.json 2>> $O/err.log
  python3 - "$label" $O/ins_$label.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 5), "grad/s", round(d["grad_steps_per_sec"]), "host", round(d["host_enqueue_ms_per_step"], 4), "native", d.get("host_calls", {}).get("steps_through_hb_chain_run"), flush=True)
PY
}
B="python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant"
run sync0 HB_INSERT_ON_LEARNER=0 $B
run sync1 HB_INSERT_ON_LEARNER=1 $B
run sync0 HB_INSERT_ON_LEARNER=0 $B
run sync1 HB_INSERT_ON_LEARNER=1 $B
run lag0 HB_INSERT_ON_LEARNER=0 $B --actor-lag 1
run lag1 HB_INSERT_ON_LEARNER=1 $B --actor-lag 1
run lag0 HB_INSERT_ON_LEARNER=0 $B --actor-lag 1
run lag1 HB_INSERT_ON_LEARNER=1 $B --actor-lag 1
run p5_0 HB_INSERT_ON_LEARNER=0 $B --players 5 --steps 100
run p5_1 HB_INSERT_ON_LEARNER=1 $B --players 5 --steps 100
