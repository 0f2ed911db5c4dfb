# NOTE: HB_DBG_SKIP was a temporary, timing-only switch in rlax_dqn/fused_learner.py (tail kernels left out of the captured update:
# results are wrong by construction); it is no longer in the tree: results in profiles/r03/ab_update_tail.txt.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
run() {
  label=$1; shift
  env "$@" python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > $O/skip_$label.json 2>> $O/err.log
  python3 - "$label" $O/skip_$label.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 5), "host", round(d["host_enqueue_ms_per_step"], 4), "update alone ms", round(d["roofline_qnet"]["learner_update"]["ms"], 4), flush=True)
PY
}
run base HB_DBG_SKIP=
run noadam HB_DBG_SKIP=a
run nogemm HB_DBG_SKIP=g
run noadam_nogemm HB_DBG_SKIP=ag
run nopack HB_DBG_SKIP=p
run none HB_DBG_SKIP=agp
run base HB_DBG_SKIP=
