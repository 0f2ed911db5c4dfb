# A/B of the update-tail co-residency switches inside the benched loop (same box, alternating).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
run() {  # label, env assignments...
  label=$1; shift
  for rep in 1 2; do
    env "$@" python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > $O/ab_$label.json 2>> $O/err.log
    python3 - "$label" $O/ab_$label.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 5), "grad/s", round(d["grad_steps_per_sec"]), "host", round(d["host_enqueue_ms_per_step"], 4),
      "update alone ms", round(d["roofline_qnet"]["learner_update"]["ms"], 4), flush=True)
PY
  done
}
run base HB_ADAM_WIDTH=4 HB_TREE_UPDATE_PATH=fused
run adam2 HB_ADAM_WIDTH=2 HB_TREE_UPDATE_PATH=fused
run tree HB_ADAM_WIDTH=4 HB_TREE_UPDATE_PATH=x
run both HB_ADAM_WIDTH=2 HB_TREE_UPDATE_PATH=x
run base HB_ADAM_WIDTH=4 HB_TREE_UPDATE_PATH=fused
run both HB_ADAM_WIDTH=2 HB_TREE_UPDATE_PATH=x
