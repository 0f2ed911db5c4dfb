cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
B="python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant"
for rep in 1 2 3; do for dt in bfloat16 float16; do
  $B --compute-dtype $dt > $O/dt_$dt.json 2>> $O/err.log
  python3 -c "
import json;d=json.load(open('$O/dt_$dt.json'));print('$dt', round(d['ms_per_step'],5), round(d['grad_steps_per_sec']), round(d['roofline_qnet']['actor_forward']['per_kernel']['hb_actor_fused_act']['avg_launch_us'],2), round(d['roofline_qnet']['learner_update']['ms'],4), flush=True)"
done; done
HB_DTYPE=float16 timeout -k 10 120 python3 scripts/train_small.py Hanabi-Full 60000 4096 4 > $O/learning_curve_full2p_fp16.log 2>&1; tail -3 $O/learning_curve_full2p_fp16.log
HB_DTYPE=bfloat16 timeout -k 10 120 python3 scripts/train_small.py Hanabi-Full 60000 4096 4 > $O/learning_curve_full2p_bf16.log 2>&1; tail -3 $O/learning_curve_full2p_bf16.log
