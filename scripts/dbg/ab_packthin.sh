cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
B="python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-nstep-variant --no-fp16-variant"
for rep in 1 2 3; do for v in 0 1; do
  HB_PACK_THIN=$v $B > $O/pt_$v.json 2>> $O/err.log
  python3 -c "
import json;d=json.load(open('$O/pt_$v.json'));print('pack_thin $v', round(d['ms_per_step'],5), round(d['grad_steps_per_sec']), 'lag', round(d['async_actor']['ms_per_step'],5), 'update alone ms', round(d['roofline_qnet']['learner_update']['ms'],4), flush=True)"
done; done
