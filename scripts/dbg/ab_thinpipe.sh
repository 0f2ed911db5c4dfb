cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1; do
HB_THIN_PIPE=$v python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('thin_pipe $v', round(d['ms_per_step'],5), round(d['grad_steps_per_sec']), 'update alone', round(d['roofline_qnet']['learner_update']['ms'],4), flush=True)"
done; done
