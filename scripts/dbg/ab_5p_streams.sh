cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for q in 2 8; do for m in 2 3 5; do
GPU_MAX_HW_QUEUES=$q HB_MAX_LEARNER_STREAMS=$m python3 bench.py --players 5 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant --steps 100 --warmup 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('q=$q streams=$m  %.4f ms/step  %.1f M  grad/s %.0f' % (d['ms_per_step'], d['value']/1e6, d['grad_steps_per_sec']), flush=True)"
done; done
