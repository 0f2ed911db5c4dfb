cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do for q in 2 8; do
GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-nstep-variant --no-fp16-variant --steps 200 --warmup 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('q=$q sync %.4f  async %.4f' % (d['ms_per_step'], d['async_actor']['ms_per_step']), flush=True)"
done; done
