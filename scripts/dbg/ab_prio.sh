cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do for combo in "0 -1" "-1 0"; do set -- $combo
python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-nstep-variant --no-fp16-variant --main-priority $1 --learner-priority $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('main $1 learner $2:', round(d['ms_per_step'],5), 'async', round(d['async_actor']['ms_per_step'],5), flush=True)"
done; done
