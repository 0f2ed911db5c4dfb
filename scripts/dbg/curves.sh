cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
run() { label=$1; shift; env "$@" > $O/curve_$label.log 2>&1; echo "== $label"; grep step $O/curve_$label.log | awk 'NR%8==0' | cut -c1-80; tail -1 $O/curve_$label.log | cut -c1-80; }
run benched_sync HB_DTYPE=bfloat16 timeout -k 10 200 python3 scripts/train_small.py Hanabi-Full 40000 32768 1
run benched_lag HB_DTYPE=bfloat16 HB_ACTOR_LAG=1 timeout -k 10 200 python3 scripts/train_small.py Hanabi-Full 40000 32768 1
run benched_fp16 HB_DTYPE=float16 timeout -k 10 200 python3 scripts/train_small.py Hanabi-Full 40000 32768 1
run lag_4096 HB_DTYPE=bfloat16 HB_ACTOR_LAG=1 timeout -k 10 200 python3 scripts/train_small.py Hanabi-Full 60000 4096 4
