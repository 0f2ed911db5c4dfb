cd /tmp; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=2; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace_sync -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > $O/bench.json 2>> $O/err.log
python3 scripts/timeline.py $O/trace_sync 100 250 > $O/timeline_sync.txt
python3 scripts/timeline_gantt.py $O/trace_sync 180 4 > $O/gantt_sync.txt
rm -rf $O/trace_sync
