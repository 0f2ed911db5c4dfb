cd /tmp; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=2; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03p; mkdir -p $O
make -C hanabi-agents_amd/csrc stamps > /dev/null 2>&1
python3 scripts/actor_fused_stamps.py 32768 2 > $O/actor_fused_stamps_2p.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace_sync -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > /dev/null 2>> $O/err.log
python3 scripts/timeline.py $O/trace_sync 100 250 > $O/timeline_sync.txt
python3 scripts/timeline_gantt.py $O/trace_sync 180 4 > $O/gantt_sync.txt
rm -rf $O/trace_sync
rocprofv3 --kernel-trace --output-format csv -d $O/trace_async -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --actor-lag 1 --no-nstep-variant --no-fp16-variant > /dev/null 2>> $O/err.log
python3 scripts/timeline.py $O/trace_async 100 250 > $O/timeline_async.txt
rm -rf $O/trace_async
HB_DIST_BACKEND=gloo HB_BENCH_DEVICE=0 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gloo_2ranks_one_gpu.json 2> $O/bench_gloo2.err; echo gloo rc=$?
tail -c 600 $O/bench_gloo_2ranks_one_gpu.json; tail -3 $O/bench_gloo2.err
