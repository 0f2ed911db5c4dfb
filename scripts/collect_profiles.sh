#!/bin/bash
# Round-2 evidence, collected on the GPU box into gpurun_out/r02 (copy what is to be judged into profiles/r02).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02
mkdir -p $O
set -x
python3 bench.py > $O/full_bench.json 2> $O/full_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_full -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline > $O/full_bench_under_rocprof.json 2>> $O/err.log
python3 bench.py --env-only --steps 300 --warmup 50 --no-cpu-baseline > $O/env_only_bench.json 2>> $O/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_env -- python3 bench.py --env-only --steps 300 --warmup 50 --no-cpu-baseline > $O/env_only_bench_under_rocprof.json 2>> $O/err.log
python3 bench.py --env-only --unpacked-obs --steps 300 --warmup 50 --no-cpu-baseline > $O/env_only_int8_bench.json 2>> $O/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_env_int8 -- python3 bench.py --env-only --unpacked-obs --steps 300 --warmup 50 --no-cpu-baseline > /dev/null 2>> $O/err.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_packed_$c -- python3 bench.py --env-only --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/err.log
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_int8_$c -- python3 bench.py --env-only --unpacked-obs --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/err.log
done
python3 scripts/pmc_traffic.py $O/pmc_packed_FETCH_SIZE $O/pmc_packed_WRITE_SIZE env_kernel $((32768*369)) "packed observation rows (hb_env_step_packed): 84 B obs + 20 legal + 9 + 2 x 128 state per game; reads also include the 64-B deck-pool row" > $O/env_kernel_pmc_traffic_packed.json
python3 scripts/pmc_traffic.py $O/pmc_int8_FETCH_SIZE $O/pmc_int8_WRITE_SIZE env_kernel $((32768*943)) "int8 observations (hb_env_step): SURVEY 8(d)'s 943 B per game" > $O/env_kernel_pmc_traffic.json
# 262 144 games (BASELINE config 5's total on one GPU): the kernel at 8x the occupancy
python3 bench.py --env-only --games 262144 --steps 100 --warmup 20 --no-cpu-baseline > $O/env_only_262144_bench.json 2>> $O/err.log
python3 bench.py --env-only --games 262144 --unpacked-obs --steps 100 --warmup 20 --no-cpu-baseline > $O/env_only_262144_int8_bench.json 2>> $O/err.log
# 5 players, vanilla
python3 bench.py --players 5 --steps 100 --warmup 30 --no-cpu-baseline > $O/bench_5p.json 2>> $O/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_5p -- python3 bench.py --players 5 --steps 100 --warmup 30 --no-cpu-baseline > /dev/null 2>> $O/err.log
python3 bench.py --vanilla --steps 200 --warmup 40 > $O/bench_vanilla.json 2>> $O/err.log
# learner alone, GEMM core probe, co-residency probe
python3 scripts/learner_probe.py 2 > $O/learner_probe.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner -- python3 scripts/learner_probe.py 2 > /dev/null 2>> $O/err.log
python3 scripts/gemm_probe.py 0,1,3 > $O/gemm_probe.log 2>&1
python3 scripts/coresidency_probe.py > $O/coresidency_probe.log 2>&1
python3 scripts/bwd_probe.py > $O/bwd_probe.log 2>&1
python3 scripts/gemm_probe.py 0,7,8 > $O/gemm_probe_wave128.log 2>&1
# kernel timelines of the synchronous loop and of the asynchronous-actor loop (per queue: busy share, in-loop kernel durations)
rocprofv3 --kernel-trace --output-format csv -d $O/trace_sync -- python3 bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-async-variant > /dev/null 2>> $O/err.log
rocprofv3 --kernel-trace --output-format csv -d $O/trace_async -- python3 bench.py --steps 300 --warmup 40 --no-cpu-baseline --actor-lag 1 > /dev/null 2>> $O/err.log
python3 scripts/timeline.py $O/trace_sync > $O/timeline_sync.txt
python3 scripts/timeline.py $O/trace_async > $O/timeline_async.txt
rm -rf $O/trace_sync $O/trace_async
# the driver's own invocation (20 timed steps)
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_settings.json 2>> $O/err.log
echo done
