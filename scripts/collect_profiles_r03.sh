#!/bin/bash
# Round-3 evidence, collected on the GPU box into gpurun_out/r03p (copy what is to be judged into profiles/r03).
# Part 1 (default), part 2 (argument "2") or the short re-collection of the headline files (argument "3"): separate calls keep
# each under gpurun's time limit.
cd /tmp
export TMPDIR=/tmp
# exported BEFORE rocprofv3 starts: the profiler's preload initialises HIP before bench.py could set it
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r03p
mkdir -p $O
set -x
if [ "$1" == "3" ]; then
python3 bench.py > $O/full_bench.json 2> $O/full_bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_settings.json 2>> $O/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_full -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > $O/full_bench_under_rocprof.json 2>> $O/err.log
cp $(ls $O/prof_full/*/*kernel_stats.csv | head -1) $O/full_kernel_stats.csv
python3 scripts/timeline.py $O/prof_full 100 250 > $O/timeline_sync.txt
rm -rf $O/prof_full
python3 bench.py --env-only --steps 300 --warmup 50 --no-cpu-baseline > $O/env_only_bench.json 2>> $O/err.log
python3 bench.py --env-only --unpacked-obs --steps 300 --warmup 50 --no-cpu-baseline > $O/env_only_int8_bench.json 2>> $O/err.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_packed_$c -- python3 bench.py --env-only --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/err.log
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_packed5_$c -- python3 bench.py --env-only --players 5 --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/err.log
  HB_ENV_DECK_ALWAYS=1 rocprofv3 --pmc $c --output-format csv -d $O/pmc_always_$c -- python3 bench.py --env-only --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/err.log
done
python3 scripts/pmc_traffic.py $O/pmc_packed_FETCH_SIZE $O/pmc_packed_WRITE_SIZE env_kernel $((32768*369)) "packed observation rows (hb_env_step_packed, plain actions in: the form the loop runs since round 3): 84 B obs + 20 legal + 9 + 2 x 128 state per game; the deck-pool row is fetched only by games that can end with the move" > $O/env_kernel_pmc_traffic_packed.json
python3 scripts/pmc_traffic.py $O/pmc_packed5_FETCH_SIZE $O/pmc_packed5_WRITE_SIZE env_kernel $((32768*601)) "5 players, packed observation rows: 160 B obs + 48 legal + 9 + 2 x 192 state per game" > $O/env_kernel_pmc_traffic_packed_5p.json
python3 scripts/pmc_traffic.py $O/pmc_always_FETCH_SIZE $O/pmc_always_WRITE_SIZE env_kernel $((32768*369)) "HB_ENV_DECK_ALWAYS=1: rounds 1-2's unconditional fetch of the 64-byte deck-pool row (A/B of the same binary)" > $O/env_kernel_pmc_traffic_packed_deck_always.json
rm -rf $O/pmc_packed_* $O/pmc_packed5_* $O/pmc_always_*
python3 bench.py --players 5 --steps 100 --warmup 30 --no-cpu-baseline > $O/bench_5p.json 2>> $O/err.log
( for q in 2 3 4 8 16; do for coll in 0 1; do GPU_MAX_HW_QUEUES=$q HB_BENCH_FORCE_COLLECTIVE=$coll python3 bench.py --no-cpu-baseline --no-nstep-variant --no-fp16-variant --steps 200 --warmup 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('GPU_MAX_HW_QUEUES=$q collective_path=$coll sync %.4f ms/step  async %.4f ms/step' % (d['ms_per_step'], d.get('async_actor',{}).get('ms_per_step', float('nan'))))"; done; done ) > $O/hw_queues.txt 2>/dev/null
echo part3 done
elif [ "$1" != "2" ]; then
python3 bench.py > $O/full_bench.json 2> $O/full_bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_settings.json 2>> $O/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_full -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > $O/full_bench_under_rocprof.json 2>> $O/err.log
cp $(ls $O/prof_full/*/*kernel_stats.csv | head -1) $O/full_kernel_stats.csv
python3 scripts/timeline.py $O/prof_full 100 250 > $O/timeline_sync.txt
python3 scripts/timeline_gantt.py $O/prof_full 180 4 > $O/gantt_sync.txt
rm -rf $O/prof_full
# env kernel: packed step (the kernel form the loop runs), HBM traffic by PMC in separate passes
python3 bench.py --env-only --steps 300 --warmup 50 --no-cpu-baseline > $O/env_only_bench.json 2>> $O/err.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_packed_$c -- python3 bench.py --env-only --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/err.log
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_packed5_$c -- python3 bench.py --env-only --players 5 --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/err.log
done
python3 scripts/pmc_traffic.py $O/pmc_packed_FETCH_SIZE $O/pmc_packed_WRITE_SIZE env_kernel $((32768*369)) "packed observation rows (hb_env_step_packed, plain actions in: the form the loop runs since round 3): 84 B obs + 20 legal + 9 + 2 x 128 state per game" > $O/env_kernel_pmc_traffic_packed.json
python3 scripts/pmc_traffic.py $O/pmc_packed5_FETCH_SIZE $O/pmc_packed5_WRITE_SIZE env_kernel $((32768*601)) "5 players, packed observation rows: 160 B obs + 48 legal + 9 + 2 x 192 state per game" > $O/env_kernel_pmc_traffic_packed_5p.json
rm -rf $O/pmc_packed_* $O/pmc_packed5_*
# the one-kernel actor: alone (probe, stamps) and its counters inside the loop
python3 scripts/actor_fused_probe.py 32768 2 > $O/actor_fused_probe_2p.log 2>&1
python3 scripts/actor_fused_probe.py 32768 5 > $O/actor_fused_probe_5p.log 2>&1
python3 scripts/actor_fused_stamps.py 32768 2 > $O/actor_fused_stamps_2p.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_loop -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > /dev/null 2>> $O/err.log
python3 scripts/pmc_summary.py $O/pmc_loop actor_fused_kernel env_kernel thin_gemm_kernel > $O/loop_kernels_pmc.json 2>> $O/err.log
rm -rf $O/pmc_loop
echo part1 done
else
python3 bench.py --players 5 --steps 100 --warmup 30 --no-cpu-baseline > $O/bench_5p.json 2>> $O/err.log
python3 bench.py --vanilla --steps 200 --warmup 40 --no-cpu-baseline > $O/bench_vanilla.json 2>> $O/err.log
python3 bench.py --n-step 3 --no-cpu-baseline --no-async-variant > $O/bench_nstep3.json 2>> $O/err.log
python3 bench.py --games 262144 --steps 60 --warmup 10 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > $O/bench_262144_full_loop.json 2>> $O/err.log
python3 bench.py --env-only --games 262144 --steps 100 --warmup 20 --no-cpu-baseline > $O/env_only_262144_bench.json 2>> $O/err.log
rocprofv3 --kernel-trace --output-format csv -d $O/trace_async -- python3 bench.py --steps 300 --warmup 40 --no-cpu-baseline --actor-lag 1 --no-nstep-variant --no-fp16-variant > /dev/null 2>> $O/err.log
python3 scripts/timeline.py $O/trace_async 100 250 > $O/timeline_async.txt
rm -rf $O/trace_async
( for q in 2 3 4 8 16; do for coll in 0 1; do GPU_MAX_HW_QUEUES=$q HB_BENCH_FORCE_COLLECTIVE=$coll python3 bench.py --no-cpu-baseline --no-nstep-variant --no-fp16-variant --steps 200 --warmup 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('GPU_MAX_HW_QUEUES=$q collective_path=$coll sync %.4f ms/step  async %.4f ms/step' % (d['ms_per_step'], d.get('async_actor',{}).get('ms_per_step', float('nan'))))"; done; done ) > $O/hw_queues.txt 2>&1
python3 scripts/learner_probe.py 2 > $O/learner_probe.log 2>&1
cp gpurun_out/dtype_parity.json $O/dtype_parity.json 2>/dev/null
echo part2 done
fi
