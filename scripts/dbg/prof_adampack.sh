cd /tmp; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=2; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
for v in 0 1; do
  export HB_ADAM_PACK=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ap$v -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-async-variant --no-nstep-variant --no-fp16-variant > /dev/null 2>> $O/err.log
  echo "== HB_ADAM_PACK=$v"
  grep -E "adam|pack_weights|fused_pack" $(ls $O/prof_ap$v/*/*kernel_stats.csv | head -1) | cut -d, -f1-4 | cut -c1-160
  rm -rf $O/prof_ap$v
done
