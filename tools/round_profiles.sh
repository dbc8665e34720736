# Round-end evidence (GPU box): one driver-style bench line per config with its cpu_baseline, the rocprofv3 kernel
# statistics of the same command, the plugin-surface line, the derived-engine and general-engine timings.
#   bash tools/round_profiles.sh r03   ->  gpurun_out/r03_final/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
O=$R/gpurun_out/${tag}_final
rm -rf $O && mkdir -p $O
for cfg in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 400 python3 $R/bench.py --config $cfg > $O/bench_line_$cfg.json 2> $O/bench_line_$cfg.err
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python3 $R/bench.py --config $cfg --no-cpu-baseline > $O/bench_line_prof_$cfg.json 2> $O/prof_$cfg.log
  f=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_bench_$cfg.csv
  rm -rf $O/prof_$cfg
  echo "$cfg done" >> $O/progress.txt
done
timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_cfg2_driver_flags.json 2> $O/bench_line_cfg2_driver_flags.err
timeout -k 10 300 python3 $R/bench.py --via plugin --no-cpu-baseline > $O/bench_line_cfg2_via_plugin.json 2> $O/bench_line_cfg2_via_plugin.err
echo "plugin done" >> $O/progress.txt
timeout -k 10 300 python3 $R/tools/bench_derived_mog.py --steps > $O/bench_derived_mog.txt 2>&1
timeout -k 10 300 python3 $R/tools/bench_generic_reparam.py > $O/bench_generic_reparam.txt 2>&1
timeout -k 10 300 python3 $R/tools/bench_configs.py > $O/bench_configs.txt 2>&1 || true
echo "tools done" >> $O/progress.txt
ls $O
