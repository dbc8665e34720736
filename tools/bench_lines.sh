# One driver-style bench line per config (with its cpu_baseline) and the rocprofv3 kernel statistics of the
# same command without the baseline leg:  bash tools/bench_lines.sh  ->  gpurun_out/lines/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lines
rm -rf $O && mkdir -p $O
for cfg in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 300 python3 $R/bench.py --config $cfg > $O/bench_line_$cfg.json 2> $O/bench_line_$cfg.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python3 $R/bench.py --config $cfg --no-cpu-baseline > $O/prof_$cfg.log 2>&1
  f=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_bench_$cfg.csv
  echo "$cfg done" >> $O/progress.txt
done
ls $O
