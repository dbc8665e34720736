# rocprofv3 kernel statistics of bench.py for the given configs (no cpu_baseline leg):
#   bash tools/prof_lines.sh <tag> cfg2 cfg3 ...   ->  gpurun_out/<tag>/kernel_stats_bench_<cfg>.csv + the bench line
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
for cfg in "$@"; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python3 $R/bench.py --config $cfg --no-cpu-baseline $BENCH_EXTRA > $O/bench_line_prof_$cfg.json 2> $O/prof_$cfg.log
  f=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_bench_$cfg.csv
  rm -rf $O/prof_$cfg
  echo "$cfg done" >> $O/progress.txt
done
ls $O
