# Config 5 after the whole-line DMA pieces: bench lines + rocprofv3 kernel statistics of the same commands (both routes),
# counters of the split pass.   bash tools/refresh_cfg5_profiles.sh r03  ->  gpurun_out/r03_cfg5/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
O=$R/gpurun_out/${tag}_cfg5
rm -rf $O && mkdir -p $O
for terms in 0 2; do
  timeout -k 10 300 python3 $R/bench.py --config cfg5 --mfma-split $terms > $O/bench_line_cfg5_split$terms.json 2> $O/bench_line_cfg5_split$terms.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config cfg5 --mfma-split $terms --no-cpu-baseline > $O/bench_line_prof_cfg5_split$terms.json 2> $O/prof.log
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_bench_cfg5_split$terms.csv
  rm -rf $O/prof
  echo "cfg5 $terms done" >> $O/progress.txt
done
timeout -k 10 200 python3 $R/tools/bench_generic_reparam.py > $O/bench_generic_reparam.txt 2>&1 || true
echo "generic done" >> $O/progress.txt
cd $R
timeout -k 10 600 bash tools/pmc_collect.sh $O/pmc cfg5 cfg5x2 > $O/pmc.log 2>&1
python3 profiles/summarize_pmc.py --all $O/pmc > $O/pmc_kernels.txt 2>&1
ls $O
