# The operand-split bf16 route (bsc_ctx_set_mfma_split): bench lines of the configs that offer it, rocprofv3 kernel
# statistics of the same commands, the kernel-level A/B tools, counters.   bash tools/split_profiles.sh r03  ->  gpurun_out/r03_split/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
O=$R/gpurun_out/${tag}_split
rm -rf $O && mkdir -p $O
for run in "cfg3 2" "cfg4 2" "cfg4 3" "cfg5 2"; do
  set -- $run
  cfg=$1; terms=$2
  timeout -k 10 300 python3 $R/bench.py --config $cfg --mfma-split $terms --no-cpu-baseline > $O/bench_line_${cfg}_split$terms.json 2> $O/bench_line_${cfg}_split$terms.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config $cfg --mfma-split $terms --no-cpu-baseline > $O/bench_line_prof_${cfg}_split$terms.json 2> $O/prof.log
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_bench_${cfg}_split$terms.csv
  rm -rf $O/prof
  echo "$cfg $terms done" >> $O/progress.txt
done
timeout -k 10 200 python3 $R/tools/bench_lda_split.py > $O/bench_lda_split.txt 2>&1
timeout -k 10 200 python3 $R/tools/bench_logreg_split.py > $O/bench_logreg_split.txt 2>&1
timeout -k 10 200 python3 $R/tools/bench_mog_split.py > $O/bench_mog_split.txt 2>&1
timeout -k 10 200 python3 $R/tools/bench_gram_split.py > $O/bench_gram_split.txt 2>&1
timeout -k 10 60 $R/tools/check_bf16_maps > $O/check_bf16_maps.txt 2>&1
echo "tools done" >> $O/progress.txt
cd $R
timeout -k 10 600 bash tools/pmc_collect.sh $O/pmc cfg3x cfg4x2 cfg5x2 gramx > $O/pmc.log 2>&1
python3 profiles/summarize_pmc.py --all $O/pmc > $O/pmc_kernels.txt 2>&1
ls $O
