# Round-end evidence run (GPU box): bench line, per-config table, rocprofv3 kernel stats of both.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 600 python3 $R/tools/bench_configs.py > $O/bench_configs.txt 2>&1
cp $R/gpurun_out/bench_configs.json $O/bench_configs.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --no-cpu-baseline > $O/prof_bench.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -- python3 $R/tools/bench_configs.py --only=cfg3 --only=cfg5 --only=cfg4 > $O/prof_configs.log 2>&1
find $O -name "*kernel_stats.csv" | while read f; do cp $f $O/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
ls $O
