# rocprofv3 kernel statistics of one tool script:  bash tools/prof_tool.sh <tag> tools/<script>.py [args]  ->  gpurun_out/<tag>/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
script=$1; shift
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/$script "$@" > $O/stdout.txt 2> $O/stderr.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
rm -rf $O/prof
tail -5 $O/stdout.txt
