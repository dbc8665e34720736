# Separate rocprofv3 --pmc passes over one command, summarised per dispatch of one kernel (GPU box):
#   bash tools/pmc_kernel.sh <kernel-name-substring> <out-subdir> <python script> [args...]
# e.g. bash tools/pmc_kernel.sh weighted_outer_kernel pmc_wo tools/run_wouter.py 4
set -e
NEEDLE=$1; SUB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$SUB
rm -rf $O && mkdir -p $O
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/"$@" > $O/log$i.txt 2>&1 || echo "pass $i failed"
done
find $O -name "*_counter_collection.csv" | while read f; do cp $f $O/$(basename $(dirname $(dirname $f)))_$(basename $f); done
python3 $R/profiles/summarize_pmc.py $O $NEEDLE
