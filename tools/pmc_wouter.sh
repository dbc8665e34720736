set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_wo && mkdir -p $R/gpurun_out/pmc_wo
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wo/p$i -- python3 $R/tools/run_wouter.py 4 > $R/gpurun_out/pmc_wo/log$i.txt 2>&1 || echo "pass $i failed"
done
find $R/gpurun_out/pmc_wo -name "*_counter_collection.csv" | while read f; do cp $f $R/gpurun_out/pmc_wo/$(basename $(dirname $(dirname $f)))_$(basename $f); done
python3 $R/profiles/summarize_pmc.py $R/gpurun_out/pmc_wo weighted_outer_kernel
