#!/bin/bash
# rocprofv3 counter passes for the dominant kernels (separate --pmc runs with --kernel-trace only:
# FETCH_SIZE and WRITE_SIZE cannot share a pass, /opt/skills/guides/MI355X_MICROARCH.md "PMC slots").
#   bash tools/pmc_collect.sh gpurun_out/pmc_r02 cfg5 cfg3 ...
# then: python3 profiles/summarize_pmc.py --all gpurun_out/pmc_r02 > profiles/r02_x_pmc.txt
set -e
out=$1; shift
export TMPDIR=/tmp
G1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for cfg in "$@"; do
  i=0
  for grp in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    d=$out/$cfg/pass$i
    mkdir -p $d
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 tools/pmc_run.py $cfg 12 > $d/log.txt 2>&1
    echo "$cfg pass$i done" >> $out/progress.txt
  done
done
