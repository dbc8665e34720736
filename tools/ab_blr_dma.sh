# A/B of the config-2 pass kernels by whole bench runs: blr_pass_mfma_kernel (tile through registers), blr_pass_dma_kernel
# (round 3: the tile by LDS-DMA) and blr_pass_q_kernel (round 4: both contractions on v_mfma_f32_4x4x1, the default).
# The library reads no environment; the Python Context honours BSC_<OPTION> in a process that says BSC_PROFILING_BUILDS=1.
#   bash tools/ab_blr_dma.sh  ->  gpurun_out/ab_blr_dma.txt       (tools/ab_q.py is the in-process, per-kernel A/B)
cd $GRAFT_REPO_ROOT
export BSC_PROFILING_BUILDS=1
: > gpurun_out/ab_blr_dma.txt
for i in 0 1 2; do
  BSC_BLR_Q=0 BSC_BLR_DMA=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_blr_a$i.json
  BSC_BLR_Q=0 BSC_BLR_DMA=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_blr_b$i.json
  BSC_BLR_Q=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_blr_c$i.json
done
python - >> gpurun_out/ab_blr_dma.txt <<PY
import json
for i in range(3):
    for tag, name in (("a", "registers (blr_q=0 blr_dma=0)"), ("b", "LDS-DMA   (blr_q=0 blr_dma=1)"), ("c", "4x4x1     (blr_q=1)          ")):
        d = json.loads(open("gpurun_out/ab_blr_%s%d.json" % (tag, i)).read()); r = d["roofline"]
        print(name, r["kernel"], "%.1f updates/s  %.1f us per update  pass %.1f us = %.3f of the HBM peak; same batch %.1f us" % (
            d["value"], d["ms_per_step"] * 1e3, r["avg_launch_us"], r["frac"], r["same_batch"]["avg_launch_us"]))
PY
cat gpurun_out/ab_blr_dma.txt
