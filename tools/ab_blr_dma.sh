# A/B of blr_pass_dma_kernel (BSC_BLR_DMA=1: the tile by LDS-DMA) against blr_pass_mfma_kernel: the pass's tests on the
# DMA kernel, then alternating default bench runs.   bash tools/ab_blr_dma.sh  ->  gpurun_out/ab_blr_dma.txt
cd $GRAFT_REPO_ROOT
BSC_BLR_DMA=1 timeout -k 10 400 python -m pytest tests/test_blr_gpu.py tests/test_plugin_route_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/ab_blr_dma.txt
for i in 0 1 2; do
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_blr_a$i.json
  BSC_BLR_DMA=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_blr_b$i.json
done
python - >> gpurun_out/ab_blr_dma.txt <<PY
import json
for i in range(3):
    for tag, name in (("a", "registers"), ("b", "LDS-DMA  ")):
        d = json.loads(open("gpurun_out/ab_blr_%s%d.json" % (tag, i)).read()); r = d["roofline"]
        print(name, "%.1f updates/s  %.1f us per update  pass %.1f us = %.3f of the HBM peak; same batch %.1f us" % (
            d["value"], d["ms_per_step"] * 1e3, r["avg_launch_us"], r["frac"], r["same_batch"]["avg_launch_us"]))
PY
cat gpurun_out/ab_blr_dma.txt
