"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of each kernel.

    python profiles/summarize_pmc.py gpurun_out/pmc_r01 [kernel-substring]
    python profiles/summarize_pmc.py --all gpurun_out/pmc_r02          # every cfg*/pass* below it
    python profiles/summarize_pmc.py --traffic gpurun_out/pmc_r02      # -> profiles/pmc_traffic.json

--traffic writes HBM bytes per launch for the dominant kernels: FETCH_SIZE and WRITE_SIZE come
from separate passes, are reported in KB, and on gfx950 FETCH_SIZE counts HALF the bytes of a
wide coalesced streaming read (/opt/skills/guides/MI355X_MICROARCH.md, section HBM), so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Each record carries the sha1 of the .hip file
the counted build came from; bench.py reports `traffic` only while that still matches.
"""
import csv
import glob
import hashlib
import json
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# config directory -> (kernel-name substring, source file, algorithmic bytes per launch)
KERNELS = {
    "cfg2": ("blr_pass_q_kernel", "bsc_blr.hip", 4.0 * 1_000_000 * 256 + 4.0 * 1_000_000),
    "cfg2stream": ("blr_pass_q_kernel", "bsc_blr.hip", 4.0 * 1_000_000 * 256 + 4.0 * 1_000_000),
    "cfg2rot": ("blr_pass_q_kernel", "bsc_blr.hip", 4.0 * 1_000_000 * 256 + 4.0 * 1_000_000),
    "cfg3": ("mog_estep_kernel", "bsc_mog.hip", 4.0 * 10_000_000 * 16),
    "cfg3l2": ("mog_estep_kernel", "bsc_mog.hip", 4.0 * 10_000_000 * 16),
    "cfg3x": ("mog_estep_bx_kernel", "bsc_mog.hip", 4.0 * 10_000_000 * 16),
    "softstats": ("gemm_softmax_stats_kernel", "bsc_rowsoftmax.hip", 4.0 * 10_000_000 * 32),
    "cfg4b": ("lda_sstats_stream_bound_kernel", "bsc_lda.hip", 4.0 * 6250 * 100_000),
    "cfg4x2": ("lda_sstats_bx2_bound_kernel", "bsc_lda.hip", 4.0 * 6250 * 100_000),
    "cfg4x3": ("lda_sstats_bx3_bound_kernel", "bsc_lda.hip", 4.0 * 6250 * 100_000),
    "cfg5x2": ("logreg_loglik_dma_bx_kernel", "bsc_bbvi.hip", 4.0 * 1_000_000 * 256 + 8.0 * 1_000_000),
    "rowsoftmax": ("gemm_softmax_rows_kernel", "bsc_rowsoftmax.hip", 4.0 * 10_000_000 * (40 + 64 + 2)),
    "cfg4": ("lda_sstats_stream_kernel", "bsc_lda.hip", 4.0 * 6250 * 100_000),
    "cfg5": ("logreg_loglik_dma_kernel", "bsc_bbvi.hip", 4.0 * 1_000_000 * 256 + 8.0 * 1_000_000),
    "wouter": ("weighted_outer_kernel", "bsc_wouter.hip", 4.0 * 10_000_000 * (64 + 16)),
    "gram": ("gemm_f32_stream_kernel", "bsc_gemm.hip", 4.0 * 1_000_000 * 256),
    "gramx": ("gram256_bx_kernel", "bsc_gram.hip", 4.0 * 1_000_000 * 256),
    "lda1": ("gemm_f32_stream_kernel", "bsc_gemm.hip", 4.0 * 6250 * 100_000 + 4.0 * 128 * 106_250),
    "lda1e": ("gemm_f32_stream_kernel", "bsc_gemm.hip", 8.0 * 6250 * 100_000 + 4.0 * 128 * 106_250),
    "lda2": ("gemm_f32_stream_kernel", "bsc_gemm.hip", 4.0 * 6250 * 100_000 + 4.0 * 128 * 206_250),
    "sq4096nt": ("gemm_f32_stream_kernel", "bsc_gemm.hip", 12.0 * 4096 * 4096),
    "sq4096tn": ("gemm_f32_stream_kernel", "bsc_gemm.hip", 12.0 * 4096 * 4096),
    "skinny": ("gemm_skinny_nt_kernel", "bsc_skinny.hip", 4.0 * 1_000_000 * 256 + 4.0 * 8 * 1_000_000),
}


def collect(d, needle):
    out = {}
    paths = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))
    for path in paths:
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(path)):
            if needle not in r["Kernel_Name"]:
                continue
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        for k, (tot, n) in acc.items():
            out[k] = (tot / n, n)
    return out


def durations(d, needle):
    """Mean kernel duration (us) from the kernel-trace CSVs of the same passes."""
    tot, n = 0.0, 0
    for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if needle in r["Kernel_Name"]:
                tot += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                n += 1
    return (tot / n, n) if n else (None, 0)


def show(out):
    for k, (v, n) in sorted(out.items()):
        print("%-26s %18.1f  (mean over %d dispatches)" % (k, v, n))


def derived(out, dur_us):
    v = {k: a for k, (a, _) in out.items()}
    if v.get("SQ_INSTS_MFMA"):
        # SQ_INSTS_VALU counts the MFMAs too
        print("  non-MFMA VALU per MFMA     %.2f" % ((v.get("SQ_INSTS_VALU", 0) - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"]))
        if v.get("SQ_INSTS_VMEM"):
            print("  MFMA per VMEM instruction  %.1f" % (v["SQ_INSTS_MFMA"] / v["SQ_INSTS_VMEM"]))
        if v.get("SQ_INSTS_LDS"):
            print("  MFMA per LDS instruction   %.2f" % (v["SQ_INSTS_MFMA"] / v["SQ_INSTS_LDS"]))
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print("  LDS bank-conflict share    %.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]))
    if v.get("SQ_BUSY_CYCLES") and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES sums 32 shader engines; MFMA busy cycles sum over 1024 SIMDs
        print("  MFMA busy / SIMD-cycles    %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * v["SQ_BUSY_CYCLES"])))
    if dur_us and v.get("SQ_BUSY_CYCLES"):
        print("  clock under load           %.2f GHz (SQ_BUSY_CYCLES / 32 / duration; profiled passes)"
              % (v["SQ_BUSY_CYCLES"] / 32.0 / (dur_us * 1e-6) / 1e9))
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        print("  HBM bytes per launch       %.4e  (2*FETCH_SIZE + WRITE_SIZE) KB" %
              ((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))


def main(argv):
    if argv and argv[0] in ("--all", "--traffic"):
        base = argv[1]
        traffic = {}
        for cfg in sorted(os.listdir(base)):
            d = os.path.join(base, cfg)
            if not os.path.isdir(d) or cfg not in KERNELS:
                continue
            needle, src, algo = KERNELS[cfg]
            out = collect(d, needle)
            if not out:
                continue
            dur, n = durations(d, needle)
            if argv[0] == "--all":
                print("-- %s  (%s; profiled mean duration %s us over %d dispatches)"
                      % (needle, cfg, ("%.1f" % dur) if dur else "?", n))
                show(out)
                derived(out, dur)
            if "FETCH_SIZE" in out and "WRITE_SIZE" in out and needle not in traffic:
                # (several configs run the same GEMM kernel: the first in directory order -- `gram` -- is its record)
                path = os.path.join(ROOT, "bayesic_amd", "csrc", src)
                traffic[needle] = {
                    "hbm_bytes_per_launch": (2 * out["FETCH_SIZE"][0] + out["WRITE_SIZE"][0]) * 1024,
                    "fetch_size_kb": out["FETCH_SIZE"][0], "write_size_kb": out["WRITE_SIZE"][0],
                    "algorithmic_bytes_per_launch": algo,
                    "source": src, "source_sha1": hashlib.sha1(open(path, "rb").read()).hexdigest(),
                    "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
                    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                              "passes of `python3 tools/pmc_run.py %s` (tools/pmc_collect.sh), mean over %d "
                              "dispatches; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM; KB x1024"
                              % (cfg, out["FETCH_SIZE"][1]),
                }
        if argv[0] == "--traffic":
            path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            old = json.load(open(path)) if os.path.exists(path) else {}
            old.update(traffic)
            json.dump(old, open(path, "w"), indent=1)
            print("wrote %s: %s" % (path, ", ".join(sorted(traffic))))
        return
    d = argv[0]
    needle = argv[1] if len(argv) > 1 else "blr_pass_kernel"
    show(collect(d, needle))


if __name__ == "__main__":
    main(sys.argv[1:])
