"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of each kernel.

    python profiles/summarize_pmc.py gpurun_out/pmc_r01 [kernel-substring]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, needle="blr_pass_kernel"):
    out = {}
    for path in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(path)):
            if needle not in r["Kernel_Name"]:
                continue
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        for k, (tot, n) in acc.items():
            out[k] = (tot / n, n)
    for k, (v, n) in sorted(out.items()):
        print("%-24s %18.1f  (mean over %d dispatches)" % (k, v, n))
    return out


if __name__ == "__main__":
    main(*sys.argv[1:])
