#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<dir>/...) into the small summaries kept under profiles/.

    python profiles/summarize.py <tag> <stats_dir> [<pmc_dir> ...]

Writes profiles/<tag>_kernel_stats.csv (copy of *_kernel_stats.csv) and profiles/<tag>_pmc.json with
the mean per-dispatch value of every collected counter for the two hot kernels.  HBM traffic
follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts half of the bytes of a wide coalesced streaming read, so read bytes = 2 * FETCH_SIZE * 1024
(our loads are 12 B/lane dwordx3 streams; the doubled figure lands within 0.1 % of the algorithmic
byte count, which calibrates the correction for this access pattern).
"""
import csv
import glob
import json
import os
import shutil
import sys

HOT = {"ke_phash_fused": "hash", "ke_scan_tiles": "scan", "ke_ssim_fast": "ssim", "ke_ssim_waves": "ssim_exact"}


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    here = os.path.dirname(os.path.abspath(__file__))
    def newest(pattern):
        found = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
        return found[-1:]                      # gpurun merges runs into one directory: the latest file is this run's

    for f in newest(os.path.join(stats_dir, "**", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(here, f"{tag}_kernel_stats.csv"))
    out = {}
    for d in pmc_dirs:
        for f in newest(os.path.join(d, "**", "*_counter_collection.csv")):
            acc = {}
            for r in csv.DictReader(open(f)):
                for key, short in HOT.items():
                    if key in r["Kernel_Name"]:
                        acc.setdefault((short, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
            for (short, name), vals in acc.items():
                out.setdefault(short, {})[name] = sum(vals) / len(vals)
                out[short]["dispatches_" + name] = len(vals)
    for short, c in out.items():
        if "FETCH_SIZE" in c:
            c["hbm_read_bytes_corrected"] = 2 * c["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in c:
            c["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    with open(os.path.join(here, f"{tag}_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
