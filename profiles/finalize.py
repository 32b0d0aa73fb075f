#!/usr/bin/env python3
"""After `bash profiles/collect.sh <tag>` on the GPU box (results merged into gpurun_out/): condense them into profiles/.
    python profiles/finalize.py r02"""
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
here = os.path.dirname(os.path.abspath(__file__))
g = os.path.join(os.path.dirname(here), "gpurun_out")
pmc = sorted(glob.glob(os.path.join(g, f"{tag}_pmc_*")))
pmc = [d for d in pmc if os.path.isdir(d)]
subprocess.check_call([sys.executable, os.path.join(here, "summarize.py"), tag, os.path.join(g, f"{tag}_stats")] + pmc, stdout=subprocess.DEVNULL)
newest = lambda pat: sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1]
shutil.copy(newest(os.path.join(g, f"{tag}_stats_ssim", "**", "*_kernel_stats.csv")), os.path.join(here, f"{tag}_ssim095_kernel_stats.csv"))
for name in (f"{tag}_bench.json", f"{tag}_bench_ssim095.json"):
    line = [l for l in open(os.path.join(g, name)) if l.startswith("{")][0]
    open(os.path.join(here, name), "w").write(line)
for name, keep in ((f"{tag}_scan_sizes.jsonl", lambda l: l.startswith("{")), (f"{tag}_hbm_read.txt", lambda l: not l.startswith(("==", "rc=")))):
    open(os.path.join(here, name), "w").writelines(l for l in open(os.path.join(g, name)) if keep(l))
if os.path.exists(os.path.join(g, f"{tag}_decode.jsonl")):
    open(os.path.join(here, f"{tag}_decode.jsonl"), "w").writelines(l for l in open(os.path.join(g, f"{tag}_decode.jsonl")) if l.startswith("{"))
    if os.path.exists(os.path.join(g, f"{tag}_fastsig.jsonl")):
        open(os.path.join(here, f"{tag}_fastsig.jsonl"), "w").writelines(l for l in open(os.path.join(g, f"{tag}_fastsig.jsonl")) if l.startswith("{"))
    for fmt in ("png", "jpeg", "pngdrawing"):
        found = glob.glob(os.path.join(g, f"{tag}_stats_{fmt}", "**", "*_kernel_stats.csv"), recursive=True)
        if found:
            shutil.copy(sorted(found, key=os.path.getmtime)[-1], os.path.join(here, f"{tag}_{fmt}_decode_kernel_stats.csv"))
# round 3 extras (collect.sh <tag> more): copied as they are, JSON lines filtered from whatever else the programs printed
for name in (f"{tag}_bench_dhash.json", f"{tag}_bench_selflaunch.json", f"{tag}_mixed.jsonl", f"{tag}_kernels.jsonl", f"{tag}_scanner.jsonl",
             f"{tag}_decode_phases.jsonl", f"{tag}_fastsig_pillow_route.jsonl", f"{tag}_fastsig_bmp.jsonl", f"{tag}_fastsig_gif.jsonl", f"{tag}_fastsig_tiff.jsonl", f"{tag}_decode_gif.jsonl", f"{tag}_fastsig_collection.jsonl"):
    src = os.path.join(g, name)
    if os.path.exists(src):
        open(os.path.join(here, name), "w").writelines(l for l in open(src) if l.startswith("{"))
for name in (f"{tag}_shape_grid_phash.txt", f"{tag}_shape_grid_both.txt"):
    src = os.path.join(g, name)
    if os.path.exists(src):
        open(os.path.join(here, name), "w").writelines(l for l in open(src) if not l.startswith(("==", "rc=")) and "amdgpu.ids" not in l)
for sub, dst in (("stats_dhash", f"{tag}_dhash_kernel_stats.csv"), ("stats_mixed", f"{tag}_mixed125k_kernel_stats.csv")):
    found = glob.glob(os.path.join(g, f"{tag}_{sub}", "**", "*_kernel_stats.csv"), recursive=True)
    if found:
        shutil.copy(sorted(found, key=os.path.getmtime)[-1], os.path.join(here, dst))
h = json.load(open(os.path.join(here, f"{tag}_pmc.json")))["hash"]
json.dump({"kernel": "ke_phash_fused_mx<8,5,false,false,3,false>", "images_per_launch": 100000, "side": 512,
           "source": f"profiles/{tag}_pmc.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1 "
                     "--ssim-threshold 0.95 (profiles/collect.sh)",
           "FETCH_SIZE_KiB": h["FETCH_SIZE"], "WRITE_SIZE_KiB": h["WRITE_SIZE"],
           "correction": "gfx950: FETCH_SIZE counts half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM): "
                         "read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 as is",
           "hbm_bytes_per_launch": int(round(h["hbm_read_bytes_corrected"] + h["hbm_write_bytes"]))},
          open(os.path.join(here, "hash_kernel_traffic.json"), "w"), indent=1)
print("profiles/ updated for", tag)
