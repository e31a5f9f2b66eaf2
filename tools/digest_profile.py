#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of tools/profile_bench.sh from gpurun_out/prof_<tag>/ into profiles/ (tracked).
usage: python tools/digest_profile.py <tag> [<name under profiles/>]"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """the most recent match (gpurun merges runs of the same tag into one directory)"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


shutil.copy(newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0], os.path.join(dst, f"{name}_kernel_stats.csv"))
rows = []
for d in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_insts"):
    fs = newest(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(fs[0])):
        if "search_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            # (rocprofv3's VGPR_Count / Accum_VGPR_Count / SGPR_Count columns are dispatch-packet granules, not what the compiler
            # allocated -- e.g. 216 / 0 for a kernel that holds 256 VGPRs + 171 AGPRs: left out; the compiler's own figures are in
            # profiles/<round>_resource_usage.txt, written by tools/resource_usage.py)
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size")}
    for k, v in acc.items():
        rows.append(dict(pass_=d, counter=k, launches=len(v), mean_per_launch=sum(v) / len(v), **meta))
with open(os.path.join(dst, f"{name}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
print(open(os.path.join(dst, f"{name}_kernel_stats.csv")).read())
for r in rows:
    print(r["pass_"], r["counter"], r["mean_per_launch"])
