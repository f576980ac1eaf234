#!/usr/bin/env python3
"""Per-round summary of pls_amd/csrc/resources.txt (`make -C pls_amd/csrc resources`): how many kernels, how many with scratch,
and every STREAMING kernel (fused / retile / deflate / xb / xty / syrk / colmoments / zscale) with scratch > 0 -- VGPRs, scratch
bytes per lane, occupancy.  usage: python tools/resource_summary.py > profiles/rN/kernel_resources_summary.txt"""
import os, re, subprocess
txt = open(os.path.join(os.path.dirname(__file__), "..", "pls_amd", "csrc", "resources.txt")).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
rows = []
for b in blocks:
    g = lambda k: int((re.search(k + r": (\d+)", b) or [0, "0"])[1])
    rows.append([b.split("\n")[0].strip(), g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")])
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, d in zip(rows, dem): r[0] = d.split("(")[0].replace("void plsk::", "")
stream = re.compile(r"fused_pass|retile|deflate|xb_|xty|syrk|colmoments|zscale")
print("%d kernels in libpls_hip.so; %d with scratch > 0, %d of them streaming kernels (listed below)" % (
    len(rows), sum(r[2] > 0 for r in rows), sum(r[2] > 0 and bool(stream.search(r[0])) for r in rows)))
print("headline kernels:")
for r in rows:
    if re.search(r"fused_pass_kernel<double, 2, 32, 512, 16, (true|false), 2, 2, false, 0, (true|false)(, (true|false))?>|syrk_glds8_kernel<double>|deflate_piece_kernel<double|retile_xty_kernel<double, 2, 32, 512, 8, 1,", r[0]):
        print("  VGPR %3d  scratch %4d B  occupancy %d  LDS %6d  %s" % (r[1], r[2], r[3], r[4], r[0]))
print("streaming kernels with scratch:")
for r in sorted(rows, key=lambda r: -r[2]):
    if r[2] > 0 and stream.search(r[0]):
        print("  VGPR %3d  scratch %4d B  occupancy %d  %s" % (r[1], r[2], r[3], r[0]))
