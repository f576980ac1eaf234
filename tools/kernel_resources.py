#!/usr/bin/env python3
"""VGPRs / scratch / occupancy / LDS of the kernels in pls_amd/csrc/resources.txt (`make -C pls_amd/csrc resources`),
filtered by substrings of the demangled name:  python tools/kernel_resources.py fused_pass_kernel 'true>'"""
import os
import re
import subprocess
import sys

txt = open(os.path.join(os.path.dirname(__file__), "..", "pls_amd", "csrc", "resources.txt")).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
rows = []
for b in blocks:
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "0"])[1]
    rows.append((b.split("\n")[0].strip(), g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                 g(r"LDS Size \[bytes/block\]")))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("VGPR scratch occ LDS  kernel")
for r, d in zip(rows, dem):
    d = d.split("(")[0]
    if all(k in d for k in sys.argv[1:]):
        print(f"{r[1]:>4} {r[2]:>7} {r[3]:>3} {r[4]:>6}  {d.replace('void plsk::', '')}")
