#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_usage.py recommender_amd/csrc/tower.hip [name filter] [extra hipcc flags ...]"""
import re, subprocess, sys, os
src = os.path.abspath(sys.argv[1]); filt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                    "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"] + extra, capture_output=True, text=True, cwd=os.path.dirname(src) or ".")
t = r.stderr
names = re.findall(r"Function Name: (\S+)", t)
dem = dict(zip(names, subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines())) if names else {}
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    d = dem.get(m.group(1), m.group(1))
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    if filt in d:
        print("%-100s VGPR %3s AGPR %3s scratch %4s occ %s" % (d[:100], m.group(2), m.group(3), m.group(4), m.group(5)))
if r.returncode:
    print(t[-3000:])
