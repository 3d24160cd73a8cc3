#!/usr/bin/env python3
"""Compare two `make asm` resource-usage reports (csrc/build/resource_usage.txt): kernels whose registers / occupancy changed."""
import re
import subprocess
import sys


def parse(path):
    d, cur = {}, None
    for ln in open(path, errors="ignore"):
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            d[cur] = {}
        m = re.search(r"(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", ln)
        if m and cur:
            d[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    return d


def dem(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()[:130]
    except OSError:
        return n


old, new = parse(sys.argv[1]), parse(sys.argv[2])
old = {dem(k).split("(")[0]: v for k, v in old.items()}
for k in sorted(new):
    name = dem(k).split("(")[0]
    o = old.get(name) or old.get(name.replace(", false>", ">"))  # a template parameter added with a default
    if o != new[k]:
        print(name, "\n   old", o, "\n   new", new[k])
