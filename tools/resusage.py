#!/usr/bin/env python3
"""Compact table of per-kernel register/LDS/scratch use from hipcc -Rpass-analysis=kernel-resource-usage output (stdin)."""
import re, subprocess, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: \s*(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split()[0]] = v
def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        return n
print(f"{'kernel':160s} {'vgpr':>5s} {'sgpr':>5s} {'scr':>5s} {'occ':>4s} {'lds':>6s}")
for r in rows:
    print(f"{demangle(r['name'])[:160]:160s} {r.get('VGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('ScratchSize','?'):>5s} {r.get('Occupancy','?'):>4s} {r.get('LDS','?'):>6s}")
