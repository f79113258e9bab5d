#!/usr/bin/env python3
"""Compile one .hip translation unit for gfx950 and print VGPR / scratch / occupancy / LDS per kernel."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "godotoceanwaves_amd/csrc/ow_frame.hip"
extra = sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-c", src,
                    "-o", "/tmp/_res.o", "-Rpass-analysis=kernel-resource-usage", "-I", "godotoceanwaves_amd/csrc"] + extra,
                   capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:])
    sys.exit(1)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:"):
        cur = txt.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in txt:
        k, v = txt.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, d in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{dem:46s} VGPR {d.get('VGPRs'):>4s} AGPR {d.get('AGPRs'):>3s} scratch {d.get('ScratchSize [bytes/lane]'):>4s} "
          f"occ {d.get('Occupancy [waves/SIMD]')} LDS {d.get('LDS Size [bytes/block]')}")
