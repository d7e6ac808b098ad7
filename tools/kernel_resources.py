#!/usr/bin/env python3
"""Compile rt_device.hip with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "accelerated-ray-tracer_amd", "csrc", "rt_device.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-c", src, "-o", "/tmp/_rt_device_res.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    n = r["name"]
    m = re.match(r"_Z\d+(rt_render_\w+?_kernel)ILb(\d)ELb(\d)ELb(\d)ELi(\d)E", n)
    short = f"{m.group(1)}<so={m.group(2)},tx={m.group(3)},uv={m.group(4)},lds={m.group(5)}>" if m else n
    print(f"{short:58s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
