#!/usr/bin/env python3
"""Register / spill table of every kernel of one HIP source: `python scripts/kres.py conv3x3_dma.hip [-DMACRO ...]`.
Compiles for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per kernel."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "srcgan_amd", "csrc")


def main():
    src = sys.argv[1]
    if not os.path.exists(src):
        src = os.path.join(CSRC, src)
    extra = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: \s*(.*?) \[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
        print(f"{name:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('TotalSGPRs','?'):>4s} "
              f"vspill {r.get('VGPRs Spill','?'):>3s} sspill {r.get('SGPRs Spill','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?')}")


if __name__ == "__main__":
    main()
