#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel of one HIP source, from hipcc's own report
(-Rpass-analysis=kernel-resource-usage). Cross-compiles, needs no GPU.

    python tools/kernel_resources.py kernels_conv.hip [-DSR3_EXPERIMENTS]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d-super-resolution-face-reconstruction_amd", "csrc")


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "kernels_conv.hip"
    extra = sys.argv[2:] + (["-fno-slp-vectorize"] if src == "kernels_edge.hip" else [])
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
           os.path.join(CSRC, src), "-o", "/tmp/_kres.o", "-Rpass-analysis=kernel-resource-usage", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]

    def field(b, key):
        m = re.search(re.escape(key) + r": (\d+)", b)
        return m.group(1) if m else "?"

    for b in blocks:
        name = b.split("\n")[0].split()[0].strip()
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("sr3::(anonymous namespace)::", "").replace("(sr3::ConvParams)", "")
        print("%-72s vgpr %4s agpr %3s sgpr %3s spill %3s scratch %4s occ %2s lds %s" % (
            dn[:72], field(b, "VGPRs"), field(b, "AGPRs"), field(b, "SGPRs"), field(b, "VGPR Spill"),
            field(b, "ScratchSize [bytes/lane]"), field(b, "Occupancy [waves/SIMD]"), field(b, "LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
