#!/usr/bin/env python3
"""Register / LDS / occupancy figures of the gfx950 kernels, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/kernel_resources.py [source.hip ...] [--filter substring]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "gem_amd" / "csrc"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")]
    extra = [a[len("--flag="):] for a in sys.argv[1:] if a.startswith("--flag=")]
    srcs = [Path(a) for a in args] or [CSRC / "gem_sort.hip", CSRC / "gem_kernels.hip"]
    for src in srcs:
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", str(src), "-o", f"{td}/o.o",
                                "-Rpass-analysis=kernel-resource-usage", *extra], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-3000:]); sys.exit(1)
        for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
            name = b.split("\n")[0]
            if flt and not any(f in name for f in flt):
                continue

            def g(k):
                m = re.search(k + r": (\d+)", b)
                return int(m.group(1)) if m else -1
            scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
            print(f"{name[:84]:84s} VGPR {g('VGPRs'):3d} AGPR {g('AGPRs'):3d} spill {g('VGPR Spill'):3d} scratch {scratch:4d} occ {occ} LDS {lds}")


if __name__ == "__main__":
    main()
