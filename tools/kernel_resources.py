#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel in one csrc unit, as hipcc's -Rpass-analysis=kernel-resource-usage reports them
(runs in the build container: no GPU needed).  usage: python tools/kernel_resources.py gemm_pp.hip [more.hip ...] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mquant_amd", "csrc")


def report(unit, extra):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-c", os.path.join(CSRC, unit), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    name, row = None, {}
    for ln in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
        if not m:
            continue
        if m.group(1) == "Function Name":
            if name:
                yield name, row
            name, row = m.group(2), {}
        else:
            row[m.group(1).split(" ")[0]] = m.group(2)
    if name:
        yield name, row


def main():
    args, extra = sys.argv[1:], []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    for unit in args:
        for name, row in report(unit, extra):
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            short = re.sub(r"\(mq::GemmArgs\)|\(mq::HadArgs\)|void mq::", "", short)
            print(f"{short[:110]:110s} VGPR {row.get('VGPRs', '?'):>4} AGPR {row.get('AGPRs', '?'):>3} SGPR {row.get('TotalSGPRs', '?'):>3} "
                  f"scratch {row.get('ScratchSize', '?'):>4} occ {row.get('Occupancy', '?')}")


if __name__ == "__main__":
    main()
