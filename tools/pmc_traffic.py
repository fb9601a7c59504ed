#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [commit] [command]

Units and corrections (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots"): both counters are in
KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so wide coalesced reads are
under-counted by exactly 2x -> doubled here.  WRITE_SIZE is uncalibrated in the guide and is
reported as read.  Counters sit on the fabric side of L2, so Infinity-Cache hits are included.
"""
import collections
import csv
import json
import sys

FAMILIES = (("gemm_w4a8", "gemm"), ("gemm_ws", "gemm"), ("splitk_reduce", "splitk_reduce"), ("hadamard_kernel", "hadamard"),
            ("quantize_act", "act_quant"), ("act_quant", "act_quant"))


def family(name):
    for pat, fam in FAMILIES:
        if pat in name:
            return fam
    return None


def collect(path, counter, by_grid=False):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        fam = family(r["Kernel_Name"])
        if fam and by_grid:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mq::", "")
            fam = f'{name} x{int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)}'
        if fam:
            out[fam].append(float(r["Counter_Value"]))
    return out


def csrc_digest():
    import glob, hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "mquant_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "mquant_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --no-graph",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 counts 128 B requests as 64 B); WRITE_SIZE as read",
           "commit": sys.argv[4] if len(sys.argv) > 4 else None,
           # digest of the kernel sources the measured binary was built from (bench.py recomputes it: a later edit of any
           # file under mquant_amd/csrc/ makes the line say traffic_stale: true)
           "csrc_sha16": csrc_digest(),
           "command": sys.argv[5] if len(sys.argv) > 5 else None,
           "kernels": {}}
    for fam in sorted(set(fetch) | set(write)):
        f, w = fetch.get(fam, []), write.get(fam, [])
        rd = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
        wr = 1024.0 * sum(w) / max(len(w), 1)
        res["kernels"][fam] = {"launches_sampled": len(f), "read_bytes_per_launch": round(rd),
                               "write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
    # the same per (kernel instantiation, workgroup count): one line per layer shape
    fetch, write = collect(sys.argv[1], "FETCH_SIZE", True), collect(sys.argv[2], "WRITE_SIZE", True)
    res["by_launch_shape"] = {}
    for key in sorted(set(fetch) | set(write)):
        f, w = fetch.get(key, []), write.get(key, [])
        res["by_launch_shape"][key] = {"launches_sampled": len(f),
                                       "read_bytes_per_launch": round(2.0 * 1024.0 * sum(f) / max(len(f), 1)),
                                       "write_bytes_per_launch": round(1024.0 * sum(w) / max(len(w), 1))}
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
