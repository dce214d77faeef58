#!/usr/bin/env python3
"""Condense the passes of profiles/tools/pmc_groups.sh: per kernel and counter the median over the dispatches, plus the median
duration from the kernel trace of the same pass.  usage: pmc_summarize.py <gpurun_out/pmc_tag> <out.json> [note]"""
import csv
import glob
import json
import os
import re
import statistics
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    return re.sub(r"\(.*$", "", name).replace("nf::", "")


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = {}
    logs = []
    for g in sorted(glob.glob(os.path.join(root, "g*")), key=lambda p: int(re.sub(r"\D", "", os.path.basename(p).split(".")[0]) or 0)):
        if g.endswith(".log"):
            with open(g) as f:
                lines = [l.strip() for l in f if l.strip()]
            logs.append(lines[-1][:300] if lines else "")
            continue
        for path in glob.glob(os.path.join(g, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    if not k.startswith("k_schur"):
                        continue
                    res.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for path in glob.glob(os.path.join(g, "**", "*kernel_trace.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    if k.startswith("k_schur"):
                        res.setdefault(k, {}).setdefault("_duration_ns_under_pmc", []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    kernels = {k: {c: statistics.median(v) for c, v in sorted(cs.items())} | {"_dispatches": len(next(iter(cs.values())))} for k, cs in res.items()}
    with open(out, "w") as f:
        json.dump(dict(what="rocprofv3 --pmc, one counter group per pass with --kernel-trace only (profiles/tools/pmc_groups.sh over pmc_apply.py); "
                            "medians over the dispatches; _sum counters are summed over the instances; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE "
                            "under-reports coalesced streaming reads by 2x on gfx950 (MI355X_MICROARCH.md)",
                       note=sys.argv[3] if len(sys.argv) > 3 else "", run_lines=logs, kernels=kernels), f, indent=1)
    print(json.dumps({k: {c: v for c, v in cs.items() if c.startswith(("TCP_UTCL1_TR", "TCP_UTCL1_REQ", "_dur", "FETCH", "WRITE"))} for k, cs in kernels.items()}, indent=1))


if __name__ == "__main__":
    main()
