"""Condense the rocprofv3 CSVs written by profiles/collect.sh.

usage: python3 profiles/summarize.py <dir with stats/ fetch/ write/> <tag>
writes gpurun_out/prof_<tag>/<tag>_kernel_stats.csv (copy of rocprofv3's kernel_stats) and <tag>_pmc_traffic.json:
per kernel the median FETCH_SIZE / WRITE_SIZE over the dispatches that did work, converted to bytes per cell.
gfx950: FETCH_SIZE under-reports coalesced streaming reads by 2x (MI355X_MICROARCH.md, HBM section) -- calibrated on
k_cg_rupdate, which reads r and q = 16 B/cell; WRITE_SIZE is exact.  Counter units are KiB.
"""
import csv
import glob
import json
import os
import re
import shutil
import statistics
import sys


def find(root, suffix):
    hits = sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("nf::", "")


def counters(root, counter):
    path = find(root, "counter_collection.csv")
    out = {}
    if not path:
        return out
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            out.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return out


def kernel_source_hash():
    """sha256 over everything that decides which kernel runs with which shape: the device kernels (nf_kernels.h, nf_assembly.h) AND the host
    file that picks variants, tile widths, chunking and the streaming-load thresholds (neutfem_hip.hip).  bench.py reports a profile's
    traffic figure only while all three are the ones that were profiled, and never under NEUTFEM_OPTS (a run with overridden launch
    parameters is not the run that was profiled)."""
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for fn in ("nf_kernels.h", "nf_assembly.h", "neutfem_hip.hip"):
        with open(os.path.join(here, "neutfem_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def main():
    root, tag = sys.argv[1], sys.argv[2]
    cells = 256 ** 3
    stats = find(os.path.join(root, "stats"), "kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(root, f"{tag}_kernel_stats.csv"))
    # rocprofv3's averages include the dispatches queued behind a converged CG solve, which exit at once (no work).  From the
    # kernel trace of the same pass: average duration over the dispatches that did work (>= 1/4 of the kernel's longest one),
    # the figure bench.py's HIP-event timing reports.
    trace = find(os.path.join(root, "stats"), "kernel_trace.csv")
    if trace:
        dur = {}
        with open(trace) as f:
            for row in csv.DictReader(f):
                dur.setdefault(short(row["Kernel_Name"]), []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        with open(os.path.join(root, f"{tag}_kernel_stats_working.csv"), "w") as f:
            f.write("Name,Calls,WorkingCalls,AverageNs_all,AverageNs_working,MaxNs\n")
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                mx = max(v); w = [d for d in v if d >= 0.25 * mx]
                f.write(f"\"{k}\",{len(v)},{len(w)},{sum(v) / len(v):.1f},{sum(w) / len(w):.1f},{mx}\n")
    fetch, write = counters(os.path.join(root, "fetch"), "FETCH_SIZE"), counters(os.path.join(root, "write"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = [v for v in fetch.get(k, []) if v > 0]; w = [v for v in write.get(k, []) if v > 0]
        # a dispatch that exited early (CG already converged) moves nothing: take the median of the upper half
        f.sort(); w.sort()
        fm = statistics.median(f[len(f) // 2:]) if f else 0.0
        wm = statistics.median(w[len(w) // 2:]) if w else 0.0
        kernels[k] = dict(dispatches=max(len(fetch.get(k, [])), len(write.get(k, []))), fetch_kib_raw=fm, write_kib=wm,
                          read_bytes_per_cell_corrected=round(2.0 * fm * 1024 / cells, 3), write_bytes_per_cell=round(wm * 1024 / cells, 3),
                          hbm_bytes_per_cell=round((2.0 * fm + wm) * 1024 / cells, 3))
    commit = os.environ.get("NEUTFEM_COMMIT")                     # the GPU box has no .git: collect.sh is given the commit by the caller
    src_hash = kernel_source_hash()
    doc = dict(tag=tag, commit=commit, kernel_source_sha256=src_hash,
               command="profiles/collect.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, each with --kernel-trace only)",
               mesh="IAEA-3D resampled 256^3", cells=cells,
               units="counter values are KiB per dispatch (median of the upper half of the dispatches); bytes_per_cell = KiB*1024/cells",
               gfx950_correction="FETCH_SIZE x2 (calibration: k_cg_rupdate reads r,q = 16 B/cell); WRITE_SIZE exact", kernels=kernels)
    with open(os.path.join(root, f"{tag}_pmc_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in kernels.items():
        print(f"{k:60s} n={v['dispatches']:5d} read {v['read_bytes_per_cell_corrected']:8.3f} write {v['write_bytes_per_cell']:8.3f} B/cell")


if __name__ == "__main__":
    main()
