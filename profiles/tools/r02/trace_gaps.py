"""per-kernel exec time and gap-to-previous from a rocprofv3 kernel_trace.csv"""
import csv, sys, glob, re, statistics, os
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        nm = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"])).replace("nf::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
rows.sort()
half = rows[len(rows)//2:]          # second (timed) solve
st = {}
for i in range(1, len(half)):
    s, e, nm = half[i]
    gap = s - half[i-1][1]
    d = st.setdefault(nm, [[], []]); d[0].append(e - s); d[1].append(gap)
tot = half[-1][1] - half[0][0]
print(f"{path}: {len(half)} dispatches over {tot/1e6:.2f} ms")
print(f"{'kernel':50s} {'calls':>6s} {'med_exec_us':>11s} {'mean_exec_us':>12s} {'med_gap_us':>10s} {'sum_ms':>8s}")
for nm, (d, g) in sorted(st.items(), key=lambda kv: -sum(kv[1][0]) - sum(kv[1][1])):
    print(f"{nm[:50]:50s} {len(d):6d} {statistics.median(d)/1e3:11.2f} {sum(d)/len(d)/1e3:12.2f} {statistics.median(g)/1e3:10.2f} {(sum(d)+sum(g))/1e6:8.2f}")
