"""Kernel-trace CSV of a SLAM run (rocprofv3 --kernel-trace --output-format csv) -> durations and gaps of the k_round launches.
usage: trace_gaps.py <kernel_trace.csv>      prints the steady-state averages per position of the repeating round pattern"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_round" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
grid = [int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) for r in rows]
n = len(rows)
dur = [en[i] - st[i] for i in range(n)]
gap = [st[i + 1] - en[i] for i in range(n - 1)] + [0]
print("k_round dispatches", n)
# steady state: the last 40 % of the trace; group by block count
lo = int(0.6 * n)
by = collections.defaultdict(list)
for i in range(lo, n - 1):
    if gap[i] < 50000:
        by[grid[i]].append((dur[i], gap[i]))
tot = 0.0
for g, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 8:
        continue
    d = sum(x[0] for x in v) / len(v); gp = sum(x[1] for x in v) / len(v)
    print(f"blocks {g:6d}: {len(v):5d} launches, duration {d / 1e3:7.2f} us, gap behind it {gp / 1e3:6.2f} us")
span = en[n - 2] - st[lo]
print("steady-state span per launch: %.2f us; sum of durations / launches %.2f us; sum of gaps / launches %.2f us" % (
    span / 1e3 / (n - 2 - lo), sum(dur[lo:n - 1]) / 1e3 / (n - 1 - lo), sum(g for g in gap[lo:n - 1] if g < 50000) / 1e3 / (n - 1 - lo)))
