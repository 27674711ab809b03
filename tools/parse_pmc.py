"""Summarise rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) and kernel stats for the dominant kernel.

usage: parse_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel_stats.csv> <out.json> [units_per_launch] [kernel-substring]
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane) -> doubled; WRITE_SIZE is exact."""
import csv, json, sys

def mean_counter(path, name, kernel):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)

KERNEL = sys.argv[6] if len(sys.argv) > 6 else "k_ensarray"
fetch_kb, nf = mean_counter(sys.argv[1], "FETCH_SIZE", KERNEL)
write_kb, nw = mean_counter(sys.argv[2], "WRITE_SIZE", KERNEL)
stats = {}
for r in csv.DictReader(open(sys.argv[3])):
    if "ssn::" in r["Name"] or "k_set_block" in r["Name"]:
        stats[r["Name"].split("(")[0].replace("void ", "")] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                             "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
kname = next((k for k in stats if KERNEL in k), KERNEL)
out = {"kernel": kname, "dtype": "f32", "units_per_launch": int(sys.argv[5]) if len(sys.argv) > 5 else 5080000,
       "launches_sampled": {"fetch": nf, "write": nw},
       "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
       "read_bytes_per_launch": 2 * fetch_kb * 1024, "write_bytes_per_launch": write_kb * 1024,
       "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
       "corrections": "FETCH_SIZE x2 on gfx950 (calibrated for 16-B/lane coalesced streaming reads; applied to the whole count, so the sparse 32-B decoder gathers may be over-counted); WRITE_SIZE exact; separate --pmc passes",
       "kernel_stats": stats}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("read_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch")}))
for k, v in stats.items():
    print("%-60s calls %5d avg %8.2f us" % (k[:60], v["calls"], v["avg_us"]))
