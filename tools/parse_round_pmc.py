"""HBM traffic of the SLAM round plan per timestep from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/experiments/slam_flags.py: sums the counters over every k_round dispatch and divides by the timesteps run.
usage: parse_round_pmc.py fetch_counter_collection.csv write_counter_collection.csv n_timesteps us_per_timestep
Corrections per MI355X_MICROARCH.md (HBM section): counters in KB; FETCH_SIZE x2 on gfx950 (calibrated for wide coalesced
reads; sparse gathers may be over-counted); WRITE_SIZE exact."""
import csv, sys


def total(path, name):
    s, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "k_round" in r["Kernel_Name"]:
            s += float(r["Counter_Value"]); n += 1
    return s, n


fetch_kb, nf = total(sys.argv[1], "FETCH_SIZE")
write_kb, nw = total(sys.argv[2], "WRITE_SIZE")
steps, us = int(sys.argv[3]), float(sys.argv[4])
rd, wr = 2 * fetch_kb * 1024 / steps, write_kb * 1024 / steps
print("k_round dispatches sampled: %d (FETCH_SIZE pass), %d (WRITE_SIZE pass); %d timesteps" % (nf, nw, steps))
print("per timestep: %.1f MB read + %.1f MB written = %.1f MB; at %.1f us per timestep (unprofiled) = %.0f GB/s = %.0f %% of the 8 TB/s HBM peak"
      % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6, us, (rd + wr) / us / 1e3, 100 * (rd + wr) / us / 1e3 / 8000))
