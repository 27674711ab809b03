"""tools/valu_issue_rate.hip output -> profiles/valu_issue_rate.json.

Per instruction and resident waves per SIMD, two bases:
  * "<name>": SIMD cycles per wave64 instruction from the MEDIAN wave's s_memtime stamps (round 2's basis; understates the
    SIMD's time when the waves of a SIMD are not served evenly - 3 waves per SIMD);
  * "ns"/"<name>": nanoseconds of SIMD time per wave64 instruction from the launch's wall time (HIP events) over all
    instructions the SIMD issued - independent of arbitration and of the clock under load; bench.py's roofline basis.

usage: parse_issue_rate.py gpurun_out/.../valu_issue_rate.txt profiles/valu_issue_rate.json"""
import json
import re
import sys

rows, ns, spread, notes = {}, {}, {}, []
for line in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+waves/SIMD (\d)\s+cycles/inst/wave\s+([\d.]+)\s+SIMD cycles per wave-instruction\s+([\d.]+)\s+clock\s+(\d+) MHz"
                 r"\s+kernel ([\d.]+) ms(?:\s+wave cycles/inst min\s+([\d.]+) max\s+([\d.]+)\s+SIMD ns per wave-instruction \(kernel time\) ([\d.]+))?", line)
    if m:
        name = m.group(1).strip().split(" ")[0] if m.group(1).startswith("v_") else m.group(1).strip()
        rows.setdefault(name, {})[m.group(2)] = float(m.group(4))
        if m.group(9):
            ns.setdefault(name, {})[m.group(2)] = float(m.group(9))
            spread.setdefault(name, {})[m.group(2)] = {"median": float(m.group(3)), "min": float(m.group(7)), "max": float(m.group(8)),
                                                        "clock_mhz": float(m.group(5))}
    elif line.startswith("#"):
        notes.append(line[1:].strip())
out = {"what": "SIMD cycles one SIMD of an MI355X CU spends per wave64 instruction, independent instruction streams, by resident waves per SIMD "
               "(tools/valu_issue_rate.hip; s_memtime around 1.28 M instructions per wave, median over all waves, one workgroup per CU); "
               "'ns': nanoseconds of SIMD time per wave64 instruction from the launch's wall time (the basis bench.py uses); "
               "'wave_cycles_per_instruction': median / min / max over the waves of the per-wave stamps",
       "notes": notes}
out.update(rows)
if ns:
    out["ns"] = ns
    out["wave_cycles_per_instruction"] = spread
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in (ns or rows).items() if k.startswith("v_")}, indent=None))
