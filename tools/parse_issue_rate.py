"""tools/valu_issue_rate.hip output -> profiles/valu_issue_rate.json: SIMD cycles per wave64 instruction by waves per SIMD.

usage: parse_issue_rate.py gpurun_out/.../valu_issue_rate.txt profiles/valu_issue_rate.json"""
import json
import re
import sys

rows, notes = {}, []
for line in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+waves/SIMD (\d)\s+cycles/inst/wave\s+([\d.]+)\s+SIMD cycles per wave-instruction\s+([\d.]+)\s+clock\s+(\d+) MHz", line)
    if m:
        name = m.group(1).strip().split(" ")[0] if m.group(1).startswith("v_") else m.group(1).strip()
        rows.setdefault(name, {})[m.group(2)] = float(m.group(4))
    elif line.startswith("#"):
        notes.append(line[1:].strip())
out = {"what": "SIMD cycles one SIMD of an MI355X CU spends per wave64 instruction, independent instruction streams, by resident waves per SIMD "
               "(tools/valu_issue_rate.hip; s_memtime around 1.28 M instructions per wave, median over all waves, one workgroup per CU)",
       "notes": notes}
out.update(rows)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in rows.items() if k.startswith("v_")}, indent=None))
