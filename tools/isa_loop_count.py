"""Instruction mix of the time loop of one k_ens_block variant (between the workgroup barriers that bracket the
neuron work).  usage: isa_loop_count.py ssn_f32.hip <mangled-name-substring> [extra hipcc flags]"""
import collections
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
asm = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", src, "-o", "-"] + sys.argv[3:],
                     capture_output=True, text=True).stdout.split("\n")
start = [i for i, l in enumerate(asm) if key in l and l.rstrip().endswith(":") or (key in l and ": ;" in l)][0]
end = [i for i in range(start, len(asm)) if "s_endpgm" in asm[i]][0]
body = asm[start:end]
bars = [i for i, l in enumerate(body) if "s_barrier" in l]
logs = [i for i, l in enumerate(body) if "v_log_f32" in l]
lo = max(b for b in bars if b < logs[0])
hi = min(b for b in bars if b > logs[-1])
c = collections.Counter()
for l in body[lo:hi]:
    l = l.strip()
    if not l or l[0] in ";." or l.endswith(":"):
        continue
    c[l.split()[0]] += 1
print("instructions per timestep (one wave):", sum(c.values()), " scratch ops:", sum(v for k, v in c.items() if k.startswith("scratch")))
trans = ("v_rcp", "v_log", "v_exp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
valu = sum(v for k, v in c.items() if k.startswith("v_"))
slots = sum(v * (2 if k.startswith(trans) else 1) for k, v in c.items() if k.startswith("v_"))
print("VALU instructions %d, issue slots of 4 cycles %d (transcendentals count 2: 8-cycle issue, MI355X_MICROARCH.md)" % (valu, slots))
for k, v in c.most_common(40):
    print("  %-28s %d" % (k, v))
