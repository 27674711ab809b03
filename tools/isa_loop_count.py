"""Instruction mix of the per-timestep loop of one k_ens_block variant, from the gfx950 assembly.

usage: isa_loop_count.py [--asm FILE | --src ssn_f32.hip] --kernel <mangled-name-substring> [--npt N] [--json OUT] [-- extra hipcc flags]

The time loop is found structurally: the innermost loop - by the compiler's block annotations - of the kernel that
contains the LIF step's transcendentals (`v_log_f32`; `v_rcp_f32` only since round 4).  (Round 1's version bracketed the loop by the workgroup barriers
around the first / last `v_log`; after a kernel change the second barrier lay outside the time loop and the count
came out as 301 instead of 608 issue slots - VERDICT r1, roofline item.)

Issue cost model (cycles one SIMD spends per wave64 instruction), measured on MI355X by tools/valu_issue_rate.hip
(profiles/round2_valu_issue_rate.txt) - pass --waves-per-simd to pick the column; the JSON keeps every column.
"""
import argparse
import collections
import json
import re
import subprocess
import sys

TRANS = ("v_rcp", "v_log", "v_exp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
# plain (one value per lane) f32 multiply / add / FMA: the only vector instructions a SIMD issues at its full SIMD-32 rate when
# a second wave is resident (tools/valu_issue_rate.hip); everything else takes a quad-cycle slot
PLAIN_FMA = ("v_fma_f32", "v_fmac_f32", "v_add_f32_e", "v_sub_f32_e", "v_mul_f32_e", "v_mac_f32")


def classify(op):
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith(PLAIN_FMA):
        return "plain_fma"
    if op.startswith("v_"):
        return "plain"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "scalar"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def loop_lines(body):
    """Lines of the innermost loop that contains the LIF step's v_log_f32, by the compiler's own block annotations
    ('=>This Inner Loop Header' / 'in Loop: Header=BBx_y'): block layout need not be contiguous (the latch block of a
    rotated loop is often placed in front of its header)."""
    blocks = []          # (block id, loop header id or None, [lines])
    cur = None
    for l in body:
        m = re.match(r"^(?:\.LBB\d+_(\d+):|; %bb\.(\d+):)(.*)$", l)
        if m:
            cur = [m.group(1) or m.group(2), None, []]
            blocks.append(cur)
            l_rest = m.group(3)
        else:
            l_rest = l
        if cur is None:
            continue
        h = re.search(r"in Loop: Header=BB\d+_(\d+) Depth=(\d+)", l_rest)
        if h:
            cur[1] = h.group(1)
        elif re.search(r"=>\s*This (Inner )?Loop Header", l_rest):
            cur[1] = cur[0]
        if not m:
            cur[2].append(l)
    # (round 4: the spike time no longer takes a v_log_f32; the reciprocal of J - 1 is the LIF step's transcendental now)
    with_log = [b for b in blocks if any("v_log_f32" in x for x in b[2])] or [b for b in blocks if any("v_rcp_f32" in x for x in b[2]) and b[1] is not None]
    if not with_log:
        raise SystemExit("no v_log_f32 / v_rcp_f32 in a loop of the kernel: not an f32 LIF block kernel")
    heads = {b[1] for b in with_log}
    if len(heads) != 1 or None in heads:
        raise SystemExit(f"the LIF step is spread over loops {heads}")
    head = heads.pop()
    return head, [x for b in blocks if b[1] == head for x in b[2]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--src")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--npt", type=int, default=0, help="neurons per lane of the variant (for per-neuron figures)")
    ap.add_argument("--json")
    ap.add_argument("--merge-into", help="JSON file {'variants': {key: counts}} to update under --variant-key")
    ap.add_argument("--variant-key", help="'tpb,npt,ldsw' as ssn_counters reports the variant")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    if a.asm:
        asm = open(a.asm).read().split("\n")
    else:
        asm = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", a.src, "-o", "-"] + a.extra,
                             capture_output=True, text=True).stdout.split("\n")
    starts = [i for i, l in enumerate(asm) if a.kernel in l and l.rstrip().split(";")[0].rstrip().endswith(":")]
    if not starts:
        raise SystemExit(f"kernel {a.kernel!r} not found")
    start = starts[0]
    end = [i for i in range(start, len(asm)) if "s_endpgm" in asm[i]][0]
    body = asm[start:end]
    head, lines = loop_lines(body)
    c = collections.Counter()
    for l in lines:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    by_class = collections.Counter()
    for op, n in c.items():
        by_class[classify(op)] += n
    valu = by_class["trans"] + by_class["packed"] + by_class["plain"] + by_class["plain_fma"]
    slots4 = by_class["packed"] + by_class["plain"] + by_class["plain_fma"] + 2 * by_class["trans"]      # 4-cycle issue slots, transcendentals 2
    out = {
        "kernel": a.kernel, "loop_header_block": head,
        "instructions_per_wave_timestep": sum(c.values()), "valu_instructions": valu,
        "by_class": dict(by_class), "issue_slots_4cycle_model": slots4,
        "scratch_ops": sum(n for op, n in c.items() if op.startswith("scratch")),
        "ops": {op: n for op, n in sorted(c.items())},          # every mnemonic of the loop: bench.py prices the VALU ones one by one
        "top": c.most_common(60),
    }
    if a.npt:
        out["neurons_per_lane"] = a.npt
        out["valu_per_neuron_step"] = valu / a.npt
        out["issue_slots_per_neuron_step_4cycle_model"] = slots4 / a.npt
    print(json.dumps({k: v for k, v in out.items() if k != "top"}, indent=1))
    for op, n in c.most_common(60):
        print("  %-28s %d" % (op, n))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
    if a.merge_into:
        import os
        doc = {"what": "instruction mix of the per-timestep loop of k_ens_block variants (tools/isa_loop_count.py on the gfx950 assembly "
                       "of csrc/ssn_f32.hip); bench.py prices the VALU-issue roofline with it", "variants": {}}
        if os.path.exists(a.merge_into):
            with open(a.merge_into) as f:
                doc = json.load(f)
        doc["variants"][a.variant_key] = {k: v for k, v in out.items() if k != "top"}
        doc["variants"][a.variant_key]["top"] = dict(c.most_common(25))
        with open(a.merge_into, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
