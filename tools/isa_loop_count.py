"""Instruction mix of the per-timestep loop of one k_ens_block variant, from the gfx950 assembly.

usage: isa_loop_count.py [--asm FILE | --src ssn_f32.hip] --kernel <mangled-name-substring> [--npt N] [--json OUT] [-- extra hipcc flags]

The time loop is found structurally: the innermost loop (a label and a later branch back to it) of the kernel that
contains the LIF step's `v_log_f32` instructions.  (Round 1's version bracketed the loop by the workgroup barriers
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


def classify(op):
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith("v_"):
        return "plain"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "scalar"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def find_loop(body):
    """(lo, hi) line range of the innermost loop that contains a v_log_f32."""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    logs = [i for i, l in enumerate(body) if "v_log_f32" in l]
    if not logs:
        raise SystemExit("no v_log_f32 in the kernel: not an f32 LIF block kernel")
    best = None
    for lo, hi in loops:
        if lo <= logs[0] and logs[-1] <= hi and (best is None or hi - lo < best[1] - best[0]):
            best = (lo, hi)
    if best is None:
        raise SystemExit("no loop encloses the LIF step")
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--src")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--npt", type=int, default=0, help="neurons per lane of the variant (for per-neuron figures)")
    ap.add_argument("--json")
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
    lo, hi = find_loop(body)
    c = collections.Counter()
    for l in body[lo:hi + 1]:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    by_class = collections.Counter()
    for op, n in c.items():
        by_class[classify(op)] += n
    valu = by_class["trans"] + by_class["packed"] + by_class["plain"]
    slots4 = by_class["packed"] + by_class["plain"] + 2 * by_class["trans"]      # 4-cycle issue slots, transcendentals 2
    out = {
        "kernel": a.kernel, "loop_label_line": lo, "loop_branch_line": hi,
        "instructions_per_wave_timestep": sum(c.values()), "valu_instructions": valu,
        "by_class": dict(by_class), "issue_slots_4cycle_model": slots4,
        "scratch_ops": sum(n for op, n in c.items() if op.startswith("scratch")),
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


if __name__ == "__main__":
    main()
