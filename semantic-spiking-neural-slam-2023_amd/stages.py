"""Stage partitioning of the operator list: pre | core | post.

A timestep of a spiking network is a strict recurrence only where state feeds back through the
neurons.  Everything *upstream* of the neurons that depends on tabulated inputs alone (``pre``: input
transforms and their synapse filters) and everything *downstream* that nothing feeds back from
(``post``: linear read-outs, output and probe filters) is feed-forward in time, so it can be executed
for a whole block of B timesteps at once: a per-step GEMV becomes one GEMM over the block (the matrix
is read once per block instead of once per step), a Lowpass synapse becomes a scan along time.
Only the ``core`` (neurons, learning rules, clean-up / gate and whatever lies on a feedback cycle
through them) is stepped one timestep at a time.

For ``PathIntegration`` this leaves one ensemble-array kernel plus one small program per timestep; the
``to_Fourier`` input chain runs before and the ``to_SSP`` read-out chain after each block.  The same
split is what the multi-GPU path exchanges across ranks: the core->post boundary of a block.

Given the merged, unscheduled operators, ``stage_ops`` returns them scheduled in stage-major order
(a valid single-step order for the oracle and for step-by-step execution) and annotated with

  stage    0 pre | 1 core | 2 post
  level    scheduling round (see builder.schedule_ops)
  border   position in the time-batched order of its stage (pre/post only)
  src_prev 1 if the op's source operand is a synapse state read *before* that state's update in the
           step, i.e. the batched executor must read the previous step's row

and the model gets ``stage_info``: boundary ranges pre->core and core->post, and per-probe stages.
"""
import numpy as np

PRE, CORE, POST = 0, 1, 2
CORE_KINDS = ("ensarray", "neurons", "pes", "voja", "cleanup", "gate")


def _merge_ranges(ranges):
    out = []
    for lo, hi in sorted(ranges):
        if out and lo <= out[-1][1]:
            out[-1][1] = max(out[-1][1], hi)
        else:
            out.append([lo, hi])
    return [(lo, hi) for lo, hi in out]


def _intersect(a, b):
    """Intersection of two merged range lists."""
    out, i, j = [], 0, 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if lo < hi:
            out.append((lo, hi))
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def _split_elementwise(ops, stage):
    """Split fill / axpy / lowpass operators of the batched stages at every range endpoint of the other
    operators of those stages (iterated to closure).  The merge pass may have glued together element
    ranges that sit at different depths of the time-series data flow (e.g. ``a.in += s1`` and
    ``b.in += s2`` where s2 filters a); operator-level dependencies would then show a false cycle."""
    def ranges(o):
        k = o["kind"]
        if k == "fill":
            return [(o["dst"], o["dst"] + o["len"])]
        if k in ("axpy", "lowpass"):
            return [(o["dst"], o["dst"] + o["len"]), (o["src"], o["src"] + o["len"])]
        if k == "matvec":
            return [(o["dst"], o["dst"] + o["rows"]), (o["src"], o["src"] + o["cols"])]
        if k == "table":
            return [(o["dst"], o["dst"] + o["width"])]
        return []
    for _ in range(64):
        pts = set()
        for o, st in zip(ops, stage):
            if st != CORE:
                for lo, hi in ranges(o):
                    pts.add(lo)
                    pts.add(hi)
        new_ops, new_stage, changed = [], [], False
        for o, st in zip(ops, stage):
            if st == CORE or o["kind"] not in ("fill", "axpy", "lowpass"):
                new_ops.append(o)
                new_stage.append(st)
                continue
            L = o["len"]
            bases = [o["dst"]] + ([o["src"]] if "src" in o else [])
            cuts = sorted({p - b for b in bases for p in pts if b < p < b + L})
            if not cuts:
                new_ops.append(o)
                new_stage.append(st)
                continue
            changed = True
            edges = [0] + cuts + [L]
            for lo, hi in zip(edges, edges[1:]):
                piece = dict(o)
                piece["dst"], piece["len"] = o["dst"] + lo, hi - lo
                if "src" in o:
                    piece["src"] = o["src"] + lo
                new_ops.append(piece)
                new_stage.append(st)
        ops, stage = new_ops, new_stage
        if not changed:
            break
    return ops, stage


def stage_ops(ops, model, r_allocs, schedule_ops, op_access, overlap, enable=True, collapse=None):
    n = len(ops)
    acc = [op_access(o, model) for o in ops]
    sig_writes = [[r for cls in (0, 1, 3) for r in a[cls]] for a in acc]
    reads = [list(a[2]) for a in acc]

    # ---- data-flow edges (any lag): writer -> reader ---------------------------------------------
    succ = [set() for _ in range(n)]
    pred = [set() for _ in range(n)]
    for i in range(n):
        for j in range(n):
            if i != j and any(overlap(w, r) for w in sig_writes[i] for r in reads[j]):
                succ[i].add(j)
                pred[j].add(i)

    def reach(start, nxt):
        seen, todo = set(start), list(start)
        while todo:
            u = todo.pop()
            for v in nxt[u]:
                if v not in seen:
                    seen.add(v)
                    todo.append(v)
        return seen

    seeds = [i for i, o in enumerate(ops) if o["kind"] in CORE_KINDS]
    stage = [CORE] * n
    if enable and not seeds:
        stage = [POST] * n          # no neurons at all (e.g. a linear read-out): everything is feed-forward
    if enable and seeds:
        # along every edge i -> j: j an ancestor of a seed => i is too; i a descendant => j is too.
        # Hence stage[i] <= stage[j] on all data-flow edges without further fixing.
        desc = reach(seeds, succ)
        anc = reach(seeds, pred)
        for i in range(n):
            if i in desc and i in anc:
                stage[i] = CORE
            elif i in anc:
                stage[i] = PRE
            else:
                stage[i] = POST
        # a synapse state must be read (before its update) no later than the stage that updates it,
        # or stage-major order would hand the reader the already-updated value
        changed = True
        while changed:
            changed = False
            for i, o in enumerate(ops):
                if o["kind"] != "lowpass":
                    continue
                for j in succ[i]:
                    if stage[j] > stage[i] and any(overlap(w, r) for w in acc[i][3] for r in reads[j]):
                        if ops[j]["kind"] in CORE_KINDS:
                            return stage_ops(ops, model, r_allocs, schedule_ops, op_access, overlap, enable=False, collapse=collapse)
                        stage[j] = stage[i]
                        changed = True
                        # ... and whatever else j reads must then be written no later than j's new stage (stage[i] <= stage[j] on
                        # every edge): a merged hand-off that copies the states of a core filter AND of a read-out filter into
                        # node inputs would otherwise run per timestep on a state the post stage only updates after the block
                        # (round 4: a second decoded connection of a dense population read zeros)
                        todo = [j]
                        while todo:
                            v = todo.pop()
                            for q in pred[v]:
                                if stage[q] > stage[v]:
                                    if ops[q]["kind"] in CORE_KINDS:
                                        return stage_ops(ops, model, r_allocs, schedule_ops, op_access, overlap, enable=False, collapse=collapse)
                                    stage[q] = stage[v]
                                    todo.append(q)

    if enable and any(st != CORE for st in stage):
        ops, stage = _split_elementwise(list(ops), list(stage))
        n = len(ops)
        acc = [op_access(o, model) for o in ops]
        sig_writes = [[r for cls in (0, 1, 3) for r in a[cls]] for a in acc]

    # ---- per-stage reset of the accumulators (R arena) ---------------------------------------------
    fills = []
    for lo, ln in r_allocs:
        rng = ("s", lo, lo + ln)
        st = [stage[i] for i in range(n) if any(overlap(w, rng) for w in sig_writes[i])]
        fills.append((lo, lo + ln, min(st) if st else POST))
    fills.sort()
    fill_ops, cur = [], None
    for lo, hi, st in fills:
        if cur is not None and cur[2] == st and cur[1] == lo:
            cur[1] = hi
        else:
            if cur is not None:
                fill_ops.append(cur)
            cur = [lo, hi, st]
    if cur is not None:
        fill_ops.append(cur)
    all_ops = list(ops)
    all_stage = list(stage)
    for k, (lo, hi, st) in enumerate(fill_ops):
        all_ops.append({"kind": "fill", "dst": lo, "len": hi - lo, "value": 0.0, "seq": -1 - k})
        all_stage.append(st)

    # ---- schedule each stage, concatenate (stage-major order is a valid single-step order) ----------
    out = []
    level0 = 0
    for st in (PRE, CORE, POST):
        sub = [dict(o) for o, s in zip(all_ops, all_stage) if s == st]
        if st == CORE and collapse is not None:
            # signals that the batched stages or a probe touch keep their values; the rest of the core's linear glue is folded
            keep = [(p["src"], p["src"] + p["width"]) for p in model.probes if "src" in p]
            for o, s in zip(all_ops, all_stage):
                if s != CORE:
                    a = op_access(o, model)
                    keep += [(r[1], r[2]) for cls in range(4) for r in a[cls] if r[0] == "s"]
            sub = collapse(sub, _merge_ranges(keep))
        sched = schedule_ops(sub, model)
        for o in sched:
            o["stage"] = st
            o["level"] += level0
            o["src_prev"] = 0
            o["border"] = -1
        level0 = (max([o["level"] for o in sched]) + 1) if sched else level0
        out.extend(sched)

    # ---- batched order + row flags for the feed-forward stages -----------------------------------
    acc_o = [op_access(o, model) for o in out]
    updater_pos = {}
    for pos, (o, a) in enumerate(zip(out, acc_o)):
        for r in a[3]:
            if r[0] == "s":
                updater_pos[(r[1], r[2])] = pos
    for pos, (o, a) in enumerate(zip(out, acc_o)):
        if o["stage"] == CORE or "src" not in o:
            continue
        ln = o.get("len", o.get("cols", 0))
        src = ("s", o["src"], o["src"] + ln)
        for (lo, hi), upos in updater_pos.items():
            if overlap(src, ("s", lo, hi)) and pos < upos:
                o["src_prev"] = 1
    for st in (PRE, POST):
        idx = [p for p, o in enumerate(out) if o["stage"] == st]
        if not idx:
            continue
        m = len(idx)
        w = [[r for cls in (0, 1, 3) for r in acc_o[p][cls]] for p in idx]
        rd = [list(acc_o[p][2]) for p in idx]
        bsucc = [set() for _ in range(m)]
        indeg = [0] * m
        for a in range(m):
            for b in range(m):
                if a == b:
                    continue
                dep = any(overlap(x, y) for x in w[a] for y in rd[b])                       # writer -> reader
                if not dep and a < b:
                    dep = any(overlap(x, y) for x in w[a] for y in w[b])                     # keep write order
                if dep and b not in bsucc[a]:
                    bsucc[a].add(b)
                    indeg[b] += 1
        ready = sorted(i for i in range(m) if indeg[i] == 0)
        order = []
        while ready:
            a = ready.pop(0)
            order.append(a)
            for b in sorted(bsucc[a]):
                indeg[b] -= 1
                if indeg[b] == 0:
                    ready.append(b)
            ready.sort()
        if len(order) != m:
            # a feedback cycle inside a feed-forward stage would be a partitioning bug: fall back
            return stage_ops(ops, model, r_allocs, schedule_ops, op_access, overlap, enable=False, collapse=collapse)
        for k, a in enumerate(order):
            out[idx[a]]["border"] = k

    # ---- boundaries --------------------------------------------------------------------------------
    def sig_ranges(sel, classes):
        rs = []
        for o, a in zip(out, acc_o):
            if sel(o):
                rs += [(r[1], r[2]) for cls in classes for r in a[cls] if r[0] == "s"]
        return _merge_ranges(rs)

    pre_w = sig_ranges(lambda o: o["stage"] == PRE, (0, 1, 3))
    core_w = sig_ranges(lambda o: o["stage"] == CORE, (0, 1, 3))
    core_need = sig_ranges(lambda o: o["stage"] == CORE, (1, 2))
    post_need = sig_ranges(lambda o: o["stage"] == POST, (1, 2))
    probe_stage = []
    for p in model.probes:
        if "src" not in p:
            probe_stage.append(CORE)
            continue
        rng = ("s", p["src"], p["src"] + p["width"])
        st = [o["stage"] for o, a in zip(out, acc_o) if any(overlap(x, rng) for cls in (0, 1, 3) for x in a[cls])]
        probe_stage.append(max(st) if st else POST)
    probe_need = _merge_ranges([(p["src"], p["src"] + p["width"]) for p, s in zip(model.probes, probe_stage)
                                if "src" in p and s == POST])
    all_w = sig_ranges(lambda o: True, (0, 1, 3))

    def bridge(ranges):
        """Join neighbouring ranges across gaps nobody ever writes (decoded rows dropped as dead, builder._live_elements):
        handing such constant elements over with their neighbours is harmless, and the device copies one range, not one
        per ensemble."""
        out_r = []
        for lo, hi in ranges:
            if out_r and not _intersect([(out_r[-1][1], lo)], all_w):
                out_r[-1] = (out_r[-1][0], hi)
            else:
                out_r.append((lo, hi))
        return out_r

    model.stage_info = {
        "enabled": bool(enable and any(o["stage"] != CORE for o in out)),
        "pre_to_core": bridge(_intersect(pre_w, core_need)),
        "core_to_post": bridge(_intersect(core_w, _merge_ranges(post_need + probe_need))),
        "probe_stage": probe_stage,
        "n_pre": sum(o["stage"] == PRE for o in out), "n_core": sum(o["stage"] == CORE for o in out),
        "n_post": sum(o["stage"] == POST for o in out),
        "batched_written": _merge_ranges(pre_w + sig_ranges(lambda o: o["stage"] == POST, (0, 1, 3))),
    }
    return out
