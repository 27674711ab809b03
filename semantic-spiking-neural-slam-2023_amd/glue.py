"""Collapse of the linear "glue" between the big operators of a timestep.

nengo lowers every ``Connection`` without a synapse into a ``Reset`` + ``ElementwiseInc`` / ``DotInc`` on the
post object's input signal, and every pass-through ``Node`` into a copy.  The
reference's networks are wired almost entirely that way (``networks/slam.py:259-307``,
``networks/binding.py:23-74``): between two populations a value travels through chains such as

    product.in_a  += 0.7071 * dft_a          (level n)
    product.in_a  += 0.7071 * dft_b          (level n + 1: a second increment of the same signal)
    ensembles.x   += product.in_a            (level n + 2: the pass-through node hands it on)

Each link is a 1 000 - 2 000 element vector operator, and each adds a dependency level - a SLAM timestep has 25 of
them, and on the GPU every level is a launch (or a grid barrier).  All of it is linear, so it can be folded at build
time: this pass rewrites the fill / axpy operators of the per-timestep core into ``lincomb`` operators

    dst = self * dst + (const + sum_k alpha_k * src_k)          (element-wise over ``len`` elements)

whose sources are only signals that something other than glue produces (decoded outputs, filter states, table rows,
matvec results).  Intermediate accumulators that nothing else reads disappear; an accumulator that a big operator, a
filter, a probe or another stage reads is materialised once, straight from those sources.  Summation order changes
(a + (b + c) instead of (a + b) + c): results move by rounding only, which the oracle - executing the same
rewritten list - shares; ``oracle/graphwalk.py`` checks the rewritten list against the un-lowered network.
"""
import numpy as np


def _sig_ranges(acc):
    """op_access tuple -> (writes, reads) lists of (lo, hi) signal ranges."""
    w = [(r[1], r[2]) for cls in (0, 1, 3) for r in acc[cls] if r[0] == "s"]
    rd = [(r[1], r[2]) for r in acc[2] if r[0] == "s"]
    return w, rd


class _Segments:
    """Elementary segments between sorted boundaries; lookup of the segments a range covers."""

    def __init__(self, bounds):
        self.b = np.array(sorted(bounds), dtype=np.int64)

    def index(self, off):
        i = int(np.searchsorted(self.b, off))
        assert i < len(self.b) and self.b[i] == off, "range endpoint is not a segment boundary"
        return i

    def cover(self, lo, hi):
        """Segment indices whose span intersects [lo, hi) (the range need not be aligned)."""
        i = int(np.searchsorted(self.b, lo, side="right")) - 1
        j = int(np.searchsorted(self.b, hi, side="left"))
        return range(max(i, 0), min(j, len(self.b) - 1))

    def span(self, i):
        return int(self.b[i]), int(self.b[i + 1])


def collapse_glue(ops, model, op_access, protected):
    """``ops``: unscheduled operators of the core stage (dicts).  ``protected``: signal ranges other stages or probes
    touch.  Returns the rewritten list (``seq`` keeps creation order for the scheduler) and a stats dict."""
    glue = [o for o in ops if o["kind"] in ("fill", "axpy") and not o.get("partial_zero")]
    hard = [dict(o) for o in ops if not (o["kind"] in ("fill", "axpy") and not o.get("partial_zero"))]
    stats = {"glue_in": len(glue), "glue_out": len(glue), "inlined_segments": 0, "retargeted_inputs": 0}
    if not any(o["kind"] == "axpy" for o in glue):
        return ops, stats
    hard_acc = [_sig_ranges(op_access(o, model)) for o in hard]

    # ---- boundaries: every range endpoint, closed under the translations dst <-> src of the axpy operators -----------
    bounds = set()
    for o in glue:
        bounds.update((o["dst"], o["dst"] + o["len"]))
        if o["kind"] == "axpy":
            bounds.update((o["src"], o["src"] + o["len"]))
    for w, rd in hard_acc:
        for lo, hi in w + rd:
            bounds.update((lo, hi))
    for lo, hi in protected:
        bounds.update((lo, hi))
    axpys = [(o["dst"], o["src"], o["len"]) for o in glue if o["kind"] == "axpy"]
    changed = True
    while changed:
        changed = False
        arr = np.array(sorted(bounds), dtype=np.int64)
        for d, s, ln in axpys:
            for a, b in ((d, s), (s, d)):
                inside = arr[(arr > a) & (arr < a + ln)]
                for p in inside:
                    q = int(p) - a + b
                    if q not in bounds:
                        bounds.add(q)
                        changed = True
    seg = _Segments(bounds)
    n_seg = len(seg.b) - 1

    # ---- per-segment facts -----------------------------------------------------------------------------------------------
    hard_w = np.zeros(n_seg, bool)
    hard_r = np.zeros(n_seg, np.int64)           # number of readers other than glue (big operators, filters, probes, other stages)
    for w, rd in hard_acc:
        for lo, hi in w:
            hard_w[list(seg.cover(lo, hi))] = True
        for lo, hi in rd:
            hard_r[list(seg.cover(lo, hi))] += 1
    for lo, hi in protected:
        hard_r[list(seg.cover(lo, hi))] += 1
    pieces = [[] for _ in range(n_seg)]          # per destination segment: (seq, "fill", value) | (seq, mode, alpha, src segment)
    for o in glue:
        d0 = o["dst"]
        for i in seg.cover(d0, d0 + o["len"]):
            lo, hi = seg.span(i)
            if o["kind"] == "fill":
                pieces[i].append((o["seq"], "fill", float(o["value"]), -1))
            else:
                j = seg.index(o["src"] + (lo - d0))
                assert seg.span(j)[1] - seg.span(j)[0] == hi - lo
                pieces[i].append((o["seq"], o["mode"], float(o["alpha"]), j))
    n_sets = np.array([sum(1 for p in ps if p[1] in ("fill", "set")) for ps in pieces])
    has_glue = np.array([len(ps) > 0 for ps in pieces])
    pure = has_glue & ~hard_w & (n_sets == 1)
    inlinable = pure & (hard_r == 0)

    # ---- expressions ---------------------------------------------------------------------------------------------------
    memo = {}

    def expr(i, stack=()):
        """(const, {source segment: coefficient}) of a pure segment, sources being non-inlinable segments."""
        if i in memo:
            return memo[i]
        if i in stack:
            raise ValueError("glue cycle within a timestep")
        c, terms = 0.0, {}
        for _, mode, val, j in sorted(pieces[i], key=lambda p: p[0]):
            if mode == "fill":
                c += val
                continue
            if inlinable[j]:
                cj, tj = expr(j, stack + (i,))
                c += val * cj
                for k, a in tj.items():
                    terms[k] = terms.get(k, 0.0) + val * a
            else:
                terms[j] = terms.get(j, 0.0) + val
        memo[i] = (c, terms)
        return memo[i]

    # ---- copy propagation ------------------------------------------------------------------------------------------------
    # An operator whose whole input vector is a plain copy of another signal (a pass-through node handing a filter state
    # or a clean-up result on: `x = 1.0 * s`) reads that signal itself.  The copy then has one reader fewer and disappears
    # if that was the last one; on the device it is one dependency level - one launch - less in front of the operator.
    # Reads precede updates within a timestep, so a filter state read directly is still the previous step's value.
    input_field = {"matvec": ("src", "cols"), "cleanup": ("src", "cols"), "lowpass": ("src", "len"), "neurons": ("j", "n")}
    filters = [o for o in hard if o["kind"] == "lowpass"]

    def _resolved_sources(lo, ln):
        """Signal ranges a read of [lo, lo + ln) really depends on: pure glue segments are looked through."""
        out_r = []
        for i in seg.cover(lo, lo + ln):
            if pure[i]:
                try:
                    _, terms = expr(i)
                except ValueError:
                    terms = {}
                out_r.extend(seg.span(j) for j in terms)
            else:
                out_r.append(seg.span(i))
        return out_r

    def _closes_update_cycle(op, new_lo, ln):
        """Would `op` (a filter: it reads its source, then updates its state) reading [new_lo, new_lo + ln) directly sit on a
        cycle of filters that read each other's STATE?  The scheduler orders every reader of a state before the state's
        update, so filter A reading B's state and B reading A's (two pass-through nodes joined by synapses in both
        directions) has no order; with the copy in between, the copy is the reader and both filters update afterwards."""
        seen, todo = set(), [(new_lo, new_lo + ln)]
        while todo:
            lo, hi = todo.pop()
            for f in filters:
                if id(f) in seen or f["dst"] >= hi or f["dst"] + f["len"] <= lo:
                    continue
                if f is op:
                    return True
                seen.add(id(f))
                todo.extend(_resolved_sources(f["src"], f["len"]))
        return False
    changed = True
    while changed:
        changed = False
        memo.clear()
        inlinable[:] = pure & (hard_r == 0)
        for o in hard:
            f = input_field.get(o["kind"])
            if f is None:
                continue
            lo0, ln = o[f[0]], o[f[1]]
            segs = list(seg.cover(lo0, lo0 + ln))
            off = None
            for i in segs:
                if not pure[i]:
                    off = None
                    break
                c, terms = expr(i)
                if c != 0.0 or len(terms) != 1:
                    off = None
                    break
                (j, a), = terms.items()
                d = seg.span(j)[0] - seg.span(i)[0]
                if a != 1.0 or (off is not None and d != off):
                    off = None
                    break
                off = d
            if off is None or off == 0:
                continue
            if o["kind"] == "lowpass" and _closes_update_cycle(o, lo0 + off, ln):
                continue              # keep the copy: two filters reading each other's state directly cannot be ordered
            src_segs = list(seg.cover(lo0 + off, lo0 + off + ln))
            hard_r[segs] -= 1
            hard_r[src_segs] += 1
            o[f[0]] = lo0 + off
            stats["retargeted_inputs"] += 1
            changed = True
            break                     # expressions depend on which copies are still read: start over
    memo.clear()
    inlinable[:] = pure & (hard_r == 0)

    out = list(hard)
    new = []                                     # (dst lo, len, self coefficient, const, ((src lo, alpha), ...), seq)
    for i in range(n_seg):
        if not has_glue[i]:
            continue
        lo, hi = seg.span(i)
        seq = min(p[0] for p in pieces[i])
        if pure[i]:
            if inlinable[i]:
                stats["inlined_segments"] += 1
                continue
            c, terms = expr(i)
            new.append((lo, hi - lo, 0.0, c, tuple(sorted((seg.span(j)[0], a) for j, a in terms.items() if a != 0.0)), seq))
            continue
        # mixed accumulator (a big operator adds into it as well, or it has no reset of its own): the reset stays a fill,
        # all glue increments become one
        c_set, c_inc, t_inc, set_terms = None, 0.0, {}, None
        for _, mode, val, j in sorted(pieces[i], key=lambda p: p[0]):
            if mode == "fill":
                c_set = (c_set or 0.0) + val
                continue
            tgt = {}
            if inlinable[j]:
                cj, tj = expr(j)
                add_c = val * cj
                for k, a in tj.items():
                    tgt[k] = tgt.get(k, 0.0) + val * a
            else:
                add_c = 0.0
                tgt[j] = val
            if mode == "set":
                set_terms = (add_c, tgt) if set_terms is None else (set_terms[0] + add_c, {k: set_terms[1].get(k, 0.0) + tgt.get(k, 0.0)
                                                                                              for k in set(set_terms[1]) | set(tgt)})
            else:
                c_inc += add_c
                for k, a in tgt.items():
                    t_inc[k] = t_inc.get(k, 0.0) + a
        if c_set is not None or set_terms is not None:
            c = (c_set or 0.0) + (set_terms[0] if set_terms else 0.0)
            tt = set_terms[1] if set_terms else {}
            new.append((lo, hi - lo, 0.0, c, tuple(sorted((seg.span(j)[0], a) for j, a in tt.items() if a != 0.0)), seq))
        if t_inc or c_inc != 0.0:
            new.append((lo, hi - lo, 1.0, c_inc, tuple(sorted((seg.span(j)[0], a) for j, a in t_inc.items() if a != 0.0)), seq + 0.5))

    # ---- merge neighbours of identical structure, emit -------------------------------------------------------------------
    new.sort(key=lambda e: (e[2], len(e[4]), e[3], tuple(a for _, a in e[4]), e[0]))
    merged = []
    for e in new:
        if merged:
            m = merged[-1]
            same = (m[2] == e[2] and m[3] == e[3] and len(m[4]) == len(e[4]) and m[0] + m[1] == e[0] and
                    all(ma == ea and ms + m[1] == es for (ms, ma), (es, ea) in zip(m[4], e[4])))
            if same:
                merged[-1] = (m[0], m[1] + e[1], m[2], m[3], m[4], min(m[5], e[5]))
                continue
        merged.append(e)
    for lo, ln, a_self, c, terms, seq in merged:
        if a_self == 0.0 and not terms:
            out.append({"kind": "fill", "dst": lo, "len": ln, "value": c, "seq": seq})
        elif len(terms) == 1 and c == 0.0:
            out.append({"kind": "axpy", "dst": lo, "src": terms[0][0], "len": ln, "alpha": terms[0][1],
                        "mode": "inc" if a_self == 1.0 else "set", "seq": seq})
        else:
            out.append({"kind": "lincomb", "dst": lo, "len": ln, "self": a_self, "const": c,
                        "srcs": [s for s, _ in terms], "alphas": [a for _, a in terms], "seq": seq})
    out.sort(key=lambda o: o["seq"])
    stats["glue_out"] = len(out) - len(hard)
    return out, stats
