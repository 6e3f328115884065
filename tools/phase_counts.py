#!/usr/bin/env python3
"""Per-phase instruction classes of one rollout-kernel variant, from `hipcc -S --cuda-device-only` output.
usage: phase_counts.py file.s KERNEL_SUBSTRING [trip counts of the depth-2 loops, in order; 0 = rarely taken]
Prints, for the step loop (the depth-1 loop that holds the barriers) and each loop nested in it, the static counts of
plain VALU / packed fp32 VALU / quarter-rate transcendentals / SALU / LDS / vector memory; with trip counts, the dynamic
per-step totals (loop bodies x trips + the straight-line rest)."""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
trips = [int(x) for x in sys.argv[3:]]
start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
body = txt[start:end]
TRANS = ("v_sqrt_f32", "v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")
def cls(op):
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_pk_"): return "packed"
    if op.startswith("v_"): return "plain"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_"): return "salu"
    return "other"
# loops: label lines carry "Loop Header: Depth=N" / "Inner Loop Header: Depth=N"; a loop runs to its back edge
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):\s*;.*Depth[= ](\d+)", l) or (re.match(r"^(\.LBB\d+_\d+):", l) and re.search(r"Loop Header: Depth=(\d+)", "".join(body[i:i + 3])) and None)
    if re.match(r"^(\.LBB\d+_\d+):", l) and re.search(r"Loop Header: Depth=(\d+)", " ".join(body[i:i + 8]).split("\t")[0]):
        name = re.match(r"^(\.LBB\d+_\d+):", l).group(1)
        depth = int(re.search(r"Loop Header: Depth=(\d+)", " ".join(body[i:i + 8])).group(1))
        last = max((j for j, b in enumerate(body) if re.search(r"s_cbranch\w*\s+" + re.escape(name) + r"\b|s_branch\s+" + re.escape(name) + r"\b", b) and j > i), default=None)
        if last: loops.append((i, last, depth, name))
def count(lo, hi, skip=()):
    c = collections.Counter()
    for j in range(lo, hi + 1):
        if any(a <= j <= b for a, b in skip): continue
        l = body[j]
        if not l.startswith("\t") or l.strip().startswith((";", ".")): continue
        c[cls(l.split()[0])] += 1
    return c
def fmt(c): return "  ".join(f"{k} {c.get(k, 0):4d}" for k in ("plain", "packed", "trans", "salu", "lds", "vmem"))
step = [l for l in loops if l[2] == 1 and any("s_barrier" in b for b in body[l[0]:l[1]])]
if not step: step = [max((l for l in loops if l[2] == 1), key=lambda l: l[1] - l[0])]
s0, s1, _, sname = step[0]
inner = [l for l in loops if l[2] == 2 and s0 <= l[0] <= s1]
print(f"step loop {sname}: lines {s0}..{s1}, {len(inner)} nested loops")
tot = collections.Counter()
for k, (a, b, d, n) in enumerate(inner):
    c = count(a, b)
    t = trips[k] if k < len(trips) else None
    print(f"  loop {k} {n:12s} body: {fmt(c)}" + (f"   x {t}" if t is not None else ""))
    if t:
        for kk, v in c.items(): tot[kk] += v * t
rest = count(s0, s1, skip=[(a, b) for a, b, _, _ in inner])
print(f"  straight-line rest   : {fmt(rest)}   (includes rarely taken blocks)")
if trips:
    for kk, v in rest.items(): tot[kk] += v
    print(f"  per step, dynamic    : {fmt(tot)}   VALU total {tot['plain'] + tot['packed'] + tot['trans']}")
