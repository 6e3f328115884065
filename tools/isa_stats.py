#!/usr/bin/env python3
"""Per-kernel ISA statistics from `hipcc -S --cuda-device-only` output: VGPRs, SGPR spills, scratch,
instruction counts inside the step loop.  usage: isa_stats.py file.s [name-filter]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
name = None; body = []; out = []
for ln in txt:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1); body = []
        continue
    if name is None: continue
    body.append(ln)
    m = re.match(r"; Occupancy: (\d+)", ln)
    if m:
        ins = [l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))]
        c = collections.Counter(ins)
        v = next(int(l.split()[-1]) for l in body if l.startswith("; NumVgprs:"))
        sc = next(int(l.split()[-1]) for l in body if l.startswith("; ScratchSize:"))
        short = re.sub(r"_ZN8uavtrack12_GLOBAL__N_1\d+", "", name).replace("EvNS_10StepParamsE", "")
        if flt in short:
            print(f"{short:48s} instr {len(ins):5d} vgpr {v:3d} occ {m.group(1)} scratch {sc:4d} readlane {c['v_readlane_b32']:3d} "
                  f"cmp {sum(n for k, n in c.items() if k.startswith('v_cmp')):3d} cndmask {sum(n for k, n in c.items() if k.startswith('v_cndmask')):3d} "
                  f"pk {sum(n for k, n in c.items() if k.startswith('v_pk')):3d}")
        name = None
