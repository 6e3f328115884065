#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into profiles/:
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_pmc.json             per-kernel average counter values per dispatch
  profiles/<tag>_summary.md           the table the DESIGN/bench numbers cite
HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced reads -- so the read side is reported both raw and doubled."""
import csv
import re
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/prof_{tag}"
key = sys.argv[3] if len(sys.argv) > 3 else None     # e.g. 4096x20x10_T200 -> profiles/traffic.json
os.makedirs("profiles", exist_ok=True)

# (gpurun merges every call's files into the same directory: take the newest file of each pass)
newest = lambda pat: sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1:]
stats = newest(f"{src}/trace/**/*_kernel_stats.csv")
kern = {}
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
    for row in csv.DictReader(open(stats[0])):
        kern[row["Name"]] = dict(calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]),
                                 min_ns=float(row["MinNs"]), max_ns=float(row["MaxNs"]), pct=float(row["Percentage"]))

pmc = defaultdict(lambda: defaultdict(list))
meta = {}
for f in [g for d in sorted(glob.glob(f"{src}/pmc_*")) if os.path.isdir(d) for g in newest(f"{d}/**/*_counter_collection.csv")]:
    per_dispatch = defaultdict(float)
    names = {}
    for row in csv.DictReader(open(f)):
        k = (row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[k] += float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = row["Kernel_Name"]
        meta[row["Kernel_Name"]] = dict(vgpr=int(row["VGPR_Count"]), sgpr=int(row["SGPR_Count"]),
                                        lds=int(row["LDS_Block_Size"]), scratch=int(row["Scratch_Size"]),
                                        wg=int(row["Workgroup_Size"]), grid=int(row["Grid_Size"]))
    for (d, c), v in per_dispatch.items():
        pmc[names[d]][c].append(v)

out = {}
for k, cs in pmc.items():
    if "rollout_kernel" not in k and "reset_kernel" not in k and "pmi" not in k:
        continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["_dispatches"] = max(len(v) for v in cs.values())
    out[k]["_launch"] = meta.get(k, {})
    if k in kern:
        out[k]["_avg_ns"] = kern[k]["avg_ns"]
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1, sort_keys=True)

# MAAC-R issues TWO pmi_score* kernels per chunk: the scorer that does the work and the gated stand-by behind it, which
# returns at once (18 KiB fetched).  The traffic record of "the scorer" is the one that moved the bytes.
def _traffic(c):
    return 2 * c.get("FETCH_SIZE", 0.0) * 1024 + c.get("WRITE_SIZE", 0.0) * 1024
scorers = [k for k in out if "pmi_score" in k and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]]
working_scorer = max(scorers, key=lambda k: _traffic(out[k])) if scorers else None

lines = [f"# rocprofv3 summary `{tag}`", "", "## kernel trace (`--kernel-trace --stats`)", "",
         "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["pct"])[:8]:
    lines.append(f"| `{k[:90]}` | {v['calls']} | {v['avg_ns']/1e3:.1f} | {v['min_ns']/1e3:.1f} | {v['max_ns']/1e3:.1f} | {v['pct']:.2f} |")
for k, c in out.items():
    if "rollout_kernel" not in k and "pmi" not in k:
        continue
    lines += ["", f"## counters, average per dispatch: `{k[:100]}`", "", f"launch: {c['_launch']}", ""]
    avg_ns = c.get("_avg_ns")
    fetch = c.get("FETCH_SIZE")
    write = c.get("WRITE_SIZE")
    if fetch is not None and write is not None:
        rb, wb = fetch * 1024, write * 1024
        lines += [f"- FETCH_SIZE {fetch:.0f} KiB -> {rb/1e6:.1f} MB raw, {2*rb/1e6:.1f} MB with the gfx950 x2 correction for wide coalesced reads",
                  f"- WRITE_SIZE {write:.0f} KiB -> {wb/1e6:.1f} MB",
                  f"- HBM traffic per launch: {(rb+wb)/1e6:.1f} MB raw / {(2*rb+wb)/1e6:.1f} MB corrected"]
        if avg_ns:
            lines.append(f"- at {avg_ns/1e3:.1f} us per launch: {(2*rb+wb)/avg_ns:.1f} GB/s of HBM traffic (corrected)")
        if key and ("rollout_kernel" in k or k == working_scorer):
            tf = "profiles/traffic.json"
            tr = json.load(open(tf)) if os.path.exists(tf) else {}
            tr[key + ("_scorer" if "pmi_score" in k else "")] = dict(hbm_bytes_per_launch=2 * rb + wb, fetch_bytes_raw=rb, write_bytes=wb,
                           kernel=(re.search(r"(\w+_kernel(?:<[^>]*>)?)", k) or [None, k[:60]])[1],
                           source=f"profiles/{tag}_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)")
            json.dump(tr, open(tf, "w"), indent=1, sort_keys=True)
        elif key and "pmi_score" in k:
            lines.append("- (the gated stand-by scorer: not the traffic record of this configuration)")
    w = c.get("SQ_WAVES")
    if w:
        lines.append(f"- waves {w:.0f}; per wave: VALU {c.get('SQ_INSTS_VALU',0)/w:.0f}, SALU {c.get('SQ_INSTS_SALU',0)/w:.0f}, "
                     f"LDS {c.get('SQ_INSTS_LDS',0)/w:.0f} instructions")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        lines.append(f"- SQ_WAVE_CYCLES {wc:.3g}; WAIT_ANY {c.get('SQ_WAIT_ANY',0)/wc:.2%}, WAIT_INST_ANY {c.get('SQ_WAIT_INST_ANY',0)/wc:.2%} of wave cycles; "
                     f"SQ_BUSY_CYCLES {c.get('SQ_BUSY_CYCLES',0):.3g}")
    if "SQ_ACTIVE_INST_VALU" in c:
        lines.append(f"- ACTIVE_INST_VALU {c['SQ_ACTIVE_INST_VALU']:.3g}, ACTIVE_INST_LDS {c.get('SQ_ACTIVE_INST_LDS',0):.3g}, ACTIVE_INST_ANY {c.get('SQ_ACTIVE_INST_ANY',0):.3g}, "
                     f"WAIT_INST_LDS {c.get('SQ_WAIT_INST_LDS',0):.3g}, LDS_BANK_CONFLICT {c.get('SQ_LDS_BANK_CONFLICT',0):.3g} / LDS_IDX_ACTIVE {c.get('SQ_LDS_IDX_ACTIVE',0):.3g}")
    if c.get("SQ_INSTS_MFMA"):
        lines.append(f"- MFMA: {c['SQ_INSTS_MFMA']:.4g} instructions, SQ_VALU_MFMA_BUSY_CYCLES {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.4g} "
                     f"({c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/c['SQ_INSTS_MFMA']:.1f} busy cycles per MFMA)")
    lines.append("")
    lines.append("all counters: " + ", ".join(f"{n}={v:.4g}" for n, v in sorted(c.items()) if not n.startswith("_")))
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
