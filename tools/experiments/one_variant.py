#!/usr/bin/env python3
"""Experiment aid (build container): compile ONE rollout_kernel instantiation to assembly in seconds instead of the whole
step_kernel.hip in minutes.  usage: one_variant.py "20, 10, 0, false, 2, false, false, true" [out.s ["old=>new" ...]]
Works on a patched COPY of csrc/step_kernel.hip whose pick_kernel() names only that variant; prints registers / scratch."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "marl-uavs-targets-tracking_amd", "csrc")
args = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/asm/one.s"
subs = [a.split("=>", 1) for a in sys.argv[3:]]          # textual patches "old=>new" applied to the copy (experiments)
os.makedirs(os.path.dirname(out), exist_ok=True)
s = open(os.path.join(src, "step_kernel.hip")).read()
for old, new in subs:
    assert s.count(old) >= 1, old
    s = s.replace(old, new)
a = s.index("KernelFn pick_kernel(int N, int M, int mode, bool z3, int *specialised")
b = s.index("}  // namespace\n\nGeometry plan_geometry")
s = s[:a] + ("KernelFn pick_kernel(int N, int M, int mode, bool z3, int *specialised, int policy = kPolicyGiven, bool allout = false,\n"
             "                     bool extras = false, bool lone = false)\n{\n    *specialised = 1;\n    return rollout_kernel<%s>;\n}\n\n" % args) + s[b:]
tmp = os.path.join(os.path.dirname(out), "one_variant.hip")
open(tmp, "w").write(s)
flags = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -I%s/include -I%s -Wno-unused-function -Wno-pass-failed "
         "-fno-convergent-functions -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-mfma-vgpr-form=1 "
         "-S --cuda-device-only" % (ROOT, src)).split()
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + [tmp, "-o", out], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "NumSgprs"):
    m = re.search(r"; %s: (\d+)" % k, txt)
    print(k, m.group(1) if m else "?", end="  ")
print()
