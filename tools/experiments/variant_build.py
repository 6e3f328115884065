#!/usr/bin/env python3
"""Experiment aid (build container): build_variants/NAME.so from a PATCHED COPY of csrc/step_kernel.hip -- csrc itself carries no
experiment switches.  usage: variant_build.py NAME "old=>new" ["FILE::old=>new" ...]   (every `old` must occur in its file; default file step_kernel.hip)
A/B the result on the GPU box with tools/sweep.py --lib build_variants/NAME.so."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "marl-uavs-targets-tracking_amd", "csrc")
name = sys.argv[1]
subs = [a.split("=>", 1) for a in sys.argv[2:]]
dst = os.path.join(ROOT, "build_variants", name + "_src")
os.makedirs(dst, exist_ok=True)
for f in os.listdir(src):
    if f.endswith((".h", ".hip")):
        open(os.path.join(dst, f), "w").write(open(os.path.join(src, f)).read())
p = os.path.join(dst, "step_kernel.hip")
for old, new in subs:        # "old=>new" patches step_kernel.hip; "FILE::old=>new" patches another file of the copy (e.g. actor.h)
    f = p
    if "::" in old.split("\n")[0] and old.split("::", 1)[0].endswith((".h", ".hip")):
        fname, old = old.split("::", 1)
        f = os.path.join(dst, fname)
    s = open(f).read()
    assert s.count(old) >= 1, old
    open(f, "w").write(s.replace(old, new))
out = os.path.join(ROOT, "build_variants", "obj_" + name)
os.makedirs(out, exist_ok=True)
flags = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -I%s/include -I%s -Wno-unused-function -Wno-pass-failed "
         "-fno-convergent-functions -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-mfma-vgpr-form=1" % (ROOT, dst)).split()
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", p, "-o", os.path.join(out, "step_kernel.o")], check=True)
objs = [os.path.join(src, "build", f + ".o") for f in ("api", "reset_kernel", "pmi_kernel", "policy_kernel")] + [os.path.join(out, "step_kernel.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "build_variants", name + ".so")] + objs, check=True)
print("built build_variants/%s.so" % name)
