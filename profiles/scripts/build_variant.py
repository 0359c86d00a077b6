#!/usr/bin/env python3
"""Build a VARIANT of libcosmomap2_hip.so for same-box A/B runs (loaded through CM2_LIB_PATH; DESIGN.md section 6:
candidates alternate five times in one GPU call): one translation unit is compiled from another source file and /
or with extra -D flags, the other objects are those of the shipped build.

    python3 profiles/scripts/build_variant.py NAME [--unit cm2_overlap_save] [--src FILE] [--flags "-DX=1 ..."]

-> profiles/scripts/_variants/lib_NAME.so (+ the unit's register table on stdout).  `--src` may be a file from
another commit: git show REV:cosmomap2_amd/csrc/cm2_overlap_save.hip > /tmp/x.hip
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cosmomap2_amd import build as B                     # noqa: E402
from cosmomap2_amd import kernel_resources as KR         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("--unit", default="cm2_overlap_save")
ap.add_argument("--src", default=None)
ap.add_argument("--flags", default="")
a = ap.parse_args()
B.build(verbose=False)                                   # the shipped objects must exist and be current
out_dir = os.path.join(ROOT, "profiles", "scripts", "_variants")
os.makedirs(out_dir, exist_ok=True)
src = a.src or os.path.join(B.CSRC, a.unit + ".hip")
obj = os.path.join(out_dir, "%s_%s.o" % (a.unit, a.name))
cmd = B.compile_command(src, obj)
cmd[1:1] = ["-I" + B.CSRC] + a.flags.split()
p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
if p.returncode:
    sys.exit(p.stderr[-3000:])
raw = KR.parse_remarks(p.stderr)
names = KR.demangle(list(raw))
rows = [dict(unit=a.unit, kernel=KR.short(names[k]), **v) for k, v in raw.items()]
print(KR.table([r for r in KR.own_kernels(rows) if "k_os_real" in r["kernel"] or a.unit != "cm2_overlap_save"]))
objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith(".o") and f != a.unit + ".o"] + [obj]
lib = os.path.join(out_dir, "lib_%s.so" % a.name)
subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-o", lib] + objs +
                      ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"])
print(lib)
