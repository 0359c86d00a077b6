"""
Build libcosmomap2_hip.so (gfx950) in-tree with hipcc.

    python -m cosmomap2_amd.build [--force]

One object per .hip translation unit under cosmomap2_amd/csrc (compiled in
parallel, rebuilt only when the source or a header is newer), linked into
cosmomap2_amd/libcosmomap2_hip.so.  -ffp-contract=off: the kernels keep the
reference's IEEE operation order so that results can be compared bit for bit with
the CPU oracle; every kernel on this path is HBM-bound, so FMA would buy nothing.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libcosmomap2_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-munsafe-fp-atomics"]


# translation units whose results are not compared bit for bit (FFT butterflies): FMA allowed
FMA_OK = {"cm2_fft.hip", "cm2_fft_real.hip"}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            flags = list(FLAGS) + os.environ.get("CM2_EXTRA_HIPCC_FLAGS", "").split()
            if os.path.basename(s) in FMA_OK:
                flags[flags.index("-ffp-contract=off")] = "-ffp-contract=fast"
            jobs.append([HIPCC] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-o", LIB] + objs +
            ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
