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
FMA_OK = {"cm2_overlap_save.hip"}


STAMP = os.path.join(HERE, "libcosmomap2_hip.sources.sha16")


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and headers the library is built from, in name
    order.  Written next to the library at link time and printed by bench.py beside the hash of the sources
    present at run time: a profile or a bench line says which sources its kernels came from even where there
    is no .git (the GPU boxes), and profiles/make_summary.py refuses counters taken from another build."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                   glob.glob(os.path.join(HERE, "..", "include", "*.h")), key=os.path.basename)
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def library_stamp():
    """{"at_build": hash stored when the library was linked (None: no stamp), "now": hash of the sources present}"""
    at_build = open(STAMP).read().strip() if os.path.exists(STAMP) else None
    if os.environ.get("CM2_LIB_PATH"):                     # a variant build loaded for an A/B run: no stamp
        at_build = "variant:" + os.path.basename(os.environ["CM2_LIB_PATH"])
    return {"sources_sha16_at_build": at_build, "sources_sha16_now": source_hash(),
            "library_matches_sources": at_build is not None and at_build == source_hash()}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def compile_command(src, obj):
    """hipcc line of one translation unit (with the per-kernel resource remarks switched on)."""
    flags = list(FLAGS) + os.environ.get("CM2_EXTRA_HIPCC_FLAGS", "").split()
    if os.path.basename(src) in FMA_OK:
        flags[flags.index("-ffp-contract=off")] = "-ffp-contract=fast"
    return [HIPCC] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]


def build(force=False, verbose=True, allow_spills=None):
    """Compile what is out of date, link, write the per-kernel register table of the shipped objects
    to profiles/kernel_resources.md and raise when a default-path kernel spills registers
    (kernel_resources.NO_SPILL; CM2_ALLOW_SPILLS=1 turns the failure into a warning for experiments)."""
    from cosmomap2_amd import kernel_resources as KR
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        res = os.path.join(OBJ, os.path.basename(s)[:-4] + ".resources.json")
        if force or _newer(o, [s] + hdrs) or not os.path.exists(res):
            jobs.append((s, compile_command(s, o)))
    # objects of translation units that no longer exist must not be linked
    for f in os.listdir(OBJ):
        stem = f.split(".")[0]
        if (f.endswith(".o") or f.endswith(".resources.json")) and \
                not os.path.exists(os.path.join(CSRC, stem + ".hip")):
            os.remove(os.path.join(OBJ, f))

    def run(job):
        src, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        rest = [ln for ln in p.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln]
        if rest and (verbose or p.returncode):
            print("\n".join(rest), file=sys.stderr, flush=True)
        if p.returncode:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        KR.store(os.path.basename(src), p.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-o", LIB] + objs + \
            ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(STAMP, "w") as fh:
            fh.write(source_hash() + "\n")
    elif not os.path.exists(STAMP):
        with open(STAMP, "w") as fh:                       # (a library linked before the stamp existed, up to date)
            fh.write(source_hash() + "\n")
    rows = KR.load_all()
    if jobs:
        prof = os.path.join(HERE, "..", "profiles")
        if os.path.isdir(prof):
            with open(os.path.join(prof, "kernel_resources.md"), "w") as fh:
                fh.write("# Register / scratch / LDS use of every kernel in libcosmomap2_hip.so\n\n"
                         "Written by `cosmomap2_amd/build.py` from hipcc's `-Rpass-analysis=kernel-resource-usage` "
                         "remarks of the compile that produced the shipped objects (gfx950). The build fails "
                         "when a kernel matching `kernel_resources.NO_SPILL` spills.\n\n")
                fh.write(KR.table(KR.own_kernels(rows)) + "\n")
    bad = KR.offenders(rows)
    if bad:
        msg = "default-path kernels spill registers: " + ", ".join(
            "%s (%d VGPRs, %d SGPRs spilled, %d B/lane scratch)" % (
                r["kernel"], r.get("vgpr_spill", 0), r.get("sgpr_spill", 0),
                r.get("scratch_bytes_per_lane", 0)) for r in bad)
        if allow_spills if allow_spills is not None else os.environ.get("CM2_ALLOW_SPILLS"):
            print("WARNING: " + msg, file=sys.stderr)
        else:
            raise RuntimeError(msg)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
