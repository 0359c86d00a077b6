"""
Per-kernel register / scratch / LDS table of the gfx950 code objects, from hipcc's own
`-Rpass-analysis=kernel-resource-usage` remarks (emitted while the unit is compiled, so the table
describes exactly the object that ships).

    python -m cosmomap2_amd.kernel_resources            # table of the last build
    python -m cosmomap2_amd.kernel_resources cm2_overlap_save.hip   # compile one unit, print its table

cosmomap2_amd.build stores the remarks of every unit it compiles under csrc/build/<unit>.resources.json,
writes the merged table to profiles/kernel_resources.md, and FAILS when a kernel that matches
NO_SPILL (the kernels the default dispatch can choose) has spilled registers or uses scratch.
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "csrc", "build")
CXXFILT = "c++filt"

# kernels that must not spill: every instantiation of the hot-path kernels the dispatchers can select
NO_SPILL = [r"\bk_os_real<", r"\bk_P_tiles", r"\bk_Pt_tiles", r"\bk_Pt_hot", r"\bk_PtNP_sell", r"\bk_Pt_sell",
            r"\bk_P_time", r"\bk_Zt_partial_wide", r"\bk_m2_finish_wide", r"\bk_gemm_tn_mfma", r"\bk_panel_gemm_mfma",
            r"\bk_Z_axpy_wide", r"\bk_filter_windows"]

# kernels that must not spill SGPRs to VGPR lanes either: the overlap-save instantiations the default dispatch
# reaches below 4 GB of TOD (at 246-254 of 256 VGPRs a lane register spent on spilled scalars is one too many);
# the flat-addressing forms of the plain and run-coded lists (<32, 1, false>, <32, 2, false>) keep 44 / 56
NO_SGPR_SPILL = [r"\bk_os_real<32, 0, false>", r"\bk_os_real<32, [123], true>", r"\bk_os_real<32, 3, false>"]

_FIELDS = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
           "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
           "LDS Size [bytes/block]": "lds_bytes"}


def parse_remarks(text):
    """{mangled kernel name: {vgpr, vgpr_spill, ...}} from the compiler's stderr."""
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", ln)
        if m and cur is not None and m.group(1).strip() in _FIELDS:
            cur[_FIELDS[m.group(1).strip()]] = int(m.group(2))
    return out


def demangle(names):
    try:
        res = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, res.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def short(name):
    """`(anonymous namespace)::k_os_real<32, 2, true>(args...)` -> `k_os_real<32, 2, true>`"""
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, cut = 0, len(name)
    for i, c in enumerate(name):
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            cut = i
            break
    return name[:cut]


def load_all():
    rows = []
    if not os.path.isdir(OBJ):
        return rows
    for f in sorted(os.listdir(OBJ)):
        if f.endswith(".resources.json"):
            unit = f[:-len(".resources.json")]
            for k, v in json.load(open(os.path.join(OBJ, f))).items():
                rows.append(dict(unit=unit, kernel=k, **v))
    return rows


def own_kernels(rows):
    """Rows of this repo's kernels (library templates of hipCUB / rocPRIM left out)."""
    return [r for r in rows if "rocprim" not in r["kernel"] and "hipcub" not in r["kernel"]]


def offenders(rows):
    bad = []
    for r in rows:
        if any(re.search(p, r["kernel"]) for p in NO_SPILL):
            if r.get("vgpr_spill", 0) or r.get("scratch_bytes_per_lane", 0):
                bad.append(r)
                continue
        if any(re.search(p, r["kernel"]) for p in NO_SGPR_SPILL) and r.get("sgpr_spill", 0):
            bad.append(r)
    return bad


def table(rows):
    lines = ["| unit | kernel | VGPRs | AGPRs | spilled VGPRs | spilled SGPRs | scratch B/lane | LDS B/block (static) | waves/SIMD |",
             "|---|---|---|---|---|---|---|---|---|"]
    for r in sorted(rows, key=lambda r: (r["unit"], r["kernel"])):
        lines.append("| %s | `%s` | %d | %d | %d | %d | %d | %d | %d |" % (
            r["unit"], r["kernel"], r.get("vgpr", 0), r.get("agpr", 0), r.get("vgpr_spill", 0),
            r.get("sgpr_spill", 0), r.get("scratch_bytes_per_lane", 0), r.get("lds_bytes", 0),
            r.get("occupancy", 0)))
    return "\n".join(lines)


def store(unit, stderr_text):
    """Called by build.py after compiling `unit` (.hip file name)."""
    raw = parse_remarks(stderr_text)
    names = demangle(list(raw))
    res = {}
    for k, v in raw.items():
        res[short(names[k])] = v
    os.makedirs(OBJ, exist_ok=True)
    json.dump(res, open(os.path.join(OBJ, unit[:-4] + ".resources.json"), "w"), indent=1, sort_keys=True)
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1:
        from cosmomap2_amd import build as B
        unit = sys.argv[1]
        cmd = B.compile_command(os.path.join(B.CSRC, unit), "/dev/null")
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        raw = parse_remarks(err)
        names = demangle(list(raw))
        rows = [dict(unit=unit[:-4], kernel=short(names[k]), **v) for k, v in raw.items()]
        errs = [ln for ln in err.splitlines() if "error" in ln]
        print("\n".join(errs))
    else:
        rows = load_all()
    print(table(own_kernels(rows)))
    bad = offenders(rows)
    if bad:
        print("\nSPILLING default-path kernels: " + ", ".join(r["kernel"] for r in bad))
        sys.exit(1)
