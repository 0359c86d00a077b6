#!/usr/bin/env python3
"""
Condense rocprofv3 output of `bench.py` into the tracked summaries of a round.

    python profiles/make_summary.py <dir with kt/ fetch/ write/ bench_default.json> <round tag>

Expects (all produced on the MI355X box, see the command block written into the .md):
  <dir>/kt/**/_kernel_stats.csv          rocprofv3 --kernel-trace --stats
  <dir>/fetch/**/_counter_collection.csv rocprofv3 --pmc FETCH_SIZE   (own pass)
  <dir>/write/**/_counter_collection.csv rocprofv3 --pmc WRITE_SIZE   (own pass)
  <dir>/bench_default.json               un-profiled `python3 bench.py`
Writes profiles/<tag>_bench_c4.json, <tag>_rocprofv3_kernel_stats_c4.csv, <tag>_pmc_c4.json/.md.
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE counts half the bytes of a
wide coalesced read (MI355X_MICROARCH.md, HBM section), so HBM traffic = 2*FETCH + WRITE.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HOT = ["k_P_tiles", "k_overlap_save", "k_os_real", "k_Pt_tiles_fixed", "k_Pt_tiles<", "k_PtNP_sell", "k_P_time",
       "k_Pt_sell", "k_Zt_partial_wide", "k_Z_axpy_wide", "k_Z_apply", "k_m2_finish_wide", "k_gemm_tn_mfma_pairs", "k_panel_gemm_mfma",
       "k_bdprecond", "k_dot_partial", "k_pcg_update_xr", "k_pcg_update_p"]


def main(src, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    bench = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
    # Which sources did the profiled kernels come from?  Every bench line carries the hash of the kernel sources
    # its library was linked from (cosmomap2_amd/build.py `source_hash`).  The summary is REFUSED unless the
    # un-profiled line and the three profiled processes (kt / fetch / write logs) all ran one build, that build
    # is the working tree's, and the working tree's kernel sources are committed -- then, and only then, `commit`
    # is a true statement about the counters.
    import subprocess
    sys.path.insert(0, os.path.dirname(here))
    from cosmomap2_amd.build import source_hash
    want = source_hash()
    seen = {"bench_default.json": (bench.get("library") or {}).get("sources_sha16_at_build")}
    for log in ("kt.log", "fetch.log", "write.log"):
        lines = [l for l in open(os.path.join(src, log)) if l.startswith('{"metric"')]
        seen[log] = (json.loads(lines[-1]).get("library") or {}).get("sources_sha16_at_build") if lines else None
    if any(v != want for v in seen.values()):
        sys.exit("make_summary: REFUSED -- the runs under %s were made with other kernel sources than the working "
                 "tree's (%s): %r" % (src, want, seen))
    dirty = subprocess.run(["git", "status", "--porcelain", "--", "cosmomap2_amd/csrc", "include"],
                           cwd=os.path.dirname(here), capture_output=True, text=True).stdout.strip()
    if dirty:
        sys.exit("make_summary: REFUSED -- uncommitted kernel sources:\n" + dirty)
    commit = subprocess.run(["git", "log", "-1", "--format=%h", "--", "cosmomap2_amd/csrc", "include"],
                            cwd=os.path.dirname(here), capture_output=True, text=True).stdout.strip()
    bench["commit"] = commit
    bench["library_sources_sha16"] = want
    os.environ["CM2_PROFILE_COMMIT"] = commit
    json.dump(bench, open(os.path.join(here, tag + "_bench_c4.json"), "w"), indent=1)
    newest = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)
    ks = newest(os.path.join(src, "kt", "**", "*_kernel_stats.csv"))
    shutil.copy(ks, os.path.join(here, tag + "_rocprofv3_kernel_stats_c4.csv"))
    rows = list(csv.DictReader(open(ks)))
    # (k_Z_apply with r = 32 and with the Arnoldi panel width differ only in arguments: the rows
    # of one kernel name are merged by rocprofv3; the deflation kernels are listed for the M2 /
    # Arnoldi legs of the bench, the first six for the timed matvec)

    def counters(kind):
        f = newest(os.path.join(src, kind, "**", "*_counter_collection.csv"))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        # bench.py launches each hot kernel many times with identical work: take the median
        return {k: sorted(v)[len(v) // 2] for k, v in agg.items()}

    fe, wr = counters("fetch"), counters("write")
    nt = bench["config"]["nt_per_gpu"]
    npix = bench["config"]["npix"]
    n = 3 * npix
    r = (bench.get("pcg") or {}).get("two_level", {}).get("rank", 32)
    zb = 8.0 * n * r
    alg = {"k_P_tiles": 28.0 * nt + 24 * npix, "k_overlap_save": 16.0 * nt, "k_os_real": 16.0 * nt,
           "k_Pt_tiles_fixed": 28.0 * nt + 24 * npix, "k_Pt_tiles<": 28.0 * nt + 24 * npix,
           "k_PtNP_sell": 28.0 * nt + 48 * npix,
           "k_P_time": 28.0 * nt + 24 * npix, "k_Pt_sell": 28.0 * nt + 24 * npix,
           "k_Zt_partial_wide": zb + 8.0 * n, "k_Z_axpy_wide": zb + 16.0 * n, "k_Z_apply": zb + 16.0 * n,
           "k_m2_finish_wide": 2 * zb + 16.0 * n + 56.0 * npix, "k_gemm_tn_mfma_pairs": 2 * zb,
           "k_panel_gemm_mfma": 8.0 * n * 32 + 2 * zb,      # one 32-column panel in, r columns updated
           "k_bdprecond": 16.0 * n + 56.0 * npix, "k_dot_partial": 16.0 * n,
           "k_pcg_update_xr": 40.0 * n, "k_pcg_update_p": 24.0 * n}
    table = {}
    for r in rows:
        for h in HOT:
            if h in r["Name"] and int(r["Calls"]) > 2:
                f = fe.get(r["Name"])
                w = wr.get(r["Name"])
                table[h] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                            "fetch_size_bytes": None if f is None else f * 1024,
                            "write_size_bytes": None if w is None else w * 1024,
                            "hbm_traffic_bytes": None if (f is None or w is None)
                            else 2 * f * 1024 + w * 1024,
                            "algorithmic_bytes": alg[h]}
    out = {"workload": bench["config"]["workload"], "nt_per_gpu": nt, "kernels": table,
           "commit": os.environ.get("CM2_PROFILE_COMMIT") or bench.get("commit"),
           "library_sources_sha16": want,
           "note": "hbm_traffic_bytes = 2*FETCH_SIZE + WRITE_SIZE per launch (gfx950 FETCH_SIZE "
                   "correction); separate --pmc passes"}
    json.dump(out, open(os.path.join(here, tag + "_pmc_c4.json"), "w"), indent=1)

    md = ["# rocprofv3 summary %s -- `python3 bench.py` (%s)" % (tag, bench["config"]["workload"]), "",
          "Commands (MI355X box, after `cd /tmp && export TMPDIR=/tmp`):", "",
          "    rocprofv3 --kernel-trace --stats -d out/kt --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-filters --no-raster",
          "    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out/fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-filters --no-raster --arnoldi-steps 40",
          "    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out/write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-filters --no-raster --arnoldi-steps 40",
          "", "Full kernel table: `%s_rocprofv3_kernel_stats_c4.csv`; machine-readable: `%s_pmc_c4.json`." % (tag, tag),
          "", "| kernel | calls | avg ms (rocprofv3) | 2*FETCH_SIZE GB | WRITE_SIZE GB | HBM traffic GB | GB/s of HBM traffic | algorithmic GB | GB/s algorithmic |",
          "|---|---|---|---|---|---|---|---|---|"]
    for h, t in table.items():
        g = lambda b: "-" if b is None else "%.2f" % (b / 1e9)
        tr = t["hbm_traffic_bytes"]
        # a rate on SURVEY's algorithmic bytes is a bandwidth only when the kernel moves at least
        # that much; the two tile kernels are built to move fewer bytes (half-angle storage)
        alg_rate = ("%.0f" % (t["algorithmic_bytes"] / 1e9 / (t["avg_ms"] * 1e-3))
                    if tr is None or tr >= 0.98 * t["algorithmic_bytes"] else "n/a (moves fewer bytes)")
        md.append("| `%s` | %d | %.4f | %s | %s | %s | %s | %.2f | %s |" % (
            h, t["calls"], t["avg_ms"], g(None if t["fetch_size_bytes"] is None else 2 * t["fetch_size_bytes"]),
            g(t["write_size_bytes"]), g(tr), "-" if tr is None else "%.0f" % (tr / 1e9 / (t["avg_ms"] * 1e-3)),
            t["algorithmic_bytes"] / 1e9, alg_rate))
    md += ["", "bench.py HIP-event averages of the same kernels in the un-profiled run (`%s_bench_c4.json`): " % tag
           + "; ".join("%s %.3f ms" % (k, v["ms"]) for k, v in bench["stages"].items()) + ".",
           "Step: %.3f ms = %.3g samples/s = %.1f %% of the 8 TB/s HBM peak on 72 B/sample + 48 B/pixel."
           % (bench["ms_per_step"], bench["value"], 100 * bench["step_frac_of_hbm_peak"])]
    # the bench line printed by the PROFILED process itself (kt.log): its event timers and rocprofv3's average
    # of the same launches see the same box in the same minute -- two processes on one box differ by 3-5 %
    try:
        inproc = [json.loads(l) for l in open(os.path.join(src, "kt.log")) if l.startswith('{"metric"')][-1]
        dom = [t for h, t in table.items() if "k_os_real" in h or "k_overlap_save" in h]
        md += ["", "Agreement check inside ONE process (the `--kernel-trace --stats` run prints its own bench line): "
               "step %.4f ms; `%s` by bench.py's HIP events %.4f ms per launch (interval between the events; "
               "%.4f net of the event gap), by rocprofv3 %.4f ms average over all %d launches of the process.  "
               "(The un-profiled line above is another process on the same box: such pairs differ by 3-5 %%.)"
               % (inproc["ms_per_step"], inproc["roofline"]["kernel"].split(" (")[0],
                  inproc["roofline"]["avg_launch_ms_event_interval"], inproc["roofline"]["avg_launch_ms"],
                  dom[0]["avg_ms"] if dom else float("nan"), dom[0]["calls"] if dom else 0)]
        summary_inproc = {"ms_per_step": inproc["ms_per_step"], "roofline": inproc["roofline"]}
    except Exception as exc:                       # noqa: BLE001
        summary_inproc = {"error": str(exc)}
    out["bench_line_of_the_profiled_process"] = summary_inproc
    json.dump(out, open(os.path.join(here, tag + "_pmc_c4.json"), "w"), indent=1)
    md += ["", "Algorithmic bytes are SURVEY 8(d)'s (P and P^T: pixel 4 + cos 8 + sin 8 + TOD 8 = 28 B per "
           "sample).  The tile plan stores a 2-byte pixel-in-tile index and, by default, one half-angle "
           "value instead of cos and sin: P is designed to move 18 B per sample, the fixed-order P^T "
           "~21.7 B (8 B TOD + padded groups of 4-byte list entries and half angles), which is why their "
           "measured traffic is below the algorithmic figure.  FETCH_SIZE is doubled as the guide "
           "prescribes for wide coalesced reads; for the overlap-save kernel (k_os_real: run-coded "
           "2-byte lists and 8-byte gathers of address runs) the doubled figure equals its designed "
           "reads (lists + windows with their 1.33 overlap).  The deflation rows use the "
           "bytes of Z / AZ (8 n r each) plus the map vectors."]
    open(os.path.join(here, tag + "_pmc_c4.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md[10:]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
