# uneven hit maps: the default plan (uniform tiles, heavy tiles shared out to several workgroups) against the
# equal-load cut of rounds 2-3 (CM2_TILE_BALANCE=cut) and against uniform tiles with one workgroup each (=0);
# two alternations, one box
set -e
for rep in 1 2; do
for mode in default cut; do
  if [ $mode = default ]; then unset CM2_TILE_BALANCE; else export CM2_TILE_BALANCE=$mode; fi
  python bench.py --no-cpu --no-filters --no-parity --no-pcg --deflation 0 > gpurun_out/r4_un_${mode}_$rep.json 2> gpurun_out/r4_un_${mode}_$rep.err
done
done
python - <<'PY'
import json
for rep in (1, 2):
    for f in ("default", "cut"):
        d = json.loads(open("gpurun_out/r4_un_%s_%d.json" % (f, rep)).read().strip().splitlines()[-1])
        u = d["uneven_hit_map"]
        print(json.dumps({"plan": f, "rep": rep, "headline_ms": round(d["ms_per_step"], 4),
                          "uniform_ms_same_method": u["uniform_ms_same_method"], "uneven_ms": u["ms_per_step"],
                          "over_uniform": u["over_uniform"], "tiles": u["tiles"],
                          "stages_ms_in_sequence": u["stages_ms_in_sequence"], "pt_parts": u.get("pt_parts"),
                          "hot_pixel_ms": u["hot_pixel"]["ms_per_step"], "hot_PT": u["hot_pixel"]["PT_ms"]["fixed_chunks"],
                          "hot_over_uniform": u["hot_pixel"]["over_uniform"], "hot_tiles": u["hot_pixel"]["tiles"],
                          "hot_parts": u["hot_pixel"]["pt_parts"]}))
PY
