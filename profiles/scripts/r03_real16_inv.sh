for v in "real32 rc" "real16 rc" "real16 inv" "real32 inv"; do set -- $v; echo "# $v"; CM2_OS_KERNEL=$1 CM2_OS_LISTS=$2 python profiles/scripts/uneven_probe.py uniform uneven 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('   ', r['hit_map'], r['tiles'], r['N^-1'], r['os']['os_lists'], r['os']['tile_bytes_per_sample'])"; done
