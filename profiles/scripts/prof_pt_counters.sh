cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_prof1
mkdir -p $O
export CM2_TILE_PIXELS=1536
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pcg --no-filters --no-raster --deflation 0"
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d $O/lds --output-format csv -- $B > $O/lds.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace -d $O/wait --output-format csv -- $B > $O/wait.log 2>&1 || exit 1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace -d $O/valu --output-format csv -- $B > $O/valu.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM --kernel-trace -d $O/vmem --output-format csv -- $B > $O/vmem.log 2>&1 || true
cd $O && find . -name "*counter_collection.csv" | head
