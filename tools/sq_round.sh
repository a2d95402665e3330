# SQ counters (LDS bank conflicts, waits, MFMA busy) of the train step: cfg2 (fused path) and cfg4 (long memories: stand-alone attention)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
for w in cfg2 cfg3 cfg4; do
  timeout 400 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/sq_$w -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --windows 0 --steps 2 --warmup 1 > /tmp/sq_$w.log 2>&1
  python $R/tools/sq_summary.py /tmp/sq_$w "rocprofv3 --pmc $PMC --kernel-trace -- python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --windows 0" > $R/gpurun_out/${TAG:-r06}_sq_counters_$w.txt
done
