run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --windows 2 2>&1 | tail -1 | cut -c60-150; }
run A=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_USE_FGS_KERNARG=0
run GPU_FLUSH_ON_EXECUTION=0
run A=1
