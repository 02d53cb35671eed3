cd $GRAFT_REPO_ROOT
run() { echo "== $1"; env $1 python scripts/graph_gap.py 2>/dev/null | head -1; env $1 python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110; }
run "X=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "GPU_MAX_HW_QUEUES=2"
run "HSA_ENABLE_INTERRUPT=0"
run "X=2"
