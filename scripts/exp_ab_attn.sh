#!/bin/bash
# same-box A/B of attention kernel build variants: VARIANTS="name:ENV=val,... ..." scripts/exp_ab_attn.sh
cd $GRAFT_REPO_ROOT
VARIANTS=${VARIANTS:-"stage1: dma:SGAM_ATTN_STAGE=0"}
for v in $VARIANTS; do
  name=${v%%:*}; envs=${v#*:}
  mkdir -p /tmp/ab/$name
  ( IFS=,; for e in $envs; do [ -n "$e" ] && export "$e"; done; SGAM_LIB_DIR=/tmp/ab/$name python -m sgam_neurips22_amd.build 2>&1 | grep -E "error" )
done
for rep in 1 2; do
  for v in $VARIANTS; do
    name=${v%%:*}
    echo "== $name (rep $rep)"
    export SGAM_HIP_LIB=/tmp/ab/$name/libsgam_hip.so
    python scripts/attn_time.py 4096 fused | tail -1
    [ -n "$BENCH" ] && python bench.py --steps 40 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110
  done
done
