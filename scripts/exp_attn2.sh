for cfg in "2 0" "1 0" "2 1" "1 1"; do
  set -- $cfg
  SGAM_ATTN_SACC=$1 SGAM_ATTN_DMA_SPREAD=$2 python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
  echo "== sacc $1 spread $2: $(python scripts/attn_time.py 4096 fused 2>&1 | grep fused)"
done
python -m sgam_neurips22_amd.build 2>&1 | grep -E " error"
python -m pytest tests/test_gpu_ops.py -x -q -k fused_attention 2>&1 | tail -1
