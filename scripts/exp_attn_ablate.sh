for ab in ${ABLATES:-0 1 2 3 7 11 15 16}; do
  SGAM_ATTN_ABLATE=$ab python -m sgam_neurips22_amd.build 2>&1 | grep -E " error"
  echo "== ablate $ab: $(python scripts/attn_time.py 4096 fused 2>&1 | grep fused)"
done
python -m sgam_neurips22_amd.build 2>&1 | grep -E " error"
true
