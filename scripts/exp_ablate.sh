K5="f32x|B1|256x256x128|256x256|N128|k3x3s1u0"
for ab in 0 1 21 24 25; do
  SGAM_XABLATE=$ab python -m sgam_neurips22_amd.build 2>&1 | grep -E "error" 
  echo "== ablate $ab"
  python scripts/shape_time.py "$K5" 128,128,1 2>&1 | grep plan
done
