K3="f32x|B1|128x128x128|128x128|N128|k3x3s1u0"
for ab in 0 1 24 21 25; do
  SGAM_XABLATE=$ab python -m sgam_neurips22_amd.build 2>&1 | grep -E "error" 
  echo "== ablate $ab"
  python scripts/shape_time.py "$K3" 64,128,1 128,128,1 2>&1 | grep plan
done
python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
