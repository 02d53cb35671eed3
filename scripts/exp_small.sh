# isolated (weights hot) timing of the small-map shapes, with ablations of the halo kernel
S16="f32x|B1|16x16x512|16x16|N512|k3x3s1u0"
S32="f32x|B1|32x32x256|32x32|N256|k3x3s1u0"
S64="f32x|B1|64x64x256|64x64|N256|k3x3s1u0"
for ab in 0 1 24; do
  SGAM_XABLATE=$ab python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
  echo "== ablate $ab"
  python scripts/shape_time.py "$S16" 64,128,16 64,128,8 64,128,4 2>&1 | grep plan
  python scripts/shape_time.py "$S32" 64,128,8 64,128,4 2>&1 | grep plan
  python scripts/shape_time.py "$S64" 64,128,2 64,128,1 2>&1 | grep plan
done
python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
