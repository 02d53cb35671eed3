K5="f32x|B1|256x256x128|256x256|N128|k3x3s1u0"
K3="f32x|B1|128x128x128|128x128|N128|k3x3s1u0"
K1="f32x|B1|16x16x512|16x16|N512|k3x3s1u0"
K7="f32x|B1|1x4096x256|1x4096|N4096|k1x1s1u0"
for sb in 0 1; do
  SGAM_XSB=$sb python -m sgam_neurips22_amd.build > /dev/null 2>&1
  echo "== sched_barrier $sb"
  python scripts/shape_time.py "$K5" 128,128,1 2>&1 | grep plan
  python scripts/shape_time.py "$K3" 128,128,1 64,64,1 2>&1 | grep plan
  python scripts/shape_time.py "$K1" 64,64,8 2>&1 | grep plan
  python scripts/shape_time.py "$K7" 128,128,1 2>&1 | grep plan
done
