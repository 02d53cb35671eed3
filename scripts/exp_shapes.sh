U1="f32x|B1|128x128x128|256x256|N128|k3x3s1u1"
U2="f32x|B1|64x64x256|128x128|N256|k3x3s1u1"
U3="f32x|B1|32x32x256|64x64|N256|k3x3s1u1"
U4="f32x|B1|16x16x512|32x32|N512|k3x3s1u1"
for h in 2 0; do
echo "halo=$h"
SGAM_F32X_HALO=$h python scripts/shape_time.py "$U1" 128,128,1 64,128,1 2>&1 | grep plan
SGAM_F32X_HALO=$h python scripts/shape_time.py "$U2" 128,128,1 128,128,2 64,128,1 64,128,2 2>&1 | grep plan
SGAM_F32X_HALO=$h python scripts/shape_time.py "$U3" 64,128,1 64,128,2 64,128,4 64,64,1 2>&1 | grep plan
SGAM_F32X_HALO=$h python scripts/shape_time.py "$U4" 64,128,4 64,128,8 64,64,2 2>&1 | grep plan
done
