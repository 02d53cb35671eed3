K3="f32x|B1|128x128x128|128x128|N128|k3x3s1u0"
K4="f32x|B1|64x64x256|64x64|N256|k3x3s1u0"
K5="f32x|B1|256x256x128|256x256|N128|k3x3s1u0"
K6="f32x|B1|128x128x256|128x128|N128|k3x3s1u0"
for h in 1 0; do
echo "halo=$h"
SGAM_F32X_HALO=$h python scripts/shape_time.py "$K5" 128,128,1 2>&1 | grep plan
SGAM_F32X_HALO=$h python scripts/shape_time.py "$K3" 128,128,1 128,128,2 64,64,1 2>&1 | grep plan
SGAM_F32X_HALO=$h python scripts/shape_time.py "$K4" 128,128,2 128,128,4 128,128,8 64,64,1 2>&1 | grep plan
SGAM_F32X_HALO=$h python scripts/shape_time.py "$K6" 128,128,2 128,128,4 2>&1 | grep plan
done
