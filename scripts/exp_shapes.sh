K1="f32x|B1|16x16x512|16x16|N512|k3x3s1u0"
K2="f32x|B1|32x32x256|32x32|N256|k3x3s1u0"
K3="f32x|B1|128x128x128|128x128|N128|k3x3s1u0"
K4="f32x|B1|64x64x256|64x64|N256|k3x3s1u0"
K5="f32x|B1|256x256x128|256x256|N128|k3x3s1u0"
python scripts/shape_time.py "$K1" 64,64,16 64,64,8 64,64,4 64,64,1 2>&1 | grep plan
python scripts/shape_time.py "$K2" 64,64,8 64,64,4 64,64,2 2>&1 | grep plan
python scripts/shape_time.py "$K3" 64,64,1 128,128,1 64,128,1 2>&1 | grep plan
python scripts/shape_time.py "$K4" 64,128,4 64,64,2 64,64,4 64,64,1 2>&1 | grep plan
python scripts/shape_time.py "$K5" 128,128,1 2>&1 | grep plan
