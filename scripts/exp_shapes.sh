K5="f32x|B1|256x256x128|256x256|N128|k3x3s1u0"
K3="f32x|B1|128x128x128|128x128|N128|k3x3s1u0"
K4="f32x|B1|64x64x256|64x64|N256|k3x3s1u0"
python scripts/shape_time.py "$K5" 128,128,1 64,128,1 2>&1 | grep plan
python scripts/shape_time.py "$K3" 128,128,1 64,128,1 64,128,2 2>&1 | grep plan
python scripts/shape_time.py "$K4" 64,128,1 64,128,2 64,128,4 128,128,2 128,128,4 2>&1 | grep plan
