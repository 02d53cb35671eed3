cd $GRAFT_REPO_ROOT
python scripts/cold_time.py "f32x|B1|16x16x512|16x16|N512|k3x3s1u0" 64,128,16 32,32,1 2>&1 | grep plan
python scripts/cold_time.py "f32x|B1|32x32x256|32x32|N256|k3x3s1u0" 64,128,8 32,32,1 2>&1 | grep plan
python scripts/cold_time.py "f32x|B1|64x64x256|64x64|N256|k3x3s1u0" 64,128,2 2>&1 | grep plan
python scripts/cold_time.py "f32x|B1|128x128x128|128x128|N128|k3x3s1u0" 64,128,1 2>&1 | grep plan
python scripts/cold_time.py "f32x|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>&1 | grep plan
