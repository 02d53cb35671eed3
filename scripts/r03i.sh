#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout=600 -k "weight_stationary or splitk_combine or small_map" > gpurun_out/pytest_r03i.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_r03i.log | cut -c1-220
for k in "f32x|B1|16x16x512|16x16|N512|k3x3s1u0" "f32x|B1|16x16x256|16x16|N512|k3x3s1u0" "f32x|B1|16x16x512|16x16|N256|k3x3s1u0"; do
  python scripts/cold_time.py "$k" 64,128,16 64,128,8 256,32,16 256,32,8 256,32,4 2>/dev/null | grep plan | cut -c1-260
done
for k in "f32x|B1|32x32x256|32x32|N256|k3x3s1u0" "f32x|B1|32x32x512|32x32|N256|k3x3s1u0"; do
  python scripts/cold_time.py "$k" 64,128,8 64,128,4 256,32,8 256,32,4 256,32,2 2>/dev/null | grep plan | cut -c1-260
done
