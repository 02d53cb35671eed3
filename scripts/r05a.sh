#!/bin/bash
# round 5, call 1: the 16-bit 128-row halo kernel — what bounds a cheaper swish (GroupNorm without swish), the 64-row tile at four
# workgroups per CU on the 256^2 layer, and the scheduling variants of this round (ablib/*: peeled last slabs + residual prefetch,
# six-deep weight ring, halo-load position, packed-fp16 swish): bit-identity, layer timings (hot / chain / cold), frame
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib
echo "== bit identity against the base build"
timeout 1500 python scripts/h16_variant_check.py $A/base/libsgam_hip.so $A/peel/libsgam_hip.so $A/nbr6/libsgam_hip.so $A/nbr6peel/libsgam_hip.so $A/lt6/libsgam_hip.so $A/lt8/libsgam_hip.so 2>&1 | tail -20
lt() { echo -n "$1 $2: "; SGAM_HIP_LIB=$A/$1/libsgam_hip.so timeout 200 python scripts/h16_layer_time.py $2 2>&1 | tail -1 | cut -c60-330; }
echo "== base: swish on/off, 128- vs 64-row tile, bf16 / fp16, B = 1 / 8"
for a in "1 bf16 256 128 128 1" "1 bf16 256 128 128 0" "1 bf16 256 128 64 1" "1 bf16 256 128 64 0" "1 fp16 256 128 128 1" "1 fp16 256 128 64 1" "8 bf16 256 128 128 1" "8 bf16 256 128 128 0" "8 bf16 256 128 64 1"; do lt base "$a"; done
echo "== variants, 256^2 x 128, B = 1"
for v in base peel nbr6 nbr6peel lt6 lt8 sw1 peelsw1 base; do lt $v "1 bf16 256 128 128 1"; done
for v in base peel nbr6 sw1 peelsw1; do lt $v "1 bf16 256 128 64 1"; done
for v in base peel sw1 peelsw1; do lt $v "1 fp16 256 128 128 1"; done
echo "== variants, B = 8"
for v in base peel nbr6peel sw1 peelsw1; do lt $v "8 bf16 256 128 128 1"; done
echo "== variants, small maps (64-row tile): 128^2 x 128, 64^2 x 256"
for v in base peel nbr6 sw1 peelsw1; do lt $v "1 bf16 128 128 64 1"; lt $v "1 bf16 64 256 64 1"; done
echo "== frame (bf16)"
for v in base peel sw1 peelsw1 base peel; do echo -n "$v: "; SGAM_HIP_LIB=$A/$v/libsgam_hip.so timeout 300 python scripts/h16_frame.py bf16 2>&1 | tail -9 | head -4; done
