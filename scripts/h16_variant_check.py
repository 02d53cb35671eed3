#!/usr/bin/env python
"""Two builds of libsgam_hip.so on the cases of h16_pc_check.py (the 16-bit halo kernel: plain / GroupNorm / swish / residual / fp32
output, 128- and 64-row tiles, whole K and split-K): outputs and GroupNorm chunk statistics must be BIT-IDENTICAL — the acceptance
test of every scheduling variant of the kernel (ring depth, peeled slabs, residual prefetch ...).
    python scripts/h16_variant_check.py <lib A> <lib B> [more libs ...]      (each compared with lib A)"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PLANS = [("128", "1"), ("64", "1"), ("64", "2")]


def dump(lib, bm, ks):
    r = subprocess.run(["timeout", "300", sys.executable, os.path.join(HERE, "h16_pc_check.py"), "dump"], capture_output=True, text=True,
                       env=dict(os.environ, SGAM_HIP_LIB=os.path.abspath(lib), SGAM_HPC="0", PC_BM=bm, PC_KS=ks))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DUMP ")]
    if r.returncode != 0 or not line:
        print(f"{lib} bm={bm} ks={ks}: rc {r.returncode}\n{r.stderr[-1200:]}")
        return None
    return json.loads(line[0][5:])


if __name__ == "__main__":
    libs = sys.argv[1:]
    rc = 0
    for bm, ks in PLANS:
        ref = dump(libs[0], bm, ks)
        for lib in libs[1:]:
            got = dump(lib, bm, ks)
            if ref is None or got is None:
                rc = 1
                continue
            bad = [k for k in ref if ref[k][:2] != got[k][:2]]
            print(f"plan bm={bm} ksplit={ks}: {lib}: " + ("BIT-IDENTICAL" if not bad else f"MISMATCH in cases {bad} "
                  + str([(ref[k], got[k]) for k in bad])), flush=True)
            rc = rc or (2 if bad else 0)
    sys.exit(rc)
