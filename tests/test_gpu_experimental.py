"""Opt-in kernels that are selected per PROCESS by an environment switch (README "Opt-in switches"): each must stay bit-identical
to the default path it replaces.  `SGAM_TEST_EXPERIMENTAL=1 pytest -m experimental` runs them; the driver's `-m gpu` pass skips
them (conftest.py).  The switches that can be flipped inside a process have their tests next to the default kernels:
SGAM_STATS_ACC / SGAM_XFIXUP -> test_gpu_fixup.py, test_gpu_vqgan.py::test_full_model_parity_with_launch_free_folds;
SGAM_SPLAT_TILED -> test_gpu_warp.py (tiled / auto / two streams)."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.experimental]
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_persistent_producer_consumer_kernel_is_bit_identical_to_the_one_role_kernel():
    """SGAM_HPC=1 (conv3x3_h16_pc_kernel, csrc/h16_halo.hip): the 128-row 16-bit halo tile as a persistent producer / consumer
    workgroup.  scripts/h16_pc_check.py runs eight cases (256^2 x 128 at B = 1, 64^2 x 128 at B = 8, ragged 24 x 32 maps at B = 5,
    Cin = 32 / 128 / 256, N = 128 / 256, bf16 and fp16, GroupNorm with and without swish, residual, fp32 output) once per kernel in
    separate processes and compares SHA-256 digests of the output tensor and of the GroupNorm chunk statistics."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "h16_pc_check.py")], capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k != "SGAM_HPC"})
    assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_flash_h16_with_256_query_workgroups_is_bit_identical():
    """SGAM_ATTN_H8=1 (attn_flash_h16_kernel<HT, 8>, csrc/attention.hip): eight wavefronts share ONE K / V stream — half the LDS-DMA
    pieces per wavefront and key — on a grid of half as many workgroups.  Same per-wavefront arithmetic over the same key ranges:
    scripts/attn_h8_check.py compares SHA-256 digests of the outputs of both forms (bf16 and fp16; B = 1 / 2 at n = 4096, n = 1024)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "attn_h8_check.py")], capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k != "SGAM_ATTN_H8"})
    assert r.returncode == 0 and "BIT-IDENTICAL" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
