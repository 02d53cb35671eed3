"""16-bit fused attention with 256-query workgroups (SGAM_ATTN_H8=1) against the default 128-query form: one process per form, SHA-256 of
the outputs.  The two forms run the same per-wavefront arithmetic over the same key ranges, so the outputs must be bit-identical."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from sgam_neurips22_amd import ops, testing
    h = hashlib.sha256()
    for dt in (torch.bfloat16, torch.float16):
        for B, n in ((1, 4096), (2, 4096), (1, 1024)):
            qkv = ops.cast((testing.seeded_tensor(f"h8.{B}.{n}", (B * n, 768)) * 1.2).cuda(), dt)
            o = ops.attention_h16(qkv, 256, 1 / 16.0, B=B)
            h.update(o.view(torch.int16).cpu().numpy().tobytes())
    print("DIGEST", h.hexdigest())
    sys.exit(0)
dig = []
for v in ("0", "1"):
    env = dict(os.environ, SGAM_ATTN_H8=v)
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env, timeout=600)
    line = [x for x in r.stdout.splitlines() if x.startswith("DIGEST")]
    if r.returncode != 0 or not line:
        print(r.stdout[-800:], r.stderr[-800:])
        sys.exit(1)
    dig.append(line[0])
print(dig)
print("BIT-IDENTICAL" if dig[0] == dig[1] else "DIFFERENT")
sys.exit(0 if dig[0] == dig[1] else 2)
