"""Build recipe for libsgam_hip.so — hipcc, gfx950 only, in-tree (so the built library travels to the
GPU box with the repo snapshot).  `python -m sgam_neurips22_amd.build` or `__graft_entry__.build()`."""
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.environ.get("SGAM_LIB_DIR") or os.path.join(PKG, "lib")     # SGAM_LIB_DIR: variant builds for A/B experiments
LIB = os.path.join(LIBDIR, "libsgam_hip.so")
ARCH = "gfx950"

# exact-semantics translation units: no fp contraction (every fused op is an explicit __fmaf_rn)
SOURCES = {
    "conv_gemm.hip": [f"-DSGAM_SCHED={os.environ.get('SGAM_SCHED', '2')}"],
    "norm_softmax.hip": [],
    "groupnorm.hip": [],
    "h16.hip": [],
    "conv_f32x.hip": [f"-DSGAM_XPF_BIG={os.environ.get('SGAM_XPF_BIG', '1')}",
                      f"-DSGAM_XPF_SMALL={os.environ.get('SGAM_XPF_SMALL', '2')}",
                      f"-DSGAM_XABLATE={os.environ.get('SGAM_XABLATE', '0')}",
                      f"-DSGAM_XSB={os.environ.get('SGAM_XSB', '1')}",
                      f"-DSGAM_XNT={os.environ.get('SGAM_XNT', '0')}",
                      f"-DSGAM_XWGM={os.environ.get('SGAM_XWGM', '1')}",
                      f"-DSGAM_XSOFF={os.environ.get('SGAM_XSOFF', '1')}",
                      f"-DSGAM_XPEEL={os.environ.get('SGAM_XPEEL', '1')}",
                      f"-DSGAM_XLB64={os.environ.get('SGAM_XLB64', '2')}",
                      f"-DSGAM_XNBR64={os.environ.get('SGAM_XNBR64', '6')}",
                      f"-DSGAM_XRWARM={os.environ.get('SGAM_XRWARM', '0')}"],
    "h16_halo.hip": [f"-DSGAM_HABLATE={os.environ.get('SGAM_HABLATE', '0')}",
                     f"-DSGAM_HDIRECT={os.environ.get('SGAM_HDIRECT', '1')}",
                     f"-DSGAM_HWGM={os.environ.get('SGAM_HWGM', '1')}",
                     f"-DSGAM_HSB={os.environ.get('SGAM_HSB', '2')}",
                     f"-DSGAM_HFD2={os.environ.get('SGAM_HFD2', '1')}",
                     f"-DSGAM_HFD4={os.environ.get('SGAM_HFD4', '1')}",
                     f"-DSGAM_HNBR={os.environ.get('SGAM_HNBR', '3')}",
                     f"-DSGAM_HNBR64={os.environ.get('SGAM_HNBR64', '6')}",
                     f"-DSGAM_HNBRF={os.environ.get('SGAM_HNBRF', '6')}",
                     f"-DSGAM_HLT={os.environ.get('SGAM_HLT', '0')}",
                     f"-DSGAM_HPEEL={os.environ.get('SGAM_HPEEL', '1')}",
                     f"-DSGAM_HRPF={os.environ.get('SGAM_HRPF', '1')}",
                     f"-DSGAM_HSWISH={os.environ.get('SGAM_HSWISH', '0')}"],
    "attention.hip": [f"-DSGAM_ATTN_ABLATE={os.environ.get('SGAM_ATTN_ABLATE', '0')}"],
    "vq.hip": ["-ffp-contract=off"],
    "layout.hip": ["-ffp-contract=off"],
    "warp.hip": ["-ffp-contract=off"],
    "gemm_gn_f32x.hip": [],
    "train.hip": [],
    "build_info.hip": [],       # flags = the build stamp, filled in by build()
    "tsdf.hip": ["-ffp-contract=off", f"-DSGAM_TSDF_ZG={os.environ.get('SGAM_TSDF_ZG', '2')}",
                 f"-DSGAM_TSDF_LB={os.environ.get('SGAM_TSDF_LB', '8')}",
                 f"-DSGAM_TSDF_TOUCH_ABLATE={os.environ.get('SGAM_TSDF_TOUCH_ABLATE', '0')}"] +
                (["-DSGAM_TSDF_DEBUG_STEPS"] if os.environ.get("SGAM_TSDF_DEBUG_STEPS") else []),
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsgam_hip.so cannot be built (ROCm toolchain required)")


def _git(*args):
    try:
        r = subprocess.run(["git", "-C", os.path.join(PKG, "..")] + list(args), capture_output=True, text=True, timeout=10)
        return r.stdout.strip() if r.returncode == 0 else None
    except (OSError, subprocess.SubprocessError):
        return None


_STAMP_PATHS = ["sgam_neurips22_amd/csrc", "include", "sgam_neurips22_amd/build.py"]


def build_stamp(headers):
    """(commit, digest) compiled into the library (csrc/build_info.hip).  digest: sha256 over every source, header and flag.
    commit: the last commit that touched the library's sources, '+dirty' when the tree differs from it — from git where the
    build runs; without git (the GPU box) the stamp file of the build that produced the same digest is reused."""
    srcs = sorted(os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f)) and f != "build_info.hip")
    digest = _digest(srcs + headers, [COMMON] + [SOURCES[os.path.basename(p)] for p in srcs])[:12]
    stamp_file = os.path.join(LIBDIR, "BUILD_STAMP")
    commit = _git("log", "-1", "--format=%h", "--abbrev=12", "--", *_STAMP_PATHS)
    if commit:
        if _git("status", "--porcelain", "--", *_STAMP_PATHS):
            commit += "+dirty"
    else:
        commit = "nogit"
        if os.path.exists(stamp_file):
            old = open(stamp_file).read().split()
            if len(old) == 2 and old[1] == digest:
                commit = old[0]
    return commit, digest, stamp_file


def _digest(paths, flags):
    h = hashlib.sha256()
    for p in sorted(paths, key=os.path.basename):
        h.update(os.path.basename(p).encode())          # (names, not locations: the GPU box unpacks the tree under another root)
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(repr(flags).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libsgam_hip.so.  Incremental: a unit is
    recompiled only when its source, a shared header or its flags changed."""
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(PKG, "..", "include", "sgam_hip.h"))
    objs, relink = [], force or not os.path.exists(LIB)
    commit, digest, stamp_file = build_stamp(headers)
    SOURCES["build_info.hip"] = [f'-DSGAM_BUILD_COMMIT="{commit}"', f'-DSGAM_BUILD_DIGEST="{digest}"']
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest([path] + headers, COMMON + extra)
        old = open(stamp).read() if os.path.exists(stamp) else ""
        if force or not os.path.exists(obj) or old != dig:
            cmd = [hipcc] + COMMON + extra + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(dig)
            relink = True
        objs.append(obj)
    if relink:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(stamp_file, "w") as f:
        f.write(f"{commit} {digest}\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
