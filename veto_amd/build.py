"""Builds veto_amd/csrc/libveto_amd.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libveto_amd.so")
SOURCES = ["gemm_split_ps.hip", "ffn_fused.hip", "rowops.hip", "attention.hip", "postprocess.hip", "roialign.hip", "sgg_eval.hip", "losses.hip", "backward.hip", "train.hip", "veto_abi.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "veto_amd.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_native(force=False, verbose=False):
    """Compiles every HIP source for gfx950 into one shared library. Returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % LIB)
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
