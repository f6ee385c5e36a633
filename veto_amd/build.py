"""Builds veto_amd/csrc/libveto_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

Every source is compiled to its own object (in parallel, cached by modification time under build/obj, which is neither tracked nor
shipped) and the objects are linked into the one shared library the C ABI lives in."""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libveto_amd.so")
OBJ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "obj")
SOURCES = ["gemm_split_ps.hip", "ffn_fused.hip", "qkv_attn_fused.hip", "rowops.hip", "attention.hip", "postprocess.hip", "roialign.hip",
           "sgg_eval.hip", "losses.hip", "backward.hip", "train.hip", "veto_abi.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "veto_amd.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]


def _mtime(f):
    return os.path.getmtime(os.path.join(CSRC, f))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_mtime(f) > t for f in SOURCES + HEADERS)


def _toolchain_stamp(hipcc):
    """What the cached objects depend on besides their sources: the flags and the compiler."""
    try:
        version = subprocess.run([hipcc, "--version"], capture_output=True, text=True, check=True).stdout
    except (OSError, subprocess.CalledProcessError):
        version = "unknown"
    return hashlib.sha256((" ".join(FLAGS) + "\n" + version).encode()).hexdigest()


def build_native(force=False, verbose=False, jobs=None):
    """Compiles every HIP source for gfx950 into one shared library. Returns its path.  force=True recompiles every object."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % LIB)
    os.makedirs(OBJ, exist_ok=True)
    newest_header = max(_mtime(h) for h in HEADERS)
    # objects of other flags or another compiler are stale whatever their age (tools/variants.sh writes its objects under other names)
    stamp_file, stamp = os.path.join(OBJ, "toolchain.stamp"), _toolchain_stamp(hipcc)
    if not os.path.exists(stamp_file) or open(stamp_file).read() != stamp:
        force = True

    def compile_one(src):
        obj = os.path.join(OBJ, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(_mtime(src), newest_header):
            return obj
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, cwd=CSRC, check=True)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    with open(stamp_file, "w") as f:
        f.write(stamp)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
