"""Build libemadapt_hip.so (all HIP kernels + the C ABI of include/emadapt.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The .so stays next to this file so that it travels with the
repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures
import fcntl
import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(PKG_DIR, "libemadapt_hip.so")
# diagnostic build (-DEMADAPT_ABLATE_BUILD: E-step phase exits, igemm ablation bits); never loaded by the product path,
# only by scripts/ (they bind it explicitly with _lib._bind before using any operator)
ABLATE_LIB_PATH = os.path.join(PKG_DIR, "libemadapt_hip_ablate.so")
ARCH = "gfx950"
# No -ffast-math: the E-step must keep IEEE fp32 semantics (bit-exact label maps), and the conv
# accumulation order is part of the documented numerics.
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP toolchain (ROCm) is required to build em-adapt-tensorflow_amd")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and the C ABI header, in name order: what a committed counter profile
    was taken on.  bench.py refuses to quote roofline.traffic from a profile whose hash differs (a claim carried forward would go
    stale silently when the kernels change)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(PKG_DIR, "..", "include", "emadapt.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _deps_mtime():
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h"))
    return max(os.path.getmtime(p) for p in hdrs)


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in sources()) or _deps_mtime() > t


def _compile(src, obj_dir=OBJ_DIR, extra=()):
    obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _deps_mtime()):
        return obj
    subprocess.check_call([_hipcc()] + CXXFLAGS + list(extra) + ["-c", src, "-o", obj])
    return obj


def build(force=False, jobs=None, verbose=True, ablate=False):
    """Compile every csrc/*.hip for gfx950 and link libemadapt_hip.so.  Returns the .so path.
    Serialised across processes by a file lock (ranks of a data-parallel job may all find the library stale)."""
    lib_path = ABLATE_LIB_PATH if ablate else LIB_PATH
    obj_dir = OBJ_DIR + ("_ablate" if ablate else "")
    os.makedirs(obj_dir, exist_ok=True)
    with open(os.path.join(obj_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            stale = is_stale() if not ablate else (
                not os.path.exists(lib_path) or os.path.getmtime(lib_path) < max(
                    [os.path.getmtime(s) for s in sources()] + [_deps_mtime()]))
            if not force and not stale:
                return lib_path
            if force:
                for o in glob.glob(os.path.join(obj_dir, "*.o")):
                    os.remove(o)
            srcs = sources()
            jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) // 2))
            extra = ("-DEMADAPT_ABLATE_BUILD",) + tuple(os.environ.get("EMADAPT_LAB_CXXFLAGS", "").split()) if ablate else ()   # (lab flags: diagnostic build only)
            with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
                objs = list(ex.map(lambda src: _compile(src, obj_dir, extra), srcs))
            tmp = "%s.tmp.%d" % (lib_path, os.getpid())
            subprocess.check_call([_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", tmp] + objs)
            os.replace(tmp, lib_path)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if verbose:
        print("built %s (%d sources)" % (lib_path, len(srcs)))
    return lib_path


if __name__ == "__main__":
    import sys

    build(force="--force" in sys.argv, ablate="--ablate" in sys.argv)
