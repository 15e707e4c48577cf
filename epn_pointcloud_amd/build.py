"""Build recipe for libepn_so3conv.so: hipcc --offload-arch=gfx950 on every csrc/*.hip, linked in-tree.

Counterpart of the reference's vgtk/setup.py:30-55 (three CUDAExtensions); here one C-ABI shared
library, no torch headers involved, so it cross-compiles on a GPU-less box in seconds.
"""
import concurrent.futures
import glob
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libepn_so3conv.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps_mtime():
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hdrs)


TUNING = os.environ.get("EPN_TUNING", "0") == "1"      # tools/ builds: keep the A/B kernel policies (epn_set_kernel_policy 0x100..0x4ff)
# A/B builds of the tools: EPN_BUILD_TAG=x EPN_EXTRA_FLAGS="-DFOO=1" -> libepn_so3conv_x.so (load with EPN_LIB=...)
TAG = os.environ.get("EPN_BUILD_TAG", "")
EXTRA = os.environ.get("EPN_EXTRA_FLAGS", "").split()


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + (f".{TAG}" if TAG else "") + (".tuning.o" if TUNING else ".o"))
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= _deps_mtime()):
        return obj, False
    subprocess.check_call([HIPCC] + FLAGS + (["-DEPN_TUNING"] if TUNING else []) + EXTRA + ["-c", src, "-o", obj])
    return obj, True


def build(force=False, verbose=False, tuning=None):
    """Compile (if stale) and link the shared library; returns its path.  tuning=True (or EPN_TUNING=1 / --tuning) builds
    libepn_so3conv_tuning.so with -DEPN_TUNING for the tools/ A/B scripts (load it with EPN_LIB=...)."""
    global TUNING, LIB
    if tuning is not None:
        TUNING = bool(tuning)
    lib_path = os.path.join(PKG, "libepn_so3conv_tuning.so") if TUNING else (
        os.path.join(PKG, f"libepn_so3conv_{TAG}.so") if TAG else LIB)
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    stale = not os.path.exists(lib_path) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs)
    if force or stale or any(c for _, c in results):
        subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs)
        if verbose:
            print("linked", lib_path)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True, tuning="--tuning" in os.sys.argv or None))
