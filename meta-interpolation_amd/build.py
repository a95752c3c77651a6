"""Build recipe for libsavfi_hip.so (gfx950 only).

`python meta-interpolation_amd/build.py` or `build_library()` compiles every translation unit under
csrc/ with hipcc and links them into ``meta-interpolation_amd/lib/libsavfi_hip.so`` IN-TREE, so the
shared object travels with the repository snapshot to the GPU box.  hipcc cross-compiles for gfx950
without a GPU present.
"""
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsavfi_hip.so")
STAMP = os.path.join(LIB_DIR, "libsavfi_hip.stamp")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _fingerprint():
    h = hashlib.sha256()
    files = _sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    files.append(os.path.join(REPO_DIR, "include", "savfi_hip.h"))
    for f in files:
        h.update(f.encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def find_hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libsavfi_hip.so cannot be built on this machine")
    return exe


def build_library(force=False, verbose=True):
    """Compile csrc/*.hip -> lib/libsavfi_hip.so.  Returns the path.  Skips when up to date."""
    os.makedirs(LIB_DIR, exist_ok=True)
    fp = _fingerprint()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == fp:
                return LIB_PATH
    # one object per translation unit, compiled in parallel and cached by content (a scratch directory outside the tree: only the linked
    # library travels with the snapshot), then one link
    hipcc = find_hipcc()
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-I", os.path.join(REPO_DIR, "include"), "-I", CSRC]
    cache = os.environ.get("SAVFI_BUILD_CACHE", "/tmp/savfi_build_cache")
    os.makedirs(cache, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [os.path.join(REPO_DIR, "include", "savfi_hip.h")]
    hh = hashlib.sha256()
    for f in headers:
        with open(f, "rb") as fh:
            hh.update(fh.read())
    hh.update(" ".join(cflags).encode())

    def compile_unit(src):
        h = hh.copy()
        with open(src, "rb") as fh:
            h.update(fh.read())
        obj = os.path.join(cache, os.path.basename(src)[:-4] + "." + h.hexdigest()[:16] + ".o")
        if not os.path.exists(obj):
            tmp = obj + ".tmp%d" % os.getpid()
            subprocess.run([hipcc] + cflags + ["-c", src, "-o", tmp], check=True)
            os.replace(tmp, obj)
        return obj

    if verbose:
        print("[savfi build]", hipcc, " ".join(cflags), "-c <%d units> (cache %s)" % (len(_sources()), cache), flush=True)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(len(_sources()), (os.cpu_count() or 4)))) as pool:
        objs = list(pool.map(compile_unit, _sources()))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", LIB_PATH], check=True)
    with open(STAMP, "w") as fh:
        fh.write(fp + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
