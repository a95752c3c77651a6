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
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-I", os.path.join(REPO_DIR, "include"), "-I", CSRC]
    cmd += _sources() + ["-o", LIB_PATH]
    if verbose:
        print("[savfi build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(fp + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
