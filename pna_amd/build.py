"""Builds libpna_amd.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m pna_amd.build [--force]

The shared object lands in pna_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).  hipcc
cross-compiles for gfx950 without a GPU being present.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpna_amd.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # op-by-op fp32 rounding like the reference's torch code: no fused multiply-add contraction
    "-ffp-contract=off",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every .hip under csrc/ into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpna_amd.so (and no prebuilt copy is current)")
    os.makedirs(LIBDIR, exist_ok=True)
    tmp = LIB + ".tmp"
    cmd = [hipcc] + HIPCC_FLAGS + sources() + ["-o", tmp]
    if verbose:
        print("[pna_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
