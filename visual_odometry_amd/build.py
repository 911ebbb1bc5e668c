"""In-tree build of the HIP shared library (``hipcc`` for gfx950, no cmake).

``python -m visual_odometry_amd.build`` compiles every ``csrc/*.hip`` into one
object each (in parallel) and links ``lib/libvo_hip.so``.  hipcc cross-compiles
without a GPU, so this runs in the authoring container; the built ``.so`` travels
to the GPU box with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(LIBDIR, "libvo_hip.so")
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc", "-DNDEBUG"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (ROCm toolchain required to build libvo_hip.so)")
    return exe


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(PKG), "include", "vo_hip.h"))
    return hs


def _compile(src: str, extra=()) -> str:
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    if _newer(obj, [src] + headers()):
        return obj
    cmd = [hipcc(), *CFLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False, jobs: int = 8) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    if not force and _newer(LIB, srcs + headers()):
        return LIB
    with ThreadPoolExecutor(max_workers=min(jobs, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_host_checks() -> str:
    """CPU-only shared object exposing the host+device math for the CPU tests."""
    root = os.path.dirname(PKG)
    src = os.path.join(root, "tests", "host_checks.hip")
    out = os.path.join(root, "tests", "_host_checks.so")
    if _newer(out, [src] + headers()):
        return out
    cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
    print(build_host_checks())
