"""Build the native library: hipcc cross-compiles every .hip source for gfx950 (no GPU needed)
into srcgan_amd/lib/libsrcgan_amd.so, in-tree so it travels with the repo snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsrcgan_amd.so")
SOURCES = ["conv_igemm.hip", "conv_par4.hip", "conv3x3_dma.hip", "conv_wgrad.hip", "wgrad_dense.hip", "elementwise.hip", "groupnorm.hip", "metrics.hip", "colour.hip", "nets.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_params.h"), os.path.join(HERE, "..", "include", "srcgan_amd.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[srcgan_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    import ctypes
    ctypes.CDLL(LIB)          # a kernel whose host stub was not emitted links fine but fails here (undefined symbol)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
