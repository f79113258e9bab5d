"""In-tree build of libocean_waves.so (HIP, gfx950 only) with hipcc.  No JIT cache, no fallback."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libocean_waves.so")
ARCH = "gfx950"

# translation unit -> extra flags.  ow_spectrum.hip is built with FP contraction off: its omega(k)
# plane must be bit-identical to the oracle's; ow_consumer.hip likewise (the sampling
# arithmetic is checked operation for operation).
UNITS = {
    "ow_frame.hip": [],
    "ow_spectrum.hip": ["-ffp-contract=off"],
    "ow_runtime.hip": [],
    "ow_consumer.hip": ["-ffp-contract=off"],
    "ow_group.hip": [],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libocean_waves.so cannot be built (there is no CPU fallback)")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "ocean_waves.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(item):
        src, extra = item
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if force or _stale(obj, [srcp] + headers):
            cmd = [hipcc] + COMMON + extra + ["-c", srcp, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        results = list(ex.map(compile_one, UNITS.items()))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
