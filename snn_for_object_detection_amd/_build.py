"""In-tree build of the gfx950 C-ABI library (``libsnn_hip.so``) with hipcc.

``python -m snn_for_object_detection_amd._build`` or ``__graft_entry__.build()``.
The library links only against the HIP runtime - no torch types cross the ABI
(``include/snn_hip.h``).  Objects are rebuilt when a source or header is newer.
"""

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_NAME = "libsnn_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ("elementwise.hip", "neuron.hip", "conv.hip", "wgrad_halo.hip", "detect.hip", "targets.hip")
ARCH = "gfx950"
# -ffp-contract=off: the pointwise kernels must round like the reference's unfused torch ops.
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-ffp-contract=off", "-std=c++17", "-Wall",
         "-Wno-unused-function", f"-I{INCLUDE}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, tuning: bool = False, stamp: bool = False,
          clock: bool = False) -> str:
    """``tuning=True`` (``--tuning``): compile the bisecting / tuning environment knobs in (``-DSNN_TUNING``) and write
    ``libsnn_hip_tuning.so`` (load it with ``SNN_HIP_LIB=...``); the product library reads no environment.
    ``stamp=True`` (``--stamp``): additionally ``-DSNN_STAMP`` (in-kernel cycle stamps of the conv main loop,
    ``tools/stamp_conv.py``) -> ``libsnn_hip_stamp.so``.
    ``clock=True`` (``--clock``): ``-DSNN_CLOCK`` only (begin / end stamps of the shader and the wall clock per block: the
    clock the chip holds under the kernel's load, ``tools/clock_conv.py``) -> ``libsnn_hip_clock.so``."""
    hipcc = _hipcc()
    if tuning or stamp or clock:
        return _build_tuning(hipcc, verbose, stamp, clock)
    headers = [os.path.join(CSRC, "snn_common.h"), os.path.join(INCLUDE, "snn_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        opath = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(opath)
        if force or _newer(opath, [spath] + headers):
            jobs.append([hipcc, *FLAGS, "-c", spath, "-o", opath])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(run, jobs))
    if force or jobs or _newer(LIB_PATH, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH])
    return LIB_PATH


def _build_tuning(hipcc: str, verbose: bool, stamp: bool = False, clock: bool = False) -> str:
    out = os.path.join(HERE, "libsnn_hip_stamp.so" if stamp else "libsnn_hip_clock.so" if clock else "libsnn_hip_tuning.so")
    cmd = [hipcc, *FLAGS, "-DSNN_TUNING", *(["-DSNN_STAMP"] if stamp else []), *(["-DSNN_CLOCK"] if clock else []),
           "-shared",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv, stamp="--stamp" in sys.argv,
                clock="--clock" in sys.argv))
