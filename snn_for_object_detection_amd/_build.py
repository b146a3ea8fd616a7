"""In-tree build of the gfx950 C-ABI library (``libsnn_hip.so``) with hipcc.

``python -m snn_for_object_detection_amd._build`` or ``__graft_entry__.build()``.
The library links only against the HIP runtime - no torch types cross the ABI
(``include/snn_hip.h``).

What is rebuilt is decided by CONTENT, not by file times: every object carries a stamp
(``csrc/<name>.o.buildstamp``) holding the sha256 of its source, the shared headers, the compiler
flags and ``hipcc --version``; the library's stamp (``libsnn_hip.so.buildstamp``) holds the
fingerprint of all kernel sources (``source_fingerprint()``, the same digest ``bench.py`` ties its PMC
traffic files to).  A shipped binary next to edited sources is therefore detected - by ``build()``,
which recompiles, and by ``_hip.load()``, which refuses to run a stale library.
"""

import hashlib
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SCRATCH = os.path.join(os.path.dirname(HERE), "build")   # tuning / stamp / clock libraries: never in the package
LIB_NAME = "libsnn_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
STAMP_PATH = LIB_PATH + ".buildstamp"
SOURCES = ("elementwise.hip", "neuron.hip", "conv.hip", "wgrad_halo.hip", "conv_halo.hip", "detect.hip",
           "targets.hip")
HEADERS = (os.path.join(CSRC, "snn_common.h"), os.path.join(INCLUDE, "snn_hip.h"))
ARCH = "gfx950"
# -ffp-contract=off: the pointwise kernels must round like the reference's unfused torch ops.
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-ffp-contract=off", "-std=c++17", "-Wall",
         "-Wno-unused-function", f"-I{INCLUDE}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _sha(*chunks: bytes) -> str:
    h = hashlib.sha256()
    for c in chunks:
        h.update(c)
    return h.hexdigest()


def source_fingerprint() -> str:
    """sha256 over every ``csrc/*.hip`` / ``*.h`` (name + bytes, sorted) and ``include/snn_hip.h``."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(open(os.path.join(INCLUDE, "snn_hip.h"), "rb").read())
    return h.hexdigest()


def _read_stamp(path: str) -> dict:
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def library_is_current() -> bool:
    """True when ``libsnn_hip.so`` exists and was built from the kernel sources as they are now."""
    return os.path.exists(LIB_PATH) and _read_stamp(STAMP_PATH).get("sources") == source_fingerprint()


def _compiler_id(hipcc: str) -> str:
    res = subprocess.run([hipcc, "--version"], capture_output=True, text=True)
    return _sha(res.stdout.encode())[:16]


def build(force: bool = False, verbose: bool = False, tuning: bool = False, stamp: bool = False,
          clock: bool = False) -> str:
    """``tuning=True`` (``--tuning``): compile the bisecting / tuning environment knobs in (``-DSNN_TUNING``) and write
    ``build/libsnn_hip_tuning.so`` (load it with ``SNN_HIP_LIB=...``); the product library reads no environment.
    ``stamp=True`` (``--stamp``): additionally ``-DSNN_STAMP`` (in-kernel cycle stamps of the conv main loop,
    ``tools/stamp_conv.py``) -> ``build/libsnn_hip_stamp.so``.
    ``clock=True`` (``--clock``): ``-DSNN_CLOCK`` only (begin / end stamps of the shader and the wall clock per block: the
    clock the chip holds under the kernel's load, ``tools/clock_conv.py``) -> ``build/libsnn_hip_clock.so``."""
    if not (force or tuning or stamp or clock) and library_is_current():
        return LIB_PATH   # the shipped binary matches the sources: nothing to do (and no hipcc needed)
    hipcc = _hipcc()
    if tuning or stamp or clock:
        return _build_tuning(hipcc, verbose, stamp, clock)
    cc = _compiler_id(hipcc)
    header_bytes = b"".join(open(h, "rb").read() for h in HEADERS)
    objs, jobs = [], []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        opath = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(opath)
        key = _sha(open(spath, "rb").read(), header_bytes, " ".join(FLAGS[:-1]).encode(), cc.encode())  # (no -I path)
        if force or not os.path.exists(opath) or _read_stamp(opath + ".buildstamp").get("key") != key:
            jobs.append(([hipcc, *FLAGS, "-c", spath, "-o", opath], opath, key))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)

    def compile_one(job):
        cmd, opath, key = job
        run(cmd)
        with open(opath + ".buildstamp", "w") as f:
            json.dump({"key": key}, f)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(compile_one, jobs))
    if os.path.exists(STAMP_PATH):
        os.remove(STAMP_PATH)   # never leave a stamp that vouches for a half-written library
    run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH])
    with open(STAMP_PATH, "w") as f:
        json.dump({"sources": source_fingerprint(), "flags": FLAGS, "hipcc": cc}, f)
    return LIB_PATH


def build_sanitized(verbose: bool = False) -> dict:
    """HOST side of the C ABI with AddressSanitizer + UndefinedBehaviorSanitizer (``hipcc --offload-host-only``: argument
    checks, geometry / planning / size helpers, launch set-up; no device code) -> ``build/libsnn_hip_asan.so``.
    Sanitizers run on the CPU build only; ``tests/test_host_sanitizers.py`` drives the helpers through
    it in a child process with the ASan runtime preloaded.  Returns the library path and the runtime to preload."""
    hipcc = _hipcc()
    os.makedirs(SCRATCH, exist_ok=True)
    out = os.path.join(SCRATCH, "libsnn_hip_asan.so")
    key = _sha(source_fingerprint().encode(), _compiler_id(hipcc).encode())
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    rt = subprocess.run([clang, "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.exists(out) and _read_stamp(out + ".buildstamp").get("key") == key):
        cmd = [hipcc, "--offload-host-only", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
               "-fno-omit-frame-pointer", "-fPIC", "-ffp-contract=off", "-std=c++17", f"-I{INCLUDE}"]
        if verbose:
            print(" ".join(cmd), flush=True)
        # two steps: the host objects reference one `__hip_fatbin_<hash>` per translation unit (the device code object a
        # full build embeds); a host-only build has none, so an EMPTY bundle stands in for each - the runtime registers
        # it at load time and never looks inside (no kernel is ever launched from this library)
        objs = []
        for src in SOURCES:
            o = os.path.join(SCRATCH, "asan_" + src.replace(".hip", ".o"))
            c = cmd + ["-c", os.path.join(CSRC, src), "-o", o]
            res = subprocess.run(c, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed:\n{' '.join(c)}\n{res.stdout}\n{res.stderr}")
            objs.append(o)
        syms = set()
        for o in objs:
            for line in subprocess.run(["nm", "-u", o], capture_output=True, text=True).stdout.splitlines():
                if "__hip_fatbin_" in line:
                    syms.add(line.split()[-1])
        stub = os.path.join(SCRATCH, "asan_fatbin_stub.c")
        with open(stub, "w") as f:
            for sym in sorted(syms):
                f.write(f'__attribute__((aligned(4096))) const char {sym}[4096] = "__CLANG_OFFLOAD_BUNDLE__";\n')
        res = subprocess.run([hipcc, "-fsanitize=address,undefined", "-shared", "-fPIC", "-x", "c", stub, "-x", "none",
                              *objs, "-o", out], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link of the sanitized host library failed:\n{res.stdout}\n{res.stderr}")
        with open(out + ".buildstamp", "w") as f:
            json.dump({"key": key}, f)
    return {"lib": out, "asan_runtime": rt}


def _build_tuning(hipcc: str, verbose: bool, stamp: bool = False, clock: bool = False) -> str:
    os.makedirs(SCRATCH, exist_ok=True)
    out = os.path.join(SCRATCH, "libsnn_hip_stamp.so" if stamp else "libsnn_hip_clock.so" if clock
                       else "libsnn_hip_tuning.so")
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
