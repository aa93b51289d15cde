"""Builds libnrtgpu.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnrtgpu.so")
SOURCES = ["kernels.hip", "maxscore.hip", "knn.hip", "runtime.cpp", "segment.cpp", "planner.cpp", "search.cpp", "vectors.cpp", "dist.cpp"]
HEADERS = ["plan.h", "topk.hiph", "bm25_common.hiph", "host_math.h", "runtime_internal.h", os.path.join("..", "..", "include", "nrtgpu.h")]
# -ffp-contract=off + no fast-math: BM25 arithmetic must round exactly like Java's float ops.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-parallel-jobs=8", "-x", "hip"]   # (the translation units compile side by side: 50 s -> 23 s)


def _stale(out: str = OUT, more=()) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS + list(more)] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


DEV_OUT = os.path.join(HERE, "libnrtgpu_dev.so")
DEV_SOURCES = ["devtools.cpp"]   # measurement helpers outside the product ABI (include/nrtgpu_dev.h)


def build_dev(verbose: bool = False) -> str:
    """The development build: the product sources + devtools.cpp with -DNRTGPU_DEV (closed-loop load generator, timing
    ablations of the scan).  NRTGPU_LIB_PATH=<this file> makes nrtsearch_amd._lib load it."""
    if not _stale(DEV_OUT, DEV_SOURCES + [os.path.join("..", "..", "include", "nrtgpu_dev.h")]):
        return DEV_OUT
    return build(force=True, verbose=verbose, extra=["-DNRTGPU_DEV"] + [os.path.join(CSRC, s) for s in DEV_SOURCES], out=DEV_OUT)


def build(force: bool = False, verbose: bool = False, extra=(), out: str = OUT) -> str:
    """extra: additional hipcc flags (e.g. -DNRT_SCAN_WAVES=16 -DNRT_TILE_DOCS=768 for an A/B build
    written to `out`; NRTGPU_LIB_PATH makes nrtsearch_amd._lib load it)."""
    if not force and out == OUT and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


_BUILD_ID_CACHE = {}


def build_id(path: str = OUT):
    """Identity of the DEVICE code in a built library: sha256 (first 16 hex digits) over the `.text` sections of its gfx950
    code objects (llvm-objdump --offloading unbundles them; the bundle itself is not reproducible -- it embeds temporary
    file names -- but the machine code is).  A measurement can then name the kernels it was taken from: bench.py stamps
    `roofline.build_id`, profiles/pmc_traffic.json records carry the id of the build they profiled, and bench.py refuses to
    reuse their traffic for another one.  Host-side edits do not change it; any kernel edit does.  None when the LLVM
    binutils are not there."""
    import hashlib
    import shutil
    import tempfile

    key = (os.path.abspath(path), os.path.getmtime(path))
    if key in _BUILD_ID_CACHE:
        return _BUILD_ID_CACHE[key]
    llvm = os.environ.get("NRTGPU_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    objdump, objcopy = os.path.join(llvm, "llvm-objdump"), os.path.join(llvm, "llvm-objcopy")
    out = None
    if os.path.exists(objdump) and os.path.exists(objcopy):
        tmp = tempfile.mkdtemp(prefix="nrtgpu_id_")
        try:
            local = os.path.join(tmp, "lib.so")
            shutil.copy(path, local)
            subprocess.run([objdump, "--offloading", local], cwd=tmp, check=True, capture_output=True)
            digests = []
            for f in sorted(os.listdir(tmp)):
                if "gfx950" not in f:
                    continue
                text = os.path.join(tmp, f + ".text")
                subprocess.run([objcopy, "-O", "binary", "--only-section=.text", os.path.join(tmp, f), text], check=True, capture_output=True)
                digests.append(hashlib.sha256(open(text, "rb").read()).hexdigest())
            if digests:
                out = hashlib.sha256("".join(sorted(digests)).encode()).hexdigest()[:16]
        except (OSError, subprocess.CalledProcessError):
            out = None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    _BUILD_ID_CACHE[key] = out
    return out


LOADGEN_SRC = os.path.join(HERE, "..", "bench", "loadgen", "loadgen.cpp")
LOADGEN_OUT = os.path.join(HERE, "..", "bench", "loadgen", "libloadgen.so")


def build_loadgen() -> str:
    """bench.py's closed-loop load generator: a host-only helper that knows nothing but include/nrtgpu.h (not part of the product)."""
    src, out = os.path.abspath(LOADGEN_SRC), os.path.abspath(LOADGEN_OUT)
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "include", "nrtgpu.h"))):
        return out
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", src, "-o", out])
    return out


if __name__ == "__main__":
    print(build_dev(verbose=True) if "--dev" in sys.argv else build(force="--force" in sys.argv, verbose=True))
