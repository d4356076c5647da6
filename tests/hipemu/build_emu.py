"""TEST INFRASTRUCTURE: compiles the product's kernel sources for x86 against tests/hipemu/include
(a CPU stand-in for <hip/hip_runtime.h>) into tests/hipemu/_build/libmphsir_emu.so.

The emulated library exports the same C ABI as libmphsir.so and is loaded only by tests (see
tests/emu.py) to check kernel logic on machines without a GPU.  The product never loads it.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "mp-hsir_amd", "csrc")
# HIPEMU_SANITIZE=1: the same sources with AddressSanitizer + UBSan (the GPU pool has no sanitizers: the kernels' LDS tiles and global
# accesses are checked on the CPU build).  Run the emulator tests with the runtime preloaded:
#   HIPEMU_SANITIZE=1 LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 pytest tests/test_emu_kernels.py
SANITIZE = os.environ.get("HIPEMU_SANITIZE", "0") == "1"
BUILD = os.path.join(HERE, "_build_asan" if SANITIZE else "_build")
OUT = os.path.join(BUILD, "libmphsir_emu.so")
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-ffp-contract=off", "-mavx2", "-mfma", "-mf16c",
         "-I", os.path.join(HERE, "include"), "-Wno-unused-function", "-Wno-unknown-attributes",
         "-fno-omit-frame-pointer"] + (["-fsanitize=address,undefined", "-fno-sanitize=alignment,float-cast-overflow,vptr", "-fno-sanitize-recover=undefined", "-shared-libasan"] if os.environ.get("HIPEMU_SANITIZE", "0") == "1" else [])


def _mtime_deps():
    deps = [os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "mphsir.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return max(os.path.getmtime(d) for d in deps)


def _compile(src, dep_mtime):
    obj = os.path.join(BUILD, os.path.basename(src)[:-4] + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), dep_mtime):
        return obj
    extra = ["-DHIPEMU_IMPLEMENTATION"] if os.path.basename(src) == "common.hip" else []
    r = subprocess.run([CXX] + FLAGS + extra + ["-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipemu compile failed for %s:\n%s" % (src, r.stderr[-8000:]))
    return obj


def build(jobs=6):
    os.makedirs(BUILD, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    dm = _mtime_deps()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, dm), srcs))
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(o) for o in objs):
        return OUT
    r = subprocess.run([CXX, "-shared", "-fPIC"] + (["-fsanitize=address,undefined", "-shared-libasan"] if SANITIZE else []) + ["-o", OUT] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipemu link failed:\n" + r.stderr[-8000:])
    return OUT


if __name__ == "__main__":
    print(build())
