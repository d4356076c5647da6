"""Register / LDS / scratch figures of every kernel in the built objects, from the code-object metadata hipcc writes.

    python tools/kernel_meta.py [--spills] [pattern]     # name, vgpr, agpr, vgpr spills, scratch bytes, LDS bytes

The same parser backs tests/test_kernel_meta.py (no instantiation the training step uses may spill)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "mp-hsir_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"

_KEYS = ("name", "vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
         "group_segment_fixed_size", "max_flat_workgroup_size")


def _demangle(names):
    r = subprocess.run([shutil.which("c++filt") or "c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")[:len(names)] if r.returncode == 0 else names


def object_kernels(obj):
    """[{name, vgpr_count, ...}] of one object file built by mp-hsir_amd/build.py"""
    out = []
    with tempfile.TemporaryDirectory() as td:
        o = os.path.join(td, os.path.basename(obj))
        shutil.copy(obj, o)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", o], capture_output=True, text=True, check=True)
        cos = [os.path.join(td, f) for f in os.listdir(td) if "amdgcn" in f]
        for co in cos:
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "agpr_count" or (k == "args" and cur is None):
                    pass
                if k == "agpr_count":            # first key of a kernel record in the metadata's (alphabetical) order
                    cur = {"agpr_count": int(v)}
                    out.append(cur)
                elif cur is not None and k in _KEYS:
                    cur[k] = v.strip("'\"") if k == "name" else int(v)
    out = [k for k in out if "name" in k]
    for k, d in zip(out, _demangle([k["name"] for k in out])):
        k["demangled"] = d
    return out


def all_kernels(build_dir=BUILD):
    ks = []
    for f in sorted(os.listdir(build_dir)):
        if f.endswith(".o"):
            for k in object_kernels(os.path.join(build_dir, f)):
                k["object"] = f
                ks.append(k)
    return ks


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = re.compile(args[0]) if args else None
    only_spills = "--spills" in sys.argv
    print("%-110s %5s %5s %6s %8s %7s" % ("kernel", "vgpr", "agpr", "spill", "scratch", "lds"))
    for k in all_kernels():
        if pat and not pat.search(k["demangled"]):
            continue
        if only_spills and not (k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)):
            continue
        print("%-110s %5d %5d %6d %8d %7d" % (k["demangled"][:110], k.get("vgpr_count", -1), k.get("agpr_count", -1), k.get("vgpr_spill_count", 0),
                                              k.get("private_segment_fixed_size", 0), k.get("group_segment_fixed_size", 0)))
