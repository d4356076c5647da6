"""Helpers shared by the tests: deterministic parameter dicts built from the committed manifest."""
import torch

from golden.detfill import det_value


def params_from_manifest(man, prefix="", dtype=torch.float64, rename_from=None):
    """{key (relative to prefix): tensor} for every floating-point entry of a manifest, filled with
    the same crc32(key)-seeded values make_golden.py gave the reference.  Keys are seeded by their
    name *relative to prefix* when rename_from is None (stand-alone sub-module goldens), which is how
    the generator filled stand-alone reference sub-modules."""
    P = {}
    for k, (shape, dt) in man.items():
        if not k.startswith(prefix) or not dt.startswith("float") or k.endswith("attn_mask"):
            continue
        rel = k[len(prefix):]
        P[rel] = det_value(rel, shape).to(torch.float32).to(dtype)   # generator filled an fp32 module
    return P


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
