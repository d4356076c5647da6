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


def golden_grad_err(got, g, name, key):
    """Relative error of gradient tensor `got` against the block fixture entry (blocks.npz): full tensors are compared
    in rel-L2; the large ones are stored as norm + seeded samples (golden.cases.GRAD_FULL_MAX) and compared as
    max(|norm - want| / want, rel-L2 over the samples)."""
    from golden.cases import GRAD_SAMPLES, sample_indices
    got = torch.as_tensor(got).detach().double().cpu()
    full = "%s/%s" % (name, key)
    if full in g.files:
        return rel_l2(got, g[full])
    idx = sample_indices(name + ":" + key, got.numel(), GRAD_SAMPLES)
    want_n = float(g["%s/norm/%s" % (name, key)])
    e_norm = abs(float(got.norm()) - want_n) / (want_n + 1e-30)
    return max(e_norm, rel_l2(got.flatten()[idx], g["%s/samp/%s" % (name, key)]))
