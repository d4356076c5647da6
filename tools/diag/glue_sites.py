"""Which lines of the package issue framework (at::native) kernels in a training step?  A TorchDispatchMode logs every aten op that
launches a kernel, with the innermost package frames.  Runs on the GPU box (full net) or here on the CPU-emulated library (tiny net):
    python tools/diag/glue_sites.py [emu] [fwd]        (fwd: the no-grad forward instead of a training step)"""
import collections, os, sys, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import torch
from torch.utils._python_dispatch import TorchDispatchMode

emu = "emu" in sys.argv[1:]
fwd = "fwd" in sys.argv[1:]
if emu:
    import emu as E
    E.bind_emulator()
from mp_hsir_amd.data import SyntheticPatchSource
from mp_hsir_amd.engine import DataParallelEngine
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net

NOKERNEL = ("aten::empty", "aten::view", "aten::_unsafe_view", "aten::as_strided", "aten::detach", "aten::alias", "aten::reshape", "aten::permute",
            "aten::transpose", "aten::t", "aten::slice", "aten::select", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::split", "aten::unbind",
            "aten::narrow", "aten::_local_scalar_dense", "aten::is_", "aten::size", "aten::stride", "aten::empty_like", "aten::empty_strided", "aten::new_empty",
            "aten::lift_fresh", "aten::set_", "aten::record_stream", "aten::chunk", "aten::unflatten", "aten::flatten", "aten::view_as", "aten::_reshape_alias",
            "aten::unfold", "aten::movedim", "aten::is_same_size", "aten::resize_", "aten::result_type", "aten::_has_compatible_shallow_copy_type", "aten::sym_")
log = collections.OrderedDict()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name()
        if not name.startswith(NOKERNEL):
            fr = [f for f in traceback.extract_stack() if "hsir" in f.filename and "glue_sites" not in f.filename][-3:]
            shp = next((tuple(a.shape) for a in args if torch.is_tensor(a)), None)
            if shp is None and args and isinstance(args[0], (list, tuple)) and args[0] and torch.is_tensor(args[0][0]):
                shp = ("list", len(args[0]))
            key = (name, tuple("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr))
            e = log.setdefault(key, [0, set()])
            e[0] += 1
            e[1].add(shp)
        return func(*args, **(kwargs or {}))


if emu:
    from golden.cases import TINY_CFG
    from golden.detfill import surrogate_clip_prompt
    dev = torch.device("cpu")
    net = MP_HSIR_Net(**TINY_CFG, clip_prompt=surrogate_clip_prompt(6)).train()
    net.set_compute_dtype(torch.bfloat16)
    src = SyntheticPatchSource(8, 32, 2, 6, dev, 2024, 0)
else:
    dev = torch.device("cuda")
    net = MP_HSIR_Net(compute_dtype=torch.bfloat16, clip_prompt="surrogate").to(dev).train()
    src = SyntheticPatchSource(31, 512, 1, 6, dev, 2024, 0) if fwd else SyntheticPatchSource(31, 64, 32, 6, dev, 2024, 0)
if fwd:
    net.eval()
    with torch.no_grad():
        for _ in range(2):
            _, x, c, p = src.next(); net(x, p)
        with Log():
            _, x, c, p = src.next(); net(x, p)
else:
    eng = DataParallelEngine(net, lr=2e-4, use_graph=False)
    for _ in range(2):
        _, x, c, p = src.next(); eng.train_step(x, c, p)
    with Log():
        _, x, c, p = src.next(); eng.train_step(x, c, p)
print("%d aten ops that launch kernels" % sum(v[0] for v in log.values()))
for (name, fr), (n, shp) in sorted(log.items(), key=lambda kv: -kv[1][0]):
    print("x%-3d %-28s %s   %s" % (n, name, " <- ".join(reversed(fr)), sorted(shp, key=str)[:3]))
