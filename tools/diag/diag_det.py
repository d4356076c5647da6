import sys, torch, warnings
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
warnings.filterwarnings("ignore")
import model_checks as M
c, clean, degraded = M.full_case_inputs("natural_mode0")
net = M.build_net(c["cfg"], "cuda", torch.bfloat16)
x, t = degraded.cuda(), torch.tensor(c["task"]).cuda()
caps = []
def mk(name):
    def hook(m, i, o):
        caps[-1][name] = o.detach().clone()
    return hook
for name, m in net.named_modules():
    if name and name.count('.') <= 2 and not name.startswith('text_prompt'):
        m.register_forward_hook(mk(name))
with torch.no_grad():
    for r in range(3):
        caps.append({})
        net(x, t)
a, b = caps[1], caps[2]
for k in a:
    if not torch.equal(a[k], b[k]):
        print("DIFF", k, float((a[k].float()-b[k].float()).abs().max()))
print("done", len(a))
