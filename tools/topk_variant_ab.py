import sys, os, torch, json
sys.path.insert(0, os.getcwd())
from recommender_amd import ops
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
torch.manual_seed(0)
x = torch.randn(256, 1_000_000, device='cuda')
y = torch.randn(6038, 3533, device='cuda')
z = torch.randn(1024, 200_000, device='cuda')
out = {"v": os.environ.get("V")}
for k in (10, 100):
    out["256x1M k%d" % k] = round(t(lambda: ops.topk_rows(x, k)), 4)
out["6038x3533 k10"] = round(t(lambda: ops.topk_rows(y, 10)), 4)
out["1024x200k k100"] = round(t(lambda: ops.topk_rows(z, 100)), 4)
print(json.dumps(out), flush=True)
