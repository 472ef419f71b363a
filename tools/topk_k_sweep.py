"""Long-row top-k scan (256 x 1M) over k, plus ascending / descending rows: separates streaming from list upkeep."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommender_amd import ops
x = torch.randn(256, 1_000_000, device='cuda')
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for k in (1, 2, 10, 32, 64, 65, 100, 128):
    print(k, round(t(lambda: ops.topk_rows(x, k)), 4), flush=True)
# sorted rows: ascending (every element is a new maximum: worst case) and descending (nothing enters after the first k)
xs = torch.sort(x[:64], dim=1).values.contiguous()
print('asc k=100', round(t(lambda: ops.topk_rows(xs, 100), 5), 4))
xd = xs.flip(1).contiguous()
print('desc k=100', round(t(lambda: ops.topk_rows(xd, 100), 5), 4), '(64 rows)')
print('rand 64 rows k=100', round(t(lambda: ops.topk_rows(x[:64], 100), 5), 4))
