#!/usr/bin/env python3
"""Time ops.linear over a shape sweep (exact and split): separates per-tile overhead, K-loop rate and tile quantisation."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommender_amd import ops

dev = torch.device("cuda", 0)


def t_ms(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]; e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        e0[i].record(); fn(); e1[i].record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in zip(e0, e1))[reps // 2]


if len(sys.argv) > 1 and sys.argv[1] == "--short":
    SHORT = [(3706, 6040, 500), (4096, 4096, 32), (4096, 4096, 128), (4096, 4096, 512), (4096, 4096, 2048), (3706, 500, 6040)]
else:
    SHORT = None
shapes = SHORT or [(3706, 6040, 500), (3712, 6144, 512), (3712, 6144, 1024), (3712, 6144, 2048), (3712, 6144, 4096),
          (4096, 4096, 512), (4096, 4096, 4096), (8192, 8192, 512), (2048, 4096, 512), (4096, 8192, 512),
          (3706, 500, 6040), (3712, 512, 6144), (3712, 512, 4096), (4096, 512, 8192)]
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for prec in ("f32", "bf16x3"):
        ms = t_ms(lambda: ops.linear(x, w, b, "clamp", 1.0, 5.0, out=out, precision=prec))
        tf = 2.0 * M * N * K / ms / 1e9
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        print(json.dumps({"M": M, "N": N, "K": K, "precision": prec, "ms": round(ms, 4), "TFLOPs": round(tf, 1),
                          "frac": round(tf / 157.3, 3), "tiles128": tiles, "rounds_of_512": round(tiles / 512, 2)}), flush=True)
