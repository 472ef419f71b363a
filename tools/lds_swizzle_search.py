#!/usr/bin/env python3
"""Brute-force search for the LDS layout of linear_kernel's transposed-A staging (gemm.hip at_swz): row stride (in 16-byte slots),
and an XOR of the k-chunk position with ((row >> a) & m), such that BOTH the transposing writes (16 consecutive lanes write rows
4 apart at one chunk) and the fragment reads (16 consecutive lanes read 16 consecutive rows at one chunk) hit 16 distinct slots."""
from collections import Counter


def worst(slots):
    return max(Counter(s % 16 for s in slots).values())


res = []
for a in range(6):
    for m in (0, 1, 3, 7):
        for stride in (8, 9, 10, 11, 12):
            f = lambda r: (r >> a) & m
            w = max(worst([(4 * tc + i4) * stride + (c ^ f(4 * tc + i4)) for tc in range(g, g + 16)])
                    for g in (0, 16) for i4 in range(4) for c in range(8))
            r = max(worst([row * stride + (c ^ f(row)) for row in range(g, g + 16)]) for g in (0, 16) for c in range(8))
            res.append((w, r, stride, a, m))
for w, r, stride, a, m in sorted(set(res))[:10]:
    print("write conflicts %d-way, read %d-way: stride %d slots, chunk ^ ((row >> %d) & %d)" % (w, r, stride, a, m))
print("padded stride alone (9 slots, no xor):", [x for x in res if x[2] == 9 and x[4] == 0][0][:2])
