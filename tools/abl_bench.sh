#!/bin/bash
# diagnostic: time the tile kernel of ablation builds under tools/abl/ (outputs are wrong by construction; only kernel_ms matters)
for f in "" tools/abl/*.so; do
  RECO_HIP_LIB=$([ -n "$f" ] && echo $PWD/$f) timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${f:-baseline}', d['roofline']['kernel_ms'])" || exit 1
done
