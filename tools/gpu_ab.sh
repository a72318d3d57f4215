#!/bin/bash
# same-box A/B of the attention kernel choice at B=32 and B=8 (whole sample())
for w in 0 -1 0 -1; do
  python tools/bench_flags.py wide=$w -- --batch 32 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=32 wide=$w', round(d['ms_per_step'],1))"
done
for w in 0 -1; do
  python tools/bench_flags.py wide=$w -- --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 wide=$w', round(d['ms_per_step'],1))"
done
