#!/bin/bash
# quick bench sweep on the GPU box (no profiler): tools/sweep.sh "<engines list>" "<batch list>" ["<VB_MF_TARGET_WG list>"]
cd $GRAFT_REPO_ROOT
for w in ${3:-512}; do for e in ${1:-1 2 3}; do for b in ${2:-1 4 16}; do
  VB_MF_TARGET_WG=$w python bench.py --steps 4000 --warmup 400 --engines $e --batch $b --no-cpu-baseline --no-fullrank 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('wg=$w engines=$e batch=$b value=%.0f us/step=%.2f sync=%.0f k1_us/launch=%.2f evals/launch=%.1f GB/s=%.0f frac=%.3f' % (d['value'], 1e3*d['ms_per_step'], d['sync_call_evals_per_s'], r['avg_kernel_us'], r['evals_per_launch'], r['achieved'], r['frac']))"
done; done; done
