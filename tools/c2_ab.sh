#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  r=$(env $v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['c2_fullrank_d512']; print('%.1f us' % c['whole_evaluation']['us_per_eval'], {k:round(x['avg_kernel_us'],1) for k,x in c['per_kernel'].items()})")
  echo "[$v]: $r"
done
