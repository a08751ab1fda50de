#!/bin/bash
# usage: tools/fr_ab.sh "<env assignments A>" "<env assignments B>" ...   (GPU box): headline us/eval of each variant, 2 rounds
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  r=$(env $v python bench.py --no-legs --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us  %.0f evals/s' % (1e3*d['ms_per_step'], d['value']))")
  echo "round $round [$v]: $r"
done; done
