#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
for v in 1 0; do
  if [ $v = 1 ]; then export DLKA_DW_NO_WL=1; else unset DLKA_DW_NO_WL; fi
  python bench.py --no-cpu-baseline --no-tblock --no-companion --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NO_WL=$v', d['value'], d['ms_per_step'], (d.get('other_dtype') or {}).get('value'))"
done; done
