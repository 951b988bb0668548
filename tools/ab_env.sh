#!/bin/bash
# same-box A/B of one environment switch on the bench loop: ab_env.sh VAR "values" [bench args]; alternates the values twice
V="$1"; VALS="$2"; shift 2
for rep in 1 2; do for x in $VALS; do
  env $V=$x python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gen "$@" 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$V=$x', j['value'], j['per_lesson_ms'], j.get('replay'))"
done; done
