#!/bin/bash
# throughput against an artificial host delay per C-ABI call (tools/host_slack.py): usage host_slack.sh "<delays>" [bench args]
D="$1"; shift
for d in $D; do
  python tools/host_slack.py $d --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('delay_us $d', j['value'], j['ms_per_step'], j['per_lesson_ms'])"
done
