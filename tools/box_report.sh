#!/bin/bash
# what kind of box is this? CPU model / cores / load next to a bench line's clocks, per-lesson times and kernel efficiency
lscpu | grep -E "Model name|^CPU\(s\)|MHz|NUMA node\(s\)" | head -6
cat /proc/loadavg
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gen 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('steps/s', j['value'], 'sclk', j['clocks']['sclk_mhz']['mean'], 'power', j['clocks']['power_w']['mean'], 'frac', j['roofline']['frac'], 'gd', j['roofline']['gd_conv_stack']['frac'], 'minnec', j['minimum_necessary']['value'])
print(j['per_lesson_ms'])"
python tools/host_time.py 2>/dev/null | tail -9
