#!/bin/bash
# one compact bench line per workload of bench.py's table (kernel, us per launch, Msamples/s, fraction of the 8 TB/s roofline, parity), one box
python - <<'PY' > /tmp/wl.txt
import sys; sys.path.insert(0, '.')
import bench
print(" ".join(bench.WORKLOADS.keys()))
PY
for w in $(cat /tmp/wl.txt); do
  python bench.py --workload $w --no-cpu-baseline --no-host-paths --no-n1-reference --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.readline())
print('%-8s %-16s %8.1f us  %9.0f Msamples/s  frac %.3f  parity %s   %s' % ('$w', l['roofline']['kernel'], l['ms_per_step']*1e3, l['value'], l['roofline']['frac'], l.get('parity_full_stream'), l['config']['workload'].split(':',1)[1].split(', ONE')[0].strip()))"
done
