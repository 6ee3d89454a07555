#!/bin/bash
# Dev tool (GPU box): verify-kernel time against database size: t = a + b N  (a = ramp + tail + first-tile flood)
for q in 1 8; do for n in 6.25e7 1.25e8 2.5e8 5e8 1e9; do
python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-traffic --db-size $n --queries $q 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); n=j['config']['n_codes']; t=j['roofline']['avg_launch_ms']; print('qt=$q n=%-11d scan %.4f ms  %.1f ns per 1e3 codes  step-scan %.1f us' % (n, t, t*1e6/(n/1e3), (j['ms_per_step']-t)*1e3))"
done; done
