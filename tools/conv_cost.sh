#!/bin/bash
# Dev tool (GPU box): does the threshold's convergence cost verify time?  Default bootstrap against an exact threshold
# (VC_SAMPLE1 = n) at 1e9, where the 256 MB the sampling pass leaves in the Infinity Cache are 1.6 % of the database
# (at 1.25e8 they are 12 % and fake a 13 % gain: profiles/r02_sweeps.md).
for rep in 1 2; do
for s in default 1000000000; do
  if [ $s = default ]; then ev="VC_X=0"; else ev="VC_SAMPLE1=$s"; fi
  x=$(env $ev timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-traffic --db-size 1e9 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/step  scan %.4f ms  %s' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['results_check']))") || exit 1
  echo "rep$rep sample=$s  $x"
done
done
