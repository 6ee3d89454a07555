for rep in 1 2 3; do
for b in 8 4 2 16; do
  r=$(VC_SAMPLE_BLOCKS_PER_CU=$b timeout -k 10 300 python bench.py --steps 60 --warmup 5 --cpu-seconds 0 --no-traffic --db-size 1.25e8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/step  scan %.4f ms  fixed %.1f us  %s' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], (j['ms_per_step']-j['roofline']['avg_launch_ms'])*1e3, j['results_check']))") || exit 1
  echo "sample blocks/CU=$b rep$rep  $r"
done
done
