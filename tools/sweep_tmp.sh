run() { python bench.py --steps 60 --warmup 10 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps %.0f  step %.4f ms  scan %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for rep in 1 2 3; do
for s2 in default 16000000 32000000 64000000; do
  if [ $s2 = default ]; then unset VC_SAMPLE2; else export VC_SAMPLE2=$s2; fi
  echo "== 1e9 sample2=$s2"; run
done; done
