# second-stage bootstrap size vs step time (0 = stage off)
for n in 1.25e8 1e9; do
for s2 in 0 1000000 2000000 4000000 8000000 16000000; do
  VC_SAMPLE2=$s2 python bench.py --db-size $n --cpu-seconds 0 --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n', d['config']['n_codes'], 'sample2 $s2', round(d['value'],1), 'qps', round(d['ms_per_step'],4), 'ms/step  scan', round(d['roofline']['avg_launch_ms'],4))"
done; done
