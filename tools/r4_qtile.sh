cd $GRAFT_REPO_ROOT
L=$PWD/verticut_amd/lib/variants/libvc_qt16k.so
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1  %.0f q/s  step %.4f ms  kernel %.4f ms  %s' % (j['value'], j['ms_per_step'], r.get('avg_launch_ms') or 0, j.get('results_check')))"; }
for n in 1e8 1e9; do
for rep in 1 2; do
  python3 bench.py --workload knn_mih --db-size $n --steps 24 --warmup 4 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | line "n=$n tile 4096 Q=4096 rep$rep"
  VERTICUT_GPU_LIB=$L python3 bench.py --workload knn_mih --db-size $n --queries 8192 --steps 12 --warmup 2 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | line "n=$n tile 16384 Q=8192 rep$rep"
  VERTICUT_GPU_LIB=$L python3 bench.py --workload knn_mih --db-size $n --queries 16384 --steps 6 --warmup 2 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | line "n=$n tile 16384 Q=16384 rep$rep"
  python3 bench.py --workload knn_mih --db-size $n --queries 16384 --steps 6 --warmup 2 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | line "n=$n tile 4096 Q=16384 rep$rep"
done; done
