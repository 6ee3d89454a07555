#!/bin/bash
# Dev tool (GPU box): bitmap granule width of the radius search (MQ_LO_RADIUS = 7 / 8 / 9 / 10: 16 / 32 / 64 / 128 bytes per
# granule), one build per width, configs[1] through each inside one gpurun call.
OUT=${1:-gpurun_out/granule.txt}
L=$GRAFT_REPO_ROOT/verticut_amd/lib
for lo in 7 8 9 10; do
  VC_BUILD_EXTRA="-DMQ_LO_RADIUS=${lo}u" python -m verticut_amd.build --force > /dev/null 2>&1 || exit 1
  cp $L/libverticut_gpu.so $L/libverticut_gpu_lo$lo.so
done
python -m verticut_amd.build --force > /dev/null 2>&1
for rep in 1 2; do
for lo in 7 8 9 10; do
  r=$(VERTICUT_GPU_LIB=$L/libverticut_gpu_lo$lo.so timeout -k 10 300 python bench.py --workload c2 --tables 2 --cpu-seconds 0 --no-traffic 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f q/s  %.4f ms/step  %s' % (j['value'], j['ms_per_step'], j['results_check']))") || exit 1
  echo "granule lo=$lo rep$rep  $r" | tee -a $OUT
done
done
rm -f $L/libverticut_gpu_lo*.so
