L=$GRAFT_REPO_ROOT/verticut_amd/lib
for lo in 7 8; do
  VC_BUILD_EXTRA="-DMQ_LO_KNN=${lo}u" python -m verticut_amd.build --force > /dev/null 2>&1 || exit 1
  cp $L/libverticut_gpu.so $L/libverticut_gpu_klo$lo.so
done
python -m verticut_amd.build --force > /dev/null 2>&1
for rep in 1 2; do
for lo in 7 8; do
  r=$(VERTICUT_GPU_LIB=$L/libverticut_gpu_klo$lo.so timeout -k 10 300 python bench.py --workload knn_mih --cpu-seconds 0 --no-traffic 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f q/s  %.4f ms/step  %s' % (j['value'], j['ms_per_step'], j['results_check']))") || exit 1
  echo "knn granule lo=$lo rep$rep  $r"
done
done
rm -f $L/libverticut_gpu_klo*.so
