for shape in 4,256,2 4,256,3 2,256,3 2,256,2; do
  VC_SCAN_SHAPE=$shape timeout -k 10 200 python tools/sweep_scan.py 1e9 128 1,6,8,10,12 0 || exit 1
done
