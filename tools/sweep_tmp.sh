timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in R1 R8 R32 R128; do
if [ $v = R32 ]; then unset VERTICUT_GPU_LIB; else export VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/ab/lib$v.so; fi
echo "== $v 125M"; timeout -k 10 300 python tools/sweep_scan.py 1.25e8 128 1,8,16 0
done; done
for v in R1 R32 R128; do
if [ $v = R32 ]; then unset VERTICUT_GPU_LIB; else export VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/ab/lib$v.so; fi
echo "== $v 1e9"; timeout -k 10 300 python tools/sweep_scan.py 1e9 128 1,6,8,10,16 0
done
