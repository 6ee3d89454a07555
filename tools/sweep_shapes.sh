VC_BUILD_EXTRA=-DVC_SCAN_NT=0 python -m verticut_amd.build --force > /dev/null 2>&1 || exit 1
echo "== default-policy loads (no nt)"
timeout -k 10 200 python tools/sweep_scan.py 1e9 128 1,8,12 0 || exit 1
python -m verticut_amd.build --force > /dev/null 2>&1 || exit 1
echo "== nt loads"
timeout -k 10 200 python tools/sweep_scan.py 1e9 128 1,8,12 0 || exit 1
