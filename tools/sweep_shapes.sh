VC_SCAN_WRAP=512 timeout -k 10 200 python tools/sweep_scan.py 1e9 128 1,4,8,12,16,32 1024 || exit 1
VC_SCAN_WRAP=4096 timeout -k 10 200 python tools/sweep_scan.py 1e9 128 1,8,16 1024 || exit 1
