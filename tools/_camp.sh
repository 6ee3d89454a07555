mkdir -p gpurun_out/r3camp
timeout -k 10 1100 python tests/campaign/parity_campaign.py 1500 0 > gpurun_out/r3camp/linear_0.txt 2>&1; tail -2 gpurun_out/r3camp/linear_0.txt
