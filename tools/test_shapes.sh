#!/bin/bash
# Dev tool: the linear-path parity tests under every selectable verify-kernel shape (VC_SCAN_SHAPE=U,BLK,NB).
for shape in 4,256,3 4,256,1 2,256,2 2,256,3 1,256,2 4,512,2 2,512,3; do
  echo "== VC_SCAN_SHAPE=$shape"
  VC_SCAN_SHAPE=$shape timeout -k 10 200 python -m pytest tests/test_linear_gpu.py tests/test_fixtures_gpu.py tests/test_edge_gpu.py -x -q 2>&1 | tail -2
done
