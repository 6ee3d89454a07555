#!/bin/bash
# dev tool: build libverticut_gpu.so variants with extra -D flags for same-box A/B runs
#   tools/build_variant.sh NAME "-DMW_WAVES=4 -DMW_G=2u"   ->  verticut_amd/lib/variants/libvc_NAME.so  (select with VERTICUT_GPU_LIB)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=verticut_amd/lib/variants; mkdir -p $out/obj_$name
pids=()
for f in vc_scan vc_mih vc_sort vc_engine vc_sharded; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-pass-failed $@ -c verticut_amd/csrc/$f.hip -o $out/obj_$name/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libvc_$name.so $out/obj_$name/*.o -ldl
rm -rf $out/obj_$name
echo $out/libvc_$name.so
