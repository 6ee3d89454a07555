"""Build-time guard of the verify kernel's hand-issued loads (verticut_amd/build.py): every vc_scan_kernel instantiation
must come out of hipcc with no scratch and no register spills -- a spill between the inline-asm global_load and
vc_tile_wait would read a register whose load has not landed.  Runs on the CPU: hipcc cross-compiles."""
import os
import subprocess

import pytest


def test_shipped_scan_kernels_have_no_scratch_and_no_spills(vc):
    from verticut_amd import build as vb
    obj = os.path.join(vb.LIBDIR, "vc_scan.o")
    if not os.path.exists(obj):
        vb.build(force=True)
    n = vb.check_scan_code_objects(obj)
    assert n >= 40                                   # W x tile shapes x small-tile forms
    res = vb.kernel_resources(obj)
    assert all(set(v) >= {"private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count"} for v in res.values())


def test_guard_rejects_a_scan_kernel_with_scratch(tmp_path):
    from verticut_amd import build as vb
    src = tmp_path / "bad.hip"
    src.write_text('''#include <hip/hip_runtime.h>
template <int W> __global__ void vc_scan_kernel(const unsigned* in, unsigned* out, unsigned n) {
  unsigned a[256];                                    // dynamically indexed local array: lives in scratch
  for (unsigned i = 0; i < 256; ++i) a[i] = in[(i * 7 + threadIdx.x) % n];
  unsigned s = 0;
  for (unsigned i = 0; i < n; ++i) s += a[(in[i] + threadIdx.x) & 255];
  out[threadIdx.x] = s;
}
template __global__ void vc_scan_kernel<2>(const unsigned*, unsigned*, unsigned);
''')
    obj = tmp_path / "bad.o"
    subprocess.check_call([vb._hipcc(), "-O3", "--offload-arch=gfx950", "-fPIC", "-c", str(src), "-o", str(obj)])
    res = vb.kernel_resources(str(obj))
    assert any(v["private_segment_fixed_size"] > 0 for v in res.values())
    with pytest.raises(RuntimeError, match="scratch"):
        vb.check_scan_code_objects(str(obj))
