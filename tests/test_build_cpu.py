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


def test_shipped_query_kernels_of_the_configs_shapes_have_no_scratch(vc):
    """mih_query_kernel for 64- and 128-bit codes (BASELINE configs[1] and the exact / approximate k-NN of configs[2]) is
    compiled for 4 waves per SIMD and sits at the 128-VGPR limit: a change that tips it over shows up
    here as scratch, not as a slower bench line three steps later.  SGPR spills (to VGPR lanes) are bounded, not zero."""
    from verticut_amd import build as vb
    obj = os.path.join(vb.LIBDIR, "vc_mih.o")
    if not os.path.exists(obj):
        vb.build(force=True)
    res = {k: v for k, v in vb.kernel_resources(obj).items() if "mih_query_kernel" in k}
    assert len(res) == 12                             # W = 1, 2, 4, 8 x {radius granules, k-NN granules, k-NN over directory lines}
    narrow = {k: v for k, v in res.items() if "ILi1E" in k or "ILi2E" in k}
    assert len(narrow) == 6
    for k, v in narrow.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0, (k, v)
        assert v["vgpr_count"] <= 128 and v["sgpr_spill_count"] <= 64, (k, v)   # (to the lanes of ONE VGPR; scratch and VGPR spills must stay zero)
    stream = {k: v for k, v in vb.kernel_resources(obj).items() if "mih_bucket_stream_kernel" in k}
    assert stream and all(v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 for v in stream.values())
