"""Build libverticut_gpu.so (gfx950 only) in-tree with hipcc.  `python -m verticut_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libverticut_gpu.so")
SOURCES = ["vc_scan.hip", "vc_mih.hip", "vc_sort.hip", "vc_engine.hip", "vc_sharded.hip"]
HEADERS = ["vc_common.hpp", "vc_internal.hpp", "vc_mih.hpp", os.path.join("..", "..", "include", "verticut_gpu.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-pass-failed"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libverticut_gpu.so cannot be built")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(HERE, "host", f) for f in ("verticut_host.hpp", "distributed_image_search.cc", "accuracy_test.cc")]
    if not all(os.path.exists(os.path.join(HERE, "bin", b)) for b in ("distributed-image-search", "accuracy-test")):
        return True
    return any(os.path.getmtime(d) > t for d in deps)


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def kernel_resources(obj):
    """{kernel name: {private_segment_fixed_size, sgpr_spill_count, vgpr_spill_count, vgpr_count, sgpr_count}} of the
    gfx950 code object inside a hipcc object file: .hip_fatbin section -> offload bundle -> AMDGPU metadata note."""
    import tempfile
    tools = {t: os.path.join(LLVM_BIN, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")}
    for t, path in tools.items():
        if not os.path.exists(path):
            raise RuntimeError("%s not found: cannot inspect the code object of %s" % (path, obj))
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "gfx950.co")
        subprocess.check_call([tools["llvm-objcopy"], "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(td, "copy.o")])
        subprocess.check_call([tools["clang-offload-bundler"], "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        notes = subprocess.check_output([tools["llvm-readelf"], "--notes", co]).decode(errors="replace")
    # the note is YAML: "amdhsa.kernels:" holds one "  - .agpr_count: ..." record per kernel, keys at four spaces
    res, rec, in_kernels = {}, None, False
    keys = ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "sgpr_count")

    def close(r):
        if r and "name" in r:
            res[r.pop("name")] = r

    for line in notes.splitlines():
        if not line.startswith(" "):
            close(rec)
            rec, in_kernels = None, line.startswith("amdhsa.kernels:")
            continue
        if not in_kernels:
            continue
        if line.startswith("  - "):
            close(rec)
            rec, line = {}, "    " + line[4:]
        if rec is None or not line.startswith("    .") :
            continue
        key, _, val = line[5:].partition(":")
        if key == "name":
            rec["name"] = val.strip().strip("'\"")
        elif key in keys:
            rec[key] = int(val)
    close(rec)
    return res


def check_scan_code_objects(obj=None):
    """The verify kernel issues its tile loads by hand (inline asm, vc_scan.hip): correctness depends on hipcc never
    spilling or moving a tile register between the load and vc_tile_wait, so EVERY vc_scan_kernel instantiation must
    have no scratch and no spills.  Raises otherwise; returns the number of instantiations checked."""
    obj = obj or os.path.join(LIBDIR, "vc_scan.o")
    res = kernel_resources(obj)
    scan = {k: v for k, v in res.items() if "vc_scan_kernel" in k}
    if not scan:
        raise RuntimeError("no vc_scan_kernel instantiation found in %s (metadata layout changed?)" % obj)
    bad = {k: v for k, v in scan.items()
           if v.get("private_segment_fixed_size", 1) != 0 or v.get("vgpr_spill_count", 1) != 0 or v.get("sgpr_spill_count", 1) != 0}
    if bad:
        raise RuntimeError("vc_scan_kernel must not use scratch or spill next to its hand-issued loads: %s" % bad)
    return len(scan)


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        extra = ["-DVC_SCAN_DIAGNOSTICS=1"] if os.environ.get("VC_BUILD_DIAG") == "1" else []
        extra += os.environ.get("VC_BUILD_EXTRA", "").split()   # dev experiments, e.g. -DVC_SCAN_NT=0
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on %s" % src)
    if os.environ.get("VC_BUILD_EXTRA") is None and os.environ.get("VC_BUILD_DIAG") != "1":
        check_scan_code_objects()      # product build only: dev experiments may spill on purpose
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    build_host_tools()
    return LIB


HOST = os.path.join(HERE, "host")
BINDIR = os.path.join(HERE, "bin")
DRIVER = os.path.join(BINDIR, "distributed-image-search")


def build_host_tools():
    """C++ host layer above the C ABI (plain g++): the reference-shaped query driver."""
    os.makedirs(BINDIR, exist_ok=True)
    cxx = shutil.which("g++") or "g++"
    for exe, src in ((DRIVER, "distributed_image_search.cc"), (os.path.join(BINDIR, "accuracy-test"), "accuracy_test.cc")):
        subprocess.check_call([cxx, "-O2", "-std=c++14", "-Wall", "-o", exe, os.path.join(HOST, src), "-I", HOST,
                               "-L", LIBDIR, "-lverticut_gpu", "-Wl,-rpath,$ORIGIN/../lib",
                               "-Wl,-rpath-link," + "/opt/rocm/lib", "-L/opt/rocm/lib"])
    return DRIVER


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
