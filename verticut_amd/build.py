"""Build libverticut_gpu.so (gfx950 only) in-tree with hipcc.  `python -m verticut_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libverticut_gpu.so")
SOURCES = ["vc_scan.hip", "vc_mih.hip", "vc_sort.hip", "vc_engine.hip"]
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
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
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
