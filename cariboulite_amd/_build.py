"""Build the in-tree native libraries (no JIT cache: the .so files travel with the tree).

    libcariboulite_hip.so   HIP kernels + C-ABI shim, hipcc --offload-arch=gfx950
    libcariboulite_host.so  host C layer with the reference's call surface (gcc)
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)
INC = os.path.join(ROOT, "include")
OBJ = os.path.join(CSRC, "build")
HIP_LIB = os.path.join(PKG, "libcariboulite_hip.so")
HOST_LIB = os.path.join(PKG, "libcariboulite_host.so")
ARCH = "gfx950"


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(INC, "cariboulite_hip.h"))
    jobs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        if force or _newer(obj, [src] + hdrs):
            jobs.append([_hipcc(), "-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17",
                         "-Wall", "-Wno-unused-function", "-fno-slp-vectorize", "-I", INC, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in srcs]
    if force or jobs or _newer(HIP_LIB, objs):
        run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", HIP_LIB] + objs)
    return HIP_LIB


def build_hip_variant(name, defines, source="clhip_rx_pipe.hip", verbose=False):
    """A diagnostic build of the shim with one source recompiled under extra -D flags (timing / energy ablations,
    A/B experiments): abl/<name>/libcariboulite_hip.so, loaded through CLHIP_LIB.  The other objects are the shipped ones."""
    build_hip(False, verbose)
    out_dir = os.path.join(ROOT, "abl", name)
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, source[:-4] + ".o")
    cmd = [_hipcc(), "-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
           "-Wno-unused-but-set-variable", "-fno-slp-vectorize", "-I", INC] + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, source), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    others = [os.path.join(OBJ, f[:-4] + ".o") for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f != source]
    lib = os.path.join(out_dir, "libcariboulite_hip.so")
    subprocess.run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
    return lib


def build_host(force=False, verbose=False):
    hdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hdir):
        return None
    srcs = sorted(os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".c"))
    if not srcs:
        return None
    deps = srcs + [os.path.join(INC, "cariboulite_hip.h")] + \
        [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith((".h", ".inc"))]
    if force or _newer(HOST_LIB, deps + [HIP_LIB]):
        cmd = ["gcc", "-O2", "-g", "-std=gnu11", "-Wall", "-Wextra", "-fPIC", "-shared", "-I", INC,
               "-o", HOST_LIB] + srcs + ["-L", PKG, "-lcariboulite_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return HOST_LIB


CPP_LIB = os.path.join(PKG, "libcariboulite_cpp.so")


def build_cpp(force=False, verbose=False):
    """The C++ API seam (CaribouLiteRadio) over the host C layer."""
    cdir = os.path.join(CSRC, "cpp_api")
    srcs = sorted(os.path.join(cdir, f) for f in os.listdir(cdir) if f.endswith(".cpp"))
    deps = srcs + [os.path.join(cdir, f) for f in os.listdir(cdir) if f.endswith(".hpp")] + [HOST_LIB]
    if force or _newer(CPP_LIB, deps):
        cmd = ["g++", "-O2", "-g", "-std=c++14", "-Wall", "-Wextra", "-fPIC", "-shared", "-I", INC, "-I", cdir,
               "-o", CPP_LIB] + srcs + ["-L", PKG, "-lcariboulite_host", "-lcariboulite_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return CPP_LIB


FANOUT_LIB = os.path.join(PKG, "libcariboulite_fanout.so")


def build_fanout(force=False, verbose=False):
    """The multi-GPU fan-out / fan-in of raw stream buffers over RCCL point-to-point (include/cariboulite_fanout.h)."""
    src = os.path.join(CSRC, "fanout", "clfan.cpp")
    deps = [src, os.path.join(INC, "cariboulite_fanout.h")]
    if force or _newer(FANOUT_LIB, deps):
        cmd = [_hipcc(), "-O2", "-fPIC", "-shared", "-std=c++17", "-Wall", src, "-o", FANOUT_LIB, "-lrccl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return FANOUT_LIB


PCIE_PROBE = os.path.join(ROOT, "tools", "microbench", "pcie_duplex")


def build_pcie_probe(force=False, verbose=False):
    """tools/microbench/pcie_duplex: the box's concurrent H2D + D2H ceiling, which bench.py --pcie and tools/bench_group.py price
    the stream group against (a measuring tool, not part of the product libraries)."""
    src = PCIE_PROBE + ".hip"
    if force or _newer(PCIE_PROBE, [src]):
        cmd = [_hipcc(), "-O3", "-mavx2", "-w", f"--offload-arch={ARCH}", src, "-o", PCIE_PROBE, "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return PCIE_PROBE


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_host(force, verbose)
    build_cpp(force, verbose)
    try:
        build_fanout(force, verbose)
    except (subprocess.CalledProcessError, OSError) as e:
        # only the multi-GPU fan-out (shard.fanout_streams) needs RCCL: a box without its headers or library still gets
        # the data path; cariboulite_amd.fanout raises ImportError when someone asks for the missing library
        import warnings
        warnings.warn(f"libcariboulite_fanout.so not built ({e}); the RCCL fan-out is unavailable")
    try:
        build_pcie_probe(force, verbose)
    except (subprocess.CalledProcessError, OSError) as e:
        import warnings
        warnings.warn(f"tools/microbench/pcie_duplex not built ({e}); bench.py --pcie reports no roof")


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv, verbose=True)
