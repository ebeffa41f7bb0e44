"""Build libadas_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python vehicle-cv-adas_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  The bit-exact post-processing units are compiled with -ffp-contract=off.
"""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# ADAS_BUILD_TAG=<tag>: a scratch build (instrumented / experimental flags through ADAS_CFLAGS) beside the product library --
# _scratch/libadas_hip_<tag>.so with its own object directory; tools load it through ADAS_LIB=
TAG = os.environ.get("ADAS_BUILD_TAG", "")
OUT = os.path.join(HERE, "_scratch", f"libadas_hip_{TAG}.so") if TAG else os.path.join(HERE, "libadas_hip.so")
OBJ = os.path.join(HERE, "_scratch", f"_obj_{TAG}") if TAG else os.path.join(HERE, "_obj")
ARCH = "gfx950"

# (source, extra flags)
UNITS = [
    ("api_common.cpp", []),
    ("post_kernels.hip", ["-ffp-contract=off"]),
    ("pre_kernels.hip", ["-ffp-contract=off"]),
    ("conv_kernels.hip", []),
    ("conv_x3.hip", []),
    ("conv_halo.hip", []),
    ("conv_ml.hip", []),
    ("conv_halo_rw.hip", []),
    ("conv_halo_s2.hip", []),
    ("conv_halo8.hip", []),
    ("conv_halo8_x3.hip", []),
    ("conv_pw_x3.hip", []),
    ("conv_pair.hip", []),
    ("conv_c2f.hip", []),
    ("conv_c2f_x3.hip", []),
    ("conv_fc.hip", []),
    ("conv_pw.hip", []),
    ("conv_pwg.hip", []),
    ("conv_stem.hip", []),
    ("conv_stem_x3.hip", []),
    ("aux_kernels.hip", []),
    ("dw_attn.hip", []),
    ("fuse_ops.hip", []),
    ("engine.cpp", []),
    ("pipeline.cpp", []),
]
EXTRA = os.environ.get("ADAS_CFLAGS", "").split()   # scratch builds only (e.g. -DADAS_HALO_PROF)
COMMON = EXTRA + ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-Wno-unused-value", "-x", "hip"]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "adas_hip.h"))
    return hdrs


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _deps())
    objs, todo = [], []
    for src, extra in UNITS:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t):
            todo.append([hipcc, *COMMON, *extra, "-c", sp, "-o", op])
    rebuilt = bool(todo)
    if todo:   # the units are independent: compile them side by side (ADAS_BUILD_JOBS, default = min(8, cores))
        from concurrent.futures import ThreadPoolExecutor
        jobs = max(1, int(os.environ.get("ADAS_BUILD_JOBS", min(8, os.cpu_count() or 1))))

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(jobs) as ex:
            list(ex.map(run, todo))
    if rebuilt or not os.path.exists(OUT):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        # the link step accepts undefined symbols in a shared object (hipcc once dropped the host stubs of a kernel template without a
        # diagnostic): loading the library is the check that every symbol resolves
        import ctypes
        ctypes.CDLL(OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
