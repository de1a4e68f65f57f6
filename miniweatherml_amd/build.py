"""Builds libmw_cdna4.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m miniweatherml_amd.build [--force]

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmw_cdna4.so")
SOURCES = ["mw_host.cpp", "mw_dycore.hip", "mw_kessler.hip", "mw_mlp.hip", "mw_column.hip", "mw_output.hip", "mw_netcdf.cpp", "mw_rccl.cpp", "mw_h5.cpp"]
HEADERS = ["mw_common.h", "mw_weno.h", "mw_weno79.h", "mw_march.h", "mw_calib.h", "mw_glibc_pow.h", "mw_glibc_pow_tables.h", os.path.join("..", "..", "include", "mw_cdna4.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=on", "-I/opt/rocm/include"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for s in srcs:
        o = os.path.splitext(s)[0] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        # RCCL is NOT linked: mw_rccl.cpp resolves it at run time from the librccl already mapped in the process (one RCCL)
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_examples(verbose=True):
    """The C++ callers of examples/ (mirrors of the reference's drivers) against the C++ facade + C ABI."""
    root = os.path.dirname(HERE)
    exes = []
    for name in ("supercell_driver", "simple_city_driver", "inference_ponni_driver", "supercell_multirank"):
        src = os.path.join(root, "examples", name + ".cpp")
        exe = os.path.join(root, "examples", name)
        deps = [src, os.path.join(HERE, "host", "mw_facade.h"), os.path.join(HERE, "host", "mw_ponni.h"), os.path.join(root, "include", "mw_cdna4.h"), LIB]
        if _stale(exe, deps):
            cmd = [HIPCC, "-O2", "-std=c++17", "-x", "c++", src, "-o", exe, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                   "-L" + HERE, "-lmw_cdna4", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN/../miniweatherml_amd",
                   "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        exes.append(exe)
    # a module with its own HIP kernels over core::MultiField (MultipleFields.h:10-96): compiled as HIP for gfx950
    src = os.path.join(root, "examples", "multifield_module.cpp")
    exe = os.path.join(root, "examples", "multifield_module")
    if _stale(exe, [src, os.path.join(HERE, "host", "mw_facade.h"), os.path.join(root, "include", "mw_cdna4.h"), LIB]):
        cmd = [HIPCC, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", src, "-o", exe, "-I/opt/rocm/include", "-L" + HERE, "-lmw_cdna4",
               "-Wl,-rpath,$ORIGIN/../miniweatherml_amd", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    exes.append(exe)
    return exes[0]


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_examples()
