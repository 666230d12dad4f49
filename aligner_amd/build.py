"""Builds aligner_amd/lib/libaligner_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the dev container; the built .so travels to the GPU box
with the repo snapshot.  `python -m aligner_amd.build [--force] [--remarks]`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libaligner_hip.so")
SOURCES = ["aln_kernels.hip", "aln_host.hip"]
# aln_kernels.hip is compiled as several translation units side by side (-DALN_TU=<mask of its ALN_PART_* families>): the fast
# core-local batch kernel alone is half of the compile time
KERNEL_UNITS = [("generic", 1), ("fast_cl", 2), ("fast_rest", 4), ("single", 8), ("tb", 16), ("fast_cl_solo", 32), ("fast_rest_solo", 64)]
HEADERS = ["aln_device.h", "aln_fast.h", "aln_single_unit.inc", os.path.join("..", "..", "include", "aligner_hip.h")]
# host-only helper of the synthetic workloads (splitmix64 residues; aligner_amd/workloads.py only LOADS it)
SYNTH_LIB = os.path.join(LIBDIR, "libaln_synth.so")
SYNTH_SRC = os.path.join(CSRC, "aln_synth.c")
# -ffp-contract=off: the f64 kernels must be the reference's add/sub/max/compare, never an fma
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_synth(force=False):
    if force or not os.path.exists(SYNTH_LIB) or os.path.getmtime(SYNTH_SRC) > os.path.getmtime(SYNTH_LIB):
        os.makedirs(LIBDIR, exist_ok=True)
        tmp = SYNTH_LIB + ".tmp%d" % os.getpid()
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-o", tmp, SYNTH_SRC])
        os.replace(tmp, SYNTH_LIB)
    return SYNTH_LIB


# tests/abi_harness.c: include/aligner_hip.h compiled as C99 and linked against the library (the check that the header itself is a
# usable C interface; run on the GPU box by tests/test_gpu_parity.py)
HARNESS_SRC = os.path.join(HERE, "..", "tests", "abi_harness.c")
HARNESS = os.path.join(HERE, "..", "tests", "bin", "abi_harness")


def build_harness(force=False):
    src, out = os.path.abspath(HARNESS_SRC), os.path.abspath(HARNESS)
    hdr = os.path.join(HERE, "..", "include", "aligner_hip.h")
    if force or not os.path.exists(out) or max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(LIB)) > os.path.getmtime(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = out + ".tmp%d" % os.getpid()
        subprocess.check_call([os.environ.get("CC", "gcc"), "-std=c99", "-Wall", "-Werror", "-I", os.path.join(HERE, "..", "include"), src, "-o", tmp,
                               "-L", LIBDIR, "-laligner_hip", "-Wl,-rpath,$ORIGIN/../../aligner_amd/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
        os.replace(tmp, out)
    return out


def build(force=False, remarks=False):
    lib = _build_lib(force, remarks)
    build_harness(force)
    return lib


def _build_lib(force=False, remarks=False):
    build_synth(force)
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    # ALN_CXXFLAGS: extra flags for an instrumented build (-DALN_STAMPS: per-pair time stamps for tools/tail_timeline.py)
    base = [hipcc] + [f for f in FLAGS if f != "-shared"] + os.environ.get("ALN_CXXFLAGS", "").split() + \
        (["-Rpass-analysis=kernel-resource-usage"] if remarks else [])
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = [(base + ["-DALN_TU=%d" % mask, "-c", os.path.join(CSRC, "aln_kernels.hip"), "-o", os.path.join(objdir, "aln_kernels_%s.o" % name)])
            for name, mask in KERNEL_UNITS]
    jobs.append(base + ["-c", os.path.join(CSRC, "aln_host.hip"), "-o", os.path.join(objdir, "aln_host.o")])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
        for rc, cmd in zip(pool.map(subprocess.call, jobs), jobs):
            if rc != 0:
                raise subprocess.CalledProcessError(rc, cmd)
    tmp = LIB + ".tmp%d" % os.getpid()
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [j[-1] for j in jobs] + ["-o", tmp])
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, remarks="--remarks" in sys.argv))
