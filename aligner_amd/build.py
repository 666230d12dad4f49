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


def build(force=False, remarks=False):
    build_synth(force)
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    # ALN_CXXFLAGS: extra flags for an instrumented build (-DALN_STAMPS: per-pair time stamps for tools/tail_timeline.py)
    cmd = [hipcc] + FLAGS + os.environ.get("ALN_CXXFLAGS", "").split() + (["-Rpass-analysis=kernel-resource-usage"] if remarks else []) + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, remarks="--remarks" in sys.argv))
