"""Builds libsr3hip.so (gfx950 only) in-tree with hipcc. No torch involved.

    python -m build            (from inside the package directory)   or   build_library()
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsr3hip.so")
SOURCES = ["sr3_api.hip", "kernels_conv.hip", "kernels_conv_ws.hip", "kernels_misc.hip", "kernels_edge.hip", "kernels_pre.hip", "kernels_post.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-source flags: the fp32 VALU contraction of kernels_edge.hip must stay scalar v_fma_f32 (the SLP
# vectoriser packs it into v_pk_fma_f32 pairs: register-pair shuffles, 2 KB of scratch per lane)
EXTRA_FLAGS = {"kernels_edge.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(out: str, deps: list[str]) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    """experiments=True builds libsr3hip_exp.so with -DSR3_EXPERIMENTS: the timing switches of
    tools/conv_bench.py (SR3_CONV_DBG) exist only there, never in the product library."""
    hipcc = _hipcc()
    # SR3_EXP_TAG / SR3_EXP_DEFINES (experiments only): several variant libraries side by side, e.g.
    #   SR3_EXP_TAG=v1 SR3_EXP_DEFINES="-DSR3_GA_VARIANT=1" python build.py --experiments  -> libsr3hip_exp_v1.so
    tag = os.environ.get("SR3_EXP_TAG", "") if experiments else ""
    extra = os.environ.get("SR3_EXP_DEFINES", "").split() if experiments else []
    objdir = os.path.join(HERE, ("build_exp" + ("_" + tag if tag else "")) if experiments else "build")
    lib = os.path.join(HERE, "libsr3hip_exp%s.so" % ("_" + tag if tag else "")) if experiments else LIB
    flags = FLAGS + (["-DSR3_EXPERIMENTS"] + extra if experiments else [])
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "sr3_internal.h"),
               os.path.join(HERE, "..", "include", "sr3hip.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            jobs.append([hipcc, *flags, *EXTRA_FLAGS.get(src, []), "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
