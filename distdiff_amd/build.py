"""Builds libdistdiff_hip.so (hand-written gfx950 kernels + engine + C ABI) in-tree with hipcc.

    python -m distdiff_amd.build            # incremental
    python -m distdiff_amd.build --force

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels with gpurun.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# DD_BUILD_OBJ / DD_BUILD_LIB: a variant build beside the product one (A/B of compile-time switches: DD_EXTRA_CFLAGS=-D... ; run with DD_LIB=)
OBJ = os.environ.get("DD_BUILD_OBJ") or os.path.join(HERE, "csrc", "_obj")
LIB = os.environ.get("DD_BUILD_LIB") or os.path.join(HERE, "libdistdiff_hip.so")
SOURCES = ["conv_gemm.hip", "conv_gemm2.hip", "conv_halo.hip", "gemm_ws.hip", "norm.hip", "attention.hip", "attention_shortk.hip", "attention_gemm.hip", "elementwise.hip", "guide_f32.hip", "weights.cpp", "ops_abi.cpp",
           "engine_weights.cpp", "engine_graph.cpp", "engine_exec.cpp", "engine.cpp"]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs. With the default heuristic the attention kernels put them in AccVGPRs
# and paid 144 v_accvgpr_read/write per KV tile to run the softmax on them (found in the ISA; attention family 475 -> see DESIGN.md)
FLAGS = ["--offload-arch=gfx950", "-O3", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + os.environ.get("DD_EXTRA_CFLAGS", "").split() + ["-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-result",
         "-I", os.path.join(HERE, "..", "include")]


# per-file switches.  gemm_ws.hip: the SLP vectoriser turns the GEGLU / LayerNorm-fold epilogue into v_pk_fma_f32 / v_pk_mul_f32, which
# cost more issue time beside MFMAs than the scalar pairs they replace (MI355X guide, cycle constants)
PER_FILE = {"gemm_ws.hip": ["-fno-slp-vectorize"]}


def _newer(src, dst):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        if force or _newer(src, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + PER_FILE.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print("[build]", os.path.basename(src), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in srcs]
    if jobs or force or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
        if verbose:
            print("[build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
