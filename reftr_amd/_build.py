"""Builds libreftr_hip.so (the C-ABI kernel library declared in include/reftr_hip.h) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container as well as on the
MI355X box.  The .so lives IN-TREE (reftr_amd/libreftr_hip.so): it is git-ignored but travels with
the gpurun snapshot, and the round-end check sees it among the loaded shared objects.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
# REFTR_LAB=1 selects the LAB library: the same sources compiled with -DRT_LAB, in which the kernels' tuning switches (RT_TUNE in
# csrc/rt_common.h: tile heuristics, split targets, ablation probes ...) are read from the environment.  The product library fixes
# every one of them at its measured-best value and cannot be re-tuned (or put into a wrong-results probe mode) from outside.
# The lab library also carries every tile / stage / schedule variant of rt_conv_gemm that was built and measured but is not chosen by the
# product heuristics (reachable through tile_hint: tests/test_gemm_gpu.py, benchmarks/tile_sweep.py); the product library instantiates
# only what it can launch.
LAB = os.environ.get("REFTR_LAB", "0") == "1"
ARCH = "gfx950"


def _paths(lab):
    return (os.path.join(_HERE, "build", "obj_lab" if lab else "obj"),
            os.path.join(_HERE, "libreftr_hip_lab.so" if lab else "libreftr_hip.so"))


OBJ_DIR, LIB_PATH = _paths(LAB)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-inline-asm"]



def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(d) for d in deps)


def _compile_one(src, hdr_mtime, force, lab):
    obj = os.path.join(_paths(lab)[0], os.path.basename(src)[:-4] + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= hdr_mtime):
        return obj, False
    cmd = [HIPCC] + FLAGS + (["-DRT_LAB"] if lab else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=False, lab=None):
    """Compile every csrc/*.hip for gfx950 and link libreftr_hip.so (lab=True: libreftr_hip_lab.so; None: what REFTR_LAB selects).
    Returns the library path."""
    lab = LAB if lab is None else bool(lab)
    OBJ_DIR, LIB_PATH = _paths(lab)
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    hdr_mtime = _deps_mtime()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile_one(s, hdr_mtime, force, lab), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(ch for _, ch in results)
    stale = os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs)   # interrupted link
    if rebuilt or stale or not os.path.exists(LIB_PATH) or force:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[reftr_amd] {LIB_PATH} ({'rebuilt' if rebuilt or stale else 'up to date'}, {len(objs)} objects)")
    return LIB_PATH


def build_id():
    """sha256[:16] over the kernel sources, the C-ABI header and the package's Python files: identifies the build a
    measurement artefact (profiles/*_pmc_traffic.json) was taken on.  (The GPU box has no .git, so the commit cannot be used.)"""
    import hashlib
    h = hashlib.sha256()
    files = []
    for d, _, fs in os.walk(_HERE):
        if os.sep + "build" in d or "__pycache__" in d:
            continue
        files += [os.path.join(d, f) for f in fs if f.endswith((".hip", ".h", ".py"))]
    files += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    for f in sorted(files):
        h.update(os.path.relpath(f, os.path.dirname(_HERE)).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, lab=True if "--lab" in sys.argv else None)
