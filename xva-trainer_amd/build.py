"""Build libxvahip.so (all HIP kernels + the C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU container and the resulting
.so travels to the GPU box with the repo snapshot.  Objects are rebuilt only when a source
or header is newer (plain mtime check; there is no cmake/ninja dependency).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libxvahip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"] + os.environ.get("XVA_CFLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    m = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


# translation units that #include another .hip (a second instantiation set of the same kernels)
INCLUDES = {"gemm_glds_f16.hip": ["gemm_glds.hip"]}


def _compile(src, verbose, force=False):
    obj = os.path.join(CSRC, "build", src[:-4] + ".o")
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    spath = os.path.join(CSRC, src)
    newest = max([os.path.getmtime(spath), _headers_mtime()] + [os.path.getmtime(os.path.join(CSRC, f)) for f in INCLUDES.get(src, [])])
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest:
        return obj, False
    cmd = ["hipcc"] + FLAGS + ["-I", INCLUDE, "-c", spath, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def build(verbose=True, force=False):
    """force: a from-source rebuild — every object is recompiled and the library relinked (`python xva-trainer_amd/build.py --force`)."""
    if force and os.path.exists(LIB):
        os.remove(LIB)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, verbose, force), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
