"""Build libadaprompt_hip.so (gfx950) in-tree with hipcc.  No cmake, no JIT cache:
the .so sits next to the sources and travels to the GPU box with the repo snapshot."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libadaprompt_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "adaprompt_hip.h")
SOURCES = ["capi.hip", "conv_gemm.hip", "norms.hip", "attention.hip", "misc.hip", "optim.hip", "wgrad.hip", "vae.hip", "comm.hip", "regloss.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value"]


def _stamp():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".h")):
            continue
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    with open(HEADER, "rb") as fh:                    # the public C ABI is part of what the library was built from
        h.update(b"adaprompt_hip.h")
        h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_current():
    """the library exists and was built from exactly the sources / header / flags on disk."""
    stamp_file = OUT + ".stamp"
    return os.path.exists(OUT) and os.path.exists(stamp_file) and open(stamp_file).read() == _stamp()


def build(force=False, verbose=True):
    stamp_file = OUT + ".stamp"
    stamp = _stamp()
    if not force and is_current():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(7, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
