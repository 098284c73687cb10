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
SOURCES = ["capi.hip", "conv_gemm.hip", "norms.hip", "attention.hip", "misc.hip", "optim.hip", "wgrad.hip", "vae.hip", "comm.hip", "regloss.hip", "blocks.hip", "stage2loss.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value"]
# per-source additions.  attention.hip: the SLP vectoriser packs the softmax's independent f32 multiplies into v_pk_mul_f32 and the
# bf16 conversions into shuffles -- 28 v_mov + 12 v_alignbit + 4 v_perm of register re-alignment per dK/dV tile step beside the
# MFMAs (the guide's "packed f32 VALU is an anti-lever beside MFMAs"); without it the loops are exp / mul / cvt only
FILE_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


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
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def is_current():
    """the library exists and was built from exactly the sources / header / flags on disk."""
    stamp_file = OUT + ".stamp"
    return os.path.exists(OUT) and os.path.exists(stamp_file) and open(stamp_file).read() == _stamp()


def build(force=False, verbose=True):
    """Build (or rebuild a stale) library.  Safe when several ranks of one node call it at once (torchrun, bench --gpus N, the
    two-rank tests): an flock serialises the builders, every builder compiles into its own temporary directory and the
    library + stamp are moved into place atomically, so nobody ever dlopens or links a half-written file; whoever gets the
    lock second finds the stamp current and returns."""
    import fcntl
    import shutil
    import tempfile
    stamp_file = OUT + ".stamp"
    if not force and is_current():
        return OUT
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_current():          # another process built it while this one waited
                return OUT
            return _build_locked(stamp_file, verbose, tempfile, shutil)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(stamp_file, verbose, tempfile, shutil):
    stamp = _stamp()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    objdir = tempfile.mkdtemp(prefix=f"obj{os.getpid()}_", dir=os.path.join(HERE, "build"))

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(7, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp_out = os.path.join(objdir, "libadaprompt_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", tmp_out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(tmp_out + ".stamp", "w") as fh:
        fh.write(stamp)
    if os.path.exists(stamp_file):
        os.remove(stamp_file)                     # never a current stamp next to an older library
    os.replace(tmp_out, OUT)
    os.replace(tmp_out + ".stamp", stamp_file)
    # keep the objects of the last build where tools/ expect them (A/B links), drop the temporary directory
    for o in objs:
        os.replace(o, os.path.join(HERE, "build", os.path.basename(o)))
    shutil.rmtree(objdir, ignore_errors=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
