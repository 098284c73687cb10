"""ctypes binding of libadaprompt_hip.so -- the only doorway from the Python host code to the
HIP kernels.  Signatures are read from include/adaprompt_hip.h, so header and binding cannot
drift.  There is no CPU fallback: if the library is missing and cannot be built, or a call
returns a non-zero status, a RuntimeError is raised."""
import ctypes
import os
import re

# torch first: its bundled HIP runtime must be the one (and only) libamdhip64 in the process, otherwise the
# kernels would be launched through a second, uninitialised runtime on streams it does not know.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "adaprompt_hip.h")
LIB_PATH = os.path.join(HERE, "libadaprompt_hip.so")

ABI_VERSION = 3        # bumped whenever an entry point's argument list changes (capi.hip returns the same number)

_SCALARS = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    protos = {}
    for m in re.finditer(r"\b(const char\*|int|long)\s+(adap_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = {"int": ctypes.c_int, "long": ctypes.c_long, "const char*": ctypes.c_char_p}[ret]
        argtypes, argnames = [], []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                typ, an = mm.group(1).strip(), mm.group(2)
                if "*" in typ:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_SCALARS[typ.replace("const ", "")])
                argnames.append(an)
        protos[name] = (restype, argtypes, argnames)
    return protos


_lib = None
_protos = None


def load(build_if_missing=True):
    """dlopen the in-tree library (building it with hipcc first if it is absent)."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    from . import build as _build
    alt = os.environ.get("ADAP_LIB_PATH")          # A/B timing of two builds on one box (tools/): no stamp check, no rebuild
    if alt:
        lib = ctypes.CDLL(alt)
        _protos = parse_header()
        for name, (restype, argtypes, _) in _protos.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return lib
    if build_if_missing:
        # no-op when the source stamp matches; otherwise a stale library (sources, header or flags changed since it was
        # built -- the .so is git-ignored but ships with the snapshot) is rebuilt instead of being dlopen'ed
        _build.build(verbose=False)
    elif not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found; run `python -m adaprompt_amd.build`")
    elif not _build.is_current():
        raise RuntimeError(f"{LIB_PATH} is stale (csrc/ or include/adaprompt_hip.h changed since it was built); "
                           "run `python -m adaprompt_amd.build`")
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (restype, argtypes, _) in _protos.items():
        fn = getattr(lib, name)       # AttributeError here == header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.adap_abi_version() != ABI_VERSION:
        raise RuntimeError("libadaprompt_hip.so: ABI version mismatch")
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


_fns = {}


def _fn(name):
    f = _fns.get(name)
    if f is None:
        f = _fns[name] = getattr(load(), name)
    return f


def call(name, *args):
    """Call an int-returning entry point; raise HipError with adap_last_error() on failure."""
    rc = _fn(name)(*args)
    if rc != 0:
        raise HipError(f"{name} failed ({rc}): {load().adap_last_error().decode()}")


def call_long(name, *args):
    return _fn(name)(*args)


_sizes = {}


def size_query(name, *args):
    """memoised ``call_long`` for the pure workspace-size functions (hundreds of identical queries per step)."""
    key = (name, args)
    v = _sizes.get(key)
    if v is None:
        v = _sizes[key] = _fn(name)(*args)
    return v


def current_stream():
    """raw hipStream_t of torch's current stream on the current device, without building a torch.cuda.Stream
    object (torch.cuda.current_stream() costs ~9 us per call; there are ~1400 launches per micro-batch)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
