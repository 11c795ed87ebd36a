"""ctypes binding of libdose_hip.so (the C ABI declared in include/dose_hip.h).

The prototypes are parsed from the header itself, so the Python side can never drift from the C ABI.
The product path has NO CPU fallback: if the shared library is missing, importing any op raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "dose_hip.h")
# (DOSE_HIP_LIB: another build of the same C ABI, for A/B measurements on one box -- tools/build_ab.sh)
LIB_PATH = os.path.abspath(os.environ["DOSE_HIP_LIB"]) if os.environ.get("DOSE_HIP_LIB") else os.path.join(_HERE, "libdose_hip.so")

_CTYPES = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "int32_t": ctypes.c_int32,
    "double": ctypes.c_double,
}


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(dp_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int)
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                argnames.append(re.split(r"[\s\*]+", a)[-1])
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[a.replace("const", "").split()[0]])
        protos[name] = (restype, argtypes, argnames)
    return protos


PROTOS = parse_header()
_lib = None


class DoseHipError(RuntimeError):
    pass


_FN = {}            # entry point name -> the callable call() uses (fast binding when built, else the ctypes function)
BINDING = None      # "fastcall" | "ctypes" once lib() has run


def lib():
    """The loaded library as an object with one attribute per entry point of include/dose_hip.h.  The symbols are always resolved
    through ctypes first (a declared but missing symbol raises AttributeError); calls then go through the generated METH_FASTCALL
    module (dose_prediction_amd/_fastgen.py: ~0.3 us per call instead of 4-6 us of ctypes argument conversion -- the launch thread
    makes ~1 300 calls per training step) when it has been built, unless DOSE_HIP_CTYPES=1."""
    global _lib, BINDING
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DoseHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes, _) in PROTOS.items():
            fn = getattr(L, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = restype, argtypes
        fast = None
        if not os.environ.get("DOSE_HIP_CTYPES"):
            try:
                from . import _dose_fastcall as fast
                fast.bind(LIB_PATH)
                if any(not hasattr(fast, name) for name in PROTOS):
                    raise ImportError("built from another version of include/dose_hip.h")
            except ImportError as e:
                import warnings
                warnings.warn(f"dose_prediction_amd: fast C binding unavailable ({e}); calling the library through ctypes "
                              "(same results, ~5 ms more host time per training step) -- run __graft_entry__.build()")
                fast = None
        if fast is not None:
            import types
            ns = types.SimpleNamespace(**{name: getattr(fast, name) for name in PROTOS})
            ns._cdll = L
            _FN.update({name: getattr(fast, name) for name in PROTOS})
            _lib, BINDING = ns, "fastcall"
        else:
            _FN.update({name: getattr(L, name) for name in PROTOS})
            _lib, BINDING = L, "ctypes"
    return _lib


# Optional per-launch timing (bench.py): when PROFILE is a list and ``name`` is in PROFILE_NAMES, every call is bracketed by
# HIP events recorded on the stream the kernel is launched on (torch's current stream) and appended as
# (name, args, start_event, end_event).
PROFILE = None
PROFILE_NAMES = ("dp_conv3d", "dp_conv3d_tiled", "dp_conv3d_tiled2", "dp_conv3d_tiled_stats", "dp_conv3d_wgrad", "dp_conv3d_wgrad_tiled", "dp_conv3d_wgrad_tiled2",
                 "dp_gemm_nt", "dp_gemm_tn", "dp_attention_fwd", "dp_attention_bwd")


# entry points that may answer 3 = "not this kernel's shape, nothing launched" (the caller then takes its general path)
SOFT_DECLINE = ("dp_tconv2x_fwd", "dp_stats_partial_finalize", "dp_norm_act_bwd_partial_finalize", "dp_norm_act_cat_bwd_partial_finalize")


def call(name, *args):
    """Call an int-returning entry point; raise DoseHipError(dp_last_error()) on a non-zero status (3 is returned for SOFT_DECLINE names)."""
    fn = _FN.get(name)
    if fn is None:
        lib()
        fn = _FN[name]
    if PROFILE is not None and name in PROFILE_NAMES:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        PROFILE.append((name, args, e0, e1))
    else:
        rc = fn(*args)
    if rc != 0 and not (rc == 3 and name in SOFT_DECLINE):
        raise DoseHipError(f"{name} failed (rc={rc}): {_FN['dp_last_error']().decode()}")
    return rc
