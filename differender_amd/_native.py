"""ctypes binding of libdifferender_hip.so (C ABI: include/differender_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFFERENDER_HIP_LIB lets tools/ab.sh time two builds of the library in one job (A/B on the same device). A what-if build
# (kernels that compute wrong results on purpose, csrc/dr_experiment.h) is refused unless DIFFERENDER_ALLOW_EXPERIMENT=1.
LIB_PATH = os.environ.get("DIFFERENDER_HIP_LIB") or os.path.join(_HERE, "libdifferender_hip.so")

ABI_VERSION = 9
BUILD_WRONG_RESULTS, BUILD_DIAGNOSTIC = 1, 2   # dr_build_flags()
DR_F32, DR_F16 = 0, 1
DR_MODE_DIFF, DR_MODE_NONDIFF = 0, 1
DR_VARIANT_AUTO, DR_VARIANT_BASELINE = 0, 1
DR_HINT_NO_EARLY_TERMINATION, DR_HINT_EARLY_TERMINATION = 0x100, 0x200   # OR-ed into `variant` of dr_march_fwd[_rows]
DR_COUNT_EVALUATED = 0x400   # ... and of dr_march_bwd[_rows]: measurement only (workspace header words 58-63)
DR_TAPE_TF = 0x800           # forward (DIFF) + the TF-only backward of the same inputs: per-sample tape of (intensity, lighting)

_c = ctypes
_P, _I, _L, _F, _D, _U, _Z = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_double, _c.c_uint32, _c.c_size_t

# name -> (restype, argtypes); must list every symbol the header declares (tests/test_abi.py checks)
SIGNATURES = {
    "dr_abi_version": (_I, []),
    "dr_build_flags": (_I, []),
    "dr_error_string": (_c.c_char_p, [_I]),
    "dr_ray_setup": (_I, [_P, _I, _I, _I, _I, _I, _I, _D, _D, _F, _U, _U, _P, _P, _P, _P, _P]),
    "dr_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "dr_workspace_bytes_tape": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I, _F]),
    "dr_march_fwd": (_I, [_P, _I, _I, _I, _I, _L, _L, _L, _L, _P, _I, _L, _P, _P, _P, _P, _P,
                          _I, _I, _I, _I, _F, _D, _D, _I, _I, _P, _P, _P, _Z, _P]),
    "dr_march_bwd": (_I, [_P, _I, _I, _I, _I, _L, _L, _L, _L, _P, _I, _L, _P, _P, _P, _P, _P,
                          _I, _I, _I, _I, _F, _D, _D, _I, _P, _P, _P, _L, _L, _L, _L, _P, _L, _P, _Z, _P]),
    "dr_ray_setup_rows": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _D, _D, _F, _U, _U, _P, _P, _P, _P, _P]),
    "dr_march_fwd_rows": (_I, [_P, _I, _I, _I, _I, _L, _L, _L, _L, _P, _I, _L, _P, _P, _P, _P, _P,
                               _I, _I, _I, _I, _F, _D, _D, _I, _I, _P, _P, _P, _Z, _I, _I, _P]),
    "dr_march_bwd_rows": (_I, [_P, _I, _I, _I, _I, _L, _L, _L, _L, _P, _I, _L, _P, _P, _P, _P, _P,
                               _I, _I, _I, _I, _F, _D, _D, _I, _P, _P, _P, _L, _L, _L, _L, _P, _L, _P, _Z, _I, _I, _P]),
    "dr_march_bwd_variant": (_I, [_I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _I, _I, _I]),
    "dr_comm_unique_id": (_I, [_P]),
    "dr_comm_init_rank": (_I, [_c.POINTER(_P), _I, _P, _I]),
    "dr_comm_init_all": (_I, [_c.POINTER(_P), _I, _c.POINTER(_I)]),
    "dr_allreduce_f32": (_I, [_P, _P, _Z, _P]),
    "dr_allreduce_gradients_f32": (_I, [_P, _P, _Z, _P, _Z, _P]),
    "dr_comm_destroy": (_I, [_P]),
    "dr_mse_loss_grad": (_I, [_P, _P, _L, _F, _P, _P, _P]),
    "dr_tf_momentum_step": (_I, [_P, _P, _P, _I, _F, _F, _F, _P]),
}

_lib = None


def lib():
    """Load the HIP library (once). Raises ImportError with a build hint if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C differender_amd/csrc`). differender_amd has no CPU or PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        rebuild = "rebuild it (`make -C differender_amd/csrc`), or unset DIFFERENDER_HIP_LIB"
        # the version FIRST: a stale library lacks newer symbols, and the loop over SIGNATURES below would die on the first of
        # them with a bare AttributeError before the "ABI version ..., rebuild it" hint was ever reached
        try:
            handle.dr_abi_version.restype = ctypes.c_int
            version = handle.dr_abi_version()
        except AttributeError as exc:
            raise ImportError(f"{LIB_PATH} does not export dr_abi_version: not this package's library; {rebuild}") from exc
        # (same-device A/B against the library of an EARLIER round -- tools/abn.sh with ab_libs/r05.so: DIFFERENDER_AB_OLD_ABI=8 accepts
        #  that one older version; entry points it lacks are simply absent, and whatever needs them fails when it is called)
        ab_old = os.environ.get("DIFFERENDER_AB_OLD_ABI") == str(abs(version)) and abs(version) < ABI_VERSION
        if abs(version) != ABI_VERSION and not ab_old:
            raise ImportError(f"{LIB_PATH}: ABI version {version}, expected {ABI_VERSION}; {rebuild}")
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as exc:
                if ab_old:
                    continue
                raise ImportError(f"{LIB_PATH} reports ABI version {version} but does not export `{name}`; {rebuild}") from exc
            fn.restype = res
            fn.argtypes = args
        flags = handle.dr_build_flags()
        if (version < 0 or flags & BUILD_WRONG_RESULTS) and os.environ.get("DIFFERENDER_ALLOW_EXPERIMENT") != "1":
            raise ImportError(
                f"{LIB_PATH} is a what-if build whose kernels compute WRONG results on purpose (dr_build_flags() = {flags}); "
                "it is for timing experiments only. Set DIFFERENDER_ALLOW_EXPERIMENT=1 to load it anyway.")
        if flags & BUILD_WRONG_RESULTS:
            import warnings
            warnings.warn(f"{LIB_PATH}: what-if build, results are WRONG by construction", RuntimeWarning)
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().dr_error_string(code)
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else code} (code {code})")
