"""ctypes binding of libmocodad_hip.so (include/mocodad_hip.h).  No CPU fallback: if the shared
library is missing or fails to load, importing callers get a RuntimeError telling them to build it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCD_LIB", os.path.join(_HERE, "libmocodad_hip.so"))   # MCD_LIB: tuning builds only

MCD_MAX_FRAMES = 32
MCD_MAX_COND_LAYERS = 8

STRATEGY = {"inject": 0, "concat": 1, "no_condition": 2, "inbetween_imp": 3, "random_imp": 4}
LOSS = {"smooth_l1": 0, "l1": 1, "mse": 2}
COND_UNET = -1  # MCD_COND_UNET
AGGR = {"all": 0, "best": 1, "worst": 2, "mean": 3, "median": 4, "mean_pose": 5, "median_pose": 6, "quantile": 7}
OPT = {"variant": 0, "cond_generic": 1, "generic_unet": 2, "split": 3, "phase": 4}     # MCD_OPT_*
ABI_VERSION = 4


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class ModelCfg(C.Structure):
    _fields_ = [("num_coords", C.c_int32), ("n_joints", C.c_int32), ("t_unet", C.c_int32), ("t_cond", C.c_int32),
                ("emb_dim", C.c_int32), ("strategy", C.c_int32), ("cond_layers", C.c_int32),
                ("cond_channels", C.c_int32 * MCD_MAX_COND_LAYERS)]


class ScoreCfg(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("n_samples", C.c_int32), ("noise_steps", C.c_int32), ("seg_len", C.c_int32),
                ("n_cond", C.c_int32), ("n_corrupt", C.c_int32), ("cond_idx", C.c_int32 * MCD_MAX_FRAMES),
                ("corrupt_idx", C.c_int32 * MCD_MAX_FRAMES), ("loss_fn", C.c_int32)]


class WindowView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("stride_c", C.c_int64), ("stride_t", C.c_int64), ("trans", C.c_void_p),
                ("affine", C.c_void_p), ("cond_mask", C.c_void_p)]


class FrameCfg(C.Structure):
    _fields_ = [("n_clips", C.c_int32), ("num_transform", C.c_int32), ("n_persons", C.c_int32), ("max_frames", C.c_int32),
                ("pad_size", C.c_int32), ("frames_shift", C.c_int32), ("gauss_radius", C.c_int32),
                ("clip_keys", C.c_void_p), ("clip_n_frames", C.c_void_p), ("frame_dst", C.c_void_p), ("clip_out_len", C.c_void_p),
                ("clip_out_off", C.c_void_p), ("gauss_weights", C.c_void_p)]


_SIGS = {
    "mcd_pack_weights": (C.c_int, [C.POINTER(Tensor), C.c_int32, C.POINTER(ModelCfg), C.c_int32, C.POINTER(C.c_void_p)]),
    "mcd_free_weights": (None, [C.c_void_p]),
    "mcd_set_option": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mcd_layer_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcd_pass_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32]),
    "mcd_philox_noise": (C.c_int, [C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mcd_debug_set_prof": (None, [C.c_void_p]),
    "mcd_debug_poison_lds": (C.c_int, [C.c_void_p]),
    "mcd_cond_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mcd_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcd_score_workspace_bytes": (C.c_int64, [C.c_void_p, C.POINTER(ScoreCfg)]),
    "mcd_plan_split": (C.c_int32, [C.c_void_p, C.POINTER(ScoreCfg)]),
    "mcd_score": (C.c_int, [C.c_void_p, C.POINTER(ScoreCfg), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcd_score_view": (C.c_int, [C.c_void_p, C.POINTER(ScoreCfg), C.c_void_p, C.POINTER(WindowView), C.c_void_p, C.c_uint64,
                                 C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcd_score_fused": (C.c_int, [C.c_void_p, C.POINTER(ScoreCfg), C.c_void_p, C.POINTER(WindowView), C.c_void_p, C.c_uint64,
                                  C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcd_aggregate": (C.c_int, [C.POINTER(ScoreCfg), C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcd_scatter_max": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p]),
    "mcd_frame_scores_workspace_bytes": (C.c_int64, [C.POINTER(FrameCfg)]),
    "mcd_frame_scores": (C.c_int, [C.POINTER(FrameCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcd_last_error": (C.c_char_p, []),
    "mcd_abi_version": (C.c_int32, []),
}

EXPORTS = tuple(_SIGS.keys())
_lib = None


def lib():
    """Load (once) and return the C-ABI library.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch bundles its own libamdhip64.so.7; it must be the HIP runtime instance this library binds to,
        # otherwise streams / device pointers would belong to a second runtime.  Load torch's first.
        import torch  # noqa: F401
        if torch.cuda.is_available():
            torch.cuda.init()
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
        # a stale library (the .so is git-ignored but travels with the tree) would silently misroute options and arguments:
        # check the ABI before binding anything else
        try:
            L.mcd_abi_version.restype = C.c_int32
            have = int(L.mcd_abi_version())
        except AttributeError:
            have = -1
        if have != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {have}, this package needs {ABI_VERSION}: rebuild it with "
                               "`python -m mocodad_amd.build` (or __graft_entry__.build())")
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f"mocodad_hip error {rc}: {lib().mcd_last_error().decode()}")
