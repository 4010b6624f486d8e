"""ctypes binding of the C ABI in include/inversus_hip.h (libinversus_hip.so).

There is NO fallback: if the HIP library is missing or does not load, importing callers get a
RuntimeError.  The package never routes through oracle/ or any CPU path.
"""
import ctypes as C
import os

from . import _build

P = C.POINTER

INV_OK = 0
INV_ERR_BAD_ARG, INV_ERR_BAD_DIMS, INV_ERR_HIP = -1, -2, -3
INV_ERR_BAD_ACTION, INV_ERR_BULLET_OVERFLOW, INV_ERR_NO_OPPONENT = -4, -5, -6
INV_OPP_EXTERNAL, INV_OPP_DUMMY = 0, 1
INV_EASY, INV_HARD = 0, 1
INV_PHASE_ACTIONS, INV_PHASE_RELOAD, INV_PHASE_BULLETS, INV_PHASE_WRAPPER = 1, 2, 4, 8
INV_STEP_AUTO_RESET, INV_STEP_OBS_BITS, INV_STEP_DEFAULT = 16, 32, 15
INV_INFO_LANDED_HIT, INV_INFO_GOT_HIT, INV_INFO_WIN, INV_INFO_LOSE = 1, 2, 4, 8
INV_GAE_REFERENCE_FLAT, INV_GAE_PER_ENV = 0, 1
INV_STEP_KERNEL_AUTO, INV_STEP_KERNEL_WAVE, INV_STEP_KERNEL_ROW4 = 0, 1, 2
INV_OBS_STORE_AUTO, INV_OBS_STORE_PLAIN = 0, 1


class InvConfig(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("max_bullets", C.c_int32), ("max_episode_steps", C.c_int32), ("opponent", C.c_int32),
                ("difficulty", C.c_int32), ("device", C.c_int32), ("opp_seed", C.c_uint32),
                ("env_offset", C.c_int64)]


class InvStepOut(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("extra", C.c_void_p), ("obs_p2", C.c_void_p), ("extra_p2", C.c_void_p),
                ("reward", C.c_void_p), ("reward64", C.c_void_p), ("done", C.c_void_p),
                ("info_flags", C.c_void_p), ("episode_steps", C.c_void_p), ("episode_return", C.c_void_p)]


class InvStateView(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("players", C.c_void_p), ("nbullets", C.c_void_p), ("bullets", C.c_void_p),
                ("step_count", C.c_void_p), ("episode_return", C.c_void_p), ("prev_alive", C.c_void_p),
                ("opp_ctr", C.c_void_p), ("mt", C.c_void_p), ("mt_index", C.c_void_p)]


class InvPolicyLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("ln", C.c_void_p)]


class InvPolicyParams(C.Structure):
    _fields_ = [("layer", InvPolicyLayer * 4)]


class InvPolicySaved(C.Structure):
    _fields_ = [("xhat", C.c_void_p * 4), ("act", C.c_void_p * 3), ("rstd", C.c_void_p)]


class InvPolicyConvGrads(C.Structure):
    _fields_ = [("dw", C.c_void_p * 4)]


class InvPolicyLnGrads(C.Structure):
    _fields_ = [("gw", C.c_void_p * 4), ("gb", C.c_void_p * 4)]


class InvPolicyBandSaved(C.Structure):
    _fields_ = [("z", C.c_void_p * 4), ("stats", C.c_void_p)]


class InvPolicyBwdParams(C.Structure):
    _fields_ = [("wt", C.c_void_p * 3), ("ln", C.c_void_p * 4)]


class InvPolicyGrads(C.Structure):
    _fields_ = [("dy", C.c_void_p * 4), ("dz", C.c_void_p * 4)]


class InvHeadsParams(C.Structure):
    _fields_ = [("w2", C.c_void_p * 2), ("w2t", C.c_void_p * 2), ("w3", C.c_void_p * 2), ("w3t", C.c_void_p * 2),
                ("we", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p * 2), ("b3", C.c_void_p * 2)]


class InvHeadsPackItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("ld", C.c_int32),
                ("transposed", C.c_int32), ("mpad", C.c_int32), ("kpad", C.c_int32)]


INV_HEADS_PACK_MAX = 8


class InvHeadsPackList(C.Structure):
    _fields_ = [("item", InvHeadsPackItem * INV_HEADS_PACK_MAX)]


class InvHeadsGrads(C.Structure):
    _fields_ = [("db1", C.c_void_p), ("dwe", C.c_void_p), ("db2", C.c_void_p * 2), ("dw3", C.c_void_p * 2), ("db3", C.c_void_p * 2)]


_V = C.c_void_p
SIGNATURES = {
    "inv_last_error": (C.c_char_p, []),
    "inv_abi_version": (C.c_int, []),
    "inv_create": (C.c_int, [P(InvConfig), P(_V)]),
    "inv_destroy": (C.c_int, [_V]),
    "inv_get_config": (C.c_int, [_V, P(InvConfig)]),
    "inv_set_step_kernel": (C.c_int, [_V, C.c_int32]),
    "inv_set_obs_store_policy": (C.c_int, [_V, C.c_int32]),
    "inv_obs_floats_per_env": (C.c_int64, [_V]),
    "inv_grid_words": (C.c_int32, [_V]),
    "inv_obs_words_per_env": (C.c_int32, [_V]),
    "inv_seed": (C.c_int, [_V, _V, _V]),
    "inv_seed_masked": (C.c_int, [_V, _V, _V, _V]),
    "inv_reset": (C.c_int, [_V, _V, _V]),
    "inv_step": (C.c_int, [_V, _V, _V, C.c_uint32, P(InvStepOut), _V]),
    "inv_step_masked": (C.c_int, [_V, _V, _V, _V, C.c_uint32, P(InvStepOut), _V]),
    "inv_dummy_actions": (C.c_int, [_V, C.c_int32, _V, _V]),
    "inv_render_obs": (C.c_int, [_V, C.c_int32, _V, _V, _V]),
    "inv_render_obs_bits": (C.c_int, [_V, C.c_int32, _V, _V, _V]),
    "inv_unpack_obs": (C.c_int, [_V, C.c_int64, C.c_int32, _V, _V]),
    "inv_get_state": (C.c_int, [_V, P(InvStateView), _V]),
    "inv_set_state": (C.c_int, [_V, P(InvStateView), _V]),
    "inv_check": (C.c_int, [_V, _V]),
    "inv_gae": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int32, C.c_float, C.c_float, _V, C.c_int32, _V, _V, _V]),
    "inv_uniform_actions": (C.c_int, [_V, C.c_int32, C.c_uint32, C.c_int64, C.c_uint32, _V]),
    "inv_rollout": (C.c_int, [_V, C.c_int32, _V, _V, C.c_uint32, P(InvStepOut), _V]),
    "inv_policy_trunk": (C.c_int, [P(InvPolicyParams), _V, _V, C.c_int32, _V]),
    "inv_policy_trunk_bits": (C.c_int, [P(InvPolicyParams), _V, _V, C.c_int32, _V]),
    "inv_policy_trunk_train_fwd": (C.c_int, [P(InvPolicyParams), _V, C.c_int32, _V, P(InvPolicySaved), C.c_int32, _V]),
    "inv_policy_trunk_bwd": (C.c_int, [P(InvPolicyBwdParams), P(InvPolicySaved), _V, P(InvPolicyGrads), C.c_int32, _V]),
    "inv_policy_wgrad": (C.c_int, [C.c_int32, _V, _V, _V, _V, C.c_int32, _V]),
    "inv_policy_param_grads": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int32, C.c_int32, _V]),
    "inv_policy_param_grads_all": (C.c_int, [P(InvPolicyGrads), P(InvPolicySaved), P(InvPolicyLnGrads), C.c_int32, C.c_int32, _V]),
    "inv_policy_param_grads_all2": (C.c_int, [P(InvPolicyGrads), P(InvPolicySaved), P(InvPolicyLnGrads), _V, _V, C.c_int32, C.c_int32, _V]),
    "inv_policy_wgrad_to_module": (C.c_int, [_V, P(InvPolicyConvGrads), _V]),
    "inv_policy_pack_conv": (C.c_int, [_V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _V, _V]),
    "inv_policy_pack_ln": (C.c_int, [_V, _V, C.c_int32, C.c_int32, _V, _V]),
    "inv_policy_trunk_banded_scratch_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "inv_policy_trunk_banded": (C.c_int, [P(InvPolicyParams), _V, C.c_int32, C.c_int32, C.c_int32, _V, _V, C.c_int64, C.c_int32, _V]),
    "inv_policy_obs_to_x0": (C.c_int, [_V, C.c_int32, _V, C.c_int32, _V]),
    "inv_policy_trunk_banded_train_fwd": (C.c_int, [P(InvPolicyParams), _V, C.c_int32, C.c_int32, C.c_int32, _V, P(InvPolicyBandSaved), C.c_int32, _V]),
    "inv_policy_band_ln_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V, _V, _V, _V, C.c_int32, _V]),
    "inv_policy_band_ln_sums": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V, _V, C.c_int32, _V]),
    "inv_policy_band_ln_bwd2": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V, _V, _V, _V, _V, C.c_int32, C.c_int32, _V]),
    "inv_policy_band_conv_t": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V, C.c_int32, _V]),
    "inv_policy_band_conv_t_sums": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V, _V, _V, _V, _V, C.c_int32, _V]),
    "inv_policy_band_act": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V, C.c_int32, _V]),
    "inv_policy_band_wgrad": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V, C.c_int32, _V]),
    "inv_policy_band_param_grads": (C.c_int, [C.c_int32, _V, _V, _V, _V, _V, C.c_int32, C.c_int32, C.c_int32, _V]),
    "inv_policy_obs_to_x0_any": (C.c_int, [_V, C.c_int32, C.c_int32, _V, C.c_int32, _V]),
    "inv_heads_pack": (C.c_int, [P(InvHeadsPackList), C.c_int32, _V]),
    "inv_heads_pack_fc1": (C.c_int, [_V, C.c_int32, C.c_int32, _V, _V, _V]),
    "inv_heads_fwd": (C.c_int, [P(InvHeadsParams), _V, C.c_int32, _V, _V, _V, _V, _V, C.c_int32, C.c_int32, _V]),
    "inv_heads_bwd": (C.c_int, [P(InvHeadsParams), _V, _V, _V, _V, _V, _V, _V, C.c_int32, _V, P(InvHeadsGrads), C.c_int32, _V]),
    "inv_heads_dw2": (C.c_int, [_V, _V, C.c_int32, _V, _V, _V]),
    "inv_fc1_fwd_slabs": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "inv_fc1_fwd": (C.c_int, [_V, _V, C.c_int32, C.c_int32, C.c_int32, _V, _V]),
    "inv_fc1_dfeat": (C.c_int, [_V, _V, C.c_int32, C.c_int32, C.c_int32, _V, _V]),
    "inv_fc1_dw_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "inv_fc1_dw": (C.c_int, [_V, _V, _V, C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V]),
    "inv_transpose_bf16": (C.c_int, [_V, C.c_int32, C.c_int32, _V, _V]),
    "inv_ppo_loss_blocks": (C.c_int32, [C.c_int32]),
    "inv_ppo_loss": (C.c_int, [_V, _V, _V, _V, _V, _V, C.c_int32, C.c_float, C.c_float, C.c_float, _V, _V, _V, _V]),
    "inv_sample_actions": (C.c_int, [_V, C.c_int32, _V, _V, _V, _V, _V]),
    "inv_clip_adam_workspace_bytes": (C.c_int64, []),
    "inv_clip_adam_step": (C.c_int, [_V, _V, _V, _V, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float,
                                     C.c_float, _V, _V]),
}

_lib = None


def library_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load libinversus_hip.so; build it first if the sources are newer.  Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    override = os.environ.get("INVERSUS_HIP_LIB")          # experiments: load another build of the same ABI
    if override:
        path, build_if_missing = override, False
    if build_if_missing:
        try:
            path = _build.build_library()
        except Exception as exc:                               # hipcc missing on a box that has the prebuilt .so
            if not os.path.exists(path):
                raise RuntimeError(f"libinversus_hip.so is not built and cannot be built here: {exc}") from exc
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as exc:
        raise RuntimeError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)                                # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class InversusError(RuntimeError):
    pass


def raise_for(code, lib=None):
    """Map a negative inv_status to the exception class the reference raises for it."""
    if code == INV_OK:
        return
    lib = lib or load()
    msg = (lib.inv_last_error() or b"").decode("utf-8", "replace")
    if code in (INV_ERR_BAD_ACTION, INV_ERR_BAD_DIMS, INV_ERR_NO_OPPONENT, INV_ERR_BAD_ARG):
        raise ValueError(msg)                                  # env_wrappers.py:66,309,316; core.py:69-70,198
    raise InversusError(f"inv_status {code}: {msg}")
