"""ctypes binding of libmarl_amd.so (the C ABI declared in include/marl_amd.h).

There is no CPU fallback: if the HIP library has not been built the import of any product
path fails with the build command.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MARL_AMD_LIB: a diagnostic build of this library (profiles/diag/libvar_*.so) instead of the product's;
# never silent: the load is announced on stderr
LIB_PATH = os.environ.get("MARL_AMD_LIB") or os.path.join(_HERE, "libmarl_amd.so")

EXPORTS = (
    "marl_last_error", "marl_env_create", "marl_env_destroy", "marl_env_set_auto_reset", "marl_env_obs_dim", "marl_env_num_rows",
    "marl_env_num_pots", "marl_env_grid_cells", "marl_env_state_words", "marl_env_state_bytes",
    "marl_env_reset", "marl_env_step", "marl_env_step_many", "marl_env_featurize", "marl_env_export_state",
    "marl_env_import_state", "marl_gae", "marl_gae_workspace_bytes", "marl_row_sums",
    "marl_reward_events", "marl_reward_events_workspace_bytes", "marl_reward_events_noalloc", "marl_reward_events_many", "marl_reward_events_many_workspace_bytes", "marl_sample_categorical",
    "marl_ppo_loss", "marl_ppo_loss_workspace_bytes",
)

_lib = None


class MarlAmdError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MarlAmdError(
            f"{LIB_PATH} is missing: build it with `make -C marl_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    if os.environ.get("MARL_AMD_LIB"):
        import sys

        print(f"marl_amd: MARL_AMD_LIB is set, loading {LIB_PATH} instead of the product library", file=sys.stderr, flush=True)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.marl_last_error.restype = ctypes.c_char_p
    L.marl_last_error.argtypes = []
    L.marl_env_create.restype = ci
    L.marl_env_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_char_p), ci, ci, ci, ci, ci,
                                  ctypes.c_uint64, ctypes.c_int64, ci]
    L.marl_env_destroy.restype = None
    L.marl_env_destroy.argtypes = [vp]
    for name in ("marl_env_obs_dim", "marl_env_num_rows", "marl_env_num_pots", "marl_env_grid_cells",
                 "marl_env_state_words"):
        getattr(L, name).restype = ci
        getattr(L, name).argtypes = [vp]
    L.marl_env_set_auto_reset.restype = ci
    L.marl_env_set_auto_reset.argtypes = [vp, ci]
    L.marl_env_state_bytes.restype = ctypes.c_size_t
    L.marl_env_state_bytes.argtypes = [vp]
    L.marl_env_reset.restype = ci
    L.marl_env_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    L.marl_env_step.restype = ci
    L.marl_env_step.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp]
    L.marl_env_featurize.restype = ci
    L.marl_env_step_many.restype = ci
    L.marl_env_step_many.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]
    L.marl_env_featurize.argtypes = [vp, vp, vp, vp]
    L.marl_env_export_state.restype = ci
    L.marl_env_export_state.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.marl_env_import_state.restype = ci
    L.marl_env_import_state.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.marl_gae.restype = ci
    L.marl_gae.argtypes = [vp, vp, vp, vp, ci, ci, ctypes.c_double, ctypes.c_double, vp, vp, ci, vp, ci, vp]
    L.marl_gae_workspace_bytes.restype = ctypes.c_size_t
    L.marl_gae_workspace_bytes.argtypes = [ci, ci]
    L.marl_row_sums.restype = ci
    L.marl_row_sums.argtypes = [vp, ci, ci, vp, vp, vp]
    L.marl_reward_events.restype = ci
    L.marl_reward_events.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    L.marl_reward_events_workspace_bytes.restype = ctypes.c_size_t
    L.marl_reward_events_workspace_bytes.argtypes = [ci, ci]
    L.marl_reward_events_many.restype = ci
    L.marl_reward_events_many.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp]
    L.marl_reward_events_many_workspace_bytes.restype = ctypes.c_size_t
    L.marl_reward_events_many_workspace_bytes.argtypes = [ci, ci, ci]
    L.marl_reward_events_noalloc.restype = ci
    L.marl_reward_events_noalloc.argtypes = [vp, ci, ci, vp, vp, vp]
    L.marl_sample_categorical.restype = ci
    L.marl_sample_categorical.argtypes = [vp, ci, ci, ci, ctypes.c_uint64, vp, ctypes.c_int64, vp, vp, vp, vp, vp]
    L.marl_ppo_loss.restype = ci
    L.marl_ppo_loss.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                vp, vp, vp, vp, vp]
    L.marl_ppo_loss_workspace_bytes.restype = ctypes.c_size_t
    L.marl_ppo_loss_workspace_bytes.argtypes = [ci]
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().marl_last_error()
        raise MarlAmdError(f"{what} failed ({rc}): {msg.decode() if msg else 'unknown error'}")
