"""ctypes binding of libflyhip.so (include/flyhip.h).  Fails loudly: no library, no product."""
import ctypes as C
import os

from .params import FlyParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLYHIP_LIB") or os.path.join(_HERE, "libflyhip.so")
_lib = None


class FlyHipError(RuntimeError):
    pass


class FlyBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("root", "dof_state", "targets", "contact", "pot", "prev_pot",
                                          "obs", "reward", "reset", "progress", "ep_return", "ep_length",
                                          "done_return", "done_length", "done_count")]


ABI_VERSION = 12        # include/flyhip.h as this package binds it (fly_abi_version(): argument lists changed between versions)
# name -> argtypes; every entry point returns int except fly_last_error
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "fly_abi_version": [],
    "fly_create": [C.POINTER(FlyParams), C.POINTER(_P)],
    "fly_destroy": [_P],
    "fly_step": [_P, _P, C.POINTER(FlyBuffers), _P],
    "fly_scale_actions": [_P, _P, _P, _P],
    "fly_reset_masked": [_P, C.POINTER(FlyBuffers), _P],
    "fly_integrate": [_P, C.POINTER(FlyBuffers), _P],
    "fly_pack_obs": [_P, C.POINTER(FlyBuffers), _P],
    "fly_pack_reward": [_P, C.POINTER(FlyBuffers), _I, _P],
    "ppo_sample_logprob": [_P, _P, _P, _P, _P, _L, _P],
    "ppo_td_gae": [_P, _P, _P, _P, _F, _F, _L, _L, _P, _P, _I, _P],
    "ppo_adv_stats": [_P, _L, _P, _P],
    "ppo_adv_apply": [_P, _L, _P, _F, _F, _P],
    "ppo_step_bookkeeping": [_P, _L, _P, _F, _P, _I, _F, _F, _P],
    "ppo_rollout_step": [_P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _P, _P],
    "ppo_rollout_all": [_P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "ppo_rollout_bookkeeping": [_P, _L, _L, _P, _P, _F, _P, _I, _F, _F, _P, _P],
    "mlp_forward": [_P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P],
    "mlp_forward_sample": [_P, _P, _P, _L, _P, _P, _I, _F, _F, _P, _P, _P, _P, _P, _P],
    "mlp_grad_workspace_floats": [],
    "mlp_backward_dx": [_P] * 10 + [_L, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "mlp_forward_backward": [_P] * 4 + [_L] + [_P] * 9 + [_F, _F] + [_P] * 6 + [_I, _P, _P, _P, _I, _P],
    "mlp_grad_w": [_P] * 8 + [_L, _P, _P, _P, _P, _P, _P, _I, _P],
    "mlp_fused_workspace_floats": [],
    "mlp_fused_grad": [_P] * 3 + [_P, _L] + [_P] * 5 + [_F, _F] + [_P] * 6 + [C.POINTER(_P), _P],
    "mlp_fused_h2_workspace_floats": [],
    "mlp_fused_grad_h2": [_P] * 5 + [_I, _P, _L] + [_P] * 5 + [_F, _F] + [_P] * 6 + [C.POINTER(_P), _P],
    "mlp_h2_rescale": [_P] * 7,
    "mlp_adam_step": [_P] * 10 + [_F, _F, _F, _F, _F, _F, _P, _I, _P, _P, _P, _P, _P, _P] + [_P] * 3 + [_I, _P],
    "dqn_eps_greedy": [_P, _P, _P, _F, _I, _P, _L, _P],
    "dqn_huber_td": [_P, _P, _P, _P, _P, _F, _I, _L, _P, _P, _P],
    "dqn_forward": [_P, _P, _P, _L, _P, _P],
    "dqn_act": [_P, _P, _P, _L, _P, _P, _F, _P, _P, _P],
    "dqn_td_step": [_P] * 10 + [_L, _F, _F] + [_P] * 7,
    "dqn_grad_workspace_floats": [],
    "dqn_grad_w": [_P] * 6 + [_L, _P, _P, _I, _P],
    "dqn_fused_workspace_floats": [],
    "dqn_fused_image_halves": [_L],
    "dqn_fused_update": [_P] * 6 + [_I, _L, _F, _F, _P, _P, _P, _P, _I, _P],
    "dqn_fused_h2_workspace_floats": [],
    "dqn_fused_h2_image_halves": [_L],
    "dqn_fused_update_h2": [_P] * 10 + [_I, _L, _F, _F, _P, _P, _P, _P, _I, _I, _P],
    "dqn_adam_soft_update": [_P] * 12 + [_F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "dp_p2p_alloc": [_L, C.POINTER(_P)],
    "dp_p2p_free": [_P],
    "dp_ipc_export": [_P, _P],
    "dp_ipc_import": [_P, C.POINTER(_P)],
    "dp_ipc_close": [_P],
    "dp_allreduce_p2p": [_P, _L, C.POINTER(_P), _I, _I, C.c_uint32, _P, _L, _P],
}


def build(verbose=False):
    """Compile libflyhip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlyHipError(
            "libflyhip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fly_bproject_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.fly_last_error.restype = C.c_char_p
    lib.fly_last_error.argtypes = []
    for name, argtypes in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name.endswith(("_workspace_floats", "_image_halves")) else C.c_int
    if lib.fly_abi_version() != ABI_VERSION:
        raise FlyHipError("stale libflyhip.so (ABI %d, this package was written for %d): rebuild it "
                          "(`make -C fly_bproject_amd/csrc`)" % (lib.fly_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().fly_last_error()
        raise FlyHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
