"""Loader for the engine's C-ABI library (host-side mirror of reference python/magent/c_lib.py:11-42).

The product library is magent_amd/lib/libmagent.so, built by ``__graft_entry__.build()`` from
magent_amd/csrc (HIP, gfx950).  There is NO CPU fallback: if the library is missing, loading fails loudly.  `load()` takes no environment override;
`load(path)` loads the named library -- the tests use it (tests/helpers.py: `world_on`) to drive the CPU checkers under
oracle/ through the same ctypes declarations.
"""
import ctypes
import os
import sys

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_PKG_DIR, "lib", "libmagent.so")

# every symbol include/magent_runtime_api.h declares: (name, restype, argtypes)
_c = ctypes
_vp, _i, _cp, _ip, _fp = _c.c_void_p, _c.c_int, _c.c_char_p, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_float)
REFERENCE_ABI = [
    ("env_new_game", [_c.POINTER(_vp), _cp]),
    ("env_delete_game", [_vp]),
    ("env_config_game", [_vp, _cp, _vp]),
    ("env_reset", [_vp]),
    ("env_get_observation", [_vp, _i, _c.POINTER(_vp)]),      # float *buffers[2]: addresses (numpy's .ctypes.data costs a third of data_as)
    ("env_set_action", [_vp, _i, _vp]),
    ("env_step", [_vp, _ip]),
    ("env_get_reward", [_vp, _i, _vp]),
    ("env_get_info", [_vp, _i, _cp, _vp]),
    ("env_render", [_vp]),
    ("env_render_next_file", [_vp]),
    ("gridworld_register_agent_type", [_vp, _cp, _i, _c.POINTER(_cp), _fp]),
    ("gridworld_new_group", [_vp, _cp, _ip]),
    ("gridworld_add_agents", [_vp, _i, _i, _cp, _ip, _ip, _ip]),
    ("gridworld_clear_dead", [_vp]),
    ("gridworld_set_goal", [_vp, _i, _cp, _ip]),
    ("gridworld_define_agent_symbol", [_vp, _i, _i, _i]),
    ("gridworld_define_event_node", [_vp, _i, _i, _ip, _i]),
    ("gridworld_add_reward_rule", [_vp, _i, _ip, _fp, _i, _c.c_bool, _c.c_bool]),
    ("discrete_snake_clear_dead", [_vp]),
    ("discrete_snake_add_object", [_vp, _i, _i, _cp, _ip]),
]
DEVICE_ABI = [
    ("env_get_observation_device", [_vp, _i, _c.POINTER(_vp)]),
    ("env_get_observation_device_bf16", [_vp, _i, _c.POINTER(_vp)]),
    ("env_set_action_device", [_vp, _i, _vp]),
    ("env_get_reward_device", [_vp, _i, _vp]),
    ("env_get_info_device", [_vp, _i, _cp, _vp]),
    ("env_step_many", [_c.POINTER(_vp), _i, _ip]),
    ("env_cycle_many", [_c.POINTER(_vp), _i, _i, _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp), _ip, _i]),
    ("env_num_many", [_c.POINTER(_vp), _i, _i, _ip]),
    ("env_sync", [_vp]),
    ("env_get_stream", [_vp, _c.POINTER(_vp)]),
    ("env_streams_many", [_c.POINTER(_vp), _i, _c.POINTER(_vp)]),
    ("env_get_action_stream", [_vp, _c.POINTER(_vp)]),
    ("env_profile_enable", [_vp, _i]),
    ("env_profile_read", [_vp, _cp, _ip, _fp]),
]
# include/magent_policy.h (struct arguments are passed by reference from magent_amd/builtin/torch_model/hip_policy.py)
POLICY_ABI = [
    ("policy_dqn_supported", [_vp]),
    ("policy_dqn_act_bytes", [_vp, _i, _c.POINTER(_c.c_size_t)]),
    ("policy_dqn_infer", [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("policy_dqn_infer_bf16", [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("policy_dqn_f32_supported", [_vp]),
    ("policy_dqn_f32_act_bytes", [_vp, _i, _c.POINTER(_c.c_size_t)]),
    ("policy_dqn_infer_f32", [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
]

_cache = {}
_lock = __import__("threading").Lock()


def load(path=None):
    with _lock:   # environments may be constructed from several threads at once
        return _load(path)


def _load(path=None):
    """dlopen the engine library and declare argtypes for every entry point it exports.

    RTLD_LOCAL on purpose: the CPU checkers under oracle/ export the same names and must be loadable
    next to the product in one test process."""
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise OSError(
            "magent_amd: engine library %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    if path == os.path.abspath(DEFAULT_LIB) and "torch" not in sys.modules:
        # PyTorch-ROCm bundles its own libamdhip64; if torch gets imported later the process would hold two HIP
        # runtimes.  Importing it first makes both share one.  Opt out for torch-free use.
        if os.environ.get("MAGENT_AMD_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
    lib = ctypes.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name, argtypes in REFERENCE_ABI:
        fn = getattr(lib, name)  # AttributeError if a reference symbol is missing: that is a bug
        fn.restype, fn.argtypes = ctypes.c_int, argtypes
    lib.has_device_api = True
    for name, argtypes in DEVICE_ABI:
        fn = getattr(lib, name, None)
        if fn is None:
            lib.has_device_api = False
            continue
        fn.restype, fn.argtypes = ctypes.c_int, argtypes
    for name, argtypes in POLICY_ABI:
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = ctypes.c_int, argtypes
    _cache[path] = lib
    return lib
