"""Host-side mirror of the reference's ``magent.GridWorld`` operator interface.

Same class / method names, argument meaning and return shapes as reference python/magent/gridworld.py:14-482
(GridWorld) and :571-800 (Config / Event / AgentSymbol / CircleRange / SectorRange), written from scratch on top
of the C-ABI declared in include/magent_runtime_api.h.  On top of the reference surface it adds the
device-resident calls (``*_device``) of the MI355X engine.

The engine library is always magent_amd/lib/libmagent.so (the HIP engine); loading fails if it is absent.  The class
attribute ``_engine_path`` is what the tests override in a subclass of their own (tests/helpers.py) to drive the CPU
checkers under oracle/ through the same wrapper -- nothing in the package sets it.
"""
import atexit
import ctypes
import importlib
import os
import weakref

import numpy as np

from . import c_lib

_F32P = ctypes.POINTER(ctypes.c_float)
_I32P = ctypes.POINTER(ctypes.c_int32)


def _i32(a):
    return a.ctypes.data_as(_I32P)


def _f32(a):
    return a.ctypes.data_as(_F32P)


def _gid(handle):
    """group handles are ctypes.c_int32 objects in the reference (gridworld.py:94-96); accept ints too"""
    return handle.value if hasattr(handle, "value") else int(handle)


# key -> python type that decides how env_config_game's void* is filled (reference gridworld.py:46-63)
_CONFIG_KINDS = {
    "map_width": int, "map_height": int, "embedding_size": int, "device_id": int,
    "food_mode": bool, "turn_mode": bool, "minimap_mode": bool, "revive_mode": bool, "goal_mode": bool,
    "render_dir": str,
}


_live_worlds = weakref.WeakSet()


@atexit.register
def _close_worlds():
    """environments still alive at interpreter exit are closed while the HIP runtime is still up (afterwards their streams and
    device memory cannot be returned any more)"""
    for env in list(_live_worlds):
        env.close()


class GridWorld(object):
    OBS_INDEX_VIEW = 0
    OBS_INDEX_HP = 1

    _engine_path = None      # None: the product library (c_lib.DEFAULT_LIB)

    def __init__(self, config, device_obs=None, **kwargs):
        """config: name of a built-in game ("battle", "gather", "pursuit", kwargs -> its get_config) or a Config
        device_obs: get_observation() returns torch tensors living on the engine's GPU instead of numpy arrays (the
                    same reused buffers, no PCIe); default from the environment variable MAGENT_DEVICE_OBS=1.  Lets an
                    unmodified training script keep observations, policy and replay memory on the device.
                    device_obs="bf16": the views come as torch.bfloat16 [n, H, W, 8] (get_observation_device_bf16) -- what
                    DeepQNetwork.infer_action feeds to its MFMA kernels without a conversion; for acting, not for the
                    float32 replay memory of training."""
        self._lib = c_lib.load(type(self)._engine_path)
        L = self._lib
        if device_obs is None:
            device_obs = {"0": False, "1": True}.get(os.environ.get("MAGENT_DEVICE_OBS", "0"), os.environ.get("MAGENT_DEVICE_OBS"))
        self._device_obs = bool(device_obs) and getattr(L, "has_device_api", False)
        self._obs_bf16 = device_obs == "bf16"      # views as bf16 cells of 8 channels, the policy kernels' input format
        self._dev_cache, self._dev_slot, self._dev_guard = ({}, {}), {}, {}
        if isinstance(config, str):
            config = _builtin_config(config, **kwargs)

        self.game = ctypes.c_void_p()
        self._num = ctypes.c_int32(0)
        self._num_ref = ctypes.byref(self._num)       # (get_num's out-parameter; one environment is driven by one thread at a time)
        L.env_new_game(ctypes.byref(self.game), b"GridWorld")
        _live_worlds.add(self)

        self._device_id = int(config.config_dict.get(
            "device_id", os.environ.get("MAGENT_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
        for key, val in config.config_dict.items():
            kind = _CONFIG_KINDS[key]
            if key == "device_id" and not getattr(L, "has_device_api", False):
                continue  # additive key: only the MI355X engine knows it
            self._config(key, kind, val)

        # agent types: parallel key/value arrays; a range object expands to (radius, angle) (gridworld.py:66-86)
        for name, attr in config.agent_type_dict.items():
            flat = {}
            for k, v in attr.items():
                if k in ("view_range", "attack_range"):
                    stem = k.split("_")[0]
                    flat[stem + "_radius"], flat[stem + "_angle"] = v.radius, v.angle
                else:
                    flat[k] = v
            keys = (ctypes.c_char_p * len(flat))(*[k.encode() for k in flat])
            vals = (ctypes.c_float * len(flat))(*[float(v) for v in flat.values()])
            L.gridworld_register_agent_type(self.game, name.encode(), len(flat), keys, vals)

        self._send_reward_rules(config)

        self.group_handles = []
        for type_name in config.groups:
            h = ctypes.c_int32()
            L.gridworld_new_group(self.game, type_name.encode(), ctypes.byref(h))
            self.group_handles.append(h)

        self._obs_cache = ({}, {})
        self.view_space, self.feature_space, self.action_space = {}, {}, {}
        tmp = np.empty(3, dtype=np.int32)
        for h in self.group_handles:
            L.env_get_info(self.game, h.value, b"view_space", tmp.ctypes.data)
            self.view_space[h.value] = (int(tmp[0]), int(tmp[1]), int(tmp[2]))
            L.env_get_info(self.game, h.value, b"feature_space", tmp.ctypes.data)
            self.feature_space[h.value] = (int(tmp[0]),)
            L.env_get_info(self.game, h.value, b"action_space", tmp.ctypes.data)
            self.action_space[h.value] = (int(tmp[0]),)

    def _config(self, key, kind, val):
        """env_config_game takes a void* whose pointee type depends on the key (GridWorld.cc:120-149); the ctypes
        object is held in a local until the call returns"""
        if kind is int:
            box = ctypes.c_int(val)
        elif kind is bool:
            box = ctypes.c_bool(val)
        else:
            box = ctypes.create_string_buffer(val.encode())
        self._lib.env_config_game(self.game, key.encode(), ctypes.addressof(box))
        del box

    # ------------------------------------------------------------------ setup
    def reset(self):
        self._lib.env_reset(self.game)

    def add_walls(self, method, **kwargs):
        """method 'random' (n=...) | 'custom' (pos=[(x,y),...]) | 'fill' (pos=(x,y), size=(w,h))"""
        kwargs["dir"] = 0
        self.add_agents(-1, method, **kwargs)

    def new_group(self, name):
        h = ctypes.c_int32()
        self._lib.gridworld_new_group(self.game, name.encode(), ctypes.byref(h))
        return h

    def add_agents(self, handle, method, **kwargs):
        """method 'random' (n=...) | 'custom' (pos=[(x,y[,dir]),...]) | 'fill' (pos=(x,y), size=(w,h)[, dir])"""
        L, g = self._lib, _gid(handle)
        if method == "random":
            L.gridworld_add_agents(self.game, g, int(kwargs["n"]), b"random", None, None, None)
        elif method == "custom":
            pos = np.asarray(kwargs["pos"], dtype=np.int32)
            if pos.size == 0:
                return
            xs = np.ascontiguousarray(pos[:, 0])
            ys = np.ascontiguousarray(pos[:, 1])
            ds = np.ascontiguousarray(pos[:, 2]) if pos.shape[1] >= 3 else np.zeros(len(pos), dtype=np.int32)
            L.gridworld_add_agents(self.game, g, len(pos), b"custom", _i32(xs), _i32(ys), _i32(ds))
        elif method == "fill":
            (x, y), (w, h) = kwargs["pos"][:2], kwargs["size"][:2]
            packed = np.array([x, y, w, h, kwargs.get("dir", 0)], dtype=np.int32)
            L.gridworld_add_agents(self.game, g, 0, b"fill", _i32(packed), None, None)
        else:
            raise ValueError("unknown placement method %r" % (method,))

    # ------------------------------------------------------------------ run (host buffers: reference ABI)
    def _buf(self, which, g, shape):
        """observation buffer of group g: grown when the group grows, otherwise the same memory (a leading slice) --
        like the reference's in-place resize (gridworld.py:203-213) it keeps the pages touched and the address stable"""
        cache = self._obs_cache[which]
        buf = cache.get(g)
        if buf is None or buf.shape[1:] != shape[1:] or buf.shape[0] < shape[0]:
            buf = cache[g] = np.empty(shape, dtype=np.float32)
        return buf[:shape[0]]

    def get_observation(self, handle):
        """-> (view float32[n,H,W,C], feature float32[n,F]); buffers are reused between calls like the reference"""
        g = _gid(handle)
        n = self.get_num(g)
        if self._device_obs:
            return self._observe_device_cached(g, n)
        view = self._buf(0, g, (n,) + self.view_space[g])
        feat = self._buf(1, g, (n,) + self.feature_space[g])
        bufs = (ctypes.c_void_p * 2)(view.ctypes.data, feat.ctypes.data)
        self._lib.env_get_observation(self.game, g, bufs)
        return view, feat

    def _observe_device_cached(self, g, n):
        import torch
        out = []
        spaces = ((0, self.view_space[g], torch.float32), (1, self.feature_space[g], torch.float32))
        if self._obs_bf16:
            spaces = ((0, self.view_space[g][:2] + (8,), torch.bfloat16), spaces[1])
        # Validity contract: the tensors returned for a group stay valid until the NEXT get_observation call for that group (the reference
        # reuses its numpy buffers the same way); the cache holds two sets per group -- twice the observation memory, 7.8 GB at
        # 2 x 400k float32 agents.
        # Two sets of buffers per group, handed out in turn.  The set written now was handed out two calls ago, and (like the
        # reference's reused buffers) stopped being valid at the previous call for this group: whatever torch work reads it was
        # queued before that call, where an event was recorded on torch's stream.  The render waits for THAT event -- not for
        # torch's whole queue: a policy that is still running on the other group's observation does not hold this render back --
        # and torch's stream for the render; stream-to-stream, the host does not block.
        slot = self._dev_slot.get(g, 0) ^ 1
        self._dev_slot[g] = slot
        dev = torch.device("cuda", self._device_id)
        for which, space, dtype in spaces:
            buf = self._dev_cache[which].get((g, slot))
            if buf is None or buf.shape[0] < n:
                buf = self._dev_cache[which][(g, slot)] = torch.empty((n,) + space, dtype=dtype, device=dev)
            out.append(buf[:n])
        # (the event only covers work queued on the torch stream it was recorded on: a caller that has switched streams since -- a
        # worker stream, another thread -- is ordered against its CURRENT stream as well, not instead: readers queued on the old
        # stream still hold this set; ADVICE rounds 3 and 4)
        guard = self._dev_guard.get((g, slot))
        cur = torch.cuda.current_stream(dev)
        if guard is not None:
            for st in self._streams():
                st.wait_event(guard[0])
        if guard is None or guard[1] != cur.cuda_stream:
            self.order_after_torch()
        if self._obs_bf16:
            self.get_observation_device_bf16(g, out[0], out[1])
        else:
            self.get_observation_device(g, out[0], out[1])
        self.order_torch_after()
        ev = torch.cuda.Event()
        ev.record(cur)                                  # everything queued so far may still read the OTHER set: its next render waits for this
        self._dev_guard[(g, slot ^ 1)] = (ev, cur.cuda_stream)
        return out[0], out[1]

    def use_bf16_observations(self, on=True):
        """device_obs mode: get_observation hands out the views as bf16 cells of 8 channels (get_observation_device_bf16)"""
        if on and max(v[2] for v in self.view_space.values()) > 7:
            raise ValueError("bf16-cell observations hold at most 7 channels; this game has %d" % max(v[2] for v in self.view_space.values()))
        if bool(on) != self._obs_bf16:
            self._obs_bf16, self._dev_cache, self._dev_slot, self._dev_guard = bool(on), ({}, {}), {}, {}

    def set_action(self, handle, actions):
        if not isinstance(actions, np.ndarray):   # a torch int32 tensor on the engine's device
            import torch
            assert isinstance(actions, torch.Tensor) and actions.dtype == torch.int32 and actions.is_cuda
            actions = actions.contiguous()            # (a copy, if any, is queued on torch's stream: before the hand-over)
            self.order_after_torch()                  # the producer of `actions` runs on torch's stream
            self.set_action_device(handle, actions)
            self.order_torch_after()                  # ... and whatever torch's stream does with that memory next comes after the read
            return
        assert isinstance(actions, np.ndarray) and actions.dtype == np.int32
        actions = np.ascontiguousarray(actions)
        self._lib.env_set_action(self.game, _gid(handle), actions.ctypes.data)

    def step(self):
        done = ctypes.c_int32(0)
        self._lib.env_step(self.game, ctypes.byref(done))
        return bool(done.value)

    def get_reward(self, handle):
        g = _gid(handle)
        out = np.empty(self.get_num(g), dtype=np.float32)
        self._lib.env_get_reward(self.game, g, out.ctypes.data)
        return out

    def clear_dead(self):
        self._lib.gridworld_clear_dead(self.game)

    # ------------------------------------------------------------------ info
    def get_handles(self):
        return self.group_handles

    def _info(self, g, name, buf):
        self._lib.env_get_info(self.game, g, name, buf.ctypes.data)
        return buf

    def get_num(self, handle):
        # (called by nearly every other method: a ctypes int kept for the purpose, not a fresh numpy array per call -- 6 -> 1.5 us)
        self._lib.env_get_info(self.game, handle.value if hasattr(handle, "value") else int(handle), b"num", self._num_ref)
        return self._num.value

    def get_action_space(self, handle):
        return self.action_space[_gid(handle)]

    def get_view_space(self, handle):
        return self.view_space[_gid(handle)]

    def get_feature_space(self, handle):
        return self.feature_space[_gid(handle)]

    def get_agent_id(self, handle):
        g = _gid(handle)
        return self._info(g, b"id", np.empty(self.get_num(g), dtype=np.int32))

    def get_alive(self, handle):
        g = _gid(handle)
        return self._info(g, b"alive", np.empty(self.get_num(g), dtype=np.bool_))

    def get_pos(self, handle):
        g = _gid(handle)
        return self._info(g, b"pos", np.empty((self.get_num(g), 2), dtype=np.int32))

    def get_view2attack(self, handle):
        """-> (attack_base, int32[H,W]): view cell -> attack action index or -1 (GridWorld.cc:853-872)"""
        g = _gid(handle)
        table = self._info(g, b"view2attack", np.empty(self.view_space[g][:2], dtype=np.int32))
        base = self._info(g, b"attack_base", np.zeros(1, dtype=np.int32))
        return int(base[0]), table

    def get_global_minimap(self, height, width):
        buf = np.empty((height, width, len(self.group_handles)), dtype=np.float32)
        buf.reshape(-1)[:2] = height, width  # in-params travel in the out-buffer's first two floats (GridWorld.cc:741-742)
        return self._info(-1, b"global_minimap", buf)

    def get_mean_info(self, handle):
        """-> float32[2 + n_action]: mean x, mean y, the share of every action among the group's last actions ("deprecated" in the reference,
        gridworld.py:375-380, GridWorld.cc:765-786)"""
        g = _gid(handle)
        return self._info(g, b"mean_info", np.empty(2 + self.action_space[g][0], dtype=np.float32))

    def set_seed(self, seed):
        self._config("seed", int, int(seed))

    # ------------------------------------------------------------------ render (host-side text dump)
    def set_render_dir(self, name):
        if not os.path.exists(name):
            os.mkdir(name)
        self._config("render_dir", str, name)

    def render(self):
        self._lib.env_render(self.game)

    def _get_groups_info(self):
        return self._info(-1, b"groups_info", np.empty((len(self.group_handles), 5), dtype=np.int32))

    def _get_walls_info(self):
        buf = self._info(-1, b"walls_info", np.empty((100 * 100, 2), dtype=np.int32))
        return buf[1:1 + buf[0, 0]]

    def _get_render_info(self, x_range, y_range):
        n = sum(self.get_num(h) for h in self.group_handles)
        buf = np.empty((n + 1, 4), dtype=np.int32)
        buf[0] = x_range[0], y_range[0], x_range[1], y_range[1]
        self._info(-1, b"render_window_info", buf)
        n_agent, n_event = int(buf[0, 0]), int(buf[0, 1])
        agents = {int(r[0]): [int(r[1]), int(r[2]), int(r[3])] for r in buf[1:1 + n_agent]}
        events = self._info(-1, b"attack_event", np.empty((n_event, 3), dtype=np.int32))
        return agents, events

    def set_goal(self, handle, method, *args, **kwargs):
        if method != "random":
            raise NotImplementedError
        self._lib.gridworld_set_goal(self.game, _gid(handle), b"random", None)

    def close(self):
        """gives the engine's resources back (device memory, stream); the object is unusable afterwards"""
        game, self.game = getattr(self, "game", None), None
        self._ext_stream = self._ext_side = None
        self._dev_cache, self._dev_slot, self._dev_guard = ({}, {}), {}, {}
        if game:
            try:
                self._lib.env_delete_game(game)
            except Exception:   # interpreter shutdown: ctypes / the library may already be gone
                pass

    def __del__(self):
        self.close()

    # ------------------------------------------------------------------ MI355X extensions (device buffers)
    def _require_device_api(self):
        if not getattr(self._lib, "has_device_api", False):
            raise RuntimeError("this library does not export the device-resident API")

    def get_observation_device(self, handle, view=None, feature=None):
        """Render observations straight into torch CUDA(HIP) tensors.

        Asynchronous contract of every *_device call: the work is queued on the engine's own (non-blocking) HIP stream
        and the call returns at once -- the outputs are valid after env.sync(), or for torch's stream after
        env.order_torch_after(); inputs / output buffers that torch's stream still produces or reads must be handed
        over with env.order_after_torch() first.  Buffers allocated inside these calls (view / feature / out = None)
        come from torch's caching allocator: keep them alive until the engine's work on them is done."""
        import torch
        self._require_device_api()
        g = _gid(handle)
        n = self.get_num(g)
        if view is None:
            view = torch.empty((n,) + self.view_space[g], dtype=torch.float32, device=torch.device("cuda", self.device_id))
        if feature is None:
            feature = torch.empty((n,) + self.feature_space[g], dtype=torch.float32, device=torch.device("cuda", self.device_id))
        assert view.is_contiguous() and feature.is_contiguous()
        vs = self.view_space[g]
        assert view.numel() >= n * vs[0] * vs[1] * vs[2] and feature.numel() >= n * self.feature_space[g][0]
        ptrs = (ctypes.c_void_p * 2)(view.data_ptr(), feature.data_ptr())
        self._lib.env_get_observation_device(self.game, g, ptrs)
        return view, feature

    def get_observation_device_bf16(self, handle, view=None, feature=None):
        """The observation in the policy kernels' input format: view as torch.bfloat16 [n, H, W, 8] (the channels of
        get_observation rounded to nearest even, zeros, 1.0 in channel 7), feature float32 [n, F].  Same asynchronous contract
        as get_observation_device; needs a game with at most 7 observation channels."""
        import torch
        self._require_device_api()
        g = _gid(handle)
        n = self.get_num(g)
        dev = torch.device("cuda", self.device_id)
        h, w, c = self.view_space[g]
        if c > 7:     # (the engine would abort the process: 8 bf16 per cell = 7 channels + the constant-1 bias channel)
            raise ValueError("bf16-cell observations hold at most 7 channels; this game has %d (use get_observation_device)" % c)
        if view is None:
            view = torch.empty((n, h, w, 8), dtype=torch.bfloat16, device=dev)
        if feature is None:
            feature = torch.empty((n,) + self.feature_space[g], dtype=torch.float32, device=dev)
        assert view.is_contiguous() and feature.is_contiguous() and view.dtype == torch.bfloat16
        assert view.numel() >= n * h * w * 8 and feature.numel() >= n * self.feature_space[g][0]
        ptrs = (ctypes.c_void_p * 2)(view.data_ptr(), feature.data_ptr())
        self._lib.env_get_observation_device_bf16(self.game, g, ptrs)
        return view, feature

    def set_action_device(self, handle, actions):
        """actions: int32 torch tensor on the env's device (must be complete on the env stream's timeline)"""
        self._require_device_api()
        assert actions.dtype.is_floating_point is False and actions.element_size() == 4 and actions.is_contiguous()
        self._lib.env_set_action_device(self.game, _gid(handle), actions.data_ptr())

    def get_reward_device(self, handle, out=None):
        import torch
        self._require_device_api()
        g = _gid(handle)
        if out is None:
            out = torch.empty(self.get_num(g), dtype=torch.float32, device=torch.device("cuda", self.device_id))
        self._lib.env_get_reward_device(self.game, g, out.data_ptr())
        return out

    def get_info_device(self, handle, name, out):
        self._require_device_api()
        self._lib.env_get_info_device(self.game, _gid(handle), name.encode(), out.data_ptr())
        return out

    def sync(self):
        self._require_device_api()
        self._lib.env_sync(self.game)

    @property
    def stream(self):
        """the engine's HIP stream as a torch.cuda.ExternalStream (every *_device call is asynchronous on it)"""
        # (asked for every time: an environment that joins an EnvBatch later gives up its own stream for the batch's, and a
        # wrapper of the destroyed one must not be used again)
        self._require_device_api()
        ptr = ctypes.c_void_p()
        self._lib.env_get_stream(self.game, ctypes.byref(ptr))
        st = getattr(self, "_ext_stream", None)
        if st is None or st.cuda_stream != ptr.value:
            import torch
            st = self._ext_stream = torch.cuda.ExternalStream(ptr.value, device=torch.device("cuda", self.device_id))
        return st

    def _streams(self):
        """the engine's stream(s): large worlds read their actions (and run the read-only head of the step) on a second stream
        beside the observation renders -- env_get_action_stream (include/magent_runtime_api.h)"""
        main = self.stream
        ptr = ctypes.c_void_p()
        self._lib.env_get_action_stream(self.game, ctypes.byref(ptr))
        if not ptr.value or ptr.value == main.cuda_stream:
            return (main,)
        side = getattr(self, "_ext_side", None)
        if side is None or side.cuda_stream != ptr.value:
            import torch
            side = self._ext_side = torch.cuda.ExternalStream(ptr.value, device=main.device)
        return (main, side)

    def order_after_torch(self):
        """engine work queued from now on waits for what is queued on torch's current stream (no host blocking)"""
        import torch
        for st in self._streams():
            cur = torch.cuda.current_stream(st.device)
            if cur.query():        # nothing is pending there: nothing to wait for (a stream query costs a tenth of an event record + wait:
                continue           # measured 15 us per 32-environment round, bench.py extra.battle_200_2x2000.cycle_32env_default_ordering)
            st.wait_stream(cur)

    def order_torch_after(self):
        """torch's current stream waits for the engine work queued so far (no host blocking)"""
        import torch
        for st in self._streams():
            torch.cuda.current_stream(st.device).wait_stream(st)

    @property
    def device_id(self):
        return self._device_id

    def engine_stats(self):
        """additive: (steps finished by the host-checked driver, attack rounds, move rounds of the last such step, attack
        rounds launched in the last step, steps whose ATTACK rounds ran out, steps whose MOVE rounds ran out, the kernel of the last
        observation render: 0 k_render, 1 k_render_fast, 4 k_render_sweep2)"""
        buf = np.zeros(8, dtype=np.int32)
        self._lib.env_get_info(self.game, 0, b"engine_stats", buf.ctypes.data)
        return tuple(int(v) for v in buf)

    def pipeline_stats(self):
        """additive (tests): (steps of the plain pipeline, of which launched with two optimistic pairs of death-rank rounds, with one, refills
        of the claim words for such steps, position in the current window of 63 epochs, steps whose optimistic rounds ran out, cycles through the BATCHED pipeline of env_cycle_many, of which rendered by the batch's sweeping kernel)"""
        buf = np.zeros(8, dtype=np.int32)
        self._lib.env_get_info(self.game, 0, b"pipeline_stats", buf.ctypes.data)
        return tuple(int(v) for v in buf)

    def round_hist(self):
        """additive (tuning): plain steps since the last call, by the last attack round that still changed a death rank"""
        buf = np.zeros(9, dtype=np.int32)
        self._lib.env_get_info(self.game, 0, b"round_hist", buf.ctypes.data)
        return tuple(int(v) for v in buf)

    def profile_enable(self, on=True):
        """on: False / 0 off, True / 1 every named phase, 2 only the observation render launches (cheap: for timed regions)"""
        self._require_device_api()
        self._lib.env_profile_enable(self.game, int(on))

    def profile_read(self, name):
        """-> (n_launches, total_ms) measured with HIP events on the env stream since the last read"""
        self._require_device_api()
        n, ms = ctypes.c_int32(0), ctypes.c_float(0)
        self._lib.env_profile_read(self.game, name.encode(), ctypes.byref(n), ctypes.byref(ms))
        return n.value, ms.value

    # ------------------------------------------------------------------ reward-rule serialisation
    def _send_reward_rules(self, config):
        """Flatten the event expressions into numbered symbols / nodes (protocol of gridworld.py:493-565).

        Numbering order is part of the protocol: receivers first, then the symbols met in a pre-order walk of the
        rule's expression; nodes in pre-order.  The reference passes 6 of the 7 arguments of
        gridworld_add_reward_rule (auto_value missing, gridworld.py:564-565); only OP_ALIGN reads it, so False here.
        """
        L, game = self._lib, self.game
        sym_no, node_no = {}, {}

        def walk_symbols(node):
            for item in node.inputs:
                if isinstance(item, EventNode):
                    walk_symbols(item)
                elif isinstance(item, AgentSymbol):
                    sym_no.setdefault(item, len(sym_no))

        def walk_nodes(node):
            node_no.setdefault(node, len(node_no))
            for item in node.inputs:
                if isinstance(item, EventNode):
                    walk_nodes(item)

        for on, receivers, _values, _terminal in config.reward_rules:
            for s in receivers:
                sym_no.setdefault(s, len(sym_no))
            walk_symbols(on)
        for on, _r, _v, _t in config.reward_rules:
            walk_nodes(on)

        for s, no in sym_no.items():
            L.gridworld_define_agent_symbol(game, no, s.group, s.index)
        for node, no in node_no.items():
            args = np.array([node_no[i] if isinstance(i, EventNode) else sym_no[i] if isinstance(i, AgentSymbol) else i
                             for i in node.inputs], dtype=np.int32)
            L.gridworld_define_event_node(game, no, node.op, _i32(args), len(args))
        for on, receivers, values, terminal in config.reward_rules:
            recv = np.array([sym_no[s] for s in receivers], dtype=np.int32)
            if len(values) == 1 and values[0] == "auto":
                vals = np.zeros(len(recv), dtype=np.float32)
            else:
                vals = np.array(values, dtype=np.float32)
            L.gridworld_add_reward_rule(game, node_no[on], _i32(recv), _f32(vals), len(recv), bool(terminal), False)


# ---------------------------------------------------------------------- reward description DSL
class EventNode(object):
    """AST node of an event expression; op codes are the engine's EventOp enum (grid_def.h:18-24)"""
    OP_AND, OP_OR, OP_NOT = 0, 1, 2
    OP_KILL, OP_AT, OP_IN, OP_COLLIDE, OP_ATTACK, OP_DIE, OP_IN_A_LINE, OP_ALIGN = 3, 4, 5, 6, 7, 8, 9, 10
    _BINARY = {"kill": OP_KILL, "attack": OP_ATTACK, "collide": OP_COLLIDE}
    _UNARY = {"die": OP_DIE, "in_a_line": OP_IN_A_LINE, "align": OP_ALIGN}

    def __init__(self, op=None, inputs=(), predicate=None):
        self.op, self.inputs, self.predicate = op, list(inputs), predicate

    def __call__(self, subject, predicate, *args):
        if predicate in self._BINARY:
            return EventNode(self._BINARY[predicate], [subject, args[0]], predicate)
        if predicate in self._UNARY:
            return EventNode(self._UNARY[predicate], [subject], predicate)
        if predicate == "at":
            return EventNode(self.OP_AT, [subject, args[0][0], args[0][1]], predicate)
        if predicate == "in":
            (xa, ya), (xb, yb) = args[0]
            return EventNode(self.OP_IN, [subject, min(xa, xb), min(ya, yb), max(xa, xb), max(ya, yb)], predicate)
        raise Exception("invalid predicate of event " + predicate)

    def __and__(self, other):
        return EventNode(self.OP_AND, [self, other])

    def __or__(self, other):
        return EventNode(self.OP_OR, [self, other])

    def __invert__(self):
        return EventNode(self.OP_NOT, [self])


Event = EventNode()


class AgentSymbol(object):
    """a group member in an event: index 'any' (-1), 'all' (-2) or a fixed int"""
    def __init__(self, group, index):
        self.group = -1 if group is None else group
        self.index = {"any": -1, "all": -2}.get(index, index)
        assert isinstance(self.index, int), "index must be 'any', 'all' or an int"

    def __str__(self):
        return "agent(%d,%d)" % (self.group, self.index)


class Config(object):
    """game description: global settings, agent types, groups, reward rules"""
    def __init__(self):
        self.config_dict, self.agent_type_dict, self.groups, self.reward_rules = {}, {}, [], []

    def set(self, args):
        self.config_dict.update(args)

    def register_agent_type(self, name, attr):
        if name in self.agent_type_dict:
            raise Exception("type name %s already exists" % name)
        self.agent_type_dict[name] = attr
        return name

    def add_group(self, agent_type):
        self.groups.append(agent_type)
        return len(self.groups) - 1

    def add_reward_rule(self, on, receiver, value, terminal=False):
        if not isinstance(receiver, (tuple, list)):
            receiver, value = [receiver], [value]
        if len(receiver) != len(value):
            raise Exception("the length of receiver and value should be equal")
        self.reward_rules.append([on, list(receiver), list(value), terminal])


class CircleRange(object):
    def __init__(self, radius):
        self.radius, self.angle = radius, 360

    def __str__(self):
        return "circle(%g)" % self.radius


class SectorRange(object):
    def __init__(self, radius, angle):
        if angle >= 180:
            raise Exception("the angle of a sector should be smaller than 180 degree")
        self.radius, self.angle = radius, angle

    def __str__(self):
        return "sector(%g, %g)" % (self.radius, self.angle)


def step_many(envs):
    """env.step() for several independent environments with their device work overlapped (one host thread): every
    step is enqueued on its environment's stream before the first is waited for.  Returns the list of done flags."""
    lib = envs[0]._lib
    n = len(envs)
    handles = (ctypes.c_void_p * n)(*[e.game for e in envs])
    done = (ctypes.c_int32 * n)()
    lib.env_step_many(handles, n, done)
    return [bool(d) for d in done]


class EnvBatch(object):
    """Several independent environments on one GPU, cycled together by host threads inside the library.

    cycle(views, feats, actions, rewards): per environment and group, observe into the given device tensors, set the
    device actions, step, fetch rewards, clear_dead -- one library call for the whole batch (env_cycle_many)."""

    def __init__(self, envs, n_threads=8):
        self.envs, self.n_threads = list(envs), n_threads
        self._lib = self.envs[0]._lib
        self.n_group = len(self.envs[0].group_handles)
        n = len(self.envs)
        self._handles = (ctypes.c_void_p * n)(*[e.game for e in self.envs])
        self._done = (ctypes.c_int32 * n)()
        self._adopted = False
        self.order_streams = True
        self._stream_ptrs = (ctypes.c_void_p * (2 * n))()
        self._uniq, self._uniq_key = None, None
        self._nums = None

    def pointers(self, tensors):
        """the device-pointer array of a list (per env) of lists (per group) of CUDA tensors (None entries allowed).  cycle()
        takes such an array in place of the nested list: a caller that reuses its buffers builds it once -- at 8 small
        environments the 64 data_ptr() calls per cycle cost as much as the step itself.  The tensors must stay alive."""
        n = len(self.envs) * self.n_group
        arr = (ctypes.c_void_p * n)()
        if tensors is not None:
            for e, per_env in enumerate(tensors):
                for g, t in enumerate(per_env):
                    arr[e * self.n_group + g] = None if t is None else t.data_ptr()
        return arr

    def _ptrs(self, tensors):
        return tensors if isinstance(tensors, ctypes.Array) else self.pointers(tensors)

    def nums(self):
        """agent counts [env][group] (host mirror, no device work)"""
        return self.nums_array().tolist()

    def nums_array(self):
        """the same as an int32 numpy array [env][group] over a buffer of the batch's own (valid until the next call): one library call and
        no per-environment Python work -- at 32 environments the nested lists cost a caller's loop 8 us of a 225 us round"""
        if self._nums is None:
            self._nums_c = (ctypes.c_int32 * (len(self.envs) * self.n_group))()
            self._nums = np.frombuffer(self._nums_c, dtype=np.int32).reshape(len(self.envs), self.n_group)
        self._lib.env_num_many(self._handles, len(self.envs), self.n_group, self._nums_c)
        return self._nums

    def cycle(self, views=None, feats=None, actions=None, rewards=None):
        """each argument: list (per env) of lists (per group) of CUDA tensors or None, or the result of pointers();
        returns the done flags.

        Stream contract.  Outputs: the call returns after the host has seen every environment's step record, which the step
        kernel publishes behind an agent-scope release and a workgroup barrier -- rewards, observations and the compacted state
        are complete in device memory, and a kernel launched afterwards on ANY stream sees them (no stream ordering needed;
        measured: an event pair per cycle costs a 4000-agent world 30 us of its 110).  Inputs: the library reads `actions` on
        the environments' own streams; with `order_streams` (default on) those streams are first ordered after torch's current
        stream, the producer of the actions, stream to stream.  Callers that synchronise themselves switch it off."""
        if self.order_streams:
            for e in self._distinct():
                e.order_after_torch()
        return self._cycle_raw(views, feats, actions, rewards)

    def _distinct(self):
        """one environment per distinct engine stream.  Batched environments share their leader's stream, and an environment may join
        the batch in a later cycle: the streams are asked for every cycle -- ONE library call for all of them (env_streams_many) -- and
        the list is rebuilt only when a pointer has changed (two ctypes calls per environment and cycle cost a 64-
        environment batch as much as the cycle itself)"""
        self._lib.env_streams_many(self._handles, len(self.envs), self._stream_ptrs)
        key = bytes(self._stream_ptrs)
        if key != self._uniq_key:
            seen, uniq = set(), []
            for k, e in enumerate(self.envs):
                pair = (self._stream_ptrs[2 * k], self._stream_ptrs[2 * k + 1])
                if pair not in seen:
                    seen.add(pair)
                    uniq.append(e)
            self._uniq, self._uniq_key = uniq, key
        return self._uniq

    def _cycle_raw(self, views, feats, actions, rewards):
        self._lib.env_cycle_many(self._handles, len(self.envs), self.n_group, self._ptrs(views), self._ptrs(feats),
                                 self._ptrs(actions), self._ptrs(rewards), self._done, self.n_threads)
        if not self._adopted:      # environments cycled together share the first one's stream from now on
            self._adopted = True
            self._uniq_key = None
            for e in self.envs:
                e._ext_stream = None
        return [bool(d) for d in self._done]


def _builtin_config(name, **kwargs):
    try:
        mod = importlib.import_module("magent_amd.builtin.config." + name)
    except ImportError:
        raise BaseException('unknown built-in game "' + name + '"')
    return mod.get_config(**kwargs)
