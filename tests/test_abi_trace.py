"""Replay of a C-ABI transcript recorded from the reference's own Python wrapper.

tests/golden/abi_trace_battle.bin.gz holds every call /root/reference/python/magent/gridworld.py (unmodified) made during an episode --
game and agent-type configuration, reward rules, walls, random / custom placement, nine steps of observe / set_action / step /
rewards / alive / pos / clear_dead, a mid-episode add_agents -- with the arguments as they crossed src/runtime_api.h, and the bytes
the COMPILED REFERENCE wrote back (tests/golden/make_abi_trace.py, tests/abi_trace/trace_shim.c).  Here the transcript is replayed,
call by call, through ctypes WITHOUT argtypes -- the way the reference's c_lib.py calls (c_lib.py:25-41: plain Python ints, byref
of c_int / c_bool, numpy pointers) -- against another engine, and every returned byte is compared.  The GPU leg is the answer to
"does the reference's wrapper, as it is, drive the new library on an MI355X": the reference tree does not travel to the GPU box, its
exact call sequence does."""
import ctypes
import gzip
import os
import struct

import numpy as np
import pytest

import helpers as H

TRACE = os.path.join(H.GOLDEN_DIR, "abi_trace_battle.bin.gz")
(F_NEW_GAME, F_DELETE_GAME, F_CONFIG_GAME, F_RESET, F_GET_OBSERVATION, F_SET_ACTION, F_STEP, F_GET_REWARD, F_GET_INFO, F_RENDER, F_RENDER_NEXT_FILE,
 F_REGISTER_AGENT_TYPE, F_NEW_GROUP, F_ADD_AGENTS, F_CLEAR_DEAD, F_SET_GOAL, F_DEFINE_AGENT_SYMBOL, F_DEFINE_EVENT_NODE, F_ADD_REWARD_RULE) = range(19)


def records():
    data = gzip.open(TRACE, "rb").read()
    off = 0
    while off < len(data):
        func, n = struct.unpack_from("<II", data, off)
        off += 8
        fields = []
        for _ in range(n):
            kind, nb = struct.unpack_from("<II", data, off)
            off += 8
            fields.append((kind, data[off:off + nb]))
            off += nb
        yield func, fields


def i32(b):
    return struct.unpack("<i", b)[0]


def replay(lib_path):
    lib = ctypes.CDLL(lib_path, mode=os.RTLD_LOCAL)      # no argtypes, no restype: as the reference's c_lib.py
    game = ctypes.c_void_p()
    calls = compared = 0
    for func, f in records():
        calls += 1
        if func == F_NEW_GAME:
            lib.env_new_game(ctypes.byref(game), f[0][1])
        elif func == F_DELETE_GAME:
            lib.env_delete_game(game)
        elif func == F_CONFIG_GAME:
            key, val = f[0][1], f[1][1]
            if key == b"render_dir":
                lib.env_config_game(game, key, ctypes.c_char_p(val))
            elif len(val) == 1:
                lib.env_config_game(game, key, ctypes.byref(ctypes.c_bool(val != b"\x00")))
            else:
                lib.env_config_game(game, key, ctypes.byref(ctypes.c_int(i32(val))))
        elif func == F_RESET:
            lib.env_reset(game)
        elif func == F_REGISTER_AGENT_TYPE:
            name, n = f[0][1], i32(f[1][1])
            keys = (ctypes.c_char_p * n)(*[f[2 + 2 * k][1] for k in range(n)])
            values = (ctypes.c_float * n)(*[struct.unpack("<f", f[3 + 2 * k][1])[0] for k in range(n)])
            lib.gridworld_register_agent_type(game, name, n, keys, values)
        elif func == F_NEW_GROUP:
            handle = ctypes.c_int32()
            lib.gridworld_new_group(game, ctypes.c_char_p(f[0][1]), ctypes.byref(handle))
            assert handle.value == i32(f[1][1]); compared += 1
        elif func == F_ADD_AGENTS:
            group, n, method = ctypes.c_int32(i32(f[0][1])), i32(f[1][1]), f[2][1]
            if method == b"custom":
                xs, ys, ds = (np.frombuffer(f[k][1], dtype=np.int32).copy() for k in (3, 4, 5))
                lib.gridworld_add_agents(game, group, n, method, xs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                         ys.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ds.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
            elif method in (b"fill", b"maze"):
                bind = np.frombuffer(f[3][1], dtype=np.int32).copy()
                lib.gridworld_add_agents(game, group, 0, method, bind.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 0, 0, 0)
            else:
                lib.gridworld_add_agents(game, group, n, method, 0, 0, 0)
        elif func == F_DEFINE_AGENT_SYMBOL:
            lib.gridworld_define_agent_symbol(game, i32(f[0][1]), i32(f[1][1]), i32(f[2][1]))
        elif func == F_DEFINE_EVENT_NODE:
            inputs = np.frombuffer(f[2][1], dtype=np.int32).copy()
            lib.gridworld_define_event_node(game, i32(f[0][1]), i32(f[1][1]), inputs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(inputs))
        elif func == F_ADD_REWARD_RULE:
            recv, val = np.frombuffer(f[1][1], dtype=np.int32).copy(), np.frombuffer(f[2][1], dtype=np.float32).copy()
            lib.gridworld_add_reward_rule(game, i32(f[0][1]), recv.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                          val.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(recv), bool(i32(f[3][1])))   # six arguments, like gridworld.py:564-565
        elif func == F_GET_INFO:
            group, name, want = ctypes.c_int32(i32(f[0][1])), f[1][1], f[2][1]
            buf = np.full(len(want), 0x5A, dtype=np.uint8)
            lib.env_get_info(game, group, name, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
            assert buf.tobytes() == want, "env_get_info(%s) of call %d" % (name.decode(), calls); compared += 1
        elif func == F_GET_OBSERVATION:
            group = ctypes.c_int32(i32(f[0][1]))
            view, feat = np.full(len(f[1][1]), 0x5A, dtype=np.uint8), np.full(len(f[2][1]), 0x5A, dtype=np.uint8)
            bufs = (ctypes.POINTER(ctypes.c_float) * 2)()
            bufs[0], bufs[1] = view.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), feat.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            lib.env_get_observation(game, group, bufs)
            assert view.tobytes() == f[1][1], "view of call %d" % calls
            assert feat.tobytes() == f[2][1], "feature of call %d" % calls
            compared += 2
        elif func == F_SET_ACTION:
            acts = np.frombuffer(f[1][1], dtype=np.int32).copy()
            lib.env_set_action(game, ctypes.c_int32(i32(f[0][1])), acts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        elif func == F_STEP:
            done = ctypes.c_int32()
            lib.env_step(game, ctypes.byref(done))
            assert done.value == i32(f[0][1]), "done of call %d" % calls; compared += 1
        elif func == F_GET_REWARD:
            buf = np.full(len(f[1][1]), 0x5A, dtype=np.uint8)
            lib.env_get_reward(game, ctypes.c_int32(i32(f[0][1])), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
            assert buf.tobytes() == f[1][1], "rewards of call %d" % calls; compared += 1
        elif func == F_CLEAR_DEAD:
            lib.gridworld_clear_dead(game)
        else:
            raise AssertionError("call %d: function %d is not in the transcript's set" % (calls, func))
    return calls, compared


def test_transcript_covers_the_wrapper_surface():
    seen = {func for func, _ in records()}
    for must in (F_NEW_GAME, F_CONFIG_GAME, F_REGISTER_AGENT_TYPE, F_NEW_GROUP, F_DEFINE_AGENT_SYMBOL, F_DEFINE_EVENT_NODE, F_ADD_REWARD_RULE, F_RESET,
                 F_ADD_AGENTS, F_GET_INFO, F_GET_OBSERVATION, F_SET_ACTION, F_STEP, F_GET_REWARD, F_CLEAR_DEAD, F_DELETE_GAME):
        assert must in seen


def test_replay_on_the_oracle_and_on_the_emulated_kernels():
    for lib in (H.ensure_oracle(), H.ensure_emu()):
        calls, compared = replay(lib)
        assert calls > 250 and compared > 150


@pytest.mark.gpu
def test_replay_on_the_hip_engine():
    calls, compared = replay(H.HIP_LIB)
    assert calls > 250 and compared > 150
