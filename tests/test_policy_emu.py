"""The MFMA policy kernels (magent_amd/csrc/policy.hip) on the CPU emulator (tests/hipemu: every lane a fiber, v_mfma_f32_32x32x16_bf16
as a wave-wide meeting with the lane maps measured on the MI355X) against the same PyTorch f32 computation tests/test_policy.py
uses on the GPU.  What this checks in the GPU-less container: the LDS images and their swizzles, the look-up tables, the tile /
chunk bookkeeping, the prefetch rings and every barrier (the emulator runs the lanes of a workgroup one after the other between
meetings: a missing barrier shows as stale data).  TEST INFRASTRUCTURE: the product is the HIP build, checked by test_policy.py."""
import ctypes
import os

import numpy as np
import pytest

import helpers as H


def _emu():
    os.environ["MAGENT_TUNE"] = "policy_grid=3"      # (read once, at the emulated library's first call: three workgroups walk every tile of the conv kernel)
    lib = ctypes.CDLL(H.ensure_emu())
    lib.policy_dqn_infer.restype = ctypes.c_int
    lib.policy_dqn_infer_bf16.restype = ctypes.c_int
    return lib


def _infer(lib, pol, view, feat, cells16=False):
    import torch
    if pol.dirty:
        pol.pack()
    n = view.shape[0]
    actions = torch.empty(n, dtype=torch.int32)
    q = torch.empty((n, pol.shape.n_action), dtype=torch.float32)
    nbytes = ctypes.c_size_t(0)
    lib.policy_dqn_act_bytes(ctypes.byref(pol.shape), ctypes.c_int(n), ctypes.byref(nbytes))
    work = torch.zeros(nbytes.value, dtype=torch.uint8)
    call = lib.policy_dqn_infer_bf16 if cells16 else lib.policy_dqn_infer
    rc = call(ctypes.byref(pol.shape), ctypes.byref(pol._w), ctypes.c_void_p(view.data_ptr()), ctypes.c_void_p(feat.data_ptr()), ctypes.c_int(n),
              ctypes.c_void_p(work.data_ptr()), ctypes.c_void_p(actions.data_ptr()), ctypes.c_void_p(q.data_ptr()), None)
    assert rc == 0
    return actions, q


@pytest.mark.parametrize("view_space,feat,n_action,n", [((13, 13, 7), 34, 21, 128 + 37), ((9, 9, 5), 18, 9, 77), ((13, 11, 6), 40, 31, 70),
                                                         ((7, 7, 3), 5, 5, 131), ((13, 13, 7), 34, 21, 1), ((5, 5, 1), 1, 2, 40), ((16, 16, 4), 36, 13, 9), ((15, 15, 7), 64, 31, 6)])
def test_emulated_policy_matches_torch_reference(view_space, feat, n_action, n):
    import torch
    from test_policy import _reference
    from magent_amd.builtin.torch_model.dqn import _QNet
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicy
    lib = _emu()
    torch.manual_seed(4321 + n)
    qnet = _QNet(view_space, (feat,), n_action, True, True)
    with torch.no_grad():
        for p in qnet.parameters():
            p.mul_(3.0)
    view = ((torch.rand((n,) + view_space) < 0.3).float() * torch.rand((n,) + view_space)).contiguous()
    featv = (torch.rand((n, feat)) * 2 - 0.5).contiguous()
    pol = HipDqnPolicy(qnet, view_space, (feat,), n_action, "cpu")
    actions, q = _infer(lib, pol, view, featv)
    with torch.no_grad():
        ref = _reference(qnet, view, featv)
    scale = float(ref.abs().max())
    err = (q - ref).abs().max().item()
    assert err <= 2e-3 * scale + 2e-3, (err, scale)
    assert torch.equal(actions.long(), q.argmax(dim=1))
    # the bf16-cell entry point on the same views: identical operands reach the same MFMAs
    cells = torch.zeros((n,) + view_space[:2] + (8,), dtype=torch.bfloat16)
    cells[..., :view_space[2]] = view.to(torch.bfloat16)
    cells[..., 7] = 1.0
    a16, q16 = _infer(lib, pol, cells.contiguous(), featv, cells16=True)
    assert torch.equal(q16, q) and torch.equal(a16, actions)


@pytest.mark.parametrize("seed", ["5", "19"])
def test_emulated_policy_in_scrambled_order(seed):
    """HIPEMU_SCRAMBLE: the waves of a workgroup (and the workgroups of a launch) run in a pseudo-random order that changes at every
    scheduling pass -- a wave may run a whole chunk ahead of its neighbours between two barriers.  The three-buffer ring of the head
    and the staging of the conv kernel must not care: the same cases, in a process of their own (the variable is read at the first launch)"""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import test_policy_emu as T\n"
            "for case in (((13, 13, 7), 34, 21, 128 + 37), ((9, 9, 5), 18, 9, 77), ((7, 7, 3), 5, 5, 131)):\n"
            "    T.test_emulated_policy_matches_torch_reference(*case)\n"
            "print('scrambled ok')\n") % os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPEMU_SCRAMBLE=seed, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "scrambled ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_emulated_policy_random_shapes():
    """seeded random shapes inside policy_dqn_supported: views 5..16 on a side with H x W <= 256, 1..7 channels, 1..64 features, 1..31 actions,
    agent counts that leave partial conv tiles and partial head groups"""
    rs = np.random.RandomState(2024)
    done = 0
    while done < 6:
        h, w = int(rs.randint(5, 17)), int(rs.randint(5, 17))
        if h * w > 256:
            continue
        c, feat, n_action = int(rs.randint(1, 8)), int(rs.randint(1, 65)), int(rs.randint(1, 32))
        n = int(rs.randint(1, 60)) if h * w > 100 else int(rs.randint(1, 200))
        test_emulated_policy_matches_torch_reference((h, w, c), feat, n_action, n)
        done += 1


# ---------------------------------------------------------------------------------------------------- float32 (policy_f32.hip)
def _infer_f32(lib, pol, view, feat):
    import torch
    lib.policy_dqn_infer_f32.restype = ctypes.c_int
    if pol.dirty:
        pol.pack()
    n = view.shape[0]
    actions = torch.empty(n, dtype=torch.int32)
    q = torch.empty((n, pol.shape.n_action), dtype=torch.float32)
    nbytes = ctypes.c_size_t(0)
    lib.policy_dqn_f32_act_bytes(ctypes.byref(pol.shape), ctypes.c_int(n), ctypes.byref(nbytes))
    work = torch.zeros(nbytes.value, dtype=torch.uint8)
    rc = lib.policy_dqn_infer_f32(ctypes.byref(pol.shape), ctypes.byref(pol._w), ctypes.c_void_p(view.data_ptr()), ctypes.c_void_p(feat.data_ptr()), ctypes.c_int(n),
                                  ctypes.c_void_p(work.data_ptr()), ctypes.c_void_p(actions.data_ptr()), ctypes.c_void_p(q.data_ptr()), None)
    assert rc == 0
    return actions, q


@pytest.mark.parametrize("view_space,feat,n_action,n", [((13, 13, 7), 34, 21, 128 + 37), ((9, 9, 5), 18, 9, 77), ((13, 11, 6), 40, 31, 70),
                                                         ((7, 7, 3), 5, 5, 131), ((13, 13, 7), 34, 21, 1), ((5, 5, 1), 1, 2, 40), ((15, 15, 7), 56, 31, 6)])
def test_emulated_f32_policy_matches_the_torch_network(view_space, feat, n_action, n):
    """k_dqn_conv_f32 + k_dqn_head_f32 (v_mfma_f32_32x32x2_f32 as a wave-wide meeting: an exact k-ordered fmaf chain, as on the hardware) against
    the PyTorch float32 network itself -- no rounding anywhere, so the tolerance is float32 round-off: 1e-5 of max |Q|"""
    import torch
    from magent_amd.builtin.torch_model.dqn import _QNet
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicyF32
    lib = _emu()
    torch.manual_seed(977 + n)
    qnet = _QNet(view_space, (feat,), n_action, True, True)
    with torch.no_grad():
        for p in qnet.parameters():
            p.mul_(3.0)
    view = ((torch.rand((n,) + view_space) < 0.3).float() * torch.rand((n,) + view_space)).contiguous()
    featv = (torch.rand((n, feat)) * 2 - 0.5).contiguous()
    pol = HipDqnPolicyF32(qnet, view_space, (feat,), n_action, "cpu")
    actions, q = _infer_f32(lib, pol, view, featv)
    with torch.no_grad():
        ref = qnet(view, featv)
    scale = float(ref.abs().max())
    err = (q - ref).abs().max().item()
    assert err <= 1e-5 * scale + 1e-6, (err, scale)
    assert torch.equal(actions.long(), q.argmax(dim=1))


def test_emulated_f32_policy_in_scrambled_order():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import test_policy_emu as T\n"
            "for case in (((13, 13, 7), 34, 21, 128 + 37), ((9, 9, 5), 18, 9, 77), ((7, 7, 3), 5, 5, 131)):\n"
            "    T.test_emulated_f32_policy_matches_the_torch_network(*case)\n"
            "print('scrambled ok')\n") % os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPEMU_SCRAMBLE="23", OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "scrambled ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
