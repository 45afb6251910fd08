"""The hand-written bf16 MFMA inference kernels of the reference's deep Q network (magent_amd/csrc/policy.hip) against a plain
PyTorch f32 computation of the same network that rounds to bf16 at the same points (inputs, weights, activations between
layers) -- so that what is left is the order of the f32 accumulation.

Tolerance: |Q_hip - Q_ref| <= 2e-3 * max|Q_ref| + 2e-3 per entry.  A bf16 activation that sits on a rounding boundary may round
the other way under a different summation order (one bf16 ulp = 2^-8 relative, on one of 2592 + 256 inputs of the next layer):
that, not the f32 sums themselves (1e-6), is what the bound covers.  Greedy actions must agree wherever the reference's best and
second-best Q are further apart than that bound."""
import numpy as np
import pytest



def _reference(qnet, view, feat):
    import torch
    import torch.nn.functional as F
    r = lambda t: t.to(torch.bfloat16).float()
    x = r(view).permute(0, 3, 1, 2)
    x = r(F.relu(F.conv2d(x, r(qnet.conv1.weight), r(qnet.conv1.bias))))      # (conv1's bias rides in the MFMA as a weight: bf16)
    x = r(F.relu(F.conv2d(x, r(qnet.conv2.weight), qnet.conv2.bias.float())))
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    hv = r(F.relu(F.linear(x, r(qnet.dense_view.weight), qnet.dense_view.bias.float())))
    he = r(F.relu(F.linear(r(feat), r(qnet.dense_emb.weight), qnet.dense_emb.bias.float())))
    h = torch.cat([hv, he], dim=1)
    adv = F.linear(h, r(qnet.advantage.weight))
    val = F.linear(h, r(qnet.value.weight), qnet.value.bias.float())
    return val + adv - adv.mean(dim=1, keepdim=True)


@pytest.mark.gpu
@pytest.mark.parametrize("view_space,feat,n_action,n", [((13, 13, 7), 34, 21, 1000), ((13, 13, 7), 34, 21, 64 * 6 + 5),
                                                         ((9, 9, 5), 18, 9, 777), ((13, 11, 6), 40, 31, 300), ((7, 7, 3), 5, 5, 131),
                                                         ((13, 13, 7), 34, 21, 1), ((13, 13, 7), 34, 21, 7), ((5, 5, 1), 1, 2, 40),
                                                         ((16, 16, 4), 36, 13, 300), ((15, 15, 7), 64, 31, 200)])
def test_hip_policy_matches_torch_reference(view_space, feat, n_action, n):
    import torch
    from magent_amd.builtin.torch_model.dqn import _QNet
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicy
    torch.manual_seed(1234 + n)
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.allow_tf32 = False
    qnet = _QNet(view_space, (feat,), n_action, True, True).to(dev)
    with torch.no_grad():
        for p in qnet.parameters():          # larger weights than the default init: every layer's output matters in Q
            p.mul_(3.0)
    # observation-like inputs: sparse 0/1 channels, fractions, a few large values
    view = (torch.rand((n,) + view_space, device=dev) < 0.3).float() * torch.rand((n,) + view_space, device=dev)
    featv = torch.rand((n, feat), device=dev) * 2 - 0.5
    pol = HipDqnPolicy(qnet, view_space, (feat,), n_action, dev, chunk=512)     # several chunks on the larger cases
    actions, q = pol.infer(view, featv, want_q=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = _reference(qnet, view, featv)
    scale = float(ref.abs().max())
    err = (q - ref).abs().max().item()
    assert err <= 2e-3 * scale + 2e-3, (err, scale)
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * (2e-3 * scale + 2e-3)
    assert n < 50 or clear.float().mean().item() > 0.5          # the comparison below is not vacuous
    assert torch.equal(actions[clear].long(), ref.argmax(dim=1)[clear])
    assert torch.equal(actions.long(), q.argmax(dim=1))      # the kernel's own argmax (first index on ties)
    # actions only (no Q output) give the same answer
    assert torch.equal(pol.infer(view, featv), actions)


@pytest.mark.gpu
def test_hip_policy_many_tiles_per_workgroup():
    """more agents per launch than the persistent conv grid has workgroups x 4: every workgroup walks several tiles (the register
    prefetch a tile ahead, the staging in the middle of a tile), the head runs several rounds of workgroups; float32 and bf16-cell views"""
    import torch
    from magent_amd.builtin.torch_model.dqn import _QNet
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicy
    torch.manual_seed(77)
    dev = torch.device("cuda", 0)
    view_space, feat, n_action, n = (13, 13, 7), 34, 21, 4 * 512 * 5 + 77 * 4 + 3
    qnet = _QNet(view_space, (feat,), n_action, True, True).to(dev)
    with torch.no_grad():
        for p in qnet.parameters():
            p.mul_(3.0)
    view = (torch.rand((n,) + view_space, device=dev) < 0.3).float() * torch.rand((n,) + view_space, device=dev)
    featv = torch.rand((n, feat), device=dev) * 2 - 0.5
    pol = HipDqnPolicy(qnet, view_space, (feat,), n_action, dev, chunk=1 << 20)
    actions, q = pol.infer(view, featv, want_q=True)
    with torch.no_grad():
        ref = _reference(qnet, view, featv)
    scale = float(ref.abs().max())
    assert (q - ref).abs().max().item() <= 2e-3 * scale + 2e-3
    assert torch.equal(actions.long(), q.argmax(dim=1))
    cells = torch.zeros((n,) + view_space[:2] + (8,), dtype=torch.bfloat16, device=dev)
    cells[..., :7] = view.to(torch.bfloat16); cells[..., 7] = 1
    a16, q16 = pol.infer(cells, featv, want_q=True)
    assert torch.equal(q16, q) and torch.equal(a16, actions)


@pytest.mark.gpu
def test_bf16_policy_against_the_f32_network():
    """the bf16 MFMA kernels against the reference's OWN arithmetic -- the float32 network of tf_model/dqn.py:151-189 (here the
    PyTorch _QNet, pinned to a NumPy restatement of the TensorFlow graph in tests/test_training.py) -- on real observations: battle
    1000 x 1000, 2 x 400k agents, several steps into an episode (attacks, deaths, hp fractions and minimaps in the views).
    Stated bound: max |Q_bf16 - Q_f32| <= 2 % of max |Q_f32|, and the greedy actions agree for >= 97 % of the agents (every
    disagreement sits where the f32 network's best two Q values are closer than twice the measured error).  The numbers are
    printed; DeepQNetwork only takes this path when the caller opts in (infer_dtype="bf16")."""
    import torch
    import magent_amd
    from magent_amd.builtin.torch_model import DeepQNetwork
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    env = magent_amd.GridWorld("battle", map_size=1000, device_obs=True)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=400000)
    torch.manual_seed(7)
    m = DeepQNetwork(env, hs[0], "pin", memory_size=16, infer_dtype="bf16")
    assert m._hip is not None and m.infer_dtype == "bf16"
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicyF32
    assert isinstance(DeepQNetwork(env, hs[1], "dflt", memory_size=16)._hip, HipDqnPolicyF32)           # float32 unless asked otherwise
    rs = np.random.RandomState(5)
    for step in range(6):
        for h in hs:
            env.set_action(h, torch.from_numpy(rs.randint(21, size=env.get_num(h)).astype(np.int32)).cuda())
        env.step(); env.clear_dead()
    worst, agree, total, unexplained = 0.0, 0, 0, 0
    for h in hs:
        view, feat = env.get_observation(h); env.sync()
        a16, q16 = m._hip.infer(view, feat, want_q=True)
        q32 = torch.cat([m.qnet(view[b:b + 65536], feat[b:b + 65536]) for b in range(0, len(view), 65536)]).detach()
        torch.cuda.synchronize()
        scale = float(q32.abs().max())
        err = float((q16 - q32).abs().max())
        worst = max(worst, err / scale)
        same = a16.long() == q32.argmax(dim=1)
        top2 = q32.topk(2, dim=1).values
        unexplained += int((~same & ((top2[:, 0] - top2[:, 1]) > 2 * err)).sum())
        agree += int(same.sum()); total += len(same)
    print("bf16 policy vs f32 network on %d real observations: max |dQ| / max |Q| = %.4f, greedy actions equal for %.2f %%" % (total, worst, 100.0 * agree / total))
    assert worst <= 0.02, worst
    assert agree >= 0.97 * total, (agree, total)
    assert unexplained == 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("view_space,feat,n_action,n", [((13, 13, 7), 34, 21, 1000), ((13, 13, 7), 34, 21, 4 * 256 * 5 + 77 * 4 + 3), ((9, 9, 5), 18, 9, 777),
                                                         ((13, 11, 6), 40, 31, 300), ((7, 7, 3), 5, 5, 131), ((13, 13, 7), 34, 21, 1), ((5, 5, 1), 1, 2, 40),
                                                         ((15, 15, 7), 36, 33 - 2, 200), ((16, 16, 4), 56, 13, 300)])
def test_hip_f32_policy_matches_the_torch_network(view_space, feat, n_action, n):
    """k_dqn_conv_f32 + k_dqn_head_f32 (magent_amd/csrc/policy_f32.hip: float32 in, float32 accumulate, v_mfma_f32_32x32x2_f32) against the
    PyTorch float32 network itself.  Nothing is rounded anywhere: the only difference is the order of the float32 sums (the matrix
    instruction adds in k order, MIOpen / rocBLAS in theirs) -- tolerance 1e-4 of max |Q|, measured ~1e-6"""
    import torch
    from magent_amd.builtin.torch_model.dqn import _QNet
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicyF32
    torch.manual_seed(4321 + n)
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    qnet = _QNet(view_space, (feat,), n_action, True, True).to(dev)
    with torch.no_grad():
        for p in qnet.parameters():
            p.mul_(3.0)
    view = (torch.rand((n,) + view_space, device=dev) < 0.3).float() * torch.rand((n,) + view_space, device=dev)
    featv = torch.rand((n, feat), device=dev) * 2 - 0.5
    pol = HipDqnPolicyF32(qnet, view_space, (feat,), n_action, dev, chunk=2048)     # several chunks on the larger cases
    actions, q = pol.infer(view, featv, want_q=True)
    with torch.no_grad():
        ref = qnet(view, featv)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    err = (q - ref).abs().max().item()
    assert err <= 1e-4 * scale, (err, scale)
    assert torch.equal(actions.long(), q.argmax(dim=1))
    assert torch.equal(pol.infer(view, featv), actions)


@pytest.mark.gpu
def test_f32_policy_on_real_observations_and_its_rate():
    """BASELINE config 5's policy step at the reference's precision: battle 1000 x 1000, 2 x 400k agents six steps into an episode, every
    agent's Q values from the HIP float32 kernels against the PyTorch float32 network (max |dQ| <= 1e-4 max |Q|; greedy actions equal
    except where the network's best two Q values are closer than twice the measured error), and both forward passes timed: the line is
    printed (profiles/r06_summary.md keeps it), the kernels must beat PyTorch's rate"""
    import time
    import torch
    import magent_amd
    from magent_amd.builtin.torch_model import DeepQNetwork
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicyF32
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    env = magent_amd.GridWorld("battle", map_size=1000, device_obs=True)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=400000)
    torch.manual_seed(7)
    m = DeepQNetwork(env, hs[0], "pin", memory_size=16)
    assert isinstance(m._hip, HipDqnPolicyF32) and m.infer_dtype == "f32"
    rs = np.random.RandomState(5)
    for step in range(6):
        for h in hs:
            env.set_action(h, torch.from_numpy(rs.randint(21, size=env.get_num(h)).astype(np.int32)).cuda())
        env.step(); env.clear_dead()
    worst, agree, total, unexplained = 0.0, 0, 0, 0
    t_hip = t_torch = 0.0
    for h in hs:
        view, feat = env.get_observation(h); env.sync()
        m._hip.infer(view, feat)      # (packs the weights, sizes the workspace)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a32, q_hip = m._hip.infer(view, feat, want_q=True)
        torch.cuda.synchronize(); t_hip += time.perf_counter() - t0
        m.qnet(view[:65536], feat[:65536])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q32 = torch.cat([m.qnet(view[b:b + 65536], feat[b:b + 65536]) for b in range(0, len(view), 65536)]).detach()
        torch.cuda.synchronize(); t_torch += time.perf_counter() - t0
        scale = float(q32.abs().max())
        err = float((q_hip - q32).abs().max())
        worst = max(worst, err / scale)
        same = a32.long() == q32.argmax(dim=1)
        top2 = q32.topk(2, dim=1).values
        unexplained += int((~same & ((top2[:, 0] - top2[:, 1]) > 2 * err)).sum())
        agree += int(same.sum()); total += len(same)
    flop = 2.0 * (11 * 11 * 32 * 63 + 9 * 9 * 32 * 288 + 2592 * 256 + 34 * 256 + 512 * 22) * total
    print("f32 policy on %d real observations: max |dQ| / max |Q| = %.2e, greedy actions equal for %.4f %%; HIP f32 kernels %.2f ms (%.1f TFLOP/s = %.2f of the 157.3 "
          "TFLOP/s f32 matrix peak), PyTorch f32 %.2f ms" % (total, worst, 100.0 * agree / total, t_hip * 1e3, flop / t_hip / 1e12, flop / t_hip / 157.3e12, t_torch * 1e3))
    assert worst <= 1e-4, worst
    assert unexplained == 0 and agree >= 0.999 * total, (agree, total, unexplained)
    assert t_hip < t_torch
    env.close()


@pytest.mark.gpu
def test_hip_policy_follows_parameter_updates():
    """DeepQNetwork repacks the kernel's weights after training: infer_action through the HIP path tracks the torch network"""
    import torch
    import magent_amd
    from magent_amd.builtin.torch_model import DeepQNetwork
    env = magent_amd.GridWorld("battle", map_size=40, device_obs=True)
    env.set_seed(3); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=200)
    m = DeepQNetwork(env, hs[0], "m", memory_size=64, infer_dtype="bf16")
    assert m._hip is not None
    obs = env.get_observation(hs[0]); env.sync()
    a1 = m.infer_action(obs, None, policy="greedy")
    with torch.no_grad():
        ref1 = _reference(m.qnet, obs[0], obs[1]).argmax(dim=1)
    with torch.no_grad():
        for p in m.qnet.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    m._hip.dirty = True
    a2 = m.infer_action(obs, None, policy="greedy")
    with torch.no_grad():
        ref2 = _reference(m.qnet, obs[0], obs[1]).argmax(dim=1)
    assert (a1.long() == ref1).float().mean().item() > 0.97 and (a2.long() == ref2).float().mean().item() > 0.97
    assert not torch.equal(a1, a2)
    env.close()


def test_weight_packing_is_the_documented_permutation():
    """CPU-only: include/magent_policy.h's "fragment order" and "slot order", checked by undoing them -- lane l of k-step s and tile
    T holds W[32 T + (l & 31)][16 s + 8 (l >> 5) + 0..7], and slot s of a 32-wide tile stands for channel
    (s & 3) + 8 ((s & 15) >> 2) + 4 (s >> 4)"""
    import torch
    from magent_amd.builtin.torch_model.dqn import _QNet
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicy, fragment_order, slot_channels
    torch.manual_seed(7)
    ch = slot_channels("cpu")
    assert sorted(ch.tolist()) == list(range(32))
    assert ch[:8].tolist() == [0, 1, 2, 3, 8, 9, 10, 11] and ch[16:20].tolist() == [4, 5, 6, 7]      # lane group 0 / 1 of the MFMA result
    w = torch.randn(64, 48)
    f = fragment_order(w).float()
    assert f.shape == (3, 2, 64, 8)
    for (s, T, l, e) in [(0, 0, 0, 0), (2, 1, 63, 7), (1, 0, 37, 3), (2, 1, 5, 6)]:
        assert f[s, T, l, e] == w[32 * T + (l & 31), 16 * s + 8 * (l >> 5) + e].to(torch.bfloat16).float()
    view_space, feat, n_action = (13, 13, 7), 34, 21
    qnet = _QNet(view_space, (feat,), n_action, True, True)
    pol = HipDqnPolicy(qnet, view_space, (feat,), n_action, "cpu")
    pol.pack()
    t = pol._packed
    assert t["conv1"].shape == (5, 1, 64, 8) and t["conv2"].shape == (18, 1, 64, 8) and t["dense_view"].shape == (162, 8, 64, 8)
    assert t["dense_emb"].shape == (3, 8, 64, 8) and t["head"].shape == (32, 1, 64, 8)
    bf = lambda x: x.detach().to(torch.bfloat16).float()
    # conv1: k = j * 8 + channel, j the tap's place in the order 0 3 1 4 2 5 6 7 8 pad; (tap 0, channel 7) carries the bias; the padding
    # tap and channels >= C are zero
    from magent_amd.builtin.torch_model.hip_policy import CONV1_TAP_ORDER
    c1 = t["conv1"].float()
    get1 = lambda co, tap, c: c1[(CONV1_TAP_ORDER.index(tap) * 8 + c) // 16, 0, 32 * (((CONV1_TAP_ORDER.index(tap) * 8 + c) % 16) // 8) + co, c]
    assert get1(5, 2, 3) == bf(qnet.conv1.weight)[5, 3, 0, 2] and get1(5, 7, 1) == bf(qnet.conv1.weight)[5, 1, 2, 1]
    assert get1(9, 0, 7) == bf(qnet.conv1.bias)[9] and get1(9, 9, 1) == 0 and get1(9, 4, 7) == 0
    # conv2: k = tap * 32 + slot, slot -> input channel ch[slot]
    c2 = t["conv2"].float()
    get2 = lambda co, k: c2[k // 16, 0, 32 * ((k % 16) // 8) + co, k % 8]
    assert get2(11, 4 * 32 + 17) == bf(qnet.conv2.weight)[11, int(ch[17]), 1, 1]
    # dense_view: k = position * 32 + slot; output tile T row i is output 32 T + i; its bias sits in slot order
    dv = t["dense_view"].float()
    getv = lambda o, k: dv[k // 16, o // 32, 32 * ((k % 16) // 8) + o % 32, k % 8]
    assert getv(200, 40 * 32 + 9) == bf(qnet.dense_view.weight)[200, 40 * 32 + int(ch[9])]
    assert t["dense_view_bias"][3 * 32 + 21] == qnet.dense_view.bias[3 * 32 + int(ch[21])]
    # head: k = hidden slot 32 T' + slot (T' < 8: dense_view, then dense_emb); output n_action is the value
    hd = t["head"].float()
    geth = lambda o, k: hd[k // 16, 0, 32 * ((k % 16) // 8) + o, k % 8]
    assert geth(4, 5 * 32 + 30) == bf(qnet.advantage.weight)[4, 5 * 32 + int(ch[30])]
    assert geth(n_action, 256 + 2 * 32 + 1) == bf(qnet.value.weight)[0, 256 + 2 * 32 + int(ch[1])] and geth(n_action + 1, 77) == 0


@pytest.mark.gpu
def test_bf16_cell_observation_and_policy_on_it():
    """env_get_observation_device_bf16: every window cell as 8 bf16 = the float32 observation's channels rounded to nearest even,
    zeros, and 1.0 in channel 7 -- bit for bit; the policy kernels fed with those cells give the very Q values they give on the
    float32 views (the same bf16 operands reach the same MFMAs)"""
    import torch
    import magent_amd
    from magent_amd.builtin.torch_model.dqn import _QNet
    from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicy
    for game, size, n in (("battle", 60, 700), ("gather", 60, 300)):
        env = magent_amd.GridWorld(game, map_size=size)
        env.set_seed(11); env.reset()
        hs = env.get_handles()
        if game == "battle":
            for h in hs:
                env.add_agents(h, "random", n=n)
        else:
            env.add_agents(hs[0], "random", n=400); env.add_agents(hs[1], "random", n=n)
        h = hs[-1]
        for step in range(3):
            view32, feat = env.get_observation_device(h)
            view16, feat16 = env.get_observation_device_bf16(h)
            env.sync()
            c = view32.shape[-1]
            if c > 7:
                break
            assert torch.equal(view16[..., :c].view(torch.int16), view32.to(torch.bfloat16).view(torch.int16))
            assert torch.equal(view16[..., c:7].float(), torch.zeros_like(view16[..., c:7].float())) and bool((view16[..., 7].float() == 1).all())
            assert torch.equal(feat16, feat)
            if step == 0:
                torch.manual_seed(3)
                vs, fs, na = env.get_view_space(h), env.get_feature_space(h), env.get_action_space(h)[0]
                qnet = _QNet(vs, fs, na, True, True).to(view32.device)
                try:
                    pol = HipDqnPolicy(qnet, vs, fs, na, view32.device, chunk=256)
                except ValueError:          # (gather: 33 actions, more than the head's 32-wide output tile holds: PyTorch takes it)
                    assert game == "gather"
                    continue
                a32, q32 = pol.infer(view32, feat, want_q=True)
                a16, q16 = pol.infer(view16, feat, want_q=True)
                torch.cuda.synchronize()
                assert torch.equal(q32, q16) and torch.equal(a32, a16)
            for hh in hs[-1:] if game == "gather" else hs:
                env.set_action(hh, np.random.RandomState(step).randint(env.get_action_space(hh)[0], size=env.get_num(hh)).astype(np.int32))
            env.step(); env.clear_dead()
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["battle_turn", "arrange", "tri_rect", "battle_food", "sector_turn", "pursuit"])
def test_bf16_cells_on_other_games(name):
    """the bf16-cell render in its other instantiations -- turn_mode windows, unpacked view cells (goals), three groups, food
    channel, multi-cell bodies: cells == the float32 observation rounded to nearest even, wherever the game has <= 7 channels"""
    import torch
    import helpers as H
    sc = H.scenarios()[name]
    env, hs = sc.build(H.HIP_LIB)
    rs = np.random.RandomState(1)
    checked = 0
    channels = [env.get_view_space(h)[2] for h in hs]
    for step in range(3):
        for h in hs:
            if env.get_num(h) == 0 or env.get_view_space(h)[2] > 7:
                continue
            view32, feat = env.get_observation_device(h)
            view16, feat16 = env.get_observation_device_bf16(h)
            env.sync()
            c = view32.shape[-1]
            assert torch.equal(view16[..., :c].view(torch.int16), view32.to(torch.bfloat16).view(torch.int16)), (name, step)
            assert bool((view16[..., c:7].float() == 0).all()) and bool((view16[..., 7].float() == 1).all()) and torch.equal(feat16, feat)
            checked += 1
        acting = sc.acting if sc.acting is not None else list(range(len(hs)))
        for g, h in enumerate(hs):
            if g in acting:
                env.set_action(h, rs.randint(env.get_action_space(h)[0], size=env.get_num(h)).astype(np.int32))
        env.step(); env.clear_dead()
    env.close()
    assert checked > 0 or all(c > 7 for c in channels), name      # (battle_food: 8 channels, refused by the bf16-cell format)
