"""The callers of the hot path: model hosting, episode buffer, the PyTorch DQN that stands where
the reference keeps its TensorFlow one, and the reference's own examples/train_battle.py running UNMODIFIED on the
`magent` import name.  CPU tests drive the CPU oracle as the engine (no GPU here); the gpu test drives the HIP engine."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

ROOT = H.ROOT


def test_piecewise_decay_and_friends():
    from magent_amd import utility as U
    assert U.piecewise_decay(0, [0, 700, 1400], [1, 0.2, 0.05]) == 1
    assert abs(U.piecewise_decay(350, [0, 700, 1400], [1, 0.2, 0.05]) - 0.6) < 1e-12
    assert U.piecewise_decay(5000, [0, 700, 1400], [1, 0.2, 0.05]) == 0.05
    assert U.linear_decay(10, 100, 0.1) == pytest.approx(0.91)
    assert U.rec_round([1.234, [2.345, 3.456]]) == [1.23, [2.35, 3.46]] or U.rec_round([1.234, [2.345, 3.456]])[0] == 1.23


def test_episodes_buffer_tracks_a_capped_random_subset():
    from magent_amd.utility import EpisodesBuffer
    np.random.seed(0)
    buf = EpisodesBuffer(capacity=5)
    ids = np.arange(100, 120, dtype=np.int32)
    views, feats = np.random.rand(20, 3, 3, 2).astype(np.float32), np.random.rand(20, 4).astype(np.float32)
    alive = np.ones(20, dtype=bool); alive[3] = False
    for t in range(3):
        buf.record_step(ids, (views + t, feats + t), np.arange(20), np.full(20, float(t)), alive)
    eps = list(buf.episodes())
    assert len(eps) == 5 and buf.is_full
    for key, e in buf.buffer.items():
        i = key - 100
        assert len(e.rewards) == 3 and e.rewards == [0.0, 1.0, 2.0] and e.actions == [i, i, i]
        assert np.array_equal(e.views[2], views[i] + 2) and e.terminal == (i == 3)
    new_ids = np.arange(200, 210, dtype=np.int32)     # a full buffer admits nobody new
    buf.record_step(new_ids, (views[:10], feats[:10]), np.zeros(10), np.zeros(10), np.ones(10, bool))
    assert len(buf.buffer) == 5


def test_episodes_buffer_packed_is_the_loop_over_episodes():
    """packed() -- what DeepQNetwork puts into its replay memory in one go -- is exactly what the reference's loop over episodes()
    appends episode by episode (views, features, actions, rewards; terminal on the last transition of an agent that died, mask 0 on
    the last transition of one that did not), with agents dying, leaving and new ones showing up, numpy and torch observations"""
    import torch
    from magent_amd.utility import EpisodesBuffer
    for as_torch in (False, True):
        rs = np.random.RandomState(5)
        np.random.seed(11)
        buf = EpisodesBuffer(capacity=13)
        alive_ids = list(range(50, 70))
        next_id = 70
        for t in range(9):
            ids = np.array(alive_ids, dtype=np.int32)
            n = len(ids)
            views, feats = rs.rand(n, 3, 3, 2).astype(np.float32), rs.rand(n, 4).astype(np.float32)
            acts, rewards = rs.randint(5, size=n).astype(np.int32), rs.rand(n).astype(np.float32)
            alives = rs.rand(n) > 0.15
            obs = (torch.from_numpy(views), torch.from_numpy(feats)) if as_torch else (views, feats)
            buf.record_step(ids, obs, torch.from_numpy(acts) if as_torch else acts, rewards, alives)
            alive_ids = [i for i, a in zip(alive_ids, alives) if a] + [next_id, next_id + 1]      # the dead are cleared, two are born
            next_id += 2
        assert buf.is_full and len(buf.buffer) == 13
        want = {k: [] for k in ("views", "features", "actions", "rewards", "terminal", "mask")}
        for ep in buf.episodes():                      # the loop of tf_model/dqn.py:233-262 / torch_model/dqn.py
            m = len(ep.rewards)
            assert m > 0
            mask, terminal = np.ones(m, np.float32), np.zeros(m, bool)
            if ep.terminal:
                terminal[-1] = True
            else:
                mask[-1] = 0
            want["views"] += [np.asarray(v) for v in ep.views]; want["features"] += [np.asarray(f) for f in ep.features]
            want["actions"] += ep.actions; want["rewards"] += ep.rewards
            want["terminal"] += list(terminal); want["mask"] += list(mask)
        views, feats, acts, rewards, terminal, mask = buf.packed()
        assert isinstance(views, torch.Tensor) == as_torch
        assert np.array_equal(np.asarray(views), np.stack(want["views"])) and np.array_equal(np.asarray(feats), np.stack(want["features"]))
        assert np.array_equal(np.asarray(acts), np.array(want["actions"])) and np.array_equal(rewards, np.array(want["rewards"], np.float32))
        assert np.array_equal(terminal, np.array(want["terminal"])) and np.array_equal(mask, np.array(want["mask"], np.float32))
        assert terminal.any() and (mask == 0).any()
    assert EpisodesBuffer(3).packed() is None


def test_episodes_buffer_row_lookup_equals_the_search():
    """record_step finds the tracked agents' rows by looking the tracked ids up in the step's (ascending) ids -- k log n instead of a pass
    over every id of a million-agent step -- and must record exactly what the general search records: agents dying and being cleared,
    new ones appended, a step whose ids are NOT ascending (the look-up must notice and search), a group that shrinks to nothing"""
    from magent_amd.utility import EpisodesBuffer
    for seed in range(6):
        rs = np.random.RandomState(100 + seed)
        np.random.seed(seed)
        fast, plain = EpisodesBuffer(capacity=40), EpisodesBuffer(capacity=40)
        ids = np.sort(rs.choice(5000, size=600, replace=False)).astype(np.int32)
        next_id = 5000
        for t in range(25):
            n = len(ids)
            order = rs.permutation(n) if (t % 9 == 5 and n > 1) else np.arange(n)       # (now and then: ids in no order at all)
            step_ids = ids[order]
            views, feats = rs.rand(n, 2, 2, 1).astype(np.float32), rs.rand(n, 3).astype(np.float32)
            acts, rewards = rs.randint(5, size=n).astype(np.int32), rs.rand(n).astype(np.float32)
            alives = rs.rand(n) > (0.9 if t == 20 else 0.1)
            state = np.random.get_state()
            fast.record_step(step_ids, (views, feats), acts, rewards, alives)
            np.random.set_state(state)
            plain._sorted_n = None; plain._expect_alive = 1 << 62                       # (never takes the look-up)
            plain.record_step(step_ids, (views, feats), acts, rewards, alives)
            ids = ids[np.sort(order[alives])] if False else step_ids[alives][np.argsort(step_ids[alives])]     # clear_dead keeps the order; the next step is ascending again
            if t % 4 == 1:
                ids = np.concatenate([ids, np.arange(next_id, next_id + 30, dtype=np.int32)]); next_id += 30
        a, b = fast.packed(), plain.packed()
        assert (a is None) == (b is None)
        if a is not None:
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x), np.asarray(y))
        assert fast._slot == plain._slot


def _np_qnet(params, view, feature, use_dueling=True):
    """NumPy fp32 restatement of the reference network (tf_model/dqn.py:151-189): conv3x3(32) -> conv3x3(32), both VALID, NHWC,
    relu -> flatten (h, w, c order) -> dense 256 relu || dense 256 relu on the features -> concat -> dueling head
    (value + advantage(no bias) - mean advantage).  Weights in TensorFlow layout: conv kernels HWIO, dense kernels [in, out]."""
    def conv_valid(x, k, b):                       # x [N,H,W,C], k [3,3,C,O]
        n, h, w, c = x.shape
        out = np.zeros((n, h - 2, w - 2, k.shape[3]), dtype=np.float32)
        for dy in range(3):
            for dx in range(3):
                out += np.tensordot(x[:, dy:dy + h - 2, dx:dx + w - 2, :], k[dy, dx], axes=([3], [0])).astype(np.float32)
        return np.maximum(out + b, 0).astype(np.float32)
    h1 = conv_valid(view, params["conv1/kernel"], params["conv1/bias"])
    h2 = conv_valid(h1, params["conv2/kernel"], params["conv2/bias"])
    flat = h2.reshape(h2.shape[0], -1)
    h_view = np.maximum(flat @ params["dense_view/kernel"] + params["dense_view/bias"], 0)
    h_emb = np.maximum(feature @ params["dense_emb/kernel"] + params["dense_emb/bias"], 0)
    dense = np.concatenate([h_view, h_emb], axis=1).astype(np.float32)
    if not use_dueling:
        return dense @ params["value/kernel"] + params["value/bias"]
    value = dense @ params["value/kernel"] + params["value/bias"]
    adv = dense @ params["advantage/kernel"]
    return value + adv - adv.mean(axis=1, keepdims=True)


def test_dqn_network_is_the_reference_network():
    """the PyTorch Q-network against a NumPy restatement of tf_model/dqn.py:151-189: parameter shapes (in TensorFlow's layout)
    and a forward pass on battle-shaped inputs, rtol 1e-4 (fp32 accumulation order differs)"""
    import torch
    from magent_amd.builtin.torch_model.dqn import _QNet
    torch.manual_seed(3)
    view_space, feature_space, n_action = (13, 13, 7), (34,), 21
    net = _QNet(view_space, feature_space, n_action, use_dueling=True, use_conv=True)
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    tf_params = {   # torch conv weights are OIHW, TensorFlow's HWIO; torch Linear is [out, in], tf.layers.dense [in, out]
        "conv1/kernel": sd["conv1.weight"].transpose(2, 3, 1, 0), "conv1/bias": sd["conv1.bias"],
        "conv2/kernel": sd["conv2.weight"].transpose(2, 3, 1, 0), "conv2/bias": sd["conv2.bias"],
        "dense_view/kernel": sd["dense_view.weight"].T, "dense_view/bias": sd["dense_view.bias"],
        "dense_emb/kernel": sd["dense_emb.weight"].T, "dense_emb/bias": sd["dense_emb.bias"],
        "value/kernel": sd["value.weight"].T, "value/bias": sd["value.bias"],
        "advantage/kernel": sd["advantage.weight"].T,
    }
    shapes = {k: v.shape for k, v in tf_params.items()}
    assert shapes == {"conv1/kernel": (3, 3, 7, 32), "conv1/bias": (32,), "conv2/kernel": (3, 3, 32, 32), "conv2/bias": (32,),
                      "dense_view/kernel": (9 * 9 * 32, 256), "dense_view/bias": (256,), "dense_emb/kernel": (34, 256),
                      "dense_emb/bias": (256,), "value/kernel": (512, 1), "value/bias": (1,), "advantage/kernel": (512, 21)}
    assert "advantage.bias" not in sd                   # use_bias=False (dqn.py:182)
    rs = np.random.RandomState(0)
    view = rs.rand(17, *view_space).astype(np.float32)
    feature = rs.rand(17, *feature_space).astype(np.float32)
    with torch.no_grad():
        got = net(torch.from_numpy(view), torch.from_numpy(feature)).numpy()
    want = _np_qnet(tf_params, view, feature)
    assert got.shape == (17, 21) and np.allclose(got, want, rtol=1e-4, atol=1e-5)
    assert np.array_equal(got.argmax(1), want.argmax(1))
    # the plain head (use_dueling=False, dqn.py:186) and the fully connected trunk (use_conv=False, dqn.py:171-173) exist too
    plain = _QNet(view_space, feature_space, n_action, use_dueling=False, use_conv=False)
    assert plain.value.weight.shape == (21, 512) and plain.dense_view.weight.shape == (256, 13 * 13 * 7)


def _tiny_env(lib):
    import magent_amd
    env = H.gridworld("battle", lib=lib, map_size=20)
    env.reset()
    h0, h1 = env.get_handles()
    env.add_agents(h0, "random", n=20)
    env.add_agents(h1, "random", n=20)
    return env, (h0, h1)


def _play_round(env, handles, models, steps, eps=1.0):
    """the call sequence of examples/train_battle.py:61-109"""
    for _ in range(steps):
        obs, ids, acts = {}, {}, {}
        for i, h in enumerate(handles):
            obs[i] = env.get_observation(h)
            ids[i] = env.get_agent_id(h)
            models[i].infer_action(obs[i], ids[i], "e_greedy", eps, block=False)
        for i, h in enumerate(handles):
            acts[i] = models[i].fetch_action()
            env.set_action(h, acts[i])
        done = env.step()
        for i, h in enumerate(handles):
            models[i].sample_step(env.get_reward(h), env.get_alive(h), block=False)
        env.clear_dead()
        for m in models:
            m.check_done()
        if done:
            break
    for m in models:
        m.train(print_every=1000, block=False)
    return [m.fetch_train() for m in models]


def test_dqn_learns_something_on_cpu(tmp_path):
    import torch
    import magent_amd
    from magent_amd.builtin.torch_model import DeepQNetwork
    from magent_amd.model import ProcessingModel
    torch.manual_seed(0); np.random.seed(0)
    env, handles = _tiny_env(H.ensure_oracle())
    models = [ProcessingModel(env, h, "m%d" % i, 20000 + i, 100, DeepQNetwork, batch_size=32, memory_size=4096,
                              target_update=50, train_freq=2, device="cpu") for i, h in enumerate(handles)]
    before = [p.detach().clone() for p in models[0].model.qnet.parameters()]
    results = _play_round(env, handles, models, steps=25)
    for loss, value in results:
        assert np.isfinite(loss) and np.isfinite(value) and loss > 0
    after = list(models[0].model.qnet.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(after, before))
    # acting: greedy is deterministic, epsilon = 1 is uniform over the action space
    v, f = env.get_observation(handles[0])
    a1 = models[0].model.infer_action((v, f), None, policy="greedy")
    a2 = models[0].model.infer_action((v, f), None, policy="greedy")
    assert a1.dtype == np.int32 and np.array_equal(a1, a2) and a1.min() >= 0 and a1.max() < 21
    # checkpoints round-trip
    models[0].save(str(tmp_path), 3)
    fresh = DeepQNetwork(env, handles[0], "m0", memory_size=16, device="cpu")
    fresh.load(str(tmp_path), 3)
    assert np.array_equal(fresh.infer_action((v, f), None, policy="greedy"), a1)


def test_drqn_and_a2c_train_on_cpu(tmp_path):
    """the other two model families of the reference (tf_model/drqn.py, a2c.py) behind the same call protocol"""
    import torch
    from magent_amd.builtin.torch_model import AdvantageActorCritic, DeepRecurrentQNetwork
    from magent_amd.model import ProcessingModel
    torch.manual_seed(0); np.random.seed(0)
    env, handles = _tiny_env(H.ensure_oracle())
    models = [ProcessingModel(env, handles[0], "rq", 20010, 100, DeepRecurrentQNetwork, batch_size=4, unroll_step=4,
                              memory_size=200, target_update=5, train_freq=1, device="cpu"),
              ProcessingModel(env, handles[1], "ac", 20011, 100, AdvantageActorCritic, use_comm=True, device="cpu")]
    before = [[p.detach().clone() for p in net.parameters()] for net in (models[0].model.qnet, models[1].model.net)]
    (loss, value), (losses, state_value) = _play_round(env, handles, models, steps=20)
    assert np.isfinite(loss) and loss > 0 and np.isfinite(value)
    assert len(losses) == 3 and all(np.isfinite(x) for x in losses) and np.isfinite(state_value)
    for net, old in zip((models[0].model.qnet, models[1].model.net), before):
        assert any(not torch.equal(a, b) for a, b in zip(net.parameters(), old))
    # the recurrent state follows the agent ids: same observation, different history -> the state table tracks the callers
    v, f = env.get_observation(handles[0])
    ids = env.get_agent_id(handles[0])
    drqn = models[0].model
    a = drqn.infer_action((v, f), ids, policy="greedy")
    assert a.dtype == np.int32 and len(a) == len(ids) and set(drqn.agent_states) == set(int(i) for i in ids)
    drqn.infer_action((v[:3], f[:3]), ids[:3], policy="greedy")
    assert set(drqn.agent_states) == set(int(i) for i in ids[:3])
    # the policy head samples valid actions; checkpoints round-trip
    a2c = models[1].model
    v1, f1 = env.get_observation(handles[1])
    acts = a2c.infer_action((v1, f1), None)
    assert acts.dtype == np.int32 and acts.min() >= 0 and acts.max() < 21
    for m, cls, kw in ((drqn, DeepRecurrentQNetwork, dict(memory_size=4)), (a2c, AdvantageActorCritic, dict(use_comm=True))):
        m.save(str(tmp_path), 1)
        fresh = cls(env, m.handle, m.name, device="cpu", **kw)
        fresh.load(str(tmp_path), 1)
        for p, q in zip((fresh.qnet if cls is DeepRecurrentQNetwork else fresh.net).parameters(),
                        (m.qnet if cls is DeepRecurrentQNetwork else m.net).parameters()):
            assert torch.equal(p, q)


def test_magent_alias_exposes_the_reference_names():
    import magent
    from magent.builtin.tf_model import DeepQNetwork
    from magent.builtin.rule_model import RandomActor
    assert magent.GridWorld is magent.gridworld.GridWorld and magent.ProcessingModel is magent.model.ProcessingModel
    assert callable(magent.utility.init_logger) and callable(magent.utility.sample_observation)
    assert DeepQNetwork.__name__ == "DeepQNetwork" and RandomActor is not None
    cfg = magent.gridworld.Config()
    cfg.set({"map_width": 10, "map_height": 10})
    from magent.builtin.mx_model import DeepQNetwork as MX     # train_gather.py imports the MXNet name
    from magent.builtin.tf_model import AdvantageActorCritic, DeepRecurrentQNetwork   # train_trans.py / train_against.py
    assert MX is DeepQNetwork and DeepRecurrentQNetwork.__name__ == "DeepRecurrentQNetwork" and AdvantageActorCritic is not None


@pytest.mark.skipif(not os.path.isfile("/root/reference/examples/train_battle.py"), reason="reference examples not present")
def test_reference_train_battle_runs_unmodified(tmp_path):
    """examples/train_battle.py, byte for byte the reference's file, against `import magent` = this repository.
    Engine = CPU oracle here (no GPU in this container); on a GPU box the same command runs the HIP engine."""
    (tmp_path / "build").mkdir()
    # the script is run as it is; the launcher in front of it only names the engine library the way every test does
    # (the class attribute the test suite overrides -- the product has no environment switch for it)
    launcher = ("import sys, runpy\n"
                "import magent_amd.gridworld as gw\n"
                "gw.GridWorld._engine_path = %r\n"
                "sys.argv = ['train_battle.py', '--train', '--n_round', '1', '--map_size', '20']\n"
                "runpy.run_path('/root/reference/examples/train_battle.py', run_name='__main__')\n" % H.ensure_oracle())
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-c", launcher], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "===== train =====" in p.stdout and "round time" in p.stdout


@pytest.mark.gpu
def test_training_round_on_gpu():
    """same loop, HIP engine + DQN on the same GPU"""
    import torch
    from magent_amd.builtin.torch_model import DeepQNetwork
    from magent_amd.model import ProcessingModel
    assert torch.cuda.is_available()
    env, handles = _tiny_env(H.HIP_LIB)
    models = [ProcessingModel(env, h, "g%d" % i, 20000 + i, 100, DeepQNetwork, batch_size=32, memory_size=4096,
                              target_update=50, train_freq=2) for i, h in enumerate(handles)]
    assert models[0].model.device.type == "cuda"
    for loss, value in _play_round(env, handles, models, steps=25):
        assert np.isfinite(loss) and np.isfinite(value)
    # device-resident observations feed the network without touching the host
    view, feat = env.get_observation_device(handles[0])
    env.sync()
    acts = models[0].model.infer_action((view, feat), None, policy="greedy")
    assert isinstance(acts, torch.Tensor) and acts.dtype == torch.int32 and acts.is_cuda
    env.order_after_torch()              # `acts` is produced on torch's stream
    env.set_action_device(handles[0], acts)
    env.sync()


@pytest.mark.gpu
def test_device_resident_training_round_on_gpu():
    """device_obs=True: get_observation hands out torch tensors on the engine's GPU; policy, episode buffer and replay
    memory never touch the host -- through the very same call sequence"""
    import torch
    import magent_amd
    from magent_amd.builtin.torch_model import DeepQNetwork
    from magent_amd.model import ProcessingModel
    env = H.gridworld("battle", lib=H.HIP_LIB, map_size=24, device_obs=True)
    env.reset()
    handles = env.get_handles()
    for h in handles:
        env.add_agents(h, "random", n=40)
    v, f = env.get_observation(handles[0])
    assert isinstance(v, torch.Tensor) and v.is_cuda and tuple(v.shape) == (40, 13, 13, 7)
    ref = H.gridworld("battle", lib=H.HIP_LIB, map_size=24)       # host-buffer twin: same bits
    ref.reset()
    for h in ref.get_handles():
        ref.add_agents(h, "random", n=40)
    assert v.cpu().numpy().tobytes() == ref.get_observation(ref.get_handles()[0])[0].tobytes()
    models = [ProcessingModel(env, h, "d%d" % i, 20000 + i, 50, DeepQNetwork, batch_size=32, memory_size=2048,
                              target_update=50, train_freq=2) for i, h in enumerate(handles)]
    for loss, value in _play_round(env, handles, models, steps=20):
        assert np.isfinite(loss) and np.isfinite(value)
    ep = next(iter(models[0].sample_buffer.episodes()), None)
    assert models[0].model.mem_view.buf.is_cuda


@pytest.mark.gpu
@pytest.mark.parametrize("map_size,per_side", [(1000, 40000), (3536, 499849)])
def test_c5_train_round(map_size, per_side):
    """BASELINE config 5's loop (examples/train_battle.py:45-140, `play_a_round`) at `--map_size 1000` (2 x 40,000 agents from its
    generate_map) and at the size BASELINE.json names, `--map_size 3536` (2 x 499,849: "1M agents"): device-resident observations, both
    sides acting through the DQN (bench.py: train_round_extra -- the function the bench line's `extra.c5_train_round_*` come from):
    finite loss after one train() per model, and for the first three steps every engine output (views, features, rewards, alive
    flags, and the positions after clear_dead) equals the oracle's, which is fed the very actions the policy chose."""
    import torch
    import bench
    import magent_amd
    twin = {}

    def check(step, env, handles, obs, acts, rewards, alives):
        if step >= 1:          # (train_round_extra plays two untimed steps first, numbered -2 and -1: the first three steps of the episode)
            return
        if not twin:
            assert step == -2
            o = H.gridworld("battle", lib=H.ensure_oracle(), map_size=map_size)
            o.set_seed(12345); o.reset()
            for g, pos in bench.train_battle_formation(map_size):      # the start positions as train_round_extra lays them out
                o.add_agents(o.get_handles()[g], method="custom", pos=pos)
            twin["env"] = o
        o = twin["env"]
        for i, h in enumerate(o.get_handles()):
            v, f = o.get_observation(h)
            assert obs[i][0].cpu().numpy().tobytes() == v.tobytes(), "step %d view %d" % (step, i)
            assert obs[i][1].cpu().numpy().tobytes() == f.tobytes(), "step %d feature %d" % (step, i)
            o.set_action(h, acts[i].cpu().numpy().astype(np.int32))
        o.step()
        for i, h in enumerate(o.get_handles()):
            r = rewards[i].cpu().numpy() if isinstance(rewards[i], torch.Tensor) else rewards[i]
            assert np.asarray(r).tobytes() == o.get_reward(h).tobytes(), "step %d reward %d" % (step, i)
            assert np.array_equal(np.asarray(alives[i].cpu() if isinstance(alives[i], torch.Tensor) else alives[i]).astype(bool), o.get_alive(h)), "step %d alive %d" % (step, i)
            assert np.array_equal(env.get_pos(h), o.get_pos(h)), "step %d pos %d" % (step, i)
        o.clear_dead()

    out = bench.train_round_extra(torch, magent_amd, map_size=map_size, steps=8 if map_size <= 1000 else 4, on_step=check)
    assert out["agents"] == [per_side, per_side]
    assert all(np.isfinite(x) and x > 0 for x in out["loss"]) and all(np.isfinite(x) for x in out["value"])
    assert out["env_ms_per_step"] > 0 and out["infer_ms_per_step"] > 0 and out["train_ms_per_round"] > 0
    print("c5 train round:", out)
