"""Development probe (GPU box): does the render's rate depend on where the caller's view tensor starts?
One process, one pair of big allocations; the bench cycle is played with the view tensors at different byte offsets inside them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import magent_amd
n = 400000
dev = torch.device("cuda", 0)
row = 13 * 13 * 7
slack = 64 << 20
big = [torch.empty(n * row + slack // 4, device=dev) for _ in range(2)]
feats = [torch.empty((n, 34), device=dev) for _ in range(2)]
rew = [torch.empty(n, device=dev) for _ in range(2)]
gen = torch.Generator(device=dev); gen.manual_seed(0)
acts = [[torch.randint(21, (n,), dtype=torch.int32, device=dev, generator=gen) for g in range(2)] for _ in range(8)]
offsets = [0, 16, 4096, 65536, 1 << 20, (1 << 21), (1 << 21) + 65536, 3 << 20, 16 << 20, 0, 4096, 1 << 20, 0]
for off in offsets:
    env = magent_amd.GridWorld("battle", map_size=1000)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=n)
    views = [b[off // 4: off // 4 + n * row].view(n, 13, 13, 7) for b in big]
    def cyc(s):
        for g, h in enumerate(hs):
            env.get_observation_device(h, views[g], feats[g]); env.set_action_device(h, acts[s % 8][g])
        env.step()
        for g, h in enumerate(hs):
            env.get_reward_device(h, rew[g])
        env.clear_dead()
    for s in range(5): cyc(s)
    env.sync()
    env.profile_enable(2); env.profile_read("render")
    t0 = time.perf_counter()
    for s in range(5, 25): cyc(s)
    env.sync()
    dt = (time.perf_counter() - t0) / 20
    k, ms = env.profile_read("render")
    print("offset %9d: render avg %.4f ms over %d launches, cycle %.4f ms, ptr %% 2MiB = %d" % (off, ms / k, k, dt * 1e3, views[0].data_ptr() % (1 << 21)), flush=True)
    env.close()
