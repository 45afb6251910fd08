"""Development helper (GPU box): forward time of the DQN trunk on one inference batch, a few formulations."""
import sys, time
import torch, torch.nn as nn, torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda")
torch.manual_seed(0)
view = torch.rand(B, 13, 13, 7, device=dev)
feat = torch.rand(B, 34, device=dev)
c1, c2 = nn.Conv2d(7, 32, 3).to(dev), nn.Conv2d(32, 32, 3).to(dev)
dv, de = nn.Linear(32 * 81, 256).to(dev), nn.Linear(34, 256).to(dev)
val, adv = nn.Linear(512, 1).to(dev), nn.Linear(512, 21, bias=False).to(dev)

def head(x, f):
    h = torch.cat([F.relu(dv(x)), F.relu(de(f))], dim=1)
    a = adv(h)
    return val(h) + a - a.mean(dim=1, keepdim=True)

def ref(view, feat):
    x = view.permute(0, 3, 1, 2)
    x = F.relu(c2(F.relu(c1(x))))
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    return head(x, feat)

def gemm(view, feat, dt):
    # NHWC windows -> GEMM: conv1 [B*121, 63] x [63, 32]; conv2 [B*81, 288] x [288, 32]
    w1 = c1.weight.permute(2, 3, 1, 0).reshape(63, 32).to(dt); w2 = c2.weight.permute(2, 3, 1, 0).reshape(288, 32).to(dt)
    v = view.to(dt)
    p = v.unfold(1, 3, 1).unfold(2, 3, 1)                    # [B, 11, 11, 7, 3(ky), 3(kx)]
    p = p.permute(0, 1, 2, 4, 5, 3).reshape(-1, 63)          # (ky, kx, c)
    x = F.relu(p @ w1 + c1.bias.to(dt)).reshape(-1, 11, 11, 32)
    p = x.unfold(1, 3, 1).unfold(2, 3, 1).permute(0, 1, 2, 4, 5, 3).reshape(-1, 288)
    x = F.relu(p @ w2 + c2.bias.to(dt)).reshape(x.shape[0], -1)
    h = torch.cat([F.relu(x @ dv.weight.t().to(dt) + dv.bias.to(dt)), F.relu(feat.to(dt) @ de.weight.t().to(dt) + de.bias.to(dt))], dim=1)
    a = h @ adv.weight.t().to(dt)
    return (h @ val.weight.t().to(dt) + val.bias.to(dt) + a - a.mean(dim=1, keepdim=True)).float()

def timeit(name, fn, n=5):
    with torch.no_grad():
        out = fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): out = fn()
        torch.cuda.synchronize()
    print("%-34s %8.2f ms / %d agents" % (name, (time.perf_counter() - t0) / n * 1e3, B), flush=True)
    return out

with torch.no_grad():
    q0 = timeit("fp32 NCHW conv (current)", lambda: ref(view, feat))
    def cl():
        x = view.permute(0, 3, 1, 2)          # NHWC memory == channels_last strides already
        x = F.relu(c2c(F.relu(c1c(x))))
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
        return head(x, feat)
    c1c, c2c = c1.to(memory_format=torch.channels_last), c2.to(memory_format=torch.channels_last)
    q1 = timeit("fp32 channels_last conv", cl)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        q2 = timeit("bf16 autocast channels_last", cl)
    q3 = timeit("fp32 unfold + matmul", lambda: gemm(view, feat, torch.float32))
    q4 = timeit("bf16 unfold + matmul", lambda: gemm(view, feat, torch.bfloat16))
    for n_, q in (("channels_last", q1), ("bf16 autocast", q2), ("fp32 gemm", q3), ("bf16 gemm", q4)):
        print(n_, "max |dq| %.4g  argmax agreement %.4f" % (float((q - q0).abs().max()), float((q.argmax(1) == q0.argmax(1)).float().mean())))
