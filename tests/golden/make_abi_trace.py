"""The C-ABI transcript of an episode played by the reference's UNMODIFIED Python wrapper against the compiled reference.

    python tests/golden/make_abi_trace.py      (build container: needs /root/reference and oracle/_ref/libmagent_ref.so)

/root/reference/python is copied to a scratch directory next to a `build/libmagent.so` that is tests/abi_trace/trace_shim.c -- a
library that forwards every call to oracle/_ref/libmagent_ref.so and writes the call down: arguments as they cross the boundary,
input buffers, and the bytes the reference engine wrote back.  Output: tests/golden/abi_trace_battle.bin.gz, replayed call by call by
tests/test_abi_trace.py against the oracle and the emulated kernels (CPU) and against the HIP engine on the GPU box, where the
reference tree does not exist."""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "libmagent_ref.so")

EPISODE = r'''
import sys
sys.path.insert(0, %(py)r)
import numpy as np
import magent
env = magent.GridWorld("battle", map_size=40)
env.set_seed(2024)
env.reset()
handles = env.get_handles()
env.add_walls(method="random", n=30)
env.add_agents(handles[0], method="random", n=28)
env.add_agents(handles[1], method="custom", pos=[(5 + 2 * k, 20 + (k %% 3)) for k in range(14)])
env.add_agents(handles[1], method="random", n=12)
for h in handles:
    env.get_view_space(h); env.get_feature_space(h); env.get_action_space(h); env.get_view2attack(h)
rs = np.random.RandomState(7)
for step in range(9):
    for h in handles:
        env.get_observation(h)
        env.get_agent_id(h)
        env.set_action(h, rs.randint(env.get_action_space(h)[0], size=env.get_num(h)).astype(np.int32))
    done = env.step()
    for h in handles:
        env.get_reward(h); env.get_alive(h); env.get_pos(h); env.get_num(h)
    env.clear_dead()
    if step == 4:
        env.add_agents(handles[0], method="random", n=5)
del env
'''


def main():
    assert os.path.exists(REF), "build oracle/_ref first: make -C oracle ref"
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree("/root/reference/python", os.path.join(tmp, "python"))
        os.makedirs(os.path.join(tmp, "build"))
        subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-o", os.path.join(tmp, "build", "libmagent.so"),
                               os.path.join(ROOT, "tests", "abi_trace", "trace_shim.c"), "-ldl"])
        raw = os.path.join(tmp, "trace.bin")
        env = dict(os.environ, MAGENT_TRACE_TARGET=REF, MAGENT_TRACE_OUT=raw, OMP_NUM_THREADS="1")
        subprocess.check_call([sys.executable, "-c", EPISODE % {"py": os.path.join(tmp, "python")}], env=env)
        data = open(raw, "rb").read()
    out = os.path.join(HERE, "abi_trace_battle.bin.gz")
    with gzip.GzipFile(out, "wb", compresslevel=9, mtime=0) as f:
        f.write(data)
    print("%s: %d bytes of transcript, %d compressed" % (out, len(data), os.path.getsize(out)))


if __name__ == "__main__":
    main()
