"""Builds tests/hipemu/_build/libmagent_emu.so: the engine's HIP sources compiled as plain C++ against the hipemu shim.

TEST INFRASTRUCTURE ONLY -- see tests/hipemu/hip/hip_runtime.h.  The sources are used as they are, except for one textual
change made on copies (sources and headers): `extern __shared__ T name[];` (dynamic LDS) becomes a pointer to the emulator's LDS block."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "magent_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libmagent_emu.so")
SOURCES = ["render.hip", "step.hip", "pipe.hip", "cycle.hip", "engine.hip", "engine_rules.hip", "engine_observe.hip", "engine_step.hip", "engine_batch.hip", "runtime_api.hip", "policy.hip", "policy_f32.hip"]     # (MFMAs: hipemu::mfma_32x32x16_bf16)
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-g1", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing", "-w",
         "-I", HERE, "-I", OUT, "-I", CSRC, "-I", os.path.join(ROOT, "include")]
DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?((?:unsigned\s+)?\w+)\s+(\w+)\[\];")


def build(force=False):
    # one builder at a time (pytest-xdist workers, the tests' subprocesses): whoever waited finds the library fresh and returns
    import fcntl
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force)


def _build(force):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "emu_runtime.cc"), os.path.join(HERE, "hip", "hip_runtime.h"),
                                                                 os.path.abspath(__file__)]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs = []
    sub = lambda text: DYN.sub(lambda m: "%s *%s = (%s *)hipemu::dynamic_lds();" % (m.group(1), m.group(2), m.group(1)), text)
    for h in sorted(os.listdir(CSRC)):          # (the headers get the same textual change, on copies that are found first)
        if h.endswith(".h"):
            open(os.path.join(OUT, h), "w").write('#line 1 "%s"\n' % os.path.join(CSRC, h) + sub(open(os.path.join(CSRC, h)).read()))
    for src in SOURCES:
        text = open(os.path.join(CSRC, src)).read()
        text = sub(text)
        cc = os.path.join(OUT, src.replace(".hip", "_emu.cc"))
        open(cc, "w").write('#line 1 "%s"\n' % os.path.join(CSRC, src) + text)
        obj = cc.replace(".cc", ".o")
        subprocess.check_call([CXX] + FLAGS + ["-c", cc, "-o", obj])
        objs.append(obj)
    obj = os.path.join(OUT, "emu_runtime.o")
    subprocess.check_call([CXX] + FLAGS + ["-c", os.path.join(HERE, "emu_runtime.cc"), "-o", obj])
    objs.append(obj)
    # (linked beside the target and renamed over it: a test process that has the old library mapped keeps its own copy)
    tmp = LIB + ".%d.tmp" % os.getpid()
    subprocess.check_call([CXX, "-shared", "-fPIC", "-o", tmp] + objs + ["-Wl,-Bsymbolic", "-lpthread"])
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
