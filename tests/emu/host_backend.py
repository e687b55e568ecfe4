"""TEST-ONLY memory backend: numpy host arrays standing in for device memory when FixedWingVecEnv is driven against
tests/emu/libfwgym_emu.so (the host emulation build of the HIP kernels).  Not part of the product."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EMU_LIB = os.path.join(HERE, "libfwgym_emu.so")
SRC = os.path.join(ROOT, "fixed-wing-gym_amd", "csrc")


def build_emu_omp(force=False):
    """Bench-only CPU baseline (bench.py cpu_baseline "native"): the same emulation build with -O2 and the workgroups
    spread over OpenMP threads."""
    return build_emu(force, out=os.path.join(HERE, "libfwgym_emu_omp.so"), extra=["-O2", "-fopenmp", "-DFWG_EMU_OMP"])


def build_emu(force=False, out=None, extra=None):
    srcs = [os.path.join(SRC, f) for f in os.listdir(SRC) if os.path.isfile(os.path.join(SRC, f))] + [os.path.join(HERE, "hip", "hip_runtime.h"),
                                                               os.path.join(ROOT, "include", "fwgym.h")]
    inc = os.path.join(SRC, "generated", "specs.inc")
    srcs = srcs + ([inc] if os.path.exists(inc) else [])
    out = out or EMU_LIB
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    cmd = ["g++", "-x", "c++", "-std=c++17", "-O1", "-shared", "-fPIC", "-pthread", "-w", "-I" + HERE] + (extra or []) + \
          (["-DFWG_WITH_SPECS"] if os.path.exists(inc) else []) + [
           "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "-o", out + ".tmp{}".format(os.getpid()),
           os.path.join(SRC, "fwgym.hip")]
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp{}".format(os.getpid()), out)   # atomic: parallel test workers may build at once
    return out


class HostBackend(object):
    index = 0
    _dt = {"f32": np.float32, "u8": np.uint8, "i32": np.int32}

    def zeros(self, shape, kind="f32"):
        return np.zeros(shape, dtype=self._dt[kind])

    def full(self, shape, value, kind="f32"):
        return np.full(shape, value, dtype=self._dt[kind])

    def ptr(self, t):
        assert t.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(t.ctypes.data)

    def stream(self):
        return ctypes.c_void_p()

    def sync(self):
        pass

    def to_host(self, t):
        return np.array(t)

    def from_host(self, a, kind="f32"):
        return np.ascontiguousarray(a, dtype=self._dt[kind])

    def as_device(self, x, kind="f32"):
        return np.ascontiguousarray(np.asarray(x), dtype=self._dt[kind])

    def view_i32(self, t):
        return t.view(np.int32)
