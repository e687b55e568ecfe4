#!/usr/bin/env python3
"""Developer tool: a quick-to-build libfwgym with ONE frozen configuration (a preset of gym_fixed_wing/presets.py) and only
its two-wave / fused / reset kernels (-DFWG_DEV_FAST_BUILD: no generic kernels, no one-wave k_step): about a minute of hipcc
instead of eight for the product library.  For A/B measurements on the GPU box:

    python tools/devlib.py c5_examples_lean _abl/libfwgym_c5dev.so [-DFLAG ...]
    gpurun -- python bench.py --workload c5 --lib fixed-wing-gym_amd/gym_fixed_wing/_abl/libfwgym_c5dev.so ...

Never the product: FixedWingVecEnv(_lib_path=...) only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa  # noqa: E402


def main():
    preset, out = sys.argv[1], sys.argv[2]
    flags = sys.argv[3:]
    if not os.path.isabs(out):
        out = os.path.join(ROOT, "fixed-wing-gym_amd", "gym_fixed_wing", out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    inc = out + ".specs.inc"
    isa.spec_file(preset, inc)
    csrc = os.environ.get("DEVLIB_CSRC", os.path.join(ROOT, "fixed-wing-gym_amd", "csrc"))   # (another tree: an earlier commit's csrc/)
    inc_dir = os.path.join(os.path.dirname(os.path.dirname(csrc)), "include") if "DEVLIB_CSRC" in os.environ else os.path.join(ROOT, "include")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=fast",
           "-fno-slp-vectorize", "-I" + inc_dir, "-I" + csrc, '-DFWG_SPECS_FILE="{}"'.format(inc),
           "-DFWG_DEV_FAST_BUILD", "-o", out, os.path.join(csrc, "fwgym.hip")] + flags
    subprocess.run(cmd, check=True)
    os.remove(inc)
    print("built", out)


if __name__ == "__main__":
    main()
