#!/usr/bin/env python3
"""Development helper: regenerates csrc/generated/specs.inc from the HOST-emulation build of the library (fwg_dump_spec is
pure host code), so that the CPU test loop does not need the 7-minute hipcc build after a change of DevCfg / the layout.
__graft_entry__.build() does the same through the real library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from emu.host_backend import build_emu  # noqa: E402
from gym_fixed_wing import _native as nat, specialize  # noqa: E402

lib = nat.load_library(build_emu(force=True))
print(specialize.write_specs(lib))
build_emu(force=True)
