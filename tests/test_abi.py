"""The C-ABI library loads and exports every symbol include/fwgym.h declares; struct sizes agree between the header
(as compiled into the library) and the ctypes mirror; host-only entry points work without a GPU."""
import copy
import ctypes
import os
import re

import pytest

from gym_fixed_wing import _native as nat
from gym_fixed_wing.config import EnvConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "fwgym.h")) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(fwg_[a-z_]+)\s*\(", text)) - {"fwg_config", "fwg_layout"})


def test_library_exports_every_declared_symbol():
    if not os.path.exists(nat.DEFAULT_LIB):
        pytest.skip("libfwgym.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(nat.DEFAULT_LIB)
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(nat.EXPORTS) == set(names)


def test_layout_and_version_without_gpu():
    if not os.path.exists(nat.DEFAULT_LIB):
        pytest.skip("libfwgym.so not built")
    lib = nat.load_library()
    assert lib.fwg_abi_version() == nat.FWG_ABI_VERSION
    c = EnvConfig().compile()
    lay = nat.Layout()
    nat.check(lib, lib.fwg_get_layout(ctypes.byref(c), ctypes.byref(lay)))
    assert lay.sim == 0 and lay.rows > 100 and lay.rows % 4 == 0 and lay.window == 5 and lay.lag_depth == 0
    # a struct of the wrong size is refused, not misread
    c.struct_bytes -= 8
    assert lib.fwg_get_layout(ctypes.byref(c), ctypes.byref(lay)) == -2
    assert b"mismatch" in lib.fwg_last_error()
    # the build froze the preset configurations
    assert lib.fwg_num_specs() >= 3
    buf = (ctypes.c_uint32 * 4096)()
    c = EnvConfig().compile()
    assert lib.fwg_dump_spec(ctypes.byref(c), buf, 4096) > 500


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(nat.NativeError):
        nat.load_library(str(tmp_path / "nope.so"))


def test_unsupported_configurations_raise():
    from gym_fixed_wing import presets
    cfg = presets.default()
    cfg["integration_window"] = 10   # (built since round 3: lowers -- to 0 while nothing reads the windowed sums)
    assert EnvConfig(cfg).compile().integration_window == 0
    cfg["observation"]["states"].append({"type": "target", "name": "roll", "value": "integrator"})
    assert EnvConfig(cfg).compile().integration_window == 10
    cfg["integration_window"] = 50
    with pytest.raises(NotImplementedError):
        EnvConfig(cfg)
    cfg = presets.default()
    cfg["target"]["states"][0]["class"] = "attitude_angular"
    with pytest.raises(NotImplementedError):
        EnvConfig(cfg)
    cfg = presets.default()
    cfg["reward"]["factors"][0]["function_class"] = "quadratic"   # no matching term: the reference raises KeyError too
    with pytest.raises(KeyError):
        EnvConfig(cfg).compile()


def test_every_configuration_file_the_reference_ships_is_a_build_time_preset():
    """gym_fixed_wing/fixed_wing_config.json, fixed_wing_config_dev.json, examples/fixed_wing_config.json and the two
    examples/models/*_controller configurations, constructed the way FixedWingVecEnv does by default (derived views on, row log
    where it applies): each lowers to the words of a frozen preset, i.e. runs a constexpr-specialised kernel without hipcc on the
    machine (the interpreting kernel is ~30x slower, bench.py `generic_kernel`)."""
    from gym_fixed_wing import presets, specialize
    from gym_fixed_wing.config import EnvConfig
    lib = nat.load_library()
    frozen = []
    for name, kind, ckw, skw in presets.SPECIALISED:
        ec = EnvConfig(presets.preset(kind), config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw))
        frozen.append(specialize.spec_words(lib, ec, store_derived="_lean" not in name,
                                            obs_log_rows=presets.OBS_LOG_ROWS if name.endswith("_log") else 0))
    for kind in ("default", "examples", "mlp", "cnn", "dev"):
        ec = EnvConfig(presets.preset(kind))
        ob = ec.cfg["observation"]
        noise = ob.get("noise", None)
        noisy = noise is not None and (noise.get("var", 0) != 0 or noise.get("mean", 0) != 0)
        rows = presets.OBS_LOG_ROWS if (int(ob.get("length", 1)) > 1 and not noisy and len(ob["states"]) % 4 == 0) else 0
        words = specialize.spec_words(lib, ec, store_derived=True, obs_log_rows=rows)
        assert words in frozen, kind
