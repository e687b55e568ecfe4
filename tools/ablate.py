#!/usr/bin/env python3
"""Builds measurement-only variants of libfwgym (one -DFWG_ABL_* switch each) for timing what each part of k_step costs.
Run here (hipcc cross-compiles), then `gpurun -- python tools/ablate.py --time`.  The variants are NOT functional."""
import os, subprocess, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fixed-wing-gym_amd")
OUT = os.path.join(PKG, "gym_fixed_wing", "_abl")
PRESETS = {"base": [], "empty": ["-DFWG_ABL_EMPTY"], "no_sim": ["-DFWG_ABL_NO_SIM"], "no_lag": ["-DFWG_ABL_NO_LAG"],
           "no_obswrite": ["-DFWG_ABL_NO_OBSWRITE"], "no_simstore": ["-DFWG_ABL_NO_SIMSTORE"],
           "no_stores": ["-DFWG_ABL_NO_OBSWRITE", "-DFWG_ABL_NO_GYMSTORE", "-DFWG_ABL_NO_SIMSTORE"],
           "loads_only": ["-DFWG_ABL_NO_OBSWRITE", "-DFWG_ABL_NO_GYMSTORE", "-DFWG_ABL_NO_SIMSTORE", "-DFWG_ABL_NO_SIM"],
           "nothing": ["-DFWG_ABL_NO_OBSWRITE", "-DFWG_ABL_NO_GYMSTORE", "-DFWG_ABL_NO_SIMSTORE", "-DFWG_ABL_NO_SIM", "-DFWG_ABL_NO_LAG"]}
# default: the preset list; NAME="-DFLAG ..." arguments replace it
VARIANTS = dict(PRESETS) if not any("=" in a for a in sys.argv[1:]) else {}
VARIANTS.update({k: v.split() for k, v in (a.split("=", 1) for a in sys.argv[1:] if "=" in a)})

def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for name, flags in VARIANTS.items():
        out = os.path.join(OUT, "libfwgym_{}.so".format(name))
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-DFWG_WITH_SPECS"] + flags + \
              ["-o", out, os.path.join(PKG, "csrc", "fwgym.hip")]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) >= 4:
            for n, p in procs:
                o = p.communicate()[0]
                print(n, "rc", p.returncode, o[-400:] if p.returncode else "")
            procs = []
    for n, p in procs:
        o = p.communicate()[0]
        print(n, "rc", p.returncode, o[-400:] if p.returncode else "")

def timeit():
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    from gym_fixed_wing import presets
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    wl = os.environ.get("FWG_ABL_WORKLOAD", "c3")
    cfg, ckw, skw, n, desc = presets.workload(wl)
    n = int(os.environ.get("FWG_ABL_ENVS", n))
    res = {}
    for name in sorted(f[9:-3] for f in os.listdir(OUT) if f.endswith(".so")):
        vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, seed=1, derived_views=False,
                              obs_log_rows=int(os.environ.get("FWG_ABL_LOG_ROWS", "0")),
                              _lib_path=os.path.join(OUT, "libfwgym_{}.so".format(name)))
        assert vec.spec_index >= 0
        vec.reset()
        acts = [torch.rand((n, 3), device="cuda") * 2 - 1 for _ in range(16)]
        for t in range(100): vec.step_device(acts[t % 16])
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(400): vec.step_device(acts[t % 16])
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 400 * 1e3)
        res[name] = best
        # the same launches replayed from one hipGraph (no host launch cost)
        gbest = float("nan")
        if os.environ.get("FWG_ABL_GRAPH", "1") == "1" and name not in ("no_gymstore",):
            vec.set_graph_mode(True)
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for t in range(2): vec.step_device(acts[t])
            torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); vec.capture_begin()
            with torch.cuda.graph(g):
                for t in range(200): vec.step_device(acts[t % 16])
            vec.capture_end()
            g.replay(); vec.note_replayed_steps(200); torch.cuda.synchronize()
            gbest = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); vec.note_replayed_steps(200); torch.cuda.synchronize()
                gbest = min(gbest, e0.elapsed_time(e1) / 200 * 1e3)
            res[name + "_graph"] = gbest
        print("%-16s %7.2f us/step   graph replay %7.2f us/step" % (name, best, gbest), flush=True)
        vec.close()
    print(json.dumps(res))

if __name__ == "__main__":
    timeit() if "--time" in sys.argv else build()
