#!/usr/bin/env python3
"""Per-wave phase timeline of k_step from s_memtime stamps (measurement build, -DFWG_TIMELINE; never the product).

  python tools/timeline.py build            # here: cross-compiles gym_fixed_wing/_abl/libfwgym_timeline.so
  gpurun -- python tools/timeline.py run    # GPU box: C3 dense and row-log, prints the phase table + JSON

Stamps (csrc/fwgym.hip FWG_TL): 0 entry | 1 loads issued, before the integration | 2 integration done | 3 simulator rows
stored | 4 streamed windows landed (vmcnt 0) | 5 gym logic done | 6 bookkeeping rows stored | 7 observation built |
8 episode-end branch done | 9 outputs issued | 10 all stores acknowledged."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fixed-wing-gym_amd")
OUT = os.path.join(PKG, "gym_fixed_wing", "_abl")
LIB = os.path.join(OUT, os.environ.get("TL_LIB", "libfwgym_timeline.so"))
# (table of phase names kept for reference; the run prints the stamps of each wave relative to the block's start)
TLW = 32   # stamps per wave (csrc/fwgym_env.h FWG_TL_W)
NAMES = ["entry->loads issued", "integration (incl. wait for state)", "store sim rows", "wait streamed windows (vmcnt0)",
         "gym logic", "store gym rows", "observation build", "episode-end branch", "outputs issued", "stores acknowledged"]


def build(extra=()):
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=fast",
           "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-DFWG_WITH_SPECS",
           "-DFWG_TIMELINE"] + list(extra) + ["-o", LIB, os.path.join(PKG, "csrc", "fwgym.hip")]
    subprocess.run(cmd, check=True)


def run():
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import numpy as np
    import torch
    from gym_fixed_wing import presets
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    res = {}
    # TL_WL="c3:dense" etc.: the workloads and observation layouts to run (a one-preset tools/devlib.py build holds one of them)
    sel = [w.split(":") for w in os.environ.get("TL_WL", "c3:log,c5:dense").split(",")]
    for wl, log_rows in [(w, presets.OBS_LOG_ROWS if lay == "log" else 0) for w, lay in sel]:
        cfg, ckw, skw, n, desc = presets.workload(wl)
        n = 65536
        vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, seed=1, derived_views=False,
                              obs_log_rows=log_rows, _lib_path=LIB)
        assert vec.spec_index >= 0
        vec.reset()
        trace = torch.zeros((n // 64 * 2, TLW), dtype=torch.int64, device="cuda")
        vec._lib.fwg_debug_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        vec._lib.fwg_debug_set_trace(vec._handle, ctypes.c_void_p(trace.data_ptr()))
        acts = [torch.rand((n, 3), device="cuda") * 2 - 1 for _ in range(16)]
        stagger = int(os.environ.get("TL_STAGGER", "0"))
        if stagger:   # spread the episode ages (and, with constant actions, provoke failures: TL_CONST=1)
            const = os.environ.get("TL_CONST", "0") == "1"
            per = int(vec.cfg["steps_max"]) // stagger
            perm = np.random.RandomState(5).permutation(n) if os.environ.get("TL_PERM", "0") == "1" else np.arange(n)
            for k in range(stagger):
                vec.reset(indices=np.sort(perm[k::stagger]))
                for t in range(per):
                    vec.step_device(acts[0] if const else acts[t % 16], want_obs=False)
                vec.finish_episodes()   # (as a training loop does: an end then never finds its env's last record uncollected)
            if const:
                acts = [acts[0]] * 16
        for t in range(300):
            vec.step_device(acts[t % 16])
        vec.finish_episodes()
        torch.cuda.synchronize()
        rows = []
        for rep in range(20):
            trace.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            vec.step_device(acts[rep % 16])
            e1.record()
            torch.cuda.synchronize()
            t = trace.cpu().numpy().astype(np.float64).reshape(n // 64, 2, TLW)
            rows.append((t, e0.elapsed_time(e1) * 1e3))
        T = np.stack([r[0] for r in rows])                    # [rep][block][wave][16]
        T = np.where(T == 0, np.nan, T)
        t0 = np.nanmin(T[:, :, :, 0], axis=2, keepdims=True)[..., None]     # the block's first stamp
        rel = T - t0                                          # ticks since the block started
        key = "{}{}".format(wl, "_log" if log_rows else "")
        med = np.nanmedian(rel.reshape(-1, 2, TLW), axis=0)    # [wave][stamp]
        start_spread = float(np.nanmedian(np.nanmax(T[:, :, :, 0], axis=(1, 2)) - np.nanmin(T[:, :, :, 0], axis=(1, 2))))
        life = np.nanmedian(np.nanmax(rel, axis=(2, 3)))
        res[key] = {"stamp_ticks_median": [[None if np.isnan(x) else float(x) for x in w] for w in med],
                    "first_to_last_wave_start_ticks": start_spread, "block_lifetime_ticks_median": float(life),
                    "event_us_median": float(np.median([r[1] for r in rows]))}
        print("== {} ({}): event {:.2f} us; ticks: block lifetime {:.0f}, start spread {:.0f}".format(
            key, desc, res[key]["event_us_median"], life, start_spread))
        for w in range(2):
            if np.all(np.isnan(med[w])):
                continue
            print("   wave {}: ".format(w) + "  ".join("{}:{:.0f}".format(i, med[w][i]) for i in np.argsort(np.where(np.isnan(med[w]), 1e18, med[w])) if not np.isnan(med[w][i])))
        # the slowest block of the last launch: where did ITS time go?
        last = rel[-1]
        worst = int(np.nanargmax(np.nanmax(last, axis=(1, 2))))
        order = np.argsort(-np.nanmax(last, axis=(1, 2)))
        steps_now = vec.field("steps_count").cpu().numpy()
        # lifetimes by what the block hosted in the last launch: an ending lane, lanes in their first steps, a piece of the next
        # episode's draw (flags bits 4..6 below the ready mark), or nothing special
        life_b = np.nanmax(last, axis=(1, 2))
        done_b = vec._done.cpu().numpy().reshape(-1, 64).sum(axis=1) > 0
        flags = vec.field("flags").cpu().numpy()
        early_b = (steps_now.reshape(-1, 64) <= 8).sum(axis=1) > 0 if steps_now is not None else np.zeros_like(done_b)
        stage = (flags.reshape(-1, 64) >> 4) & 7
        draw_b = ((stage > 0) & (stage < 6)).sum(axis=1) > 0        # (a piece was computed in this or an earlier launch)
        classes = {"episode end": done_b, "early lanes (no end)": early_b & ~done_b, "draw piece (no end, not early)": draw_b & ~done_b & ~early_b,
                   "plain": ~done_b & ~early_b & ~draw_b}
        res[key]["lifetime_by_class"] = {}
        for name, m in classes.items():
            if m.sum():
                res[key]["lifetime_by_class"][name] = {"blocks": int(m.sum()), "median": float(np.median(life_b[m])), "p90": float(np.percentile(life_b[m], 90)),
                                                        "max": float(life_b[m].max())}
                print("   {:34s} blocks {:4d}  lifetime median {:.0f}  p90 {:.0f}  max {:.0f}".format(name, int(m.sum()), np.median(life_b[m]),
                                                                                                       np.percentile(life_b[m], 90), life_b[m].max()))
        # by block index (dispatch order): lifetime and the gym wave's stamp 1 (its first rows consumed) per eighth of the grid
        nb = last.shape[0]
        print("   by block index eighth: lifetime median / p90 | gym stamp 1 median / p90 | physics stamp 1 median")
        for k in range(8):
            sl = slice(k * nb // 8, (k + 1) * nb // 8)
            print("      {:4d}..{:4d}: {:6.0f} {:6.0f} | {:6.0f} {:6.0f} | {:6.0f}".format(sl.start, sl.stop - 1, np.nanmedian(life_b[sl]), np.nanpercentile(life_b[sl], 90),
                  np.nanmedian(last[sl, 1, 1]), np.nanpercentile(last[sl, 1, 1], 90), np.nanmedian(last[sl, 0, 1])) +
                  " | physics " + " ".join("{}:{:.0f}".format(i, np.nanmedian(last[sl, 0, i])) for i in (27, 29, 30, 31, 2, 3) if not np.all(np.isnan(last[sl, 0, i]))) +
                  " | gym " + " ".join("{}:{:.0f}".format(i, np.nanmedian(last[sl, 1, i])) for i in (24, 16, 23, 2, 25, 4, 5, 6, 9) if not np.all(np.isnan(last[sl, 1, i]))))
        # the median stamps of each class of block (where does an ending block lose its time?)
        for name, m in classes.items():
            if m.sum() >= 3:
                cm = np.nanmedian(last[m], axis=0)
                for w in range(2):
                    print("      {} wave {}: ".format(name[:12], w) + "  ".join(
                        "{}:{:.0f}".format(i, cm[w][i]) for i in np.argsort(np.where(np.isnan(cm[w]), 1e18, cm[w])) if not np.isnan(cm[w][i])))
        # which piece of the draw costs what: lifetime and the gym wave's arrival at barrier A (stamp 2) by the stage reached
        g2 = last[:, 1, 2]
        for st in range(1, 7):
            m = ((stage == st).sum(axis=1) > 0) & ~done_b
            if m.sum():
                print("   draw stage now {}: blocks {:4d}  lifetime median {:.0f} p90 {:.0f} | gym wave at barrier A median {:.0f} p90 {:.0f}".format(
                    st, int(m.sum()), np.median(life_b[m]), np.percentile(life_b[m], 90), np.nanmedian(g2[m]), np.nanpercentile(g2[m], 90)))
        m = ~done_b & ~early_b & ~draw_b
        print("   plain: gym wave at barrier A median {:.0f} p90 {:.0f}; physics wave median {:.0f} p90 {:.0f}".format(
            np.nanmedian(g2[m]), np.nanpercentile(g2[m], 90), np.nanmedian(last[m, 0, 2]), np.nanpercentile(last[m, 0, 2], 90)))
        # dispatch ramp: the counter is per XCD (block b runs on XCD b % 8), so starts and ends are compared within an XCD only
        Tl = T[-1]
        st_b, en_b = np.nanmin(Tl[:, :, 0], axis=1), np.nanmax(Tl, axis=(1, 2))
        ramp, span, last_is = [], [], []
        for x in range(8):
            idx = np.arange(x, Tl.shape[0], 8)
            z = np.nanmin(st_b[idx])
            ramp.append(np.nanmax(st_b[idx]) - z)
            span.append(np.nanmax(en_b[idx]) - z)
            j = idx[int(np.nanargmax(en_b[idx]))]
            last_is.append((int(j), float(st_b[j] - z), float(life_b[j]), bool(done_b[j])))
        res[key]["xcd_start_ramp_ticks"] = [float(r) for r in ramp]
        res[key]["xcd_span_ticks"] = [float(r) for r in span]
        print("   per XCD: first-to-last block start {}  | first start to last end {}".format(
            " ".join("{:.0f}".format(r) for r in ramp), " ".join("{:.0f}".format(r) for r in span)))
        print("   last block to finish per XCD (block, started at, lifetime, hosted an end): " +
              "  ".join("{}:{:.0f}+{:.0f}{}".format(j, a, l, "E" if d else "") for j, a, l, d in last_is))
        for worst in [int(order[0]), int(order[8]), int(order[40]), int(order[200])]:
            print("   block {} of the last launch (lifetime {:.0f}, done lanes in the wave: {}{}):".format(
                worst, np.nanmax(last[worst]), int(vec._done[worst * 64:(worst + 1) * 64].sum().item()),
                "" if steps_now is None else ", lanes in their first 8 steps: {}".format(int((steps_now[worst * 64:(worst + 1) * 64] <= 8).sum()))))
            for w in range(2):
                if not np.all(np.isnan(last[worst][w])):
                    print("      wave {}: ".format(w) + "  ".join("{}:{:.0f}".format(i, last[worst][w][i]) for i in np.argsort(np.where(np.isnan(last[worst][w]), 1e18, last[worst][w])) if not np.isnan(last[worst][w][i])))
        vec.close()
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "timeline.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    else:
        build(sys.argv[2:])
