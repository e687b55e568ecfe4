import copy, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from gym_fixed_wing import presets
from gym_fixed_wing.vec_env import FixedWingVecEnv
import test_shape_instance as T
warnings.simplefilter("ignore")
cfg, ckw, skw, _, _ = presets.workload("c3")
ckw = T._merged(ckw, T.VALUE_TWEAKS); skw = T._merged(skw, {"turbulence_intensity": "light"})
n = 1100
layout = sys.argv[1]
def make(mode):
    os.environ.pop("FWGYM_SHAPE", None)
    kw = dict(specialize=False)
    if mode == "generic": os.environ["FWGYM_SHAPE"] = "0"
    if mode == "jit": kw = dict(specialize=True)
    if mode.endswith(".so"): kw = dict(_lib_path=os.path.join(ROOT, "fixed-wing-gym_amd", "gym_fixed_wing", "_abl", mode))
    v = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw),
                        derived_views=False, seed=5, obs_layout=layout, **kw)
    os.environ.pop("FWGYM_SHAPE", None)
    return v
a, b = make(sys.argv[2]), make(sys.argv[3])
print("instances", a.spec_index, b.spec_index)
a.reset(); b.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
shown = 0
hist = []   # newest record of every env after each step (generic env b)
hist.append(b.reset().reshape(n, 5, 12)[:, 0, :].clone()) if False else None
for t in range(200):
    act = (torch.rand((n, 3), device="cuda", generator=gen) * 2 - 1) * (2.5 if t % 13 == 0 else 1.3)
    fl_before = a.field("flags").clone()
    age_before = a.field("steps_count").clone()
    oa, ra, da, _ = a.step(act); ob, rb, db, _ = b.step(act)
    assert torch.equal(da, db), t
    hist.append(ob.reshape(n, 5, 12)[:, 0, :].clone())
    if bool(da.any()):
        m = da.bool()
        d = (a._term_obs[m] - b._term_obs[m]).abs()
        if float(d.max()) > 1e-3 and shown < 6:
            idx = torch.nonzero(m).flatten()
            k = int(d.max(dim=1).values.argmax())
            e = int(idx[k])
            print("step", t, "env", e, "lane", e % 64, "ends in wave", int(m[(e//64)*64:(e//64)*64+64].sum()), "term code", int(a._term[e]), "steps_count", "max diff", float(d[k].max()))
            print("  a", np.round(a._term_obs[e].cpu().numpy().reshape(5, 12)[:, :4], 3).tolist())
            print("  b", np.round(b._term_obs[e].cpu().numpy().reshape(5, 12)[:, :4], 3).tolist())
            r4 = a._term_obs[e].reshape(5, 12)[4]
            best = min(range(len(hist)), key=lambda k: float((hist[k][e] - r4).abs().max()))
            print("  a's oldest row is closest to env's newest record after step", best, "err", float((hist[best][e] - r4).abs().max()),
                  "| new episode's first obs row0 err", float((oa.reshape(n, 5, 12)[e, 0] - r4).abs().max()))
            nbad = int(((a._term_obs[m] - b._term_obs[m]).abs().reshape(int(m.sum()), 5, 12).amax(dim=2) > 1e-3).sum(dim=0)[4])
            print("  lanes with a wrong oldest row this step:", nbad, "of", int(m.sum()), "ending")
            # where else do these twelve values exist?  (other envs' records of this step, in either env's outputs)
            for name, ten in (("a.obs", oa), ("b.obs", ob), ("a.term", a._term_obs), ("b.term", b._term_obs)):
                rows = ten.reshape(n, 5, 12)
                hit = torch.nonzero((rows - r4).abs().amax(dim=2) < 1e-4)
                if hit.numel():
                    print("   found in", name, "at (env, row):", hit.tolist()[:6])
            print("   a row4", [round(float(x), 4) for x in r4], "\n   b row4", [round(float(x), 4) for x in b._term_obs[e].reshape(5, 12)[4]])
            print("   a new-episode obs rows:", [[round(float(x), 3) for x in oa.reshape(n, 5, 12)[e, r, :4]] for r in range(5)])
            w0 = (e // 64) * 64
            st = ((fl_before[w0:w0 + 64].to(torch.int64) >> 4) & 7).tolist()
            dn = da[w0:w0 + 64].to(torch.int64).tolist()
            print("   draw stage before the step, this lane:", st[e % 64], "| ending lanes' stages:", sorted(set(st[i] for i in range(64) if dn[i])),
                  "| lanes of the wave with stage < 6:", [(i, st[i], dn[i]) for i in range(64) if st[i] < 6])
            print("   age before the step:", int(age_before[e]), "| ages of the wave's lanes:", sorted(set(int(x) for x in age_before[w0:w0 + 64].tolist())))
            dd = (a._term_obs[w0:w0 + 64] - b._term_obs[w0:w0 + 64]).abs().reshape(-1, 5, 12).amax(dim=2)[:, 4]
            print("   ending lanes of the wave (lane, flags before, wrong):", [(i, hex(int(fl_before[w0 + i]) & 0xFFFFFFFF), int(dd[i] > 1e-3)) for i in range(min(64, n - w0)) if dn[i]])
            st_all = a.state
            g0 = int(a.layout.draw) // 4
            tags = st_all[g0 + 10, w0:w0 + 64].view(torch.int32)
            print("   draw tags (lane: generation, episode, flags) of ending lanes:", [(i, tags[i].tolist()[:3]) for i in range(min(64, n - w0)) if dn[i]][:40])
            print("   termination codes of the wave's ending lanes (lane, code, wrong):", [(i, int(a._term[w0 + i]), int(dd[i] > 1e-3)) for i in range(min(64, n - w0)) if dn[i]])
            H = torch.stack(hist)   # [steps so far][env][12]: every env's newest record after every step
            hit = torch.nonzero((H - r4).abs().amax(dim=2) < 1e-3)
            print("   the same twelve values as the newest record of (after step, env):", hit.tolist()[:6])
            shown += 1
print("done")
