"""Golden-vector generator for the third batched environment, env_3d (BASELINE config 5).

Runs the reference's environment/env_3d/particle_env.py (continuous 3-D pursuit: heading / pitch / speed rate limits,
kill-radius reward, active masks, done rule) with its own SLSQP evader (scipy) and records the evader's 3-component command
per step: the product and the oracle take that command as an input tape (SLSQP is scipy code, not under /root/reference:
parity of the minimiser is unpinned).  Pursuer actions are CONTINUOUS (a in [-1, 1]^3, Point.step :25-55); no
trainer in the reference drives this env, so the step order used here is ours: observe -> evader_step -> step(actions).
Usage:  python tests/golden/gen/make_goldens_e3d.py
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refload  # noqa: E402

refload.activate()
OUT = os.path.dirname(HERE)


def capture(seed, P, T, chase):
    from environment.env_3d import particle_env as pe
    from environment.env_3d import eva
    random.seed(seed); np.random.seed(seed)
    env = pe.ParticleEnv()
    env.initialize(P)
    env.max_step = T
    env.reset()
    logged = []
    orig = eva.e_f

    def spy(*a, **k):
        v = orig(*a, **k)
        logged.append(np.asarray(v, np.float64).copy())
        return v
    eva.e_f = spy
    rng = np.random.default_rng(seed + 70)
    rec = {k: [] for k in ("p", "e", "pp_adj", "pe_adj", "action", "e_cmd", "reward", "active", "done")}

    def full_state(is_p):
        lst, idxs = (env.p_list, env.p_idx) if is_p else (env.e_list, env.e_idx)
        return np.asarray([[lst[f"{i}"].x, lst[f"{i}"].y, lst[f"{i}"].z, lst[f"{i}"].phi, lst[f"{i}"].gamma, lst[f"{i}"].v,
                            float(lst[f"{i}"].active)] for i in idxs], np.float64)
    out = dict(target=np.asarray(env.target, np.float64), p0=full_state(True), e0=full_state(False))
    done = False
    while not done:
        ps, es = env.get_team_state(True, rules=False), env.get_team_state(False, rules=False)
        rec["p"].append(full_state(True)); rec["e"].append(full_state(False))
        rec["pp_adj"].append(env.get_adj_mat(ps, ps, env.p_comm_range, True).astype(np.uint8))
        rec["pe_adj"].append(env.get_adj_mat(ps, es, env.p_sen_range, True).astype(np.uint8))
        del logged[:]
        alive = env.get_team_state(True, rules=True)
        cmd = np.zeros((1, 3))
        if len(alive) and rec["e"][-1][0, 6] > 0:
            env.evader_step(alive)
            cmd[0] = logged[0]
        rec["e_cmd"].append(cmd)
        a = rng.uniform(-1, 1, (P, 3))
        if chase:  # steer most pursuers at the evader so captures / collisions happen
            e = es[0]
            for i in range(P):
                if rng.random() < 0.75:
                    d = np.asarray(e[:3]) - np.asarray(ps[i][:3])
                    a[i, 0] = np.arctan2(d[1], d[0]) / np.pi
                    a[i, 1] = np.arctan2(d[2], np.hypot(d[0], d[1])) / (np.pi / 2)
                    a[i, 2] = 1.0
        reward, done, active = env.step([list(map(float, v)) for v in a])
        rec["action"].append(a.astype(np.float64)); rec["reward"].append(np.asarray(reward, np.float64))
        rec["active"].append(np.asarray(active, np.uint8)); rec["done"].append(np.uint8(done))
    eva.e_f = orig
    out["p_end"] = full_state(True); out["e_end"] = full_state(False)
    for k, v in rec.items():
        out[k] = np.stack(v)
    out["meta"] = np.asarray([seed, P, 1, T], np.int64)
    out["cfg"] = np.asarray([env.p_vmax, env.e_vmax, env.p_sen_range, env.p_comm_range, env.kill_radius, env.ang_lmt, env.v_lmt, env.step_size], np.float64)
    return out


def main():
    for name, seed, P, T, chase in (("e3d_p4_s0", 0, 4, 80, True), ("e3d_p4_s1", 1, 4, 60, False), ("e3d_p8_s2", 2, 8, 120, True),
                                    ("e3d_p8_s3", 3, 8, 120, True), ("e3d_p8_s4", 4, 8, 60, False), ("e3d_p3_s5", 5, 3, 200, True)):
        o = capture(seed, P, T, chase)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **o)
        print(name, "steps", len(o["done"]), "reward sum", o["reward"].sum(), "pursuers left", int(o["active"][-1].sum()),
              "evader alive", int(o["e_end"][0, 6]), flush=True)


if __name__ == "__main__":
    main()
