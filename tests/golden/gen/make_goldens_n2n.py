"""Golden-vector generator for the second batched environment, env_n2n.

Runs the reference's environment/env_n2n/particle_env.py (continuous pursuit, heading-rate-limited kinematics,
kill-radius reward, active masks) with its own SLSQP evader (scipy) and records the evader's normalised heading
command per step: the product and the oracle take that command as an input tape (SLSQP itself is scipy code, not
under /root/reference: parity of the minimiser is unpinned).  No trainer in the reference drives this env,
so the step order used here is ours: observe -> evader_step -> step(actions).
Usage:  python tests/golden/gen/make_goldens_n2n.py
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


def capture(seed, P, E, T):
    from environment.env_n2n import particle_env as pe
    from environment.env_n2n import eva
    random.seed(seed); np.random.seed(seed)
    env = pe.ParticleEnv()
    env.initialize(P, E)
    env.episode_limit = T
    env.reset()
    logged = []
    orig = eva.e_f

    def spy(*a, **k):
        v = orig(*a, **k)
        logged.append(float(v))
        return v
    eva.e_f = spy
    rng = np.random.default_rng(seed + 50)
    rec = {k: [] for k in ("p", "e", "pp_adj", "pe_adj", "action", "e_cmd", "reward", "active", "done")}

    def full_state(is_p):
        lst, idxs = (env.p_list, env.p_idx) if is_p else (env.e_list, env.e_idx)
        return np.asarray([[lst[f"{i}"].x, lst[f"{i}"].y, lst[f"{i}"].phi, lst[f"{i}"].v, float(lst[f"{i}"].active)] for i in idxs], np.float64)
    out = dict(target=np.asarray(env.target, np.float64), p0=full_state(True), e0=full_state(False))
    done = False
    while not done:
        ps, es = env.get_team_state(True, rules=False), env.get_team_state(False, rules=False)
        rec["p"].append(full_state(True)); rec["e"].append(full_state(False))
        rec["pp_adj"].append(env.get_adj_mat(ps, ps, env.p_comm_range, True).astype(np.uint8))
        rec["pe_adj"].append(env.get_adj_mat(ps, es, env.p_sen_range, True).astype(np.uint8))
        del logged[:]
        env.evader_step(env.get_team_state(True, rules=True))
        cmd = np.zeros(E)
        k = 0
        for i in env.e_idx:  # e_f is only called for evaders that were active
            if rec["e"][-1][i, 4] > 0:
                cmd[i] = logged[k]; k += 1
        rec["e_cmd"].append(cmd)
        a = rng.integers(0, 9, P)
        if seed % 2 == 0:  # steer pursuers roughly at the evader most of the time so captures / collisions happen
            ex, ey = es[0][0], es[0][1]
            for i in range(P):
                if rng.random() < 0.7:
                    ang = np.arctan2(ey - ps[i][1], ex - ps[i][0])
                    a[i] = int(np.round(ang / (np.pi / 4))) % 8 or 8
        reward, done, active = env.step([int(v) for v in a])
        rec["action"].append(a.astype(np.int32)); rec["reward"].append(np.asarray(reward, np.float64))
        rec["active"].append(np.asarray(active, np.uint8)); rec["done"].append(np.uint8(done))
    eva.e_f = orig
    out["p_end"] = full_state(True); out["e_end"] = full_state(False)
    for k, v in rec.items():
        out[k] = np.stack(v)
    out["meta"] = np.asarray([seed, P, E, T], np.int64)
    out["cfg"] = np.asarray([env.p_vmax, env.e_vmax, env.p_sen_range, env.p_comm_range, env.kill_radius, env.ang_lmt, env.step_size], np.float64)
    return out


def main():
    for name, seed, P, E, T in (("n2n_p4_s0", 0, 4, 1, 60), ("n2n_p4_s1", 1, 4, 1, 60), ("n2n_p16_s2", 2, 16, 1, 100),
                                ("n2n_p16_s3", 3, 16, 1, 100), ("n2n_p16_e2_s4", 4, 16, 2, 100), ("n2n_p16_s5", 5, 16, 1, 100),
                                ("n2n_p8_s7", 7, 8, 1, 100)):
        o = capture(seed, P, E, T)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **o)
        print(name, "steps", len(o["done"]), "reward sum", o["reward"].sum(), "pursuers left", int(o["active"][-1].sum()),
              "evaders left", int(o["e_end"][:, 4].sum()), flush=True)


if __name__ == "__main__":
    main()
