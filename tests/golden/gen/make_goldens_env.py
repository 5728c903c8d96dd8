"""Golden-vector generator, environment part (G1-G4 of SURVEY 8c).

Runs the REAL reference (read-only under /root/reference, imported through the tiny shims in ./stubs) in the
build container and writes small .npz fixtures under tests/golden/.  The reference never travels to the GPU
box; only these data files do.  Usage:  python tests/golden/gen/make_goldens_env.py
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


def capture_trace(seed, P, W, H, blocks, variance, T, policy, save_raser):
    from environment.pursuit_evasion_game.pursuit_env import Pursuit_Env
    cfg = refload.load_cfg(num_defender=P, map_size=(W, H), blocks=blocks, variance=variance, max_steps=T)
    refload.seed_all(seed)
    env = Pursuit_Env(cfg)
    drawn = []
    orig_init_target = env.init_target

    def logged_init_target(inflated_map):
        orig_init_target(inflated_map=inflated_map)
        drawn.append(tuple(env.target[0]))
    env.init_target = logged_init_target
    env.reset()
    n_obs = len(env.boundary_map.obstacles)
    init = dict(
        grid=np.asarray(env.occupied_map.grid_map, np.uint8),
        inflated=np.asarray(env.inflated_map.grid_map, np.uint8),
        obs_xy=np.asarray(env.boundary_map.obstacles, np.int32).reshape(-1, 2),
        target0=np.asarray(env.target[0], np.int32),
        defenders0=np.asarray(env.get_state('defender'), np.float64),
        evader0=np.asarray(env.get_state('attacker')[0], np.float64),
    )
    rng = np.random.default_rng(seed + 1000)
    rec = {k: [] for k in ("p_state", "e_state", "p_adj", "e_adj", "o_adj", "reward", "action", "target", "path_len",
                           "p_after", "e_after")}
    paths = []
    done = False
    for t in range(T):
        rec["p_state"].append(np.asarray(env.get_state('defender'), np.float64))
        rec["e_state"].append(np.asarray(env.get_state('attacker'), np.float64))
        rec["p_adj"].append(np.asarray(env.communicate(), np.uint8))
        o_adj, e_adj = env.sensor()
        rec["o_adj"].append(np.asarray(o_adj, np.uint8).reshape(P, n_obs))
        rec["e_adj"].append(np.asarray(e_adj, np.uint8))
        if policy == "demon":
            a = env.demon() if (t // 25) % 2 == 0 else list(rng.integers(0, 9, P))
        elif policy == "stress":
            # round 2: the order-dependent corner of Pursuit_Env.step -- run for the nearest map border (proposals leave the clip
            # box, accepted ones are clipped in place), then herd towards the centroid (inner collisions), alternating
            st = np.asarray(env.get_state('defender'), np.float64)
            ang = np.arange(8) * np.pi / 4
            if (t // 30) % 2 == 0:
                db = np.stack((st[:, 0], W - 1 - st[:, 0], st[:, 1], H - 1 - st[:, 1]), -1)
                want = np.array([np.pi, 0.0, -np.pi / 2, np.pi / 2])[db.argmin(-1)]
            else:
                cen = st[:, :2].mean(0)
                want = np.arctan2(cen[1] - st[:, 1], cen[0] - st[:, 0])
            a = [int(np.abs(np.angle(np.exp(1j * (w - ang)))).argmin()) for w in want]
            if t % 7 == 3:
                a[int(rng.integers(0, P))] = int(rng.integers(0, 9))
        else:
            a = list(rng.integers(0, 9, P))
        env.attacker_step()
        path = np.asarray(env.attacker_list[0].path, np.int16).reshape(-1, 2)
        paths.append(path)
        rec["path_len"].append(len(path))
        r, done, _ = env.step([int(x) for x in a])
        rec["reward"].append(np.asarray(r, np.float64))
        rec["action"].append(np.asarray(a, np.int32))
        rec["target"].append(np.asarray(env.target[0], np.int32))
        rec["p_after"].append(np.asarray(env.get_state('defender'), np.float64))
        rec["e_after"].append(np.asarray(env.get_state('attacker')[0], np.float64))
    assert done
    out = dict(init)
    for k, v in rec.items():
        out[k] = np.stack(v)
    out["o_adj"] = np.packbits(out["o_adj"], axis=-1)
    out["n_obs"] = np.int32(n_obs)
    out["paths_cat"] = np.concatenate(paths, 0)
    out["drawn_targets"] = np.asarray(drawn, np.int32)  # [0] is the reset draw, the rest are mid-episode re-draws
    out["meta"] = np.asarray([seed, P, W, H, blocks, variance, T], np.int64)
    out["collision_flag"] = np.int32(bool(env.collision))
    if save_raser:
        out["raser"] = np.packbits(np.asarray(env.raser_map, np.uint8).reshape(W * H, n_obs), axis=-1)
    return out


def capture_astar_cases():
    from environment.pursuit_evasion_game.astar import AStar_2D
    cases = []
    rng = np.random.default_rng(7)

    def run(W, H, obs_grid, s, g):
        obs = [tuple(c) for c in np.argwhere(obs_grid).tolist()]
        path, closed = AStar_2D(width=W, height=H).searching(s_start=tuple(s), s_goal=tuple(g), obs=obs)
        cases.append((W, H, obs_grid.copy(), tuple(s), tuple(g), np.asarray(path, np.int16).reshape(-1, 2), len(closed)))

    for k in range(22):
        W, H = (20, 20) if k % 3 == 0 else (40, 40)
        obs = np.zeros((W + 1, H + 1), np.uint8)
        dens = [0.0, 0.1, 0.2, 0.3, 0.38][k % 5]
        obs[:W, :H] = rng.random((W, H)) < dens
        free = np.argwhere(obs[:W, :H] == 0)
        s = free[rng.integers(len(free))]; g = free[rng.integers(len(free))]
        run(W, H, obs, s, g)
    # walls with a gap / without a gap (no path) / goal inside an obstacle / start == goal / start in obstacle
    W = H = 20
    obs = np.zeros((W + 1, H + 1), np.uint8); obs[10, 0:18] = 1
    run(W, H, obs, (2, 2), (18, 3))
    obs = np.zeros((W + 1, H + 1), np.uint8); obs[10, 0:20] = 1          # closed inside the map; detour through y == H row (legal!)
    run(W, H, obs, (2, 2), (18, 3))
    obs = np.zeros((W + 1, H + 1), np.uint8); obs[5:9, 5:9] = 1; obs[6:8, 6:8] = 0  # goal enclosed: no path
    run(W, H, obs, (1, 1), (6, 6))
    obs = np.zeros((W + 1, H + 1), np.uint8); obs[12, 12] = 1            # goal in obs
    run(W, H, obs, (1, 1), (12, 12))
    obs = np.zeros((W + 1, H + 1), np.uint8)
    run(W, H, obs, (7, 7), (7, 7))                                        # start == goal
    obs = np.zeros((W + 1, H + 1), np.uint8); obs[3, 3] = 1
    run(W, H, obs, (3, 3), (10, 10))                                      # start inside an obstacle
    obs = np.zeros((W + 1, H + 1), np.uint8)
    run(W, H, obs, (0, 0), (19, 19))
    run(W, H, obs, (19, 0), (0, 19))
    obs = np.zeros((W + 1, H + 1), np.uint8); obs[0:19, 10] = 1
    run(W, H, obs, (5, 2), (5, 18))
    out = {"n": np.int32(len(cases))}
    for i, (W, H, obs, s, g, path, nclosed) in enumerate(cases):
        out[f"c{i}_WH"] = np.asarray([W, H], np.int32)
        out[f"c{i}_obs"] = obs
        out[f"c{i}_sg"] = np.asarray([*s, *g], np.int32)
        out[f"c{i}_path"] = path
        out[f"c{i}_nclosed"] = np.int32(nclosed)
    return out


def capture_reward_norm():
    from DHGN.normalization import Normalization
    rng = np.random.default_rng(3)
    nz = Normalization(shape=8)
    xs, ys = [], []
    for t in range(40):
        x = [int(v) for v in rng.integers(-2, 2, 8)]
        xs.append(x)
        ys.append(np.asarray(nz(x), np.float64))
    return dict(x=np.asarray(xs, np.float64), y=np.stack(ys), mean=np.asarray(nz.running_ms.mean, np.float64),
                S=np.asarray(nz.running_ms.S, np.float64), n=np.int64(nz.running_ms.n))


def main():
    jobs = [
        # name, seed, P, W, H, blocks, variance, T, policy, raser
        ("env_trace_20x20_p4_s0", 0, 4, 20, 20, 2, 4, 60, "demon", True),
        ("env_trace_20x20_p4_s1", 1, 4, 20, 20, 2, 4, 60, "random", False),
        ("env_trace_20x20_p4_s2", 2, 4, 20, 20, 2, 4, 60, "demon", False),
        ("env_trace_40x40_p8_s0", 0, 8, 40, 40, 5, 10, 150, "demon", True),
        ("env_trace_40x40_p8_s1", 1, 8, 40, 40, 5, 10, 150, "random", False),
        ("env_trace_40x40_p8_s2", 2, 8, 40, 40, 5, 10, 150, "demon", True),
        ("env_trace_40x40_p8_s3", 3, 8, 40, 40, 5, 10, 150, "random", False),
        ("env_trace_40x40_p8_s4", 4, 8, 40, 40, 5, 10, 150, "demon", False),
    ]
    if "--stress-only" in sys.argv:   # round 2 additions (the round-1 fixtures above are unchanged)
        jobs = []
    jobs += [
        ("env_trace_20x20_p4x_s5", 5, 4, 20, 20, 2, 4, 120, "stress", False),
        ("env_trace_20x20_p4x_s6", 6, 4, 20, 20, 2, 4, 120, "stress", False),
        ("env_trace_40x40_p8_s5", 5, 8, 40, 40, 5, 10, 150, "stress", False),
    ]
    for (name, seed, P, W, H, blocks, var, T, pol, ras) in jobs:
        out = capture_trace(seed, P, W, H, blocks, var, T, pol, ras)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "n_obs", int(out["n_obs"]), "redraws", len(out["drawn_targets"]) - 1, "sum r", out["reward"].sum(),
              "e_adj hits", int(out["e_adj"].sum()), flush=True)
    if "--stress-only" in sys.argv:
        return
    np.savez_compressed(os.path.join(OUT, "astar_cases.npz"), **capture_astar_cases())
    np.savez_compressed(os.path.join(OUT, "reward_norm.npz"), **capture_reward_norm())
    print("done")


if __name__ == "__main__":
    main()
