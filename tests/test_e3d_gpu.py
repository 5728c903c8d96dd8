"""GPU parity of the batched env_3d kernels (C ABI include/e3d_env.h) against the oracle and the reference goldens."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import e3d_oracle as eo
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(GOLDEN, "e3d_*.npz")))


def load(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["seed"], d["P"], d["E"], d["T"] = [int(v) for v in d["meta"]]
    return d


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_e3d_trace_parity(path):
    """Recorded initial condition, continuous pursuer actions and evader commands in; every discrete output (rewards, active,
    done, adjacency) identical to the reference, f64 state within 1e-9 (device cos/sin vs libm).  11 copies of the trace: the
    wave holds 8 environments, so copies sit in different lane groups, waves and workgroups."""
    from distributed_multi_agent_reinforcement_learning_amd.e3d_env import ParticleEnv
    d = load(path)
    M = 11
    env = ParticleEnv(num_envs=M, max_step=d["T"])
    env.initialize(d["P"])
    env.reset(init=(np.stack([d["p0"]] * M), np.stack([d["e0"][0]] * M), np.stack([d["target"]] * M)))
    for t in range(len(d["done"])):
        p = env.p.permute(0, 2, 1).cpu().numpy(); e = env.e.cpu().numpy()
        for n in range(M):
            assert np.max(np.abs(p[n] - d["p"][t])) <= 1e-9 and np.max(np.abs(e[n] - d["e"][t][0])) <= 1e-9, (t, n)
            assert np.array_equal(env.obs["pp_adj"][n].cpu().numpy(), d["pp_adj"][t].astype(np.float32)), t
            assert np.array_equal(env.obs["pe_adj"][n].cpu().numpy(), d["pe_adj"][t].astype(np.float32)), t
            assert np.allclose(env.obs["p_state"][n].cpu().numpy(), d["p"][t][:, :6].astype(np.float32), atol=1e-5)
            assert np.allclose(env.obs["e_state"][n].cpu().numpy(), d["e"][t][:, :6].astype(np.float32), atol=1e-5)
        env.evader_step(np.stack([d["e_cmd"][t][0]] * M))
        r, done, act = env.step(np.stack([d["action"][t]] * M))
        for n in range(M):
            assert np.array_equal(r[n].cpu().numpy(), d["reward"][t].astype(np.float32)), (t, n)
            assert np.array_equal(act[n].cpu().numpy(), d["active"][t]) and bool(done[n]) == bool(d["done"][t]), (t, n)


@pytest.mark.parametrize("P", [3, 8, 12])
def test_e3d_random_batch_matches_oracle_and_seeded_reset(P):
    from distributed_multi_agent_reinforcement_learning_amd.e3d_env import ParticleEnv
    N, T = 150, 50
    env = ParticleEnv(num_envs=N, seeds=list(range(300, 300 + N)), max_step=T)
    env.initialize(P)
    env.reset()
    p0, e0, tg = env.last_init
    for n in (0, 7, N - 1):  # the C++ reset replays numpy's legacy generator
        np.random.seed(300 + n)
        t_ref, p_ref, e_ref = eo.reset_oracle(P)
        assert np.array_equal(tg[n], t_ref) and np.array_equal(p0[n], p_ref) and np.array_equal(e0[n], e_ref[0])
    cfg = eo.make_cfg(P, T)
    oenvs = [eo.OracleE3d(cfg, p0[n], e0[n], tg[n]) for n in range(N)]
    rng = np.random.default_rng(P)
    for t in range(T):
        acts = rng.uniform(-1, 1, (N, P, 3))
        # half of the pursuers chase the evader so that captures / collisions happen
        pe = env.e[:, None, :3].cpu().numpy() - env.p.permute(0, 2, 1)[:, :, :3].cpu().numpy()
        chase = rng.random((N, P)) < 0.5
        acts[..., 0] = np.where(chase, np.arctan2(pe[..., 1], pe[..., 0]) / np.pi, acts[..., 0])
        acts[..., 1] = np.where(chase, np.arctan2(pe[..., 2], np.hypot(pe[..., 0], pe[..., 1])) / (np.pi / 2), acts[..., 1])
        acts[..., 2] = np.where(chase, 1.0, acts[..., 2])
        cmd = rng.uniform(-1, 1, (N, 3))
        env.evader_step(cmd)
        r, done, act = env.step(acts)
        r, done, act = r.cpu().numpy(), done.cpu().numpy(), act.cpu().numpy()
        p = env.p.permute(0, 2, 1).cpu().numpy(); e = env.e.cpu().numpy()
        pp, pe_adj = env.obs["pp_adj"].cpu().numpy(), env.obs["pe_adj"].cpu().numpy()
        for n, oe in enumerate(oenvs):
            if oe.e[0, 6] > 0 and oe.p[:, 6].sum() > 0:
                oe.evader_step(cmd[n])
            ro, do, ao = oe.step(acts[n])
            assert np.array_equal(r[n], ro.astype(np.float32)) and np.array_equal(act[n], ao) and bool(done[n]) == do, (t, n)
            assert np.max(np.abs(p[n] - oe.p)) <= 1e-9 and np.max(np.abs(e[n] - oe.e[0])) <= 1e-9, (t, n)
            _, _, pp_o, pe_o = oe.observe()
            assert np.array_equal(pp[n], pp_o) and np.array_equal(pe_adj[n], pe_o), (t, n)
    assert (env.active_t.sum(1) < P).any() and (env.e[:, 6] == 0).any()  # pursuers collided, evaders were caught
