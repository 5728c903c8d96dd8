"""BASELINE-size checks (4096 environments, P=8, 40x40, T=150) through size-independent properties: reset invariants
(SURVEY Q16), determinism, structural properties of the observations, sampled oracle spot checks, GAE linearity."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_env():
    from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    cfg = baseline_config("cfg3", **{"runtime.device_reset": False})  # host resetter: env.last_init holds the host arrays
    env = Pursuit_Env(cfg, num_envs=4096)
    env.reset()
    return cfg, env


def test_reset_invariants_at_full_size(big_env):
    cfg, env = big_env
    init = env.last_init
    N = 4096
    grid = init["grid"].reshape(N, 40, 40)
    assert (grid.reshape(N, -1).sum(1) <= 5 * 36).all() and (grid.reshape(N, -1).sum(1) > 0).any()
    d = init["defenders"]
    diff = d[:, :, None, :2] - d[:, None, :, :2]
    dist = np.sqrt((diff ** 2).sum(-1)) + np.eye(8)[None] * 1e9
    assert (dist.min((1, 2)) >= 4.0).all()                                   # defenders pairwise >= min_dist
    for k in range(1, 8):                                                     # each later defender within comm range of 1-2 earlier ones
        conn = (dist[:, k, :k] < 16).sum(1)
        assert ((conn >= 1) & (conn <= 2)).all()
    assert (d[..., 2:] == 0).all() and (d[..., 0] >= 0).all() and (d[..., 0] <= 39).all()
    cells = np.rint(d[..., :2]).astype(int)
    infl = np.zeros((N, 44, 44), bool)                                        # static map inflated by 2 (padded by 2)
    g = np.pad(grid.astype(bool), ((0, 0), (2, 2), (2, 2)))
    for dx in range(-2, 3):
        for dy in range(-2, 3):
            infl |= np.roll(np.roll(g, dx, 1), dy, 2)
    n_idx = np.arange(N)
    tg = init["target"]
    assert not infl[n_idx, tg[:, 0] + 2, tg[:, 1] + 2].any()                  # target on a free cell of the inflated map
    e = init["evader"]
    ec = np.rint(e[:, :2]).astype(int)
    assert not infl[n_idx, ec[:, 0] + 2, ec[:, 1] + 2].any()
    near = np.sqrt(((cells - e[:, None, :2]) ** 2).sum(-1)).min(1)
    assert (near < 8.0).all()                                                 # evader perceived by some defender cell
    assert (init["n_obs"] <= 176).all() and (init["n_obs"] > 0).mean() > 0.99


def test_observation_structure_and_determinism_at_full_size(big_env):
    cfg, env = big_env
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    obs = {k: v.clone() for k, v in env.observe().items()}
    again = env.observe()
    for k in obs:
        assert torch.equal(obs[k], again[k]), k                               # observe is idempotent
    pa = obs["p_adj"]
    assert (pa[:, :, 1] == 1).all()                                           # column-1 quirk (Q2)
    assert (torch.tril(pa, -1)[:, :, [0] + list(range(2, 8))] == 0).all()     # lower triangle empty except column 1
    assert (torch.diagonal(pa, dim1=1, dim2=2) == 1).all()
    oa = obs["o_adj"]
    assert set(oa.unique().tolist()) <= {0.0, 1.0} and (oa.sum(-1) <= 36).all()   # one hit per beam at most
    pad = torch.arange(176, device="cuda")[None, None, :] >= env.n_obs[:, None, None]
    assert (oa[pad.expand_as(oa)] == 0).all()                                 # padded obstacle slots stay zero
    other = Pursuit_Env(cfg, num_envs=4096)                                   # same seeds -> same episode
    other.reset()
    o2 = other.observe()
    for k in obs:
        assert torch.equal(obs[k], o2[k]), k
    a = torch.randint(0, 9, (4096, 8), dtype=torch.int32, device="cuda")
    for e in (env, other):
        e.attacker_step()
        for _ in range(12):
            e.tick(a, e.observe(), torch.zeros(4096, 8, device="cuda"))
    assert torch.equal(env.sim.defs, other.sim.defs) and torch.equal(env.sim.eva, other.sim.eva) and torch.equal(env.sim.meta[:, :7], other.sim.meta[:, :7])  # [7] is a cycle-count diagnostic
    assert not env.sim.status().any()


def test_sampled_environments_match_oracle_at_full_size():
    """64 of the 4096 environments replayed in the CPU oracle for 30 ticks: bit-identical state."""
    from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    from oracle import pe_oracle
    cfg = baseline_config("cfg3", **{"runtime.device_reset": False})
    env = Pursuit_Env(cfg, num_envs=4096, seeds=list(range(7000, 7000 + 4096)))
    env.reset()
    init = env.last_init
    pick = np.linspace(0, 4095, 64).astype(int)
    ocfg = pe_oracle.make_config(W=40, H=40, P=8, O=176, max_steps=150, tape_len=16)
    oes = []
    for n in pick:
        oe = pe_oracle.OracleEnv(ocfg)
        k = int(init["n_obs"][n])
        oe.load(init["grid"][n], init["obs_xy"][n, :k], init["defenders"][n], init["evader"][n], init["target"][n], init["tape"][n])
        oes.append(oe)
    obs = env.observe(); env.attacker_step()
    for oe in oes:
        oe.observe(); oe.evader_step()
    g = torch.Generator(device="cuda").manual_seed(0)
    rew = torch.zeros(4096, 8, device="cuda")
    for t in range(30):
        a = torch.randint(0, 9, (4096, 8), dtype=torch.int32, device="cuda", generator=g)
        env.tick(a, obs, rew)
        an = a.cpu().numpy()
        for n, oe in zip(pick, oes):
            oe.step(an[n]); oe.observe(); oe.evader_step()
    defs, eva = env.sim.defenders_aos().cpu().numpy(), env.sim.eva.cpu().numpy()
    oadj = obs["o_adj"].cpu().numpy()
    for n, oe in zip(pick, oes):
        st = oe.state()
        assert np.array_equal(defs[n], st["defenders"]) and np.array_equal(eva[n], st["evader"]), n
        assert np.array_equal(oadj[n], oe.observe()[4]), n


def test_gae_is_linear_in_rewards_at_full_size():
    from distributed_multi_agent_reinforcement_learning_amd import ops
    N, T, P = 4096, 150, 8
    g = torch.Generator(device="cuda").manual_seed(1)
    r1, r2 = torch.randn(N, T, P, device="cuda", generator=g), torch.randn(N, T, P, device="cuda", generator=g)
    v = torch.zeros(N, T + 1, P, device="cuda")
    act = torch.ones(N, T, P, device="cuda")
    a1, _ = ops.gae_advnorm(r1, v, act, 0.99, 0.95, False)
    a2, _ = ops.gae_advnorm(r2, v, act, 0.99, 0.95, False)
    a12, vt = ops.gae_advnorm(r1 + 2 * r2, v, act, 0.99, 0.95, False)
    assert torch.allclose(a12, a1 + 2 * a2, rtol=1e-4, atol=1e-4)
    assert torch.equal(vt, a12)                                               # v == 0: target equals the advantage
    an, _ = ops.gae_advnorm(r1, v, act, 0.99, 0.95, True)
    assert abs(float(an.mean())) < 1e-4 and abs(float(an.std()) - 1.0) < 1e-3


def test_gae_advnorm_is_deterministic_at_full_size():
    """The advantage normaliser's statistics are per-workgroup f64 partials added in index order (k_gae_scan / k_gae_center; rounds
    1-3 used f64 atomicAdd, whose order varies from launch to launch): the same input gives the same BITS every time, at the benchmark
    batch (4096 x 150 x 8) and at a ragged size, and the statistics are those of an f64 torch evaluation (DHGN/mappo_parallel.py:643-658)."""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    for (N, T, P) in ((4096, 150, 8), (1031, 37, 5)):
        g = torch.Generator(device="cuda").manual_seed(N)
        r = torch.randn(N, T, P, device="cuda", generator=g) * 3 + 0.5
        v = torch.randn(N, T + 1, P, device="cuda", generator=g)
        act = (torch.rand(N, T, P, device="cuda", generator=g) < 0.95).float()
        first = ops.gae_advnorm(r, v, act, 0.99, 0.95, True)
        for _ in range(5):
            again = ops.gae_advnorm(r, v, act, 0.99, 0.95, True)
            assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1])
        raw, _ = ops.gae_advnorm(r, v, act, 0.99, 0.95, False)
        mean, std = raw.double().mean(), raw.double().std()
        want = ((raw.double() - mean) / (std + 1e-5) * act.double())
        assert float((first[0].double() - want).abs().max()) < 2e-6


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("name", ["cfg2", "cfg3"])
def test_full_size_training_iteration(name):
    """One rollout + PPO update at BASELINE size (4096 envs x 150 steps, mini-batches of 410 episodes = 492 000 rows): every
    kernel sees its production shape (split-K weight gradients, persistent GRU, fused cell, depth-3 FCRA for cfg3); the
    buffer obeys the domain invariants and the update moves every parameter by a finite amount."""
    import torch
    from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
    from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
    tr = Trainer(baseline_config(name))
    before = [p.detach().clone() for p in tr.agent.ac_parameters]
    steps, exp_r = tr.iterate()
    torch.cuda.synchronize()
    assert steps == 4096 * 150
    buf = tr.agent.minibuffer.buffer
    a = buf["a_n"]
    assert a.min().item() >= 0 and a.max().item() <= 8 and torch.equal(a, a.round())
    assert torch.isfinite(buf["r"]).all() and torch.isfinite(buf["v_n"]).all() and (buf["a_logprob_n"] <= 0).all()
    adj = buf["p_adj"]
    assert ((adj == 0) | (adj == 1)).all()
    import math
    assert all(math.isfinite(float(v)) for v in tr.last_log) and math.isfinite(float(exp_r))
    moved = 0
    for p, b in zip(tr.agent.ac_parameters, before):
        assert torch.isfinite(p).all()
        moved += int(not torch.equal(p.detach(), b))
    assert moved == len(before)
    tr.env.check_status()


# ---- BASELINE configs 4 and 5 at their sizes (configs_untested) ---------------------------------------------------
def test_config4_env_n2n_16_pursuers_8192_envs_sampled_oracle_replay():
    """env_n2n at BASELINE config 4's size (16 pursuers, 8192 environments): a seeded batch stepped for 40 ticks; 96 sampled
    environments (first / last lanes of waves, first / last workgroups included) replayed in the CPU oracle -- every discrete
    output identical, f64 state within 1e-9 -- plus whole-batch invariants."""
    from distributed_multi_agent_reinforcement_learning_amd.n2n_env import ParticleEnv
    from oracle import n2n_oracle as no
    P, E, N, T = 16, 2, 8192, 40
    env = ParticleEnv(num_envs=N, seeds=list(range(5000, 5000 + N)), episode_limit=T)
    env.initialize(P, E)
    env.reset()
    p0, e0, tg = env.last_init
    pick = np.unique(np.concatenate([np.arange(0, 8), np.arange(N - 8, N), np.linspace(0, N - 1, 80).astype(int)]))
    cfg = no.make_cfg(P, E, T)
    oenvs = {int(n): no.OracleN2n(cfg, p0[n], e0[n], tg[n]) for n in pick}
    rng = np.random.default_rng(4)
    for t in range(T):
        acts = rng.integers(0, 9, (N, P)).astype(np.int32)
        cmd = rng.uniform(-1, 1, (N, E))
        env.evader_step(cmd)
        r, done, act = env.step(acts)
        r, done, act = r.cpu().numpy(), done.cpu().numpy(), act.cpu().numpy()
        assert np.isfinite(r).all() and set(np.unique(act).tolist()) <= {0, 1}
        p = env.p.permute(0, 2, 1).cpu().numpy()
        for n, oe in oenvs.items():
            oe.evader_step(cmd[n])
            ro, do, ao = oe.step(acts[n])
            assert np.array_equal(r[n], ro.astype(np.float32)) and np.array_equal(act[n], ao) and bool(done[n]) == do, (t, n)
            assert np.max(np.abs(p[n] - oe.p)) <= 1e-9, (t, n)
    pp = env.obs["pp_adj"]
    assert ((pp == 0) | (pp == 1)).all() and pp.shape == (N, P, P)
    assert done.all()                                            # episode_limit reached everywhere
    assert (env.active_t.sum(1) < P).any()


def test_config5_env_3d_8_pursuers_2048_envs_sampled_oracle_replay():
    """env_3d at BASELINE config 5's size (8 pursuers, 2048 environments): as above, 64 sampled environments in the oracle."""
    from distributed_multi_agent_reinforcement_learning_amd.e3d_env import ParticleEnv
    from oracle import e3d_oracle as eo
    P, N, T = 8, 2048, 50
    env = ParticleEnv(num_envs=N, seeds=list(range(9000, 9000 + N)), max_step=T)
    env.initialize(P)
    env.reset()
    p0, e0, tg = env.last_init
    pick = np.unique(np.concatenate([np.arange(0, 8), np.arange(N - 8, N), np.linspace(0, N - 1, 48).astype(int)]))
    cfg = eo.make_cfg(P, T)
    oenvs = {int(n): eo.OracleE3d(cfg, p0[n], e0[n], tg[n]) for n in pick}
    rng = np.random.default_rng(5)
    for t in range(T):
        acts = rng.uniform(-1, 1, (N, P, 3))
        pe = env.e[:, None, :3].cpu().numpy() - env.p.permute(0, 2, 1)[:, :, :3].cpu().numpy()
        chase = rng.random((N, P)) < 0.5     # half of the pursuers chase the evader so that captures / collisions happen
        acts[..., 0] = np.where(chase, np.arctan2(pe[..., 1], pe[..., 0]) / np.pi, acts[..., 0])
        acts[..., 1] = np.where(chase, np.arctan2(pe[..., 2], np.hypot(pe[..., 0], pe[..., 1])) / (np.pi / 2), acts[..., 1])
        acts[..., 2] = np.where(chase, 1.0, acts[..., 2])
        cmd = rng.uniform(-1, 1, (N, 3))
        env.evader_step(cmd)
        r, done, act = env.step(acts)
        r, done, act = r.cpu().numpy(), done.cpu().numpy(), act.cpu().numpy()
        assert np.isfinite(r).all()
        p = env.p.permute(0, 2, 1).cpu().numpy(); e = env.e.cpu().numpy()
        pp, pe_adj = env.obs["pp_adj"].cpu().numpy(), env.obs["pe_adj"].cpu().numpy()
        for n, oe in oenvs.items():
            if oe.e[0, 6] > 0 and oe.p[:, 6].sum() > 0:
                oe.evader_step(cmd[n])
            ro, do, ao = oe.step(acts[n])
            assert np.array_equal(r[n], ro.astype(np.float32)) and np.array_equal(act[n], ao) and bool(done[n]) == do, (t, n)
            assert np.max(np.abs(p[n] - oe.p)) <= 1e-9 and np.max(np.abs(e[n] - oe.e[0])) <= 1e-9, (t, n)
            _, _, pp_o, pe_o = oe.observe()
            assert np.array_equal(pp[n], pp_o) and np.array_equal(pe_adj[n], pe_o), (t, n)
    assert (env.active_t.sum(1) < P).any() and (env.e[:, 6] == 0).any()


@pytest.mark.timeout(300)
def test_config4_shapes_training_iteration_at_1024_envs_per_rank():
    """The stand-in for BASELINE config 4's trainer (env_n2n has no trainer in the reference): 16 defenders on 64 x 64,
    DHGN depth 3, 1024 environments = one rank's share of 8192 over 8 GPUs; one rollout + update at that size."""
    import math
    from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
    from distributed_multi_agent_reinforcement_learning_amd.trainer import Trainer
    tr = Trainer(baseline_config("cfg4"))
    before = [p.detach().clone() for p in tr.agent.ac_parameters]
    steps, exp_r = tr.iterate()
    torch.cuda.synchronize()
    assert steps == 1024 * 150
    buf = tr.agent.minibuffer.buffer
    a = buf["a_n"]
    assert a.shape == (1024, 150, 16) and a.min().item() >= 0 and a.max().item() <= 8 and torch.equal(a, a.round())
    assert torch.isfinite(buf["r"]).all() and torch.isfinite(buf["v_n"]).all() and (buf["a_logprob_n"] <= 0).all()
    assert ((buf["p_adj"] == 0) | (buf["p_adj"] == 1)).all()
    assert all(math.isfinite(float(v)) for v in tr.last_log) and math.isfinite(float(exp_r))
    for p, b in zip(tr.agent.ac_parameters, before):
        assert torch.isfinite(p).all() and not torch.equal(p.detach(), b)
    tr.env.check_status()
    # 8 sampled environments of the rollout replayed in the CPU oracle from the recorded actions: the P = 16 tick (sequential
    # step scoring) at this size produced the reference observations and rewards
    from oracle import pe_oracle
    # (device reset: the oracle replay needs host initial conditions, so a second, host-reset environment repeats the check)
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    cfg = baseline_config("cfg4", **{"runtime.device_reset": False})
    env = Pursuit_Env(cfg, num_envs=1024, seeds=list(range(3000, 4024)))
    env.reset()
    init = env.last_init
    pick = np.linspace(0, 1023, 8).astype(int)
    ocfg = pe_oracle.make_config(W=64, H=64, P=16, O=cfg.map.num_max_obstacle, max_steps=150, tape_len=16)
    oes = []
    for n in pick:
        oe = pe_oracle.OracleEnv(ocfg)
        k = int(init["n_obs"][n])
        oe.load(init["grid"][n], init["obs_xy"][n, :k], init["defenders"][n], init["evader"][n], init["target"][n], init["tape"][n])
        oes.append(oe)
    obs = env.observe(); env.attacker_step()
    for oe in oes:
        oe.observe(); oe.evader_step()
    g = torch.Generator(device="cuda").manual_seed(0)
    rew = torch.zeros(1024, 16, device="cuda")
    for t in range(25):
        a = torch.randint(0, 9, (1024, 16), dtype=torch.int32, device="cuda", generator=g)
        env.tick(a, obs, rew)
        an = a.cpu().numpy()
        for n, oe in zip(pick, oes):
            oe.step(an[n]); oe.observe(); oe.evader_step()
    defs, eva = env.sim.defenders_aos().cpu().numpy(), env.sim.eva.cpu().numpy()
    oadj = obs["o_adj"].cpu().numpy()
    for n, oe in zip(pick, oes):
        st = oe.state()
        assert np.array_equal(defs[n], st["defenders"]) and np.array_equal(eva[n], st["evader"]), n
        assert np.array_equal(oadj[n], oe.observe()[4]), n
