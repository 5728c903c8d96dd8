"""The update's PRODUCTION backward kernels under the reference goldens and under a production-width oracle.

ops.py routes weight gradients, ReLU-backward / bias sums and skinny products to the HIP kernels (k_sb_wgrad / k_wgrad,
k_relu_bwd_colsum, k_wgrad_skinny) only from 4096 rows up; the reference fixtures have a few hundred rows, so the gradient
goldens of tests/test_mappo_gpu.py exercise torch.mm / aten there.  Here
  (a) the same reference goldens (DHGN/mappo_parallel.py:660-723 captured by tests/golden/gen/*) are re-run with every size
      threshold forced to 1 and call counters on the C-ABI entry points, in both `runtime.matmul` modes, and
  (b) one cfg3-shaped batch of 64 episodes x 150 steps x 8 agents (two mini-batches of 38 400 rows: above every threshold, the
      grouped epoch included) rolled out by the product is re-evaluated by oracle/model_oracle.py `train` in f64 on the GPU
      (plain torch; the Python reference never travels) -- losses and every gradient tensor within the noise-scaled tolerance of
      tests/test_mappo_gpu.py::test_train_full_gradient_tensors, the noise being the oracle's own fp32-vs-f64 difference.
"""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, buffer_tensors, load_model_golden, sharpen
from tests.test_mappo_gpu import close, make_agent

pytestmark = pytest.mark.gpu

ENTRY_POINTS = ("wgrad_split_tn", "wgrad_tn", "relu_bwd_colsum", "wgrad_skinny", "sb_gemm", "sb_gemm_masked", "sb_gemm_masked_bits", "sb_gemm_signs")


@pytest.fixture
def forced_kernels(monkeypatch):
    """every size threshold of the update's kernel routing at 1 + call counters on the library's entry points; restores the matmul mode"""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    for name in ("WGRAD_MIN_ROWS", "RELU_BWD_MIN_ROWS", "SKINNY_MIN_ROWS", "SORTED_ONES_MIN_QDIV", "MASKED_GRAD_MIN_ROWS"):
        monkeypatch.setattr(ops, name, 1)
    L = ops.load_library()
    calls = {n: 0 for n in ENTRY_POINTS}

    def wrap(name):
        real = getattr(L, name)

        def counted(*a):
            calls[name] += 1
            return real(*a)
        monkeypatch.setattr(L, name, counted)
    for n in ENTRY_POINTS:
        wrap(n)
    modes = ops.matmul_modes()
    yield calls
    ops.restore_matmul_modes(modes)


def _expected_entry_points(E, mode):
    """the HIP entry points that must have been on the path for a fixture of embedding width E"""
    want = {"wgrad_skinny"}
    if E % 128 == 0:      # the MFMA weight-gradient kernels cover multiples of 128 features
        want.add("wgrad_split_tn" if mode == "split_bf16" else "wgrad_tn")
        if mode == "split_bf16":
            want.add("sb_gemm")
    # ReLU backward + bias sums: in the input-gradient GEMM's epilogue (E = 128, split mode: every ReLU of the networks has such a
    # consumer), else the separate pass
    # consumer (E = 128, split mode: every ReLU of the networks has such a consumer; relu' travels as sign bits written by the ReLU
    # layer's own GEMM), else the separate pass
    want.update({"sb_gemm_masked_bits", "sb_gemm_signs"} if E == 128 and mode == "split_bf16" else {"relu_bwd_colsum"})
    return want


@pytest.mark.parametrize("mode", ["split_bf16", "fp32"])
@pytest.mark.parametrize("name", ["model_p4_20x20_d1", "model_p8_40x40_d3"])
def test_train_full_gradient_tensors_through_the_production_kernels(name, mode, forced_kernels):
    """tests/test_mappo_gpu.py::test_train_full_gradient_tensors with the kernels of the benchmark's update on the path: every gradient
    tensor of MAPPO.train against the reference's, element by element, same tolerance."""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    from distributed_multi_agent_reinforcement_learning_amd.mappo import ReplayBuffer
    GRAD_NOISE_FACTOR, GRAD_REL_FLOOR = 4.0, 2e-5
    d = load_model_golden(name)
    z = np.load(os.path.join(GOLDEN, f"grads_{name}.npz"))
    cfg, agent = make_agent(d, "Learner", **{"runtime.matmul": mode})
    assert ops.MATMUL_MODE == mode and ops.WGRAD_MODE == mode and ops.PROJ_MODE == mode
    sharpen(d, agent.actor)
    rb = ReplayBuffer.from_tensors(cfg, buffer_tensors(d), d["init_n_obs"], agent.device)
    objC, objA, ag, cg = agent.train(rb, int(d["steps"]))
    assert close(agent.last_adv, d["gae_adv"], 1e-4) and close(agent.last_v_target, d["gae_v_target"], 1e-4)
    assert abs(objC - float(z["objC"])) <= 1e-4 * (1 + abs(float(z["objC"]))) and abs(objA - float(z["objA"])) <= 1e-4 * (1 + abs(float(z["objA"])))
    for who, grads, names in (("a", ag, z["actor_names"]), ("c", cg, z["critic_names"])):
        for n, g in zip(names, grads):
            n = str(n)
            ref, noise, scale = z[f"{who}grad_{n}"], float(z[f"{who}noise_{n}"]), float(z[f"{who}scale_{n}"])
            err = float(np.max(np.abs(np.asarray(g, np.float64) - ref)))
            assert err <= GRAD_NOISE_FACTOR * noise + GRAD_REL_FLOOR * scale, (who, n, err, noise, scale)
    calls = forced_kernels
    for ep in _expected_entry_points(d["E"], mode):
        assert calls[ep] > 0, (ep, "was not on the update's path", calls)
    other = "wgrad_tn" if mode == "split_bf16" else "wgrad_split_tn"
    if d["E"] % 128 == 0:
        # the other mode's MFMA weight-gradient kernel is only reached for shapes the split kernel does not cover
        assert calls[other] == 0 or mode == "split_bf16", calls
    if mode == "fp32":
        assert calls["sb_gemm"] == 0 and calls["wgrad_split_tn"] == 0, calls


@pytest.mark.parametrize("mode", ["split_bf16", "fp32"])
def test_sequence_mode_train_and_update_through_the_production_kernels(mode, forced_kernels):
    """tests/test_mappo_gpu.py::test_sequence_mode_and_train_reproduce_reference on the E = 128 fixture with the thresholds at 1: mode-1
    outputs, GAE, losses, gradient digests and the Adam step of Learner.set_gradients_and_update (runner.py:72-78)."""
    from distributed_multi_agent_reinforcement_learning_amd.mappo import ReplayBuffer
    from distributed_multi_agent_reinforcement_learning_amd.runner import Learner
    from tests.helpers import digest
    d = load_model_golden("model_p8_40x40_d3")
    cfg, agent = make_agent(d, "Learner", **{"runtime.matmul": mode})
    sharpen(d, agent.actor)
    rb = ReplayBuffer.from_tensors(cfg, buffer_tensors(d), d["init_n_obs"], agent.device)
    objC, objA, ag, cg = agent.train(rb, int(d["steps"]))
    assert abs(objC - float(d["objC"])) <= 1e-4 * (1 + abs(float(d["objC"]))) and abs(objA - float(d["objA"])) <= 1e-4 * (1 + abs(float(d["objA"])))
    for grads, dg, m in ((ag, d["actor_grad_digest"], agent.actor), (cg, d["critic_grad_digest"], agent.critic)):
        names = [n for n, _ in m.named_parameters()]
        for k, g, ref in zip(names, grads, dg):
            mine = digest(g)
            scale = max(1e-6, ref[2] / ref[0])
            assert abs(mine[2] - ref[2]) <= 3e-3 * ref[2] + 1e-7, (k, mine[2], ref[2])
            assert np.max(np.abs(mine[4:] - ref[4:])) <= 3e-3 * max(scale, np.max(np.abs(ref[4:]))) + 1e-7, k
    lrn = Learner.__new__(Learner)
    lrn.agent, lrn.learner_device, lrn.use_lr_decay = agent, agent.device, True
    lrn.set_gradients_and_update(ag, cg, int(d["steps"]))
    for m, dg, keys in ((agent.actor, d["actor_upd_digest"], d["actor_keys"]), (agent.critic, d["critic_upd_digest"], d["critic_keys"])):
        sd = m.state_dict()
        for k, ref in zip(keys, dg):
            k = str(k)
            if k.endswith(("weight_u", "weight_v")):
                continue
            mine = digest(sd[k])
            assert abs(mine[1] - ref[1]) <= 1e-4 * ref[2] + 1e-6, (k, mine[1], ref[1])
            assert np.max(np.abs(mine[4:] - ref[4:])) <= 3e-4, k
    for ep in _expected_entry_points(128, mode):
        assert forced_kernels[ep] > 0, (ep, forced_kernels)


# ---- (b) production width against the f64 oracle ---------------------------------------------------------------------------------
def _linear_variant(k):
    """F.linear with the contraction evaluated in another order (k = 0: as the library does it): the same mathematics, other fp32
    roundings -- an independent sample of which near-zero ReLU inputs fall on which side"""
    import torch.nn.functional as F
    if k == 0:
        return lambda x, sd, key: F.linear(x, sd[key + ".weight"], sd[key + ".bias"])
    if k == 1:   # the features in reverse order
        return lambda x, sd, key: F.linear(x.flip(-1), sd[key + ".weight"].flip(-1), sd[key + ".bias"])

    def split(x, sd, key):   # split-K: k - 1 cuts
        W, K = sd[key + ".weight"], x.shape[-1]
        cuts = [round(K * j / k) for j in range(k + 1)]
        y = sd[key + ".bias"]
        for a_, b_ in zip(cuts[:-1], cuts[1:]):
            y = y + F.linear(x[..., a_:b_], W[:, a_:b_])
        return y
    return split


def _oracle_train(sd_a, sd_c, batch, depth, mb, cfg, dtype, variant=0):
    """oracle/model_oracle.py train on the GPU in `dtype`: -> objC, objA, {('a'|'c', name): grad}; variant: see _linear_variant"""
    from oracle import model_oracle as mo
    a = {k: v.detach().clone().to(dtype) if v.is_floating_point() else v.clone() for k, v in sd_a.items()}
    c = {k: v.detach().clone().to(dtype) if v.is_floating_point() else v.clone() for k, v in sd_c.items()}
    b = {k: v.to(dtype) for k, v in batch.items()}
    plain = mo.linear
    mo.linear = _linear_variant(variant)
    try:
        with torch.enable_grad():
            objC, objA, ga, gc, adv, vt = mo.train(a, c, b, depth, mb, cfg.algo.gamma, cfg.algo.lamda, cfg.algo.epsilon, cfg.algo.entropy_coef)
    finally:
        mo.linear = plain
    grads = {("a", k): g for k, g in ga.items() if g is not None}
    grads.update({("c", k): g for k, g in gc.items() if g is not None and not k.startswith("shared_net.")})
    return objC, objA, grads, adv, vt


@pytest.mark.parametrize("mode", ["split_bf16", "fp32"])
def test_production_width_update_matches_the_f64_oracle(mode):
    """One cfg3-shaped batch at production width (64 episodes x 150 steps x 8 defenders, mini-batches of 32 episodes = 38 400 GRU rows /
    4 800 message rows x 176 obstacles): rolled out by the product, updated by MAPPO.train through the grouped epoch and every
    production kernel (nothing forced), and re-evaluated by the oracle's `train` in f64.  Losses within 1e-4; every gradient tensor
    within 4 x noise + 2e-5 x max|g| where noise = max |oracle fp32 - oracle f64| for that tensor over four fp32 evaluations (the tolerance
    of the reference-golden test, with the oracle's fp32 runs standing in for the reference's)."""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
    from distributed_multi_agent_reinforcement_learning_amd.mappo import BUFFER_KEYS, MAPPO
    from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
    GRAD_NOISE_FACTOR, GRAD_REL_FLOOR = 4.0, 2e-5
    N, MB = 64, 32
    modes = ops.matmul_modes()
    try:
        cfg = baseline_config("cfg3", **{"runtime.num_envs": N, "runtime.seed": 11, "runtime.matmul": mode, "algo.sample_epi_num": 1})
        torch.manual_seed(5)
        agent = MAPPO(cfg, N, MB, "Learner")
        with torch.no_grad():   # a head that is not uniform: the ratio / clip branches of the loss are all populated after a perturbation
            agent.actor.Mean.weight.mul_(20.0)
        env = Pursuit_Env(cfg, num_envs=N)
        exp_r, rb, steps = agent.explore_env(env, 1)
        T, P = cfg.env.max_steps, cfg.env.num_defender
        assert steps == N * T and T == 150 and MB * T * P >= 38400
        buf = rb.buffer
        # the update is evaluated on a policy that moved since the rollout (as in the second and later epochs): ratios leave 1, some clip
        with torch.no_grad():
            g = torch.Generator(device="cuda").manual_seed(3)
            for p in agent.ac_parameters:
                p.add_(0.02 * p.abs().mean() * torch.randn(p.shape, device=p.device, generator=g))
        sd_a = {k: v.detach().clone() for k, v in agent.actor.state_dict().items()}
        sd_c = {k: v.detach().clone() for k, v in agent.critic.state_dict().items()}
        batch = {k: buf[k].detach().clone() for k in BUFFER_KEYS}
        calls = {n: 0 for n in ENTRY_POINTS}
        L = ops.load_library()
        reals = {n: getattr(L, n) for n in ENTRY_POINTS}
        try:
            for n in ENTRY_POINTS:
                def counted(*a, _n=n):
                    calls[_n] += 1
                    return reals[_n](*a)
                setattr(L, n, counted)
            with torch.enable_grad():
                objC, objA, ag, cg = agent.train(rb, steps)
        finally:
            for n in ENTRY_POINTS:
                setattr(L, n, reals[n])
        assert agent._update_group(2, MB * P, T) == 2, "the grouped epoch was not on the path"
        for ep in _expected_entry_points(128, mode):
            assert calls[ep] > 0, (ep, calls)
        mine = {("a", n): torch.as_tensor(g) for (n, _), g in zip(agent.actor.named_parameters(), ag) if g is not None}
        mine.update({("c", n): torch.as_tensor(g) for (n, _), g in zip(agent.critic.named_parameters(), cg) if g is not None and not n.startswith("shared_net.")})
        del agent._segs
        torch.cuda.empty_cache()
        o64 = _oracle_train(sd_a, sd_c, batch, cfg.algo.depth, MB, cfg, torch.float64)
        torch.cuda.empty_cache()
        # the fp32 noise of every tensor from FOUR fp32 evaluations of the oracle whose Linear layers sum their contractions in different
        # orders: a gradient here is a sum over 3e5 rows through several ReLUs, and an activation within rounding noise of zero falls on
        # either side of it depending on the summation order -- a whole row's contribution appears or disappears.  One fp32 run samples
        # that once (and a library GEMM rounds a row the same way wherever it stands in the batch); the product's kernels (other tilings,
        # other roundings) are another sample of the same distribution.
        o32s = []
        for variant in range(4):
            o32s.append(_oracle_train(sd_a, sd_c, batch, cfg.algo.depth, MB, cfg, torch.float32, variant)[2])
            torch.cuda.empty_cache()
        assert close(agent.last_adv, o64[3].cpu().numpy(), 1e-4) and close(agent.last_v_target, o64[4].cpu().numpy(), 1e-4)
        assert abs(objC - o64[0]) <= 1e-4 * (1 + abs(o64[0])) and abs(objA - o64[1]) <= 1e-4 * (1 + abs(o64[1])), (objC, o64[0], objA, o64[1])
        assert set(mine) == set(o64[2]), set(mine) ^ set(o64[2])
        worst = (0.0, None)
        for key, ref in o64[2].items():
            ref = ref.double().cpu()
            noise = max(float((o[key].double().cpu() - ref).abs().max()) for o in o32s)
            scale = float(ref.abs().max())
            err = float((mine[key].double() - ref).abs().max())
            tol = GRAD_NOISE_FACTOR * noise + GRAD_REL_FLOOR * scale
            if err / tol > worst[0]:
                worst = (err / tol, f"{key} err {err:.3e} noise {noise:.3e} scale {scale:.3e}")
            assert err <= tol, (key, err, tol, noise, scale)
            assert scale > 0, key
        print(f"[{mode}] worst gradient error / tolerance:", worst, "calls:", calls)
    finally:
        ops.restore_matmul_modes(modes)
