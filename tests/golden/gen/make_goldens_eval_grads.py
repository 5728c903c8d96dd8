"""Golden-vector generator, round 2: full-length greedy evaluations and full gradient tensors.

Runs the REAL reference (evaluator.evaluate, MAPPO.train) in the build container on the weights / buffers the model
fixtures already hold (tests/golden/model_*.npz) and stores
  * eval150_<fixture>.npz  -- T = 150 greedy episodes (evaluator.py:106-201) for several environment seeds: initial
    conditions, target tape, the (T, P) action matrix as int8, the return.  The actor is the fixture's initial actor with
    the fixture's sharpening (argmax is degenerate on fresh orthogonal weights).
  * grads_<fixture>.npz -- every gradient tensor of MAPPO.train (DHGN/mappo_parallel.py:660-723) for actor and critic on
    the fixture's buffer, plus per-tensor tolerances derived from an fp64 run of oracle/model_oracle.py on the same
    buffer: tol = max|reference fp32 - oracle fp64| is the reference's OWN fp32 noise floor for that tensor.
Usage:  python tests/golden/gen/make_goldens_eval_grads.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refload  # noqa: E402

refload.activate()
OUT = os.path.dirname(HERE)
from tests.helpers import buffer_tensors, load_model_golden  # noqa: E402


def ref_agent(d, T, agent_type="Learner"):
    from DHGN.mappo_parallel import MAPPO
    cfg = refload.load_cfg(num_defender=d["P"], map_size=(d["W"], d["H"]), blocks=d["blocks"], variance=d["variance"],
                           depth=d["depth"], max_steps=T)
    cfg.algo.sample_epi_num = d["n_epi"]
    cfg.algo.max_train_steps = 100000
    cfg.algo.embedding_dim = d["E"]
    cfg.algo.rnn_hidden_dim = d["E"]
    refload.seed_all(d["seed"])
    agent = MAPPO(cfg, d["n_epi"], d["mb"], agent_type)
    sd_a = {k[len("w_actor_"):]: torch.as_tensor(v) for k, v in d.items() if k.startswith("w_actor_")}
    agent.actor.load_state_dict(sd_a)
    sd_c = {k[len("w_critic_"):]: torch.as_tensor(v) for k, v in d.items() if k.startswith("w_critic_")}
    agent.critic.load_state_dict(sd_c, strict=False)
    assert agent.actor.shared_net is agent.critic.shared_net
    with torch.no_grad():
        params = dict(agent.actor.named_parameters())
        for k, f in zip(d["sharpen_keys"], d["sharpen_factors"]):
            params[str(k)].mul_(float(f))
    return cfg, agent


def capture_eval(name, seeds, T=150):
    from environment.pursuit_evasion_game.pursuit_env import Pursuit_Env
    import evaluator as ref_eval
    d = load_model_golden(name)
    cfg, agent = ref_agent(d, T, "Evaluator")
    O = cfg.map.num_max_obstacle
    out = dict(meta=np.asarray([T, len(seeds)], np.int64), seeds=np.asarray(seeds, np.int64))
    torch.set_grad_enabled(False)
    for s in seeds:
        refload.seed_all(s)
        env = Pursuit_Env(cfg)
        drawn, actions_log, init = [], [], {}
        orig_reset, orig_init_target, orig_step = env.reset, env.init_target, env.step

        def logged_init_target(inflated_map):
            orig_init_target(inflated_map=inflated_map)
            drawn.append(tuple(env.target[0]))

        def logged_reset():
            orig_reset()
            init.update(grid=np.asarray(env.occupied_map.grid_map, np.uint8),
                        obs_xy=np.asarray(env.boundary_map.obstacles, np.int32).reshape(-1, 2),
                        target=np.asarray(env.target[0], np.int32),
                        defenders=np.asarray(env.get_state('defender'), np.float64),
                        evader=np.asarray(env.get_state('attacker')[0], np.float64))

        def logged_step(a):
            actions_log.append(np.asarray(a, np.int64).copy())
            return orig_step(a)
        env.reset, env.init_target, env.step = logged_reset, logged_init_target, logged_step
        R, last = ref_eval.evaluate(env, agent.actor, cfg)
        k = len(init["obs_xy"])
        assert k <= O and len(drawn) - 1 <= 16
        eo = np.zeros((O, 2), np.int32); eo[:k] = init["obs_xy"]
        tape = np.zeros((16, 2), np.int32)
        for j, tgt in enumerate(drawn[1:]):
            tape[j] = tgt
        acts = np.stack(actions_log)
        assert acts.shape == (T, d["P"]) and acts.max() < 9
        pre = f"s{s}_"
        out[pre + "grid"] = init["grid"]; out[pre + "obs_xy"] = eo; out[pre + "n_obs"] = np.int32(k)
        out[pre + "target"] = init["target"]; out[pre + "defenders"] = init["defenders"]; out[pre + "evader"] = init["evader"]
        out[pre + "tape"] = tape; out[pre + "tape_n"] = np.int32(len(drawn) - 1)
        out[pre + "actions"] = acts.astype(np.int8); out[pre + "return"] = np.float64(R); out[pre + "last_index"] = np.int64(last)
        print(name, "seed", s, "return", float(R), "action histogram", np.bincount(acts.ravel(), minlength=9), flush=True)
    torch.set_grad_enabled(True)
    np.savez_compressed(os.path.join(OUT, f"eval150_{name}.npz"), **out)


def capture_grads(name):
    from DHGN.replay_buffer import BigBuffer, ReplayBuffer
    from oracle import model_oracle as mo
    d = load_model_golden(name)
    cfg, agent = ref_agent(d, d["T"], "Learner")
    bt = buffer_tensors(d)
    rb = ReplayBuffer(cfg)
    rb.reset_buffer()
    for k in bt:
        assert rb.buffer[k].shape == bt[k].shape, (k, rb.buffer[k].shape, bt[k].shape)
        rb.buffer[k] = bt[k].clone()
    big = BigBuffer(); big.concat_buffer(rb)
    sd_a0 = {k: v.detach().clone() for k, v in agent.actor.state_dict().items()}
    sd_c0 = {k: v.detach().clone() for k, v in agent.critic.state_dict().items()}
    objC, objA, ag, cg = agent.train(big, int(d["steps"]))
    assert abs(objC - float(d["objC"])) < 1e-9 and abs(objA - float(d["objA"])) < 1e-9, (objC, d["objC"], objA, d["objA"])
    names_a = [n for n, _ in agent.actor.named_parameters()]
    names_c = [n for n, _ in agent.critic.named_parameters()]
    # fp64 run of the oracle on the same buffer: the distance of the reference's fp32 gradients from it is the noise floor
    sd_a = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd_a0.items()}
    sd_c = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd_c0.items()}
    bt64 = {k: v.double() for k, v in bt.items()}
    bt64["a_n"] = bt["a_n"].clone()
    o_objC, o_objA, ga64, gc64, _, _ = mo.train(sd_a, sd_c, bt64, d["depth"], d["mb"], cfg.algo.gamma, cfg.algo.lamda, cfg.algo.epsilon,
                                                cfg.algo.entropy_coef)
    out = dict(objC=np.float64(objC), objA=np.float64(objA), objC_fp64=np.float64(o_objC), objA_fp64=np.float64(o_objA),
               actor_names=np.asarray(names_a), critic_names=np.asarray(names_c))
    worst = 0.0
    for who, names, grads, g64 in (("a", names_a, ag, ga64), ("c", names_c, cg, gc64)):
        for n, g in zip(names, grads):
            g = np.asarray(g, np.float32)
            ref64 = g64[n].detach().numpy()
            noise = float(np.max(np.abs(g.astype(np.float64) - ref64)))
            scale = float(np.max(np.abs(ref64)))
            out[f"{who}grad_{n}"] = g
            out[f"{who}noise_{n}"] = np.float64(noise)
            out[f"{who}scale_{n}"] = np.float64(scale)
            worst = max(worst, noise / max(scale, 1e-30))
            print(f"{name} {who} {n:45s} max|g| {scale:.3e}  |ref32 - oracle64| {noise:.3e}  rel {noise / max(scale, 1e-30):.2e}", flush=True)
    print(name, "objC ref", objC, "fp64", o_objC, "objA ref", objA, "fp64", o_objA, "worst relative fp32 noise", worst, flush=True)
    np.savez_compressed(os.path.join(OUT, f"grads_{name}.npz"), **out)


def main():
    which = sys.argv[1:] or ["grads", "eval"]
    if "grads" in which:
        capture_grads("model_p4_20x20_d1")
        capture_grads("model_p8_40x40_d3")
    if "eval" in which:
        capture_eval("model_p4_20x20_d1", seeds=(101, 102, 103))
        capture_eval("model_p8_40x40_d3", seeds=(201, 202, 203))


if __name__ == "__main__":
    main()
