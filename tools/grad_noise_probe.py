"""Which kernel family moves a gradient tensor's error against the f64 oracle: the production-width update of
tests/test_update_parity_gpu.py with the matmul switches set one by one (MAPPO_PROBE_MODES="seq=fp32,wgrad=split_bf16,...")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd import ops
from distributed_multi_agent_reinforcement_learning_amd.config import baseline_config
from distributed_multi_agent_reinforcement_learning_amd.mappo import BUFFER_KEYS, MAPPO
from distributed_multi_agent_reinforcement_learning_amd.pursuit_env import Pursuit_Env
from tests.test_update_parity_gpu import _oracle_train

N, MB = 64, 32
cfg = baseline_config("cfg3", **{"runtime.num_envs": N, "runtime.seed": 11, "algo.sample_epi_num": 1})
torch.manual_seed(5)
agent = MAPPO(cfg, N, MB, "Learner")
with torch.no_grad():
    agent.actor.Mean.weight.mul_(20.0)
env = Pursuit_Env(cfg, num_envs=N)
exp_r, rb, steps = agent.explore_env(env, 1)
with torch.no_grad():
    g = torch.Generator(device="cuda").manual_seed(3)
    for p in agent.ac_parameters:
        p.add_(0.02 * p.abs().mean() * torch.randn(p.shape, device=p.device, generator=g))
sd_a = {k: v.detach().clone() for k, v in agent.actor.state_dict().items()}
sd_c = {k: v.detach().clone() for k, v in agent.critic.state_dict().items()}
batch = {k: rb.buffer[k].detach().clone() for k in BUFFER_KEYS}
u0, v0 = agent.critic.Mean.weight_u.clone(), agent.critic.Mean.weight_v.clone()
o64 = _oracle_train(sd_a, sd_c, batch, cfg.algo.depth, MB, cfg, torch.float64)
torch.cuda.empty_cache()
o32s = []
for variant in range(4):
    o32s.append(_oracle_train(sd_a, sd_c, batch, cfg.algo.depth, MB, cfg, torch.float32, variant)[2])
    torch.cuda.empty_cache()
combos = [dict(), dict(seq="fp32"), dict(wgrad="fp32"), dict(proj="fp32"), dict(seq="fp32", wgrad="fp32", proj="fp32")]
for combo in combos:
    ops.set_matmul_mode("split_bf16")
    if "seq" in combo: ops.SEQ_MODE = combo["seq"]
    if "wgrad" in combo: ops.WGRAD_MODE = combo["wgrad"]
    if "proj" in combo: ops.PROJ_MODE = combo["proj"]
    agent.critic.Mean.weight_u.copy_(u0); agent.critic.Mean.weight_v.copy_(v0)
    with torch.enable_grad():
        objC, objA, ag, cg = agent.train(rb, steps)
    mine = {("a", n): torch.as_tensor(g_) for (n, _), g_ in zip(agent.actor.named_parameters(), ag) if g_ is not None}
    mine.update({("c", n): torch.as_tensor(g_) for (n, _), g_ in zip(agent.critic.named_parameters(), cg) if g_ is not None and not n.startswith("shared_net.")})
    rows = []
    for key, ref in o64[2].items():
        ref = ref.double().cpu()
        noise = max(float((o[key].double().cpu() - ref).abs().max()) for o in o32s); scale = float(ref.abs().max())
        err = float((mine[key].double() - ref).abs().max())
        rows.append((err / (4 * noise + 2e-5 * scale), key, err, noise, scale))
    rows.sort(reverse=True)
    print("modes", combo or "all split", " worst:", "; ".join(f"{k[0]}:{k[1]} r={r:.2f} err={e:.2e} noise={n:.1e} scale={s:.1e}" for r, k, e, n, s in rows[:4]), flush=True)
