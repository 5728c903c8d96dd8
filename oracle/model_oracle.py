"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

Plain PyTorch (CPU, fp32) restatement of the reference's DHGN encoder, GRU actor/critic, rollout bookkeeping
(incl. the shared-history quirk), GAE / advantage normalisation and the PPO-clip / value-clip update.  It works on
state_dicts with the reference's key names and materialises every intermediate exactly like the reference does
(no fusion, no custom kernels), so it is the checker for the HIP ops and for the product's MAPPO.

Follows (reference paths): DHGN/mappo_parallel.py:204-348 (DHGN), :400-456 (actor), :487-527 (critic),
:638-723 (train), :742-827 (run_episode), evaluator.py:106-201 (evaluate), DHGN/replay_buffer.py:24-65.

Parity status: PINNED by tests/golden/model_*.npz captured from the reference itself (rollout buffers,
mode-1 log-probs/values, GAE, losses, gradient digests, Adam step, greedy actions).
"""
import torch
import torch.nn.functional as F


def linear(x, sd, key):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def mean_operator(sd, key, message, adjacent_mat):
    """DHGN.mean_operator (:336-348): message (*,P,K,E), adjacent_mat (*,P,1,K)"""
    a = F.normalize(adjacent_mat, p=1, dim=-1)
    return F.relu(linear(torch.matmul(a, message), sd, key))


def encoder(sd, pre, p, e, o, adj_p, adj_e, adj_o, is_critic):
    """DHGN.encoder (:241-304). p (*,P,4), e (*,1,4), o (*,O,4); returns h0 (*,P,E)."""
    embs = []
    for r, (q, adj) in enumerate(((p, adj_p), (e, adj_e), (o, adj_o))):
        if is_critic:
            adj = torch.ones_like(adj)
        rel = p.unsqueeze(-2) - q.unsqueeze(-3)                      # coordinate (:235-239): (*,P,K,4)
        if r == 0:
            t2 = (p.unsqueeze(-2) - e.unsqueeze(-3)).expand(*rel.shape)
            rel = torch.cat((rel, t2), dim=-1)
        msg = F.relu(linear(rel, sd, f"{pre}MSG_layers.{r}"))           # message (:323-334)
        embs.append(mean_operator(sd, f"{pre}AGG_layers.AGG_vertex_0", msg, adj.unsqueeze(-2)))
    v = torch.cat(embs, dim=-1)                                      # (*,P,1,3E)
    v = torch.cat((p.unsqueeze(-2), v), dim=-1)
    return linear(v, sd, f"{pre}semantic_layer").squeeze(-2)


def fcra(sd, pre, h0, hist, adj_p, is_critic, depth):
    """DHGN.fcra (:204-233) with 'mean': hist[k] (*,P,E), k = 0 most recent."""
    h = h0
    adj = torch.ones_like(adj_p) if is_critic else adj_p
    for k in range(depth):
        a = F.normalize(adj, p=1, dim=-1)
        emb = F.relu(linear(torch.matmul(a, hist[k]), sd, f"{pre}AGG_layers.AGG_fcra_{k}"))
        h = F.relu(linear(torch.cat((emb, h), dim=-1), sd, f"{pre}FCRA_layers.{k}"))
    return h


def gru_module(sd, num_layers=2):
    E = sd["GRU.weight_ih_l0"].shape[1]
    H = sd["GRU.weight_hh_l0"].shape[1]
    g = torch.nn.GRU(E, H, num_layers)
    g.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("GRU.")})
    return g


def gru_apply(sd, x, h0, num_layers=2):
    """torch.nn.GRU semantics through the functional kernel so gradients reach the tensors in sd."""
    flat = []
    for layer in range(num_layers):
        flat += [sd[f"GRU.weight_ih_l{layer}"], sd[f"GRU.weight_hh_l{layer}"], sd[f"GRU.bias_ih_l{layer}"], sd[f"GRU.bias_hh_l{layer}"]]
    # (train flag: dropout is 0 either way; the GPU library's GRU only differentiates in training mode)
    out, hn = torch._VF.gru(x, h0, flat, True, num_layers, 0.0, bool(x.is_cuda), False, False)
    return out, hn


def critic_head_weight(sd, update_uv=True):
    """old-style torch.nn.utils.spectral_norm, one power iteration per training-mode forward (critic.Mean, :485).
    Returns the effective weight; u / v buffers in sd are updated in place like the hook does."""
    if "Mean.weight_orig" not in sd:
        return sd["Mean.weight"]
    w = sd["Mean.weight_orig"]
    u, v = sd["Mean.weight_u"], sd["Mean.weight_v"]
    wm = w.reshape(w.shape[0], -1)
    with torch.no_grad():
        v_new = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
        u_new = F.normalize(torch.mv(wm, v_new), dim=0, eps=1e-12)
        if update_uv:
            u.copy_(u_new); v.copy_(v_new)
    sigma = torch.dot(u_new, torch.mv(wm, v_new))
    return w / sigma


def actor_step(sd, obs, hist, h, depth):
    """SharedActor.forward mode 0 for ONE environment: obs tensors without batch dim. Returns prob (P,A), h, emb."""
    pre = "shared_net."
    h0 = encoder(sd, pre, obs["p"], obs["e"], obs["o"], obs["p_adj"], obs["e_adj"], obs["o_adj"], False)
    emb = fcra(sd, pre, h0, hist, obs["p_adj"], False, depth)
    feat, h = gru_apply(sd, emb.unsqueeze(0), h)
    prob = torch.softmax(linear(feat.squeeze(0), sd, "Mean"), dim=-1)
    return prob, h, emb


def critic_step(sd, obs, hist, h, depth):
    pre = "shared_net."
    h0 = encoder(sd, pre, obs["p"], obs["e"], obs["o"], obs["p_adj"], obs["e_adj"], obs["o_adj"], True)
    emb = fcra(sd, pre, h0, hist, obs["p_adj"], True, depth)
    feat, h = gru_apply(sd, emb.unsqueeze(0), h)
    val = F.linear(feat.squeeze(0), critic_head_weight(sd), sd["Mean.bias"])
    return val, h, emb


def sequence_forward(sd, batch, hist_key, is_critic, depth, num_layers=2):
    """mode 1 (:426-437, :516-527): batch tensors (N,T,...) incl. zero-padded obstacles; clean per-net history
    EmbeddingDataset2 (:95-113): hop k reads buffer[:, depth-1-k : depth-1-k+T]."""
    pre = "shared_net."
    N, T, P = batch["p_state"].shape[:3]
    h0 = encoder(sd, pre, batch["p_state"], batch["e_state"], batch["o_state"], batch["p_adj"], batch["e_adj"], batch["o_adj"], is_critic)
    hist = [batch[hist_key][:, depth - (k + 1): depth - (k + 1) + T] for k in range(depth)]
    emb = fcra(sd, pre, h0, hist, batch["p_adj"], is_critic, depth)
    E = emb.shape[-1]
    x = emb.permute(1, 0, 2, 3).reshape(T, N * P, E)
    H = sd["GRU.weight_hh_l0"].shape[1]
    feat, _ = gru_apply(sd, x, torch.zeros(num_layers, N * P, H, dtype=x.dtype, device=x.device))
    feat = feat.reshape(T, N, P, H).permute(1, 0, 2, 3)
    if is_critic:
        return F.linear(feat, critic_head_weight(sd), sd["Mean.bias"]).squeeze(-1)
    return torch.softmax(linear(feat, sd, "Mean"), dim=-1)


def categorical_logprob_entropy(prob, action):
    dist = torch.distributions.Categorical(prob)
    return dist.log_prob(action), dist.entropy()


def rollout_from_observations(sd_a, sd_c, buf, depth, num_layers=2, n_obs=None):
    """Re-runs the model side of MAPPO.run_episode (:742-827) on recorded observations and actions, reproducing the
    shared history list: both nets read hops from the last `depth` items of (..., a_{t-1}, c_{t-1}).
    buf: dict of (N,T,...) tensors (reference replay-buffer layout); n_obs[n] = real obstacle count (the rollout sees
    only real obstacle rows).  Returns a_logprob (N,T,P), v (N,T+1,P), actor/critic embeddings (N,T,P,E)."""
    N, T, P = buf["p_state"].shape[:3]
    E = sd_a["shared_net.semantic_layer.weight"].shape[0]
    H = sd_a["GRU.weight_hh_l0"].shape[1]
    logp = torch.zeros(N, T, P); v = torch.zeros(N, T + 1, P)
    ea = torch.zeros(N, T, P, E); ec = torch.zeros(N, T, P, E)
    for n in range(N):
        k = int(n_obs[n])
        ha = torch.zeros(num_layers, P, H); hc = torch.zeros(num_layers, P, H)
        shared = [torch.zeros(P, E) for _ in range(depth)]
        a_cur = torch.zeros(P, E); c_cur = torch.zeros(P, E)
        for t in range(T):
            obs = dict(p=buf["p_state"][n, t], e=buf["e_state"][n, t], o=buf["o_state"][n, t, :k], p_adj=buf["p_adj"][n, t],
                       e_adj=buf["e_adj"][n, t], o_adj=buf["o_adj"][n, t, :, :k])
            shared = (shared + [a_cur, c_cur])[len(shared + [a_cur, c_cur]) - depth:] if depth else []
            hops = [shared[depth - 1 - j] for j in range(depth)]
            prob, ha, a_cur = actor_step(sd_a, obs, hops, ha, depth)
            val, hc, c_cur = critic_step(sd_c, obs, hops, hc, depth)
            lp, _ = categorical_logprob_entropy(prob, buf["a_n"][n, t].long())
            logp[n, t] = lp; v[n, t] = val.flatten(); ea[n, t] = a_cur; ec[n, t] = c_cur
        # bootstrap value is computed by the caller (needs the post-episode observation)
    return logp, v, ea, ec


def gae(r, v, active, gamma, lamda, use_adv_norm=True):
    """:643-658"""
    deltas = (r + gamma * v[:, 1:] - v[:, :-1]) * active
    adv = torch.zeros_like(r)
    g = torch.zeros_like(r[:, 0])
    for t in reversed(range(r.shape[1])):
        g = deltas[:, t] + gamma * lamda * g
        adv[:, t] = g
    v_target = adv + v[:, :-1]
    if use_adv_norm:
        adv = (adv - adv.mean()) / (adv.std() + 1e-5) * active
    return adv, v_target


def ppo_losses(logp_now, entropy, values_now, batch, adv, v_target, epsilon, entropy_coef, use_value_clip=True):
    """:692-706 for one mini-batch (tensors already indexed)"""
    ratios = torch.exp(logp_now - batch["a_logprob_n"])
    surr1 = ratios * adv
    surr2 = torch.clamp(ratios, 1 - epsilon, 1 + epsilon) * adv
    actor_loss = -torch.min(surr1, surr2) - entropy_coef * entropy
    actor_loss = (actor_loss * batch["active"]).sum() / batch["active"].sum()
    if use_value_clip:
        v_old = batch["v_n"][:, :-1]
        clip_err = torch.clamp(values_now - v_old, -epsilon, epsilon) + v_old - v_target
        orig_err = values_now - v_target
        critic_loss = torch.max(clip_err ** 2, orig_err ** 2)
    else:
        critic_loss = (values_now - v_target) ** 2
    critic_loss = (critic_loss * batch["active"]).sum() / batch["active"].sum()
    return actor_loss, critic_loss


def train(sd_a, sd_c, batch, depth, mini_batch_size, gamma, lamda, epsilon, entropy_coef, clip=5.0, adv_override=None, keep_grads=False):
    """MAPPO.train (:638-723) on tensors that share storage for shared_net.* between sd_a and sd_c.
    Gradients accumulate over mini-batches with clip_grad_norm_ after every backward.
    Returns objC, objA, {name: grad} for the actor keys and for the critic keys, adv, v_target.
    keep_grads: the tensors' existing .grad storage is accumulated into instead of being dropped first (the caller zeroed it; used by
    the data-parallel test, whose gradients are views of one flat bucket)."""
    for k in sd_c:
        if k.startswith("shared_net."):
            sd_c[k] = sd_a[k]  # one encoder object serves both nets (MAPPO.__init__, :582-616)
    params = {}
    for k, t in sd_a.items():
        params[("a", k)] = t
    for k, t in sd_c.items():
        if k.startswith("shared_net."):
            continue
        params[("c", k)] = t
    leaf = [t for (who, k), t in params.items() if t.is_floating_point() and not k.endswith(("weight_u", "weight_v"))]
    for t in leaf:
        t.requires_grad_(True)
        if not keep_grads:
            t.grad = None
    with torch.no_grad():
        adv, v_target = gae(batch["r"], batch["v_n"], batch["active"], gamma, lamda)
    if adv_override is not None:  # (adv, v_target) computed elsewhere, e.g. per data-parallel shard (main.py:105-129)
        adv, v_target = adv_override
    N = batch["r"].shape[0]
    objC = objA = 0.0
    n_upd = 0
    for s in range(0, N, mini_batch_size):
        idx = slice(s, min(s + mini_batch_size, N))
        mb = {k: v[idx] for k, v in batch.items()}
        prob = sequence_forward(sd_a, mb, "actor_historical_embedding", False, depth)
        logp, ent = categorical_logprob_entropy(prob, mb["a_n"])
        vals = sequence_forward(sd_c, mb, "critic_historical_embedding", True, depth)
        la, lc = ppo_losses(logp, ent, vals, mb, adv[idx], v_target[idx], epsilon, entropy_coef)
        (la + lc).backward()
        torch.nn.utils.clip_grad_norm_(leaf, clip)
        objC += lc.item(); objA += la.item(); n_upd += 1
    ga = {k: sd_a[k].grad for k in sd_a if sd_a[k].requires_grad}
    gc = {k: sd_c[k].grad for k in sd_c if sd_c[k].requires_grad}
    return objC / n_upd, objA / n_upd, ga, gc, adv, v_target
