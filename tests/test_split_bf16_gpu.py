"""Adversarial pins of the split-bf16 arithmetic (`runtime.matmul: split_bf16`, the default that carries the benchmark line).

The kernels evaluate fp32 products from exact three-way bf16 splits of their operands (csrc/mappo_ops.hip sb_split2 / sb_mma6).  The
tests of tests/test_ops_gpu.py draw well-conditioned `randn` operands; here
  (a) the split itself is checked bit for bit over the whole fp32 exponent range (the domain statement of the kernels as a test), and
  (b) every split kernel of the product (sb_gemm, wgrad_split_tn, gru_cell_split_fwd_multi) runs on operands that stress what the
      construction drops (a2 b3 + a3 b2 + a3 b3) and the matrix unit's internal summation: wide dynamic range inside one dot product,
      cancelling dot products, one huge element -- against f64, with the fp32-MFMA / BLAS fp32 route's error beside it.
Reference ops these kernels stand for: nn.Linear / nn.GRU of DHGN/mappo_parallel.py:284-303, :397, :432-436 and their autograd.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RECORD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_split_adversarial.jsonl")


def _record(**row):
    """one JSON line per measurement (copied to profiles/ and DESIGN 3.7 by hand); silent when gpurun_out/ is absent"""
    try:
        os.makedirs(os.path.dirname(RECORD), exist_ok=True)
        with open(RECORD, "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass


def _pieces(x):
    from distributed_multi_agent_reinforcement_learning_amd import ops
    L = ops.load_library()
    n = x.numel()
    out = torch.full((3, n), float("nan"), dtype=torch.float32, device="cuda")
    ops._check(L.sb_split_diag(n, ops._ptr(x), ops._ptr(out), ops._stream()), "sb_split_diag")
    torch.cuda.synchronize()
    return out


def _from_bits(bits):
    return torch.from_numpy(np.asarray(bits, dtype=np.uint32).view(np.float32).copy()).cuda()


MAX_EXACT_BITS = 0x7F7F7FFF     # the largest fp32 whose first piece bf16(x) is finite (0x7F7F8000 rounds to infinity)
EXACT_FROM = 2.0 ** -110        # from here up every one of x's 24 significant bits sits at or above bf16's smallest subnormal, 2^-133


def test_split_bf16_pieces_sum_exactly():
    """p1 + p2 + p3 == x BIT FOR BIT for every finite fp32 from 2^-110 up to 0x7F7F7FFF (3.3895e38), both signs: all 254 normal
    exponents x (64 random + 12 tie / carry mantissas).  Below 2^-110 the low pieces fall under bf16's subnormal grid (2^-133) and
    the sum is x to within 2^-134 absolute -- 1e-40: no activation, weight or gradient of this model is within 25 orders of magnitude
    of it -- (subnormal fp32 inputs included); +-0 split into zeros.  Each piece is a bf16 and the pieces do not overlap."""
    rng = np.random.default_rng(0)
    exps = np.arange(1, 255, dtype=np.uint32)
    special = np.array([0, 1, 0x7FFFFF, 0x7F8000, 0x7F7FFF, 0x008000, 0x018000, 0x000080, 0x000180, 0x7FFF80, 0x400000, 0x3FFFFF], np.uint32)
    man = np.concatenate([rng.integers(0, 1 << 23, (254, 64), dtype=np.uint32), np.broadcast_to(special, (254, special.size))], axis=1)
    bits = ((exps[:, None] << 23) | man).ravel()
    bits = bits[bits <= MAX_EXACT_BITS]
    sub = np.concatenate([rng.integers(1, 1 << 23, 4096, dtype=np.uint32), np.array([1, 2, 3, 0x7FFFFF, 0x400000, 0x8000, 0x10000], np.uint32)])
    bits = np.concatenate([bits, sub, np.array([0, MAX_EXACT_BITS], np.uint32)])
    bits = np.concatenate([bits, bits | np.uint32(0x80000000)])
    if bits.size % 2:
        bits = bits[:-1]
    x = _from_bits(bits)
    p = _pieces(x).cpu().numpy().astype(np.float64)     # widened on the host: nothing on the way may flush a subnormal
    xd = bits.view(np.float32).astype(np.float64)
    assert np.isfinite(p).all()
    s = p[0] + p[1] + p[2]                      # exact in f64: three bf16 numbers within 2^-24 of each other's magnitude
    big = np.abs(xd) >= EXACT_FROM
    assert big.sum() > 30000 and (~big).sum() > 8000
    bad = np.flatnonzero(big & (s != xd))
    assert bad.size == 0, ("not exact", bits[bad[:5]], xd[bad[:5]], s[bad[:5]])
    assert np.max(np.abs(s - xd)[~big]) <= 2.0 ** -134
    # pieces are bf16 numbers (8 significant bits) ...
    pb = np.ascontiguousarray(p.astype(np.float32)).view(np.uint32)
    assert not (pb & 0xFFFF).any()
    # ... that do not overlap: |p2| <= ulp_bf16(p1) / 2, |p3| <= ulp_bf16(p2) / 2  (ulp of an 8-bit significand = 2^-7 of its binade)
    with np.errstate(divide="ignore"):
        for hi, lo in ((p[0], p[1]), (p[1], p[2])):
            nz = (hi != 0) & big
            binade = 2.0 ** np.floor(np.log2(np.abs(hi[nz])))
            assert (np.abs(lo[nz]) <= binade * 2.0 ** -8).all()
    zeros = xd == 0
    assert zeros.sum() >= 2 and not p[:, zeros].any()
    _record(test="pieces_sum", exact_inputs=int(big.sum()), tiny_inputs=int((~big).sum()), tiny_max_abs_err=float(np.max(np.abs(s - xd)[~big])))


def test_split_bf16_domain_edge_is_documented_behaviour():
    """Just beyond the domain: the 32 768 fp32 values in [0x7F7F8000, 0x7F7FFFFF] (3.3961e38 .. 3.4028e38) round to a first piece of
    infinity, and the remaining pieces are not numbers (inf - inf); infinities and NaNs pass through as non-finite pieces.  The product
    never produces such operands (activations, weights and gradients are < 1e6); a caller who does gets inf / NaN results, not a
    silently wrong finite number."""
    bits = np.array([0x7F7F8000, 0x7F7FFFFF, 0xFF7F8000, 0xFF7FFFFF, 0x7F800000, 0xFF800000, 0x7FC00000, 0x7F7F7FFF], np.uint32)
    p = _pieces(_from_bits(bits)).cpu().numpy()
    assert np.isposinf(p[0, 0]) and np.isposinf(p[0, 1]) and np.isneginf(p[0, 2]) and np.isneginf(p[0, 3])
    assert np.isnan(p[2, :4]).all()
    assert np.isposinf(p[0, 4]) and np.isneginf(p[0, 5]) and np.isnan(p[0, 6])
    assert np.isfinite(p[:, 7]).all() and p[:, 7].astype(np.float64).sum() == float(bits[7:8].view(np.float32)[0])


# ---- (b) the kernels on adversarial operands ------------------------------------------------------------------------------------
def _pow2(shape, lo, hi, gen):
    return torch.exp2(torch.randint(lo, hi + 1, shape, generator=gen, device="cuda").float())


def _operands(case, R, K, N, gen):
    """x (R, K), w (N, K) whose dot products x[r] . w[n] are hard in the way `case` names"""
    x = torch.randn(R, K, device="cuda", generator=gen)
    w = torch.randn(N, K, device="cuda", generator=gen)
    if case == "randn":
        return x, w
    if case == "wide":                 # magnitudes spanning 2^-20 .. 2^20 on both sides: terms over 2^+-40 within one dot product
        return x * _pow2((R, K), -20, 20, gen), w * _pow2((N, K), -20, 20, gen)
    if case == "cancel":               # the second half of every dot product cancels the first to ~1e-4 of the terms' size
        h = K // 2
        x[:, h:2 * h] = -x[:, :h] + 1e-4 * torch.randn(R, h, device="cuda", generator=gen)
        w[:, h:2 * h] = w[:, :h]
        return x, w
    if case == "huge":                 # one element of every row is 2^100 (1.3e30) times the others
        idx = torch.randint(0, K, (R,), generator=gen, device="cuda")
        sign = torch.where(torch.rand(R, device="cuda", generator=gen) < 0.5, -1.0, 1.0)
        x[torch.arange(R, device="cuda"), idx] = sign * 2.0 ** 100 * (1.0 + torch.rand(R, device="cuda", generator=gen))
        return x, w
    raise ValueError(case)


def _norm_err(y, ref, xabs_wabs):
    """max |y - ref| / sum_k |x_k w_k|: the error in units of the dot product's own scale (what a K-term fp32 sum can lose)"""
    return float(((y.double() - ref).abs() / xabs_wabs.clamp_min(1e-300)).max())


CASES = ["randn", "wide", "cancel", "huge"]
ULP = 2.0 ** -23   # one fp32 unit in the last place, in the normalised unit above: the allowance for a route that happens to be exact


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("N,K", [(128, 128), (128, 256), (128, 384), (256, 128), (384, 128)])
def test_split_linear_on_adversarial_operands(case, N, K):
    """sb_gemm (every (outputs, inputs) variant) against f64, beside the BLAS library's fp32 GEMM the fp32 mode uses for the same layer"""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(1000 * N + K + len(case))
    R = 8192 + 37
    x, w = _operands(case, R, K, N, gen)
    ref = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    old = ops.CELL_MODE
    ops.set_cell_mode("split_bf16")
    try:
        with torch.no_grad():
            assert ops.split_linear_ok(x, w)
            y = ops.split_linear(x, w)
    finally:
        ops.set_cell_mode(old)
    lib = x @ w.t()
    e_split, e_lib = _norm_err(y, ref, scale), _norm_err(lib, ref, scale)
    _record(test="sb_gemm", case=case, N=N, K=K, rows=R, err_split=e_split, err_fp32_library=e_lib)
    assert torch.isfinite(y).all()
    assert e_split <= 2.0 * e_lib + ULP, (case, e_split, e_lib)
    assert e_split <= K * 2.0 ** -24, (case, e_split)          # the textbook bound of a K-term fp32 dot product, whatever the library does


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("M,N", [(128, 128), (384, 128), (128, 384), (128, 256), (256, 128)])
def test_split_wgrad_on_adversarial_operands(case, M, N):
    """wgrad_split_tn (k_sb_wgrad) against f64, beside the fp32-MFMA kernel (wgrad_tn): dot products over the ROW axis (20 011 rows)"""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(77 * M + N + len(case))
    rows = 20011
    at, bt = _operands(case, M, rows, N, gen)       # (M, rows), (N, rows): the dot products run along the second axis
    a, b = at.t().contiguous(), bt.t().contiguous()  # the kernels' (rows, M), (rows, N) operands
    ref = at.double() @ bt.double().t()
    scale = at.double().abs() @ bt.double().abs().t()
    old = ops.WGRAD_MODE
    errs = {}
    try:
        for mode in ("fp32", "split_bf16"):
            ops.WGRAD_MODE = mode
            c = ops.wgrad(a, b)
            assert torch.isfinite(c).all()
            errs[mode] = _norm_err(c, ref, scale)
    finally:
        ops.WGRAD_MODE = old
    _record(test="wgrad", case=case, M=M, N=N, rows=rows, err_split=errs["split_bf16"], err_fp32_mfma=errs["fp32"])
    assert errs["split_bf16"] <= 2.0 * errs["fp32"] + ULP, (case, errs)


@pytest.mark.parametrize("case", CASES)
def test_split_cell_on_adversarial_operands(case):
    """gru_cell_split_fwd_multi (k_gru_cell_sb) against an f64 torch.nn.GRU step, beside the fp32-MFMA cell (k_gru_cell): input and
    state rows with a wide dynamic range, cancelling projections, one huge element (the gates saturate: both kernels must agree with
    f64 on WHICH way)."""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(31 + len(case))
    B, H = 4096 + 19, 128
    torch.manual_seed(9)
    mods = [torch.nn.GRU(H, H, 1).cuda() for _ in range(2)]
    xs, hs = [], []
    for m in mods:
        x, wi = _operands(case, B, H, 3 * H, gen)
        h, wh = _operands(case, B, H, 3 * H, gen)
        if case == "wide":   # keep the pre-activations O(1) so the gates stay sensitive: the largest of a row's terms is ~2^38
            wi, wh = wi * 2.0 ** -38, wh * 2.0 ** -38
        else:
            wi, wh = wi * 0.1, wh * 0.1
        if case == "cancel":
            h = h.clamp(-1, 1)
        with torch.no_grad():
            m.weight_ih_l0.copy_(wi); m.weight_hh_l0.copy_(wh)
        xs.append(x); hs.append(h.unsqueeze(0).contiguous())
    with torch.no_grad():
        ref = []
        for x, h, m in zip(xs, hs, mods):
            m64 = torch.nn.GRU(H, H, 1).cuda().double()
            m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
            ref.append(m64(x.double().unsqueeze(0), h.double())[1])
        old = ops.CELL_MODE
        errs = {}
        try:
            for mode in ("fp32", "split_bf16"):
                ops.set_cell_mode(mode)
                outs = [torch.full_like(h, float("nan")) for h in hs]
                ops.gru_step_multi(xs, hs, mods, hiddens_out=outs)
                assert all(torch.isfinite(o).all() for o in outs)
                # error relative to the state's own scale (the huge case carries h' = z h with |h| up to 2^101)
                errs[mode] = max(float(((o.double() - r).abs() / r.abs().clamp_min(1.0)).max()) for o, r in zip(outs, ref))
        finally:
            ops.set_cell_mode(old)
    _record(test="gru_cell", case=case, rows=B, err_split=errs["split_bf16"], err_fp32_mfma=errs["fp32"])
    assert errs["split_bf16"] <= 2.0 * errs["fp32"] + 2e-7, (case, errs)


def _gru_seq_run(mode_fwd, mode_bwd, mod, x, h0, gout, agents, T):
    """ops.gru on the persistent recurrences with SEQ_MODE = mode_fwd for the forward and mode_bwd for the backward launch"""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    old = ops.SEQ_MODE
    try:
        xs = x.clone().requires_grad_(True)
        hs = h0.clone().requires_grad_(True)
        mod.zero_grad()
        ops.SEQ_MODE = mode_fwd
        out, _ = ops.gru(xs, hs, mod, agents=agents, steps=T) if agents else ops.gru(xs, hs, mod)
        ops.SEQ_MODE = mode_bwd
        (out * gout).sum().backward()
    finally:
        ops.SEQ_MODE = old
    return [out.detach().double(), xs.grad.double(), hs.grad.double()] + [p.grad.double().clone() for p in mod.parameters()]


@pytest.mark.parametrize("T,n,P,agents,case", [(150, 26, 8, True, "randn"), (37, 5, 8, False, "randn"), (12, 3, 4, True, "randn"), (150, 26, 8, True, "cancel"),
                                               (60, 9, 8, False, "wide")])
def test_split_gru_sequence_matches_f64_beside_the_fp32_kernels(T, n, P, agents, case):
    """The update's persistent GRU recurrences on the split-bf16 kernels (csrc/sb_gru_seq.hpp k_gru_seq_fwd_sb / k_gru_seq_bwd_sb) against
    an f64 torch.nn.GRU, with the fp32-MFMA kernels (k_gru_seq_fwd2 / bwd2) beside them: outputs and every gradient (input, initial
    state, W_ih, W_hh, both biases), two layers, the encoder's row order and the time-major one, sequence counts that leave a ragged last
    tile; a recurrent weight whose products cancel and one with a wide dynamic range.  The saved gates of the two routes are
    interchangeable: forward on one, backward on the other gives the same numbers."""
    from distributed_multi_agent_reinforcement_learning_amd import ops
    torch.manual_seed(T + n)
    B, E = n * P, 128
    mod = torch.nn.GRU(E, E, 2).cuda()
    gen = torch.Generator(device="cuda").manual_seed(5)
    if case == "cancel":      # the second half of every recurrent dot product nearly cancels the first
        with torch.no_grad():
            for w in (mod.weight_hh_l0, mod.weight_hh_l1):
                w[:, 64:] = -w[:, :64] + 1e-4 * torch.randn(384, 64, device="cuda", generator=gen)
    if case == "wide":
        with torch.no_grad():
            for w in (mod.weight_hh_l0, mod.weight_hh_l1):
                w.mul_(torch.exp2(torch.randint(-12, 3, w.shape, device="cuda", generator=gen).float()))
    x = torch.randn(n * T * P, E, device="cuda", generator=gen) if agents else torch.randn(T, B, E, device="cuda", generator=gen)
    h0 = torch.randn(2, B, E, device="cuda", generator=gen) * 0.5
    if case == "cancel":
        h0[:, :, 64:] = h0[:, :, :64]
    gout = torch.randn(T, B, E, device="cuda", generator=gen)
    m64 = torch.nn.GRU(E, E, 2).cuda().double()
    m64.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    h64 = h0.double().requires_grad_(True)
    xin = x64.reshape(n, T, P, E).permute(1, 0, 2, 3).reshape(T, B, E) if agents else x64
    o64, _ = m64(xin, h64)
    (o64 * gout.double()).sum().backward()
    ref = [o64.detach(), x64.grad, h64.grad] + [p.grad for p in m64.parameters()]
    names = ["out", "dx", "dh0"] + [k for k, _ in mod.named_parameters()]
    ag = P if agents else 0
    res = {k: _gru_seq_run(a, b, mod, x, h0, gout, ag, T) for k, (a, b) in
           dict(split=("split_bf16", "split_bf16"), fp32=("fp32", "fp32"), mixed1=("split_bf16", "fp32"), mixed2=("fp32", "split_bf16")).items()}
    worst = {}
    for k, r in res.items():
        for nm, a, b in zip(names, r, ref):
            scale = float(b.abs().max())
            worst[(k, nm)] = float((a - b).abs().max()) / max(scale, 1e-30)
    for nm in names:
        e_s, e_f = worst[("split", nm)], worst[("fp32", nm)]
        _record(test="gru_seq", case=case, T=T, B=B, agents=ag, tensor=nm, err_split=e_s, err_fp32_mfma=e_f)
        assert e_s <= 2.0 * e_f + 2e-6, (nm, e_s, e_f)
        assert e_s < 2e-5, (nm, e_s)
        for k in ("mixed1", "mixed2"):
            assert worst[(k, nm)] <= 2.0 * max(e_s, e_f) + 2e-6, (k, nm, worst[(k, nm)], e_s, e_f)
    assert worst[("split", "out")] < 2e-6      # within 1e-6-ish of f64 like the fp32 kernels (test_gru_multi_* pin bit-identity of launch forms)
