"""ctypes binding of libmappo_ops.so (include/mappo_ops.h) with autograd glue.

The ops run only on the GPU through the HIP library; on a CPU tensor or without the built library they raise
(there is deliberately no eager fallback).
"""
import ctypes as C
import os

import torch
import torch.nn.functional as F

from . import build as _build

ADJ_TENSOR, ADJ_ONES, ADJ_VALID, ADJ_BITS = 0, 1, 2, 3
EXPORTS = ("dhgn_msg_agg_fwd", "dhgn_msg_agg3_fwd", "dhgn_msg_agg_bwd", "dhgn_msg_agg_bwd_workspace", "dhgn_msg_agg_ones_sorted_ok", "dhgn_msg_agg_ones_sorted_fwd",
           "dhgn_msg_agg_ones_sorted_bwd", "dhgn_msg_agg_ones_sorted_workspace", "gae_advnorm", "gae_advnorm_workspace", "categorical_sample",
           "categorical_sample_counter",
           "gru_gates_fwd", "gru_gates_bwd", "gru_cell_fwd", "gru_cell_fwd_multi", "gru_cell_split_fwd_multi", "sb_gemm_n128", "sb_gemm", "gru_seq_fwd", "gru_seq_fwd_multi", "gru_seq_split_fwd_multi", "gru_seq_split_bwd_multi", "gru_seq_save_elems", "gru_seq_bwd", "gru_seq_bwd_multi", "gru_seq_bwd_workspace", "wgrad_tn", "wgrad_tn_workspace", "wgrad_split_tn", "wgrad_split_tn2", "wgrad_split_workspace", "rollout_record", "ppo_loss_fwd_bwd", "ppo_loss_prob_fwd_bwd", "ppo_loss_workspace",
           "mappo_ops_error_string")

_lib = None


def lib_path():
    return _build.lib_path("libmappo_ops.so")


def load_library():
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                               "the fused MAPPO ops have no fallback")
        L = C.CDLL(path)
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        L.dhgn_msg_agg_fwd.argtypes = [i32, i32, i32, i32, i32, vp, i64, vp, i64, i32, vp, i64, vp, i64, i32, vp, vp, vp, vp, i64, vp]
        L.dhgn_msg_agg_bwd.argtypes = [i32, i32, i32, i32, i32, vp, i64, vp, i64, i32, vp, i64, vp, i64, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp]
        L.dhgn_msg_agg3_fwd.argtypes = [vp, i32, i32, i32, vp, i64, vp, i64, vp]
        L.dhgn_msg_agg3_pair_fwd.argtypes = [vp, i32, i32, i32, vp, i64, vp, vp, vp, i64, vp]
        L.dhgn_msg_agg3_pair01_fwd.argtypes = [vp, i32, i32, i32, vp, i64, vp, vp, i64, vp]
        L.dhgn_msg_agg_bwd_pair.argtypes = [i32, i32, i32, i32, i32, vp, i64, vp, i64, i32, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp]
        L.dhgn_msg_agg3_pair_pos_fwd.argtypes = [vp, i32, i32, i32, vp, i64, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp]
        L.spectral_norm_weight.argtypes = [i32, i32, vp, vp, vp, f32, i32, vp, vp]
        L.dhgn_msg_agg_bwd_workspace.argtypes = [i32, i32]
        L.dhgn_msg_agg_bwd_workspace.restype = i64
        L.dhgn_msg_agg_ones_sorted_ok.argtypes = [i32] * 6
        L.dhgn_msg_agg_ones_sorted_workspace.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp]
        L.dhgn_msg_agg_ones_sorted_workspace.restype = i64
        L.dhgn_msg_agg_ones_sorted_fwd.argtypes = [i32, i32, i32, i32, vp, i64, vp, i64, i32, vp, vp, vp, i64, vp, vp, vp]
        L.dhgn_msg_agg_ones_sorted_bwd.argtypes = [i32, i32, i32, i32, vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, vp]
        L.gae_advnorm.argtypes = [i32, i32, i32, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp]
        L.gae_advnorm_workspace.restype = i64
        L.categorical_sample.argtypes = [i32, i32, vp, C.c_uint64, C.c_uint64, i32, vp, vp, vp]
        L.categorical_sample_counter.argtypes = [i32, i32, vp, C.c_uint64, vp, i32, vp, vp, vp]
        L.head_linear.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp]
        L.head_sample.argtypes = [i32, i32, i32, vp, vp, vp, C.c_uint64, vp, vp, i32, vp, vp, vp]
        L.gru_gates_fwd.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp]
        L.gru_gates_bwd.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.gru_cell_fwd.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.gru_seq_fwd.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]
        L.gru_seq_bwd.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
        L.gru_cell_fwd_multi.argtypes = [i32, vp, i32, i32, vp]
        L.gru_cell_split_fwd_multi.argtypes = [i32, vp, i32, i32, vp]
        L.sb_gemm_n128.argtypes = [i64, i32, vp, i64, vp, i64, vp, i32, vp, i64, vp, i64, vp]
        L.sb_gemm.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i32, vp, i64, vp, i64, vp]
        L.gru_seq_fwd_multi.argtypes = [i32, vp, i32, i32, i32, i32, vp]
        L.gru_seq_bwd_multi.argtypes = [i32, vp, i32, i32, i32, i32, vp]
        L.gru_seq_split_fwd_multi.argtypes = [i32, vp, i32, i32, i32, i32, vp]
        L.gru_seq_split_bwd_multi.argtypes = [i32, vp, i32, i32, i32, i32, vp]
        L.gru_seq_save_elems.argtypes = [i32, i32]
        L.gru_seq_save_elems.restype = i64
        L.gru_seq_bwd_workspace.argtypes = [i32]
        L.gru_seq_bwd_workspace.restype = i64
        L.wgrad_tn_workspace.argtypes = [i32, i32]
        L.wgrad_tn_workspace.restype = i64
        L.wgrad_tn.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i32, vp, vp]
        L.wgrad_split_workspace.restype = i64
        L.wgrad_split_workspace.argtypes = [i32, i32]
        L.wgrad_split_tn.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i32, vp, vp]
        L.wgrad_split_tn2.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64, vp, i64, vp, i32, vp, vp]
        L.fcra_neighbour_mean.argtypes = [i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, vp, i32, vp, vp, i64, vp]
        L.fcra_neighbour_mean_multi.argtypes = [i32, vp, i32, i32, i32, i32, i64, i64, i64, i64, vp, i64, i32, i64, vp]
        L.sb_gemm_signs.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i32, vp, i64, vp, i64, vp, i64, vp]
        L.sb_gemm_masked_bits.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp]
        L.sb_gemm_masked_workspace.argtypes = [i32]
        L.sb_gemm_masked_workspace.restype = i64
        L.sb_gemm_masked.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp]
        L.relu_bwd_colsum_workspace.argtypes = [i32]
        L.relu_bwd_colsum_workspace.restype = i64
        L.relu_bwd_colsum.argtypes = [i64, i32, vp, i64, vp, i64, vp, vp, vp, vp]
        L.wgrad_skinny_workspace.argtypes = [i32, i32]
        L.wgrad_skinny_workspace.restype = i64
        L.wgrad_skinny.argtypes = [i64, i32, i32, vp, i64, vp, i64, i32, vp, vp, vp, vp, vp]
        L.rollout_record.argtypes = [i32, i32, vp, vp, vp, i32, vp]
        L.ppo_loss_workspace.restype = i64
        L.ppo_loss_fwd_bwd.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp]
        L.ppo_loss_prob_fwd_bwd.argtypes = [i64, i32, vp, vp, i64, i64, i64, i64, i64, vp, vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp]
        L.sb_split_diag.argtypes = [i64, vp, vp, vp]
        L.mappo_ops_error_string.argtypes = [C.c_int]
        L.mappo_ops_error_string.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load_library().mappo_ops_error_string(rc).decode()} (code {rc})")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the fused HIP ops run on the GPU only (no CPU fallback)")


def adj_row_words(K):
    """include/mappo_ops.h MO_ADJ_ROW_WORDS: 32-bit words of one bit-packed adjacency row of K neighbours"""
    return (((K + 31) >> 5) + 3) & ~3


def pack_adj_bits(adj):
    """(..., K) 0/1 float adjacency -> (..., adj_row_words(K)) int32, bit j of the row = adj[..., j] (the env's o_adj_bits
    layout).  Plain torch: used for fixtures and tests, the rollout gets the packed rows straight from the env kernel."""
    K = adj.shape[-1]
    RW = adj_row_words(K)
    b = (adj != 0).to(torch.int64)
    pad = RW * 32 - K
    if pad:
        b = torch.cat((b, b.new_zeros(*b.shape[:-1], pad)), -1)
    b = b.reshape(*b.shape[:-1], RW, 32)
    w = (b << torch.arange(32, device=adj.device, dtype=torch.int64)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def unpack_adj_bits(bits, K):
    """inverse of pack_adj_bits: (..., RW) int32 -> (..., K) float32"""
    sh = torch.arange(32, device=bits.device, dtype=torch.int32)
    b = (bits.unsqueeze(-1) >> sh) & 1
    return b.reshape(*bits.shape[:-1], -1)[..., :K].to(torch.float32)


def _vec_stride(t):
    """t (..., E): the E-vectors of all leading positions, in C order, are `ld` elements apart (ld = E: a dense tensor; ld > E: a
    column block of a dense (rows, ld) matrix).  Returns ld; asserts that layout."""
    E = t.shape[-1]
    ld = t.stride(-2) if t.dim() > 1 else E
    assert t.stride(-1) == 1 and ld >= E
    n = 1
    for k in range(t.dim() - 2, -1, -1):
        assert t.shape[k] == 1 or t.stride(k) == n * ld, "not a column block of a dense matrix"
        n *= t.shape[k]
    return ld


def _rows_ok(t):
    """each row (dim 0) is a dense block; rows may be strided (a slice buffer[:, t] of an (N, T, ...) tensor)"""
    return t[0].is_contiguous() and (t.shape[0] == 1 or t.stride(0) >= t[0].numel())


_workspaces = {}


def _workspace(device, E, din):
    key = (device, E, din, torch.cuda.current_stream().cuda_stream)  # per stream: actor and critic backward may overlap
    ws = _workspaces.get(key)
    if ws is None:
        n = load_library().dhgn_msg_agg_bwd_workspace(E, din)
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _msg_call(kind, L, p, q, e, adj, kvalid, W, b, adj_mode, q_div, io, io_stride, extra=()):
    """one relation: io = out (fwd) or gout (bwd) slot pointer with io_stride elements between [E] vectors"""
    R, P = p.shape[0], p.shape[1]
    K = q.shape[1]
    E, din = W.shape
    assert p.dtype == torch.float32 and p.shape[2] == 4 and _rows_ok(p)
    assert _rows_ok(q) and q.shape[2] == 4 and q.shape[0] * q_div == R and q.stride(0) % 4 == 0
    if adj_mode == ADJ_TENSOR:
        assert adj.shape == (R, P, K) and adj.dtype == torch.float32 and _rows_ok(adj)
    if adj_mode == ADJ_BITS:
        assert adj.shape == (R, P, adj_row_words(K)) and adj.dtype == torch.int32 and _rows_ok(adj)
    if e is not None:
        assert e.shape == (R, 4) and _rows_ok(e)
    if adj_mode == ADJ_VALID:
        assert kvalid.dtype == torch.int32 and kvalid.is_contiguous() and kvalid.shape[0] * q_div == R
    args = (R, P, K, E, din, _ptr(p), p.stride(0), _ptr(q), q.stride(0), q_div, _ptr(e), e.stride(0) if e is not None else 0,
            _ptr(adj) if adj_mode in (ADJ_TENSOR, ADJ_BITS) else None, adj.stride(0) if adj_mode in (ADJ_TENSOR, ADJ_BITS) else 0, adj_mode,
            _ptr(kvalid) if adj_mode == ADJ_VALID else None, _ptr(W), _ptr(b), io, io_stride)
    if kind == "fwd":
        _check(L.dhgn_msg_agg_fwd(*args, _stream()), "dhgn_msg_agg_fwd")
    else:
        _check(L.dhgn_msg_agg_bwd(*args, *extra, _stream()), "dhgn_msg_agg_bwd")


class _MsgAgg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, q, e, adj, kvalid, W, b, adj_mode, q_div):
        L = load_library()
        _need_gpu(p, "dhgn_msg_agg")
        R, P = p.shape[0], p.shape[1]
        E = W.shape[0]
        Wc, bc = W.detach().contiguous(), b.detach().contiguous()
        out = torch.empty((R, P, E), dtype=torch.float32, device=p.device)
        _msg_call("fwd", L, p, q, e, adj, kvalid, Wc, bc, adj_mode, q_div, _ptr(out), E)
        ctx.save_for_backward(p, q, e, adj, kvalid, Wc, bc)
        ctx.meta = (adj_mode, q_div)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = load_library()
        p, q, e, adj, kvalid, W, b = ctx.saved_tensors
        adj_mode, q_div = ctx.meta
        E, din = W.shape
        gout = gout.contiguous()
        dW, db = torch.empty_like(W), torch.empty_like(b)
        ws = _workspace(p.device, E, din)
        _msg_call("bwd", L, p, q, e, adj, kvalid, W, b, adj_mode, q_div, _ptr(gout), E, (_ptr(dW), _ptr(db), _ptr(ws)))
        return None, None, None, None, None, dW, db, None, None


def msg_agg(p, q, e, adj, W, b, adj_mode=ADJ_TENSOR, kvalid=None, q_div=1):
    """sum_j abar_ij ReLU(W [p_i - q_j, p_i - e] + b): DHGN.coordinate/message/mean_operator fused
    (reference DHGN/mappo_parallel.py:235-239, 323-334, 346-347).  p (R,P,4), q (R/q_div,K,4), e (R,4)|None,
    adj (R,P,K)|None -> (R,P,E)."""
    return _MsgAgg.apply(p, q, e, adj if adj_mode in (ADJ_TENSOR, ADJ_BITS) else None, kvalid if adj_mode == ADJ_VALID else None, W, b,
                         adj_mode, q_div)


class MsgRel(C.Structure):
    """include/mappo_ops.h mo_msg_rel"""
    _fields_ = [("K", C.c_int32), ("din", C.c_int32), ("q_div", C.c_int32), ("adj_mode", C.c_int32), ("q", C.c_void_p), ("q_rs", C.c_int64),
                ("e", C.c_void_p), ("e_rs", C.c_int64), ("adj", C.c_void_p), ("adj_rs", C.c_int64), ("kvalid", C.c_void_p), ("W", C.c_void_p),
                ("b", C.c_void_p)]


def _msg3_rels(p, rels):
    """the mo_msg_rel[3] of three (q, e, adj, kvalid, W, b, adj_mode, q_div)"""
    R, P = p.shape[0], p.shape[1]
    arr = (MsgRel * 3)()
    for r, (q, e, adj, kv, W, b, mode, qd) in enumerate(rels):
        K, din = q.shape[1], W.shape[1]
        assert _rows_ok(q) and q.shape[2] == 4 and q.shape[0] * qd == R and q.stride(0) % 4 == 0
        m = arr[r]
        m.K, m.din, m.q_div, m.adj_mode = K, din, qd, mode
        m.q, m.q_rs = q.data_ptr(), q.stride(0)
        if e is not None:
            assert e.shape == (R, 4) and _rows_ok(e)
            m.e, m.e_rs = e.data_ptr(), e.stride(0)
        if mode in (ADJ_TENSOR, ADJ_BITS):
            assert adj.shape == (R, P, K if mode == ADJ_TENSOR else adj_row_words(K)) and _rows_ok(adj)
            assert adj.dtype == (torch.float32 if mode == ADJ_TENSOR else torch.int32)
            m.adj, m.adj_rs = adj.data_ptr(), adj.stride(0)
        if mode == ADJ_VALID:
            assert kv.dtype == torch.int32 and kv.is_contiguous() and kv.shape[0] * qd == R
            m.kvalid = kv.data_ptr()
        m.W, m.b = W.data_ptr(), b.data_ptr()
    assert p.dtype == torch.float32 and p.shape[2] == 4 and _rows_ok(p)
    return arr


def _msg3_call(L, p, rels, out, E):
    """one launch for the three relations"""
    arr = _msg3_rels(p, rels)
    _check(L.dhgn_msg_agg3_fwd(C.cast(arr, C.c_void_p), p.shape[0], p.shape[1], E, _ptr(p), p.stride(0), _ptr(out), 3 * E, _stream()),
           "dhgn_msg_agg3_fwd")


def msg_agg3_pair(p, e, o, adj_p, adj_e, adj_o, W0, b0, W1, b1, W2, b2, o_kvalid=None, q_div=1, out=None, pos=None):
    """(2, R, P, 3, E): msg_agg3 of the actor (slot 0: the observed adjacency) and of the critic (slot 1: ones, for the obstacle
    relation over the first o_kvalid[row] obstacles when given) from ONE pass over the messages -- the two networks share the
    encoder weights (DHGN/mappo_parallel.py:582-616).  Rollout only (no autograd); bit-identical to two msg_agg3 calls."""
    assert not (torch.is_grad_enabled() and any(t.requires_grad for t in (W0, b0, W1, b1, W2, b2))), "msg_agg3_pair has no backward"
    L = load_library()
    _need_gpu(p, "dhgn_msg_agg3_pair")
    R, P = p.shape[0], p.shape[1]
    E = W0.shape[0]
    ws = [t.detach().contiguous() for t in (W0, b0, W1, b1, W2, b2)]
    if out is None:
        out = torch.empty((2, R, P, 3, E), dtype=torch.float32, device=p.device)
    assert out.shape == (2, R, P, 3, E) and out.is_contiguous() and out.dtype == torch.float32
    mode_o = ADJ_BITS if adj_o.dtype == torch.int32 else ADJ_TENSOR
    arr = _msg3_rels(p, ((p, e.reshape(R, 4), adj_p, None, ws[0], ws[1], ADJ_TENSOR, 1), (e, None, adj_e, None, ws[2], ws[3], ADJ_TENSOR, 1),
                         (o, None, adj_o, None, ws[4], ws[5], mode_o, q_div)))
    if o_kvalid is not None:
        assert o_kvalid.dtype == torch.int32 and o_kvalid.is_contiguous() and o_kvalid.shape[0] * q_div == R
    if pos is not None:
        # pos = (Wp (E, 4) -- a column slice of the semantic layer's weight --, bp (E,), h0 (2, R, P, E)): the same launch writes the
        # layer's position part bp + Wp p for both networks into h0
        Wp, bp, h0 = pos
        Wp, bp = Wp.detach(), bp.detach().contiguous()
        ld = _vec_stride(h0)   # E (dense) or the row length of a wider matrix whose column block h0 is
        assert Wp.shape == (E, 4) and Wp.stride(1) == 1 and h0.shape == (2, R, P, E)
        _check(L.dhgn_msg_agg3_pair_pos_fwd(C.cast(arr, C.c_void_p), R, P, E, _ptr(p), p.stride(0), _ptr(o_kvalid), _ptr(out[0]), _ptr(out[1]), 3 * E,
                                            _ptr(Wp), Wp.stride(0), _ptr(bp), _ptr(h0[0]), _ptr(h0[1]), ld, _stream()), "dhgn_msg_agg3_pair_pos_fwd")
        return out
    _check(L.dhgn_msg_agg3_pair_fwd(C.cast(arr, C.c_void_p), R, P, E, _ptr(p), p.stride(0), _ptr(o_kvalid), _ptr(out[0]), _ptr(out[1]), 3 * E,
                                    _stream()), "dhgn_msg_agg3_pair_fwd")
    return out


def spectral_norm_weight(weight_orig, u, v, eps=1e-12, n_power_iterations=1, out=None):
    """torch.nn.utils.spectral_norm's pre-forward hook (SpectralNorm.compute_weight) for a small head as one launch: the power
    iteration updates u and v IN PLACE (n_power_iterations = 0: the eval-mode form), returns weight_orig / sigma.  No autograd."""
    L = load_library()
    _need_gpu(weight_orig, "spectral_norm_weight")
    A, H = weight_orig.shape
    W = weight_orig.detach()
    assert W.is_contiguous() and u.is_contiguous() and v.is_contiguous() and u.shape == (A,) and v.shape == (H,)
    if out is None:
        out = torch.empty_like(W)
    assert out.shape == W.shape and out.is_contiguous()
    _check(L.spectral_norm_weight(A, H, _ptr(W), _ptr(u), _ptr(v), float(eps), int(n_power_iterations), _ptr(out), _stream()),
           "spectral_norm_weight")
    return out


SORTED_ONES_MIN_QDIV = 8  # rows sharing one neighbour set from which the sort + binary-search kernels beat the O(K) loop


def _sorted_ones_ok(L, p, q, W, adj_mode, q_div):
    """the critic's all-ones obstacle relation over a neighbour set shared by q_div rows: O(log K) kernels (mappo_ops.h)"""
    return (adj_mode == ADJ_ONES and q_div >= SORTED_ONES_MIN_QDIV and p.stride(0) % 4 == 0 and p.data_ptr() % 16 == 0 and q.data_ptr() % 16 == 0
            and bool(L.dhgn_msg_agg_ones_sorted_ok(p.shape[0], p.shape[1], q.shape[1], W.shape[0], W.shape[1], q_div)))


class _MsgAgg3(torch.autograd.Function):
    """The three relations of DHGN.encoder in one autograd node writing one (R, P, 3, E) tensor (relation r in slot r), so
    the shared AGG_vertex_0 layer runs as a single GEMM on a view and no stack / concatenate copies exist."""

    @staticmethod
    def forward(ctx, p, e, o, adj_p, adj_e, adj_o, kvalid, W0, b0, W1, b1, W2, b2, mode, mode_o, q_div):
        L = load_library()
        _need_gpu(p, "dhgn_msg_agg")
        R, P = p.shape[0], p.shape[1]
        E = W0.shape[0]
        ws = [t.detach().contiguous() for t in (W0, b0, W1, b1, W2, b2)]
        e2 = e.reshape(R, 4)
        out = torch.empty((R, P, 3, E), dtype=torch.float32, device=p.device)
        slot = lambda r: C.c_void_p(out.data_ptr() + 4 * r * E)
        save_m = qtab = None
        ctx.sorted_o = _sorted_ones_ok(L, p, o, ws[4], mode_o, q_div)
        if not ctx.sorted_o:   # the three relations in one launch
            _msg3_call(L, p, ((p, e2, adj_p, None, ws[0], ws[1], mode, 1), (e, None, adj_e, None, ws[2], ws[3], mode, 1),
                              (o, None, adj_o, kvalid, ws[4], ws[5], mode_o, q_div)), out, E)
            ctx.save_for_backward(p, e, o, adj_p, adj_e, adj_o, kvalid, save_m, qtab, *ws)
            ctx.meta = (mode, mode_o, q_div)
            return out
        _msg_call("fwd", L, p, p, e2, adj_p, None, ws[0], ws[1], mode, 1, slot(0), 3 * E)
        _msg_call("fwd", L, p, e, None, adj_e, None, ws[2], ws[3], mode, 1, slot(1), 3 * E)
        if ctx.sorted_o:
            K = o.shape[1]
            if any(ctx.needs_input_grad):
                save_m = torch.empty((R, P, E), dtype=torch.uint8, device=p.device)
                qtab = torch.empty((R // q_div, E, 4, K + 1), dtype=torch.float32, device=p.device)
            _check(L.dhgn_msg_agg_ones_sorted_fwd(R, P, K, E, _ptr(p), p.stride(0), _ptr(o), o.stride(0), q_div, _ptr(ws[4]), _ptr(ws[5]), slot(2), 3 * E,
                                                  _ptr(save_m), _ptr(qtab), _stream()), "dhgn_msg_agg_ones_sorted_fwd")
        else:
            _msg_call("fwd", L, p, o, None, adj_o, kvalid, ws[4], ws[5], mode_o, q_div, slot(2), 3 * E)
        ctx.save_for_backward(p, e, o, adj_p, adj_e, adj_o, kvalid, save_m, qtab, *ws)
        ctx.meta = (mode, mode_o, q_div)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = load_library()
        p, e, o, adj_p, adj_e, adj_o, kvalid, save_m, qtab, W0, b0, W1, b1, W2, b2 = ctx.saved_tensors
        mode, mode_o, q_div = ctx.meta
        R, P = p.shape[0], p.shape[1]
        E = W0.shape[0]
        gout = gout.contiguous()
        slot = lambda r: C.c_void_p(gout.data_ptr() + 4 * r * E)
        grads = []
        for r, (q, ee, adj, kv, W, b, m, qd) in enumerate(((p, e.reshape(R, 4), adj_p, None, W0, b0, mode, 1), (e, None, adj_e, None, W1, b1, mode, 1),
                                                           (o, None, adj_o, kvalid, W2, b2, mode_o, q_div))):
            dW, db = torch.empty_like(W), torch.empty_like(b)
            if r == 2 and ctx.sorted_o:
                part = torch.empty((R // q_div) * 5 * E, dtype=torch.float32, device=p.device)
                _check(L.dhgn_msg_agg_ones_sorted_bwd(R, P, o.shape[1], E, _ptr(p), p.stride(0), q_div, slot(2), 3 * E, _ptr(save_m), _ptr(qtab),
                                                      _ptr(dW), _ptr(db), _ptr(part), _stream()), "dhgn_msg_agg_ones_sorted_bwd")
                grads += [dW, db]
                continue
            ws = _workspace(p.device, E, W.shape[1])
            _msg_call("bwd", L, p, q, ee, adj, kv, W, b, m, qd, slot(r), 3 * E, (_ptr(dW), _ptr(db), _ptr(ws)))
            grads += [dW, db]
        return (None,) * 7 + tuple(grads) + (None, None, None)


class _MsgAgg3Pair(torch.autograd.Function):
    """The update's message pass of ACTOR and CRITIC together (they hold the same DHGN instance, DHGN/mappo_parallel.py:582-616: same
    MSG weights, same p / e / o rows, the observed adjacency vs ones): -> (m3_actor, m3_critic), each (R, P, 3, E).
    Forward: one launch for the actor's three relations plus the critic's relations 0 and 1 (the messages are computed once); the
    critic's obstacle relation (ones over all padded slots in training) is the sorted all-ones kernel.  Backward: relations
    0 and 1 are ONE pass each for both networks (dhgn_msg_agg_bwd_pair: the two gradients share the ReLU mask and sum into one dW),
    relation 2 the actor's bit-walk kernel plus the sorted kernel.  The numbers are those of two _MsgAgg3 nodes."""

    @staticmethod
    def forward(ctx, p, e, o, adj_p, adj_e, adj_o, W0, b0, W1, b1, W2, b2, q_div):
        L = load_library()
        _need_gpu(p, "dhgn_msg_agg")
        R, P = p.shape[0], p.shape[1]
        E, K = W0.shape[0], o.shape[1]
        ws = [t.detach().contiguous() for t in (W0, b0, W1, b1, W2, b2)]
        e2 = e.reshape(R, 4)
        out_a = torch.empty((R, P, 3, E), dtype=torch.float32, device=p.device)
        out_c = torch.empty((R, P, 3, E), dtype=torch.float32, device=p.device)
        mode_o = ADJ_BITS if adj_o.dtype == torch.int32 else ADJ_TENSOR
        arr = _msg3_rels(p, ((p, e2, adj_p, None, ws[0], ws[1], ADJ_TENSOR, 1), (e, None, adj_e, None, ws[2], ws[3], ADJ_TENSOR, 1),
                             (o, None, adj_o, None, ws[4], ws[5], mode_o, q_div)))
        _check(L.dhgn_msg_agg3_pair01_fwd(C.cast(arr, C.c_void_p), R, P, E, _ptr(p), p.stride(0), _ptr(out_a), _ptr(out_c), 3 * E, _stream()),
               "dhgn_msg_agg3_pair01_fwd")
        need = any(ctx.needs_input_grad)
        save_m = torch.empty((R, P, E), dtype=torch.uint8, device=p.device) if need else None
        qtab = torch.empty((R // q_div, E, 4, K + 1), dtype=torch.float32, device=p.device) if need else None
        _check(L.dhgn_msg_agg_ones_sorted_fwd(R, P, K, E, _ptr(p), p.stride(0), _ptr(o), o.stride(0), q_div, _ptr(ws[4]), _ptr(ws[5]),
                                              C.c_void_p(out_c.data_ptr() + 4 * 2 * E), 3 * E, _ptr(save_m), _ptr(qtab), _stream()),
               "dhgn_msg_agg_ones_sorted_fwd")
        ctx.save_for_backward(p, e, o, adj_p, adj_e, adj_o, save_m, qtab, *ws)
        ctx.meta = (mode_o, q_div)
        return out_a, out_c

    @staticmethod
    def backward(ctx, ga, gc):
        L = load_library()
        p, e, o, adj_p, adj_e, adj_o, save_m, qtab, W0, b0, W1, b1, W2, b2 = ctx.saved_tensors
        mode_o, q_div = ctx.meta
        R, P = p.shape[0], p.shape[1]
        E = W0.shape[0]
        ga, gc = ga.contiguous(), gc.contiguous()
        slot = lambda g, r: C.c_void_p(g.data_ptr() + 4 * r * E)
        grads = []
        for r, (q, ee, adj, W, b) in enumerate(((p, e.reshape(R, 4), adj_p, W0, b0), (e, None, adj_e, W1, b1))):
            K, din = q.shape[1], W.shape[1]
            dW, db = torch.empty_like(W), torch.empty_like(b)
            ws = _workspace(p.device, E, din)
            assert adj.shape == (R, P, K) and adj.dtype == torch.float32 and _rows_ok(adj) and _rows_ok(q) and q.shape[0] == R
            _check(L.dhgn_msg_agg_bwd_pair(R, P, K, E, din, _ptr(p), p.stride(0), _ptr(q), q.stride(0), 1, _ptr(ee), ee.stride(0) if ee is not None else 0,
                                           _ptr(adj), adj.stride(0), _ptr(W), _ptr(b), slot(ga, r), slot(gc, r), 3 * E, _ptr(dW), _ptr(db), _ptr(ws),
                                           _stream()), "dhgn_msg_agg_bwd_pair")
            grads += [dW, db]
        # the obstacle relation: actor (bit-packed / float adjacency) and critic (sorted all-ones), summed
        dWa, dba = torch.empty_like(W2), torch.empty_like(b2)
        _msg_call("bwd", L, p, o, None, adj_o, None, W2, b2, mode_o, q_div, slot(ga, 2), 3 * E, (_ptr(dWa), _ptr(dba), _ptr(_workspace(p.device, E, 4))))
        dWc, dbc = torch.empty_like(W2), torch.empty_like(b2)
        part = torch.empty((R // q_div) * 5 * E, dtype=torch.float32, device=p.device)
        _check(L.dhgn_msg_agg_ones_sorted_bwd(R, P, o.shape[1], E, _ptr(p), p.stride(0), q_div, slot(gc, 2), 3 * E, _ptr(save_m), _ptr(qtab),
                                              _ptr(dWc), _ptr(dbc), _ptr(part), _stream()), "dhgn_msg_agg_ones_sorted_bwd")
        grads += [dWa + dWc, dba + dbc]
        return (None,) * 6 + tuple(grads) + (None,)


def msg_agg3_pair_train_ok(p, o, W2, q_div):
    """shapes the paired update path covers (the critic's obstacle relation must take the sorted all-ones kernels)"""
    return p.is_cuda and _sorted_ones_ok(load_library(), p, o, W2, ADJ_ONES, q_div)


def msg_agg3_pair_train(p, e, o, adj_p, adj_e, adj_o, W0, b0, W1, b1, W2, b2, q_div=1):
    """-> (m3_actor, m3_critic) with autograd: msg_agg3(.., False) and msg_agg3(.., True) of one mini-batch from shared passes"""
    return _MsgAgg3Pair.apply(p, e, o, adj_p, adj_e, adj_o, W0, b0, W1, b1, W2, b2, q_div)


def msg_agg3(p, e, o, adj_p, adj_e, adj_o, W0, b0, W1, b1, W2, b2, is_critic, o_kvalid=None, q_div=1):
    """(R, P, 3, E): the message/aggregate of the defender, evader and obstacle relation (DHGN/mappo_parallel.py:256-281)."""
    mode = ADJ_ONES if is_critic else ADJ_TENSOR
    mode_o = ADJ_VALID if (is_critic and o_kvalid is not None) else mode
    if mode_o == ADJ_TENSOR and adj_o.dtype == torch.int32:  # bit-packed LiDAR rows (env o_adj_bits): same results, 1/29 of the bytes
        mode_o = ADJ_BITS
    return _MsgAgg3.apply(p, e, o, adj_p, adj_e, adj_o, o_kvalid if mode_o == ADJ_VALID else None, W0, b0, W1, b1, W2, b2, mode,
                          mode_o, q_div)


def gae_advnorm(r, v, active, gamma, lamda, use_adv_norm=True):
    """GAE + v_target + advantage normalisation (reference DHGN/mappo_parallel.py:643-658). r/active (N,T,P), v (N,T+1,P)."""
    L = load_library()
    _need_gpu(r, "gae_advnorm")
    N, T, P = r.shape
    assert v.shape == (N, T + 1, P) and active.shape == r.shape
    r, v, active = r.contiguous(), v.contiguous(), active.contiguous()
    adv = torch.empty_like(r)
    v_target = torch.empty_like(r)
    stats = torch.empty(L.gae_advnorm_workspace() // 8, dtype=torch.float64, device=r.device)
    _check(L.gae_advnorm(N, T, P, _ptr(r), _ptr(v), _ptr(active), float(gamma), float(lamda), 1 if use_adv_norm else 0,
                         _ptr(adv), _ptr(v_target), _ptr(stats), _stream()), "gae_advnorm")
    return adv, v_target


def categorical_sample(probs, seed, offset, greedy=False, counter=None, out=None):
    """Categorical(probs).sample() and log_prob, or argmax when greedy (reference DHGN/mappo_parallel.py:442-448).
    `counter` (int64 device tensor of one element) replaces the host-side offset: the stream position lives on the
    device and advances by the number of rows, which makes the call replayable inside a captured graph."""
    L = load_library()
    _need_gpu(probs, "categorical_sample")
    shape = probs.shape[:-1]
    A = probs.shape[-1]
    pr = probs.reshape(-1, A).contiguous()
    R = pr.shape[0]
    if out is not None:  # (action int32, logp float32) dense tensors of `shape`, written in place (static rollout storage)
        action, logp = out
        assert action.dtype == torch.int32 and logp.dtype == torch.float32 and action.is_contiguous() and logp.is_contiguous()
        assert action.numel() == R and logp.numel() == R
    else:
        action = torch.empty(R, dtype=torch.int32, device=probs.device)
        logp = torch.empty(R, dtype=torch.float32, device=probs.device)
    if counter is not None:
        assert counter.dtype == torch.int64 and counter.numel() == 1 and counter.is_cuda
        _check(L.categorical_sample_counter(R, A, _ptr(pr), int(seed), _ptr(counter), 1 if greedy else 0, _ptr(action), _ptr(logp),
                                            _stream()), "categorical_sample_counter")
    else:
        _check(L.categorical_sample(R, A, _ptr(pr), int(seed), int(offset), 1 if greedy else 0, _ptr(action), _ptr(logp), _stream()),
               "categorical_sample")
    return action.reshape(shape), logp.reshape(shape)


HEAD_MAX_OUT, HEAD_FEATURES = 16, 128


def _head_ok(feat, W, b):
    return (feat.is_cuda and feat.dtype == torch.float32 and feat.is_contiguous() and W.shape[1] == HEAD_FEATURES and feat.shape[-1] == HEAD_FEATURES
            and W.shape[0] <= HEAD_MAX_OUT and b is not None and not torch.is_grad_enabled()
            and feat.data_ptr() % 16 == 0)   # the kernel reads feature rows with 16-byte loads (an offset view falls back to F.linear)


def head_linear(feat, W, b, out=None):
    """feat W^T + b of a head with <= 16 outputs on 128 features in one launch (rollout, no autograd); out: dense storage of
    feat.shape[:-1] + (A,) elements, written in place.  Other shapes: F.linear."""
    if not _head_ok(feat, W, b):
        y = F.linear(feat, W, b)
        return y if out is None else out.copy_(y.reshape(out.shape))
    L = load_library()
    A = W.shape[0]
    R = feat.numel() // HEAD_FEATURES
    if out is None:
        out = torch.empty(feat.shape[:-1] + (A,), dtype=torch.float32, device=feat.device)
    assert out.is_contiguous() and out.numel() == R * A and out.dtype == torch.float32
    _check(L.head_linear(R, A, HEAD_FEATURES, _ptr(feat), _ptr(W.detach().contiguous()), _ptr(b.detach().contiguous()), _ptr(out), _stream()),
           "head_linear")
    return out


def head_sample(feat, W, b, seed, counter, ticket, out, greedy=False):
    """Categorical(softmax(feat W^T + b)).sample() and its log-probability (argmax when greedy) for an action head with <= 16
    outputs on 128 features in one launch; the stream position `counter` (int64, one element) advances by the number of rows
    (replayable in a captured graph).  ticket: a zero uint32 tensor of one element.  out = (action int32, logp float32)."""
    assert _head_ok(feat, W, b)
    L = load_library()
    action, logp = out
    R = feat.numel() // HEAD_FEATURES
    assert action.dtype == torch.int32 and logp.dtype == torch.float32 and action.is_contiguous() and logp.is_contiguous()
    assert action.numel() == R and logp.numel() == R and counter.dtype == torch.int64 and counter.numel() == 1
    assert ticket.dtype == torch.int32 and ticket.numel() == 1
    _check(L.head_sample(R, W.shape[0], HEAD_FEATURES, _ptr(feat), _ptr(W.detach().contiguous()), _ptr(b.detach().contiguous()), int(seed),
                         _ptr(counter), _ptr(ticket), 1 if greedy else 0, _ptr(action), _ptr(logp), _stream()), "head_sample")
    return action, logp


WGRAD_MIN_ROWS = 4096  # below this the BLAS library's single-workgroup-tile GEMMs are as fast


# ---- how the fp32 matrix products of the hot path are evaluated ------------------------------------------------------------------
# "split_bf16" (default): every fp32 operand is split EXACTLY into three bf16 numbers and a product is the six piece products with
# i + j <= 4 on the bf16 matrix pipe with fp32 accumulation (csrc/mappo_ops.hip k_gru_cell_sb, k_sb_gemm_n128, k_sb_wgrad) -- fp32
# inputs, outputs and stored tensors, the error against f64 of an fp32 GEMM (tests/test_ops_gpu.py), 2.67 x the fp32 matrix rate.
# "fp32": v_mfma_f32_16x16x4_f32 kernels / the BLAS library's fp32 GEMMs.  One switch, `runtime.matmul` (MAPPO.__init__) or
# set_matmul_mode(); the three parts can be switched separately through the environment for A/B measurements.
MATMUL_MODE = os.environ.get("MAPPO_MATMUL", "split_bf16")
WGRAD_MODE = os.environ.get("MAPPO_WGRAD", MATMUL_MODE)       # weight gradients (k_sb_wgrad | k_wgrad)
SPLIT_WGRAD_SHAPES = {(128, 128), (128, 256), (256, 128), (128, 384), (384, 128)}


def _wgrad_ok(a, b):
    return (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2
            and a.shape[0] == b.shape[0] and a.shape[0] >= WGRAD_MIN_ROWS
            and a.shape[1] % 128 == 0 and b.shape[1] % 128 == 0 and a.shape[1] <= 1024 and b.shape[1] <= 1024
            and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def wgrad(a, b, out=None, accumulate=False):
    """a^T b for a (K, M), b (K, N) row-major (rows may be strided): the weight gradient of a Linear / GRU projection over
    the K rows of a minibatch (the `grad_output.t() @ input` of autograd, DHGN/mappo_parallel.py:660-708).  Shapes the
    split-K MFMA kernel covers (M, N multiples of 128) run in csrc/mappo_ops.hip k_wgrad, the rest in the BLAS library."""
    if not _wgrad_ok(a, b):
        if out is None:
            return torch.mm(a.t(), b)
        return out.addmm_(a.t(), b) if accumulate else torch.mm(a.t(), b, out=out)
    L = load_library()
    M, N = a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
        accumulate = False
    assert out.is_contiguous() and out.shape == (M, N)
    if WGRAD_MODE == "split_bf16" and (M, N) in SPLIT_WGRAD_SHAPES:
        ws = torch.empty(L.wgrad_split_workspace(M, N), dtype=torch.uint8, device=a.device)
        _check(L.wgrad_split_tn(a.shape[0], M, N, _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), int(bool(accumulate)), _ptr(ws), _stream()),
               "wgrad_split_tn")
        return out
    ws = torch.empty(L.wgrad_tn_workspace(M, N), dtype=torch.uint8, device=a.device)
    _check(L.wgrad_tn(a.shape[0], M, N, _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), int(bool(accumulate)), _ptr(ws), _stream()),
           "wgrad_tn")
    return out


def _nbr_view(z):
    """(pointer tensor, T, episode stride, step stride) of a dense (R, P, E) tensor or an (n, T, P, E) slice of a history buffer"""
    if z.dim() == 3:
        assert z.is_contiguous()
        return z, 1, z.shape[1] * z.shape[2], 0
    assert z.dim() == 4 and z.stride(3) == 1 and z.stride(2) == z.shape[3]
    return z, z.shape[1], z.stride(0), z.stride(1)


def fcra_mean(z_actor=None, z_critic=None, adj=None, bias=None, relu=False, out=None):
    """DHGN.fcra's neighbour mean `matmul(normalize(adj, p=1), z)` (actor) and / or `matmul(normalize(ones), z)` (critic), plus an
    optional bias and ReLU, in one launch (csrc/mappo_ops.hip k_nbr_mean).  z_*: dense (R, P, E) or a (n, T, P, E) slice of a
    history buffer, read in place.  Returns (R, P, E) for one network, (2, R, P, E) (actor, critic) for both.  No autograd:
    the inputs are stored data."""
    L = load_library()
    z = z_actor if z_actor is not None else z_critic
    _need_gpu(z, "fcra_neighbour_mean")
    P, E = z.shape[-2], z.shape[-1]
    R = z.numel() // (P * E)
    both = z_actor is not None and z_critic is not None
    if out is None:
        out = torch.empty(((2, R, P, E) if both else (R, P, E)), dtype=torch.float32, device=z.device)
    assert out.numel() == (2 if both else 1) * R * P * E and out.shape[-1] == E
    out_ld = _vec_stride(out)   # dense, or the left half of an [agg | h] operand
    za = zc = None
    T = 1
    a_es = a_ts = c_es = c_ts = 0
    if z_actor is not None:
        za, T, a_es, a_ts = _nbr_view(z_actor)
        assert adj is not None and adj.dtype == torch.float32 and adj.shape[-2:] == (P, P) and adj.numel() == R * P * P and _rows_ok(adj.reshape(R, P, P))
    if z_critic is not None:
        zc, Tc, c_es, c_ts = _nbr_view(z_critic)
        assert z_actor is None or Tc == T
        T = Tc
    o_a = (out[0] if both else out) if z_actor is not None else None
    o_c = (out[1] if both else out) if z_critic is not None else None
    adj3 = adj.reshape(R, P, P) if adj is not None else None
    _check(L.fcra_neighbour_mean(R, P, E, T, _ptr(za), a_es, a_ts, _ptr(zc), c_es, c_ts, _ptr(adj3), adj3.stride(0) if adj3 is not None else 0,
                                 _ptr(bias.detach() if bias is not None else None), int(bool(relu)), _ptr(o_a), _ptr(o_c), out_ld, _stream()),
           "fcra_neighbour_mean")
    return out


class NbrJob(C.Structure):
    """include/mappo_ops.h mo_nbr_job"""
    _fields_ = [(n, C.c_void_p) for n in ("z_actor", "z_critic", "bias", "out_actor", "out_critic")]


def fcra_mean_pair_multi(zas, zcs, adj, biases, outs):
    """fcra_mean(z_actor=za, z_critic=zc, adj=adj, bias=b, relu=True, out=o) for several hops (za, zc, b, o) of one shape in ONE launch
    (fcra_neighbour_mean_multi): the hops of a rollout tick read stored history slots, none reads another hop's result.
    za, zc: dense (R, P, E); o: (2, R, P, E) storage (possibly the left half of an [agg | h] operand)."""
    L = load_library()
    n = len(zas)
    assert 1 <= n <= 4 and len(zcs) == n and len(biases) == n and len(outs) == n
    P, E = zas[0].shape[-2], zas[0].shape[-1]
    R = zas[0].numel() // (P * E)
    out_ld = _vec_stride(outs[0])
    arr = (NbrJob * n)()
    keep = []
    for k in range(n):
        za, zc, o = zas[k], zcs[k], outs[k]
        assert za.is_contiguous() and zc.is_contiguous() and za.shape == zas[0].shape and zc.shape == zas[0].shape and _vec_stride(o) == out_ld
        b = biases[k].detach()
        keep.append(b)
        a = arr[k]
        a.z_actor, a.z_critic, a.bias, a.out_actor, a.out_critic = za.data_ptr(), zc.data_ptr(), b.data_ptr(), o[0].data_ptr(), o[1].data_ptr()
    adj3 = adj.reshape(R, P, P)
    assert adj3.dtype == torch.float32 and _rows_ok(adj3)
    _check(L.fcra_neighbour_mean_multi(n, C.cast(arr, C.c_void_p), R, P, E, 1, P * E, 0, P * E, 0, _ptr(adj3), adj3.stride(0), 1, out_ld, _stream()),
           "fcra_neighbour_mean_multi")


RELU_BWD_MIN_ROWS = 4096


def relu_bwd_colsum(g, y):
    """(g * (y > 0), its column sums): ReLU backward from the saved output and the bias gradient of the Linear in front of it
    in one pass (csrc/mappo_ops.hip k_relu_bwd_colsum); small or odd shapes take the two torch ops.  g, y: dense, or column
    blocks of wider matrices (last stride 1, one row stride); the result is dense."""
    F_ = g.shape[-1]
    rows = g.numel() // max(F_, 1)

    def ld(t):
        try:
            return _vec_stride(t)
        except AssertionError:
            return None
    lg, ly = ld(g), ld(y)
    if not (g.is_cuda and g.dtype == torch.float32 and y.shape == g.shape and rows >= RELU_BWD_MIN_ROWS and lg is not None and ly is not None
            and lg % 4 == 0 and ly % 4 == 0 and F_ % 4 == 0 and F_ <= 1024 and 256 % (F_ // 4) == 0 and g.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0):
        gin = torch.ops.aten.threshold_backward(g, y, 0.0)
        return gin, gin.reshape(-1, F_).sum(0)
    L = load_library()
    gin = torch.empty(g.shape, dtype=g.dtype, device=g.device)
    cs = torch.empty(F_, dtype=g.dtype, device=g.device)
    ws = torch.empty(L.relu_bwd_colsum_workspace(F_), dtype=torch.uint8, device=g.device)
    _check(L.relu_bwd_colsum(rows, F_, _ptr(g), lg, _ptr(y), ly, _ptr(gin), _ptr(cs), _ptr(ws), _stream()), "relu_bwd_colsum")
    return gin, cs


SKINNY_MAX = 16        # input / output features up to which a Linear's weight gradient takes the streaming kernel
SKINNY_MIN_ROWS = 4096


def _skinny_ok(s, x):
    return (s.is_cuda and s.dtype == torch.float32 and x.dtype == torch.float32 and s.dim() == 2 and x.dim() == 2 and s.shape[0] == x.shape[0]
            and s.shape[0] >= SKINNY_MIN_ROWS and 1 <= s.shape[1] <= SKINNY_MAX and x.shape[1] % 64 == 0 and 64 <= x.shape[1] <= 1024
            and s.stride(1) == 1 and x.stride(1) == 1)


def wgrad_skinny(s, x, transposed=False, colsum_x=False, colsum_s=False):
    """s^T x for s (R, NS <= 16), x (R, F): (NS, F), or (F, NS) when transposed; optionally x.sum(0) and s.sum(0) from the same
    pass (csrc/mappo_ops.hip k_wgrad_skinny).  Returns (C, colsum_x or None, colsum_s or None)."""
    if not _skinny_ok(s, x):
        C = torch.mm(s.t(), x)
        return (C.t().contiguous() if transposed else C), (x.sum(0) if colsum_x else None), (s.sum(0) if colsum_s else None)
    L = load_library()
    R, NS, F = s.shape[0], s.shape[1], x.shape[1]
    C = torch.empty((F, NS) if transposed else (NS, F), dtype=torch.float32, device=x.device)
    cx = torch.empty(F, dtype=torch.float32, device=x.device) if colsum_x else None
    cs = torch.empty(NS, dtype=torch.float32, device=x.device) if colsum_s else None
    ws = torch.empty(L.wgrad_skinny_workspace(NS, F), dtype=torch.uint8, device=x.device)
    _check(L.wgrad_skinny(R, NS, F, _ptr(s), s.stride(0), _ptr(x), x.stride(0), int(bool(transposed)), _ptr(C), _ptr(cx), _ptr(cs), _ptr(ws),
                          _stream()), "wgrad_skinny")
    return C, cx, cs


class _SkinnyLinear(torch.autograd.Function):
    """x W^T + b for a Linear with at most SKINNY_MAX inputs or outputs: forward and input gradient are the library's GEMMs,
    weight and bias gradient one streaming pass (wgrad_skinny)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x2 = x.reshape(-1, W.shape[1])          # a permuted input is gathered once, here; backward reuses the copy
        ctx.save_for_backward(x2, W)
        ctx.has_bias = b is not None
        ctx.x_shape = x.shape
        y = torch.empty(x.shape[:-1] + (W.shape[0],), dtype=x.dtype, device=x.device)   # a base tensor: callers may consume it in place
        if b is None:
            torch.mm(x2, W.t(), out=y.view(-1, W.shape[0]))
        else:
            torch.addmm(b, x2, W.t(), out=y.view(-1, W.shape[0]))
        return y

    @staticmethod
    def backward(ctx, g):
        x2, W = ctx.saved_tensors
        n_out, n_in = W.shape
        g2 = g.reshape(-1, n_out)
        if g2.stride(1) != 1:
            g2 = g2.contiguous()
        dx = torch.mm(g2, W).reshape(ctx.x_shape) if ctx.needs_input_grad[0] else None
        dW = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_db:
            if n_in <= SKINNY_MAX and n_in <= n_out:   # few inputs: dW [n_out][n_in] = g^T x, lanes over g's columns
                dW, db, _ = wgrad_skinny(x2, g2, transposed=True, colsum_x=want_db)
            else:                                      # few outputs: dW [n_out][n_in], lanes over x's columns
                dW, _, db = wgrad_skinny(g2, x2, transposed=False, colsum_s=want_db)
        return dx, dW, db


def linear_skinny(x, W, b=None):
    """F.linear for layers with at most SKINNY_MAX inputs or outputs (the position part of DHGN's semantic layer, the action and
    value heads); under autograd the weight / bias gradients use wgrad_skinny."""
    if torch.is_grad_enabled() and (W.requires_grad or (b is not None and b.requires_grad) or x.requires_grad) and x.is_cuda:
        return _SkinnyLinear.apply(x, W, b)
    return F.linear(x, W, b)


def _linear_fwd(x, W, b, out=None, relu=False, consume_addend=False, y_link=None):
    """x W^T + b; b is a bias (out,) or a full addend of the output's shape (the add rides in the GEMM epilogue, beta = 1);
    relu with a 1-D bias rides in the hipBLASLt epilogue as well (bit-identical to relu(linear)).
    consume_addend: the full addend b is a temporary the caller no longer needs -- the product accumulates INTO it
    (no copy of b into a fresh result, no extra relu output); the returned tensor is b itself."""
    x2 = x.reshape(-1, W.shape[1])
    oshape = x.shape[:-1] + (W.shape[0],)
    if out is None and x2.numel() and x2.stride(-1) == 1 and split_linear_ok(x2, W, mode=PROJ_MODE, shapes=UPDATE_SPLIT_SHAPES):
        # the update's Linear layers on the split-bf16 kernel: bias / full addend / ReLU in its epilogue
        if b is None or b.dim() == 1:
            bits = None
            if y_link is not None and relu and sign_bits_ok(W.shape[0], W.shape[1]):   # relu'(y) as bits for the consumer's backward
                bits = y_link.bits = torch.empty((x2.shape[0], W.shape[0] // 8), dtype=torch.uint8, device=x2.device)
            return split_linear(x2, W, b, relu, sign_bits=bits).view(oshape)
        if b.is_contiguous():
            b2 = b.view(-1, W.shape[0])
            if consume_addend:
                split_linear(x2, W, None, relu, out=b2, addend=b2)
                return b
            return split_linear(x2, W, None, relu, addend=b2).view(oshape)
    if consume_addend and b is not None and b.dim() > 1 and out is None and b.is_contiguous():
        y = b.view(-1, W.shape[0]).addmm_(x2, W.t())
        if relu:
            y.clamp_min_(0.0)
        return b
    if relu:
        if b is not None and b.dim() == 1 and out is None:
            return torch._addmm_activation(b, x2, W.t(), use_gelu=False).reshape(oshape)
        y = _linear_fwd(x, W, b)
        return torch.clamp_min(y, 0.0, out=out.view(y.shape) if out is not None else None)
    res = None
    if out is None:  # the result is allocated in its final shape: a base tensor, not a view (it may be consumed in place later)
        res = torch.empty(oshape, dtype=x.dtype, device=x.device)
        out = res.view(-1, W.shape[0])
    elif out.dim() != 2:  # caller's storage in the result's shape
        res = out
        out = out.view(-1, W.shape[0])
    if b is None:
        y = torch.mm(x2, W.t(), out=out)
    else:
        y = torch.addmm(b if b.dim() == 1 else b.reshape(-1, W.shape[0]), x2, W.t(), out=out)
    return res if res is not None else y.reshape(oshape)


class _Linear(torch.autograd.Function):
    """x W^T + b whose weight gradient runs in the split-K MFMA kernel (wgrad)."""

    @staticmethod
    def forward(ctx, x, W, b, relu, consume_addend, x_link=None, y_link=None):
        ctx.bias_kind = 0 if b is None else (1 if b.dim() == 1 else 2)
        ctx.b_shape = None if b is None else b.shape
        ctx.relu = relu
        ctx.x_link, ctx.y_link = x_link, (y_link if relu else None)
        inplace = bool(consume_addend and ctx.bias_kind == 2 and b.is_contiguous() and b.shape == x.shape[:-1] + (W.shape[0],))
        if inplace:
            ctx.mark_dirty(b)
        y = _linear_fwd(x, W, b, None, relu, inplace, ctx.y_link if any(ctx.needs_input_grad) else None)
        ctx.save_for_backward(x, W, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        db = None
        want_db = bool(ctx.bias_kind and ctx.needs_input_grad[2])
        if ctx.relu:  # relu'(pre-activation) from the saved output; with a 1-D bias its gradient comes out of the same pass
            if ctx.y_link is not None and ctx.y_link.db is not None:
                # the consumer's input-gradient GEMM applied relu' and summed the columns (ReluLink)
                if want_db and ctx.bias_kind == 1:
                    db = ctx.y_link.take().view(-1, W.shape[0]).sum(0)
            elif want_db and ctx.bias_kind == 1:
                g, db = relu_bwd_colsum(g, y)
            else:
                g = torch.ops.aten.threshold_backward(g, y, 0.0)
        g2 = g.reshape(-1, W.shape[0])
        x2 = x.reshape(-1, W.shape[1])
        dx = None
        if ctx.needs_input_grad[0]:
            fused = None
            if ctx.x_link is not None:
                bits = ctx.x_link.bits
                fused = input_grad_masked(g2, W, x2, W.shape[1], bits.view(x2.shape[0], -1) if bits is not None else None)
            if fused is not None:
                dx, ctx.x_link.db = fused
            else:
                dx = input_grad(g2, W)
            dx = dx.reshape(x.shape)
        dW = wgrad(g2, x2) if ctx.needs_input_grad[1] else None
        if want_db and db is None:
            db = g2.sum(0) if ctx.bias_kind == 1 else g.reshape(ctx.b_shape)
        return dx, dW, db, None, None, None, None


def input_grad(g2, W):
    """g2 W for g2 (rows, N), W (N, K): the input gradient of a Linear layer -- on the split-bf16 kernel (as g2 (W^T)^T) where it
    covers the shape, else the library"""
    if g2.dim() == 2 and g2.stride(-1) == 1 and g2.is_cuda and PROJ_MODE == "split_bf16" and (W.shape[1], W.shape[0]) in UPDATE_SPLIT_SHAPES:
        Wt = W.detach().t().contiguous()
        if split_linear_ok(g2, Wt, mode=PROJ_MODE, shapes=UPDATE_SPLIT_SHAPES):
            return split_linear(g2, Wt)
    return torch.mm(g2, W)


class ReluLink:
    """Hand-over of a ReLU layer's backward to the ONE layer that consumes its output y (the update's relu(Linear) -> Linear chains,
    DHGN/mappo_parallel.py:148-233).  Autograd runs threshold_backward and the bias sum as passes of their own over d y; here the
    consumer's input-gradient GEMM applies relu'(y) in its epilogue and sums the columns on the way out (input_grad_masked: y is the
    consumer's saved input).  The producer is built with `y_link=link`, the consumer with `x_link=link`; in backward the consumer
    stores the column sums in `db` and returns the MASKED gradient, and the producer, finding `db` set, skips its own pass.  When the
    consumer cannot fuse (shape, mode), `db` stays None and both sides behave as without a link.  y must have no other consumer:
    autograd would add the other gradient to a masked one."""
    __slots__ = ("db", "bits")

    def __init__(self):
        self.db = None
        self.bits = None     # relu'(y) as sign bits (rows, width / 8) uint8 when the producer's GEMM wrote them (sb_gemm_signs), else None

    def take(self):
        db, self.db = self.db, None
        return db


MASKED_GRAD_SHAPES = {(256, 128), (384, 128), (128, 384)}   # (inputs of the layer = columns of the gradient, outputs)
MASKED_GRAD_MIN_ROWS = 4096
RELU_BITS = os.environ.get("MAPPO_RELU_BITS", "1") != "0"   # A/B switch: off = the masked input gradients read the saved fp32 activations as mask
RELU_LINK = os.environ.get("MAPPO_RELU_LINK", "1") != "0"   # A/B switch (tools/ab_switch.py): off = the separate relu' / bias-sum passes


def input_grad_masked(g2, W, y, mask_cols, bits=None):
    """(g2 W) * (y > 0) on the first mask_cols columns, and the column sums of the result (include/mappo_ops.h sb_gemm_masked):
    the input gradient of a Linear layer (W: (outputs, inputs)) whose input y (rows, inputs) came out of a ReLU, with that ReLU's
    backward and the bias gradient behind it.  bits: (rows, inputs / 8) uint8 sign bits of y (sb_gemm_signs), read instead of y.
    -> (gradient (rows, inputs), sums (inputs,)), or None where the kernel does not apply."""
    n_out, n_in = W.shape
    if (not RELU_LINK or PROJ_MODE != "split_bf16" or (n_in, n_out) not in MASKED_GRAD_SHAPES or not g2.is_cuda or g2.dim() != 2 or y.dim() != 2
            or g2.shape[0] < MASKED_GRAD_MIN_ROWS or y.shape != (g2.shape[0], n_in) or g2.dtype != torch.float32 or y.dtype != torch.float32):
        return None
    Wt = W.detach().t().contiguous()
    for t in (g2, y):
        if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
            return None
    L = load_library()
    R = g2.shape[0]
    out = torch.empty((R, n_in), dtype=torch.float32, device=g2.device)
    cs = torch.empty(n_in, dtype=torch.float32, device=g2.device)
    ws = torch.empty(L.sb_gemm_masked_workspace(n_in), dtype=torch.uint8, device=g2.device)
    if bits is not None and RELU_BITS:
        assert bits.dtype == torch.uint8 and bits.dim() == 2 and bits.shape == (R, n_in // 8) and bits.stride(1) == 1
        _check(L.sb_gemm_masked_bits(R, n_in, n_out, _ptr(g2), g2.stride(0), _ptr(Wt), Wt.stride(0), _ptr(bits), bits.stride(0), int(mask_cols), _ptr(out),
                                     out.stride(0), _ptr(cs), _ptr(ws), _stream()), "sb_gemm_masked_bits")
        return out, cs
    _check(L.sb_gemm_masked(R, n_in, n_out, _ptr(g2), g2.stride(0), _ptr(Wt), Wt.stride(0), _ptr(y), y.stride(0), int(mask_cols), _ptr(out), out.stride(0),
                            _ptr(cs), _ptr(ws), _stream()), "sb_gemm_masked")
    return out, cs


def linear(x, W, b=None, out=None, relu=False, consume_addend=False, x_link=None, y_link=None):
    """F.linear(x, W, b), optionally followed by relu (W may be a column slice of a larger weight; b a bias or a full
    addend); under autograd the weight gradient uses wgrad.  out (no autograd): written in place.  consume_addend: see
    _linear_fwd (b must be a temporary: it becomes the result).  x_link / y_link: ReluLink."""
    if torch.is_grad_enabled() and (W.requires_grad or x.requires_grad or (b is not None and b.requires_grad)):
        assert out is None
        return _Linear.apply(x, W, b, relu, consume_addend, x_link, y_link)
    return _linear_fwd(x, W, b, out, relu, consume_addend)


_gemm_lib = None
_gemm_ws = {}


def load_gemm_library():
    """libmappo_gemm.so (include/mappo_gemm.h): the Linear GEMM with a strided output and a fused epilogue on hipBLASLt"""
    global _gemm_lib
    if _gemm_lib is None:
        path = _build.lib_path("libmappo_gemm.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        Lg = C.CDLL(path)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        Lg.mo_gemm_workspace_bytes.restype = i64
        Lg.mo_gemm_nt.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, vp, i64, i32, vp, i64, vp, i64, vp]
        _gemm_lib = Lg
    return _gemm_lib


def gemm_workspace(device):
    """hipBLASLt's scratch for gemm_nt: one per (device, host thread) -- a thread issues its GEMMs on one stream at a time (the trainer's
    loop, the background evaluator's actor), so two streams never share one.  It must exist BEFORE a tick program is captured
    (_RolloutState.__init__ calls this): memory allocated during capture belongs to that graph's private pool and must not be
    cached beyond it."""
    import threading
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), threading.get_ident())
    ws = _gemm_ws.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("gemm_nt: no workspace for this thread yet and the stream is capturing -- call ops.gemm_workspace(device) first")
        ws = _gemm_ws[key] = torch.empty(load_gemm_library().mo_gemm_workspace_bytes(), dtype=torch.uint8, device=device)
    return ws


def gemm_nt(x, W, bias=None, relu=False, out=None, addend=None):
    """out = act(x W^T + bias + addend) for 2-D row-major x (rows, K), W (N, K), out / addend (rows, N); every matrix may be a
    column block of a wider one (last stride 1, any row stride) -- the form torch's GEMM epilogue path cannot write.  No autograd."""
    Lg = load_gemm_library()
    _need_gpu(x, "gemm_nt")
    M, K = x.shape
    N = W.shape[0]
    W = W.detach()
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    for t in (x, W, out) + ((addend,) if addend is not None else ()):
        assert t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.data_ptr() % 16 == 0
    assert W.shape[1] == K and out.shape == (M, N) and (addend is None or addend.shape == (M, N))
    b = bias.detach().contiguous() if bias is not None else None
    ws = gemm_workspace(x.device)
    rc = Lg.mo_gemm_nt(M, N, K, _ptr(x), x.stride(0), _ptr(W), W.stride(0), _ptr(b), _ptr(addend), addend.stride(0) if addend is not None else 0,
                       int(bool(relu)), _ptr(out), out.stride(0), _ptr(ws), ws.numel(), _stream())
    if rc != 0:
        raise RuntimeError(f"mo_gemm_nt failed (code {rc})")
    return out


def block2d(t):
    """(rows, E) 2-D view of a dense tensor or of a column block of a dense matrix (see _vec_stride); never copies"""
    E = t.shape[-1]
    return t.as_strided((t.numel() // E, E), (_vec_stride(t), 1))


def _hop_gemm(x, W, b, out, sign_bits=None):
    """relu(x W^T + b) into a column block: the split-bf16 kernel where the update routes its Linear layers to it, else hipBLASLt.
    sign_bits: (rows, 16) uint8 block that receives relu'(result) as bits -> True when it was written (split kernel only)"""
    if split_linear_ok(x, W, out, mode=PROJ_MODE, shapes=UPDATE_SPLIT_SHAPES):
        bits = sign_bits if sign_bits is not None and sign_bits_ok(W.shape[0], W.shape[1]) else None
        split_linear(x, W, b, True, out=out, sign_bits=bits)
        return bits is not None
    gemm_nt(x, W, b, True, out=out)
    return False


class _FcraHop(torch.autograd.Function):
    """One hop of DHGN.fcra (DHGN/mappo_parallel.py:204-233):  h' = relu([agg | h] Wf^T + bf),  agg = relu(nb Wagg^T + bagg).
    The reference concatenates; rounds 1-2 ran the FCRA layer as two accumulating GEMMs followed by a separate ReLU pass (and its
    backward as threshold + bias-sum passes).  Here [agg | h] is ONE (rows, 2E) operand that is never assembled by a copy: the AGG
    GEMM writes its (bias + ReLU epilogue) result into the left half, the previous hop's FCRA GEMM wrote h into the right half
    (`cat`: that buffer; None on the first hop, whose h is copied in), so the layer is one K = 2E GEMM with bias + ReLU in its
    epilogue.  Backward: relu'/bias-gradient in one pass per layer (relu_bwd_colsum on column blocks), one GEMM for
    [d agg | d h], split-K MFMA weight gradients.  nb is stored data (no gradient)."""

    @staticmethod
    def forward(ctx, nb, h, cat, Wagg, bagg, Wf, bf, last, out, box, h_link=None, cat_bits=None):
        E = Wf.shape[0]
        rows = h.numel() // E
        nb2 = nb.reshape(rows, E)
        train = any(ctx.needs_input_grad) and out is None
        if cat is None:
            cat = torch.empty((rows, 2 * E), dtype=h.dtype, device=h.device)
            cat[:, E:].copy_(block2d(h))
        else:
            assert cat.shape == (rows, 2 * E) and cat.is_contiguous() and h.data_ptr() == cat.data_ptr() + E * cat.element_size()
        # relu' of this hop's operand [agg | h] as sign bits (rows, 2E / 8): the left half is written by the AGG GEMM here, the right half
        # was written by the previous hop's FCRA GEMM (cat_bits); the backward's masked input gradient reads them instead of cat
        if train and cat_bits is None:
            cat_bits = torch.empty((rows, 2 * E // 8), dtype=torch.uint8, device=h.device)
        ok_left = _hop_gemm(nb2, Wagg, bagg, cat[:, :E], cat_bits[:, :E // 8] if train else None)
        nxt = nxt_bits = y_bits = None
        if out is not None:     # rollout: the caller's static storage (no autograd)
            ctx.mark_dirty(out)
            dst = block2d(out)
        elif last:
            dst = torch.empty((rows, E), dtype=h.dtype, device=h.device)
            y_bits = torch.empty((rows, E // 8), dtype=torch.uint8, device=h.device) if train else None
        else:
            nxt = torch.empty((rows, 2 * E), dtype=h.dtype, device=h.device)
            dst = nxt[:, E:]
            if train:
                nxt_bits = torch.empty((rows, 2 * E // 8), dtype=torch.uint8, device=h.device)
                y_bits = nxt_bits[:, E // 8:]
        ok_y = _hop_gemm(cat, Wf, bf, dst, y_bits)
        box.append(nxt)
        ctx.save_for_backward(nb2, Wagg, Wf)
        # the two operand buffers are kept as plain references: the next hop writes the OTHER half of `nxt` in place, which bumps
        # the version counter the saved view `dst` shares with it although its own columns are never touched again
        ctx.cat, ctx.y = cat, dst
        ctx.h_shape = h.shape
        # h_link: h is the previous hop's ReLU output (the right half of cat) and this hop its only consumer; y_link: the same
        # offer to whoever consumes this hop's output
        ctx.h_link, ctx.y_link = h_link, ReluLink()
        box.append(ctx.y_link)
        # usable in backward when every masked half has its bits: the left one from this hop, the right one (if masked) from the previous hop
        ctx.cat_bits = cat_bits if train and ok_left and (h_link is None or h_link.bits is not None) else None
        if ok_y:
            ctx.y_link.bits = y_bits
        box.append(nxt_bits if ok_y else None)
        return dst.reshape(h.shape) if out is None else out

    @staticmethod
    def backward(ctx, g):
        nb2, Wagg, Wf = ctx.saved_tensors
        cat, y = ctx.cat, ctx.y
        E = Wf.shape[0]
        rows = cat.shape[0]
        try:
            g2 = block2d(g)
        except AssertionError:
            g2 = g.reshape(rows, E).contiguous()
        if ctx.y_link.db is not None:                  # the consumer masked the gradient and summed its columns (ReluLink)
            gin, dbf = g2, ctx.y_link.take()
        else:
            gin, dbf = relu_bwd_colsum(g2, y)
        # [d agg | d h]; relu' of the AGG layer -- and of the previous hop's FCRA layer, whose output the right half of cat is --
        # in the GEMM's epilogue, the two bias gradients its column sums
        fused = input_grad_masked(gin, Wf, cat, 2 * E if ctx.h_link is not None else E, ctx.cat_bits)
        if fused is not None:
            dcat, cs = fused
            ga, dbagg = dcat[:, :E], cs[:E]
            if ctx.h_link is not None:
                ctx.h_link.db = cs[E:]
        else:
            dcat = input_grad(gin, Wf)
            ga, dbagg = relu_bwd_colsum(dcat[:, :E], cat[:, :E])
        dWf = wgrad(gin, cat)
        dWagg = wgrad(ga, nb2)
        d_h = dcat[:, E:].reshape(ctx.h_shape)         # a view (column block)
        return None, d_h, None, dWagg, dbagg, dWf, dbf, None, None, None, None, None


def fcra_hop(nb, h, carry, Wagg, bagg, Wf, bf, last, out=None):
    """-> (h', carry'): see _FcraHop.  carry (None for the first hop) is what a hop hands to the next one, its only consumer: the buffer
    whose right half h' is (None after the last hop / when `out` is given), the ReluLink of h', and the sign-bit block of that buffer;
    carry'[1] is also the link for whoever consumes the last hop's output."""
    cat, h_link, cat_bits = carry if carry is not None else (None, None, None)
    box = []
    y = _FcraHop.apply(nb, h, cat, Wagg, bagg, Wf, bf, bool(last), out, box, h_link, cat_bits)
    return y, (box[0], box[1], box[2])


FUSED_CELL_MIN_ROWS = 1024  # single-step batches at least this large take the fused cell kernel
class _PPOLoss(torch.autograd.Function):
    """(actor_loss, critic_loss) of one mini-batch; the gradients are computed in the forward launch and scaled here."""

    @staticmethod
    def forward(ctx, logp_now, entropy, values_now, logp_old, adv, active, values_old, v_target, epsilon, entropy_coef, use_value_clip):
        L = load_library()
        _need_gpu(logp_now, "ppo_loss")
        ts = [t.contiguous() for t in (logp_now, entropy, logp_old, adv, active, values_now, v_target)]
        vo = values_old.contiguous() if values_old is not None else None
        n = ts[0].numel()
        assert all(t.numel() == n and t.dtype == torch.float32 for t in ts)
        dev = logp_now.device
        asum = active.sum().reshape(1)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        g = torch.empty((3,) + tuple(logp_now.shape), dtype=torch.float32, device=dev)
        ws = torch.empty(L.ppo_loss_workspace(), dtype=torch.uint8, device=dev)
        _check(L.ppo_loss_fwd_bwd(n, _ptr(ts[0]), _ptr(ts[1]), _ptr(ts[2]), _ptr(ts[3]), _ptr(ts[4]), _ptr(ts[5]), _ptr(vo), _ptr(ts[6]),
                                  _ptr(asum), float(epsilon), float(entropy_coef), int(bool(use_value_clip)), _ptr(losses), _ptr(g[0]),
                                  _ptr(g[1]), _ptr(g[2]), _ptr(ws), _stream()), "ppo_loss_fwd_bwd")
        ctx.save_for_backward(g)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, ga, gc):
        (g,) = ctx.saved_tensors
        return g[0] * ga, g[1] * ga, g[2] * gc, None, None, None, None, None, None, None, None


class _PPOLossProb(torch.autograd.Function):
    """(actor_loss, critic_loss) of one mini-batch from the policy's probabilities: Categorical(prob).log_prob / .entropy() inside the
    loss launch (ppo_loss_prob_fwd_bwd), which also writes the gradient with respect to prob."""

    @staticmethod
    def forward(ctx, prob, values_now, action, logp_old, adv, active, values_old, v_target, epsilon, entropy_coef, use_value_clip):
        L = load_library()
        _need_gpu(prob, "ppo_loss_prob")
        A = prob.shape[-1]
        d0, d1, d2 = prob.shape[:3]
        n = d0 * d1 * d2
        ts = [t.contiguous() for t in (action, logp_old, adv, active, v_target)]
        vo = values_old.contiguous() if values_old is not None else None
        assert all(t.numel() == n and t.dtype == torch.float32 for t in ts) and values_now.shape == prob.shape[:3]
        dev = prob.device
        asum = active.sum().reshape(1)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        g_prob = torch.empty_strided(prob.shape, prob.stride(), dtype=torch.float32, device=dev)    # the layout of prob (a time-major view)
        g_v = torch.empty(values_now.shape, dtype=torch.float32, device=dev)
        ws = torch.empty(L.ppo_loss_workspace(), dtype=torch.uint8, device=dev)
        ps, vs = prob.stride(), values_now.stride()
        _check(L.ppo_loss_prob_fwd_bwd(n, A, _ptr(prob), _ptr(g_prob), d1, d2, ps[0], ps[1], ps[2], _ptr(ts[0]), _ptr(ts[1]), _ptr(ts[2]), _ptr(ts[3]),
                                       _ptr(values_now), vs[0], vs[1], vs[2], _ptr(vo), _ptr(ts[4]), _ptr(asum), float(epsilon), float(entropy_coef),
                                       int(bool(use_value_clip)), _ptr(losses), _ptr(g_v), _ptr(ws), _stream()), "ppo_loss_prob_fwd_bwd")
        ctx.save_for_backward(g_prob, g_v)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, ga, gc):
        g_prob, g_v = ctx.saved_tensors
        return g_prob * ga, g_v * gc, None, None, None, None, None, None, None, None, None


PPO_FROM_PROB = os.environ.get("MAPPO_PPO_FROM_PROB", "1") != "0"   # A/B switch: off = torch.distributions.Categorical + ppo_loss


def ppo_loss_prob_ok(prob, values_now):
    return (PPO_FROM_PROB and prob.is_cuda and prob.dtype == torch.float32 and prob.dim() == 4 and prob.stride(3) == 1 and 1 <= prob.shape[-1] <= 16
            and values_now.dtype == torch.float32 and values_now.shape == prob.shape[:3] and prob.numel() > 0)


def ppo_loss_prob(prob, action, values_now, logp_old, adv, active, values_old, v_target, epsilon, entropy_coef, use_value_clip=True):
    """ppo_loss(Categorical(prob).log_prob(action), Categorical(prob).entropy(), ...) in one launch (DHGN/mappo_parallel.py:451-456,
    :692-706; csrc/mappo_ops.hip k_ppo_loss_prob).  prob (mb, T, P, A), values_now (mb, T, P): any strides over the first three
    dimensions (the heads' outputs are time-major views)."""
    return _PPOLossProb.apply(prob, values_now, action, logp_old, adv, active, values_old, v_target, epsilon, entropy_coef, use_value_clip)


def ppo_loss(logp_now, entropy, values_now, logp_old, adv, active, values_old, v_target, epsilon, entropy_coef, use_value_clip=True):
    """Masked-mean PPO policy loss and (clipped) value loss of a mini-batch, one launch with the gradients
    (DHGN/mappo_parallel.py:692-706; csrc/mappo_ops.hip k_ppo_loss)."""
    return _PPOLoss.apply(logp_now, entropy, values_now, logp_old, adv, active, values_old, v_target, epsilon, entropy_coef, use_value_clip)


class RecordItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_row_stride", C.c_int64), ("row_bytes", C.c_int32), ("i32_to_f32", C.c_int32)]


def rollout_record(pairs, raw=None, episode_return=None):
    """pairs: (src, dst) tensors; src dense (N, ...), dst a row-strided view of the same trailing shape (buffer[k][rows, t]).
    One launch for all of them (csrc/mappo_ops.hip k_rollout_record); int32 sources are converted to dst's float32."""
    L = load_library()
    n = len(pairs)
    arr = (RecordItem * max(n, 1))()
    N = pairs[0][0].shape[0]
    for k, (src, dst) in enumerate(pairs):
        _need_gpu(src, "rollout_record")
        assert src.is_contiguous() and src.shape == dst.shape and src.shape[0] == N and (N == 1 or dst[0].is_contiguous())
        conv = src.dtype == torch.int32 and dst.dtype == torch.float32
        assert conv or src.dtype == dst.dtype
        it = arr[k]
        it.src, it.dst = src.data_ptr(), dst.data_ptr()
        it.row_bytes = src[0].numel() * src.element_size()
        it.dst_row_stride = dst.stride(0) * dst.element_size()
        it.i32_to_f32 = int(conv)
    P = raw.shape[1] if raw is not None else 0
    _check(L.rollout_record(N, n, C.cast(arr, C.c_void_p), _ptr(raw), _ptr(episode_return), P, _stream()), "rollout_record")


PERSISTENT_GRU_MIN_T = 2  # sequences at least this long take the one-launch recurrence (H = 128)
# the persistent recurrences of the update: "fp32" = k_gru_seq_fwd2 / bwd2 (v_mfma_f32_16x16x4_f32), "split_bf16" = the same arithmetic
# from exact three-way bf16 operand splits (csrc/sb_gru_seq.hpp); interchangeable save layout.  Follows `runtime.matmul`.
SEQ_MODE = os.environ.get("MAPPO_GRU_SEQ", os.environ.get("MAPPO_MATMUL", "split_bf16"))


def _seq_fwd(L, arr, n, T, Bmax, H, agents):
    if SEQ_MODE == "split_bf16":
        _check(L.gru_seq_split_fwd_multi(n, C.cast(arr, C.c_void_p), T, Bmax, H, agents, _stream()), "gru_seq_split_fwd_multi")
    else:
        _check(L.gru_seq_fwd_multi(n, C.cast(arr, C.c_void_p), T, Bmax, H, agents, _stream()), "gru_seq_fwd_multi")


def _seq_bwd(L, arr, n, T, Bmax, H, agents):
    if SEQ_MODE == "split_bf16":
        _check(L.gru_seq_split_bwd_multi(n, C.cast(arr, C.c_void_p), T, Bmax, H, agents, _stream()), "gru_seq_split_bwd_multi")
    else:
        _check(L.gru_seq_bwd_multi(n, C.cast(arr, C.c_void_p), T, Bmax, H, agents, _stream()), "gru_seq_bwd_multi")


class _GRULayer(torch.autograd.Function):
    """One torch.nn.GRU layer over a sequence: the input projection is one fp32 MFMA GEMM (rocBLAS / hipBLASLt), the
    recurrence runs in the persistent kernel pair gru_seq_{fwd,bwd} (or, for short sequences / other sizes, a GEMM and the
    gate kernels per step).  x: (T, B, I) time-major, or -- agents = P > 0 -- the (T B, I) rows of the encoder's output in
    (episode, step, agent) order (sequence b = n P + p): the projection and its gradients then use the embedding as it lies
    in memory (no permuted copy on the way in, none for the gradient on the way out).  out: (T, B, H) time-major."""

    @staticmethod
    def forward(ctx, x, h0, w_ih, w_hh, b_ih, b_hh, T, B, agents):
        L = load_library()
        _need_gpu(x, "gru")
        I = x.shape[-1]
        H = w_hh.shape[1]
        persistent = bool(H == 128 and T >= PERSISTENT_GRU_MIN_T)
        ctx.x_shape = x.shape
        ctx.ungather = 0
        if agents and not persistent:  # the per-step path is time-major only: gather the rows once
            x = x.reshape(B // agents, T, agents, I).permute(1, 0, 2, 3).reshape(T, B, I)
            ctx.ungather = int(agents)
            agents = 0
        x = x.contiguous()
        h0 = h0.contiguous()
        need = any(ctx.needs_input_grad)
        out = torch.empty((T, B, H), dtype=x.dtype, device=x.device)
        gi = gru_input_projection(x.reshape(T * B, I), w_ih, b_ih)
        # the saved gates: (T, 4, B, H) planes for the per-step path; the persistent pair keeps them in its own lane order
        save = None
        if need:
            save = (torch.empty(L.gru_seq_save_elems(T, B), dtype=x.dtype, device=x.device) if persistent
                    else torch.empty((T, 4, B, H), dtype=x.dtype, device=x.device))
        b_hh = b_hh.contiguous()
        st = _stream()
        ctx.persistent, ctx.dims, ctx.agents = persistent, (T, B, I), int(agents)
        if persistent:  # whole recurrence in one launch, W_hh in registers (csrc/mappo_ops.hip k_gru_seq_fwd2 | csrc/sb_gru_seq.hpp)
            whh = w_hh.detach().contiguous()
            arr = (GruSeqNet * 1)()
            a = arr[0]
            a.gi, a.w_hh, a.b_hh, a.h0, a.out = gi.data_ptr(), whh.data_ptr(), b_hh.data_ptr(), h0.data_ptr(), out.data_ptr()
            a.save = save.data_ptr() if need else None
            a.B = B
            _seq_fwd(L, arr, 1, T, B, H, int(agents))
            if need:
                ctx.save_for_backward(x, h0, w_ih, whh, out, save)
            return out
        gi = gi.reshape(T, B, 3 * H)
        gh = torch.empty((B, 3 * H), dtype=x.dtype, device=x.device)
        w_hh_t = w_hh.t()
        hprev = h0
        for t in range(T):
            torch.mm(hprev, w_hh_t, out=gh)
            _check(L.gru_gates_fwd(B, H, _ptr(gi[t]), _ptr(gh), _ptr(b_hh), _ptr(hprev), _ptr(out[t]),
                                   _ptr(save[t]) if need else None, st), "gru_gates_fwd")
            hprev = out[t]
        if need:
            ctx.save_for_backward(x, h0, w_ih, w_hh, out, save)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = load_library()
        x, h0, w_ih, w_hh, out, save = ctx.saved_tensors
        T, B, I = ctx.dims
        H = w_hh.shape[1]
        dout = dout.contiguous()
        dgi = torch.empty((T * B, 3 * H), dtype=x.dtype, device=x.device)   # rows in x's order
        dh_direct = torch.empty((B, H), dtype=x.dtype, device=x.device)
        dcarry = None
        st = _stream()
        db_ih = db_hh = None
        # dW_hh = [dr dz dnr]^T h_prev.  Time-major dgi rows line up with h_prev = out[t - 1], so dr, dz are stored once (dgi)
        # and only dnr separately; with the encoder's row order (agents) the kernel writes the full time-major dgh as well.
        split = bool(ctx.persistent and not ctx.agents)
        dgh = None if split else torch.empty((T, B, 3 * H), dtype=x.dtype, device=x.device)
        dnr = torch.empty((T, B, H), dtype=x.dtype, device=x.device) if split else None
        if ctx.persistent:
            db_ih = torch.empty(3 * H, dtype=x.dtype, device=x.device)
            db_hh = torch.empty(3 * H, dtype=x.dtype, device=x.device)
            ws = torch.empty(L.gru_seq_bwd_workspace(B), dtype=torch.uint8, device=x.device)
            arr = (GruSeqBwdNet * 1)()
            a = arr[0]
            a.dout, a.save, a.out, a.h0, a.w_hh, a.dgi = dout.data_ptr(), save.data_ptr(), out.data_ptr(), h0.data_ptr(), w_hh.data_ptr(), dgi.data_ptr()
            a.dgh = dgh.data_ptr() if dgh is not None else None
            a.dnr = dnr.data_ptr() if dnr is not None else None
            a.dh0, a.db_ih, a.db_hh, a.workspace = dh_direct.data_ptr(), db_ih.data_ptr(), db_hh.data_ptr(), ws.data_ptr()
            a.B = B
            _seq_bwd(L, arr, 1, T, B, H, ctx.agents)
            dcarry = dh_direct
        else:
            dgi3 = dgi.view(T, B, 3 * H)
            for t in range(T - 1, -1, -1):
                hprev = out[t - 1] if t > 0 else h0
                _check(L.gru_gates_bwd(B, H, _ptr(dout[t]), _ptr(dcarry), _ptr(save[t]), _ptr(hprev), _ptr(dgi3[t]), _ptr(dgh[t]),
                                       _ptr(dh_direct), st), "gru_gates_bwd")
                dcarry = torch.addmm(dh_direct, dgh[t], w_hh)
        dw_hh = _gru_dw_hh(dgi, dgh, dnr, out, h0, T, B, H)
        x2 = x.reshape(T * B, I)
        dw_ih = wgrad(dgi, x2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = input_grad(dgi, w_ih)
            if ctx.ungather:   # back to the caller's (episode, step, agent) rows
                P = ctx.ungather
                dx = dx.reshape(T, B // P, P, I).permute(1, 0, 2, 3)
            dx = dx.reshape(ctx.x_shape)
        if db_ih is None:
            db_ih, db_hh = dgi.sum(0), dgh.reshape(T * B, 3 * H).sum(0)
        return dx, dcarry, dw_ih, dw_hh, db_ih, db_hh, None, None, None


class GruCellNet(C.Structure):
    """include/mappo_ops.h mo_gru_cell_net"""
    _fields_ = [(n, C.c_void_p) for n in ("x", "h_prev", "w_ih", "w_hh", "b_ih", "b_hh", "h_out")]


# The rollout's GRU cells: "fp32" = v_mfma_f32_16x16x4_f32 (k_gru_cell), "split_bf16" = the same fp32 arithmetic from exact three-way
# bf16 splits of the operands on v_mfma_f32_16x16x32_bf16 (k_gru_cell_sb, include/mappo_ops.h gru_cell_split_fwd_multi): same
# results to fp32 rounding, 2.67 x the matrix rate.  `runtime.gru_cell` (MAPPO.__init__) or set_cell_mode().
CELL_MODE = os.environ.get("MAPPO_GRU_CELL", MATMUL_MODE)     # the rollout: GRU cells and Linear layers


def set_cell_mode(mode):
    global CELL_MODE
    if mode not in ("fp32", "split_bf16"):
        raise ValueError(f"matmul mode {mode!r}: 'fp32' or 'split_bf16'")
    CELL_MODE = mode


def matmul_modes():
    """the five switches set_matmul_mode() sets together (tests and A/B tools save and restore them)"""
    return MATMUL_MODE, WGRAD_MODE, PROJ_MODE, CELL_MODE, SEQ_MODE


def restore_matmul_modes(modes):
    global MATMUL_MODE, WGRAD_MODE, PROJ_MODE, CELL_MODE, SEQ_MODE
    MATMUL_MODE, WGRAD_MODE, PROJ_MODE, CELL_MODE, SEQ_MODE = modes


def set_matmul_mode(mode):
    """all three parts at once (`runtime.matmul`)"""
    global MATMUL_MODE, WGRAD_MODE, PROJ_MODE, SEQ_MODE
    set_cell_mode(mode)
    MATMUL_MODE = WGRAD_MODE = PROJ_MODE = SEQ_MODE = mode


SPLIT_LINEAR_SHAPES = {(128, 128), (128, 256), (128, 384), (256, 128), (384, 128)}    # (outputs, inputs) sb_gemm covers
# the update routes all of them as well (5e5-row operands: the 384-input kernel is matrix + vector bound there, 398 us against the
# library's 420-484 us, worth 1 ms per iteration; the 128-input kernels run at HBM speed, 2 x the library)
UPDATE_SPLIT_SHAPES = SPLIT_LINEAR_SHAPES


def split_linear_ok(x, W, out=None, addend=None, mode=None, shapes=None):
    """shapes and strides sb_gemm covers (fp32, rows 16-byte aligned, no autograd); mode: the switch that governs the caller
    (default CELL_MODE, the rollout's kernels)"""
    if ((CELL_MODE if mode is None else mode) != "split_bf16" or torch.is_grad_enabled() or not x.is_cuda
            or tuple(W.shape) not in (SPLIT_LINEAR_SHAPES if shapes is None else shapes)):
        return False
    for t in (x, W) + ((out,) if out is not None else ()) + ((addend,) if addend is not None else ()):
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
            return False
    return x.shape[1] == W.shape[1]


def split_linear(x, W, bias=None, relu=False, out=None, addend=None, sign_bits=None):
    """out = act(x W^T + bias + addend) for 2-D x (rows, K), W (N, K) with (N, K) in SPLIT_LINEAR_SHAPES, out / addend (rows, N), any
    row strides, in fp32 arithmetic on the bf16 matrix pipe (exact three-way operand splits, include/mappo_ops.h sb_gemm).  The
    rollout's Linear layers and the update's GRU input projections; no autograd.  addend may be out (accumulate in place)."""
    L = load_library()
    R, K = x.shape
    N = W.shape[0]
    W = W.detach()
    if out is None:
        out = torch.empty((R, N), dtype=torch.float32, device=x.device)
    b = bias.detach() if bias is not None else None
    if sign_bits is not None:   # (rows, N / 8) uint8, possibly a column block of a wider byte matrix: relu'(out) for the backward (sb_gemm_signs)
        assert sign_bits.dtype == torch.uint8 and sign_bits.shape == (R, N // 8) and sign_bits.stride(1) == 1 and sign_bits_ok(N, K)
        _check(L.sb_gemm_signs(R, N, K, _ptr(x), x.stride(0), _ptr(W), W.stride(0), _ptr(b), int(bool(relu)), _ptr(addend),
                               addend.stride(0) if addend is not None else 0, _ptr(out), out.stride(0), _ptr(sign_bits), sign_bits.stride(0), _stream()),
               "sb_gemm_signs")
        return out
    _check(L.sb_gemm(R, N, K, _ptr(x), x.stride(0), _ptr(W), W.stride(0), _ptr(b), int(bool(relu)), _ptr(addend),
                     addend.stride(0) if addend is not None else 0, _ptr(out), out.stride(0), _stream()), "sb_gemm")
    return out


def sign_bits_ok(N, K):
    """the shapes whose split-bf16 GEMM can write relu'(result) as bits beside the result (ReluLink.bits)"""
    return RELU_LINK and RELU_BITS and N == 128 and K in (128, 256)


# the update's Linear layers, their input gradients and the GRU input projections gi = x W_ih^T + b_ih: the BLAS library | sb_gemm
PROJ_MODE = os.environ.get("MAPPO_PROJ", MATMUL_MODE)


def gru_input_projection(x2, w_ih, b_ih):
    """gi (rows, 384) = x2 w_ih^T + b_ih"""
    if x2.stride(-1) == 1 and split_linear_ok(x2, w_ih, mode=PROJ_MODE, shapes=UPDATE_SPLIT_SHAPES):
        return split_linear(x2, w_ih, b_ih)
    return torch.addmm(b_ih, x2, w_ih.t())


def gru_step_multi(xs, hiddens, modules, hiddens_out=None):
    """One rollout step of several independent GRU modules of one shape (actor and critic) without autograd: layer by layer, the
    cells of all modules in ONE launch.  xs: (B, 128) each; hiddens: per module (num_layers, B, 128).  hiddens_out None: each state
    is updated IN PLACE (gru_cell_fwd_multi: every 16-row tile is read before it is written by the one workgroup that owns it).
    hiddens_out given (same shapes, other storage): the new state is written there and `hiddens` is left as it was -- the form
    the split-bf16 cell needs (CELL_MODE; two workgroups share a row tile).  Returns the top layer's output per module (views of the
    new states).  Other shapes: ops.gru per module."""
    n = len(xs)
    B = xs[0].shape[0]
    outs_h = hiddens if hiddens_out is None else hiddens_out
    ok = (not torch.is_grad_enabled() and n <= 4 and B >= FUSED_CELL_MIN_ROWS and all(x.shape == (B, 128) and x.is_cuda for x in xs)
          and all(h.is_contiguous() and h.shape[1:] == (B, 128) for h in list(hiddens) + list(outs_h))
          and all(m.num_layers == modules[0].num_layers and m.weight_hh_l0.shape == (384, 128) and m.weight_ih_l0.shape == (384, 128) for m in modules))
    if not ok:
        outs = []
        for x, h, ho, m in zip(xs, hiddens, outs_h, modules):
            out, hn = gru(x.unsqueeze(0), h, m, inplace_hidden=hiddens_out is None)
            if hn is not ho:
                ho.copy_(hn)          # the per-step path returns a fresh state: keep the contract of this function
            outs.append(ho[-1])
        return outs
    L = load_library()
    split = CELL_MODE == "split_bf16" and hiddens_out is not None
    fn, what = (L.gru_cell_split_fwd_multi, "gru_cell_split_fwd_multi") if split else (L.gru_cell_fwd_multi, "gru_cell_fwd_multi")
    inps = [x.contiguous() for x in xs]
    for layer in range(modules[0].num_layers):
        arr = (GruCellNet * n)()
        keep = []
        for k, m in enumerate(modules):
            ws = [getattr(m, f"{nm}_l{layer}").detach().contiguous() for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            a = arr[k]
            a.x, a.h_prev, a.h_out = inps[k].data_ptr(), hiddens[k][layer].data_ptr(), outs_h[k][layer].data_ptr()
            a.w_ih, a.w_hh, a.b_ih, a.b_hh = (w.data_ptr() for w in ws)
            keep.append(ws)
        _check(fn(n, C.cast(arr, C.c_void_p), B, 128, _stream()), what)
        inps = [outs_h[k][layer] for k in range(n)]
    return inps


GRU_MULTI_MAX_WORKGROUPS = 256   # CUs of an MI355X: one persistent workgroup each
GRU_MULTI_MAX_NETS = 24          # include/mappo_ops.h MO_GRU_MAX_NETS


class GruSeqNet(C.Structure):
    """include/mappo_ops.h mo_gru_seq_net"""
    _fields_ = [(n, C.c_void_p) for n in ("gi", "w_hh", "b_hh", "h0", "out", "save")] + [("B", C.c_int32), ("pad0", C.c_int32)]


class GruSeqBwdNet(C.Structure):
    """include/mappo_ops.h mo_gru_seq_bwd_net"""
    _fields_ = ([(n, C.c_void_p) for n in ("dout", "save", "out", "h0", "w_hh", "dgi", "dgh", "dnr", "dh0", "db_ih", "db_hh", "workspace")]
                + [("B", C.c_int32), ("pad0", C.c_int32)])


class _GRULayerMulti(torch.autograd.Function):
    """The same layer of several independent GRUs (the actor's and the critic's, for one or several mini-batches: own weights, own
    inputs, own number of sequences; same T and row order) with the recurrences of all of them in ONE persistent launch each way
    (gru_seq_fwd_multi / gru_seq_bwd_multi); per network the arithmetic is that of _GRULayer's persistent path.
    args: T, Bs (sequences per network), agents, x_links (None, or per network the ReluLink of an input that came out of a ReLU and
    feeds this layer only -- its relu' and bias sum then ride in the input gradient), zero_h0 (the caller vouches that every h0 is all
    zeros -- the update's sequences start from the zero state: the first step then adds nothing to dW_hh), then per network
    (x, h0, w_ih, w_hh, b_ih, b_hh)."""

    @staticmethod
    def forward(ctx, T, Bs, agents, x_links, zero_h0, *ts):
        L = load_library()
        n = len(ts) // 6
        H = ts[3].shape[1]
        assert H == 128 and T >= PERSISTENT_GRU_MIN_T and len(ts) == 6 * n and len(Bs) == n
        need = any(ctx.needs_input_grad)
        outs, saved = [], []
        arr = (GruSeqNet * n)()
        keep = []
        for k in range(n):
            x, h0, w_ih, w_hh, b_ih, b_hh = ts[6 * k: 6 * k + 6]
            _need_gpu(x, "gru")
            I, B = x.shape[-1], Bs[k]
            x, h0 = x.contiguous(), h0.contiguous()
            out = torch.empty((T, B, H), dtype=x.dtype, device=x.device)
            gi = gru_input_projection(x.reshape(T * B, I), w_ih, b_ih)
            save = torch.empty(L.gru_seq_save_elems(T, B), dtype=x.dtype, device=x.device) if need else None
            whh, bhh = w_hh.detach().contiguous(), b_hh.detach().contiguous()
            a = arr[k]
            a.gi, a.w_hh, a.b_hh, a.h0, a.out = gi.data_ptr(), whh.data_ptr(), bhh.data_ptr(), h0.data_ptr(), out.data_ptr()
            a.save = save.data_ptr() if need else None
            a.B = B
            keep.append((gi, whh, bhh))
            outs.append(out)
            saved += [x, h0, w_ih, whh, out, save]
        _seq_fwd(L, arr, n, T, max(Bs), H, int(agents))
        ctx.dims, ctx.agents, ctx.n = (T, tuple(Bs)), int(agents), n
        ctx.x_links, ctx.zero_h0 = x_links, bool(zero_h0)
        ctx.x_shapes = [ts[6 * k].shape for k in range(n)]
        if need:
            ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        L = load_library()
        T, Bs = ctx.dims
        n = ctx.n
        sv = ctx.saved_tensors
        H = 128
        split = not ctx.agents      # time-major dgi rows: dr, dz stored once (see _GRULayer.backward)
        arr = (GruSeqBwdNet * n)()
        per = []
        for k in range(n):
            x, h0, w_ih, w_hh, out, save = sv[6 * k: 6 * k + 6]
            dev, dt, B = x.device, x.dtype, Bs[k]
            dout = douts[k].contiguous()
            dgi = torch.empty((T * B, 3 * H), dtype=dt, device=dev)
            dgh = None if split else torch.empty((T, B, 3 * H), dtype=dt, device=dev)
            dnr = torch.empty((T, B, H), dtype=dt, device=dev) if split else None
            dh0 = torch.empty((B, H), dtype=dt, device=dev)
            db_ih, db_hh = torch.empty(3 * H, dtype=dt, device=dev), torch.empty(3 * H, dtype=dt, device=dev)
            ws = torch.empty(L.gru_seq_bwd_workspace(B), dtype=torch.uint8, device=dev)
            a = arr[k]
            a.dout, a.save, a.out, a.h0, a.w_hh, a.dgi = dout.data_ptr(), save.data_ptr(), out.data_ptr(), h0.data_ptr(), w_hh.data_ptr(), dgi.data_ptr()
            a.dgh = dgh.data_ptr() if dgh is not None else None
            a.dnr = dnr.data_ptr() if dnr is not None else None
            a.dh0, a.db_ih, a.db_hh, a.workspace = dh0.data_ptr(), db_ih.data_ptr(), db_hh.data_ptr(), ws.data_ptr()
            a.B = B
            per.append((dout, dgi, dgh, dnr, dh0, db_ih, db_hh, ws))
        _seq_bwd(L, arr, n, T, max(Bs), H, ctx.agents)
        grads = [None, None, None, None, None]
        for k in range(n):
            x, h0, w_ih, w_hh, out, save = sv[6 * k: 6 * k + 6]
            dout, dgi, dgh, dnr, dh0, db_ih, db_hh, ws = per[k]
            B = Bs[k]
            dw_hh = _gru_dw_hh(dgi, dgh, dnr, out, h0, T, B, H, ctx.zero_h0)
            I = x.shape[-1]
            x2 = x.reshape(T * B, I)
            dw_ih = wgrad(dgi, x2)
            dx = None
            if ctx.needs_input_grad[5 + 6 * k]:
                link = ctx.x_links[k] if ctx.x_links is not None else None
                fused = input_grad_masked(dgi, w_ih, x2, I, link.bits) if link is not None else None
                if fused is not None:
                    dx, link.db = fused
                else:
                    dx = input_grad(dgi, w_ih)
                dx = dx.reshape(ctx.x_shapes[k])
            grads += [dx, dh0, dw_ih, dw_hh, db_ih, db_hh]
        return tuple(grads)


def _gru_dw_hh(dgi, dgh, dnr, out, h0, T, B, H, zero_h0=False):
    """dW_hh = [dr dz dnr]^T h_prev over all steps: from the full time-major dgh, or (dr, dz stored once) from dgi's first 2H
    columns and dnr.  zero_h0: h0 is known to be all zeros (the first step's term vanishes: two small library products less)."""
    if dgh is not None:
        dw_hh = torch.mm(dgh[0].t(), h0)
        if T > 1:
            wgrad(dgh[1:].reshape((T - 1) * B, 3 * H), out[:-1].reshape((T - 1) * B, H), out=dw_hh, accumulate=True)
        return dw_hh
    dw_hh = torch.empty((3 * H, H), dtype=dgi.dtype, device=dgi.device)
    drz, hp = dgi[:, :2 * H], out[:-1].reshape((T - 1) * B, H)     # (dr, dz): a column slice of dgi, rows strided, no copy
    acc = True
    if zero_h0 and T > 1:
        acc = False                                                # the products below write dw_hh instead of adding to the first step's
    elif zero_h0:
        dw_hh.zero_()
    else:
        torch.mm(drz[:B].t(), h0, out=dw_hh[:2 * H])
        torch.mm(dnr[0].t(), h0, out=dw_hh[2 * H:])
    if T > 1:
        a1, a2 = drz[B:], dnr[1:].reshape((T - 1) * B, H)
        if WGRAD_MODE == "split_bf16" and H == 128 and _wgrad_ok(a1, hp) and _wgrad_ok(a2, hp):
            # [dr dz | dnr]^T h_prev in ONE pass over h_prev (k_sb_wgrad with its left operand in two tensors)
            L = load_library()
            ws = torch.empty(L.wgrad_split_workspace(3 * H, H), dtype=torch.uint8, device=dgi.device)
            _check(L.wgrad_split_tn2(a1.shape[0], 2 * H, H, H, _ptr(a1), a1.stride(0), _ptr(a2), a2.stride(0), _ptr(hp), hp.stride(0), _ptr(dw_hh), int(acc),
                                     _ptr(ws), _stream()), "wgrad_split_tn2")
        else:
            wgrad(a1, hp, out=dw_hh[:2 * H], accumulate=acc)
            wgrad(a2, hp, out=dw_hh[2 * H:], accumulate=acc)
    return dw_hh


def gru_multi(xs, h0s, modules, agents=0, steps=None, grouped=False, x_links=None, zero_state=False):
    """ops.gru for several independent GRU modules of one architecture (actor and critic, for one or -- `grouped` -- several
    mini-batches: the inputs may differ in their number of sequences): layer by layer, the recurrences of all of them in one launch
    (see _GRULayerMulti).  `modules` need only carry num_layers and the weight_* / bias_* attributes of torch.nn.GRU.  Returns the
    list of outputs (T, B_k, H) (no h_n: sequences start from the given h0 and the final state is out[-1]).  Shapes the persistent
    kernels do not cover take ops.gru per module.  x_links: per input its ReluLink or None; zero_state: every h0 is all zeros (see
    _GRULayerMulti)."""
    n = len(xs)
    if agents:
        assert xs[0].dim() == 2 and steps and all(x.shape[0] % (steps * agents) == 0 for x in xs)
        T, Bs = int(steps), [x.shape[0] // int(steps) for x in xs]
    else:
        T, Bs = xs[0].shape[0], [x.shape[1] for x in xs]
    tiles = sum((B + 15) // 16 for B in Bs)
    # two layers of ONE mini-batch share a launch only while their workgroups (16 rows each) are resident at once: at the full
    # benchmark mini-batch (205 + 205 workgroups on 256 CUs) the second layer's would queue behind the first's anyway and the extra
    # concurrency only adds HBM contention (measured: update +13 ms at 4096 environments, -16 ms at 512).  A group of mini-batches
    # (MAPPO.train, small batches) is launched together whatever its size: that is what fills the chip.
    ok = (n <= GRU_MULTI_MAX_NETS and T >= PERSISTENT_GRU_MIN_T and all(x.shape[-1] == xs[0].shape[-1] and x.dim() == xs[0].dim() for x in xs)
          and (grouped or (tiles <= GRU_MULTI_MAX_WORKGROUPS and all(B == Bs[0] for B in Bs)))
          and all(m.num_layers == modules[0].num_layers and m.weight_hh_l0.shape == (384, 128) for m in modules))
    if not ok:
        return [gru(x, h0, m, agents=agents, steps=steps)[0] for x, h0, m in zip(xs, h0s, modules)]
    inps = list(xs)
    for layer in range(modules[0].num_layers):
        ts = []
        for k, m in enumerate(modules):
            ts += [inps[k], h0s[k][layer], getattr(m, f"weight_ih_l{layer}"), getattr(m, f"weight_hh_l{layer}"),
                   getattr(m, f"bias_ih_l{layer}"), getattr(m, f"bias_hh_l{layer}")]
        links = tuple(x_links) if layer == 0 and x_links is not None and any(l is not None for l in x_links) else None
        inps = list(_GRULayerMulti.apply(T, tuple(Bs), int(agents) if layer == 0 else 0, links, bool(zero_state), *ts))
    return inps


def gru(x, h0, gru_module, inplace_hidden=False, agents=0, steps=None):
    """torch.nn.GRU(x, h0) semantics (seq-first, unidirectional, no dropout) on the fused path.
    x (T, B, I), h0 (num_layers, B, H) -> out (T, B, H), h_n (num_layers, B, H).
    agents = P > 0: x is instead the (B T, I) rows of the encoder's output in (episode, step, agent) order with T = steps
    (sequence b = n P + p); the first layer reads them in place (see _GRULayer), the result is time-major all the same.
    inplace_hidden (single rollout step only): the new hidden state overwrites h0 (each 16-row tile is read before it is
    written by the one workgroup that owns it) and h_n IS h0 -- no stack, no copy back into the rollout state."""
    hn = []
    inp = x
    if agents:
        assert x.dim() == 2 and steps and x.shape[0] % (steps * agents) == 0
        T, B = int(steps), x.shape[0] // int(steps)
    else:
        T, B = x.shape[0], x.shape[1]
    for layer in range(gru_module.num_layers):
        w_ih, w_hh = getattr(gru_module, f"weight_ih_l{layer}"), getattr(gru_module, f"weight_hh_l{layer}")
        b_ih, b_hh = getattr(gru_module, f"bias_ih_l{layer}"), getattr(gru_module, f"bias_hh_l{layer}")
        if T == 1 and not torch.is_grad_enabled() and w_hh.shape[1] == 128 and w_ih.shape[1] == 128 and B >= FUSED_CELL_MIN_ROWS:
            # rollout step: both projections + gates in one persistent launch (csrc/mappo_ops.hip k_gru_cell)
            _need_gpu(inp, "gru")
            inplace = inplace_hidden and h0.is_contiguous()
            out = h0[layer].unsqueeze(0) if inplace else torch.empty((1, B, 128), dtype=inp.dtype, device=inp.device)
            _check(load_library().gru_cell_fwd(B, 128, _ptr(inp.contiguous()), _ptr(h0[layer].contiguous()), _ptr(w_ih.detach().contiguous()),
                                               _ptr(w_hh.detach().contiguous()), _ptr(b_ih.detach().contiguous()),
                                               _ptr(b_hh.detach().contiguous()), _ptr(out), _stream()), "gru_cell_fwd")
            inp = out
        else:
            inp = _GRULayer.apply(inp, h0[layer], w_ih, w_hh, b_ih, b_hh, T, B, int(agents) if layer == 0 else 0)
        hn.append(inp[-1])
    if all(t.data_ptr() == h0[k].data_ptr() for k, t in enumerate(hn)):
        return inp, h0
    return inp, torch.stack(hn, 0)
