"""Persistent GRU layer of the update (T = 150): forward and backward kernel times of the fp32-MFMA recurrences (k_gru_seq_fwd2 / bwd2)
and of the split-bf16 ones (csrc/sb_gru_seq.hpp), one layer of one mini-batch (B = 3280 sequences: 205 workgroups) and the form the
update launches (all ten mini-batches and both networks in one launch: 20 records, 4 096 workgroups)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_multi_agent_reinforcement_learning_amd import ops
L = ops.load_library()
T, H = 150, 128
dev = "cuda"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def records(Bs):
    keep, fa, ba = [], (ops.GruSeqNet * len(Bs))(), (ops.GruSeqBwdNet * len(Bs))()
    for k, B in enumerate(Bs):
        gi = torch.randn(T, B, 3 * H, device=dev); w = torch.randn(3 * H, H, device=dev) * 0.08; b = torch.zeros(3 * H, device=dev)
        h0 = torch.zeros(B, H, device=dev); out = torch.empty(T, B, H, device=dev); save = torch.empty(L.gru_seq_save_elems(T, B), device=dev)
        dout = torch.randn(T, B, H, device=dev); dgi = torch.empty(T, B, 3 * H, device=dev); dnr = torch.empty(T, B, H, device=dev)
        dh0 = torch.empty(B, H, device=dev); dbi = torch.empty(3 * H, device=dev); dbh = torch.empty(3 * H, device=dev)
        ws = torch.empty(L.gru_seq_bwd_workspace(B), dtype=torch.uint8, device=dev)
        keep.append((gi, w, b, h0, out, save, dout, dgi, dnr, dh0, dbi, dbh, ws))
        a = fa[k]
        a.gi, a.w_hh, a.b_hh, a.h0, a.out, a.save, a.B = gi.data_ptr(), w.data_ptr(), b.data_ptr(), h0.data_ptr(), out.data_ptr(), save.data_ptr(), B
        a = ba[k]
        a.dout, a.save, a.out, a.h0, a.w_hh, a.dgi, a.dgh, a.dnr = dout.data_ptr(), save.data_ptr(), out.data_ptr(), h0.data_ptr(), w.data_ptr(), dgi.data_ptr(), None, dnr.data_ptr()
        a.dh0, a.db_ih, a.db_hh, a.workspace, a.B = dh0.data_ptr(), dbi.data_ptr(), dbh.data_ptr(), ws.data_ptr(), B
    return keep, fa, ba


for label, Bs in (("one layer, 3280 sequences", [3280]), ("20 layers of 3280 / 3248 sequences (the grouped update)", [3280] * 18 + [3248] * 2),
                  ("20 layers of 416 sequences (512 environments per rank)", [416] * 18 + [352] * 2)):
    keep, fa, ba = records(Bs)
    n, Bmax, rows = len(Bs), max(Bs), sum(Bs)
    fl = 2.0 * T * rows * H * 3 * H
    for mode, ff, fb in (("fp32 ", L.gru_seq_fwd_multi, L.gru_seq_bwd_multi), ("split", L.gru_seq_split_fwd_multi, L.gru_seq_split_bwd_multi)):
        tf = timeit(lambda: ff(n, C.cast(fa, C.c_void_p), T, Bmax, H, 0, st))
        tb = timeit(lambda: fb(n, C.cast(ba, C.c_void_p), T, Bmax, H, 0, st))
        bf, bb = 4096.0 * rows * T, 5120.0 * rows * T
        print(f"{label}: {mode} fwd {tf:9.1f} us ({fl / tf / 1e6:6.1f} TFLOP/s fp32-equivalent, {bf / tf / 1e6:5.2f} TB/s)   bwd {tb:9.1f} us ({fl / tb / 1e6:6.1f} TFLOP/s, "
              f"{bb / tb / 1e6:5.2f} TB/s)")
    del keep
    torch.cuda.empty_cache()
