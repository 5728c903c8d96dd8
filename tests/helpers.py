"""Shared test helpers: golden loading and the oracle replay of an environment trace."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def trace_files(pattern="env_trace_*.npz"):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load_trace(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    seed, P, W, H, blocks, variance, T = [int(v) for v in d["meta"]]
    d.update(seed=seed, P=P, W=W, H=H, blocks=blocks, variance=variance, T=T)
    n_obs = int(d["n_obs"])
    d["o_adj"] = np.unpackbits(d["o_adj"], axis=-1)[..., :n_obs]
    if "raser" in d:
        d["raser"] = np.unpackbits(d["raser"], axis=-1)[..., :n_obs].reshape(W, H, n_obs)
    d["tape"] = d["drawn_targets"][1:]
    return d


def padded_tape(d, tape_len):
    tape = np.zeros((tape_len, 2), np.int32)
    k = min(len(d["tape"]), tape_len)
    tape[:k] = d["tape"][:k]
    return tape
