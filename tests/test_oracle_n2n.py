"""env_n2n oracle (oracle/n2n_oracle.c + reset restatement) against goldens captured from the reference.  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import n2n_oracle as no
from tests.helpers import GOLDEN

FILES = sorted(glob.glob(os.path.join(GOLDEN, "n2n_*.npz")))


def load(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["seed"], d["P"], d["E"], d["T"] = [int(v) for v in d["meta"]]
    return d


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_n2n_trace_matches_reference(path):
    d = load(path)
    cfg = no.make_cfg(d["P"], d["E"], d["T"])
    env = no.OracleN2n(cfg, d["p0"], d["e0"], d["target"])
    for t in range(len(d["done"])):
        assert np.array_equal(env.p, d["p"][t]) and np.array_equal(env.e, d["e"][t]), t        # f64 state bit for bit
        ps, es, pp, pe = env.observe()
        assert np.array_equal(pp, d["pp_adj"][t].astype(np.float32)) and np.array_equal(pe, d["pe_adj"][t].astype(np.float32)), t
        env.evader_step(d["e_cmd"][t])
        r, done, act = env.step(d["action"][t])
        assert np.array_equal(r, d["reward"][t]) and np.array_equal(act, d["active"][t]) and done == bool(d["done"][t]), t
    assert np.array_equal(env.p, d["p_end"]) and np.array_equal(env.e, d["e_end"])


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_n2n_reset_restatement(path):
    d = load(path)
    np.random.seed(d["seed"])
    target, p, e = no.reset_oracle(d["P"], d["E"])
    assert np.array_equal(target, d["target"]) and np.array_equal(p, d["p0"]) and np.array_equal(e, d["e0"])
