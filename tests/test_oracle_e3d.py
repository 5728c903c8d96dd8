"""env_3d oracle (oracle/e3d_oracle.c + reset restatement) against goldens captured from the reference.  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import e3d_oracle as eo
from tests.helpers import GOLDEN

FILES = sorted(glob.glob(os.path.join(GOLDEN, "e3d_*.npz")))


def load(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["seed"], d["P"], d["E"], d["T"] = [int(v) for v in d["meta"]]
    return d


def test_fixture_set_covers_the_domain_events():
    ds = [load(p) for p in FILES]
    assert len(ds) >= 6
    assert any(d["e_end"][0, 6] == 0 for d in ds)                                   # an evader capture
    assert any(d["active"][-1].sum() < d["P"] for d in ds)                           # pursuers lost to collisions
    assert any(len(d["done"]) < d["T"] and d["e_end"][0, 6] == 1 for d in ds)        # the evader reached its target
    assert any(len(d["done"]) == d["T"] for d in ds)                                 # an episode that ran to max_step


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_e3d_trace_matches_reference(path):
    d = load(path)
    cfg = eo.make_cfg(d["P"], d["T"])
    env = eo.OracleE3d(cfg, d["p0"], d["e0"], d["target"])
    for t in range(len(d["done"])):
        assert np.array_equal(env.p, d["p"][t]) and np.array_equal(env.e, d["e"][t]), t        # f64 state bit for bit
        ps, es, pp, pe = env.observe()
        assert np.array_equal(pp, d["pp_adj"][t].astype(np.float32)) and np.array_equal(pe, d["pe_adj"][t].astype(np.float32)), t
        assert np.array_equal(ps, d["p"][t][:, :6].astype(np.float32)) and np.array_equal(es, d["e"][t][:, :6].astype(np.float32))
        if d["e"][t][0, 6] > 0 and d["p"][t][:, 6].sum() > 0:
            env.evader_step(d["e_cmd"][t][0])
        r, done, act = env.step(d["action"][t])
        assert np.array_equal(r, d["reward"][t]) and np.array_equal(act, d["active"][t]) and done == bool(d["done"][t]), t
    assert np.array_equal(env.p, d["p_end"]) and np.array_equal(env.e, d["e_end"])


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_e3d_reset_restatement(path):
    d = load(path)
    np.random.seed(d["seed"])
    target, p, e = eo.reset_oracle(d["P"])
    assert np.array_equal(target, d["target"]) and np.array_equal(p, d["p0"]) and np.array_equal(e, d["e0"])
