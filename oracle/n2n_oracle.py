"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/libn2n_oracle.so (n2n_oracle.c) plus the numpy-RNG
restatement of ParticleEnv.reset (reference environment/env_n2n/particle_env.py:200-281)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libn2n_oracle.so")


class N2nCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("P", "E", "episode_limit", "pad0")] + \
               [(n, C.c_double) for n in ("p_vmax", "e_vmax", "p_sen_range", "p_comm_range", "kill_radius", "ang_lmt", "step_size")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "n2n_oracle.c")):
            subprocess.check_call(["make", "-C", HERE, "libn2n_oracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(os.environ.get("DMARL_N2N_ORACLE_LIB") or LIB)  # override: the sanitizer build (tools/sanitize_host.py)
        vp = C.c_void_p
        L.n2n_evader_step.argtypes = [vp, vp, vp]
        L.n2n_step.argtypes = [vp] * 8
        L.n2n_step.restype = C.c_int
        L.n2n_observe.argtypes = [vp] * 7
        _lib = L
    return _lib


def make_cfg(P, E, episode_limit=100, p_vmax=0.3, e_vmax=1.0, p_sen_range=3.0, p_comm_range=6.0, kill_radius=0.5,
             ang_lmt=np.pi / 4, step_size=0.5):
    c = N2nCfg()
    c.P, c.E, c.episode_limit = P, E, episode_limit
    c.p_vmax, c.e_vmax, c.p_sen_range, c.p_comm_range = p_vmax, e_vmax, p_sen_range, p_comm_range
    c.kill_radius, c.ang_lmt, c.step_size = kill_radius, ang_lmt, step_size
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleN2n:
    def __init__(self, cfg, p, e, target):
        self.c = cfg
        self.p = np.ascontiguousarray(p, np.float64).copy()
        self.e = np.ascontiguousarray(e, np.float64).copy()
        self.target = np.ascontiguousarray(target, np.float64).copy()
        self.t = C.c_int32(0)

    def observe(self):
        P, E = self.c.P, self.c.E
        ps, es = np.zeros((P, 3), np.float32), np.zeros((E, 3), np.float32)
        pp, pe = np.zeros((P, P), np.float32), np.zeros((P, E), np.float32)
        lib().n2n_observe(C.byref(self.c), _p(self.p), _p(self.e), _p(ps), _p(es), _p(pp), _p(pe))
        return ps, es, pp, pe

    def evader_step(self, cmd):
        cmd = np.ascontiguousarray(cmd, np.float64)
        lib().n2n_evader_step(C.byref(self.c), _p(self.e), _p(cmd))

    def step(self, action):
        a = np.ascontiguousarray(action, np.int32)
        r = np.zeros(self.c.P); act = np.zeros(self.c.P, np.uint8)
        done = lib().n2n_step(C.byref(self.c), _p(self.p), _p(self.e), _p(self.target), _p(a), C.byref(self.t), _p(r), _p(act))
        return r, bool(done), act


def reset_oracle(P, E, e_vmax=1.0, nprnd=np.random):
    """ParticleEnv.reset (:200-281): target, pursuers around (10, 10), evaders around (20, 20) - target; numpy global RNG."""
    target = [nprnd.rand() * 20, nprnd.rand() * 20]

    def sample(n, loc, lo, hi):
        pts = []
        while len(pts) < n:
            newp = nprnd.normal(loc=loc, scale=2, size=(2,)).clip(lo, hi)
            if not any(np.linalg.norm(newp - q) < 2 for q in pts):
                pts.append(newp)
        return pts
    pp = sample(P, 0, -8, 8)
    p = np.zeros((P, 5))
    for i in range(P):
        p[i] = [pp[i][0] + 10, pp[i][1] + 10, np.pi / 4, 0.0, 1.0]
    ep = sample(E, np.array([20 - target[0], 20 - target[1]]), 0, 20)
    e = np.zeros((E, 5))
    for i in range(E):
        e[i] = [ep[i][0], ep[i][1], np.pi / 4, e_vmax, 1.0]
    return np.asarray(target), p, e
