"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/libe3d_oracle.so (e3d_oracle.c) plus the numpy-RNG
restatement of ParticleEnv.reset (reference environment/env_3d/particle_env.py:137-203)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libe3d_oracle.so")
MAX_DRAWS = 100000  # include/e3d_env.h E3D_RESET_MAX_DRAWS: bound on the reference's unbounded placement loop


class E3dCfg(C.Structure):
    _fields_ = [("P", C.c_int32), ("max_step", C.c_int32)] + \
               [(n, C.c_double) for n in ("p_vmax", "e_vmax", "p_sen_range", "p_comm_range", "kill_radius", "ang_lmt", "v_lmt", "step_size")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "e3d_oracle.c")):
            subprocess.check_call(["make", "-C", HERE, "libe3d_oracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(os.environ.get("DMARL_E3D_ORACLE_LIB") or LIB)
        vp = C.c_void_p
        L.e3d_evader_step.argtypes = [vp, vp, vp]
        L.e3d_step.argtypes = [vp] * 8
        L.e3d_step.restype = C.c_int
        L.e3d_observe.argtypes = [vp] * 7
        _lib = L
    return _lib


def make_cfg(P, max_step=200, p_vmax=0.7, e_vmax=1.0, p_sen_range=3.0, p_comm_range=6.0, kill_radius=0.5, ang_lmt=np.pi / 4, v_lmt=0.4,
             step_size=0.5):
    c = E3dCfg()
    c.P, c.max_step = P, max_step
    c.p_vmax, c.e_vmax, c.p_sen_range, c.p_comm_range = p_vmax, e_vmax, p_sen_range, p_comm_range
    c.kill_radius, c.ang_lmt, c.v_lmt, c.step_size = kill_radius, ang_lmt, v_lmt, step_size
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleE3d:
    """state rows: x, y, z, phi, gamma, v, active"""

    def __init__(self, cfg, p, e, target):
        self.c = cfg
        self.p = np.ascontiguousarray(p, np.float64).copy()
        self.e = np.ascontiguousarray(e, np.float64).reshape(1, 7).copy()
        self.target = np.ascontiguousarray(target, np.float64).copy()
        self.t = C.c_int32(0)

    def observe(self):
        P = self.c.P
        ps, es = np.zeros((P, 6), np.float32), np.zeros((1, 6), np.float32)
        pp, pe = np.zeros((P, P), np.float32), np.zeros((P, 1), np.float32)
        lib().e3d_observe(C.byref(self.c), _p(self.p), _p(self.e), _p(ps), _p(es), _p(pp), _p(pe))
        return ps, es, pp, pe

    def evader_step(self, cmd):
        cmd = np.ascontiguousarray(cmd, np.float64).reshape(3)
        lib().e3d_evader_step(C.byref(self.c), _p(self.e), _p(cmd))

    def step(self, action):
        a = np.ascontiguousarray(action, np.float64).reshape(self.c.P, 3)
        r = np.zeros(self.c.P); act = np.zeros(self.c.P, np.uint8)
        done = lib().e3d_step(C.byref(self.c), _p(self.p), _p(self.e), _p(self.target), _p(a), C.byref(self.t), _p(r), _p(act))
        return r, bool(done), act


class ResetFailed(RuntimeError):
    pass


def reset_oracle(P, nprnd=np.random):
    """ParticleEnv.reset (:137-203): target, pursuers ~ N(10, 2).clip(5, 15) at least 4 apart, evader at 20 - target; every
    heading / pitch uniform; numpy global RNG, draws in the reference's order."""
    target = [nprnd.rand() * 20, nprnd.rand() * 20, nprnd.rand() * 20]
    pts, draws = [], 0
    while len(pts) < P:
        draws += 1
        if draws > MAX_DRAWS:
            raise ResetFailed("gen_init_p_pos")
        newp = nprnd.normal(loc=10, scale=2, size=(3,)).clip(5, 15)
        if not any(np.linalg.norm(newp - q) < 4 for q in pts):
            pts.append(newp)
    p = np.zeros((P, 7))
    for i in range(P):
        p[i] = [pts[i][0], pts[i][1], pts[i][2], (2 * nprnd.rand() - 1) * np.pi, (2 * nprnd.rand() - 1) * np.pi / 2, 0.0, 1.0]
    e = np.array([[20 - target[0], 20 - target[1], 20 - target[2], (2 * nprnd.rand() - 1) * np.pi, (2 * nprnd.rand() - 1) * np.pi / 2, 0.0, 1.0]])
    return np.asarray(target), p, e
