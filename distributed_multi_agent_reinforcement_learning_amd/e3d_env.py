"""`ParticleEnv` (env_3d): the reference's continuous 3-D pursuit environment batched on the GPU.

Mirrors environment/env_3d/particle_env.py:76-404 of the reference for N independent environments: `initialize`, `reset`,
`evader_step`, `step`, `get_team_state`, `get_adj_mat`, `get_active` keep their names; tensors carry a leading environment
dimension.  Pursuer actions are continuous, a in [-1, 1]^3 (heading, pitch, speed; Point.step :25-55).  Kinematics / reward /
culling / done run in csrc/e3d_env.hip (C ABI include/e3d_env.h), the reset in the same library's host part with a replica
of numpy's legacy generator per environment.  The reference's evader is driven by scipy's SLSQP (eva.py:87-148): here its
command is an input (`evader_step(cmd)`), or, when none is given, a closed-form rule (head for the target at full speed) that
is NOT the reference's optimiser.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import build as _build


class E3dConfig(C.Structure):
    _fields_ = [("P", C.c_int32), ("max_step", C.c_int32)] + \
               [(n, C.c_double) for n in ("p_vmax", "e_vmax", "p_sen_range", "p_comm_range", "kill_radius", "ang_lmt", "v_lmt", "step_size")]


class E3dState(C.Structure):
    _fields_ = [("N", C.c_int32), ("pad0", C.c_int32)] + [(n, C.c_void_p) for n in ("p", "e", "target", "time_step")]


class E3dObsOut(C.Structure):
    _fields_ = [("p_state", C.c_void_p), ("p_state_stride", C.c_int64), ("e_state", C.c_void_p), ("e_state_stride", C.c_int64),
                ("pp_adj", C.c_void_p), ("pp_adj_stride", C.c_int64), ("pe_adj", C.c_void_p), ("pe_adj_stride", C.c_int64)]


_lib = None


def load_library():
    global _lib
    if _lib is None:
        path = _build.lib_path("libe3d_env.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with __graft_entry__.build(); env_3d has no CPU fallback")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.e3d_config_check.argtypes = [vp]
        L.e3d_env_load.argtypes = [vp] * 6
        L.e3d_env_observe.argtypes = [vp] * 4
        L.e3d_env_tick.argtypes = [vp] * 9
        L.e3d_resetter_create.argtypes = [vp, C.c_int32, vp]
        L.e3d_resetter_create.restype = vp
        L.e3d_resetter_destroy.argtypes = [vp]
        L.e3d_resetter_reset.argtypes = [vp, vp, vp, vp, C.c_int32]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        extra = " (no pursuer placement within E3D_RESET_MAX_DRAWS draws: too many pursuers for the 10^3 box at distance 4)" if rc == 40003 else ""
        raise RuntimeError(f"{what} failed with code {rc}{extra}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class ParticleEnv:
    """cfg values are the reference's hard-coded defaults (particle_env.py:78-121)."""

    def __init__(self, num_envs=1, seeds=None, device="cuda", p_vmax=0.7, e_vmax=1.0, p_sen_range=3.0, p_comm_range=6.0, kill_radius=0.5,
                 ang_lmt=math.pi / 4, v_lmt=0.4, step_size=0.5, max_step=200):
        self.L = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("ParticleEnv needs a GPU (MI355X); there is no CPU path")
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.seeds = list(seeds) if seeds is not None else list(range(self.num_envs))
        self.p_obs_dim = self.e_obs_dim = 6
        self.state_dim, self.action_dim = 12, 3
        self.env_name = "ParticleEnvBoundGra"
        self.max_step, self.step_size, self.kill_radius = max_step, step_size, kill_radius
        self._kw = dict(p_vmax=p_vmax, e_vmax=e_vmax, p_sen_range=p_sen_range, p_comm_range=p_comm_range, kill_radius=kill_radius,
                        ang_lmt=ang_lmt, v_lmt=v_lmt, step_size=step_size)
        self.time_step = 0
        self.n_episode = 0
        self.p_num = None
        self.e_num = 1
        self.resetter = None

    def initialize(self, p_num):
        """particle_env.py:133-135, plus the allocation of the device records"""
        self.p_num = int(p_num)
        c = E3dConfig()
        c.P, c.max_step = self.p_num, self.max_step
        for k, v in self._kw.items():
            setattr(c, k, v)
        _check(self.L.e3d_config_check(C.byref(c)), "e3d_config_check")
        self.c = c
        N, dev, P = self.num_envs, self.device, self.p_num
        self.p = torch.zeros((N, 7, P), dtype=torch.float64, device=dev)
        self.e = torch.zeros((N, 7), dtype=torch.float64, device=dev)
        self.target = torch.zeros((N, 3), dtype=torch.float64, device=dev)
        self.t_dev = torch.zeros((N,), dtype=torch.int32, device=dev)
        self.st = E3dState()
        self.st.N = N
        self.st.p, self.st.e, self.st.target, self.st.time_step = self.p.data_ptr(), self.e.data_ptr(), self.target.data_ptr(), self.t_dev.data_ptr()
        f = lambda *s: torch.zeros((N, *s), dtype=torch.float32, device=dev)
        self.obs = dict(p_state=f(P, 6), e_state=f(1, 6), pp_adj=f(P, P), pe_adj=f(P, 1))
        self.reward_t = f(P)
        self.active_t = torch.ones((N, P), dtype=torch.uint8, device=dev)
        self.done_t = torch.zeros((N,), dtype=torch.uint8, device=dev)
        s = np.ascontiguousarray(self.seeds, np.uint32)
        self.resetter = self.L.e3d_resetter_create(C.byref(c), N, s.ctypes.data_as(C.c_void_p))
        if not self.resetter:
            raise RuntimeError("e3d_resetter_create failed (bad configuration or out of memory)")
        self._obs_struct = E3dObsOut()
        for k, t in self.obs.items():
            setattr(self._obs_struct, k, t.data_ptr())
            setattr(self._obs_struct, k + "_stride", t.stride(0))

    def __del__(self):
        try:
            if self.resetter:
                self.L.e3d_resetter_destroy(self.resetter)
        except Exception:
            pass

    def reset(self, init=None):
        """particle_env.py:137-203.  init = (p [N,P,7], e [N,7], target [N,3]) injects recorded initial conditions."""
        N = self.num_envs
        if init is None:
            p = np.empty((N, self.p_num, 7)); e = np.empty((N, 7)); tg = np.empty((N, 3))
            _check(self.L.e3d_resetter_reset(self.resetter, p.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                             tg.ctypes.data_as(C.c_void_p), min(16, os.cpu_count() or 1)), "e3d_resetter_reset")
        else:
            p, e, tg = (np.ascontiguousarray(a, np.float64) for a in init)
            e = e.reshape(N, 7)
        self.last_init = (p, e, tg)
        _check(self.L.e3d_env_load(C.byref(self.c), C.byref(self.st), p.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                   tg.ctypes.data_as(C.c_void_p), _stream()), "e3d_env_load")
        torch.cuda.current_stream().synchronize()
        self.time_step = 0
        self.n_episode += 1
        self._cmd = torch.zeros((N, 3), dtype=torch.float64, device=self.device)
        self.active_t.fill_(1)
        self.observe()

    def observe(self):
        _check(self.L.e3d_env_observe(C.byref(self.c), C.byref(self.st), C.byref(self._obs_struct), _stream()), "e3d_env_observe")
        return self.obs

    def get_team_state(self, is_pursuer, rules=False):
        """(N, A, 6) [x, y, z, phi, gamma, v] of every agent (the reference's rules=False form, :258-265)"""
        return self.obs["p_state"] if is_pursuer else self.obs["e_state"]

    def get_adj_mat(self, which="pp"):
        """:328-340 with the pursuers as observers: 'pp' (communication range) or 'pe' (sensing range)"""
        return self.obs["pp_adj" if which == "pp" else "pe_adj"]

    def get_active(self):
        return self.active_t

    def evader_step(self, cmd=None):
        """Sets the evader's command (heading, pitch, speed) in [-1, 1]^3 for the next step (the reference computes it with
        SLSQP, :354-378).  Without `cmd`: full speed straight at the target."""
        if cmd is None:
            d = self.target - self.e[:, :3]
            cmd = torch.stack((torch.atan2(d[:, 1], d[:, 0]) / math.pi, torch.atan2(d[:, 2], torch.hypot(d[:, 0], d[:, 1])) / (math.pi / 2),
                               torch.ones_like(d[:, 0])), -1)
        self._cmd = torch.as_tensor(cmd, dtype=torch.float64, device=self.device).reshape(self.num_envs, 3).contiguous()

    def step(self, action):
        """:205-219 (preceded by the evader's move with the command of evader_step) -> (reward (N,P), done (N,), active (N,P));
        action (N, P, 3) in [-1, 1]"""
        a = torch.as_tensor(action, device=self.device).to(torch.float64).reshape(self.num_envs, self.p_num, 3).contiguous()
        ptr = lambda t: C.c_void_p(t.data_ptr())
        _check(self.L.e3d_env_tick(C.byref(self.c), C.byref(self.st), ptr(a), ptr(self._cmd), ptr(self.reward_t), ptr(self.active_t),
                                   ptr(self.done_t), C.byref(self._obs_struct), _stream()), "e3d_env_tick")
        self.time_step += 1
        return self.reward_t, self.done_t, self.active_t
