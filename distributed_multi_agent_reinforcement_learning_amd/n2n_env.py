"""`ParticleEnv` (env_n2n): the reference's continuous 2-D pursuit environment batched on the GPU.

Mirrors environment/env_n2n/particle_env.py:105-462 of the reference for N independent environments: `initialize`,
`reset`, `evader_step`, `step`, `get_team_state`, `get_adj_mat`, `get_active` keep their names; tensors carry a leading
environment dimension.  The kinematics / reward / culling / done logic runs in csrc/n2n_env.hip (C ABI include/n2n_env.h),
the reset in the same library's host part with a replica of numpy's legacy generator per environment.
The reference's evader is driven by scipy's SLSQP (eva.py:36-53): here the heading command is an input
(`evader_step(cmd)`), or, when none is given, a simple closed-form rule (head for the target, turn away from the nearest
pursuer in sensing range) that is NOT the reference's optimiser.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import build as _build


class N2nConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("P", "E", "episode_limit", "pad0")] + \
               [(n, C.c_double) for n in ("p_vmax", "e_vmax", "p_sen_range", "p_comm_range", "kill_radius", "ang_lmt", "step_size")]


class N2nState(C.Structure):
    _fields_ = [("N", C.c_int32), ("pad0", C.c_int32)] + [(n, C.c_void_p) for n in ("p", "e", "target", "time_step")]


class N2nObsOut(C.Structure):
    _fields_ = [("p_state", C.c_void_p), ("p_state_stride", C.c_int64), ("e_state", C.c_void_p), ("e_state_stride", C.c_int64),
                ("pp_adj", C.c_void_p), ("pp_adj_stride", C.c_int64), ("pe_adj", C.c_void_p), ("pe_adj_stride", C.c_int64)]


_lib = None


def load_library():
    global _lib
    if _lib is None:
        path = _build.lib_path("libn2n_env.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with __graft_entry__.build(); env_n2n has no CPU fallback")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.n2n_config_check.argtypes = [vp]
        L.n2n_env_load.argtypes = [vp] * 6
        L.n2n_env_observe.argtypes = [vp] * 4
        L.n2n_env_tick.argtypes = [vp] * 9
        L.n2n_resetter_create.argtypes = [vp, C.c_int32, vp]
        L.n2n_resetter_create.restype = vp
        L.n2n_resetter_destroy.argtypes = [vp]
        L.n2n_resetter_reset.argtypes = [vp, vp, vp, vp, C.c_int32]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class ParticleEnv:
    """cfg values are the reference's hard-coded defaults (particle_env.py:108-121,143-147)."""

    def __init__(self, num_envs=1, seeds=None, device="cuda", p_vmax=0.3, e_vmax=1.0, p_sen_range=3.0, p_comm_range=6.0,
                 kill_radius=0.5, ang_lmt=math.pi / 4, step_size=0.5, episode_limit=100):
        self.L = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("ParticleEnv needs a GPU (MI355X); there is no CPU path")
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.seeds = list(seeds) if seeds is not None else list(range(self.num_envs))
        self.p_obs_dim = self.e_obs_dim = 3
        self.env_name = "ParticleEnvBoundGra"
        self.episode_limit, self.step_size, self.kill_radius = episode_limit, step_size, kill_radius
        self._kw = dict(p_vmax=p_vmax, e_vmax=e_vmax, p_sen_range=p_sen_range, p_comm_range=p_comm_range, kill_radius=kill_radius,
                        ang_lmt=ang_lmt, step_size=step_size)
        self.time_step = 0
        self.n_episode = 0
        self.p_num = self.e_num = None
        self.resetter = None

    def initialize(self, p_num, e_num):
        """particle_env.py:160-162, plus the allocation of the device records"""
        self.p_num, self.e_num = int(p_num), int(e_num)
        c = N2nConfig()
        c.P, c.E, c.episode_limit = self.p_num, self.e_num, self.episode_limit
        for k, v in self._kw.items():
            setattr(c, k, v)
        _check(self.L.n2n_config_check(C.byref(c)), "n2n_config_check")
        self.c = c
        N, dev = self.num_envs, self.device
        self.p = torch.zeros((N, 5, self.p_num), dtype=torch.float64, device=dev)
        self.e = torch.zeros((N, 5, self.e_num), dtype=torch.float64, device=dev)
        self.target = torch.zeros((N, 2), dtype=torch.float64, device=dev)
        self.t_dev = torch.zeros((N,), dtype=torch.int32, device=dev)
        self.st = N2nState()
        self.st.N = N
        self.st.p, self.st.e, self.st.target, self.st.time_step = self.p.data_ptr(), self.e.data_ptr(), self.target.data_ptr(), self.t_dev.data_ptr()
        f = lambda *s: torch.zeros((N, *s), dtype=torch.float32, device=dev)
        self.obs = dict(p_state=f(self.p_num, 3), e_state=f(self.e_num, 3), pp_adj=f(self.p_num, self.p_num), pe_adj=f(self.p_num, self.e_num))
        self.reward_t = f(self.p_num)
        self.active_t = torch.ones((N, self.p_num), dtype=torch.uint8, device=dev)
        self.done_t = torch.zeros((N,), dtype=torch.uint8, device=dev)
        s = np.ascontiguousarray(self.seeds, np.uint32)
        self.resetter = self.L.n2n_resetter_create(C.byref(c), N, s.ctypes.data_as(C.c_void_p))
        if not self.resetter:
            raise RuntimeError("n2n_resetter_create failed (bad configuration or out of memory)")
        self._obs_struct = N2nObsOut()
        for k, t in self.obs.items():
            setattr(self._obs_struct, k, t.data_ptr())
            setattr(self._obs_struct, k + "_stride", t.stride(0))

    def __del__(self):
        try:
            if self.resetter:
                self.L.n2n_resetter_destroy(self.resetter)
        except Exception:
            pass

    def reset(self, init=None):
        """particle_env.py:200-237.  init = (p [N,P,5], e [N,E,5], target [N,2]) injects recorded initial conditions."""
        N = self.num_envs
        if init is None:
            p = np.empty((N, self.p_num, 5)); e = np.empty((N, self.e_num, 5)); tg = np.empty((N, 2))
            _check(self.L.n2n_resetter_reset(self.resetter, p.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                             tg.ctypes.data_as(C.c_void_p), min(16, os.cpu_count() or 1)), "n2n_resetter_reset")
        else:
            p, e, tg = (np.ascontiguousarray(a, np.float64) for a in init)
        self.last_init = (p, e, tg)
        _check(self.L.n2n_env_load(C.byref(self.c), C.byref(self.st), p.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                   tg.ctypes.data_as(C.c_void_p), _stream()), "n2n_env_load")
        torch.cuda.current_stream().synchronize()
        self.time_step = 0
        self.n_episode += 1
        self._cmd = torch.zeros((N, self.e_num), dtype=torch.float64, device=self.device)
        self.observe()

    def observe(self):
        _check(self.L.n2n_env_observe(C.byref(self.c), C.byref(self.st), C.byref(self._obs_struct), _stream()), "n2n_env_observe")
        return self.obs

    def get_team_state(self, is_pursuer, rules=False):
        """(N, A, 3) [x, y, phi] of every agent (the reference's rules=False form, :376-384)"""
        return self.obs["p_state"] if is_pursuer else self.obs["e_state"]

    def get_adj_mat(self, which="pp"):
        """:386-397 with the pursuers as observers: 'pp' (communication range) or 'pe' (sensing range)"""
        return self.obs["pp_adj" if which == "pp" else "pe_adj"]

    def get_active(self):
        return self.active_t

    def evader_step(self, cmd=None):
        """Sets the evaders' normalised heading command in [-1, 1] for the next step (the reference computes it with SLSQP,
        :179-198).  Without `cmd`: head for the target, but away from the nearest pursuer inside the sensing range."""
        if cmd is None:
            ex, ey = self.e[:, 0], self.e[:, 1]
            to_t = torch.atan2(self.target[:, 1:2] - ey, self.target[:, 0:1] - ex)
            dx, dy = ex[:, :, None] - self.p[:, 0][:, None, :], ey[:, :, None] - self.p[:, 1][:, None, :]
            d = torch.sqrt(dx * dx + dy * dy)
            dmin, imin = d.min(-1)
            away = torch.atan2(torch.gather(dy, 2, imin[..., None])[..., 0], torch.gather(dx, 2, imin[..., None])[..., 0])
            cmd = torch.where(dmin <= self._kw["p_sen_range"], away, to_t) / math.pi
        self._cmd = torch.as_tensor(cmd, dtype=torch.float64, device=self.device).reshape(self.num_envs, self.e_num).contiguous()

    def step(self, action):
        """:164-177 (preceded by the evader's move with the command of evader_step) -> (reward (N,P), done (N,), active (N,P))"""
        a = torch.as_tensor(action, device=self.device).to(torch.int32).reshape(self.num_envs, self.p_num).contiguous()
        ptr = lambda t: C.c_void_p(t.data_ptr())
        _check(self.L.n2n_env_tick(C.byref(self.c), C.byref(self.st), ptr(a), ptr(self._cmd), ptr(self.reward_t), ptr(self.active_t),
                                   ptr(self.done_t), C.byref(self._obs_struct), _stream()), "n2n_env_tick")
        self.time_step += 1
        return self.reward_t, self.done_t, self.active_t
