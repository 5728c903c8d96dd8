"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/libpe_env_oracle.so (pe_env_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpe_env_oracle.so")
MAX_BEAMS = 64
MAX_PATH = 4096


class PeoConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("W", "H", "P", "O", "max_steps", "difficulty", "extend_dis", "num_beams",
                                          "lidar_radius", "evader_view", "tape_len", "pad0")] + \
               [(n, C.c_double) for n in ("def_tau", "def_dt", "def_collision_radius", "def_comm_range", "def_sen_range",
                                          "eva_vmax", "eva_tau", "eva_dt", "eva_collision_radius", "resolution")] + \
               [("action_u", (C.c_double * 2) * 9), ("beam_dir", (C.c_double * 2) * MAX_BEAMS)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "pe_env_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "libpe_env_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.environ.get("DMARL_PE_ORACLE_LIB") or LIB_PATH)  # override: the sanitizer build (tools/sanitize_host.py)
        vp = C.c_void_p
        L.peo_create.restype = vp
        L.peo_create.argtypes = [vp]
        L.peo_destroy.argtypes = [vp]
        L.peo_load.argtypes = [vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp]
        L.peo_get.argtypes = [vp] * 6
        L.peo_get_path.argtypes = [vp, vp]
        L.peo_get_rn.argtypes = [vp, vp, vp]
        L.peo_observe.argtypes = [vp] * 7
        L.peo_evader_step.argtypes = [vp, vp]
        L.peo_replan.argtypes = [vp, vp]
        L.peo_step.argtypes = [vp] * 5
        L.peo_step.restype = C.c_int
        L.peo_reward_norm.argtypes = [vp] * 4
        L.peo_lidar_cell.argtypes = [vp, vp, C.c_int, C.c_int, vp]
        L.peo_astar.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.peo_astar.restype = C.c_int
        L.peo_tick_batch.argtypes = [vp, vp, C.c_int, vp, vp]
        L.peo_tick_batch.restype = C.c_double
        L.peo_prims.argtypes = [C.c_int, vp, vp, vp]
        _lib = L
    return _lib


def make_config(W=40, H=40, P=8, O=176, max_steps=150, difficulty=10, extend_dis=1, num_beams=36, lidar_radius=8,
                evader_view=8, tape_len=16, def_vmax=2.0, def_tau=0.2, def_dt=0.1, def_collision_radius=0.5,
                def_comm_range=16.0, def_sen_range=8.0, eva_vmax=4.0, eva_tau=0.2, eva_dt=0.1, eva_collision_radius=0.5,
                resolution=1.0):
    c = PeoConfig()
    for k, v in dict(W=W, H=H, P=P, O=O, max_steps=max_steps, difficulty=difficulty, extend_dis=extend_dis,
                     num_beams=num_beams, lidar_radius=lidar_radius, evader_view=evader_view, tape_len=tape_len,
                     def_tau=def_tau, def_dt=def_dt, def_collision_radius=def_collision_radius,
                     def_comm_range=def_comm_range, def_sen_range=def_sen_range, eva_vmax=eva_vmax, eva_tau=eva_tau,
                     eva_dt=eva_dt, eva_collision_radius=eva_collision_radius, resolution=resolution).items():
        setattr(c, k, v)
    # agent.py:55-60 and pursuit_env.py:36-39, computed as the reference does (scalar cos/sin == glibc)
    for k in range(8):
        t = k * math.pi / 4
        c.action_u[k][0] = math.cos(t) * def_vmax
        c.action_u[k][1] = math.sin(t) * def_vmax
    c.action_u[8][0] = 0.0
    c.action_u[8][1] = 0.0
    for b in range(num_beams):
        a = b * 2 * math.pi / num_beams
        c.beam_dir[b][0] = math.cos(a)
        c.beam_dir[b][1] = math.sin(a)
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """One environment instance of the CPU restatement."""

    def __init__(self, cfg: PeoConfig):
        self.c = cfg
        self.L = lib()
        self.h = self.L.peo_create(C.byref(cfg))
        self.P, self.O = cfg.P, cfg.O

    def __del__(self):
        try:
            self.L.peo_destroy(self.h)
        except Exception:
            pass

    def load(self, grid, obs_xy, defenders, evader, target, tape=None):
        c = self.c
        grid = np.ascontiguousarray(grid, np.uint8)
        obs_xy = np.ascontiguousarray(obs_xy, np.int32).reshape(-1, 2)
        defenders = np.ascontiguousarray(defenders, np.float64)
        evader = np.ascontiguousarray(evader, np.float64)
        target = np.ascontiguousarray(target, np.int32)
        if tape is None:
            tape = np.zeros((max(c.tape_len, 1), 2), np.int32)
        tape = np.ascontiguousarray(tape, np.int32)
        assert grid.shape == (c.W, c.H) and defenders.shape == (c.P, 4) and tape.shape[0] >= c.tape_len
        self.n_obs = len(obs_xy)
        self.L.peo_load(C.byref(c), self.h, _p(grid), _p(obs_xy), len(obs_xy), _p(defenders), _p(evader), _p(target), _p(tape))

    def state(self):
        d = np.zeros((self.P, 4)); e = np.zeros(4); t = np.zeros(2, np.int32); s = np.zeros(5, np.int32)
        self.L.peo_get(C.byref(self.c), self.h, _p(d), _p(e), _p(t), _p(s))
        return dict(defenders=d, evader=e, target=t, t=int(s[0]), path_len=int(s[1]), tape_pos=int(s[2]),
                    collision=int(s[3]), astar_expansions=int(s[4]))

    def path(self):
        n = self.state()["path_len"]
        out = np.zeros((n, 2), np.int16)
        self.L.peo_get_path(self.h, _p(out))
        return out

    def reward_norm_state(self):
        o = np.zeros(1 + 2 * self.P)
        self.L.peo_get_rn(C.byref(self.c), self.h, _p(o))
        return int(o[0]), o[1:1 + self.P].copy(), o[1 + self.P:].copy()

    def observe(self):
        P, O = self.P, self.O
        ps = np.zeros((P, 4), np.float32); es = np.zeros((1, 4), np.float32); pa = np.zeros((P, P), np.float32)
        ea = np.zeros((P, 1), np.float32); oa = np.zeros((P, O), np.float32)
        self.L.peo_observe(C.byref(self.c), self.h, _p(ps), _p(es), _p(pa), _p(ea), _p(oa))
        return ps, es, pa, ea, oa

    def evader_step(self):
        self.L.peo_evader_step(C.byref(self.c), self.h)

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.int32)
        r = np.zeros(self.P); ok = np.zeros(self.P, np.uint8)
        done = self.L.peo_step(C.byref(self.c), self.h, _p(a), _p(r), _p(ok))
        return r, ok, bool(done)

    def reward_norm(self, r):
        r = np.ascontiguousarray(r, np.float64); out = np.zeros(self.P)
        self.L.peo_reward_norm(C.byref(self.c), self.h, _p(r), _p(out))
        return out

    def lidar_cell(self, cx, cy):
        f = np.zeros(max(self.n_obs, self.O), np.uint8)
        self.L.peo_lidar_cell(C.byref(self.c), self.h, cx, cy, _p(f))
        return f[:self.n_obs]


def demon(defenders, evader):
    """Pursuit_Env.demon restated line by line (reference environment/pursuit_evasion_game/pursuit_env.py:211-229), numpy scalar
    arithmetic as the reference evaluates it.  defenders (P, 4) f64, evader (4,) f64 -> list of P action indices.
    Pinned by the `action` arrays of the demon-policy traces (tests/golden/env_trace_*: steps with (t // 25) % 2 == 0)."""
    theta_list = [i * np.pi / 4 for i in range(0, 8)]
    actions_mat = [[np.cos(t), np.sin(t)] for t in theta_list]
    actions_mat.append([0., 0.])
    action_list = []
    e_x, e_y = float(evader[0]), float(evader[1])
    for d in np.asarray(defenders, np.float64):
        x, y = float(d[0]), float(d[1])
        radius = np.linalg.norm([x - e_x, y - e_y])
        if math.isclose(radius, 0.0, abs_tol=0.01):
            action = [0., 0.]
        else:
            phi = np.sign(e_y - y) * np.arccos((e_x - x) / (radius + 1e-3))
            action = [np.cos(phi), np.sin(phi)]
        middle_a = [np.linalg.norm((a[0] - action[0], a[1] - action[1])) for a in actions_mat]
        action_list.append(middle_a.index(min(middle_a)))
    return action_list


def astar(W, H, obs_grid, start, goal):
    """obs_grid: (W+1, H+1) uint8. Returns (path[n,2] goal->start, n_expanded)."""
    obs = np.ascontiguousarray(obs_grid, np.uint8)
    assert obs.shape == (W + 1, H + 1)
    out = np.zeros((MAX_PATH, 2), np.int16)
    nexp = C.c_int(0)
    n = lib().peo_astar(W, H, _p(obs), int(start[0]), int(start[1]), int(goal[0]), int(goal[1]), _p(out), C.byref(nexp))
    return out[:n].copy(), nexp.value


def tick_batch(cfg, envs, actions):
    """cpu_baseline leg: one observe -> evader -> step -> reward-norm tick over a list of OracleEnv."""
    n = len(envs)
    hs = (C.c_void_p * n)(*[e.h for e in envs])
    a = np.ascontiguousarray(actions, np.int32)
    scratch = np.zeros(4 * cfg.P + 4 + cfg.P * cfg.P + cfg.P + cfg.P * cfg.O, np.float32)
    return lib().peo_tick_batch(C.byref(cfg), hs, n, _p(a), _p(scratch))


def prims(a, b):
    a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
    out = np.zeros((3, len(a)))
    lib().peo_prims(len(a), _p(a), _p(b), _p(out))
    return out
