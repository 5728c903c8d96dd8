"""ctypes binding of libpe_env.so (include/pe_env.h) plus the device-resident batched environment.

PyTorch is used for device memory and streams only; the simulation runs in the HIP kernels of csrc/pe_env.hip.
There is no CPU path: importing works anywhere, but constructing a `BatchedEnv` without the built library or a
GPU raises.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import build as _build
from . import tables

MAX_BEAMS = 64
META_INTS = 8
META_T, META_PATH_LEN, META_TAPE_POS, META_COLLISION, META_PATH_CNT, META_ASTAR_EXP, META_STATUS = range(7)


class PeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("W", "H", "P", "O", "max_steps", "difficulty", "extend_dis", "num_beams",
                                          "lidar_radius", "evader_view", "tape_len", "max_path", "use_reward_norm",
                                          "pad0")] + \
               [(n, C.c_double) for n in ("def_tau", "def_dt", "def_collision_radius", "def_comm_range", "def_sen_range",
                                          "eva_vmax", "eva_tau", "eva_dt", "eva_collision_radius", "resolution")] + \
               [("action_u", (C.c_double * 2) * 9), ("beam_dir", (C.c_double * 2) * MAX_BEAMS)]


class PeState(C.Structure):
    _fields_ = [("N", C.c_int32), ("pad0", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("grid", "bidx", "n_obs", "def_", "eva", "target", "tape", "meta", "path", "rn", "wpw", "raser")]


class PeObsOut(C.Structure):
    _fields_ = [("p_state", C.c_void_p), ("p_state_stride", C.c_int64), ("e_state", C.c_void_p), ("e_state_stride", C.c_int64),
                ("p_adj", C.c_void_p), ("p_adj_stride", C.c_int64), ("e_adj", C.c_void_p), ("e_adj_stride", C.c_int64),
                ("o_adj", C.c_void_p), ("o_adj_stride", C.c_int64), ("o_adj_bits", C.c_void_p), ("o_adj_bits_stride", C.c_int64)]


class PeStepOut(C.Structure):
    _fields_ = [("reward", C.c_void_p), ("reward_stride", C.c_int64), ("reward_raw", C.c_void_p),
                ("reward_raw_stride", C.c_int64), ("done", C.c_void_p)]


class PeHostInit(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("grid", "obs_xy", "n_obs", "def_", "eva", "target", "tape")] + \
               [("reset_rn", C.c_int32), ("pad0", C.c_int32)]


class PeResetParams(C.Structure):
    _fields_ = [("num_blocks", C.c_int32), ("min_dist", C.c_int32), ("center", C.c_double * 2), ("variance", C.c_double),
                ("fixed_grid", C.c_void_p)]


class PeHostInitOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("grid", "obs_xy", "n_obs", "def_", "eva", "target", "tape")]


EXPORTS = ("pe_config_check", "pe_tick_lds_bytes", "pe_env_load", "pe_env_observe", "pe_evader_step", "pe_env_step",
           "pe_env_demon", "pe_env_tick", "pe_env_step_observe", "pe_astar_batch", "pe_error_string")

_lib = None


def lib_path():
    return _build.lib_path("libpe_env.so")


def load_library():
    """Loads libpe_env.so; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); the environment has no CPU fallback")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.pe_config_check.argtypes = [vp]
        L.pe_tick_lds_bytes.argtypes = [vp, C.c_int32]
        L.pe_tick_lds_bytes.restype = C.c_int64
        L.pe_env_load.argtypes = [vp, vp, vp, vp]
        L.pe_env_observe.argtypes = [vp, vp, vp, vp]
        L.pe_evader_step.argtypes = [vp, vp, C.c_int32, vp]
        L.pe_env_step.argtypes = [vp, vp, vp, vp, vp]
        L.pe_env_demon.argtypes = [vp, vp, vp, vp, vp]
        L.pe_env_tick.argtypes = [vp, vp, vp, vp, vp, C.c_int32, vp]
        L.pe_env_step_observe.argtypes = [vp, vp, vp, vp, vp, vp]
        L.pe_astar_batch.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int32, vp]
        L.pe_diag_norm2.argtypes = [C.c_int32, vp, vp, vp, C.c_double, C.c_double, vp]
        L.pe_resetter_create.argtypes = [vp, vp, C.c_int32, vp]
        L.pe_resetter_create.restype = vp
        L.pe_resetter_destroy.argtypes = [vp]
        L.pe_resetter_reset.argtypes = [vp, vp, vp, C.c_int32]
        L.pe_resetter_state_bytes.argtypes = [vp]
        L.pe_resetter_state_bytes.restype = C.c_int64
        L.pe_resetter_get_state.argtypes = [vp, vp]
        L.pe_resetter_set_state.argtypes = [vp, vp]
        L.pe_reset_state_bytes.argtypes = [vp, C.c_int32]
        L.pe_reset_state_bytes.restype = C.c_int64
        L.pe_env_reset_seed.argtypes = [vp, C.c_int32, vp, vp, vp]
        L.pe_env_reset.argtypes = [vp, vp, vp, vp, C.c_int32, vp, C.c_int32, vp]
        L.pe_error_string.argtypes = [C.c_int]
        L.pe_error_string.restype = C.c_char_p
        _lib = L
    return _lib


STATUS_NAMES = ((1, "target tape exhausted (raise runtime.tape_len)"), (2, "A* iteration cap reached"),
                (4, "stored path tail underflow (raise runtime.max_path)"),
                (8, "episode reset gave up: no valid placement after PE_RESET_MAX_DRAWS draws (map too crowded for this configuration)"))


def status_or(meta):
    """bitwise OR over environments of meta[:, PE_META_STATUS] as a 0-d device tensor (no host sync)"""
    st = meta[:, META_STATUS]
    shifts = torch.arange(len(STATUS_NAMES), device=st.device, dtype=st.dtype)
    flags = ((st[:, None] >> shifts) & 1).max(0).values
    return (flags << shifts).sum()


def status_text(bits):
    return "; ".join(n for b, n in STATUS_NAMES if int(bits) & b)


def raser_row_words(O):
    """include/pe_env.h PE_RASER_ROW_WORDS"""
    return (((O + 31) >> 5) + 3) & ~3


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load_library().pe_error_string(rc).decode()} (code {rc})")


def make_pe_config(cfg, tape_len=16, max_path=128):
    """Reference config sections (config.yaml:13-54) -> pe_config."""
    c = PeConfig()
    c.W, c.H = int(cfg.map.map_size[0]), int(cfg.map.map_size[1])
    c.P = int(cfg.env.num_defender)
    c.O = int(cfg.map.num_max_obstacle)
    c.max_steps = int(cfg.env.max_steps)
    c.difficulty = int(cfg.env.difficulty)
    c.extend_dis = int(cfg.attacker.extend_dis)
    c.num_beams = int(cfg.sensor.num_beams)
    c.lidar_radius = int(cfg.sensor.radius)
    c.evader_view = int(cfg.attacker.sen_range)
    c.tape_len = int(tape_len)
    c.max_path = int(max(max_path, c.difficulty + 2))
    c.use_reward_norm = 1 if cfg.algo.use_reward_norm else 0
    c.def_tau, c.def_dt = float(cfg.defender.tau), float(cfg.defender.step_size)
    c.def_collision_radius = float(cfg.defender.collision_radius)
    c.def_comm_range, c.def_sen_range = float(cfg.defender.comm_range), float(cfg.defender.sen_range)
    c.eva_vmax, c.eva_tau, c.eva_dt = float(cfg.attacker.vmax), float(cfg.attacker.tau), float(cfg.attacker.step_size)
    c.eva_collision_radius = float(cfg.attacker.collision_radius)
    c.resolution = float(cfg.map.resolution)
    for k, (ux, uy) in enumerate(tables.action_table(float(cfg.defender.vmax))):
        c.action_u[k][0], c.action_u[k][1] = ux, uy
    for b, (bx, by) in enumerate(tables.beam_table(c.num_beams)):
        c.beam_dir[b][0], c.beam_dir[b][1] = bx, by
    return c


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _np(a):
    return a.ctypes.data_as(C.c_void_p)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class BatchedEnv:
    """N independent pursuit-evasion environments resident in HBM, stepped by the HIP kernels.

    The tensor attributes are the `pe_state` records of include/pe_env.h; `obs` holds the fp32 observation tensors in
    the reference's layouts (N leading)."""

    def __init__(self, pe_cfg: PeConfig, num_envs: int, device="cuda"):
        self.L = load_library()
        _check(self.L.pe_config_check(C.byref(pe_cfg)), "pe_config_check")
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedEnv needs a GPU (MI355X); there is no CPU path")
        self.c = pe_cfg
        self.N = int(num_envs)
        self.device = torch.device(device)
        c, N, dev = pe_cfg, self.N, self.device
        WH, P = c.W * c.H, c.P
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self._t = dict(grid=z((N, WH), torch.uint8), bidx=z((N, WH), torch.int16), n_obs=z((N,), torch.int32),
                       defs=z((N, 4, P), torch.float64), eva=z((N, 4), torch.float64), target=z((N, 2), torch.int32),
                       tape=z((N, c.tape_len, 2), torch.int32), meta=z((N, META_INTS), torch.int32),
                       path=z((N, c.max_path, 2), torch.int16), rn=z((N, 1 + 2 * P), torch.float64), wpw=z((N, 16), torch.int32),
                       raser=z((N, WH, raser_row_words(c.O)), torch.int32))
        self.o_state = z((N, c.O, 4), torch.float32)  # boundary obstacles as [x, y, 0, 0] (pursuit_env.py:22-26), padded
        self.st = PeState()
        self.st.N = N
        for name, t in self._t.items():
            setattr(self.st, "def_" if name == "defs" else name, t.data_ptr())
        self.t_host = None  # lockstep time_step known to the host (None: unknown -> always allow replans)
        # replan ticks run the evader's rescan + A* on a second stream so its stragglers overlap the next policy forward
        self.overlap_replan = True
        self._side = torch.cuda.Stream(device=self.device)
        self._ev_obs = torch.cuda.Event()
        self._ev_evader = torch.cuda.Event()
        self._pending = False

    # -- hand-over of Pursuit_Env.reset() results --------------------------------------------------------
    def load(self, init, reset_reward_norm=False):
        self._join()
        """init: dict of host numpy arrays grid [N,W,H] u8, obs_xy [N,O,2] i32 (padded), n_obs [N], defenders [N,P,4] f64,
        evader [N,4] f64, target [N,2] i32, tape [N,tape_len,2] i32."""
        c, N = self.c, self.N
        g = np.ascontiguousarray(init["grid"], np.uint8).reshape(N, c.W * c.H)
        ob = np.ascontiguousarray(init["obs_xy"], np.int32).reshape(N, c.O, 2)
        no = np.ascontiguousarray(init["n_obs"], np.int32).reshape(N)
        if int(no.max()) > c.O:
            raise ValueError(f"an environment has {int(no.max())} boundary obstacles > num_max_obstacle={c.O}")
        d = np.ascontiguousarray(init["defenders"], np.float64).reshape(N, c.P, 4)
        e = np.ascontiguousarray(init["evader"], np.float64).reshape(N, 4)
        tg = np.ascontiguousarray(init["target"], np.int32).reshape(N, 2)
        tp = np.ascontiguousarray(init["tape"], np.int32).reshape(N, c.tape_len, 2)
        h = PeHostInit()
        h.grid, h.obs_xy, h.n_obs, h.def_, h.eva, h.target, h.tape = (_np(g), _np(ob), _np(no), _np(d), _np(e), _np(tg), _np(tp))
        h.reset_rn = 1 if reset_reward_norm else 0
        self._keep = (g, ob, no, d, e, tg, tp)  # the async copies read these host arrays
        with torch.cuda.device(self.device):
            _check(self.L.pe_env_load(C.byref(self.c), C.byref(self.st), C.byref(h), _stream()), "pe_env_load")
            torch.cuda.current_stream().synchronize()
        self._keep = None
        os_ = np.zeros((N, c.O, 4), np.float32)
        os_[:, :, :2] = ob.astype(np.float32)
        valid = np.arange(c.O)[None, :] < no[:, None]
        os_[~valid] = 0.0
        self.o_state.copy_(torch.from_numpy(os_))
        self.t_host = 0

    # -- observation / step entry points -----------------------------------------------------------------
    def new_obs(self, packed=False):
        """fp32 observation tensors in the reference's layouts; packed=True: the LiDAR rows as `o_adj_bits` (N, P, RW) int32
        (bit k of row i = o_adj[i][k]) instead of the (N, P, O) float `o_adj` -- the form the product's rollout consumes."""
        c, N, dev = self.c, self.N, self.device
        f = lambda *s: torch.empty((N, *s), dtype=torch.float32, device=dev)
        obs = dict(p_state=f(c.P, 4), e_state=f(1, 4), p_adj=f(c.P, c.P), e_adj=f(c.P, 1))
        if packed:
            obs["o_adj_bits"] = torch.zeros((N, c.P, raser_row_words(c.O)), dtype=torch.int32, device=dev)
        else:
            obs["o_adj"] = f(c.P, c.O)
        return obs

    @staticmethod
    def _obs_struct(obs):
        o = PeObsOut()
        for k in ("p_state", "e_state", "p_adj", "e_adj", "o_adj", "o_adj_bits"):
            t = obs.get(k) if obs else None
            if t is not None:
                assert t.dtype == (torch.int32 if k == "o_adj_bits" else torch.float32) and t[0].is_contiguous(), k
                setattr(o, k, t.data_ptr())
                setattr(o, k + "_stride", t.stride(0))
        return o

    @staticmethod
    def _step_struct(reward, reward_raw, done):
        s = PeStepOut()
        if reward is not None:
            assert reward.dtype == torch.float32 and reward[0].is_contiguous()
            s.reward, s.reward_stride = reward.data_ptr(), reward.stride(0)
        if reward_raw is not None:
            assert reward_raw.dtype == torch.float32 and reward_raw[0].is_contiguous()
            s.reward_raw, s.reward_raw_stride = reward_raw.data_ptr(), reward_raw.stride(0)
        if done is not None:
            assert done.dtype == torch.uint8 and done.is_contiguous()
            s.done = done.data_ptr()
        return s

    def _may_replan(self):
        return 1 if (self.t_host is None or self.t_host % self.c.difficulty == 0) else 0

    def __getattr__(self, name):
        # state tensors (grid, bidx, n_obs, defs, eva, target, tape, meta, path, rn): reading them orders the caller's stream
        # after an evader step that may still run on the side stream
        t = self.__dict__.get("_t")
        if t is not None and name in t:
            self._join()
            return t[name]
        raise AttributeError(name)

    def _join(self):
        """every entry point that reads or writes environment state first waits for an evader step still in flight"""
        if self._pending:
            torch.cuda.current_stream().wait_event(self._ev_evader)
            self._pending = False
        ev = self.__dict__.get("_ev_reset")
        if ev is not None:      # a device reset issued ahead on the side stream (Pursuit_Env.prefetch_reset)
            torch.cuda.current_stream().wait_event(ev)
            self.__dict__["_ev_reset"] = None

    def observe(self, obs=None):
        self._join()
        obs = obs if obs is not None else self.new_obs()
        o = self._obs_struct(obs)
        _check(self.L.pe_env_observe(C.byref(self.c), C.byref(self.st), C.byref(o), _stream()), "pe_env_observe")
        return obs

    def evader_step(self):
        self._join()
        _check(self.L.pe_evader_step(C.byref(self.c), C.byref(self.st), self._may_replan(), _stream()), "pe_evader_step")

    def step(self, actions, reward=None, reward_raw=None, done=None):
        self._join()
        a = self._actions(actions)
        if reward is None:
            reward = torch.empty((self.N, self.c.P), dtype=torch.float32, device=self.device)
        s = self._step_struct(reward, reward_raw, done)
        _check(self.L.pe_env_step(C.byref(self.c), C.byref(self.st), _ptr(a), C.byref(s), _stream()), "pe_env_step")
        if self.t_host is not None:
            self.t_host += 1
        return reward

    def tick(self, actions, obs, reward, reward_raw=None, done=None):
        """Fused step(actions) -> observe -> attacker_step (one launch)."""
        a = self._actions(actions)
        s = self._step_struct(reward, reward_raw, done)
        o = self._obs_struct(obs)
        self._join()
        if self.t_host is not None:
            self.t_host += 1
        replan = self._may_replan()
        if replan and self.overlap_replan:
            _check(self.L.pe_env_step_observe(C.byref(self.c), C.byref(self.st), _ptr(a), C.byref(s), C.byref(o), _stream()),
                   "pe_env_step_observe")
            self._ev_obs.record(torch.cuda.current_stream())
            self._side.wait_event(self._ev_obs)
            _check(self.L.pe_evader_step(C.byref(self.c), C.byref(self.st), 1, C.c_void_p(self._side.cuda_stream)), "pe_evader_step")
            self._ev_evader.record(self._side)
            self._pending = True
            return
        _check(self.L.pe_env_tick(C.byref(self.c), C.byref(self.st), _ptr(a), C.byref(s), C.byref(o), replan, _stream()), "pe_env_tick")

    def demon(self, out=None):
        """Pursuit_Env.demon (pursuit_env.py:211-229) for every environment -> (N, P) int32 actions (csrc/pe_env.hip k_demon)."""
        self._join()
        if out is None:
            out = torch.empty((self.N, self.c.P), dtype=torch.int32, device=self.device)
        assert out.dtype == torch.int32 and out.is_contiguous() and out.shape == (self.N, self.c.P)
        dirs = (C.c_double * 18)(*[v for cs in tables.action_table(1.0) for v in cs])
        _check(self.L.pe_env_demon(C.byref(self.c), C.byref(self.st), dirs, _ptr(out), _stream()), "pe_env_demon")
        return out

    def _actions(self, actions):
        if actions.dtype != torch.int32:
            actions = actions.to(torch.int32)
        actions = actions.contiguous()
        assert actions.shape == (self.N, self.c.P) and actions.device.type == "cuda"
        return actions

    # -- host read-back (tests, logging) ---------------------------------------------------------------------
    def defenders_aos(self):
        self._join()
        return self.defs.permute(0, 2, 1).contiguous()  # [N][P][4] like get_state('defender')

    def status(self):
        return self.meta[:, META_STATUS]


def astar_batch(W, H, obs, sg, max_path=256):
    """Diagnostic/test entry: one weighted-A* problem per workgroup. obs [n,(W+1),(H+1)] u8, sg [n,4] int."""
    L = load_library()
    n = obs.shape[0]
    obs_d = torch.as_tensor(np.ascontiguousarray(obs, np.uint8)).cuda()
    sg_d = torch.as_tensor(np.ascontiguousarray(sg, np.int32)).cuda()
    path = torch.zeros((n, max_path, 2), dtype=torch.int16, device="cuda")
    lens = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    _check(L.pe_astar_batch(W, H, n, _ptr(obs_d), _ptr(sg_d), _ptr(path), _ptr(lens), max_path, _stream()), "pe_astar_batch")
    torch.cuda.synchronize()
    return path.cpu().numpy(), lens.cpu().numpy()


def _reset_params(cfg):
    prm = PeResetParams()
    prm.num_blocks = int(cfg.map.num_obstacle_block)
    prm.min_dist = 4  # pursuit_env.py:71
    prm.center[0], prm.center[1] = float(cfg.map.center[0]), float(cfg.map.center[1])
    prm.variance = float(cfg.map.variance)
    return prm


def _seed_array(seeds):
    s = np.ascontiguousarray(seeds, np.uint64)
    if (s >> np.uint64(32)).any():
        raise ValueError("seeds must fit 32 bits (numpy.random.seed range)")
    return s


class DeviceResetter:
    """Pursuit_Env.reset() on the GPU (csrc/pe_env.hip k_reset): the generator streams of every environment live in device
    memory; same streams, same draws, same results as HostResetter, with no host arrays and no upload."""

    def __init__(self, sim, cfg, seeds, map_bank=0, bank_seed=0):
        """map_bank = B > 0: B maps are generated once (the product's host map generator with seeds bank_seed .. bank_seed + B - 1);
        every reset picks one slot for ALL environments (random.Random(bank_seed).randrange(B), like the older reference driver's
        one `map_info` per node and iteration, MAPPO_parallel_main.py:103-124) and only draws targets / defenders / evader."""
        self.L = load_library()
        self.sim = sim
        self.c = sim.c
        self.N = len(seeds)
        assert self.N == sim.N
        self.prm = _reset_params(cfg)
        self.bank = None
        if map_bank:
            import random as _pyrandom
            gen = HostResetter(sim.c, cfg, [int(bank_seed) + k for k in range(int(map_bank))])
            self.bank = torch.from_numpy(gen.reset()["grid"].reshape(int(map_bank), -1)).to(sim.device).contiguous()   # (B, W*H) u8
            self.bank_rng = _pyrandom.Random(int(bank_seed))
            self.bank_slot = -1
        s = _seed_array(seeds)
        n = self.L.pe_reset_state_bytes(C.byref(self.c), self.N)
        self.state = torch.empty(n, dtype=torch.uint8, device=sim.device)
        with torch.cuda.device(sim.device):
            _check(self.L.pe_env_reset_seed(C.byref(self.c), self.N, _np(s), _ptr(self.state), _stream()), "pe_env_reset_seed")
        self.first = True

    def get_state(self, snapshot=None):
        """the resume-bundle form (host arrays); snapshot: a snapshot_device() result to convert instead of the live state"""
        src = snapshot if snapshot is not None else dict(blob=self.state, first=self.first, bank_rng=self.bank_rng.getstate() if self.bank is not None else None)
        d = dict(blob=src["blob"].cpu().numpy(), first=src["first"], device=True)
        if self.bank is not None:
            d["bank_rng"] = src["bank_rng"]
        return d

    def snapshot_device(self):
        """the generator streams as they stand, kept on the device (no host copy, no synchronisation)"""
        return dict(blob=self.state.clone(), first=self.first, bank_rng=self.bank_rng.getstate() if self.bank is not None else None)

    def set_state(self, state):
        buf = np.ascontiguousarray(state["blob"], np.uint8)
        if not state.get("device") or buf.size != self.state.numel():
            raise ValueError("resetter state does not match this configuration / number of environments / reset mode")
        self.state.copy_(torch.from_numpy(buf))
        self.first = bool(state["first"])
        if self.bank is not None and "bank_rng" in state:
            self.bank_rng.setstate(state["bank_rng"])

    def reset(self, reset_reward_norm=False):
        """Next episode of every environment, in place in the simulator state (reads the finished episode's tape position
        from the device)."""
        self.launch(reset_reward_norm)
        self.check()

    def launch(self, reset_reward_norm=False):
        """the reset kernels on the current stream, no read-back (Pursuit_Env.prefetch_reset issues them on a side stream under the
        PPO update; check() follows when the episode starts)"""
        sim = self.sim
        sim._join()
        if self.bank is not None:
            self.bank_slot = self.bank_rng.randrange(self.bank.shape[0])
            self.prm.fixed_grid = self.bank[self.bank_slot].data_ptr()
        with torch.cuda.device(sim.device):
            _check(self.L.pe_env_reset(C.byref(self.c), C.byref(sim.st), C.byref(self.prm), _ptr(self.state), 1 if self.first else 0,
                                       _ptr(sim.o_state), 1 if reset_reward_norm else 0, _stream()), "pe_env_reset")
        self.first = False
        sim.t_host = 0

    def check(self):
        sim = self.sim
        # one blocking read-back per episode: the largest obstacle count and the sticky kernel status bits (pe_env.h: the
        # finished episode's tape-exhausted / A*-cap / path-underflow bits are carried into the new meta record by the reset)
        worst, bits = torch.stack((sim.n_obs.max(), status_or(sim.meta))).tolist()
        if bits:
            raise RuntimeError("environment kernel status: " + status_text(bits))
        if worst > self.c.O:
            raise ValueError(f"an environment has {worst} boundary obstacles > num_max_obstacle={self.c.O}")


class HostResetter:
    """Host side of Pursuit_Env.reset() for N environments (csrc/pe_reset.cpp): per-environment re-implementations of
    the reference's `random` / `numpy.random` streams, seeded like random.seed(s); np.random.seed(s)."""

    def __init__(self, pe_cfg: PeConfig, cfg, seeds, n_threads=None):
        self.L = load_library()
        self.c = pe_cfg
        self.N = len(seeds)
        prm = _reset_params(cfg)
        s = _seed_array(seeds)
        self.h = self.L.pe_resetter_create(C.byref(pe_cfg), C.byref(prm), self.N, _np(s))
        if not self.h:
            raise RuntimeError("pe_resetter_create failed (bad configuration)")
        world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
        self.n_threads = int(n_threads or max(1, min(16, (os.cpu_count() or 1) // world)))
        self.first = True

    def __del__(self):
        try:
            if self.h:
                self.L.pe_resetter_destroy(self.h)
        except Exception:
            pass

    def get_state(self):
        buf = np.empty(self.L.pe_resetter_state_bytes(self.h), np.uint8)
        _check(self.L.pe_resetter_get_state(self.h, _np(buf)), "pe_resetter_get_state")
        return dict(blob=buf, first=self.first)

    def set_state(self, state):
        buf = np.ascontiguousarray(state["blob"], np.uint8)
        if buf.size != self.L.pe_resetter_state_bytes(self.h):
            raise ValueError("resetter state does not match this configuration / number of environments")
        _check(self.L.pe_resetter_set_state(self.h, _np(buf)), "pe_resetter_set_state")
        self.first = bool(state["first"])

    def reset(self, consumed_targets=None):
        c, N = self.c, self.N
        out = dict(grid=np.empty((N, c.W, c.H), np.uint8), obs_xy=np.empty((N, c.O, 2), np.int32), n_obs=np.empty(N, np.int32),
                   defenders=np.empty((N, c.P, 4), np.float64), evader=np.empty((N, 4), np.float64),
                   target=np.empty((N, 2), np.int32), tape=np.empty((N, c.tape_len, 2), np.int32))
        o = PeHostInitOut()
        o.grid, o.obs_xy, o.n_obs, o.def_, o.eva, o.target, o.tape = (_np(out[k]) for k in
                                                                        ("grid", "obs_xy", "n_obs", "defenders", "evader", "target", "tape"))
        ct = None
        if not self.first:
            if consumed_targets is None:
                raise ValueError("consumed_targets (tape positions of the finished episode) is required after the first reset")
            ct = np.ascontiguousarray(consumed_targets, np.int32)
        _check(self.L.pe_resetter_reset(self.h, _np(ct) if ct is not None else None, C.byref(o), self.n_threads), "pe_resetter_reset")
        self.first = False
        return out
