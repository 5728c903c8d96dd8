"""`Pursuit_Env`: the reference's environment API on top of the batched HIP simulator.

Mirrors environment/pursuit_evasion_game/pursuit_env.py:56-229 (+ base_env.py) of the reference for `num_envs`
independent environments at once: every method keeps the reference's name and meaning, tensors carry a leading
environment dimension N (one reference `Worker` each).  The simulation runs in csrc/pe_env.hip through the C ABI of
include/pe_env.h; the episode reset runs in csrc/pe_reset.cpp with per-environment replicas of the reference's RNG
streams (environment n of rank r is seeded `seed + max(1000, num_envs) * r + n`).
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import pe_env


class Pursuit_Env:
    def __init__(self, cfg, num_envs=None, rank=0, device=None, seeds=None):
        self.cfg = cfg
        rt = cfg.get("runtime", {}) if hasattr(cfg, "get") else {}
        self.num_envs = int(num_envs if num_envs is not None else rt.get("num_envs", 1))
        self.device = torch.device(device if device is not None else "cuda")
        self.map_config, self.env_config = cfg.map, cfg.env
        self.defender_config, self.attacker_config, self.sensor_config = cfg.defender, cfg.attacker, cfg.sensor
        self.max_steps = cfg.env.max_steps
        self.step_size = cfg.env.step_size
        self.num_target, self.num_defender, self.num_attacker = cfg.env.num_target, cfg.env.num_defender, cfg.env.num_attacker
        self.time_step = 0
        self.n_episode = 0
        self.pe_cfg = pe_env.make_pe_config(cfg, tape_len=int(rt.get("tape_len", 16)), max_path=int(rt.get("max_path", 128)))
        self.sim = pe_env.BatchedEnv(self.pe_cfg, self.num_envs, self.device)
        if seeds is None:
            base = int(rt.get("seed", 0)) + max(1000, self.num_envs) * int(rank)  # disjoint streams per rank (1000 * rank below 1000 envs)
            seeds = [base + n for n in range(self.num_envs)]
        self.seeds = list(seeds)
        # runtime.device_reset: the episode reset runs on the GPU (same streams and draws as the host resetter, no upload)
        self.device_reset = bool(rt.get("device_reset", False))
        self.map_bank = int(rt.get("map_bank", 0))   # > 0: resets draw their map from a pre-generated bank (device reset only)
        if self.map_bank and not self.device_reset:
            raise ValueError("runtime.map_bank needs runtime.device_reset")
        self.resetter = (pe_env.DeviceResetter(self.sim, cfg, self.seeds, self.map_bank, int(rt.get("map_bank_seed", 10 ** 6)) + 7919 * rank)
                         if self.device_reset else pe_env.HostResetter(self.pe_cfg, cfg, self.seeds))
        self.boundary_map = SimpleNamespace(obstacle_agent=self.sim.o_state)  # (N, O, 4) [x, y, 0, 0], zero padded
        self._obs = None
        self._reward = torch.zeros((self.num_envs, self.num_defender), dtype=torch.float32, device=self.device)
        self._raw = torch.zeros_like(self._reward)
        self._done = torch.zeros((self.num_envs,), dtype=torch.uint8, device=self.device)

    # ---- reference API -------------------------------------------------------------------------------------
    def reset(self, init=None):
        """pursuit_env.py:60-73.  `init` (host arrays, see BatchedEnv.load) injects recorded initial conditions."""
        self.time_step = 0
        self.n_episode += 1
        if init is None and self.device_reset:
            ev = getattr(self, "_dev_prefetch", None)
            if ev is not None:          # the reset kernels already ran on the side stream (prefetch_reset): order this stream behind them
                torch.cuda.current_stream().wait_event(ev)
                self._dev_prefetch = self._pre_reset = None
                self.resetter.check()
            else:
                self.resetter.reset()
            self.last_init = None
            return None
        if init is not None and getattr(self, "_dev_prefetch", None) is not None:
            raise RuntimeError("an injected initial condition after prefetch_reset(): the device reset of the next episode has already run")
        if init is None:
            init = self._take_prefetched()
            if init is None:
                consumed = None if self.resetter.first else self.sim.meta[:, pe_env.META_TAPE_POS].cpu().numpy()
                init = self.resetter.reset(consumed)
        self.sim.load(init)
        self.last_init = init
        return None

    # ---- host reset of the NEXT episode overlapped with device work (e.g. the PPO update) ------------------------
    def prefetch_reset(self):
        """Starts the host-side reset of the next episode in a background thread (the C++ resetter releases the GIL).  Call
        it once the running episode is over: the number of tape targets it consumed is read here."""
        import threading
        if self.device_reset:
            # device reset: the next episode's reset kernels (k_reset, k_build_bidx, k_build_raser: ~4 ms at 4096 environments, VALU
            # work) run on the simulator's side stream under whatever the caller launches next -- the HBM-bound PPO update, which
            # reads the replay buffer only, never the simulator state.  reset() then waits for the event instead of launching.
            if getattr(self, "_dev_prefetch", None) is None:
                # what a resume bundle written from now on must hold (trainer.Trainer.save_resume): the state the reset starts from
                self._pre_reset = dict(resetter=self.resetter.snapshot_device(), tape_pos=self.sim.meta[:, pe_env.META_TAPE_POS].clone())
                side = self.sim._side
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self.resetter.launch()
                    ev = torch.cuda.Event()
                    ev.record(side)
                self._dev_prefetch = ev
                self.sim._ev_reset = ev      # any reader of the simulator state on another stream waits for it (BatchedEnv._join)
            return
        if getattr(self, "_prefetch", None) is not None:
            return
        self.check_status()
        consumed = None if self.resetter.first else self.sim.meta[:, pe_env.META_TAPE_POS].cpu().numpy()
        box = {}

        def work():
            box["init"] = self.resetter.reset(consumed)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        self._prefetch = (th, box)

    def check_status(self):
        """Raises when a kernel flagged a condition that breaks parity with the reference (include/pe_env.h PE_STATUS_*)."""
        bits = int(pe_env.status_or(self.sim.meta).item()) if self.num_envs else 0
        if bits:
            raise RuntimeError("environment kernel status: " + pe_env.status_text(bits))

    def _take_prefetched(self):
        pf = getattr(self, "_prefetch", None)
        if pf is None:
            return None
        th, box = pf
        th.join()
        self._prefetch = None
        return box.get("init")

    @property
    def collision(self):
        return self.sim.meta[:, pe_env.META_COLLISION].bool()

    @property
    def target(self):
        return self.sim.target

    @property
    def n_obs(self):
        return self.sim.n_obs

    def get_state(self, agent_type):
        """base_env.py:198-209: (N, P, 4) defenders or (N, 1, 4) attacker, f64 [x, y, vx, vy]."""
        if agent_type == "defender":
            return self.sim.defenders_aos()
        if agent_type == "attacker":
            return self.sim.eva.unsqueeze(1)
        raise KeyError(agent_type)

    def observe(self, obs=None):
        """get_state + communicate + sensor in one launch; fp32 tensors in the reference's layouts."""
        self._obs = self.sim.observe(obs)
        return self._obs

    def communicate(self):
        """pursuit_env.py:182-195 -> (N, P, P) fp32"""
        return self.observe()["p_adj"]

    def sensor(self):
        """pursuit_env.py:197-209 -> (o_adj (N, P, O), e_adj (N, P, 1))"""
        o = self.observe()
        return o["o_adj"], o["e_adj"]

    def attacker_step(self):
        """pursuit_env.py:75-102 (replan every `difficulty` steps, follow the path, re-draw the target on arrival)."""
        self.sim.evader_step()
        return None

    def step(self, action):
        """pursuit_env.py:104-123 -> (rewards (N, P) raw fp32, done, info)"""
        a = torch.as_tensor(action, device=self.device)
        self.sim.step(a.reshape(self.num_envs, self.num_defender), self._reward, self._raw, self._done)
        self.time_step += 1
        return self._raw, self.time_step >= self.max_steps, None

    def demon(self):
        """pursuit_env.py:211-229: scripted pursuit, the discrete action closest to the bearing of the evader -> (N, P) int32
        (one launch, csrc/pe_env.hip k_demon; pinned by the demon actions recorded in the reference traces)."""
        return self.sim.demon()

    def get_done(self):
        return self.time_step >= self.max_steps

    # ---- fused rollout entry (one launch per environment tick) ---------------------------------------------
    def tick(self, action, obs, reward, reward_raw=None, done=None):
        """step(action) -> observe -> attacker_step fused (the tail of one run_episode iteration and the head of the next)."""
        self.sim.tick(action, obs, reward, reward_raw, done)
        self.time_step += 1
