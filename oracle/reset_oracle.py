"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

Python restatement of the reference's episode reset (map, boundary obstacles, target, defenders, evader),
consuming the *same* global RNG streams (`random`, `numpy.random`) in the same order, so that under
identical seeds it reproduces the reference's initial condition bit for bit.

Follows (reference paths):
  environment/pursuit_evasion_game/pursuit_env.py:60-73   Pursuit_Env.reset
  environment/pursuit_evasion_game/base_env.py:37-162     init_map / init_target / init_defender / init_attacker
  environment/pursuit_evasion_game/Occupied_Grid_Map.py:46-62,119-166
  environment/pursuit_evasion_game/pursuit_env.py:18-27   get_boundary_map (skimage.find_boundaries 'inner')

Parity status: PINNED by tests/golden/env_trace_*.npz (initial conditions captured from the reference) --
except the inner-boundary extraction, whose third-party arithmetic (scikit-image 0.19.3, absent offline) is
restated from its published algorithm: "parity unpinned" at that call site.
"""
import random as _random

import numpy as np

MAX_DRAWS = 100000  # include/pe_env.h PE_RESET_MAX_DRAWS: the build's bound on the reference's unbounded rejection loops


class ResetFailed(RuntimeError):
    """a placement loop gave up after MAX_DRAWS candidates (the reference would loop forever)"""


def inflate(grid, cells, ext):
    """Occupied_Grid_Map.py:157-166 / :126-135 -- stamp [-ext, ext]^2 around every cell, clipped by in_bound."""
    W, H = grid.shape
    for (x, y) in cells:
        for xx in range(x - ext, x + ext + 1):
            for yy in range(y - ext, y + ext + 1):
                if 0 <= xx < W and 0 <= yy < H:
                    grid[xx, yy] = 1


def inner_boundary(grid):
    """find_boundaries(grid, mode='inner'), connectivity 1, reflect padding: an obstacle cell with at least one
    in-map 4-neighbour that is free (pursuit_env.py:20)."""
    g = grid != 0
    W, H = g.shape
    pad = np.pad(g, 1, mode="edge")  # a mirrored edge never differs from the cell itself
    free_nb = (~pad[:-2, 1:-1]) | (~pad[2:, 1:-1]) | (~pad[1:-1, :-2]) | (~pad[1:-1, 2:])
    return g & free_nb


def draw_target(inflated, rnd=_random):
    """base_env.py:52-70"""
    W, H = inflated.shape
    for _ in range(MAX_DRAWS):
        t = (rnd.randint(0, W - 1), rnd.randint(0, H - 1))
        if inflated[t] == 0:
            return t
    raise ResetFailed("init_target")


def reset_oracle(W, H, P, num_blocks, center, variance, comm_range=16, sen_range=8, min_dist=4, tape_len=0,
                 rnd=_random, nprnd=np.random, fixed_grid=None):
    """fixed_grid (W, H) u8: a pre-generated map (map-bank slot, the older reference driver's `map_info`,
    MAPPO_parallel_main.py:103-121) replaces init_map -- its draws are not taken; everything after it is unchanged."""
    grid = np.zeros((W, H), np.uint8) if fixed_grid is None else np.array(fixed_grid, np.uint8).reshape(W, H).copy()
    # init_map -> initailize_obstacle (Occupied_Grid_Map.py:56-62) -> add_blocker_type 'r' with data (6, 7)
    for _ in range(num_blocks if fixed_grid is None else 0):
        rnd.randrange(1)
        c = nprnd.normal(center, variance, 2)
        for x in range(-3, 3):
            for y in range(-3, 3):
                px, py = round(float(x + c[0])), round(float(y + c[1]))
                if 0 <= px < W and 0 <= py < H:
                    grid[px, py] = 1
    static_cells = [tuple(c) for c in np.argwhere(grid == 1).tolist()]
    inflated = grid.copy()
    inflate(inflated, static_cells, 2)
    bmap = inner_boundary(grid)
    obs_xy = np.argwhere(bmap).astype(np.int32)  # row-major order == obstacle index (pursuit_env.py:21)
    inflated_static = inflated.copy()            # self.inflated_map (pursuit_env.py:68)
    target = draw_target(inflated, rnd)
    # init_defender (base_env.py:72-120)
    scale = np.array([W - 1, H - 1])
    positions, cells = [], []
    draws = 0
    while len(positions) < P:
        pos = tuple(nprnd.rand(2) * scale)
        draws += 1
        if draws > MAX_DRAWS:
            raise ResetFailed("init_defender")
        ok = False
        if inflated[round(pos[0]), round(pos[1])] == 0:
            if not positions:
                ok = True
            else:
                dists = [np.linalg.norm((pos[0] - p[0], pos[1] - p[1])) for p in positions]
                collision = sum(d < min_dist for d in dists)
                connectivity = sum(d < comm_range for d in dists)
                ok = (collision == 0) and (connectivity > 0) and (connectivity <= 2)
        if ok:
            positions.append(pos)
            cell = (round(pos[0]), round(pos[1]))
            inflated[cell] = 1
            if cell not in cells:
                cells.append(cell)
            inflate(inflated, cells, 2)
    # init_attacker (base_env.py:122-162), is_percepted=True
    evader = None
    draws = 0
    while evader is None:
        pos = tuple(nprnd.rand(2) * scale)
        draws += 1
        if draws > MAX_DRAWS:
            raise ResetFailed("init_attacker")
        if inflated[round(pos[0]), round(pos[1])] == 0:
            for block in cells:
                if np.linalg.norm([block[0] - pos[0], block[1] - pos[1]]) < sen_range:
                    evader = pos
                    break
    tape = np.zeros((max(tape_len, 0), 2), np.int32)
    for k in range(tape_len):
        tape[k] = draw_target(inflated_static, rnd)
    defenders = np.zeros((P, 4), np.float64)
    defenders[:, 0] = [p[0] for p in positions]
    defenders[:, 1] = [p[1] for p in positions]
    eva = np.array([evader[0], evader[1], 0.0, 0.0], np.float64)
    return dict(grid=grid, inflated=inflated_static, obs_xy=obs_xy, target=np.array(target, np.int32),
                defenders=defenders, evader=eva, tape=tape)
