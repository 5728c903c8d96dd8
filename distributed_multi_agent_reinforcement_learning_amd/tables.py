"""Exact f64 direction tables the reference environment computes with numpy at run time.

* ``ACTION_DIRS``: (cos, sin)(k*pi/4), k<8 -- reference environment/pursuit_evasion_game/agent.py:55-60
  (the desired velocity is this times ``vmax``; action 8 is (0, 0)).
* ``BEAM36_DIRS``: (cos, sin)(b*2*pi/36) -- reference pursuit_env.py:36-39 (LiDAR beams).

Captured once in the build container (numpy 2.2.6 scalar cos/sin == glibc) and committed as hex floats so
that the kernels see bit-identical constants on every host (numpy's SIMD trig may differ by an ulp across
builds; e.g. cos(27*2*pi/36) = -1.8369701987210297e-16, not 0, changes which column beam 27 samples).
For any other beam count the table is computed on the host with ``math.cos/sin``.
"""
import math

ACTION_DIRS_HEX = [
    ("0x1.0000000000000p+0", "0x0.0p+0"),
    ("0x1.6a09e667f3bcdp-1", "0x1.6a09e667f3bccp-1"),
    ("0x1.1a62633145c07p-54", "0x1.0000000000000p+0"),
    ("-0x1.6a09e667f3bccp-1", "0x1.6a09e667f3bcdp-1"),
    ("-0x1.0000000000000p+0", "0x1.1a62633145c07p-53"),
    ("-0x1.6a09e667f3bcep-1", "-0x1.6a09e667f3bccp-1"),
    ("-0x1.a79394c9e8a0ap-53", "-0x1.0000000000000p+0"),
    ("0x1.6a09e667f3bcbp-1", "-0x1.6a09e667f3bcep-1"),
]

BEAM36_DIRS_HEX = [
    ("0x1.0000000000000p+0", "0x0.0p+0"),
    ("0x1.f838b8c811c17p-1", "0x1.63a1a7e0b7389p-3"),
    ("0x1.e11f642522d1cp-1", "0x1.5e3a8748a0bf5p-2"),
    ("0x1.bb67ae8584cabp-1", "0x1.fffffffffffffp-2"),
    ("0x1.8836fa2cf5039p-1", "0x1.491b7523c161cp-1"),
    ("0x1.491b7523c161dp-1", "0x1.8836fa2cf5039p-1"),
    ("0x1.0000000000001p-1", "0x1.bb67ae8584caap-1"),
    ("0x1.5e3a8748a0bf7p-2", "0x1.e11f642522d1bp-1"),
    ("0x1.63a1a7e0b738cp-3", "0x1.f838b8c811c17p-1"),
    ("0x1.1a62633145c07p-54", "0x1.0000000000000p+0"),
    ("-0x1.63a1a7e0b7388p-3", "0x1.f838b8c811c17p-1"),
    ("-0x1.5e3a8748a0bf1p-2", "0x1.e11f642522d1cp-1"),
    ("-0x1.ffffffffffffcp-2", "0x1.bb67ae8584cabp-1"),
    ("-0x1.491b7523c161dp-1", "0x1.8836fa2cf5039p-1"),
    ("-0x1.8836fa2cf5038p-1", "0x1.491b7523c161ep-1"),
    ("-0x1.bb67ae8584ca9p-1", "0x1.0000000000003p-1"),
    ("-0x1.e11f642522d1bp-1", "0x1.5e3a8748a0bf8p-2"),
    ("-0x1.f838b8c811c17p-1", "0x1.63a1a7e0b7387p-3"),
    ("-0x1.0000000000000p+0", "0x1.1a62633145c07p-53"),
    ("-0x1.f838b8c811c18p-1", "-0x1.63a1a7e0b737ep-3"),
    ("-0x1.e11f642522d1cp-1", "-0x1.5e3a8748a0bf4p-2"),
    ("-0x1.bb67ae8584caap-1", "-0x1.0000000000001p-1"),
    ("-0x1.8836fa2cf503cp-1", "-0x1.491b7523c1619p-1"),
    ("-0x1.491b7523c161ep-1", "-0x1.8836fa2cf5038p-1"),
    ("-0x1.0000000000004p-1", "-0x1.bb67ae8584ca8p-1"),
    ("-0x1.5e3a8748a0bf2p-2", "-0x1.e11f642522d1cp-1"),
    ("-0x1.63a1a7e0b7389p-3", "-0x1.f838b8c811c17p-1"),
    ("-0x1.a79394c9e8a0ap-53", "-0x1.0000000000000p+0"),
    ("0x1.63a1a7e0b737cp-3", "-0x1.f838b8c811c18p-1"),
    ("0x1.5e3a8748a0bfap-2", "-0x1.e11f642522d1bp-1"),
    ("0x1.ffffffffffff4p-2", "-0x1.bb67ae8584caep-1"),
    ("0x1.491b7523c161cp-1", "-0x1.8836fa2cf503ap-1"),
    ("0x1.8836fa2cf5037p-1", "-0x1.491b7523c161fp-1"),
    ("0x1.bb67ae8584cacp-1", "-0x1.ffffffffffffap-2"),
    ("0x1.e11f642522d1cp-1", "-0x1.5e3a8748a0bf3p-2"),
    ("0x1.f838b8c811c17p-1", "-0x1.63a1a7e0b738bp-3"),
]

ACTION_DIRS = [(float.fromhex(c), float.fromhex(s)) for c, s in ACTION_DIRS_HEX]
BEAM36_DIRS = [(float.fromhex(c), float.fromhex(s)) for c, s in BEAM36_DIRS_HEX]


def action_table(vmax):
    """agent.py:57-60: [[cos(t) * vmax, sin(t) * vmax] for t in theta_list] + [[0., 0.]]"""
    tab = [(c * vmax, s * vmax) for c, s in ACTION_DIRS]
    tab.append((0.0, 0.0))
    return tab


def beam_table(num_beams):
    """pursuit_env.py:36-39: beam_angle = beam * 2 * pi / num_beams"""
    if num_beams == 36:
        return list(BEAM36_DIRS)
    return [(math.cos(b * 2 * math.pi / num_beams), math.sin(b * 2 * math.pi / num_beams)) for b in range(num_beams)]
