"""ASCII layout registry.

Mirrors ``cogrid.core.layouts.register_layout`` as used by the reference
(overcooked_config.py:48-62) and the layout names its drivers pass
(tests/test_load.py:49-52, Makefile:6).

Legend (overcooked_config.py:34-47):
  ``#`` wall, ``C`` counter, ``@`` delivery zone, ``=`` plate stack, ``O`` onion stack,
  ``U`` pot, space = floor, ``+`` = floor that is also a spawn point.

The four ``overcooked_*_v0`` grids are the standard Overcooked-AI rooms wrapped in a wall
border; they ship inside the third-party cogrid package, not in the reference tree, so
their text here is restated from the published layouts (cramped_room is confirmed cell by
cell by the reference's states/agent_0_dropping_onion_state.npy).
"""
from __future__ import annotations

_LAYOUTS: dict[str, list[str]] = {}


def register_layout(name: str, rows: list[str]) -> None:
    if not rows or any(len(r) != len(rows[0]) for r in rows):
        raise ValueError(f"layout {name!r} must be a non-empty rectangle")
    if len(rows) * len(rows[0]) > 256:
        raise ValueError(f"layout {name!r} has more than 256 cells")
    bad = set("".join(rows)) - set("#C@=OU +")
    if bad:
        raise ValueError(f"layout {name!r} uses unknown characters {sorted(bad)}")
    _LAYOUTS[name] = list(rows)


def get_layout(name: str) -> list[str]:
    try:
        return list(_LAYOUTS[name])
    except KeyError:
        raise KeyError(f"unknown layout {name!r}; registered: {sorted(_LAYOUTS)}") from None


def layout_names() -> list[str]:
    return sorted(_LAYOUTS)


register_layout(
    "overcooked_cramped_room_v0",
    [
        "#######",
        "#CCUCC#",
        "#O   O#",
        "#C   C#",
        "#C=C@C#",
        "#######",
    ],
)

register_layout(
    "overcooked_coordination_ring_v0",
    [
        "#######",
        "#CCCUC#",
        "#C   U#",
        "#= C C#",
        "#O   C#",
        "#CO@CC#",
        "#######",
    ],
)

# The two halves of this room are disconnected; spawn points keep one agent on each side.
register_layout(
    "overcooked_forced_coordination_v0",
    [
        "#######",
        "#CCCUC#",
        "#O C+U#",
        "#O+C C#",
        "#= C C#",
        "#CCC@C#",
        "#######",
    ],
)

register_layout(
    "overcooked_counter_circuit_v0",
    [
        "##########",
        "#CCCUUCCC#",
        "#C      C#",
        "#= CCCC @#",
        "#C      C#",
        "#CCCOOCCC#",
        "##########",
    ],
)

register_layout(
    "overcooked_asymmetric_advantages_v0",
    [
        "###########",
        "#CCCCCCCCC#",
        "#O C@COC @#",
        "#C   U   C#",
        "#C   U   C#",
        "#CCC=C=CCC#",
        "###########",
    ],
)

# overcooked_config.py:48-62
register_layout(
    "large_overcooked_layout",
    [
        "#################",
        "#C@CC=CCCCCCCUUC#",
        "#C  C     C    C#",
        "#C  C COO C    C#",
        "#C    CCCCC    C#",
        "#C             C#",
        "#C   CCCCCC    C#",
        "#C   CCOOCC C  C#",
        "#C   C      C  C#",
        "#CUUCCCCCCC=CC@C#",
        "#################",
    ],
)
