"""Tile layouts of the push tasks (data restated from
``robovat/envs/push/layouts.py:25-245``).

Tiles are addressed as (row, col) on a 0.15 m grid whose (0, 0) centre sits at
``offset`` = (0.6 - 0.76/2 + 0.075, -1.22/2 + 0.05 + 0.075); a tile list is
written here as ``{row: [cols...]}`` and expanded on import.  Field names match
the reference ``PushLayout`` tuple.
"""
import collections

PushLayout = collections.namedtuple(
    'PushLayout',
    'size offset region goal target obstacle region_rgba goal_rgba')

SIZE = 0.15
OFFSET = [0.295, -0.485]

_BLUE = [0.4667, 0.7098, 0.9961, 1]
_RED = [1, .4235, .4235, 1]
_SAND = [0.867, 0.776, 0.678, 0]
_GREY = [0.8, 0.8, 0.8, 1]
_YELLOW = [1, 0.9412, 0.4235, 1]


def _tiles(rows):
    if rows is None:
        return None
    return [[r, c] for r in sorted(rows) for c in rows[r]]


def _layout(region, goal, target, obstacle, region_rgba, goal_rgba):
    return PushLayout(SIZE, list(OFFSET), _tiles(region), _tiles(goal),
                      _tiles(target), _tiles(obstacle), region_rgba, goal_rgba)


_C0 = {0: [2, 3, 4, 5], 1: [2, 3, 4, 5], 2: [2, 3, 4, 5]}
_C0T = {1: [3, 4], 2: [3, 4]}
_C1 = {1: [2, 3, 4, 5], 2: [2, 3, 4, 5]}
_C2 = {0: [2, 3, 4, 5], 1: [2, 3, 4, 5], 2: [3, 4]}
_I0 = {0: [0, 1], 1: [0, 1], 2: [0], 3: [0, 1], 4: [0, 1]}
_I0O = {1: [3, 4], 2: [3, 4], 3: [3, 4]}
_I2 = {3: [1, 2, 5, 6], 4: [1, 2, 3, 4, 5, 6]}
_I2O = {1: [2, 3, 4, 5], 2: [2, 3, 4, 5]}
_XO = {1: [1, 2, 3, 4, 5, 6], 2: [1, 2, 3, 4, 5, 6], 3: [1, 2, 3, 4, 5, 6]}

TASK_NAME_TO_LAYOUTS = {
    'clearing': [
        _layout(_C0, None, _C0T, _C0T, _BLUE, None),
        _layout(_C1, None, _C1, _C1, _BLUE, None),
        _layout(_C2, None, _C2, _C2, _BLUE, None),
    ],
    'insertion': [
        _layout(_I0, {2: [1]}, {2: [3, 4]}, _I0O, _RED, _SAND),
        _layout(_I0, {2: [1]}, {2: [3, 4]}, _I0O, _RED, _SAND),
        _layout(_I2, {2: [1]}, {1: [2, 3, 4, 5]}, _I2O, _RED, _SAND),
    ],
    'crossing': [
        _layout({0: [0, 2, 5], 1: [0, 1, 2, 5], 2: [2, 3, 4, 5], 3: [2]},
                {1: [2]}, {2: [5]}, _XO, _GREY, _YELLOW),
        _layout({0: [0, 1, 2, 5], 1: [0, 2, 3, 4, 5], 2: [0, 2, 5], 3: [2, 5]},
                {3: [2]}, {1: [4, 5]}, _XO, _GREY, _YELLOW),
        _layout({0: [2, 5], 1: [2, 5, 6], 2: [1, 2, 3, 4, 5], 3: [1, 2, 5]},
                {1: [6]}, {1: [2], 2: [1, 2]}, _XO, _GREY, _YELLOW),
    ],
}
