"""The public vocabulary of the reference's API (inversus/game_types.py:8-71) -- enum members and
record fields that `discrete_to_action` and the state views hand back -- tied to the integer codes of
the device SoA (include/inversus_hip.h).  Names and values match the reference so that callers'
comparisons (`action.type == ActionType.MOVE`, `tile == TileColor.WHITE`) keep working."""
from dataclasses import dataclass, field, make_dataclass
from enum import Enum
from typing import Optional

TileColor = Enum("TileColor", {"BLACK": "black", "WHITE": "white"})            # grid bit 0 / 1
Direction = Enum("Direction", {"UP": "up", "DOWN": "down", "LEFT": "left", "RIGHT": "right"})
ActionType = Enum("ActionType", {"MOVE": "move", "SHOOT": "shoot", "CHARGE_SHOOT": "charge_shoot", "NONE": "none"})
PlayerId = Enum("PlayerId", {"P1": 1, "P2": 2})
for _e in (TileColor, Direction, ActionType, PlayerId):
    _e.__module__ = __name__

#: device `dir` code -> Direction, in observation-channel order (env_wrappers.py:216-221)
DIR_BY_CODE = (Direction.UP, Direction.RIGHT, Direction.DOWN, Direction.LEFT)
CODE_BY_DIR = {d: i for i, d in enumerate(DIR_BY_CODE)}
#: (action_id + 3) >> 2 -> ActionType
TYPE_BY_CODE = (ActionType.NONE, ActionType.MOVE, ActionType.SHOOT, ActionType.CHARGE_SHOOT)
_DIRECTED = frozenset(TYPE_BY_CODE[1:])


@dataclass
class Action:
    type: ActionType
    direction: Optional[Direction] = None

    def __post_init__(self):
        # same consistency rule and messages as the reference (game_types.py:42-49)
        if self.type in _DIRECTED and self.direction is None:
            raise ValueError(f"Action type {self.type} requires a direction")
        if self.type is ActionType.NONE and self.direction is not None:
            raise ValueError("Action type NONE must have direction=None")


PlayerState = make_dataclass("PlayerState", [("player_id", PlayerId), ("x", int), ("y", int), ("color", TileColor),
                                             ("ammo", int), ("max_ammo", int), ("reload_counter", int),
                                             ("alive", bool, field(default=True))])
Bullet = make_dataclass("Bullet", [("x", int), ("y", int), ("dir", Direction), ("owner", PlayerId)])
for _c in (PlayerState, Bullet):
    _c.__module__ = __name__
