"""Enums and the Action record of the reference's public API (inversus/game_types.py:8-49),
kept so that `discrete_to_action` and state views return the same vocabulary.  The integer
codes are the SoA encodings of include/inversus_hip.h."""
from dataclasses import dataclass
from enum import Enum
from typing import Optional


class TileColor(Enum):
    BLACK = "black"      # grid bit 0
    WHITE = "white"      # grid bit 1


class Direction(Enum):
    UP = "up"
    DOWN = "down"
    LEFT = "left"
    RIGHT = "right"


class ActionType(Enum):
    MOVE = "move"
    SHOOT = "shoot"
    CHARGE_SHOOT = "charge_shoot"
    NONE = "none"


class PlayerId(Enum):
    P1 = 1
    P2 = 2


#: observation-channel order of directions (env_wrappers.py:216-221) == device `dir` code
DIR_BY_CODE = (Direction.UP, Direction.RIGHT, Direction.DOWN, Direction.LEFT)
CODE_BY_DIR = {d: i for i, d in enumerate(DIR_BY_CODE)}
TYPE_BY_CODE = (ActionType.NONE, ActionType.MOVE, ActionType.SHOOT, ActionType.CHARGE_SHOOT)


@dataclass
class Action:
    type: ActionType
    direction: Optional[Direction] = None

    def __post_init__(self):
        needs_dir = self.type in (ActionType.MOVE, ActionType.SHOOT, ActionType.CHARGE_SHOOT)
        if needs_dir and self.direction is None:
            raise ValueError(f"Action type {self.type} requires a direction")
        if self.type == ActionType.NONE and self.direction is not None:
            raise ValueError("Action type NONE must have direction=None")


@dataclass
class PlayerState:
    player_id: PlayerId
    x: int
    y: int
    color: TileColor
    ammo: int
    max_ammo: int
    reload_counter: int
    alive: bool = True


@dataclass
class Bullet:
    x: int
    y: int
    dir: Direction
    owner: PlayerId
