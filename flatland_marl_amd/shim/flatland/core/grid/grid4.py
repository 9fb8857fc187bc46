"""flatland.core.grid.grid4.Grid4TransitionsEnum (grid4.py:9-24)"""
from enum import IntEnum


class Grid4TransitionsEnum(IntEnum):
    NORTH = 0
    EAST = 1
    SOUTH = 2
    WEST = 3

    @staticmethod
    def to_char(int):
        return {0: "N", 1: "E", 2: "S", 3: "W"}[int]
