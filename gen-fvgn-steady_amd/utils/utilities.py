"""NodeType enum of the reference (utils/utilities.py:7-13)."""
import enum


class NodeType(enum.IntEnum):
    NORMAL = 0
    INFLOW = 1
    OUTFLOW = 2
    WALL_BOUNDARY = 3
    PRESS_POINT = 4
    IN_WALL = 5
