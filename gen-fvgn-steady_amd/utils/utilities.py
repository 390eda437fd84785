"""`utils.utilities` of the reference, hot-path part (utils/utilities.py:7-57): the NodeType enum and the two node <-> cell
scatter helpers, on libgfv's segmented reduce."""
import enum


class NodeType(enum.IntEnum):
    NORMAL = 0
    INFLOW = 1
    OUTFLOW = 2
    WALL_BOUNDARY = 3
    PRESS_POINT = 4
    IN_WALL = 5


def calc_cell_centered_with_node_attr(node_attr, cells_node, cells_index, reduce="mean", map=True):
    """utilities.py:16-35: reduce the node values of each cell's nodes (map=True gathers them first)."""
    from gfv.scatter import scatter
    if cells_node.shape != cells_index.shape:
        raise ValueError("wrong cells_node/cells_index dim")
    if len(cells_node.shape) > 1:
        cells_node = cells_node.view(-1)
    if len(cells_index.shape) > 1:
        cells_index = cells_index.view(-1)
    mapped = node_attr[cells_node] if map else node_attr
    return scatter(src=mapped, index=cells_index, dim=0, reduce=reduce)


def calc_node_centered_with_cell_attr(cell_attr, cells_node, cells_index, reduce="mean", map=True):
    """utilities.py:38-57: reduce the cell values of the cells around each node."""
    from gfv.scatter import scatter
    if cells_node.shape != cells_index.shape:
        raise ValueError("wrong cells_node/cells_index dim ")
    if len(cells_node.shape) > 1:
        cells_node = cells_node.view(-1)
    if len(cells_index.shape) > 1:
        cells_index = cells_index.view(-1)
    mapped = cell_attr[cells_index] if map else cell_attr
    return scatter(src=mapped, index=cells_node, dim=0, reduce=reduce)
