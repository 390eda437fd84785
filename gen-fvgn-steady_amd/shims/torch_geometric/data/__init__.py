from gfv.graph import Data  # noqa: F401
