from gfv.scatter import global_add_pool  # noqa: F401
