from .defaults import cfg, CfgNode, path_config  # noqa: F401
