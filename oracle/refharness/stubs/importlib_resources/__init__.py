# import-only stand-in: transition_map.py imports `path` for loading packaged PNGs (never used on the step path)
from importlib.resources import path, read_binary  # noqa: F401
