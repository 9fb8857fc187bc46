"""import shim over flatland_marl_amd (see flatland_marl_amd/shim/__init__.py); mirrors flatland-rl 3.0.15's module names"""
__version__ = "3.0.15"
