# import-only stand-in (rendertools defines a record class at import time); never used on the step path
import collections


def recordtype(name, fields):
    return collections.namedtuple(name, fields)
