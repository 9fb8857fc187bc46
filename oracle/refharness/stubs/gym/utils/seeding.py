"""Stand-in for gym==0.14.0 ``gym.utils.seeding`` (pinned by
flatland-rl/requirements_dev.txt:21, absent from this image).

PARITY UNPINNED for the seed -> MT19937 key mapping: this is a restatement
from the published gym 0.14 algorithm (SHA-512 of str(seed), first 8 bytes,
little-endian 32-bit limbs -> RandomState.seed(list)).  Golden fixtures
therefore always carry the *post-reset* MT19937 state, which fully determines
RailEnv.step(); nothing in the product depends on this mapping.
"""
import hashlib
import os
import struct

import numpy as np


def _bigint_from_bytes(b):
    sizeof_int = 4
    padding = sizeof_int - len(b) % sizeof_int
    b += b"\0" * padding
    int_count = int(len(b) / sizeof_int)
    unpacked = struct.unpack("{}I".format(int_count), b)
    accum = 0
    for i, val in enumerate(unpacked):
        accum += 2 ** (sizeof_int * 8 * i) * val
    return accum


def _int_list_from_bigint(bigint):
    if bigint < 0:
        raise ValueError("Seed must be non-negative, not {}".format(bigint))
    elif bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def create_seed(a=None, max_bytes=8):
    if a is None:
        a = _bigint_from_bytes(os.urandom(max_bytes))
    elif isinstance(a, str):
        a = a.encode("utf8")
        a += hashlib.sha512(a).digest()
        a = _bigint_from_bytes(a[:max_bytes])
    elif isinstance(a, int):
        a = a % 2 ** (8 * max_bytes)
    else:
        raise ValueError("Invalid type for seed: {} ({})".format(type(a), a))
    return a


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(h[:max_bytes])


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, int) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
