"""Synthetic, counter-based action streams shared by bench.py, the tests and the
golden-vector capture script (oracle/refharness/capture_golden.py).

The stream is a pure function of (seed, env replica b, step t, agent a) so that the
reference capture here, the CPU oracle and the HIP kernel (csrc/fl_kernels.hip:
``synth_action``) all see identical actions without shipping action tensors around.
"""
import numpy as np

M32 = 0xFFFFFFFF


def mix32(x):
    """murmur3 finaliser on uint32 numpy arrays / python ints."""
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & M32
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & M32
    x ^= x >> 16
    return x.astype(np.uint32)


def action_hash(seed, b, t, a):
    """uint32 hash of (seed, b, t, a); broadcasting over numpy inputs."""
    b = np.asarray(b, dtype=np.uint64)
    t = np.asarray(t, dtype=np.uint64)
    a = np.asarray(a, dtype=np.uint64)
    h = (np.uint64(seed) * np.uint64(0x9E3779B1)) & M32
    h = mix32(h ^ ((b * 0x85EBCA77) & M32))
    h = mix32(h.astype(np.uint64) ^ ((t * 0xC2B2AE3D) & M32))
    h = mix32(h.astype(np.uint64) ^ ((a * 0x27D4EB2F) & M32))
    return h


def uniform_actions(seed, b, t, n_agents):
    """u8[n_agents] actions uniform in 0..4 for env replica b at step t (t = step index, 0-based)."""
    a = np.arange(n_agents)
    return (action_hash(seed, b, t, a) % 5).astype(np.uint8)


def forward_biased_actions(seed, b, t, n_agents):
    """80 % MOVE_FORWARD, 5 % LEFT, 5 % RIGHT, 5 % STOP, 5 % DO_NOTHING."""
    a = np.arange(n_agents)
    r = action_hash(seed, b, t, a) % 100
    out = np.full(n_agents, 2, dtype=np.uint8)
    out[r >= 80] = 1
    out[r >= 85] = 3
    out[r >= 90] = 4
    out[r >= 95] = 0
    return out


def spfollow_actions(seed, b, t, state, pos, direction, grid, dm_u16, target_slot, p_stop_percent=3):
    """kind 2 of the on-device stream (csrc/fl_step_body.h): shortest-path following with counter-hash stops.
    state / direction int[A], pos int[A, 2] (row, col; -1 off the map) BEFORE the step, grid u16[H, W],
    dm_u16 [U, H, W, 4] (0xFFFF = unreachable), target_slot int[A]."""
    A = len(state)
    h = action_hash(seed, b, t, np.arange(A))
    out = np.zeros(A, dtype=np.uint8)
    for i in range(A):
        st, d = int(state[i]), int(direction[i])
        r, c = int(pos[i][0]), int(pos[i][1])
        if st == 1:                       # READY_TO_DEPART
            out[i] = 2
        elif st < 3 or st > 5 or r < 0:   # not on the map
            out[i] = 0
        elif int(h[i]) % 100 < p_stop_percent:
            out[i] = 4
        else:
            bits = (int(grid[r, c]) >> ((3 - d) * 4)) & 15
            out[i] = 2
            if bin(bits).count("1") != 1:
                best = None
                for a in (1, 2, 3):
                    nd = (d + a + 2) % 4
                    if not (bits >> (3 - nd)) & 1:
                        continue
                    nr, nc = r + (-1, 0, 1, 0)[nd], c + (0, 1, 0, -1)[nd]
                    if not (0 <= nr < grid.shape[0] and 0 <= nc < grid.shape[1]) or grid[nr, nc] == 0:
                        continue
                    v = int(dm_u16[int(target_slot[i]), nr, nc, nd])
                    if v != 0xFFFF and (best is None or v < best):
                        best, out[i] = v, a
    return out
