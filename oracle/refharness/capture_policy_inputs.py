#!/usr/bin/env python3
"""Golden vectors for the policy-input boundary: plfActor.get_feature (solution/plfActor.py:48-74) casts and
Network.modify_adjacency (solution/nn/net_tree.py:105-116), run with the REAL reference functions on cutils outputs
already stored in tests/golden/cfg2_uniform.npz.  Writes tests/golden/policy_inputs.npz.  Build container only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.dont_write_bytecode = True
sys.path[:0] = ["/root/reference/solution", "/root/reference/solution/nn"]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nn.net_tree import Network  # noqa: E402

if __name__ == "__main__":
    fx = np.load(os.path.join(REPO, "tests", "golden", "cfg2_uniform.npz"))
    steps = [0, 10, 40, 70]
    B = len(steps)
    adj = torch.from_numpy(np.stack([fx["o_adjacency"][k] for k in steps])).to(torch.int64)   # [B, A, 30, 3]
    mod = Network.modify_adjacency(None, adj.clone(), "cpu")
    out = os.path.join(REPO, "tests", "golden", "policy_inputs.npz")
    np.savez_compressed(out, obs_index=np.array(steps), adjacency_mod=mod.numpy())
    print(out, mod.shape, int((mod == -2).sum()))
