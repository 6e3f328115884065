import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "marl-uavs-targets-tracking_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import json
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


@pytest.fixture(scope="session")
def pmi_state_dict():
    z = np.load(os.path.join(GOLDEN, "pmi_h128.npz"))
    return {k: z[k] for k in z.files if k != "meta"}


@pytest.fixture(scope="session")
def pmi_state_dict_h64():
    """PMINetwork at its class-default width (PMINet.py:21), recorded from the reference (gen_golden.py: gen_h64)."""
    z = np.load(os.path.join(GOLDEN, "pmi_h64.npz"))
    return {k: z[k] for k in z.files if k != "meta"}



def pmi_forward_fp64(sd, x, eps=1e-5, want_scale=False):
    """PMINetwork.forward in eval mode (PMINet.py:41-62) in fp64 numpy, from an (unfolded) state dict: x [n, 12] -> scores [n].
    want_scale also returns, per row, sum_c |w2_c| (sum_k |h_k W1_kc| + |b1_c|) + |b2|: the magnitude an fp32 evaluation's
    rounding error scales with (the score itself may be far smaller when terms cancel)."""
    g = lambda k: np.asarray(sd[k], dtype=np.float64)

    def layer(lin, bn, inp):
        scale = g(bn + ".weight") / np.sqrt(g(bn + ".running_var") + eps)
        w = g(lin + ".weight") * scale[:, None]                   # [out, in], BatchNorm folded (eval mode: an affine map)
        b = (g(lin + ".bias") - g(bn + ".running_mean")) * scale + g(bn + ".bias")
        return inp @ w.T + b, np.abs(inp) @ np.abs(w.T) + np.abs(b)

    x = np.asarray(x, dtype=np.float64)
    parts = [np.maximum(layer(l, b, x[:, lo:hi])[0], 0.0) for l, b, lo, hi in
             (("fc_comm", "bn_comm", 0, 5), ("fc_obs", "bn_obs", 5, 9), ("fc_boundary_state", "bn_boundary_state", 9, 12))]
    h = np.concatenate(parts, axis=1)
    z1, mag1 = layer("fc1", "bn1", h)
    w2, b2 = g("fc2.weight").reshape(-1), g("fc2.bias").reshape(-1)[0]
    s = np.maximum(z1, 0.0) @ w2 + b2
    if want_scale:
        return s, mag1 @ np.abs(w2) + abs(b2)
    return s


def adversarial_pmi_state_dict(hidden, seed=0):
    """Weights built to stress the scorer's three-way bf16 split (csrc/pmi_kernel.hip, pmi_score_x6_kernel): fc1 entries
    log-uniform over 2^-20 .. 2^4 with random signs, and hidden units in adjacent PAIRS whose branch-layer rows are equal
    (equal activations) while their fc1 columns are almost opposite -- every 3H-term sum is a large cancellation with a
    small remainder.  BatchNorm statistics non-trivial.  Returns an unfolded PMINetwork-shaped state dict (fp32)."""
    r = np.random.RandomState(seed)
    H = hidden
    sd = {}
    for lin, bn, fan_in in (("fc_comm", "bn_comm", 5), ("fc_obs", "bn_obs", 4), ("fc_boundary_state", "bn_boundary_state", 3)):
        w = r.uniform(-1.0, 1.0, (H // 2, fan_in))
        b = r.uniform(0.0, 1.0, H // 2)                            # mostly positive pre-activations: the ReLU passes them
        sd[lin + ".weight"] = np.repeat(w, 2, axis=0).astype(np.float32)      # units 2m and 2m + 1 are twins
        sd[lin + ".bias"] = np.repeat(b, 2).astype(np.float32)
        sd[bn + ".weight"] = np.repeat(r.uniform(0.5, 1.5, H // 2), 2).astype(np.float32)
        sd[bn + ".bias"] = np.repeat(r.randn(H // 2) * 0.2, 2).astype(np.float32)
        sd[bn + ".running_mean"] = np.repeat(r.randn(H // 2) * 0.3, 2).astype(np.float32)
        sd[bn + ".running_var"] = np.repeat(r.uniform(0.5, 2.0, H // 2), 2).astype(np.float32)
    mag = np.exp2(r.uniform(-20.0, 4.0, (H, 3 * H // 2)))
    w_even = mag * r.choice([-1.0, 1.0], size=mag.shape)
    w1 = np.empty((H, 3 * H))
    w1[:, 0::2] = w_even
    w1[:, 1::2] = -w_even * (1.0 + r.uniform(-1e-3, 1e-3, mag.shape))       # the twin's column: almost the negative
    sd["fc1.weight"] = w1.astype(np.float32)
    sd["fc1.bias"] = r.uniform(-0.5, 0.5, H).astype(np.float32)
    sd["bn1.weight"] = r.uniform(0.5, 1.5, H).astype(np.float32)
    sd["bn1.bias"] = (r.randn(H) * 0.2).astype(np.float32)
    sd["bn1.running_mean"] = (r.randn(H) * 0.3).astype(np.float32)
    sd["bn1.running_var"] = r.uniform(0.5, 2.0, H).astype(np.float32)
    sd["fc2.weight"] = (r.uniform(-1.0, 1.0, (1, H)) / np.sqrt(H)).astype(np.float32)
    sd["fc2.bias"] = r.uniform(-0.1, 0.1, 1).astype(np.float32)
    return sd
