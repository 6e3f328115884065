"""BatchNorm folding of the reference PMINetwork for device inference.

PMINetwork.inference (src/models/PMINet.py:64-72) always runs in eval() mode, so each
Linear -> BatchNorm1d pair (PMINet.py:29-37, 50-60) is an affine map:
    W' = W * gamma / sqrt(var + eps),   b' = (b - mean) * gamma / sqrt(var + eps) + beta.
The blob layout is the one uavtrack_set_pmi_weights documents (input-major)."""
from __future__ import annotations

from typing import Mapping

import numpy as np

BN_EPS = 1e-5   # torch.nn.BatchNorm1d default, PMINet.py:30


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float64)


def pmi_blob_size(hidden: int) -> int:
    H = hidden
    return 12 * H + 3 * H + 3 * H * H + H + H + 1


def fold_pmi_state_dict(sd: Mapping[str, object], eps: float = BN_EPS) -> tuple:
    """state_dict of a reference PMINetwork -> (fp32 blob, hidden)."""
    def fold(lin: str, bn: str):
        w, b = _np(sd[lin + ".weight"]), _np(sd[lin + ".bias"])            # [H, in], [H]
        scale = _np(sd[bn + ".weight"]) / np.sqrt(_np(sd[bn + ".running_var"]) + eps)
        wf = (w * scale[:, None]).T                                         # [in, H]
        bf = (b - _np(sd[bn + ".running_mean"])) * scale + _np(sd[bn + ".bias"])
        return wf, bf

    wc, bc = fold("fc_comm", "bn_comm")
    wo, bo = fold("fc_obs", "bn_obs")
    wb, bb = fold("fc_boundary_state", "bn_boundary_state")
    w1, b1 = fold("fc1", "bn1")
    w2, b2 = _np(sd["fc2.weight"]).reshape(-1), _np(sd["fc2.bias"]).reshape(-1)
    H = wc.shape[1]
    assert wc.shape == (5, H) and wo.shape == (4, H) and wb.shape == (3, H) and w1.shape == (3 * H, H)
    blob = np.concatenate([wc.ravel(), bc, wo.ravel(), bo, wb.ravel(), bb, w1.ravel(), b1, w2, b2])
    assert blob.size == pmi_blob_size(H)
    return np.ascontiguousarray(blob, dtype=np.float32), int(H)


def make_pmi_net(hidden_dim: int = 128):
    """A trainable network with the reference PMINetwork's architecture and parameter names (PMINet.py:20-62:
    three branch Linear+BatchNorm1d+ReLU over x[0:5] / x[5:9] / x[9:12], concat, Linear(3H,H)+BN+ReLU, Linear(H,1)),
    so state_dicts move freely between the two and `BatchedUavEnv.set_pmi(net.state_dict())` folds and uploads
    it.  Training stays plain PyTorch (learner side); see examples/train_maac.py --method maac-r."""
    import torch

    class PmiNet(torch.nn.Module):
        def __init__(self, H: int):
            super().__init__()
            self.fc_comm, self.bn_comm = torch.nn.Linear(5, H), torch.nn.BatchNorm1d(H)
            self.fc_obs, self.bn_obs = torch.nn.Linear(4, H), torch.nn.BatchNorm1d(H)
            self.fc_boundary_state, self.bn_boundary_state = torch.nn.Linear(3, H), torch.nn.BatchNorm1d(H)
            self.fc1, self.bn1 = torch.nn.Linear(3 * H, H), torch.nn.BatchNorm1d(H)
            self.fc2 = torch.nn.Linear(H, 1)

        def forward(self, x):
            relu = torch.relu
            parts = (relu(self.bn_comm(self.fc_comm(x[:, 0:5]))), relu(self.bn_obs(self.fc_obs(x[:, 5:9]))),
                     relu(self.bn_boundary_state(self.fc_boundary_state(x[:, 9:12]))))
            return self.fc2(relu(self.bn1(self.fc1(torch.cat(parts, dim=1)))))

    return PmiNet(hidden_dim)
