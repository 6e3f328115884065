"""PMI training data path: the sampling / gather half of `PMINetwork.train_pmi`
(reference src/models/PMINet.py:74-100).  The reference draws b2_size (timestep, uav-pair) index triples and
copies the rows one at a time in a Python loop (:83-84); with the rollout's observations already on the device
it is one advanced-indexing gather.  The loss is the reference's CustomLoss (PMINet.py:10-17); the optimiser
step stays with the caller (learner side, out of scope)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def sample_pmi_pairs(train_data: torch.Tensor, n_uav: int, b2_size: int,
                     generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """train_data: [timesteps * n_uav, 12] or [timesteps, (B,) n_uav, 12] observations (train.py:183 order).
    -> (selected [b2_size, 2, 12], timestep_indices [b2_size], uav_indices [b2_size, 2]); selected[i] =
    train_data.view(T, n_uav, 12)[timestep_indices[i], uav_indices[i]]  (PMINet.py:78-84)."""
    D = train_data.shape[-1]
    data = train_data.reshape(-1, n_uav, D)
    T = data.shape[0]
    dev = data.device
    t_idx = torch.randint(0, T, (b2_size,), device=dev, generator=generator)
    u_idx = torch.randint(0, n_uav, (b2_size, 2), device=dev, generator=generator)
    return data[t_idx.unsqueeze(1), u_idx], t_idx, u_idx


def pmi_contrastive_loss(output1: torch.Tensor, output2: torch.Tensor) -> torch.Tensor:
    """CustomLoss.forward (PMINet.py:15-17): mean(log(1 + exp(-o1)) + log(1 + exp(o2))), in the overflow-safe
    softplus form."""
    return torch.mean(torch.nn.functional.softplus(-output1) + torch.nn.functional.softplus(output2))


def pmi_batches(selected: torch.Tensor, batch_size: int):
    """The mini-batches train_pmi iterates (PMINet.py:87-92): b2_size // batch_size full batches of
    (input_1_2 [bs,12], input_1_3 [bs,12])."""
    for i in range(selected.shape[0] // batch_size):
        chunk = selected[i * batch_size:(i + 1) * batch_size]
        yield chunk[:, 0], chunk[:, 1]
