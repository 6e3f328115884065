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
    train_data.view(T, n_uav, 12)[timestep_indices[i], uav_indices[i]]  (PMINet.py:78-84).

    The index triples come from torch's CPU generator -- the global one unless `generator` is given -- in the reference's
    order (timestep_indices, then uav_indices; PMINet.py:80-81), so under the same `torch.manual_seed` this selects the very
    rows `PMINetwork.train_pmi` selects (tests/golden/f3_pmi_train.npz); they are then moved to the data's device, where
    the gather runs.  A device generator draws there instead (no host round trip, another stream)."""
    D = train_data.shape[-1]
    data = train_data.reshape(-1, n_uav, D)
    T = data.shape[0]
    dev = data.device
    draw_dev = generator.device if generator is not None else torch.device("cpu")
    t_idx = torch.randint(low=0, high=T, size=(b2_size,), device=draw_dev, generator=generator).to(dev)
    u_idx = torch.randint(low=0, high=n_uav, size=(b2_size, 2), device=draw_dev, generator=generator).to(dev)
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


def train_pmi_epoch(net, optimizer, train_data: torch.Tensor, n_uav: int, b2_size: int, batch_size: int,
                    generator: Optional[torch.Generator] = None) -> float:
    """One call of `PMINetwork.train_pmi` (PMINet.py:74-100) for a network that lives where `train_data` lives: train
    mode, the (timestep, uav-pair) selection above, b2_size // batch_size mini-batches of CustomLoss + optimizer step,
    returns the mean of |loss| like the reference.  With the reference's initial weights, Adam(lr=1e-3) and the same
    torch.manual_seed it reproduces the reference's batches exactly and its losses to fp32 rounding
    (tests/golden/f3_pmi_train.npz).  The optimiser is the caller's (learner side)."""
    net.train()
    selected, _, _ = sample_pmi_pairs(train_data, n_uav, b2_size, generator=generator)
    total, n = 0.0, 0
    for x12, x13 in pmi_batches(selected, batch_size):
        optimizer.zero_grad()
        loss = pmi_contrastive_loss(net(x12), net(x13))
        total += abs(float(loss.item()))
        loss.backward()
        optimizer.step()
        n += 1
    return total / max(n, 1)
