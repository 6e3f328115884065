"""uavtrack -- MI355X-native batched multi-UAV target-tracking environment.

Host side (Python, like the reference) of the C-ABI library libuavtrack.so whose HIP
kernels replace the reset/step path of the reference's src/environment.py.
There is no CPU fallback: without the built library and a gfx950 GPU every
compute entry point raises.
"""
from .config import EnvConfig, RewardMode  # noqa: F401
from .env import BatchedUavEnv  # noqa: F401
from .compat import Environment  # noqa: F401
from .pmi import fold_pmi_state_dict, make_pmi_net  # noqa: F401
from .sharding import (shard_range, gather_rollout_summary, gather_rollout_summary_async,  # noqa: F401
                       sample_local_transitions, gather_transitions, gather_transitions_async)
from .rollout import ActorMLP, BatchedRollout, sample_actions  # noqa: F401
from .export import (uav_tracks_from_obs, save_uav_positions, save_covered_num, target_tracks,  # noqa: F401
                     save_target_positions, save_rollout)
from .replay import DeviceReplayBuffer, PrioritizedDeviceReplayBuffer, transitions_from_rollout  # noqa: F401
from .pmi_data import sample_pmi_pairs, pmi_contrastive_loss, pmi_batches, train_pmi_epoch  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["EnvConfig", "RewardMode", "BatchedUavEnv", "Environment", "fold_pmi_state_dict",
           "shard_range", "gather_rollout_summary", "gather_rollout_summary_async", "sample_local_transitions", "gather_transitions",
           "gather_transitions_async", "ActorMLP", "BatchedRollout", "sample_actions",
           "DeviceReplayBuffer", "PrioritizedDeviceReplayBuffer", "transitions_from_rollout",
           "sample_pmi_pairs", "pmi_contrastive_loss", "pmi_batches", "train_pmi_epoch", "make_pmi_net"]
