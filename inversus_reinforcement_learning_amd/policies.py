"""CNN policy of the reference (inversus_rl/policies.py:11-128) as a PyTorch-ROCm module.

Same sub-module names, creation order and shapes, so `state_dict()` keys / tensors are
interchangeable with the reference's checkpoints (`conv{1-4}`, `norm{1-4}`, `fc_actor.{0,2,4}`,
`fc_critic.{0,2,4}`; 10 249 582 parameters at 12x10x15) and `torch.manual_seed(s)` followed by
construction yields the same initial weights.  The observation it consumes is written by the
HIP kernels straight into device memory ([N,12,H,W] f32 + [N,4] f32); nothing leaves the GPU.
"""
import torch
import torch.nn as nn


class InversusCNNPolicy(nn.Module):
    def __init__(self, channels, height, width, extra_dim, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, 32, kernel_size=3, padding=1)
        self.norm1 = nn.LayerNorm([32, height, width])
        self.conv2 = nn.Conv2d(32, 64, kernel_size=3, padding=1)
        self.norm2 = nn.LayerNorm([64, height, width])
        self.conv3 = nn.Conv2d(64, 128, kernel_size=3, padding=1)
        self.norm3 = nn.LayerNorm([128, height, width])
        self.conv4 = nn.Conv2d(128, 128, kernel_size=3, padding=1)
        self.norm4 = nn.LayerNorm([128, height, width])
        self.relu = nn.ReLU()
        self.flatten = nn.Flatten()
        conv_out_dim = 128 * height * width           # 3x3, pad 1 keeps H x W (policies.py:48-57 probes it)
        in_dim = conv_out_dim + extra_dim
        self.fc_actor = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.ReLU(),
                                      nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(),
                                      nn.Linear(hidden_dim // 2, 13))
        self.fc_critic = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.ReLU(),
                                       nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(),
                                       nn.Linear(hidden_dim // 2, 1))

    def trunk(self, grid_tensor):
        x = self.relu(self.norm1(self.conv1(grid_tensor)))
        x = self.relu(self.norm2(self.conv2(x)))
        x = self.relu(self.norm3(self.conv3(x)))
        x = self.relu(self.norm4(self.conv4(x) + x))   # residual before norm4 (policies.py:95-96)
        return self.flatten(x)

    def forward(self, grid_tensor, extra_vector):
        """(B,12,H,W), (B,4) -> logits (B,13), value (B,1)."""
        x = torch.cat([self.trunk(grid_tensor), extra_vector], dim=1)
        return self.fc_actor(x), self.fc_critic(x)


def make_policy_from_env(env):
    """Policy sized from an env's observation (policies.py:111-128).  `env`: a runner / single env
    of this package, or anything with (height, width) via its device batch."""
    batch = getattr(env, "batch", None) or getattr(getattr(env, "_runner", None), "batch", None)
    if batch is None:
        raise TypeError("make_policy_from_env expects a MultiEnvRunner or SingleInversusRLEnv of this package")
    return InversusCNNPolicy(12, batch.H, batch.W, 4)
