"""MI355X-native batched Inversus environment + rollout path.

Hand-written HIP kernels for gfx950 (csrc/) behind a C ABI (include/inversus_hip.h); this
package is the thin Python host side that mirrors the reference's env_wrappers / ppo_agent
interface.  Importing the package does not touch the GPU; constructing an env does, and
raises if the HIP library or the GPU is missing (there is no CPU fallback).
"""
from .vocab import Action, ActionType, Bullet, Direction, PlayerId, PlayerState, TileColor  # noqa: F401

__all__ = ["Action", "ActionType", "Bullet", "Direction", "PlayerId", "PlayerState", "TileColor",
           "MultiEnvRunner", "SingleInversusRLEnv", "build_observation", "discrete_to_action", "dummy_opponent_policy",
           "DeviceEnvBatch", "gae"]


def __getattr__(name):          # lazy: keep `import inversus_reinforcement_learning_amd` torch-free and fast
    if name in ("MultiEnvRunner", "SingleInversusRLEnv", "build_observation", "discrete_to_action", "dummy_opponent_policy"):
        from . import env_wrappers
        return getattr(env_wrappers, name)
    if name in ("DeviceEnvBatch", "gae"):
        from . import device_env
        return getattr(device_env, name)
    raise AttributeError(name)
