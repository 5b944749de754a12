from .BaseAviary import BaseAviary
from .BaseRLAviary import BaseRLAviary
from .CtrlAviary import CtrlAviary
from .HoverAviary import HoverAviary
from .MultiHoverAviary import MultiHoverAviary
from .SwarmAviary import HaloPlan, LocalSwarmGroup, NativeHaloExchange, NativeSlabExchange, TorchHaloExchange, SwarmAviary, TorchSlabExchange, swarm_first_drone, swarm_partition, swarm_spatial_order
from .VelocityAviary import VelocityAviary
from .VectorAviary import (GymVectorEnvAdapter, VecEnvAdapter, VectorAviary, VectorCtrlAviary, VectorHoverAviary, VectorMultiHoverAviary,
                           VectorVelocityAviary)

__all__ = ["BaseAviary", "BaseRLAviary", "CtrlAviary", "HoverAviary", "MultiHoverAviary", "VelocityAviary", "VectorAviary",
           "VectorCtrlAviary", "VectorHoverAviary", "VectorMultiHoverAviary", "VectorVelocityAviary", "VecEnvAdapter", "GymVectorEnvAdapter", "SwarmAviary",
           "LocalSwarmGroup", "NativeSlabExchange", "TorchSlabExchange", "HaloPlan", "NativeHaloExchange", "TorchHaloExchange", "swarm_first_drone", "swarm_partition", "swarm_spatial_order"]
