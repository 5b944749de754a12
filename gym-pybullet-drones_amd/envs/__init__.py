from .BaseAviary import BaseAviary
from .BaseRLAviary import BaseRLAviary
from .CtrlAviary import CtrlAviary
from .HoverAviary import HoverAviary
from .MultiHoverAviary import MultiHoverAviary
from .VectorAviary import VectorAviary, VectorCtrlAviary, VectorHoverAviary, VectorMultiHoverAviary

__all__ = ["BaseAviary", "BaseRLAviary", "CtrlAviary", "HoverAviary", "MultiHoverAviary", "VectorAviary",
           "VectorCtrlAviary", "VectorHoverAviary", "VectorMultiHoverAviary"]
