from .BaseControl import BaseControl
from .DSLPIDControl import DSLPIDControl, DSLPIDControlBatch

__all__ = ["BaseControl", "DSLPIDControl", "DSLPIDControlBatch"]
