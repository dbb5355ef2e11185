from .ehem import EHEM
from .oct_attention import OctAttention

__all__ = ["EHEM", "OctAttention"]
