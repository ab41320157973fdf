"""miq — MI355X-native fake-quantization operators behind the reference's torch.nn.Module surface.

Importing the package does not load the native library; the first op call does and raises if
libmiq.so is missing (there is no fallback implementation)."""
from .quantizer import UniformAffineQuantizer  # noqa: F401
from .int_linear import QuantLinear  # noqa: F401
from .int_matmul import QuantMatMul  # noqa: F401
from .omni_norm import OmniLayerNorm, OmniLlamaRMSNorm  # noqa: F401

__all__ = ["UniformAffineQuantizer", "QuantLinear", "QuantMatMul", "OmniLayerNorm", "OmniLlamaRMSNorm"]
