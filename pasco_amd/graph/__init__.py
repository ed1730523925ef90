"""PaSCo's sparse U-Net + mask-transformer graph on the HIP operator library (inference)."""
from .unet import PascoNet, UNet3DV2, CylinderFeat, merge_subnet_inputs  # noqa: F401
from .transformer import TransformerPredictorV2  # noqa: F401
from .ensemble import Ensembler  # noqa: F401
from .panoptic import panoptic_inference  # noqa: F401
