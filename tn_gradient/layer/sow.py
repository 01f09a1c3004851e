from sow_amd.layer import SoWLinear, SoWParameter  # noqa: F401
from sow_amd.utils import qr_weight  # noqa: F401
