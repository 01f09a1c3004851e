from sow_amd.layer import SoWLinear  # noqa: F401
from sow_amd.prepare import SoWConfig, SoWModel, accumulate, export_alignment, load_sow, prepare_sow  # noqa: F401
