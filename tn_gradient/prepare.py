from sow_amd.layer import SoWLinear  # noqa: F401
from sow_amd.prepare import SoWConfig, accumulate, load_sow, prepare_sow  # noqa: F401
