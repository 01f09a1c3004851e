from sow_amd.optimizer import TTSGD  # noqa: F401
