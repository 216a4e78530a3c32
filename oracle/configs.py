"""Model configurations and the deterministic synthetic checkpoint are shared data (single source:
ladi_vton_amd/configs.py); re-exported here for the oracle and the tests."""
from ladi_vton_amd.configs import *  # noqa: F401,F403
from ladi_vton_amd.configs import _is_norm  # noqa: F401
