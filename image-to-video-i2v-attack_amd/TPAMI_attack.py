"""Drop-in for `/root/reference/TPAMI_attack.py`: the adaptive ENS-I2V class."""
from i2v_amd.attacks import Attack, AENS_I2V_MF  # noqa: F401
