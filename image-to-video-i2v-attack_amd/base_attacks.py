"""Drop-in for the sign/clip/project update family of `/root/reference/base_attacks.py`
(FGSM / BIM / MIFGSM): the white-box model and its gradient stay the caller's torch module, the
L_inf update runs in `libi2v_hip.so`."""
from i2v_amd.sign_attacks import FGSM, BIM, MIFGSM, norm_grads  # noqa: F401
