"""Drop-in for the image-model white-box attacks of `/root/reference/base_attacks.py` (selected by name, `attack.py:72`:
`getattr(base_attacks, args.attack_method)(model, steps=args.step)`): FGSM / BIM / MIFGSM (`:236-340`), DIFGSM (`:342-409`), TIFGSM
(`:411-469`), SIM (`:554-610`), TIFGSM3D (`:612-675`), SGM (`:481-551`), TAP (`:685-799`).  The L_inf update, DI's resampling and its transpose, and TI's smoothing run in
`libi2v_hip.so`; the attacked classifier is a native `VideoModel` (then the cross-entropy gradient is native too) or the caller's
torch module."""
from i2v_amd.sign_attacks import FGSM, BIM, MIFGSM, DIFGSM, TIFGSM, TIFGSM3D, SIM, SGM, TAP, norm_grads  # noqa: F401
