"""Drop-in for the reference module of the same name (`/root/reference/image_attacks.py`):
`getattr(image_attacks, args.attack_method)(...)` (`image_main.py:68,71,80`) resolves to the
MI355X-native classes."""
from i2v_amd.attacks import (Attack, ImageGuidedFMDirection_Adam, ImageGuidedFML2_Adam_MultiModels,  # noqa: F401
                             ImageGuidedStd_Adam)
from i2v_amd.sign_attacks import ILAF  # noqa: F401
