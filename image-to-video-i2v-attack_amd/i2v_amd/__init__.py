"""MI355X-native I2V adversarial-perturbation engine -- host side.

Python mirror of the reference's attack-class API (`image_attacks.py`, `TPAMI_attack.py`,
`base_attacks.py`) over the C ABI of `libi2v_hip.so` (hand-written HIP for gfx950)."""
__all__ = ["graphs", "weights", "lib", "engine"]
