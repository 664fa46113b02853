"""MI355X-native (gfx950) lip->speech inference hot path, drop-in for DomhnallBoyle/lip2speech-unit's
multi_target_lip2speech (AV-HuBERT variant) + multi_input_vocoder inference path.  See DESIGN.md."""
__version__ = "0.1.0"
