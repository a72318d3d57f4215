"""MI355X-native (gfx950) F5-TTS flow-matching sampling engine with the f5_tts_mlx API surface.

    from f5_tts_mlx_amd import F5TTS            # reference: f5_tts_mlx/__init__.py:1
    from f5_tts_mlx_amd.generate import generate
"""
from .cfm import F5TTS  # noqa: F401
