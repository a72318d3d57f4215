"""MI355X-native (gfx950) F5-TTS flow-matching sampling engine with the f5_tts_mlx API surface.

    from f5_tts_mlx_amd import F5TTS            # reference: f5_tts_mlx/__init__.py:1
    from f5_tts_mlx_amd.generate import generate
"""
import os as _os

# hipGraph replay on ROCm 7.x: with the runtime's "graph packet capture" (the default) every kernel node of a replayed graph costs
# ~0.3 us more on the GPU than the same kernel launched from a stream -- 1.6-2 ms on the 5 084 nodes of a 32-point sample() at batch 1
# (69.0 vs 67.5 ms; round-6 probe tools/r6_graph_probe.py + tools/gpu_r6_graph.sh, profiles/r06/graph_knobs.jsonl: batch sizes, queue
# counts and one exec per ODE step change nothing, this switch does: 67.6 ms).  The HIP runtime reads the variable when it initialises,
# i.e. at the first HIP call of the process: setting it here works as long as this package is imported before torch touches the GPU
# (an already initialised runtime ignores it; a value the user exported wins).  Non-Python hosts: export it.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from .cfm import F5TTS  # noqa: F401,E402
