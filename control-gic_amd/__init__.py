"""control_gic_amd -- MI355X (gfx950) implementation of Control-GIC's
granularity-adaptive VQ + router + entropy-coder hot path behind the
reference's own module API (see DESIGN.md / INTEGRATION.md).

Import name: `control_gic_amd` (the directory is `control-gic_amd/`; the
root-level control_gic_amd.py maps one to the other).
"""
from . import _lib
from ._lib import CgicError, LIB_PATH
from .quantize import VectorQuantize2, VectorQuantizer
from .router import TripleGrainFixedEntropyRouter
from .entropy import Entropy, entropy_maps, entropy_maps_u8, entropy_maps_tiles
from .indices_coding import HuffmanCoding
from .mask_coding import BinaryCoding
from .codec import GrainCodec, CompressedBatch, mode_streams, STREAM_NAMES, decoder_mode
from . import pipeline, highres, container, model, ops, experimental
from .pipeline import HotPathPipeline, LaneStream, GraphLanes, capture_graph
from .model import install, compress_batch, grain_merge, avg_pool, decoder_blend_medium, decoder_blend_fine

__all__ = ["VectorQuantize2", "VectorQuantizer", "TripleGrainFixedEntropyRouter", "Entropy", "entropy_maps", "entropy_maps_u8", "entropy_maps_tiles",
           "HuffmanCoding", "BinaryCoding", "GrainCodec", "CompressedBatch", "mode_streams", "STREAM_NAMES",
           "HotPathPipeline", "LaneStream", "GraphLanes", "capture_graph", "decoder_mode", "install", "compress_batch", "grain_merge", "avg_pool", "decoder_blend_medium", "decoder_blend_fine", "highres", "container", "CgicError", "LIB_PATH"]
