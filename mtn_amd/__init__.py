"""mtn_amd — MI355X-native (gfx950 HIP kernels behind a C ABI) implementation of the MTN transformer hot path,
keeping the reference's make_model()/train.py batch-loop surface.  See DESIGN.md."""
from .mtn import make_model, EncoderDecoder  # noqa: F401
from .data_utils import Batch, LabelSmoothing, NoamOpt, FusedAdam, SimpleLossCompute, subsequent_mask  # noqa: F401

__version__ = "0.1.0"
from .decode import beam_search_decode, beam_search_decode_many, greedy_decode, DecodeSession  # noqa: F401
