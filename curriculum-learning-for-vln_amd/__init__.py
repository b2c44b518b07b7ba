"""MI355X-native training hot path for the R2R navigation agents of
IMNearth/Curriculum-Learning-For-VLN (tasks/R2R-judy/src/model): hand-written
gfx950 HIP kernels behind a C ABI (include/vln_hip.h), wrapped in nn.Modules
that keep the reference's constructor / forward / state_dict surface.

Import as `import vln_amd` (root shim) -- the directory name is not a Python
identifier."""
from . import _lib, ops, runtime, dp  # noqa: F401
from ._lib import VlnError, LIB_PATH  # noqa: F401
from .encoder import EncoderLSTM  # noqa: F401
from .envdrop_decoder import EnvDropDecoder, Critic  # noqa: F401
from . import functional, staging, losses, optim, metrics, graphs  # noqa: F401
from .runtime import DeviceClock  # noqa: F401
from .graphs import IterationGraph, SegmentedIterationGraph, HandshakeIterationGraph  # noqa: F401
from .staging import DeviceFeatureStore, PinnedStager, HostBatchFeed  # noqa: F401
from .speaker import SpeakerEncoder, SpeakerDecoder, Speaker, back_translate, env_drop_mask  # noqa: F401
from . import synthetic, batches, trainers  # noqa: F401
from .batches import LiveBatch, LiveSteps  # noqa: F401
from .trainers import (EnvDropILIteration, EnvDropA2CIteration, SelfMonitorIteration, FollowerIteration,  # noqa: F401
                       SpeakerIteration)
from .decoders import (SoftDotAttention, VisualSoftDotAttention, ActionScoring, PositionalEncoding, MLPwithBN,  # noqa: F401
                       AttnDecoderLSTM, MonitorDecoder)

__all__ = ["_lib", "ops", "runtime", "dp", "functional", "VlnError", "LIB_PATH", "EncoderLSTM", "EnvDropDecoder",
           "Critic", "SoftDotAttention", "VisualSoftDotAttention", "ActionScoring", "PositionalEncoding", "MLPwithBN",
           "AttnDecoderLSTM", "MonitorDecoder", "SpeakerEncoder", "SpeakerDecoder", "Speaker", "back_translate",
           "env_drop_mask", "synthetic", "batches", "trainers", "LiveBatch", "LiveSteps", "EnvDropILIteration",
           "EnvDropA2CIteration", "SelfMonitorIteration", "FollowerIteration", "SpeakerIteration"]
