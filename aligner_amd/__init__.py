"""aligner_amd -- MI355X-native DP matrix fill + traceback behind ikramanop/aligner's aligner API.

Scope: ONE hot path of the reference -- `perform_alignment` of SimpleGlobalAligner / SimpleLocalAligner
(aligner-core/src/simple/mod.rs) and the legacy SimpleAligner (src/align/aligner_core.rs) -- as hand-written
HIP kernels for gfx950 behind the C ABI of include/aligner_hip.h, plus the batch driver that shards independent
pairs over GPUs.  Importing the package does not touch the GPU; the first alignment call creates the context and
fails loudly if the native library or the device is missing.
"""
from .alignment import Alignment, AlignmentResult
from .batch import BatchResult, PairBatch, StagedBatch, align_batch
from .enums import DNA, Direction, Protein
from .errors import AlignerError, DeviceError, ErrorKind, ReferencePanic
from .heuristic import HeuristicAligner, HeuristicPWMAligner, get_threshold, transform_matrix
from .legacy import SimpleAligner
from .matrices import get_blosum62, nucleotide_matrix
from .pwm import PWMAligner, PWMAlignment, align_windows
from .simple import Heuristics, SimpleGlobalAligner, SimpleLocalAligner

__all__ = [
    "Alignment", "AlignmentResult", "BatchResult", "PairBatch", "StagedBatch", "align_batch", "DNA", "Direction",
    "Protein", "AlignerError", "DeviceError", "ErrorKind", "ReferencePanic", "SimpleAligner", "get_blosum62",
    "nucleotide_matrix", "PWMAligner", "PWMAlignment", "align_windows", "Heuristics", "SimpleGlobalAligner", "SimpleLocalAligner",
    "HeuristicAligner", "HeuristicPWMAligner", "get_threshold", "transform_matrix",
]
