"""Error surface of the hot path.

Mirrors `aligner_core::Error` / `Result` (aligner-core/src/lib.rs:47-59).  Conditions on which the
reference *panics* (empty sequence, residue code outside the matrix, local alignment with no positive
cell: simple/mod.rs:103-104, :85/:198, :214-215) surface as ReferencePanic so a caller can tell them
from the recoverable `Err(..)` values.
"""
from enum import Enum


class ErrorKind(Enum):
    ProteinNotFound = "ProteinNotFound"
    CharIsNotMatchable = "CharIsNotMatchable"
    UnnecessaryArgument = "UnnecessaryArgument"
    MissingArgument = "MissingArgument"
    ResultIsEmpty = "ResultIsEmpty"
    CalculationError = "CalculationError"
    ValidationError = "ValidationError"
    MatrixShapeError = "MatrixShapeError"


class AlignerError(Exception):
    """`Err(aligner_core::Error::<kind>)`."""

    def __init__(self, kind, detail=""):
        super().__init__(kind.value + (": " + detail if detail else ""))
        self.kind = kind


class ReferencePanic(RuntimeError):
    """The reference would have panicked here (status code kept in .status)."""

    def __init__(self, status, detail):
        super().__init__(detail)
        self.status = status


class DeviceError(RuntimeError):
    """HIP runtime / kernel failure reported by the native library."""

    def __init__(self, status, detail):
        super().__init__(detail)
        self.status = status
