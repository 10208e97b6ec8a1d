"""Enums of the reference's src/types.rs; integer values = declaration order = C-ABI values."""
from enum import IntEnum


class AutoscaleStrategy(IntEnum):  # types.rs:115-123
    Standard = 0
    Robust = 1
    Adaptive = 2
    Equalized = 3
    Clahe = 4
    Tamed = 5
    Default = 6


class BitDepth(IntEnum):  # types.rs:170-173
    U8 = 0
    U16 = 1


class PolarizationOperation(IntEnum):  # types.rs:8-14
    Sum = 0
    Diff = 1
    Ratio = 2
    NDiff = 3
    LogRatio = 4


class SyntheticRgbMode(IntEnum):  # types.rs:177-182
    Default = 0
    RgbRatio = 1
    SarUrban = 2
    Enhanced = 3
