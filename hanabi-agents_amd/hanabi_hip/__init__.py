"""hanabi_hip — thin Python host layer over the C-ABI of include/hanabi_hip.h.

PyTorch is used only for device memory and streams: every call hands raw device pointers
(`tensor.data_ptr()`) and the current HIP stream to `libhanabi_hip.so` through ctypes.
There is no CPU path: constructing an env or a tree without the compiled library or
without a GPU raises.

Public names: `HanabiEnv`, `SumTree`, `make_config`, `HbConfig`, `lib`, flag constants.
"""
from ._capi import (FLAG_AUTO_RESET, FLAG_LENIENT_REWARD, FLAG_RESET_START_NEXT, GAME_TYPES, HbConfig, HbError,
                    lib, library_path, make_config)
from . import ops
from .env import HanabiEnv
from .tree import SumTree

__all__ = ["ops", "HanabiEnv", "SumTree", "make_config", "HbConfig", "HbError", "lib", "library_path", "GAME_TYPES",
           "FLAG_AUTO_RESET", "FLAG_RESET_START_NEXT", "FLAG_LENIENT_REWARD"]
