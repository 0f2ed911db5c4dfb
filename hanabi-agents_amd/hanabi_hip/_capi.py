"""ctypes declarations for every symbol of include/hanabi_hip.h (the drop-in boundary)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_AUTO_RESET, FLAG_RESET_START_NEXT, FLAG_LENIENT_REWARD = 1, 2, 4
STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2


class HbError(RuntimeError):
    pass


class HbConfig(C.Structure):
    """`hb_config` of include/hanabi_hip.h."""

    _fields_ = [(n, C.c_int32) for n in ("players", "colors", "ranks", "hand_size", "max_info", "max_life", "flags")]

    def __repr__(self):
        return "HbConfig(" + ", ".join(f"{n}={getattr(self, n)}" for n, _ in self._fields_) + ")"


# HanabiGame parameter presets of rl_env.make (hanabi_agents/rainbow/run_experiment.py:119-128)
GAME_TYPES = {
    "Hanabi-Full": dict(colors=5, ranks=5, hand_size=lambda p: 5 if p < 4 else 4, max_info=8, max_life=3),
    "Hanabi-Full-CardKnowledge": dict(colors=5, ranks=5, hand_size=lambda p: 5 if p < 4 else 4, max_info=8, max_life=3),
    "Hanabi-Small": dict(colors=2, ranks=5, hand_size=lambda p: 2, max_info=3, max_life=1),
    "Hanabi-Very-Small": dict(colors=1, ranks=5, hand_size=lambda p: 2, max_info=3, max_life=1),
}


def make_config(game="Hanabi-Full", players=2, flags=0):
    g = GAME_TYPES[game]
    return HbConfig(players, g["colors"], g["ranks"], g["hand_size"](players), g["max_info"], g["max_life"], flags)


class HbAdamTensor(C.Structure):
    """`hb_adam_tensor` of include/hanabi_hip.h."""

    _fields_ = [(n, C.c_void_p) for n in ("w", "w_mu", "w_sigma", "noise", "grad", "m_w", "v_w", "m_mu", "v_mu", "m_sigma",
                                          "v_sigma", "eff")] + [("n", C.c_int64), ("cols", C.c_int32), ("eff_ld", C.c_int32),
                                                                                   ("grad_dtype", C.c_int32), ("grad_ld", C.c_int32)]


class HbAdamPack(C.Structure):
    """`hb_adam_pack` of include/hanabi_hip.h."""

    _fields_ = [("wt", C.c_void_p), ("frag", C.c_void_p), ("col_map_dev", C.c_void_p), ("bias_f32", C.c_void_p),
                ("wt_ld", C.c_int32), ("frag_kind", C.c_int32)]


class HbPackJob(C.Structure):
    """`hb_pack_job` of include/hanabi_hip.h."""

    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("wt", C.c_void_p), ("bias_out", C.c_void_p)] + [
        (n, C.c_int32) for n in ("k_rows", "n_cols", "w_ld", "group_cols", "k_pad")]


class HbCmd(C.Structure):
    """`hb_cmd` of include/hanabi_hip.h (hb_chain_run)."""

    _fields_ = [("op", C.c_int32), ("var", C.c_int32), ("fvar", C.c_int32), ("cond", C.c_int32), ("stream", C.c_void_p),
                ("p", C.c_void_p * 16), ("i", C.c_int64 * 10), ("f", C.c_double * 2)]


(CMD_WAIT_EVENT, CMD_RECORD_EVENT, CMD_REPLAY_INSERT, CMD_ACTOR_FUSED_ACT, CMD_ENV_STEP_PACKED, CMD_TREE_FILL_RANGE,
 CMD_PER_SAMPLE_GATHER, CMD_GRAPH_LAUNCH, CMD_PER_UPDATE, CMD_ACTOR_FUSED_PACK, CMD_ACTOR_PACK_WEIGHTS) = range(1, 12)


class HbRule(C.Structure):
    """`hb_rule` of include/hanabi_hip.h."""

    _fields_ = [("kind", C.c_int32), ("arg", C.c_int32), ("threshold", C.c_float)]


(RULE_LEGAL_RANDOM, RULE_DISCARD_OLDEST_FIRST, RULE_OSAWA_DISCARD, RULE_TELL_UNKNOWN, RULE_TELL_RANDOMLY,
 RULE_PLAY_SAFE_CARD, RULE_PLAY_IF_CERTAIN, RULE_TELL_PLAYABLE_CARD_OUTER, RULE_TELL_DISPENSABLE, RULE_DISCARD_RANDOMLY,
 RULE_PLAY_PROBABLY_SAFE, RULE_DISCARD_PROBABLY_USELESS, RULE_HAIL_MARY, RULE_TELL_ANYONE_USELESS_CARD,
 RULE_TELL_PLAYABLE_CARD, RULE_TELL_MOST_INFORMATION) = range(16)
MAX_RULES = 16


def library_path():
    # HANABI_HIP_LIB lets scripts/env_stamps.py load the diagnostic (-DHB_STAMPS) build of the same ABI
    return os.environ.get("HANABI_HIP_LIB") or os.path.join(_HERE, "libhanabi_hip.so")


# name -> (restype, argtypes); one entry per symbol declared in include/hanabi_hip.h
_P, _I32, _I64, _U64, _F64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
_CFG = C.POINTER(HbConfig)
SIGNATURES = {
    "hb_last_error": (C.c_char_p, []),
    "hb_abi_version": (C.c_int, []),
    "hb_config_validate": (C.c_int, [_CFG]),
    "hb_num_actions": (C.c_int, [_CFG]),
    "hb_obs_len": (C.c_int, [_CFG]),
    "hb_deck_size": (C.c_int, [_CFG]),
    "hb_state_words": (C.c_int, [_CFG]),
    "hb_obs_words": (C.c_int, [_CFG]),
    "hb_env_create": (C.c_int, [_CFG, _I64, _U64, _I64, C.POINTER(_P)]),
    "hb_env_destroy": (C.c_int, [_P]),
    "hb_env_num_games": (_I64, [_P]),
    "hb_env_set_decks": (C.c_int, [_P, _P]),
    "hb_env_reset": (C.c_int, [_P, _P, _I32, _P]),
    "hb_env_observe": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "hb_env_step": (C.c_int, [_P] + [_P] * 8 + [_P]),
    "hb_env_observe_packed": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "hb_env_step_packed": (C.c_int, [_P] + [_P] * 9 + [_P]),
    "hb_env_step_select_packed": (C.c_int, [_P, _P, _P, C.c_float, _U64, _U64, _I64, _P] + [_P] * 8 + [_P]),
    "hb_obs_pack": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "hb_obs_unpack": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "hb_env_illegal_count": (C.c_int, [_P, C.POINTER(_I64)]),
    "hb_env_stats": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "hb_env_export_state": (C.c_int, [_P, _P, _P]),
    "hb_env_import_state": (C.c_int, [_P, _P, _P]),
    "hb_env_state": (_P, [_P]),
    "hb_rule_act": (C.c_int, [_CFG, _P, _I64, _I64, C.POINTER(HbRule), _I32, _U64, _U64, _P, _P, _P]),
    "hb_random_legal_actions": (C.c_int, [_P, _I64, _I32, _U64, _U64, _I64, _P, _P]),
    "hb_env_set_games_per_wave": (C.c_int, [_P, _I32]),
    "hb_env_set_async_refill": (C.c_int, [_P, _I32]),
    "hb_env_set_refill_period": (C.c_int, [_P, _I32]),
    "hb_env_set_profile_events": (C.c_int, [_P, _P, _P]),
    "hb_event_create": (C.c_int, [C.POINTER(_P)]),
    "hb_event_destroy": (C.c_int, [_P]),
    "hb_event_record": (C.c_int, [_P, _P]),
    "hb_stream_wait_event": (C.c_int, [_P, _P]),
    "hb_tree_create": (C.c_int, [_I64, C.POINTER(_P)]),
    "hb_tree_destroy": (C.c_int, [_P]),
    "hb_tree_capacity": (_I64, [_P]),
    "hb_tree_nodes": (_P, [_P]),
    "hb_tree_export_nodes": (C.c_int, [_P, _P, _P]),
    "hb_tree_import_nodes": (C.c_int, [_P, _P, _P]),
    "hb_tree_set_lazy_top": (C.c_int, [_P, _I32]),
    "hb_tree_update": (C.c_int, [_P, _P, _P, _I64, _P]),
    "hb_tree_fill_range": (C.c_int, [_P, _I64, _I64, _P, _P]),
    "hb_tree_sample": (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    "hb_tree_get": (C.c_int, [_P, _P, _P, _I64, _P]),
    "hb_tree_total": (C.c_int, [_P, _P, _P]),
    "hb_tree_error_count": (C.c_int, [_P, C.POINTER(_I64)]),
    "hb_per_sample": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P]),
    "hb_per_sample_philox": (C.c_int, [_P, _U64, _P, _I64, _P, _P, _P]),
    "hb_per_sample_gather": (C.c_int, [_P, _U64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _I32, _I32, _P, _P, _P, _P, _I32,
                                       C.c_float, _I64, _I64, _P, _P]),
    "hb_per_update": (C.c_int, [_P, _P, _P, _I64, _F64, _P, _P, _P]),
    "hb_obs_cast": (C.c_int, [_P, _P, _I32, _I64, _I32, _I32, _P]),
    "hb_policy_act": (C.c_int, [_P, _I32, _P, _P, _I64, _I32, _I32, _I32, C.c_float, _U64, _U64, _I64, _P, _P, _P]),
    "hb_replay_gather": (C.c_int, [_P] * 6 + [_I64, _I32, _P, _I32, _I32, _P, _P, _P, _P, _I32, C.c_float, _I64, _I64, _P, _P]),
    "hb_replay_gather_packed": (C.c_int, [_P] * 6 + [_I64, _I32, _P, _I32, _I32, _P, _P, _P, _P, _I32, C.c_float, _I64, _I64, _P, _P]),
    "hb_c51_loss_grad": (C.c_int, [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "hb_c51_loss_sparse": (C.c_int, [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "hb_thin_gemm": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I32, _P]),
    "hb_dqn_loss_sparse": (C.c_int, [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "hb_c51_backward": (C.c_int, [_P, _P, _P, _I32, _P, _I32, _I32, _I64, _I32, _I32, _I32, _P, _P, _P, _I32, _P, _P]),
    "hb_colsum": (C.c_int, [_P, _I32, _I64, _I64, _P, _P]),
    "hb_noisy_adam": (C.c_int, [_P] * 11 + [_P, _P, _I32, _I64, _I32, _I32, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "hb_noisy_adam_multi": (C.c_int, [C.POINTER(HbAdamTensor), _I32, _P, C.c_float, _I32, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "hb_actor_pack_weights": (C.c_int, [C.POINTER(HbPackJob), _I32, _P]),
    "hb_actor_hidden": (C.c_int, [_P, _I64, _I32, _P, _I32, _P, _I32, _P, _P]),
    "hb_actor_hidden_packed": (C.c_int, [_P, _I64, _I32, _P, _I32, _P, _I32, _P, _P]),
    "hb_actor_q": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _I32, _I32, _P, _P]),
    "hb_policy_select": (C.c_int, [_P, _P, _I64, _I32, C.c_float, _U64, _U64, _I64, _P, _P]),
    "hb_actor_q_select": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _I32, _I32, _P, _P, C.c_float, _U64, _U64, _I64, _P, _P, _P]),
    "hb_actor_act": (C.c_int, [_P, _I32, _P, _I64, _I32, _P, _I32, _P, _I32, _P, _P, _P, _P, _I32, _I32, _P, C.c_float, _U64, _U64, _I64,
                               _P, _P, _P]),
    "hb_actor_fused_supported": (C.c_int, [_I32, _I32, _I32, _I32]),
    "hb_actor_fused_sizes": (C.c_int, [_I32, _I32, _I32, _I32, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I32)]),
    "hb_actor_fused_pack": (C.c_int, [_P, _I32, _P, _P, _I32, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "hb_actor_fused_q": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "hb_actor_fused_act": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, C.c_float, _U64, _U64, _I64, _P, _P]),
    "hb_actor_fused_columns": (C.c_int, [_I32, C.POINTER(_I32)]),
    "hb_noisy_adam_multi_pack": (C.c_int, [_P, _P, _I32, _P, C.c_float, _I32, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "hb_actor_fused_pack_dt": (C.c_int, [_P, _I32, _P, _P, _I32, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _I32, _P]),
    "hb_actor_fused_pack_thin": (C.c_int, [_P, _I32, _P, _P, _I32, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P, _I32, _I32, _P]),
    "hb_actor_fused_q_dt": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _I32, _P]),
    "hb_actor_fused_act_dt": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, C.c_float, _U64, _U64, _I64, _P, _I32, _P]),
    "hb_chain_run": (C.c_int, [C.POINTER(HbCmd), _I32, C.POINTER(_I64), C.POINTER(_F64)]),
    "hb_relu_bwd_colsum": (C.c_int, [_P, _P, _I64, _I32, _I64, _I64, _P, _P]),
    "hb_replay_insert": (C.c_int, [_P] * 12 + [_I64, _I32, _I32, _I64, _I64, _P]),
}


def lib():
    """Load libhanabi_hip.so (built by __graft_entry__.build() / csrc/Makefile). Raises if it is missing."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise HbError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(there is no CPU fallback)")
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise HbError(f"hanabi_hip error {rc}: {lib().hb_last_error().decode()}")


_RAW_STREAM = None


def current_stream():
    """hipStream_t of torch's current stream on the current device. torch.cuda.current_stream() costs ~8 us of Python per
    call (device-index plumbing) and is needed for every launch: the raw accessor is ~20x cheaper."""
    global _RAW_STREAM
    import torch

    if _RAW_STREAM is None:
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        _RAW_STREAM = (lambda: raw(torch._C._cuda_getDevice())) if raw is not None else (
            lambda: torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(_RAW_STREAM())


def dptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


class Event:
    """hipEvent_t (timing off) through the C-ABI: record(stream) / wait(stream) take raw hipStream_t values (c_void_p or
    int; None = torch's current stream). ~1 us per call where torch.cuda.Event's methods cost ~8 us."""

    __slots__ = ("h", "_lib", "recorded")

    def __init__(self):
        self._lib = lib()
        h = C.c_void_p()
        check(self._lib.hb_event_create(C.byref(h)))
        self.h = h
        self.recorded = False

    def record(self, stream=None):
        check(self._lib.hb_event_record(self.h, current_stream() if stream is None else stream))
        self.recorded = True

    def wait(self, stream=None):
        """Make `stream` wait for the most recent record() (no-op if never recorded)."""
        if self.recorded:
            check(self._lib.hb_stream_wait_event(current_stream() if stream is None else stream, self.h))

    def __del__(self):
        try:
            if self.h:
                self._lib.hb_event_destroy(self.h)
        except Exception:
            pass


def set_stream(stream):
    """torch.cuda.set_stream without the device-index plumbing of the context manager (~1 us instead of ~20 us for
    `with torch.cuda.stream(s)`). The caller restores the previous stream itself."""
    import torch

    torch._C._cuda_setStream(stream_id=stream.stream_id, device_index=stream.device_index, device_type=stream.device_type)
