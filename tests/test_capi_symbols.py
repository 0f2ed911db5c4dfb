"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every
symbol include/hanabi_hip.h declares; pure-host size functions agree with the oracle; the
product refuses to run without a GPU instead of falling back to anything on the CPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "hanabi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import hanabi_hip

    L = hanabi_hip.lib()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/hanabi_hip.h but not exported"
    from hanabi_hip import _capi

    assert sorted(_capi.SIGNATURES) == names, "ctypes table and header disagree"
    assert L.hb_abi_version() == 1


def test_host_size_functions_match_oracle():
    import hanabi_hip
    from oracle import oracle_py as O

    L, OL = hanabi_hip.lib(), O.lib()
    for game in ("Hanabi-Full", "Hanabi-Small", "Hanabi-Very-Small"):
        for players in (2, 3, 4, 5):
            cfg = hanabi_hip.make_config(game, players)
            ocfg = O.make_config(game, players)
            assert L.hb_config_validate(C.byref(cfg)) == 0
            assert L.hb_obs_len(C.byref(cfg)) == OL.orc_obs_len(C.byref(ocfg))
            assert L.hb_num_actions(C.byref(cfg)) == OL.orc_num_actions(C.byref(ocfg))
            assert L.hb_deck_size(C.byref(cfg)) == OL.orc_deck_size(C.byref(ocfg))
            assert L.hb_state_words(C.byref(cfg)) == OL.orc_state_words(C.byref(ocfg))
    cfg = hanabi_hip.make_config()
    assert (L.hb_obs_len(C.byref(cfg)), L.hb_num_actions(C.byref(cfg))) == (658, 20)
    bad = hanabi_hip.HbConfig(6, 5, 5, 5, 8, 3, 0)
    assert L.hb_config_validate(C.byref(bad)) != 0 and b"players" in L.hb_last_error()


def test_no_cpu_fallback():
    import torch

    import hanabi_hip

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hanabi_hip.HbError):
        hanabi_hip.HanabiEnv(n_games=2)
    with pytest.raises(hanabi_hip.HbError):
        hanabi_hip.SumTree(8)
    # and the C-ABI itself reports "no device" rather than computing anything
    h = C.c_void_p()
    cfg = hanabi_hip.make_config()
    assert hanabi_hip.lib().hb_env_create(C.byref(cfg), 4, 1, 0, C.byref(h)) == -2
    assert hanabi_hip.lib().hb_tree_create(8, C.byref(h)) == -2


def test_product_never_touches_the_oracle():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may use oracle/."""
    pkg = os.path.join(ROOT, "hanabi-agents_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("oracle_py", "liboracle", "from oracle", "import oracle", "orc_"):
                    assert needle not in text, f"{f} references the oracle ({needle})"
