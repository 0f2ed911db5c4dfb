"""GPU parity: the fused HIP env/encoder kernel against the CPU oracle, bit for bit, through
the C-ABI (hanabi_hip.HanabiEnv -> libhanabi_hip.so)."""
import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _pair(game, players, n, seed=7, flags=O.FLAG_AUTO_RESET | O.FLAG_RESET_START_NEXT, decks=None, gpw=None,
          first_game_id=0, packed=False):
    import hanabi_hip

    cfg = hanabi_hip.make_config(game, players, flags)
    env = hanabi_hip.HanabiEnv(config=cfg, n_games=n, seed=seed, first_game_id=first_game_id, decks=decks,
                               games_per_wave=gpw, packed=packed)
    orc = O.OracleEnv(O.make_config(game, players, flags), n, seed=seed, first_game_id=first_game_id, decks=decks)
    return env, orc


def _assert_same(env, orc, out, what):
    import torch

    torch.cuda.synchronize()
    if env.packed:
        # the kernel wrote ONLY the bit rows (hb_env_step_packed): they must be the oracle's observation packed 32 per word,
        # pad bits zero; env.obs below is then hb_obs_unpack of them
        want = np.zeros((env.n, env.obs_words * 32), np.uint8)
        want[:, :env.obs_len] = out["obs"]
        want = np.packbits(want, axis=1, bitorder="little").view(np.uint32)
        got = env.obs_bits.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), f"{what}: packed observation differs in {np.argwhere(got != want)[:5]}"
    for name, t in (("obs", env.obs), ("legal", env.legal), ("agent_reward", env.agent_reward),
                    ("agent_step_type", env.agent_step_type)):
        got = t.cpu().numpy()
        assert np.array_equal(got, out[name]), f"{what}: {name} differs in {np.argwhere(got != out[name])[:5]}"
    if "reward" in out:
        for name, t in (("reward", env.reward), ("terminal", env.terminal), ("score", env.score)):
            assert np.array_equal(t.cpu().numpy(), out[name]), f"{what}: {name} differs"
    st = env.export_state().cpu().numpy().view(np.uint32)
    assert np.array_equal(st, orc.export_state()), f"{what}: state rows differ"


@pytest.mark.parametrize("game,players,n,gpw", [
    ("Hanabi-Full", 2, 1000, 64), ("Hanabi-Full", 2, 777, 16), ("Hanabi-Full", 3, 300, 32),
    ("Hanabi-Full", 4, 257, 64), ("Hanabi-Full", 5, 512, 64), ("Hanabi-Full", 5, 100, 16),
    ("Hanabi-Small", 2, 500, 64), ("Hanabi-Small", 5, 130, 32), ("Hanabi-Very-Small", 2, 64, 64),
    ("Hanabi-Very-Small", 5, 65, 16), ("Hanabi-Full", 2, 1, 64), ("Hanabi-Small", 3, 17, 64),
])
@pytest.mark.parametrize("packed", [False, True])
def test_random_self_play_bit_exact(game, players, n, gpw, packed):
    """Random-legal self-play with auto-reset: obs (int8 form, and the bit-packed form when the env emits that), legal,
    rewards, step types and the raw state rows stay identical to the oracle for hundreds of moves (several episodes per game)."""
    env, orc = _pair(game, players, n, gpw=gpw, first_game_id=12345, packed=packed)
    _assert_same(env, orc, orc.observe(), "after reset")
    steps = 150 if game == "Hanabi-Full" else 80
    for t in range(steps):
        act = env.random_legal_actions(seed=4321, draw=t)
        a = act.cpu().numpy()
        assert np.array_equal(a, O.random_legal_actions(orc.observe()["legal"], 4321, t, first_game_id=12345))
        env.step(act)
        _assert_same(env, orc, orc.step(a), f"step {t}")
    assert env.illegal_count() == 0 == orc.illegal_count()


def test_without_auto_reset_games_stay_terminal():
    env, orc = _pair("Hanabi-Small", 2, 200, flags=0)
    for t in range(60):
        act = env.random_legal_actions(seed=5, draw=t)
        env.step(act)
        _assert_same(env, orc, orc.step(act.cpu().numpy()), f"step {t}")
    status = (env.export_state().cpu().numpy().view(np.uint32)[:, 0] >> 19) & 3
    assert (status != 0).all()
    # explicit masked reset of half of them
    mask = (np.arange(200) % 2).astype(np.uint8)
    env.reset(mask=mask, start_player=1)
    orc.reset(mask=mask, start_player=1)
    _assert_same(env, orc, orc.observe(), "masked reset")


def test_illegal_and_out_of_range_actions():
    import torch

    env, orc = _pair("Hanabi-Full", 2, 128)
    bad = np.zeros(128, np.int32)          # discard at 8 info tokens: illegal
    bad[::3] = 99
    bad[1::3] = -5
    env.step(torch.as_tensor(bad).cuda())
    _assert_same(env, orc, orc.step(bad), "illegal step")
    assert env.illegal_count() == 128 == orc.illegal_count()


def test_explicit_decks_scripted_game():
    """Same hand-worked scenario as tests/test_oracle_env.py::test_scripted_full_game_prefix."""
    cfg = O.make_config()
    deck = []
    for c in range(5):
        for r in range(5):
            deck += [c * 5 + r] * (3 if r == 0 else (1 if r == 4 else 2))
    decks = np.tile(np.array(deck, np.uint8), (4, 1))
    env, orc = _pair("Hanabi-Full", 2, 4, flags=0, decks=decks)
    for uid in (5, 10, 8, 9, 5, 8):
        act = np.full(4, uid, np.int32)
        env.step(act)
        _assert_same(env, orc, orc.step(act), f"uid {uid}")
    assert env.terminal.cpu().numpy().tolist() == [1] * 4 and env.reward.cpu().numpy().tolist() == [-2.0] * 4


def test_full_size_properties_32768_games():
    """BASELINE config (2p full, 32 768 games): size-independent properties on the GPU output
    alone, plus a sampled comparison against the oracle on a slice of the games."""
    import hanabi_hip
    import torch

    n = 32768
    flags = O.FLAG_AUTO_RESET | O.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=n, seed=1234)
    sl = slice(20000, 20256)
    orc = O.OracleEnv(O.make_config("Hanabi-Full", 2, flags), 256, seed=1234, first_game_id=20000)
    copies = np.array([3, 2, 2, 2, 1] * 5)
    total_terminals = 0
    for t in range(200):
        act = env.random_legal_actions(seed=4321, draw=t)
        env.step(act)
        out = orc.step(act[sl].cpu().numpy())
        if t % 20 == 0 or t == 199:
            obs = env.obs.cpu().numpy()
            assert set(np.unique(obs)) <= {0, 1}
            assert np.array_equal(obs[sl], out["obs"]) and np.array_equal(env.legal[sl].cpu().numpy(), out["legal"])
            # one-hot / thermometer structure
            hands = obs[:, 0:125].reshape(n, 5, 25)
            assert (hands.sum(2) <= 1).all()
            deck = obs[:, 127:167]
            assert (np.diff(deck.astype(np.int8), axis=1) <= 0).all()          # thermometer is monotone
            fw = obs[:, 167:192].reshape(n, 5, 5)
            assert (fw.sum(2) <= 1).all()
            disc = obs[:, 203:253]
            assert (disc.sum(1) + (fw.argmax(2) + fw.max(2)).sum(1) <= 50).all()
            st = env.export_state().cpu().numpy().view(np.uint32)
            info, life = (st[:, 0] >> 6) & 15, (st[:, 0] >> 10) & 7
            assert (obs[:, 192:200].sum(1) == info).all() and (obs[:, 200:203].sum(1) == life).all()
            assert (((st[:, 0] >> 13) & 7) == (t + 1) % 2).all()               # lock-step seat
            # card conservation from the raw state rows
            dsz = st[:, 0] & 63
            cnt = np.zeros((n, 25), np.int64)
            d64 = st[:, 8].astype(np.uint64) | (st[:, 9].astype(np.uint64) << np.uint64(32))
            pos = 0                                                             # words 8-9: thermometer per card identity
            for i in range(25):
                for k in range(int(copies[i])):
                    cnt[:, i] += ((d64 >> np.uint64(pos + k)) & np.uint64(1)).astype(np.int64)
                pos += int(copies[i])
            assert np.array_equal(np.unpackbits(d64.view(np.uint8).reshape(n, 8), axis=1, bitorder="little")[:, :50],
                                  obs[:, 203:253].astype(np.uint8))             # and it IS the observation's discard section
            for c in range(5):
                f = (st[:, 1] >> (3 * c)) & 7
                for r in range(5):
                    cnt[:, c * 5 + r] += (f > r)
            for p in range(2):
                hn = (st[:, 1] >> (15 + 3 * p)) & 7
                for i in range(5):
                    card = (st[:, 10 + p] >> (5 * i)) & 31
                    ok = i < hn
                    np.add.at(cnt, (np.flatnonzero(ok), card[ok]), 1)
            deckb = st[:, 16:29].copy().view(np.uint8).reshape(n, 52)[:, :50]
            for pos in range(50):
                ok = pos >= 50 - dsz
                np.add.at(cnt, (np.flatnonzero(ok), deckb[ok, pos]), 1)
            assert (cnt == copies[None]).all()
        total_terminals += int(env.terminal.sum().item())
    assert total_terminals > n  # every game finished at least once on average
    assert env.illegal_count() == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("packed", [False, True])
def test_full_size_properties_5_players_32768_games(packed):
    """BASELINE config 4's per-GPU leg (5-player full Hanabi, 32 768 games, obs 1280, 48 moves, 192-byte state rows):
    size-independent structure of the GPU output plus an oracle comparison on a 256-game slice, int8 and bit-packed output."""
    import hanabi_hip
    import torch

    n, P = 32768, 5
    flags = O.FLAG_AUTO_RESET | O.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", P, flags), n_games=n, seed=77, packed=packed)
    assert (env.obs_len, env.num_actions, env.state_words, env.obs_words) == (1280, 48, 48, 40)
    sl = slice(30000, 30256)
    orc = O.OracleEnv(O.make_config("Hanabi-Full", P, flags), 256, seed=77, first_game_id=30000)
    terminals = 0
    for t in range(160):
        act = env.random_legal_actions(seed=99, draw=t)
        env.step(act)
        out = orc.step(act[sl].cpu().numpy())
        terminals += int(env.terminal.sum().item())
        if t % 20 == 0 or t == 159:
            obs = env.obs.cpu().numpy()
            assert set(np.unique(obs)) <= {0, 1}
            assert np.array_equal(obs[sl], out["obs"]) and np.array_equal(env.legal[sl].cpu().numpy(), out["legal"])
            assert np.array_equal(env.agent_reward[sl].cpu().numpy(), out["agent_reward"])
            assert np.array_equal(env.agent_step_type[sl].cpu().numpy(), out["agent_step_type"])
            st = env.export_state().cpu().numpy().view(np.uint32)
            assert np.array_equal(st[sl], orc.export_state())
            # sections (App. A.6, 5 players): hands 400 | flags 5 | deck 30 | fireworks 25 | info 8 | life 3 | discards 50 | last 59 | knowledge 700
            hands = obs[:, 0:400].reshape(n, 16, 25)
            assert (hands.sum(2) <= 1).all()
            deck = obs[:, 405:435]
            assert (np.diff(deck.astype(np.int8), axis=1) <= 0).all() and (deck.sum(1) == (st[:, 0] & 63)).all()
            fw = obs[:, 435:460].reshape(n, 5, 5)
            assert (fw.sum(2) <= 1).all()
            assert (obs[:, 460:468].sum(1) == ((st[:, 0] >> 6) & 15)).all() and (obs[:, 468:471].sum(1) == ((st[:, 0] >> 10) & 7)).all()
            assert (((st[:, 0] >> 13) & 7) == (t + 1) % P).all()                       # lock-step seat
            last = obs[:, 521:580]
            assert (last[:, 0:5].sum(1) <= 1).all() and (last[:, 5:9].sum(1) == last[:, 0:5].sum(1)).all()   # actor and type together
            kn = obs[:, 580:1280].reshape(n, 20, 35)
            assert (kn[:, :, 25:30].sum(2) <= 1).all() and (kn[:, :, 30:35].sum(2) <= 1).all()
            # a hand that is short shows its flag, its empty card slot and an empty knowledge slot
            hn = np.stack([(st[:, 1] >> (15 + 3 * p)) & 7 for p in range(P)], axis=1)
            cur = (st[:, 0] >> 13) & 7
            for rel in range(P):
                short = hn[np.arange(n), (cur + rel) % P] < 4
                assert np.array_equal(obs[:, 400 + rel] == 1, short)
            assert (legal_rows := env.legal.cpu().numpy()).sum(1).min() >= 1 and legal_rows.shape == (n, 48)
            if packed:
                from hanabi_agents.rlax_dqn import bitpack

                assert torch.equal(bitpack.unpack(env.obs_bits, 1280), env.obs)
    assert terminals > n // 2 and env.illegal_count() == 0


def test_unaligned_output_is_rejected():
    import ctypes as C

    import hanabi_hip
    import torch
    from hanabi_hip import _capi as K

    env = hanabi_hip.HanabiEnv(n_games=8)
    buf = torch.zeros(8 * 658 + 64, dtype=torch.int8, device="cuda")
    rc = K.lib().hb_env_observe(env.h, C.c_void_p(buf.data_ptr() + 1), K.dptr(env.legal), None, None, K.current_stream())
    assert rc == -4 and b"aligned" in K.lib().hb_last_error()


def test_episode_statistics_counted_in_kernel():
    env, orc = _pair("Hanabi-Small", 2, 300)
    episodes = score = 0
    for t in range(60):
        act = env.random_legal_actions(seed=9, draw=t)
        env.step(act)
        out = orc.step(act.cpu().numpy())
        episodes += int(out["terminal"].sum())
        score += int((out["score"] * out["terminal"]).sum())
    assert episodes > 100 and env.stats() == (episodes, score)


@pytest.mark.parametrize("game", ["Hanabi-Full", "Hanabi-Small", "Hanabi-Very-Small"])
@pytest.mark.parametrize("players", [2, 3, 4, 5])
def test_arbitrary_uids_all_variants(game, players):
    """Every compiled variant under move uids drawn uniformly from [-2, A+2): about half of them illegal (discard at
    max tokens, hints nobody matches, empty slots late in a game, out of range). Illegal moves leave the state alone
    and are counted; everything stays bit-identical to the oracle, with and without the lenient-reward flag."""
    import torch

    for extra in (0, O.FLAG_LENIENT_REWARD):
        env, orc = _pair(game, players, 96, seed=31 + players, flags=O.FLAG_AUTO_RESET | extra, first_game_id=7,
                         packed=bool(extra))
        rng = np.random.default_rng(players * 10 + len(game))
        for t in range(70):
            act = rng.integers(-2, env.num_actions + 2, 96).astype(np.int32)
            if t % 3 == 0:                                     # keep the games moving: a third of the steps are legal
                act = env.random_legal_actions(seed=9, draw=t).cpu().numpy()
            env.step(torch.as_tensor(act).cuda())
            _assert_same(env, orc, orc.step(act), f"{game}/{players}p flags {extra} step {t}")
        assert env.illegal_count() == orc.illegal_count() > 0


@pytest.mark.parametrize("game,players,n", [("Hanabi-Full", 2, 3000), ("Hanabi-Full", 5, 700), ("Hanabi-Small", 3, 500),
                                            ("Hanabi-Very-Small", 2, 130), ("Hanabi-Full", 2, 32768)])
def test_selection_fused_into_the_env_step_equals_select_then_step(game, players, n):
    """hb_env_step_select_packed (each game's lane picks its move from q and the legal mask by hb_policy_select's rule and
    draws, then applies it) against hb_policy_select followed by hb_env_step_packed: same actions, and the same state rows,
    observations, legal masks, rewards and step types after every one of 60 steps (greedy and exploring, q with ties)."""
    import torch

    import hanabi_hip
    from hanabi_hip import _capi as K

    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    mk = lambda: hanabi_hip.HanabiEnv(config=hanabi_hip.make_config(game, players, flags), n_games=n, seed=21, first_game_id=777,
                                      packed=True)
    a, b = mk(), mk()
    A = a.num_actions
    g = torch.Generator(device="cuda").manual_seed(3)
    L = K.lib()
    for t in range(60):
        q = torch.randn(n, A, device="cuda", generator=g)
        if t % 3 == 0:
            q = torch.round(q * 2) / 2            # many exact ties between the best moves
        eps = (0.0, 0.3, 1.0)[t % 3]
        act_a = torch.empty(n, dtype=torch.int32, device="cuda")
        K.check(L.hb_policy_select(K.dptr(q), K.dptr(a.legal), n, A, eps, 99, 1000 + t, 777, K.dptr(act_a), K.current_stream()))
        a.step(act_a)
        act_b = b.step_select(q, eps, 99, 1000 + t, 777)[0]
        torch.cuda.synchronize()
        assert torch.equal(act_a, act_b), f"step {t}: {(act_a != act_b).sum().item()} actions differ"
        assert torch.equal(a.export_state(), b.export_state()), f"step {t}: state rows differ"
        for x, y in ((a.obs_bits, b.obs_bits), (a.legal, b.legal), (a.reward, b.reward), (a.agent_reward, b.agent_reward),
                     (a.agent_step_type, b.agent_step_type), (a.terminal, b.terminal)):
            assert torch.equal(x, y)
    assert a.illegal_count() == 0 and b.illegal_count() == 0 and a.stats() == b.stats()
