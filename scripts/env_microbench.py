"""Env-kernel microbenchmark: time per step at several games-per-wave settings (HIP events)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
import hanabi_hip  # noqa: E402


def run(game, players, n, gpw, steps=200, warm=70, packed=False):
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config(game, players, flags), n_games=n, seed=1234,
                               games_per_wave=gpw, packed=packed)
    act = torch.empty(n, dtype=torch.int32, device="cuda")
    for t in range(warm):
        env.random_legal_actions(4321, t, out=act)
        env.step(act)
    torch.cuda.synchronize()
    # pre-generate actions? they depend on state; time policy+step and step alone separately
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step_ms = 0.0
    evs = []
    for t in range(steps):
        env.random_legal_actions(4321, warm + t, out=act)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        env.step(act)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    med = ts[len(ts) // 2]
    bytes_per = (4 * env.obs_words if packed else env.obs_len) + env.num_actions + 9 + 2 * env.state_words * 4
    print(f"{game} P={players} N={n} G={gpw} {'packed' if packed else 'int8'}: step median {med*1e3:.1f} us  min {ts[0]*1e3:.1f} us  "
          f"-> {n/med/1e3:.1f} M env-steps/s, {n*bytes_per/med/1e6:.0f} GB/s algorithmic")


if __name__ == "__main__":
    for packed in (True, False):
        for n in (32768, 262144):
            for g in (8, 16, 32, 64):
                run("Hanabi-Full", 2, n, g, packed=packed)
        for g in (16, 32, 64):
            run("Hanabi-Full", 5, 32768, g, packed=packed)
