"""Bit-packed observation rows.

The canonical observation is a vector of 0/1 (the reference keeps it as int8, one byte
per bit: hanabi_agents/rlax_dqn/experience_buffer.py:9,11). The env kernel builds it as
`words = ceil(obs_len / 32)` u32 of bits and can hand that form out directly (`hb_env_step_packed`);
with `RlaxRainbowParams(packed_obs=True)` the agent keeps `last_obs` and both observation rings in it
(84 instead of 658 bytes per observation for 2-player full Hanabi) and the actor / learner kernels
unpack while they stage their GEMM operands. Layout: observation element i = bit (i & 31) of word
i >> 5 of its row; pad bits of the last word are zero; rows are int32 tensors [n, words].

`pack` / `unpack` convert between the two forms: HIP kernels (`hb_obs_pack` / `hb_obs_unpack`) for CUDA
tensors, plain torch integer ops elsewhere (CPU tests; same bits).
"""
import torch


def words_for(obs_len: int) -> int:
    return (int(obs_len) + 31) // 32


def is_packed(obs: torch.Tensor, obs_len: int) -> bool:
    return obs.dtype == torch.int32 and obs.dim() == 2 and obs.shape[1] == words_for(obs_len)


def pack(obs: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """[n, obs_len] (any nonzero entry = 1) -> [n, words] int32."""
    n, obs_len = obs.shape
    w = words_for(obs_len)
    if out is None:
        out = torch.empty((n, w), dtype=torch.int32, device=obs.device)
    if obs.is_cuda:
        from hanabi_hip import _capi as K

        src = obs if obs.dtype == torch.int8 else (obs != 0).to(torch.int8)
        K.check(K.lib().hb_obs_pack(K.dptr(src.contiguous()), K.dptr(out), n, obs_len, K.current_stream()))
        return out
    b = torch.zeros((n, w * 32), dtype=torch.int64)
    b[:, :obs_len] = (obs != 0).to(torch.int64)
    v = (b.view(n, w, 32) << torch.arange(32, dtype=torch.int64)).sum(dim=-1)          # 0 .. 2^32-1
    out.copy_(torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32))
    return out


def unpack(bits: torch.Tensor, obs_len: int, out: torch.Tensor = None) -> torch.Tensor:
    """[n, words] int32 -> [n, obs_len] int8 0/1."""
    n, w = bits.shape
    assert w == words_for(obs_len), (w, obs_len)
    if out is None:
        out = torch.empty((n, obs_len), dtype=torch.int8, device=bits.device)
    if bits.is_cuda:
        from hanabi_hip import _capi as K

        K.check(K.lib().hb_obs_unpack(K.dptr(bits.contiguous()), K.dptr(out), n, obs_len, K.current_stream()))
        return out
    v = bits.to(torch.int64) & 0xFFFFFFFF
    out.copy_(((v[:, :, None] >> torch.arange(32, dtype=torch.int64)) & 1).reshape(n, w * 32)[:, :obs_len].to(torch.int8))
    return out
