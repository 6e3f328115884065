"""BatchedUavEnv: the batched reset()/step(actions) -> (obs, reward, done) surface over
libuavtrack.so.  Tensors are torch-ROCm device tensors owned by Python; the kernels
write them in place on torch's current stream (no host sync, no copies)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .config import EnvConfig, RewardMode
from .pmi import fold_pmi_state_dict

_STATE_KEYS = ("ux", "uy", "uz", "uh", "ua", "tx", "ty", "tz", "th")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedUavEnv:
    """B independent copies of the reference `Environment` (src/environment.py:12) on one GPU.

    reset(seed)            -> obs [B, N, 12]                           (environment.py:87-118)
    step(actions [B, N])   -> (obs [B, N, 12], reward [B, N], done [B]) (environment.py:120-164)
                              self.info = {"terms": [3, B, N], "covered": [B]}
    step_many(actions [T, B, N]) -> dict with a leading T axis, one launch
    """

    def __init__(self, cfg: EnvConfig, device: str = "cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("uavtrack runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self._lib = _lib.load()
        idx = self.device.index if self.device.index is not None else 0
        self._device_index = idx
        cstruct = cfg.c_struct(idx)
        handle = C.c_void_p()
        _lib.check(self._lib.uavtrack_create(C.byref(cstruct), C.byref(handle)), "uavtrack_create")
        self._h = handle
        self.info: Dict[str, torch.Tensor] = {}
        self._episode = 0
        self._trace: Optional[torch.Tensor] = None    # the installed target-trace buffer (kept alive: the library holds its raw pointer)
        self._raw: Optional[torch.Tensor] = None      # ... and the raw-reward buffer (set_raw_output)
        self._host_step = _lib.HostStep()             # step_host: the library's host result block and numpy views of it
        self._host_views: Optional[Dict[str, np.ndarray]] = None

    # -- plumbing ---------------------------------------------------------------------
    @property
    def B(self) -> int: return self.cfg.n_envs
    @property
    def N(self) -> int: return self.cfg.n_uav
    @property
    def M(self) -> int: return self.cfg.m_targets

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _fits(self, t, shape, dtype) -> bool:
        """True if `t` can be handed to a kernel that writes a contiguous `shape` / `dtype` array on this device."""
        return (t is not None and tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.device == self.device
                and t.is_contiguous())

    def _reuse(self, o, key, shape, dtype, want=True):
        """Output buffer `key` of a previous result `o` if it has exactly the shape this launch writes, else a
        fresh one (a kernel writing T2 rows into a T1-row buffer would corrupt device memory)."""
        if not want:
            return None
        t = o.get(key) if o else None
        return t if self._fits(t, shape, dtype) else self._empty(shape, dtype)

    def _out_arg(self, t, shape, dtype, name):
        if t is None:
            return self._empty(shape, dtype)
        if not self._fits(t, shape, dtype):
            raise ValueError(f"{name} must be a contiguous {dtype} {tuple(shape)} tensor on {self.device}")
        return t

    def _actions(self, actions, shape) -> torch.Tensor:
        a = torch.as_tensor(actions)
        if a.device != self.device or a.dtype != torch.int32:
            a = a.to(device=self.device, dtype=torch.int32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"actions shape {tuple(a.shape)} != {tuple(shape)}")
        return a.contiguous()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.uavtrack_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kernel_info(self) -> Dict[str, int]:
        out = (C.c_int64 * 5)()
        _lib.check(self._lib.uavtrack_kernel_info(self._h, out), "uavtrack_kernel_info")
        return dict(workgroup=out[0], envs_per_workgroup=out[1], workgroups=out[2], lds_bytes=out[3],
                    specialised=out[4])

    # -- reference surface ------------------------------------------------------------
    def reset(self, seed: int = 0, episode: Optional[int] = None) -> torch.Tensor:
        if episode is None:
            episode = self._episode
        self._episode = episode + 1
        obs = self._empty((self.B, self.N, _lib.OBS_DIM), torch.float32)
        _lib.check(self._lib.uavtrack_reset(self._h, C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint32(episode),
                                            _ptr(obs), self._stream()), "uavtrack_reset")
        return obs

    def step(self, actions, want_terms: bool = True, ep_sums: Optional[torch.Tensor] = None,
             out_obs: Optional[torch.Tensor] = None, out_reward: Optional[torch.Tensor] = None):
        """`ep_sums` [B, 5] (float32, this device): running episode accumulators the kernel adds
        this step's contribution to (train.py:181-192).  `out_obs` / `out_reward`: caller-owned
        buffers the kernel writes instead of fresh tensors (static graph I/O)."""
        a = self._actions(actions, (self.B, self.N))

        def take(buf, shape):
            if buf is None:
                return self._empty(shape, torch.float32)
            if tuple(buf.shape) != shape or buf.dtype != torch.float32 or not buf.is_contiguous() or buf.device != self.device:
                raise ValueError(f"output buffer must be a contiguous float32 {shape} tensor on {self.device}")
            return buf
        obs = take(out_obs, (self.B, self.N, _lib.OBS_DIM))
        reward = take(out_reward, (self.B, self.N))
        terms = self._empty((3, self.B, self.N), torch.float32) if want_terms else None
        covered = self._empty((self.B,), torch.int32)
        done = self._empty((self.B,), torch.uint8)
        if ep_sums is not None:
            if ep_sums.shape != (self.B, 5) or ep_sums.dtype != torch.float32 or not ep_sums.is_contiguous():
                raise ValueError("ep_sums must be a contiguous float32 [B, 5] tensor")
            _lib.check(self._lib.uavtrack_step_accumulate(self._h, _ptr(a), _ptr(obs), _ptr(reward), _ptr(terms),
                                                          _ptr(covered), _ptr(done), _ptr(ep_sums), self._stream()),
                       "uavtrack_step_accumulate")
        else:
            _lib.check(self._lib.uavtrack_step(self._h, _ptr(a), _ptr(obs), _ptr(reward), _ptr(terms),
                                               _ptr(covered), _ptr(done), self._stream()), "uavtrack_step")
        self.info = {"terms": terms, "covered": covered}
        if self._raw is not None:
            self.info["raw"] = self._raw[0]       # uav.raw_reward of this step (set_raw_output)
        return obs, reward, done.bool()

    def set_target_trace(self, buf: Optional[torch.Tensor]) -> None:
        """Target positions after each step, [T, B, M, 2] (environment.py:150-153), written by every later stepping
        call until replaced; None switches the output off."""
        if buf is None:
            _lib.check(self._lib.uavtrack_set_target_trace(self._h, None, 0), "uavtrack_set_target_trace")
            self._trace = None
            return
        if buf.dim() != 4 or not self._fits(buf, (buf.shape[0], self.B, self.M, 2), torch.float32):
            raise ValueError(f"target trace must be a contiguous float32 [T, {self.B}, {self.M}, 2] tensor on {self.device}")
        _lib.check(self._lib.uavtrack_set_target_trace(self._h, _ptr(buf), C.c_int32(buf.shape[0])), "uavtrack_set_target_trace")
        self._trace = buf     # every later launch writes through the raw pointer: the tensor must outlive them

    def set_raw_output(self, buf: Optional[torch.Tensor]) -> None:
        """uav.raw_reward of every UAV after each step, [T, B, N] (environment.py:219: the weighted sum of the three
        normalised terms before any cooperative sharing), written by every later stepping call until replaced; None
        switches the output off."""
        if buf is None:
            _lib.check(self._lib.uavtrack_set_raw_reward_output(self._h, None, 0), "uavtrack_set_raw_reward_output")
            self._raw = None
            return
        if buf.dim() != 3 or not self._fits(buf, (buf.shape[0], self.B, self.N), torch.float32):
            raise ValueError(f"raw-reward buffer must be a contiguous float32 [T, {self.B}, {self.N}] tensor on {self.device}")
        _lib.check(self._lib.uavtrack_set_raw_reward_output(self._h, _ptr(buf), C.c_int32(buf.shape[0])),
                   "uavtrack_set_raw_reward_output")
        self._raw = buf       # (kept alive: the library holds its raw pointer)

    def _with_raw(self, T: int, want: bool, o, launch):
        """Run `launch()` with a [T, B, N] raw-reward buffer attached when asked; returns it (or None)."""
        if not want:
            return launch(), None
        rb = self._reuse(o, "raw", (T, self.B, self.N), torch.float32)
        installed = self._raw
        self.set_raw_output(rb)
        try:
            return launch(), rb
        finally:
            self.set_raw_output(installed)

    def step_host(self, actions: np.ndarray, stream=None) -> Dict[str, np.ndarray]:
        """Environment.step for a host caller (uavtrack_step_host): `actions` is a C-contiguous int32 numpy array [B, N];
        returns numpy VIEWS of the library's page-locked result block -- obs [B, N, 12], reward [B, N], terms [3, B, N],
        raw [B, N], covered [B], done [B] and the state behind the step (ux uy uh ua tx ty th [uz tz] step_count) -- valid
        until the next step_host on this environment.  Synchronises the stream.  `stream`: a ctypes stream handle the caller
        looked up once (a per-step `torch.cuda.current_stream()` lookup is a visible share of a 30 us step); default: torch's
        current stream."""
        if actions.dtype != np.int32 or not actions.flags.c_contiguous or actions.size != self.B * self.N:
            raise ValueError(f"actions must be a C-contiguous int32 array of {self.B} x {self.N} entries")
        hs = self._host_step
        if self._lib.uavtrack_step_host(self._h, C.c_void_p(actions.ctypes.data), C.byref(hs),
                                        self._stream() if stream is None else stream) != 0:
            _lib.check(1, "uavtrack_step_host")
        if self._host_views is None:      # (the library's block is allocated once per handle: the pointers never change)
            B, N, M = self.B, self.N, self.M

            def view(ptr, shape, ctype, dtype):
                if not ptr:
                    return None
                n = int(np.prod(shape))
                return np.frombuffer((ctype * n).from_address(ptr), dtype=dtype).reshape(shape)
            f, i32, u8 = (C.c_float, np.float32), (C.c_int32, np.int32), (C.c_uint8, np.uint8)
            spec = dict(obs=((B, N, _lib.OBS_DIM), f), reward=((B, N), f), terms=((3, B, N), f), raw=((B, N), f),
                        covered=((B,), i32), done=((B,), u8), ux=((B, N), f), uy=((B, N), f), uz=((B, N), f), uh=((B, N), f),
                        ua=((B, N), i32), tx=((B, M), f), ty=((B, M), f), tz=((B, M), f), th=((B, M), f), step_count=((B,), i32))
            self._host_views = {k: view(getattr(hs, k), shape, ct, dt) for k, (shape, (ct, dt)) in spec.items()}
        return self._host_views

    def _with_targets(self, T: int, want: bool, o, launch):
        """Run `launch()` with a [T, B, M, 2] target trace attached when asked; returns the trace (or None)."""
        if not want:
            launch()
            return None
        tp = self._reuse(o, "targets", (T, self.B, self.M, 2), torch.float32)
        installed = self._trace            # a trace the caller set up earlier comes back afterwards
        self.set_target_trace(tp)
        try:
            launch()
        finally:
            self.set_target_trace(installed)
        return tp

    def step_many(self, actions, want_obs: bool = True, want_terms: bool = True, want_ep_sums: bool = True,
                  out: Optional[Dict[str, torch.Tensor]] = None, want_targets: bool = False,
                  auto_reset_seed: Optional[int] = None, want_raw: bool = False) -> Dict[str, torch.Tensor]:
        """T steps in one launch; `actions` is [T, B, N].  Pass the previous result as
        `out` to reuse its buffers.  want_targets adds "targets" [T, B, M, 2], the target tracks of t_xy<ep>.csv;
        want_raw adds "raw" [T, B, N], uav.raw_reward (environment.py:219).
        auto_reset_seed: environments whose done flag fires are reset inside the launch (reset(seed, next episode)),
        so the launch may span episodes (uavtrack_step_many_autoreset)."""
        a = torch.as_tensor(actions)
        T = int(a.shape[0])
        a = self._actions(a, (T, self.B, self.N))
        o = out or {}

        def buf(key, shape, dtype, want=True):
            return self._reuse(o, key, shape, dtype, want)

        obs = buf("obs", (T, self.B, self.N, _lib.OBS_DIM), torch.float32, want_obs)
        reward = buf("reward", (T, self.B, self.N), torch.float32)
        terms = buf("terms", (T, 3, self.B, self.N), torch.float32, want_terms)
        covered = buf("covered", (T, self.B), torch.int32)
        done = buf("done", (T, self.B), torch.uint8)
        ep = buf("ep_sums", (self.B, 5), torch.float32, want_ep_sums)
        if auto_reset_seed is None:
            launch = lambda: _lib.check(
                self._lib.uavtrack_step_many(self._h, C.c_int32(T), _ptr(a), _ptr(obs), _ptr(reward), _ptr(terms), _ptr(covered),
                                             _ptr(done), _ptr(ep), self._stream()), "uavtrack_step_many")
        else:
            launch = lambda: _lib.check(
                self._lib.uavtrack_step_many_autoreset(self._h, C.c_int32(T), C.c_uint64(auto_reset_seed & (2 ** 64 - 1)), _ptr(a),
                                                       _ptr(obs), _ptr(reward), _ptr(terms), _ptr(covered), _ptr(done), _ptr(ep),
                                                       self._stream()), "uavtrack_step_many_autoreset")
        tp, rb = self._with_raw(T, want_raw, o, lambda: self._with_targets(T, want_targets, o, launch))
        if auto_reset_seed is not None and self.cfg.horizon > 0:
            self._episode += T // self.cfg.horizon + 1      # stay ahead of the episode numbers the device has used
        res = dict(obs=obs, reward=reward, terms=terms, covered=covered, done=done, ep_sums=ep)
        if tp is not None:
            res["targets"] = tp
        if rb is not None:
            res["raw"] = rb          # uav.raw_reward per step (environment.py:219)
        return res

    def bind_step_many(self, actions: torch.Tensor, out: Dict[str, torch.Tensor]):
        """A zero-argument callable that issues `uavtrack_step_many(actions -> out)` on the stream current NOW, with
        every ctypes argument built once: what a driver replays when the per-call Python of step_many (tensor checks,
        slicing, dict building: tens of microseconds) would be a visible share of a short launch."""
        T = int(actions.shape[0])
        a = self._actions(actions, (T, self.B, self.N))
        shapes = dict(obs=((T, self.B, self.N, _lib.OBS_DIM), torch.float32), reward=((T, self.B, self.N), torch.float32),
                      terms=((T, 3, self.B, self.N), torch.float32), covered=((T, self.B), torch.int32),
                      done=((T, self.B), torch.uint8), ep_sums=((self.B, 5), torch.float32))
        for k, (shape, dtype) in shapes.items():
            if out.get(k) is not None and not self._fits(out[k], shape, dtype):
                raise ValueError(f"out[{k!r}] must be a contiguous {dtype} {shape} tensor on {self.device}")
        if out.get("reward") is None:
            raise ValueError("out['reward'] is required")
        args = (self._h, C.c_int32(T), _ptr(a), _ptr(out.get("obs")), _ptr(out["reward"]), _ptr(out.get("terms")),
                _ptr(out.get("covered")), _ptr(out.get("done")), _ptr(out.get("ep_sums")), self._stream())
        fn, keep = self._lib.uavtrack_step_many, (a, out)

        def call():
            if fn(*args) != 0:
                _lib.check(1, "uavtrack_step_many")
            return keep[1]
        return call

    def bind_reset(self, seed: int, obs: torch.Tensor):
        """Callable(episode) issuing `uavtrack_reset` into the caller's obs buffer with pre-built arguments."""
        if not self._fits(obs, (self.B, self.N, _lib.OBS_DIM), torch.float32):
            raise ValueError("obs must be a contiguous float32 [B, N, 12] tensor on this device")
        h, fn, s64, optr, st = self._h, self._lib.uavtrack_reset, C.c_uint64(seed & (2 ** 64 - 1)), _ptr(obs), self._stream()

        def call(episode: int):
            if fn(h, s64, C.c_uint32(episode), optr, st) != 0:
                _lib.check(1, "uavtrack_reset")
            return obs
        return call

    # -- state injection / checkpoint -------------------------------------------------
    def get_state(self) -> Dict[str, torch.Tensor]:
        three = self.cfg.dim == 3
        s = dict(ux=self._empty((self.B, self.N), torch.float32), uy=self._empty((self.B, self.N), torch.float32),
                 uz=self._empty((self.B, self.N), torch.float32) if three else None,
                 uh=self._empty((self.B, self.N), torch.float32), ua=self._empty((self.B, self.N), torch.int32),
                 tx=self._empty((self.B, self.M), torch.float32), ty=self._empty((self.B, self.M), torch.float32),
                 tz=self._empty((self.B, self.M), torch.float32) if three else None,
                 th=self._empty((self.B, self.M), torch.float32),
                 step_count=self._empty((self.B,), torch.int32),
                 episode=self._empty((self.B,), torch.int32))
        _lib.check(self._lib.uavtrack_get_state(self._h, *[_ptr(s[k]) for k in _STATE_KEYS],
                                                _ptr(s["step_count"]), self._stream()), "uavtrack_get_state")
        _lib.check(self._lib.uavtrack_get_episodes(self._h, _ptr(s["episode"]), self._stream()), "uavtrack_get_episodes")
        return {k: v for k, v in s.items() if v is not None}

    def set_state(self, ux, uy, uh, ua, tx, ty, th, uz=None, tz=None, step_count=None, episode=None) -> None:
        """Inject a state (parity tests) or restore a get_state() checkpoint: `env.set_state(**ckpt)`.  `episode` [B]
        (the number of each environment's last reset) keys the next automatic reset; when it is given the host-side
        episode counter moves past its largest entry, so a later reset() does not replay a used episode number."""
        def f(v, shape, dtype):
            if v is None:
                return None
            t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
            t = t.to(device=self.device, dtype=dtype).reshape(shape).contiguous()
            return t
        BN, BM = (self.B, self.N), (self.B, self.M)
        arrs = dict(ux=f(ux, BN, torch.float32), uy=f(uy, BN, torch.float32), uz=f(uz, BN, torch.float32),
                    uh=f(uh, BN, torch.float32), ua=f(ua, BN, torch.int32),
                    tx=f(tx, BM, torch.float32), ty=f(ty, BM, torch.float32), tz=f(tz, BM, torch.float32),
                    th=f(th, BM, torch.float32))
        sc = f(step_count, (self.B,), torch.int32)
        ep = f(episode, (self.B,), torch.int32)
        _lib.check(self._lib.uavtrack_set_state(self._h, *[_ptr(arrs[k]) for k in _STATE_KEYS], _ptr(sc),
                                                self._stream()), "uavtrack_set_state")
        if ep is not None:
            _lib.check(self._lib.uavtrack_set_episodes(self._h, _ptr(ep), self._stream()), "uavtrack_set_episodes")
            self._episode = max(self._episode, int(ep.max().item()) + 1)
        # the D2D copies are stream-ordered; keep the sources alive until they have run
        torch.cuda.current_stream(self.device).synchronize()

    def set_pmi(self, state_dict) -> None:
        """state_dict of a reference PMINetwork (or None to disable)."""
        if state_dict is None:
            _lib.check(self._lib.uavtrack_set_pmi_weights(self._h, None, 0, 0, self._stream()), "set_pmi")
            return
        blob, hidden = fold_pmi_state_dict(state_dict)
        _lib.check(self._lib.uavtrack_set_pmi_weights(self._h, C.c_void_p(blob.ctypes.data), blob.size, hidden,
                                                      self._stream()), "uavtrack_set_pmi_weights")

    def set_pmi_scheme(self, scheme: str = "auto") -> None:
        """Pin the MAAC-R pair scorer: "auto" (default: the fastest the weights allow), "f16x3", "bf16x6" or "fp32"
        (uavtrack_set_pmi_scheme).  Raises if the loaded weights cannot run on it."""
        if scheme not in _lib.PMI_SCHEMES:
            raise ValueError(f"scheme must be one of {_lib.PMI_SCHEMES}")
        _lib.check(self._lib.uavtrack_set_pmi_scheme(self._h, C.c_int32(_lib.PMI_SCHEMES.index(scheme))), "uavtrack_set_pmi_scheme")

    def pmi_info(self) -> Dict[str, object]:
        """{"scheme": the scorer the next MAAC-R step launches (None without weights), "hidden_padded", "f16_range_ok": the
        host-side range guard passed, "rescored_chunks": chunks the wide-range kernel scored again because an operand left
        f16's range at run time}.  Synchronises the stream."""
        out = (C.c_int64 * 4)()
        _lib.check(self._lib.uavtrack_pmi_info(self._h, out, self._stream()), "uavtrack_pmi_info")
        return dict(scheme=_lib.PMI_SCHEMES[out[0]] if out[0] else None, hidden_padded=int(out[1]), f16_range_ok=bool(out[2]),
                    rescored_chunks=int(out[3]))

    def launch_info(self) -> Dict[str, int]:
        """Geometry of the most recent rollout launch (uavtrack_launch_info)."""
        out = (C.c_int64 * 4)()
        _lib.check(self._lib.uavtrack_launch_info(self._h, out), "uavtrack_launch_info")
        return dict(workgroup=int(out[0]), envs_per_workgroup=int(out[1]), workgroups=int(out[2]), single_wavefront_variant=int(out[3]))

    def pmi_inference(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """PMINetwork.inference (PMINet.py:64-72) on a batch of pair inputs x [n, 12] (= la_i * la_j, uav.py:281) with the
        uploaded weights, on the MAAC-R scorer kernels -> scores [n]."""
        if x.dim() != 2 or not self._fits(x, (x.shape[0], _lib.OBS_DIM), torch.float32):
            raise ValueError(f"x must be a contiguous float32 [n, {_lib.OBS_DIM}] tensor on {self.device}")
        n = int(x.shape[0])
        s = self._out_arg(out, (n,), torch.float32, "out")
        _lib.check(self._lib.uavtrack_pmi_inference(self._h, _ptr(x), C.c_int64(n), _ptr(s), self._stream()),
                   "uavtrack_pmi_inference")
        return s

    def greedy_actions(self, seed: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The reference's C-METHOD baseline policy (uav.py:324-369) for every UAV -> int32 [B, N]."""
        a = self._out_arg(out, (self.B, self.N), torch.int32, "out")
        _lib.check(self._lib.uavtrack_greedy_actions(self._h, C.c_uint64(seed & (2 ** 64 - 1)), _ptr(a),
                                                     self._stream()), "uavtrack_greedy_actions")
        return a

    def run_greedy(self, T: int, seed: int = 0, want_obs: bool = True, want_terms: bool = True,
                   want_actions: bool = True, want_targets: bool = False,
                   out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """T closed-loop steps of the C-METHOD baseline (train.py:326-370) in one launch.  Pass a previous result as
        `out` to reuse its buffers."""
        o = out or {}

        def buf(key, shape, dtype, want=True):
            return self._reuse(o, key, shape, dtype, want)
        acts = buf("actions", (T, self.B, self.N), torch.int32, want_actions)
        obs = buf("obs", (T, self.B, self.N, _lib.OBS_DIM), torch.float32, want_obs)
        reward = buf("reward", (T, self.B, self.N), torch.float32)
        terms = buf("terms", (T, 3, self.B, self.N), torch.float32, want_terms)
        covered = buf("covered", (T, self.B), torch.int32)
        done = buf("done", (T, self.B), torch.uint8)
        ep = buf("ep_sums", (self.B, 5), torch.float32)
        tp = self._with_targets(T, want_targets, o, lambda: _lib.check(
            self._lib.uavtrack_run_greedy(self._h, C.c_int32(T), C.c_uint64(seed & (2 ** 64 - 1)), _ptr(acts), _ptr(obs),
                                          _ptr(reward), _ptr(terms), _ptr(covered), _ptr(done), _ptr(ep), self._stream()),
            "uavtrack_run_greedy"))
        res = dict(actions=acts, obs=obs, reward=reward, terms=terms, covered=covered, done=done, ep_sums=ep)
        if tp is not None:
            res["targets"] = tp
        return res

    # ---- the learner's shared actor on the device (actor_critic.py:85-98, 138-148) ----
    def set_actor(self, actor) -> None:
        """Upload FnnPolicyNet parameters: a module / state_dict with fc1.weight [H,12], fc1.bias, fc2.weight
        [na,H], fc2.bias (actor_critic.py:85-98), or None to remove them."""
        if actor is None:
            _lib.check(self._lib.uavtrack_set_actor_weights(self._h, None, None, None, None, C.c_int32(0),
                                                            self._stream()), "uavtrack_set_actor_weights")
            return
        sd = actor.state_dict() if hasattr(actor, "state_dict") else actor
        host = [sd[k].detach().to("cpu", torch.float32).contiguous()
                for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")]
        w1, b1, w2, b2 = host
        if w1.shape[1] != _lib.OBS_DIM or w2.shape != (self.cfg.na_total, w1.shape[0]):
            raise ValueError(f"actor shapes fc1 {tuple(w1.shape)}, fc2 {tuple(w2.shape)} do not match "
                             f"Linear({_lib.OBS_DIM}, H) / Linear(H, {self.cfg.na_total})")
        _lib.check(self._lib.uavtrack_set_actor_weights(self._h, *[C.c_void_p(t.data_ptr()) for t in host],
                                                        C.c_int32(w1.shape[0]), self._stream()),
                   "uavtrack_set_actor_weights")

    def actor_actions(self, obs: torch.Tensor, seed: int = 0, mode: int = _lib.ACTOR_SAMPLE,
                      want_probs: bool = False, out: Optional[torch.Tensor] = None):
        """take_action for every UAV (actor_critic.py:138-148) -> int32 [B, N] (and probs [B, N, na] if asked)."""
        if obs.shape != (self.B, self.N, _lib.OBS_DIM) or obs.dtype != torch.float32 or not obs.is_contiguous() \
                or obs.device != self.device:
            raise ValueError(f"obs must be a contiguous float32 [{self.B}, {self.N}, {_lib.OBS_DIM}] tensor on {self.device}")
        a = self._out_arg(out, (self.B, self.N), torch.int32, "out")
        probs = self._empty((self.B, self.N, self.cfg.na_total), torch.float32) if want_probs else None
        _lib.check(self._lib.uavtrack_actor_actions(self._h, _ptr(obs), C.c_uint64(seed & (2 ** 64 - 1)),
                                                    C.c_int32(mode), _ptr(a), _ptr(probs), self._stream()),
                   "uavtrack_actor_actions")
        return (a, probs) if want_probs else a

    def run_actor(self, T: int, obs_in: torch.Tensor, seed: int = 0, mode: int = _lib.ACTOR_SAMPLE,
                  want_terms: bool = True, out: Optional[Dict[str, torch.Tensor]] = None,
                  want_targets: bool = False) -> Dict[str, torch.Tensor]:
        """T closed-loop steps of actor + environment (the rollout of train.operate_epoch, train.py:160-192) in
        one launch.  obs_in [B, N, 12] is what the policy sees first (reset()'s return or the last obs)."""
        if obs_in.shape != (self.B, self.N, _lib.OBS_DIM) or obs_in.dtype != torch.float32 \
                or not obs_in.is_contiguous() or obs_in.device != self.device:
            raise ValueError(f"obs_in must be a contiguous float32 [{self.B}, {self.N}, {_lib.OBS_DIM}] tensor on {self.device}")
        o = out or {}

        def buf(key, shape, dtype, want=True):
            return self._reuse(o, key, shape, dtype, want)
        acts = buf("actions", (T, self.B, self.N), torch.int32)
        obs = buf("obs", (T, self.B, self.N, _lib.OBS_DIM), torch.float32)
        reward = buf("reward", (T, self.B, self.N), torch.float32)
        terms = buf("terms", (T, 3, self.B, self.N), torch.float32, want_terms)
        covered = buf("covered", (T, self.B), torch.int32)
        done = buf("done", (T, self.B), torch.uint8)
        ep = buf("ep_sums", (self.B, 5), torch.float32)
        tp = self._with_targets(T, want_targets, o, lambda: _lib.check(
            self._lib.uavtrack_run_actor(self._h, C.c_int32(T), C.c_uint64(seed & (2 ** 64 - 1)), C.c_int32(mode), _ptr(obs_in),
                                         _ptr(acts), _ptr(obs), _ptr(reward), _ptr(terms), _ptr(covered), _ptr(done), _ptr(ep),
                                         self._stream()), "uavtrack_run_actor"))
        res = dict(actions=acts, obs=obs, reward=reward, terms=terms, covered=covered, done=done, ep_sums=ep)
        if tp is not None:
            res["targets"] = tp
        return res

    def bind_run(self, T: int, out: Dict[str, torch.Tensor], policy: str = "actor", obs_in: Optional[torch.Tensor] = None,
                 seed: int = 0, mode: int = _lib.ACTOR_SAMPLE):
        """A zero-argument callable that issues `uavtrack_run_actor` / `uavtrack_run_greedy` (T steps, policy inside the
        kernel) into `out` on the stream current NOW, with every ctypes argument built once -- what a driver that issues
        short fused chunks replays (the per-call Python of run_actor is a visible share of a 10-step launch)."""
        shapes = dict(actions=((T, self.B, self.N), torch.int32), obs=((T, self.B, self.N, _lib.OBS_DIM), torch.float32),
                      reward=((T, self.B, self.N), torch.float32), terms=((T, 3, self.B, self.N), torch.float32),
                      covered=((T, self.B), torch.int32), done=((T, self.B), torch.uint8), ep_sums=((self.B, 5), torch.float32))
        for k, (shape, dtype) in shapes.items():
            if out.get(k) is not None and not self._fits(out[k], shape, dtype):
                raise ValueError(f"out[{k!r}] must be a contiguous {dtype} {shape} tensor on {self.device}")
        if out.get("reward") is None:
            raise ValueError("out['reward'] is required")
        tail = (_ptr(out.get("actions")), _ptr(out.get("obs")), _ptr(out["reward"]), _ptr(out.get("terms")), _ptr(out.get("covered")),
                _ptr(out.get("done")), _ptr(out.get("ep_sums")), self._stream())
        s64 = C.c_uint64(seed & (2 ** 64 - 1))
        if policy == "actor":
            if not self._fits(obs_in, (self.B, self.N, _lib.OBS_DIM), torch.float32):
                raise ValueError("obs_in must be a contiguous float32 [B, N, 12] tensor on this device")
            fn, name = self._lib.uavtrack_run_actor, "uavtrack_run_actor"
            args = (self._h, C.c_int32(T), s64, C.c_int32(mode), _ptr(obs_in)) + tail
        elif policy == "greedy":
            fn, name = self._lib.uavtrack_run_greedy, "uavtrack_run_greedy"
            args = (self._h, C.c_int32(T), s64) + tail
        else:
            raise ValueError("policy must be 'actor' or 'greedy'")
        keep = (out, obs_in)

        def call():
            if fn(*args) != 0:
                _lib.check(1, name)
            return keep[0]
        return call

    @staticmethod
    def clip_saturation(terms: torch.Tensor, reward: torch.Tensor) -> Dict[str, int]:
        """Diagnostic counterpart of the reference's "overstep in clip." print (data_util.py:44-47): how many values of a
        rollout's outputs sit ON a clip bound -- terms [..., 3, B, N] (tracking at 1, duplicate at -1) and reward [..., B, N]
        (at +-1).  A raw value beyond the bound ends there (the kernels clamp like `np.clip`), so a non-zero count where
        `EnvConfig.clip_can_overstep()` says it can happen is the reference's message; exactly on the bound is counted too."""
        tt, dup = terms.select(-3, 0), terms.select(-3, 2)
        return {"tracking": int((tt >= 1.0).sum()), "duplicate": int((dup <= -1.0).sum()),
                "reward": int((reward.abs() >= 1.0).sum())}

    def set_profiling(self, on: bool) -> None:
        """HIP event pairs around every kernel launch of the stepping calls (uavtrack_set_profiling); read with profile()."""
        _lib.check(self._lib.uavtrack_set_profiling(self._h, C.c_int32(1 if on else 0)), "uavtrack_set_profiling")

    def profile(self) -> Dict[str, Dict[str, float]]:
        """{kernel class: {"ms": total, "launches": n}} since the last call (synchronises the stream)."""
        ms = (C.c_double * len(_lib.PROF_CLASSES))()
        cnt = (C.c_int64 * len(_lib.PROF_CLASSES))()
        _lib.check(self._lib.uavtrack_get_profile(self._h, ms, cnt, self._stream()), "uavtrack_get_profile")
        return {k: {"ms": float(ms[i]), "launches": int(cnt[i])} for i, k in enumerate(_lib.PROF_CLASSES)}

    def pmi_pairs_scored(self) -> int:
        """Neighbour pairs the PMI network has scored so far (synchronises the stream)."""
        out = C.c_uint64(0)
        _lib.check(self._lib.uavtrack_pmi_pairs_scored(self._h, C.byref(out), self._stream()), "pmi_pairs_scored")
        return int(out.value)

    @property
    def reward_mode(self) -> RewardMode:
        return self.cfg.resolved_mode()
