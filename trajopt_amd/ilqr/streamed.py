"""`StreamedILQR` - one batch of problems as a few sub-batches, each a `BatchedILQR` on a HIP stream of its own.

The problems of a batch are independent (the reference solves one at a time, /root/reference/trajopt/ilqr/ilqr.py:136-237,
and fans them out over processes, examples/ilqr/mpc/pendulum_parallel.py:40-45), and an iteration of the batched driver is a
chain of launches of which several do not fill the chip: the backward pass of the larger models ends in a tail of the few
problems that climb the lambda ladder, the rollout of 8192 double cartpoles leaves a third of the SIMDs with one wave.
Two to four sub-batches on streams of their own run those launches side by side - one sub-batch's rollout under
another's backward pass.  Measured on MI355X (ms per iteration, one batch -> sub-batches; scripts/debug/two_streams.py):
4096 quadrotors 7.35 -> 5.97 (4; never giving up 19.4 -> 14.9), 8192 double cartpoles 2.97 -> 2.67 (2), 16 384 double
cartpoles 5.62 -> 5.19 (4); 4096 and 16 384 cartpoles lose (0.66 -> 0.76, 2.10 -> 2.35: their launches are a lone wave's
chain each, which a second stream only delays) and 65 536 gain nothing reliable, so the small models stay one batch.

Every problem runs the same algorithm on the same numbers as in one batch; which kernel variant a launch takes depends on
the batch size, and the variants agree to round-off (tests/test_ilqr_gpu.py).
"""
import numpy as np

from .ilqr import BatchedILQR


def auto_sub_batches(dm_state, dm_act, nb_problems):
    """Number of sub-batches `StreamedILQR` takes by default (the measurements in the module docstring)."""
    n, m, N = int(dm_state), int(dm_act), int(nb_problems)
    if n + m <= 5:                      # pendulum, cartpole: never (65 536 cartpoles: 6.77 -> 6.55 one day, 6.77 -> 8.18 another)
        return 1
    if n <= 6:                          # double cartpole, double pendulum: lane-per-problem backward kernels
        if 4096 <= N < 16384:           # 4096: 1.68 -> 1.60, 6144: 2.55 -> 2.07, 12 288: 4.54 -> 3.85; 32 768: 9.8 -> 10.4
            return 2
        return 4 if 16384 <= N < 32768 else 1
    return 4 if N >= 1024 else 1        # quadrotor and larger: matrix-core backward pass with the lambda ladder


def split_sizes(nb_problems, parts):
    """Sizes of `parts` contiguous sub-batches of `nb_problems` problems, as equal as possible."""
    q, r = divmod(int(nb_problems), int(parts))
    return [q + 1 if i < r else q for i in range(int(parts))]


_SIDE_STREAMS = {}   # device index -> side streams, chosen once per process and shared by every StreamedILQR
_PROBE_CYCLES = 400_000      # ~0.2 ms of torch.cuda._sleep
_PROBE_CANDIDATES = 12


def _concurrent(torch, streams, one):
    """Do one-wave spins on `streams`, started together, take about as long as one of them (`one` ms)?"""
    import time
    best = float("inf")
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(_PROBE_CYCLES)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best < 1.6 * one


def side_streams(device, count):
    """Up to `count` side streams of `device` that run CONCURRENTLY with the current stream and with each other.

    The runtime deals HIP streams out over GPU_MAX_HW_QUEUES (4) hardware queues, round robin over every stream the
    process makes (RCCL's included), and two streams on one queue run their kernels one after the other: sub-batches on
    such a pair are slower than one batch (measured: 4096 quadrotors 9.4 - 9.8 ms an iteration instead of 6.1 - 6.4, one
    batch 7.8; with fresh streams per solver that happened in every bench.py process and in one of eight runs of a loop,
    with RCCL initialised in most).  So the streams are CHOSEN, once per process: candidates from torch's pool are tried
    with a one-wave spin (torch.cuda._sleep) next to the current stream and the streams already chosen - together they must
    take about as long as one spin alone (scripts/debug/stream_probe.py shows the pattern: every fourth stream collides).
    Fewer than `count` may come back (the caller then makes fewer sub-batches)."""
    import time
    import torch
    dev = torch.device(device)
    have = _SIDE_STREAMS.get(dev.index)
    if have is None:
        have = []
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            cands = [torch.cuda.Stream(device=dev) for _ in range(_PROBE_CANDIDATES)]
            if hasattr(torch.cuda, "_sleep"):
                for st in [cur] + cands:          # first use of a stream binds its queue: not part of the timing
                    with torch.cuda.stream(st):
                        torch.cuda._sleep(1000)
                one = float("inf")
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    torch.cuda._sleep(_PROBE_CYCLES)
                    torch.cuda.synchronize()
                    one = min(one, (time.perf_counter() - t0) * 1e3)
                for st in cands:
                    if len(have) == StreamedILQR.MAX_SUB_BATCHES - 1:
                        break
                    if _concurrent(torch, [cur] + have + [st], one):
                        have.append(st)
            else:
                have = cands[:StreamedILQR.MAX_SUB_BATCHES - 1]
        _SIDE_STREAMS[dev.index] = have
    return have[:count]


class StreamedILQR:
    """`BatchedILQR` over `sub_batches` contiguous slices of the problems: the first on the caller's stream, the others
    on side streams (at most three: with the caller's stream that is one hardware queue each).

    Arguments as `BatchedILQR` (init_state (N, dm_state), kff_init (N, dm_act, nb_steps) or None: drawn once for the
    whole batch as ilqr.py:70); sub_batches None = `auto_sub_batches`.  The object is stream-ordered on the stream that is
    current when a method is called: the side streams wait for it before they start and it waits for them before the method
    returns (no host synchronisation in `iterate`).  `parts[i]` are the sub-solvers, `bounds[i]:bounds[i+1]` their problems.
    """
    MAX_SUB_BATCHES = 4

    def __init__(self, env, nb_steps, init_state, kff_init=None, sub_batches=None, **kw):
        import torch
        u = env.unwrapped if hasattr(env, "unwrapped") else env
        x0 = init_state if torch.is_tensor(init_state) else np.asarray(init_state, dtype=np.float64)
        if x0.ndim == 1:
            x0 = x0.reshape(1, -1)
        N, T, m = int(x0.shape[0]), int(nb_steps), int(u.dm_act)
        S = auto_sub_batches(u.dm_state, u.dm_act, N) if sub_batches is None else int(sub_batches)
        S = max(1, min(S, N, self.MAX_SUB_BATCHES))
        self.device = torch.device("cuda", torch.cuda.current_device()) if kw.get("device") is None else torch.device(kw["device"])
        self.streams = side_streams(self.device, S - 1) if S > 1 else []   # of sub-batches 1 .. S - 1
        S = 1 + len(self.streams)
        if kff_init is None:
            kff_init = 1e-4 * np.random.randn(N, m, T)     # ilqr.py:70, one draw for the batch
        kff_init = kff_init if torch.is_tensor(kff_init) else np.asarray(kff_init, dtype=np.float64).reshape(N, m, T)
        self.nb_problems, self.nb_steps, self.sub_batches = N, T, S
        self.bounds = np.concatenate(([0], np.cumsum(split_sizes(N, S)))).tolist()
        self._start = torch.cuda.Event()
        self._done = [torch.cuda.Event() for _ in self.streams]
        self.parts = []
        # the first env kwargs application (ilqr.py:34-36) changes the env once; the sub-solvers share it
        for i, part in self._each():
            lo, hi = self.bounds[i], self.bounds[i + 1]
            self.parts.append(BatchedILQR(env, T, x0[lo:hi], kff_init=kff_init[lo:hi], **kw))
        self._join()

    # -- stream plumbing ---------------------------------------------------------------------------------------
    def _each(self):
        """Iterate (i, sub-solver or None): sub-batch 0 on the caller's stream, sub-batch i > 0 with side stream i - 1 current
        after it has been made to wait for what the caller's stream held when the method was called."""
        import torch
        part = lambda i: self.parts[i] if i < len(self.parts) else None
        if self.streams:
            self._start.record(torch.cuda.current_stream(self.device))
        yield 0, part(0)
        for i, st in enumerate(self.streams):
            st.wait_event(self._start)
            with torch.cuda.stream(st):
                yield i + 1, part(i + 1)
            self._done[i].record(st)

    def _join(self):
        import torch
        cur = torch.cuda.current_stream(self.device)
        for ev in self._done:
            cur.wait_event(ev)

    def synchronize(self):
        import torch
        torch.cuda.current_stream(self.device).synchronize()
        for st in self.streams:
            st.synchronize()

    # -- the solver's interface --------------------------------------------------------------------------------
    def reset(self, init_state, kff_init=None, generator=None):
        """`BatchedILQR.reset` for every sub-batch (init_state (N, dm_state), kff_init (N, dm_act, nb_steps) or None).
        With a device `generator` every sub-batch draws its own block from it in turn: reproducible for a seed and a
        sub-batch count, but not the numbers one batch would draw."""
        import torch
        N = self.nb_problems
        x0 = init_state if torch.is_tensor(init_state) else np.asarray(init_state, dtype=np.float64)
        x0 = x0.reshape(N, -1)
        if kff_init is None and generator is None:
            kff_init = 1e-4 * np.random.randn(N, self.parts[0].dm_act, self.nb_steps)
        if kff_init is not None and not torch.is_tensor(kff_init):
            kff_init = np.asarray(kff_init, dtype=np.float64).reshape(N, -1, self.nb_steps)
        for i, part in self._each():
            lo, hi = self.bounds[i], self.bounds[i + 1]
            part.reset(x0[lo:hi], None if kff_init is None else kff_init[lo:hi], generator=generator)
        self._join()

    def init_rollout(self):
        for _, part in self._each():
            part.init_rollout()
        self._join()

    def iterate(self, nb_iter=1):
        """`nb_iter` iterations of every sub-batch, each enqueued on its own stream; asynchronous."""
        for _, part in self._each():
            part.iterate(nb_iter)
        self._join()

    def plan(self):
        """The plan of the first sub-batch (the others differ by at most one problem) plus `sub_batches`."""
        return dict(self.parts[0].plan(), sub_batches=self.sub_batches)

    def run(self, nb_iter=25, verbose=False):
        """iLQR.run for the whole batch: the list of per-problem cost traces (ilqr.py:137,237), in problem order."""
        for _, part in self._each():
            if not part._initialised:
                part.init_rollout()
            part.iterate(nb_iter)
        self._join()
        self.synchronize()
        traces = []
        for part in self.parts:
            tl, tr = part.trace_len.cpu().numpy(), part.trace.cpu().numpy()
            traces += [tr[:tl[p], p].tolist() for p in range(part.nb_problems)]
        if verbose:
            st = self.host("status")
            print("streamed iLQR: %d problems in %d sub-batches, mean return %.6g, status counts %s" % (
                self.nb_problems, self.sub_batches, float(self.host("last_return").mean()),
                np.bincount(st, minlength=4).tolist()))
        return traces

    def vector(self, name):
        """Device tensor (N,) of a per-problem vector (last_return, status, iters, lmbda ...) of the whole batch, in problem
        order, on the caller's stream (a copy unless the batch is one sub-batch)."""
        import torch
        self._join()
        vs = [getattr(part, name) for part in self.parts]
        return vs[0] if len(vs) == 1 else torch.cat(vs)

    def host(self, name):
        """Logical NumPy array (N, rows[, cols], T) or (N,) vector of the whole batch, in problem order."""
        self.synchronize()
        return np.concatenate([part.host(name) for part in self.parts], axis=0)
