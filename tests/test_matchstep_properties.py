"""Step-sequence invariants of the exact-final-time (MATCHSTEP) / time-span logic, SURVEY 8a-5
(reference: /root/reference/pnode/petsc_adjoint.py:640 ``setExactFinalTime(MATCHSTEP)``, :812-827 ``setTimeSpan``).

Nothing PETSc-produced exists to pin this logic, and the oracle's state machine
(oracle/petsc_ts_restated.c) was restated by the same hand as the product's (pnode_amd/csrc/pn_ts.cpp): agreement
between the two proves little (round 3 shipped a defect both shared).  So the checks here are INVARIANTS derived from
what PETSc documents the logic to do, plus a third, independent statement of the rule (`spec_sequence`: a dozen lines,
interval by interval, no cache variable at all), and only then oracle == product.

  I1  every output time is reached bit-exactly, in order;
  I2  the steps of an output interval of length D add up to D, and there are at most ceil(D/h) + 1 of them;
  I3  no step is shorter than min(h, D)/2 (minus round-off): the halving rule never cascades (exception: the first
      interval, whose first step is only clamped -- PETSc adjusts AFTER a step, so h = 0.3 towards 0.31 leaves 0.01);
  I4  after every output time a fixed-step run goes back to `step_size` unless the next interval forces a cut
      (D < 2h: D/2; D <= 1.01h: D);
  I5  the accepted-step log of the oracle and of the product's host engine are identical.
"""
import ctypes
import math
import random

import numpy as np
import pytest
import torch

from oracle.theta_oracle import step_plan
from pnode_amd import _lib


def same_log(plan, log, t_end):
    """I5: the oracle logs (end time - step, end time - start time) of every accepted step; the end times must be the
    product's bit for bit (they are what the next step starts from), the sizes agree to the rounding of that
    subtraction."""
    if len(plan) != len(log):
        return False
    ends_o = [a + b for a, b in plan]
    ends_p = [t for t, _ in log[1:]] + [t_end]
    return (np.allclose(ends_o, ends_p, rtol=1e-15, atol=0) and
            np.allclose([x for _, x in plan], [x for _, x in log], rtol=1e-13, atol=0))

STRETCH = 0.01          # TSAdapt's matchstepfac[0]: a step within 1 % of the remaining distance is stretched onto the point
HALVE = 2.0             # matchstepfac[1]: a remaining distance below two steps is split in two


def spec_sequence(h, times):
    """The rule as PETSc's manual states it, per output interval, with no state carried between intervals other than
    `h` itself: full steps while at least two fit, then the remainder in one piece when it is within 1 % of a step,
    otherwise in two halves."""
    out = []
    for i, (a, b) in enumerate(zip(times[:-1], times[1:])):
        rem = b - a
        seq = []
        if i == 0:
            # the very first step of a solve is only CLAMPED to the first target (TSSolve), never stretched or halved:
            # those adjustments are made by TSAdaptChoose, i.e. after a step
            first = min(h, rem)
            seq.append(first)
            rem = rem - first if first < rem else 0.0
        while rem > 0:
            if h * (1.0 + STRETCH) > rem:
                seq.append(rem)
                rem = 0.0
            elif h * HALVE > rem:
                seq += [rem / 2, rem / 2]
                rem = 0.0
            else:
                seq.append(h)
                rem -= h
        out.append(seq)
    return out


def product_sequence(h0, times, adapt="none", enorms=None, rk="4"):
    """Drive the product's host engine (the C++ state machine the GPU path uses, pn_ts_*): returns
    ([(t_n, h_n)], [time after the step that hit output i], steps per interval)."""
    lib = _lib.load()
    ts = ctypes.c_void_p(lib.pn_ts_create())
    try:
        _lib.check(lib.pn_ts_set_option(ts, b"ts_adapt_type", adapt.encode()))
        _lib.check(lib.pn_ts_set_option(ts, b"ts_rk_type", rk.encode()))
        _lib.check(lib.pn_ts_set_option(ts, b"ts_max_steps", b"5000"))
        n = len(times)
        _lib.check(lib.pn_ts_begin(ts, 0.0, h0, n, (ctypes.c_double * n)(*times)))
        acc, hit, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(0)
        hits, per, count = [], [], 0
        k = 0
        while not done.value:
            e = -1.0 if enorms is None else enorms[k % len(enorms)]
            k += 1
            _lib.check(lib.pn_ts_judge(ts, e, ctypes.byref(acc), ctypes.byref(hit), ctypes.byref(done)))
            if not acc.value:
                continue
            count += 1
            if hit.value >= 0:
                hits.append((hit.value, lib.pn_ts_time(ts)))
                per.append(count)
                count = 0
        if done.value != 1:
            return None, None, None          # stopped by -ts_max_steps (an injected error sequence that keeps shrinking the step)
        t, h = ctypes.c_double(), ctypes.c_double()
        log = []
        for i in range(lib.pn_ts_steps(ts)):
            _lib.check(lib.pn_ts_step_log(ts, i, ctypes.byref(t), ctypes.byref(h)))
            log.append((t.value, h.value))
        return log, hits, per
    finally:
        lib.pn_ts_destroy(ts)


def check_invariants(h, times, log, hits, per, fixed=True):
    # I1
    assert [i for i, _ in hits] == list(range(1, len(times)))
    assert [t for _, t in hits] == list(times[1:])
    k = 0
    for i, cnt in enumerate(per):
        D = times[i + 1] - times[i]
        hs = [log[k + j][1] for j in range(cnt)]
        k += cnt
        # I2
        assert math.isclose(sum(hs), D, rel_tol=1e-12, abs_tol=1e-15)
        if fixed:
            assert cnt <= math.ceil(D / h - 1e-12) + 1
            if i == 0:
                assert math.isclose(hs[0], min(h, D), rel_tol=1e-12)
                assert min(hs[:1] + hs[2:]) >= 0.5 * min(h, D) * (1 - 1e-12)
                continue
            # I3
            assert min(hs) >= 0.5 * min(h, D) * (1 - 1e-12)
            # I4
            if D >= HALVE * h * (1 + 1e-12):
                # (to 10 eps: a first step within round-off of the first output time IS that distance from then on)
                assert math.isclose(hs[0], h, rel_tol=1e-14), (h, times, i, hs)
            elif D > (1 + STRETCH) * h * (1 + 1e-12):
                assert math.isclose(hs[0], D / 2, rel_tol=1e-12)
            elif D < (1 + STRETCH) * h * (1 - 1e-12):
                assert math.isclose(hs[0], D, rel_tol=1e-12)
    assert k == len(log)


JUDGE_CASES = [      # the three sequences
    (0.3, [0.0, 0.7, 1.5], [0.3, 0.2, 0.2, 0.3, 0.25, 0.25]),
    (0.3, [0.0, 0.75, 1.5], [0.3, 0.225, 0.225, 0.3, 0.225, 0.225]),
    (0.3, [0.0, 0.8, 2.0], [0.3, 0.25, 0.25, 0.3, 0.3, 0.3, 0.3]),
]


@pytest.mark.parametrize("h,times,expected", JUDGE_CASES)
def test_the_cut_step_comes_back_after_the_output_time(h, times, expected):
    log, hits, per = product_sequence(h, times)
    assert np.allclose([x for _, x in log], expected, rtol=1e-12)
    plan, per_o = step_plan(torch.tensor(times, dtype=torch.float64), h)
    assert same_log(plan, log, times[-1])                                 # I5
    assert per_o[1:] == per
    check_invariants(h, times, log, hits, per)


def _grid():
    cases = []
    for h in (0.3, 0.1, 0.025, 0.07, 1.0 / 3.0):
        for times in ([0.0, 0.7, 1.5], [0.0, 0.75, 1.5], [0.0, 0.8, 2.0], [0.0, 0.31, 0.32, 1.0, 1.05, 2.0],
                      [0.0, 0.05, 0.1, 0.15000000000000002, 0.2], list(np.linspace(0.0, 1.0, 8)),
                      list(np.linspace(0.0, 25.0, 1000)[:7]), [0.0, 1e-3, 2.5], [0.0, 0.9999999999999999, 2.0000000000000004]):
            cases.append((h, [float(x) for x in times]))
    rng = random.Random(7)
    for _ in range(60):                  # non-commensurate and float-noisy output times
        h = rng.choice([0.3, 0.1, 0.013, 0.25]) * (1 + rng.choice([0, 0, 1e-16, -1e-16, 3e-3]))
        pts = sorted(set([0.0] + [rng.uniform(0.01, 3.0) for _ in range(rng.randint(1, 6))]))
        if rng.random() < 0.3:           # output times that ARE multiples of h up to round-off
            pts = [0.0] + [k * h * (1 + rng.choice([0, 2e-16, -2e-16])) for k in sorted(rng.sample(range(1, 12), 3))]
        cases.append((h, pts))
    return cases


def test_fixed_step_invariants_and_oracle_agreement_on_a_grid():
    for h, times in _grid():
        if min(b - a for a, b in zip(times[:-1], times[1:])) <= 0:
            continue
        log, hits, per = product_sequence(h, times)
        check_invariants(h, times, log, hits, per)
        plan, per_o = step_plan(torch.tensor(times, dtype=torch.float64), h)
        assert same_log(plan, log, times[-1]), (h, times)                      # I5
        assert per_o[1:] == per
        # the independent statement of the rule, wherever no decision sits on a boundary within round-off
        spec = spec_sequence(h, times)
        near = False
        for seq, D in zip(spec, np.diff(times)):
            rem = D
            while rem > 1e-14:
                for edge in (h * (1 + STRETCH), h * HALVE):
                    near |= abs(rem - edge) < 1e-9 * max(h, 1.0)
                rem -= h
        if not near:
            flat = [x for seq in spec for x in seq]
            assert len(flat) == len(log) and np.allclose(flat, [x for _, x in log], rtol=1e-10, atol=1e-13), (h, times)
            assert [len(s) for s in spec] == per


def test_adaptive_sequences_hit_every_output_time_and_never_reuse_a_stale_step():
    """Controller outputs injected into the host engine (error norms in a fixed pseudo-random order, some above 1):
    I1/I2 hold, and the step after an output time is the controller's choice (clipped onto the next interval), never
    a step cached several choices ago."""
    rng = random.Random(3)
    completed = 0
    for trial in range(40):
        enorms = [rng.choice([0.02, 0.3, 0.8, 0.95, 1.7, 4.0, 0.5, 0.0]) for _ in range(17)]
        pts = sorted(set([0.0] + [rng.uniform(0.05, 4.0) for _ in range(rng.randint(1, 5))]))
        h0 = rng.choice([0.01, 0.2, 1.0])
        log, hits, per = product_sequence(h0, pts, adapt="basic", enorms=enorms, rk="5dp")
        if log is None:
            continue
        completed += 1
        check_invariants(h0, pts, log, hits, per, fixed=False)
        # between two accepted steps the size changes by at most the clip factors (0.1, 10) compounded over the
        # rejected attempts in between, or it was cut for an output time; it is never LARGER than 10x the last one
        hs = [x for _, x in log]
        for a, b in zip(hs[:-1], hs[1:]):
            assert b <= 10.0 * a * (1 + 1e-12)
    assert completed >= 30


def test_adaptive_runs_of_oracle_and_product_take_the_same_steps_across_output_times():
    """dopri5 / bosh3 on the spiral with output times the steps do not divide: oracle and product (CPU stand-in for
    the device ops) log the same accepted steps, bit for bit, and count the same steps per interval."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_host_engine import _pair
    from problems import SpiralTruth, rel_err
    torch.manual_seed(1)
    y0 = torch.tensor([[2.0, 0.0], [1.0, 1.5], [-0.5, 1.0]], dtype=torch.float64)
    for method, h0 in (("dopri5", 0.3), ("bosh3", 0.2), ("dopri5", 0.01)):
        for t in ([0.0, 0.7, 1.5], [0.0, 0.75, 0.8, 2.9], [0.0, 0.05, 3.0]):
            tt = torch.tensor(t, dtype=torch.float64)
            target = torch.zeros(len(t), 3, 2, dtype=torch.float64)
            a, b = _pair(SpiralTruth, y0, tt, target, method, {}, step_size=h0)
            te, h, _ = a[3].step_log()
            log = b[3].step_log()
            # (the error norms of the two sides are formed in different orders, and an estimate far below the tolerance is mostly cancellation: the steps agree to ~1e-10)
            assert len(log) == len(h) and np.allclose([x for _, x in log], h, rtol=1e-8, atol=0), (method, h0, t)
            assert np.allclose([float(x) for x in te[:-1]], [tt for tt, _ in log[1:]], rtol=1e-9, atol=0)
            assert [float(te[sum(a[3].cur_sol_steps[:i + 1]) - 1]) for i in range(1, len(t))] == t[1:]   # oracle lands bit-exactly too
            assert b[3].cur_sol_steps == a[3].cur_sol_steps
            assert rel_err(b[0], a[0]) < 1e-12 and rel_err(b[1], a[1]) < 1e-11
            k = 0
            for i in range(1, len(t)):                       # I1/I2 on the product's log
                cnt = b[3].cur_sol_steps[i]
                assert math.isclose(sum(x for _, x in log[k:k + cnt]), t[i] - t[i - 1], rel_tol=1e-12)
                k += cnt
                assert math.isclose(log[k - 1][0] + log[k - 1][1], t[i], rel_tol=1e-14)
