"""TSTrajectory (pa.py:771-772): checkpoint slabs in HBM, the disk tier (-ts_trajectory_type basic, csrc/pn_spill.cpp)
and the two-level form; the slot PLAN is the C++ scheduler's (pn_traj_*), these classes own the memory."""
import contextlib
import ctypes
import warnings

import torch
import torch.nn as nn  # noqa: F401

from . import _lib, options
from ._lib import PnError, check  # noqa: F401


_NO_STORES = {}


class _Trajectory(object):
    """HBM-resident checkpoint store: slots planned by the C++ scheduler (pn_traj_*), memory
    owned here as torch slabs.  A slot holds `vecs` state-sized vectors (1 = the state at the
    start of a step; s_eff = state + stage values in store-all mode), each padded to a
    multiple of 64 elements so every vector starts 256-byte aligned."""

    CHUNK_BYTES = 1 << 28

    def __init__(self, lib, ops, n, vecs, mode, max_slots):
        self.lib, self.ops, self.n, self.vecs = lib, ops, n, vecs
        self.npad = (n + 63) // 64 * 64
        self.handle = ctypes.c_void_p(lib.pn_traj_create())
        check(lib.pn_traj_begin(self.handle, mode, max_slots))
        if mode == _lib.PN_TRAJ_BUDGET and vecs > 1:
            check(lib.pn_traj_set_carry(self.handle, 1))     # slots hold stage values: place the checkpoints for that cost
        esize = 4 if ops.dtype == torch.float32 else 8
        slot_bytes = self.vecs * self.npad * esize
        if mode == _lib.PN_TRAJ_BUDGET:
            self.chunk_slots = max(1, int(max_slots))
        else:
            self.chunk_slots = max(1, min(64, self.CHUNK_BYTES // max(1, slot_bytes)))
        self.chunks = []
        self.plan_cap = max(64, min(int(max_slots), 4096)) if mode == _lib.PN_TRAJ_BUDGET else 64
        self._plan_buf = None
        self.stage_step = {}          # slot -> step whose stage values Y_1.. are stored behind the state

    def __del__(self):
        try:
            self.lib.pn_traj_destroy(self.handle)
        except Exception:
            pass

    def view(self, slot):
        """(vecs, npad) tensor of `slot`."""
        c, i = divmod(slot, self.chunk_slots)
        while c >= len(self.chunks):
            self.chunks.append(self.ops.empty(self.chunk_slots, self.vecs, self.npad))
        return self.chunks[c][i]

    def fwd_slot(self, step):
        return self.lib.pn_traj_fwd_slot(self.handle, step)

    def rev_plan(self, step, cap=0):
        cap = cap or self.plan_cap
        buf = self._plan_buf
        if buf is None or buf[0] != cap:
            buf = self._plan_buf = (cap, ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(),
                                    (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)())
        _, fs, fl, ns, ss, sl = buf
        check(self.lib.pn_traj_rev_plan(self.handle, step, ctypes.byref(fs), ctypes.byref(fl), ctypes.byref(ns),
                                        ss, sl, cap))
        return fs.value, fl.value, ({ss[k]: sl[k] for k in range(ns.value)} if ns.value else _NO_STORES)

    def rev_done(self, step):
        check(self.lib.pn_traj_rev_done(self.handle, step))

    def high_water(self):
        return self.lib.pn_traj_high_water(self.handle)

    # the HBM tier needs none of these (see _DiskTrajectory)
    def claim(self, slot):
        """`slot` is about to be (re)written with a new checkpoint: its buffer, whatever it held."""
        return self.view(slot)

    def seal(self, slot):
        pass

    def begin_reverse(self):
        pass

    on_disk = False


class _DiskTrajectory(_Trajectory):
    """``-ts_trajectory_type basic`` -- PETSc's default trajectory type, the one the reference runs with
    unless ``-ts_trajectory_type memory`` is given (examples-pnode/ode_demo_petsc.py:26): every checkpoint
    is a file under ``-ts_trajectory_dirname``.  The device keeps RING checkpoint buffers as a small
    least-recently-used cache; a checkpoint leaves for its file as soon as it is complete (``seal``:
    asynchronous copy + background write, pn_spill_put) and comes back when a sweep asks for it (``view``:
    pn_spill_get), the one before it being read ahead in the reverse sweep.  Works for every placement of
    the checkpoints -- every step, or the bounded set of ``-ts_trajectory_max_cps_ram`` whose slots are
    recycled (``claim``) -- with the same slots, kernels and results as the HBM tier."""

    RING = 4             # at most three buffers are in use at once (source and destination of a step, stage values)
    STAGING = 6          # pinned staging buffers (and STAGING/2 I/O threads)
    _seq = 0
    on_disk = True

    def __init__(self, lib, ops, n, vecs, mode, max_slots, dirname, keep_files):
        import os
        _Trajectory.__init__(self, lib, ops, n, vecs, mode, max_slots)
        self.ring = ops.empty(self.RING, vecs, self.npad)
        self.holds = [-1] * self.RING
        self.stamp = [0] * self.RING
        self.clock = 0
        self.sealed = set()
        self.reverse = False
        esize = 4 if ops.dtype == torch.float32 else 8
        os.makedirs(dirname, exist_ok=True)
        _DiskTrajectory._seq += 1
        self.dir = os.path.join(dirname, "pn-%d-%d" % (os.getpid(), _DiskTrajectory._seq))
        self.spill = ctypes.c_void_p(lib.pn_spill_create(self.dir.encode(), vecs * self.npad * esize, self.STAGING,
                                                         1 if ops.device.type == "cuda" else 0, 1 if keep_files else 0))
        if not self.spill:
            raise PnError(lib.pn_last_error().decode())

    def __del__(self):
        try:
            if self.spill:
                self.lib.pn_spill_destroy(self.spill)
        except Exception:
            pass
        if _Trajectory is not None:                # (None while the interpreter shuts down: module globals go first)
            _Trajectory.__del__(self)

    def _stream(self):
        return self.ops.stream() if hasattr(self.ops, "stream") else None

    def _buffer(self, slot, load):
        self.clock += 1
        if slot in self.holds:
            r = self.holds.index(slot)
        else:
            r = min(range(self.RING), key=lambda k: self.stamp[k])      # least recently used (complete checkpoints are
            if load and slot in self.sealed:                             # in their files already: nothing to write back)
                check(self.lib.pn_spill_get(self.spill, self._stream(), slot, self.ring[r].data_ptr()))
            self.holds[r] = slot
        self.stamp[r] = self.clock
        return r

    def view(self, slot):
        r = self._buffer(slot, True)
        if self.reverse:                                 # the sweep walks backwards: read the one before it ahead
            prev = slot - 1
            if prev >= 0 and prev in self.sealed and prev not in self.holds:
                check(self.lib.pn_spill_prefetch(self.spill, prev))
        return self.ring[r]

    def claim(self, slot):
        self.sealed.discard(slot)                        # what the file holds belongs to the checkpoint that had this slot before
        return self.ring[self._buffer(slot, False)]

    def seal(self, slot):
        """The checkpoint in `slot` is complete (again): off to its file.  Called after every change of a slot's contents."""
        if slot >= 0:
            if slot not in self.holds:
                raise PnError("trajectory disk tier: slot %d sealed without being resident" % slot)
            check(self.lib.pn_spill_put(self.spill, self._stream(), slot, self.ring[self.holds.index(slot)].data_ptr()))
            self.sealed.add(slot)

    def begin_reverse(self):
        self.reverse = True

    def stats(self):
        f, w, r, wt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self.lib.pn_spill_stats(self.spill, ctypes.byref(f), ctypes.byref(w), ctypes.byref(r), ctypes.byref(wt))
        return {"files": f.value, "bytes_written": w.value, "bytes_read": r.value, "waits": wt.value}


class _TwoLevelTrajectory(_DiskTrajectory):
    """``-ts_trajectory_max_cps_ram R`` together with ``-ts_trajectory_max_cps_disk D`` (PETSc's two-level checkpointing,
    /root/reference/README.md:91-96): a bounded set of R + D checkpoints placed by the same scheduler, the first R slots in
    HBM, the other D in files behind the four-buffer device cache of the disk tier.  Same slots, kernels and results as
    a budget of R + D in HBM."""

    def __init__(self, lib, ops, n, vecs, mode, max_slots, dirname, keep_files, ram_slots):
        _DiskTrajectory.__init__(self, lib, ops, n, vecs, mode, max_slots, dirname, keep_files)
        self.ram = int(ram_slots)
        self.chunk_slots = max(1, self.ram)        # one HBM slab for the R resident slots (allocated on first use)

    def view(self, slot):
        return _Trajectory.view(self, slot) if slot < self.ram else _DiskTrajectory.view(self, slot)

    def claim(self, slot):
        return _Trajectory.view(self, slot) if slot < self.ram else _DiskTrajectory.claim(self, slot)

    def seal(self, slot):
        if slot >= self.ram:
            _DiskTrajectory.seal(self, slot)
