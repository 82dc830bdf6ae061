"""Device-resident bundle-adjustment state and the Gauss-Newton loop over it.

``DeviceProblem`` keeps cameras, points, observations, Jacobian blocks and the
reduced camera system in HBM and drives the kernels of ``libsfm_ba.so`` through
the C ABI; an iteration performs no host round trip except the scalar cost the
reference's exit rule needs.  PyTorch is used for device memory, streams and
``torch.distributed`` only.

Loop being restated: ``optimize_using_BA`` of
``/root/reference/src/bundle_adjustment.py:70-105`` (constants mu = 0.5, step
scales 1.0 / 0.1 / 0.5, exit at cost <= 100 or 50 iterations).
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np
import torch

from . import _lib

F64 = torch.float64
I32 = torch.int32


def _dev(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X/ROCm device visible: this package computes on the GPU only")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plan:
    """Owner of a ``sfm_ba_plan`` handle (index structures of one visibility pattern).

    ``cam_idx`` / ``pt_idx``: int32 index lists, either host arrays or tensors on the GPU.  With
    ``where="device"`` (default) the plan is built on the GPU from the device copies; with
    ``where="host"`` by the host builder, which then uploads it.  Both produce the same arrays.
    """

    FIELDS = ("pt_ptr", "perm", "slot_cam", "tp_ptr", "pair_entry", "grp_pt", "cam_obs", "item_cam", "item_beg",
              "item_end", "cam_item_ptr", "blk_j", "blk_k", "entries", "pit_blk", "pit_beg", "pit_end", "pit_part",
              "mb_blk", "mb_part", "mb_cnt", "run_pt0", "run_key", "slot_obs", "key_cam", "key_run", "rng_run", "fz_pt",
              "run_cam", "fb_ptr", "fb_ent", "fb_far_part", "fb_far_cnt", "far_pt", "far_obs", "far_cam", "far_lpt",
              "far_gpt", "ent_pt")
    # the arrays of the fused assembly (21..28: one set per class of near points) and of the near / far split
    CLASS_FIELDS = FIELDS[21:29]
    FUSED_FIELDS = FIELDS[21:]

    def __init__(self, cam_idx, pt_idx, n_cams_hint=0, n_pts_hint=0, check=True, where="device", _borrowed=None):
        lib = _lib.load()
        if _borrowed is not None:
            # the far plan of another plan: owned by that plan, which this object keeps alive
            self._lib = lib
            self._owner, self.handle = _borrowed
            self._read_sizes()
            return
        self._owner = None
        dev = _dev()
        handle = C.c_void_p()
        if where == "host":
            if torch.is_tensor(cam_idx):
                cam_idx, pt_idx = cam_idx.cpu().numpy(), pt_idx.cpu().numpy()
            cam = np.ascontiguousarray(cam_idx, dtype=np.int32)
            pt = np.ascontiguousarray(pt_idx, dtype=np.int32)
            assert cam.shape == pt.shape and cam.ndim == 1
            rc = lib.sfm_ba_plan_create(
                cam.ctypes.data_as(C.c_void_p), pt.ctypes.data_as(C.c_void_p), cam.shape[0],
                int(n_cams_hint), int(n_pts_hint), int(bool(check)), C.byref(handle))
        elif where == "device":
            if not torch.is_tensor(cam_idx):
                cam_idx = torch.from_numpy(np.ascontiguousarray(cam_idx, dtype=np.int32)).to(dev)
                pt_idx = torch.from_numpy(np.ascontiguousarray(pt_idx, dtype=np.int32)).to(dev)
            assert cam_idx.is_cuda and pt_idx.is_cuda and cam_idx.dtype == I32 and pt_idx.dtype == I32
            assert cam_idx.is_contiguous() and pt_idx.is_contiguous()
            assert cam_idx.shape == pt_idx.shape and cam_idx.dim() == 1
            rc = lib.sfm_ba_plan_create_device(
                _lib.ptr(cam_idx), _lib.ptr(pt_idx), cam_idx.shape[0], int(n_cams_hint), int(n_pts_hint),
                int(bool(check)), _stream(), C.byref(handle))
        else:
            raise ValueError("where must be 'device' or 'host'")
        _lib.check(rc)
        self._lib = lib
        self.handle = handle
        self._read_sizes()

    def _read_sizes(self):
        lib, handle = self._lib, self.handle
        sizes = (C.c_int64 * 8)()
        _lib.check(lib.sfm_ba_plan_sizes(handle, sizes))
        (self.m, self.n, self.n_obs, self.n_blocks, self.n_entries, self.n_pair_slots,
         self.n_items, self.device_bytes) = [int(v) for v in sizes]
        self.workspace_bytes = int(lib.sfm_ba_plan_workspace_bytes(handle))
        self.half_bandwidth = int(lib.sfm_ba_plan_half_bandwidth(handle))
        info = (C.c_int32 * 16)()
        _lib.check(lib.sfm_ba_plan_fused_info(handle, info))
        # fused point-major assembly over the near points, per class (0: windows of up to 8 cameras,
        # 1: windows of 16); no runs: the plan has no such path
        names = ("on", "slots", "n_near", "pmax", "n_runs", "n_keys", "n_steps", "n_ranges")
        self.classes = [dict(zip(names, [int(v) for v in info[8 * c: 8 * c + 8]])) for c in range(2)]
        c0 = self.classes[0]
        self.fused = int(any(c["on"] for c in self.classes))
        self.fused_slots, self.fused_pmax = c0["slots"], c0["pmax"]
        self.n_near = sum(c["n_near"] for c in self.classes)
        self.n_runs = sum(c["n_runs"] for c in self.classes)
        self.n_keys = sum(c["n_keys"] for c in self.classes)
        self.n_steps = sum(c["n_steps"] for c in self.classes)
        self.n_ranges = sum(c["n_ranges"] for c in self.classes)
        far = lib.sfm_ba_plan_far(handle)
        # far points: a plan of their own (camera-pair kernels), None when every point is near
        self.far = Plan(None, None, _borrowed=(self, C.c_void_p(far))) if far else None
        self.n_far = self.far.n if self.far is not None else (self.n - self.n_near if self.n_runs > 0 else 0)
        self.n_far_obs = self.far.n_obs if self.far is not None else 0

    def set_fused(self, enable):
        """Turn the fused point-major assembly of this plan on or off (on by default where runs exist)."""
        _lib.check(self._lib.sfm_ba_plan_set_fused(self.handle, int(bool(enable))))
        self.fused = int(bool(enable) and self.n_runs > 0)
        for c in self.classes:
            c["on"] = int(bool(enable) and c["n_runs"] > 0)

    def read(self, field, cls=0):
        """One index array of the plan as a host array (``entries``: uint32 (n, 2)).  ``cls``: class of
        near points (0 / 1) for the arrays of the fused assembly that exist per class."""
        idx = self.FIELDS.index(field) + (100 if cls else 0)
        nbytes = C.c_int64()
        _lib.check(self._lib.sfm_ba_plan_read(self.handle, idx, None, 0, C.byref(nbytes)))
        out = np.empty(nbytes.value // 4, dtype=np.uint32 if field == "entries" else np.int32)
        _lib.check(self._lib.sfm_ba_plan_read(self.handle, idx, out.ctypes.data_as(C.c_void_p), out.nbytes,
                                              C.byref(nbytes)))
        return out.reshape(-1, 2) if field == "entries" else out

    def __del__(self):
        h = getattr(self, "handle", None)
        if getattr(self, "_owner", None) is not None:
            return
        if h is not None and h.value:
            self._lib.sfm_ba_plan_destroy(h)
            self.handle = None


class DeviceProblem:
    """SoA BA problem in HBM.

    Parameters (host numpy arrays): ``K`` (3,3) or (m,3,3); ``R`` (m,3,3); ``T`` (m,3);
    ``X`` (n,3); ``cam_idx``/``pt_idx`` (N,) in caller order; ``x_true`` (N,2).
    ``n_cams``: global camera count (a shard may not observe every camera).
    ``group``: a ``torch.distributed`` process group when the points are sharded
    over ranks; the camera system is then summed over the group every iteration.
    """

    MU = 0.5       # src/bundle_adjustment.py:79
    Q_STEP = 1.0   # :88
    C_STEP = 0.1   # :94
    X_STEP = 0.5   # :100

    def __init__(self, K, R, T, X, cam_idx, pt_idx, x_true, n_cams=None, group=None, check=True, device=None,
                 sharded=None, plan_where="device", state=None, reorder=False, cam_order=None):
        self.lib = _lib.load()
        # reorder: when S is not a narrow band in the order given, look for a camera order in which
        # it is narrower (ordering.narrowest_order: a sequence that closes a loop) and keep the cameras
        # in that order on the device.  cameras() / set_state() translate; S, delta_a and everything
        # per observation are then in the device's order (cam_order[i] = caller's index of camera i).
        # cam_order: that order, given by the caller (the ranks of a sharded run must agree on it:
        # ordering.narrowest_order on the unsharded lists, before the points are dealt out).
        self.cam_order = None
        self.cam_pos = None
        self.natural_half_bandwidth = None
        self.device = _dev(device)
        # state: form the rows of A inside the assembly and back-substitution kernels instead of
        # writing A once and reading it twice (plans with runs; default on, SFM_BA_STATE=0 turns it off)
        self._want_state = (os.environ.get("SFM_BA_STATE", "1") != "0") if state is None else bool(state)
        self.group = group
        self.world = 1
        if group is not None:
            import torch.distributed as dist

            self.world = dist.get_world_size(group)
        # sharded flow: damping left out of the local U, system summed over the group, damping
        # added once afterwards (forced on for a 1-rank group by sharded=True, for testing)
        self.sharded = (self.world > 1) if sharded is None else bool(sharded)
        R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
        self.m = int(R.shape[0]) if n_cams is None else int(n_cams)
        assert R.shape[0] == self.m
        X = np.asarray(X, dtype=np.float64).reshape(-1, 3)
        self.n = int(X.shape[0])
        cam_idx = np.ascontiguousarray(cam_idx, dtype=np.int32)
        pt_idx = np.ascontiguousarray(pt_idx, dtype=np.int32)
        self.N = int(cam_idx.shape[0])
        K = np.asarray(K, dtype=np.float64)
        T = np.asarray(T, dtype=np.float64).reshape(-1, 3)
        x_true = np.ascontiguousarray(x_true, dtype=np.float64).reshape(-1, 2)
        order = pos = None
        if cam_order is not None:
            order = np.asarray(cam_order, dtype=np.int64)
            assert order.shape == (self.m,) and np.array_equal(np.sort(order), np.arange(self.m))
            pos = np.empty(self.m, dtype=np.int64)
            pos[order] = np.arange(self.m)
        elif reorder and group is None and self.N > 0:
            from . import ordering

            hb_nat = ordering.half_bandwidth(np.arange(self.m), cam_idx, pt_idx, self.n)
            self.natural_half_bandwidth = hb_nat
            if hb_nat > 64:
                hb_new, pos_new, order_new = ordering.narrowest_order(cam_idx, pt_idx, self.m, self.n, hb_nat)
                # worth it when the band halves: up to 256 it reaches a band solver, above that the
                # general factorisation still skips the zero tiles outside it
                if pos_new is not None and 2 * hb_new <= hb_nat:
                    order, pos = order_new, pos_new
        if order is not None:
            from . import ordering

            cam_idx, pt_idx, x_true, _ = ordering.reorder_observations(pos, cam_idx, pt_idx, x_true)
            R, T = R[order], T[order]
            if K.ndim == 3:
                K = K[order]
            self.cam_order, self.cam_pos = order, pos
        sharded = self.sharded
        hinted = sharded or not check
        with torch.cuda.device(self.device):
            self.cam_idx = torch.from_numpy(cam_idx).to(self.device)
            self.pt_idx = torch.from_numpy(pt_idx).to(self.device)
            self.plan = Plan(self.cam_idx if plan_where == "device" else cam_idx,
                             self.pt_idx if plan_where == "device" else pt_idx, n_cams_hint=self.m if hinted else 0,
                             n_pts_hint=self.n if hinted else 0, check=check, where=plan_where)
        if self.plan.m != self.m or self.plan.n != self.n:
            # the reference's Indices asserts that every id up to the maximum occurs
            raise AssertionError("index lists do not cover every camera / point")
        # band of the reduced camera matrix: the widest over all shards
        self.half_bandwidth = self.plan.half_bandwidth
        # rows of A formed inside the kernels that use them (plans with runs)
        self.from_state = self._want_state and bool(self.lib.sfm_ba_plan_state_path(self.plan.handle))
        if self.sharded and self.world > 1:
            import torch.distributed as dist

            hb = torch.tensor([self.half_bandwidth], dtype=torch.int64,
                              device=self.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(hb, op=dist.ReduceOp.MAX, group=group)
            self.half_bandwidth = int(hb.item())
        d = self.device
        if K.ndim == 2:
            self.K = torch.from_numpy(K.reshape(1, 9).copy()).to(d)
            self.k_stride = 0
        else:
            self.K = torch.from_numpy(K.reshape(-1, 9).copy()).to(d)
            self.k_stride = 9
            assert self.K.shape[0] == self.m
        self.R = torch.from_numpy(R.reshape(-1, 9).copy()).to(d)
        self.T = torch.from_numpy(T.copy()).to(d)
        Cc = -np.einsum("mji,mj->mi", R, T)
        self.C = torch.from_numpy(Cc).to(d)
        Kb = np.broadcast_to(K, (self.m, 3, 3)) if K.ndim == 2 else K
        P = Kb @ np.concatenate([R, T[:, :, None]], axis=2)
        self.P = torch.from_numpy(np.ascontiguousarray(P).reshape(-1, 12)).to(d)
        self.X = torch.from_numpy(X.copy()).to(d)
        self.x_true = torch.from_numpy(np.ascontiguousarray(x_true, dtype=np.float64).reshape(-1, 2)).to(d)
        self._alloc()

    def _all_reduce(self, t):
        """Sum a device buffer over the group, in place.  RCCL reduces it where it lies; any other
        backend (gloo: CPU tests, rehearsals with several ranks on one GPU) gets a host copy."""
        import torch.distributed as dist

        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(t, group=self.group)
        else:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)

    # -- state that the reference keeps on frames / map ---------------------------------------
    def set_state(self, C=None, P=None):
        """Override the derived camera centre / projection (frames may carry their own)."""
        o = slice(None) if self.cam_order is None else self.cam_order
        if C is not None:
            self.C.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(C, dtype=np.float64).reshape(-1, 3)[o])))
        if P is not None:
            self.P.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 12)[o])))
        self._blk_fresh = False
        self._sys_fresh = False

    def _alloc(self):
        d, m, n, N = self.device, self.m, self.n, self.N
        n7 = 7 * m
        self.n7 = n7
        self.A = torch.empty((N, 14), dtype=F64, device=d)
        self.B = torch.empty((N, 6), dtype=F64, device=d)
        self.x_pred = torch.empty((N, 2), dtype=F64, device=d)
        self.cam_blk = torch.empty((m, _lib.CAM_BLK), dtype=F64, device=d)
        self.flags = torch.zeros((m,), dtype=I32, device=d)
        self.upd_flags = torch.zeros((m,), dtype=I32, device=d)
        # reduced system [(n7+1) x n7] followed by the per-camera squared-residual sums:
        # one contiguous buffer so that a sharded run reduces it with a single collective
        self.sys = torch.zeros(((n7 + 1) * n7 + m,), dtype=F64, device=d)
        self.S = self.sys[: (n7 + 1) * n7].view(n7 + 1, n7)
        self.ssq = self.sys[(n7 + 1) * n7:]
        self.e = torch.empty((m, 7), dtype=F64, device=d)
        self.U = torch.empty((m, 49), dtype=F64, device=d)
        self.eps_a = torch.empty((m, 7), dtype=F64, device=d)
        self.V_inv = torch.empty((n, 9), dtype=F64, device=d)
        self.eps_b = torch.empty((n, 3), dtype=F64, device=d)
        self.delta_a = torch.zeros((m, 7), dtype=F64, device=d)
        self.delta_b = torch.zeros((n, 3), dtype=F64, device=d)
        self.workspace = torch.empty((max(self.plan.workspace_bytes, 256) // 8,), dtype=F64, device=d)
        self.chol_ws = torch.empty((int(self.lib.sfm_ba_cholesky_workspace_bytes(n7)) // 8 + 8,), dtype=F64, device=d)
        self.info = torch.zeros((1,), dtype=I32, device=d)
        # sharded run with a narrow band: what the ranks all-reduce (see assemble)
        self.band_sharded = self.sharded and self.half_bandwidth <= 64
        self._packed_mu = None
        self._s_holds_foreign = False
        self._blk_fresh = False  # cam_blk holds the records of the current R, C, K
        self._sys_fresh = False  # S, e, V_inv, eps_b and cost_buf are those of the current state (and of _sys_mu)
        self._sys_mu = None
        if self.band_sharded:
            self.packed = torch.empty((int(self.lib.sfm_ba_band_packed_elements(n7)),), dtype=F64, device=d)
            self.compact = torch.empty((int(self.lib.sfm_ba_band_compact_elements(n7, self.half_bandwidth)),),
                                       dtype=F64, device=d)
        self.tri = None  # dense sharded exchange buffer, allocated by the first such exchange
        self.cost_buf = torch.zeros((1,), dtype=F64, device=d)
        self.status_buf = torch.zeros((3,), dtype=F64, device=d)

    # -- single stages ---------------------------------------------------------------------
    def camera_prepare(self):
        p = _lib.ptr
        self._sys_fresh = False
        _lib.check(self.lib.sfm_ba_camera_prepare(p(self.R), p(self.C), p(self.K), self.k_stride, self.m,
                                                  p(self.cam_blk), p(self.flags), _stream()))
        self._blk_fresh = True

    def jacobian(self, want_pred=True, want_B=True):
        """``want_B=False``: B is not written; assemble / back_substitute called with ``B=False``
        take it from A (dF/dX = -dF/dC)."""
        p = _lib.ptr
        _lib.check(self.lib.sfm_ba_jacobian(p(self.cam_blk), p(self.P), p(self.X), p(self.cam_idx), p(self.pt_idx),
                                            self.N, p(self.A), p(self.B) if want_B else None,
                                            p(self.x_pred) if want_pred else None, _stream()))

    def predict(self):
        p = _lib.ptr
        _lib.check(self.lib.sfm_ba_jacobian(p(self.cam_blk), p(self.P), p(self.X), p(self.cam_idx), p(self.pt_idx),
                                            self.N, None, None, p(self.x_pred), _stream()))

    def _prepare_system(self, mu, reduce_packed):
        """Before an assembly: the damping that goes into U, and S cleared where the kernels will not
        write while a solver or an earlier exchange may have left something."""
        n7 = self.n7
        # ``reduce_packed`` (sharded run, narrow band): the ranks exchange the packed block-tridiagonal
        # form instead of the dense buffer; S then keeps this rank's sums only and solve() works
        # from ``self.packed``.
        self._packed_mu = float(mu) if reduce_packed else None
        if reduce_packed and self._s_holds_foreign:
            # an earlier dense exchange left other ranks' sums in blocks this rank never writes
            self.sys[: (n7 + 1) * n7].zero_()
            self._s_holds_foreign = False
        if (self.plan.n_blocks < self.m * (self.m + 1) // 2 and not reduce_packed
                and (self.sharded or self.half_bandwidth > 64)):
            # camera pairs without a shared point are not written by the kernels; the previous
            # factorisation left fill-in there (the narrow-band solve only reads S), and the
            # all-reduce of a sharded run left the other ranks' blocks there
            # (row n7, the right-hand side, included: a shard need not observe every camera)
            self.sys[: (n7 + 1) * n7].zero_()
        return 0.0 if self.sharded else mu

    def _exchange_system(self, mu, reduce_packed):
        """After an assembly of a sharded run: the ranks' camera systems added up (src/sparse_ba.py:189-271
        are sums over points), the damping of U entering once."""
        p = _lib.ptr
        n7 = self.n7
        if reduce_packed:
            # blocks and right-hand-side entries this rank never writes stay zero from allocation
            # (only the entries a band of this width can hold travel: about half of the packed form)
            _lib.check(self.lib.sfm_ba_band_compact(p(self.S), n7, self.half_bandwidth, p(self.compact), _stream()))
            self._all_reduce(self.compact)
            _lib.check(self.lib.sfm_ba_band_expand(p(self.compact), n7, self.half_bandwidth, p(self.packed), _stream()))
        elif self.sharded:
            # dense S: the lower triangle and the right-hand side travel (the solvers read nothing
            # else), half of the buffer; the damping enters on the way back
            self._s_holds_foreign = True
            if self.tri is None:
                self.tri = torch.empty((int(self.lib.sfm_ba_lower_elements(n7)),), dtype=F64, device=self.device)
            _lib.check(self.lib.sfm_ba_lower_pack(p(self.S), n7, p(self.tri), _stream()))
            self._all_reduce(self.tri)
            _lib.check(self.lib.sfm_ba_lower_unpack(p(self.tri), n7, float(mu), p(self.S), _stream()))
            self.e.copy_(self.S[n7].view(self.m, 7))

    def assemble_state(self, mu, want_cost=False):  # noqa: D401
        """Jacobian blocks and Schur assembly in one pass (plans with a fused path): the rows of A of the
        near points are formed inside the assembly kernel from the camera records and the points and
        never written; the far points go through the camera-pair kernels inside the same call.
        ``want_cost``: the per-camera sums of squared residuals of this state come out of the same
        pass (``ssq``), and ``cost_buf`` receives the cost (src/bundle_adjustment.py:9-19)."""
        p = _lib.ptr
        self._sys_fresh = False
        mu_u = self._prepare_system(mu, self.band_sharded)
        if want_cost and self.sharded:
            # the kernels write the sums of the cameras this rank observes; the others must not keep
            # the totals of the previous all-reduce
            self.ssq.zero_()
        _lib.check(self.lib.sfm_ba_schur_assemble_state(
            self.plan.handle, p(self.cam_blk), p(self.P), p(self.X), p(self.x_true), float(mu), float(mu_u),
            p(self.S), p(self.e), p(self.V_inv), p(self.eps_b), p(self.ssq) if want_cost else None,
            p(self.workspace), _stream()))
        self._exchange_system(mu, self.band_sharded)
        if want_cost:
            if self.sharded:
                self._all_reduce(self.ssq)
            _lib.check(self.lib.sfm_ba_cost_finish(p(self.ssq), self.m, p(self.cost_buf), _stream()))

    def assemble_pairs_state(self, mu, want_cost=False):
        """Schur assembly of any plan inside the loop: ``self.A`` holds the rows the Jacobian kernel wrote
        from the current state (the per-point stage and the back-substitution read them), and the
        camera-pair products are formed from the state inside their kernel instead of being gathered.
        ``want_cost`` as in ``assemble_state``."""
        p = _lib.ptr
        self._sys_fresh = False
        mu_u = self._prepare_system(mu, self.band_sharded)
        if want_cost and self.sharded:
            self.ssq.zero_()
        _lib.check(self.lib.sfm_ba_schur_assemble_pairs_state(
            self.plan.handle, p(self.A), p(self.cam_blk), p(self.X), p(self.x_true), p(self.x_pred), float(mu),
            float(mu_u), p(self.S), p(self.e), p(self.V_inv), p(self.eps_b), p(self.ssq) if want_cost else None,
            p(self.workspace), _stream()))
        self._exchange_system(mu, self.band_sharded)
        if want_cost:
            if self.sharded:
                self._all_reduce(self.ssq)
            _lib.check(self.lib.sfm_ba_cost_finish(p(self.ssq), self.m, p(self.cost_buf), _stream()))

    def back_substitute_state(self, update_points=False):
        """delta_b (and the point update) with the rows of A formed again from the state (the far
        points' rows are still in the workspace of the assembly)."""
        p = _lib.ptr
        self._sys_fresh = self._sys_fresh and not update_points
        _lib.check(self.lib.sfm_ba_back_substitute_state(
            self.plan.handle, p(self.cam_blk), p(self.V_inv), p(self.eps_b), p(self.delta_a), p(self.delta_b),
            p(self.X), 1 if update_points else 0, float(self.X_STEP), p(self.workspace), _stream()))

    def assemble(self, mu, weights=None, x_true=None, x_pred=None, A=None, B=None, by_products=True,
                 reduce_packed=False):
        """``by_products=False`` skips U and eps_a (the solve needs only S and e): the library then
        folds the per-camera sums into the camera-pair kernel and makes one pass less over A."""
        p = _lib.ptr
        self._sys_fresh = False
        reduce_packed = bool(reduce_packed) and self.band_sharded
        mu_u = self._prepare_system(mu, reduce_packed)
        _lib.check(self.lib.sfm_ba_schur_assemble(
            self.plan.handle, p(self.A if A is None else A),
            None if B is False else p(self.B if B is None else B),
            p(self.x_true if x_true is None else x_true), p(self.x_pred if x_pred is None else x_pred),
            p(weights), float(mu), float(mu_u), p(self.S), p(self.e), p(self.U) if by_products else None,
            p(self.eps_a) if by_products else None,
            p(self.V_inv), p(self.eps_b), None, p(self.workspace), _stream()))
        self._exchange_system(mu, reduce_packed)

    def solve(self):
        p = _lib.ptr
        self._sys_fresh = False  # (the general factorisation works in place)
        if self._packed_mu is not None:
            _lib.check(self.lib.sfm_ba_solve_band_packed(p(self.packed), self.n7, self._packed_mu, p(self.delta_a),
                                                         p(self.chol_ws), p(self.info), _stream()))
            return
        _lib.check(self.lib.sfm_ba_solve_reduced_banded(p(self.S), self.n7, self.half_bandwidth, p(self.delta_a),
                                                        p(self.chol_ws), p(self.info), _stream()))

    def back_substitute(self, weights=None, update_points=False, A=None, B=None):
        p = _lib.ptr
        self._sys_fresh = self._sys_fresh and not update_points
        _lib.check(self.lib.sfm_ba_back_substitute(
            self.plan.handle, p(self.A if A is None else A), None if B is False else p(self.B if B is None else B),
            p(weights), p(self.V_inv), p(self.eps_b), p(self.delta_a), p(self.delta_b),
            p(self.X) if update_points else None, float(self.X_STEP), _stream()))

    def update_cameras(self, prepare_next=False):
        """``prepare_next``: also leave the prepared camera records of the new state in ``cam_blk``
        (what ``camera_prepare`` would compute), so that the next iteration skips that launch."""
        p = _lib.ptr
        self._sys_fresh = False
        if self.cam_pos is not None:
            # the camera the loop holds is the caller's camera 0, wherever the device order put it
            _lib.check(self.lib.sfm_ba_update_cameras_held(
                p(self.R), p(self.T), p(self.C), p(self.P), p(self.K), self.k_stride, p(self.delta_a), self.m,
                int(self.cam_pos[0]), float(self.Q_STEP), float(self.C_STEP), p(self.upd_flags),
                p(self.cam_blk) if prepare_next else None, p(self.flags) if prepare_next else None, _stream()))
            self._blk_fresh = bool(prepare_next)
            return
        if prepare_next:
            _lib.check(self.lib.sfm_ba_update_cameras_prepare(
                p(self.R), p(self.T), p(self.C), p(self.P), p(self.K), self.k_stride, p(self.delta_a), self.m, 1,
                float(self.Q_STEP), float(self.C_STEP), p(self.upd_flags), p(self.cam_blk), p(self.flags), _stream()))
            self._blk_fresh = True
            return
        _lib.check(self.lib.sfm_ba_update_cameras(p(self.R), p(self.T), p(self.C), p(self.P), p(self.K),
                                                  self.k_stride, p(self.delta_a), self.m, 1, float(self.Q_STEP),
                                                  float(self.C_STEP), p(self.upd_flags), _stream()))
        self._blk_fresh = False

    def cost_async(self):
        """Enqueue the cost of the current state into ``cost_buf`` (no synchronisation)."""
        p = _lib.ptr
        if not self.sharded:
            _lib.check(self.lib.sfm_ba_cost(self.plan.handle, p(self.P), p(self.X), p(self.cam_idx), p(self.pt_idx),
                                            p(self.x_true), p(self.ssq), p(self.cost_buf), p(self.workspace),
                                            _stream()))
            return
        _lib.check(self.lib.sfm_ba_cost_partial(self.plan.handle, p(self.P), p(self.X), p(self.cam_idx),
                                                p(self.pt_idx), p(self.x_true), p(self.ssq), p(self.workspace),
                                                _stream()))
        if self.sharded:
            import torch.distributed as dist

            self._all_reduce(self.ssq)
        _lib.check(self.lib.sfm_ba_cost_finish(p(self.ssq), self.m, p(self.cost_buf), _stream()))

    def cost(self) -> float:
        self.cost_async()
        return float(self.cost_buf.item())

    # -- one Gauss-Newton iteration (src/bundle_adjustment.py:75-101) ---------------------------
    def iterate_async(self, mu=None):
        mu = self.MU if mu is None else mu
        if not self._blk_fresh:
            self.camera_prepare()
            self._sys_fresh = False
        if self.from_state and self.plan.fused and mu > 0:
            # plans with a fused path: the rows of A are formed inside the assembly and the
            # back-substitution, and the cost of a state is a by-product of its assembly: the
            # iteration ends with the assembly of the state it produced - the system the next
            # iteration solves, and the cost the loop looks at (src/bundle_adjustment.py:101)
            if not (self._sys_fresh and self._sys_mu == mu):
                self.assemble_state(mu, want_cost=True)
            self.solve()
            self.back_substitute_state(update_points=True)
            self.update_cameras(prepare_next=True)
            self.assemble_state(mu, want_cost=True)
            self._sys_fresh, self._sys_mu = True, mu
            return
        if self._want_state and mu > 0:
            # any other plan: the same shape of iteration on the camera-pair kernels, with the products
            # of a camera pair formed from the state and the cost out of the diagonal blocks
            if not (self._sys_fresh and self._sys_mu == mu):
                self.jacobian(want_B=False)
                self.assemble_pairs_state(mu, want_cost=True)
            self.solve()
            self.back_substitute(update_points=True, B=False)
            self.update_cameras(prepare_next=True)
            self.jacobian(want_B=False)
            self.assemble_pairs_state(mu, want_cost=True)
            self._sys_fresh, self._sys_mu = True, mu
            return
        else:
            self.jacobian(want_B=False)
            self.assemble(mu, by_products=False, reduce_packed=True, B=False)
            self.solve()
            self.back_substitute(update_points=True, B=False)
        self.update_cameras(prepare_next=True)
        self.cost_async()

    def status(self):
        """(cost of the last iteration, pivot info, OR of the camera flag words) with one read."""
        p = _lib.ptr
        _lib.check(self.lib.sfm_ba_iteration_status(p(self.cost_buf), p(self.info), p(self.flags), p(self.upd_flags),
                                                    self.m, p(self.status_buf), _stream()))
        c, info, f = self.status_buf.tolist()
        return c, int(info), int(f)

    def clear_flags(self):
        self.info.zero_()
        self.flags.zero_()
        self.upd_flags.zero_()

    def check_flags(self, status=None):
        """Raise what the reference would have raised during the iterations run since the flags were
        last cleared (the device words are sticky; an exception clears them)."""
        _, info, f = self.status() if status is None else status
        if info != 0 or f != 0:
            self.clear_flags()
        if info != 0:
            raise np.linalg.LinAlgError("Singular matrix")
        if f & _lib.FLAG_DET_NOT_ONE:
            raise AssertionError("Input matrix must have determinant 1 (be a rotation matrix)")
        if f & _lib.FLAG_ZERO_QUAT:
            raise ValueError("Cannot normalize quaternion with zero norm")
        if f & _lib.FLAG_NONFINITE:
            raise FloatingPointError("non-finite camera parameters")

    def run(self, max_iter=50, threshold=100.0, min_iter=0, callback=None):
        """The reference's loop: iterate while cost > threshold and fewer than
        ``max_iter`` iterations were run.  Returns the per-iteration costs."""
        costs = []
        avg_error = 1e10
        n_iter = 0
        while (avg_error > threshold or n_iter < min_iter) and n_iter < max_iter:
            self.iterate_async()
            st = self.status()   # the iteration's one host synchronisation
            avg_error = st[0]
            self.check_flags(st)
            costs.append(avg_error)
            if callback is not None:
                callback(n_iter, avg_error)
            n_iter += 1
        return costs

    # -- results back to the host --------------------------------------------------------------
    def cameras(self):
        """(R, T, C, P) of every camera, in the caller's camera order."""
        R = self.R.cpu().numpy().reshape(-1, 3, 3)
        T = self.T.cpu().numpy().reshape(-1, 3)
        Cc = self.C.cpu().numpy().reshape(-1, 3)
        P = self.P.cpu().numpy().reshape(-1, 3, 4)
        if self.cam_pos is not None:
            R, T, Cc, P = R[self.cam_pos], T[self.cam_pos], Cc[self.cam_pos], P[self.cam_pos]
        return R, T, Cc, P

    def points(self):
        return self.X.cpu().numpy()
