"""One-call forward path used by ``render_gaussians(backend="hip")``: ms_render_fwd.

The three stage calls of the reference orchestrator (mojosplat/render.py:63-101) collapse into
one library call; the projected tensors, tile ranges and the sorted intersection list stay in
caller-owned scratch that is cached per device and reused across frames (288 GB of HBM: the
scratch for 1M Gaussians at 1080p is ~100 MB and is never freed or re-sized unless a frame
needs more).  The per-stage entry points stay available for callers that want the
intermediates (project_gaussians / bin_gaussians_to_tiles / rasterize_gaussians).
"""
import ctypes
import math
import time
import os
import threading

import torch

from . import _hip
from .projection import EPS2D

LAZY_SORT = os.environ.get("MOJOSPLAT_LAZY_SORT", "1") != "0"  # mirrors csrc/pipeline.hip

class _LaneStates:
    """(device, lane) -> dict(ws, isect, host, ev, ...): the cached scratch of a lane.

    Lane 0 -- the blocking single-frame path behind render_gaussians and the differentiable frame -- is PER HOST
    THREAD (round 3): every thread that renders gets its own workspace, intersection buffer, pinned size record and
    event the first time it does, so host threads render concurrently on one device (their kernels interleave on
    whatever streams they use; 288 GB of HBM make a ~100 MB scratch set per thread a non-issue) and a thread's frame
    hints -- the previous frame's size record, its learnt sorting mode -- are its own.  The scratch dies with the
    thread.  Lanes >= 1 (multi-view batches, the asynchronous band path) are shared and driven by one thread at a
    time by design (`_frame_lock`)."""

    def __init__(self):
        self._shared = {}
        self._tls = threading.local()

    def _of(self, key):
        if key[1] != 0:
            return self._shared
        d = getattr(self._tls, "d", None)
        if d is None:
            d = self._tls.d = {}
        return d

    def get(self, key, default=None):
        return self._of(key).get(key, default)

    def __getitem__(self, key):
        return self._of(key)[key]

    def __setitem__(self, key, value):
        self._of(key)[key] = value

    def __contains__(self, key):
        return key in self._of(key)

    def clear(self):
        """Forget the shared lanes and the CALLING thread's lane 0 (tests start from no scratch this way)."""
        self._shared.clear()
        d = getattr(self._tls, "d", None)
        if d is not None:
            d.clear()


_state = _LaneStates()
# The shared lanes' scratch is one-frame-at-a-time: whole calls on them (multi-view batches) take this lock.
_frame_lock = threading.RLock()


def release_scratch():
    """Drop the cached scratch of the CALLING thread's blocking path (workspace + intersection buffer: ~0.1 GB at config
    3, ~1 GB at config 4) and of the shared lanes, back to torch's caching allocator.  For thread pools: a worker's scratch
    otherwise lives until the thread exits.  The next frame allocates afresh and starts without the previous frames' hints."""
    with _frame_lock:
        devs = set()
        for key, st in list(_state._shared.items()):
            if st.get("busy"):
                raise RuntimeError("release_scratch: a begun frame is still pending on a shared lane")
            devs.add(key[0])
        d = getattr(_state._tls, "d", None)
        if d:
            devs.update(k[0] for k in d)
        # (round 5, advisor: the buffers are used on lane streams the caching allocator does not know about -- a block handed
        # back without a sync could be re-issued while a lane's kernels still touch it; _grow() waits for the same reason)
        for dev in devs:
            torch.cuda.synchronize(dev)
        _state.clear()
        # (round 6, advisor: the scene caches pin whole scenes -- marshalled copies, prepared bounds -- on the device)
        from . import _band, scene_order
        _band.clear_scenes()
        scene_order.clear_registry()


def _dev_state(dev, lane=0):
    st = _state.get((dev, lane))
    if st is None:
        ev = torch.cuda.Event()
        with torch.cuda.device(dev):
            ev.record()  # materialises the hipEvent_t the library re-records for its size hand-off
        host = torch.zeros(16, dtype=torch.int64).pin_memory()   # (words 8..: the band pair's deferred clean-up verdict)
        # host_np: the same pinned memory as a numpy array (reading a torch tensor element costs ~1.5 us, and a
        # frame reads eight of them)
        st = dict(ws=None, isect=None, host=host, host_np=host.numpy(), ev=ev)
        _state[(dev, lane)] = st
    return st


def _grow(st, key, nbytes, dev, slack=1.0):
    buf = st[key]
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # kernels of earlier frames (possibly on a lane stream the allocator does not know this
            # buffer is used on) may still read the old block: let them drain before it is released
            torch.cuda.synchronize(dev)
        buf = torch.empty(int(nbytes * slack) + 256, dtype=torch.uint8, device=dev)
        st[key] = buf
        if key == "ws":
            st["shape"] = None   # fresh memory: the clean-up count the next frame reports is garbage
            if st.get("host_np") is not None:
                # ... and so are the per-bin depth cut-offs the library keeps in it: the record must not vouch for them
                # (bits 6-8 and 16-31 of its flag word; a frame that took them anyway would still be exact, only slow)
                st["host_np"][7] = int(st["host_np"][7]) & 0x3f
    return buf


# Optional frame statistics (bench.py's moving-camera leg): set to a dict and every finished frame adds to its
# counters -- how often a sync-free frame's bets were lost and what that cost.  None (the default): nothing is counted.
#   frames            frames finished
#   speculated        ... whose emit + rasterise were enqueued before the size record was known
#   redone_exact      ... of those, redone on the exact path (the speculation did not hold)
#   overflow          ... because the frame held more pairs than the intersection buffer had room for
#   light_bet_lost    ... because a frame that bet on "no heavy tile" (short sorts only) had one
#   other_miss        ... any other reason (a tile class that was not launched, an empty frame)
#   buffer_grown      frames that had to grow the intersection buffer (a subset of overflow / first frames)
#   redo_tiles        tiles the clean-up pass redid because their sorted front was too short (reported one frame late by
#                     the library; exact-path frames excluded)
#   cut_redo_tiles    ... because their depth cut-off was: their dropped pairs were regenerated (likewise)
#   depth_cut         frames that dropped the pairs behind their bins' depth cut-offs
#   regen_mismatch    frames whose clean-up launches brought back a pair the count kernel had not counted (a bug if ever > 0)
#   front_level_up / full_sort_on   the lane's lazy-sorting mode escalated
#   lazy_sort_retry   a lane on full sorts tried lazily sorted fronts again
FRAME_STATS = None


def _count_frame(stats, frame, host, grew, counts_valid=True):
    flags, heavy = int(host[7]), int(host[2]) + int(host[3]) + int(host[4])
    stats["frames"] = stats.get("frames", 0) + 1
    if grew:
        stats["buffer_grown"] = stats.get("buffer_grown", 0) + 1
    if flags & 1:
        stats["speculated"] = stats.get("speculated", 0) + 1
        if flags & 4:
            stats["redone_exact"] = stats.get("redone_exact", 0) + 1
            isect = frame.isect if frame.own else frame.st["isect"]
            nbytes = 0 if isect is None else isect.numel()
            # (bit 13: a differentiable frame keeps its quads' lists behind the ids -- 4 more bytes per entry and quad of a tile)
            per = 12 + (4 * 4 * max(1, frame.head[19] // 16) ** 2 if (flags & 8192) and hasattr(frame, "head") else 0)
            cap = (nbytes - 768) // 28 if flags & 8 else (nbytes - (768 if per > 12 else 512)) // per
            if grew or int(host[0]) > cap:
                why = "overflow"
            elif (flags & 32) and heavy > 0:
                why = "light_bet_lost"
            else:
                why = "other_miss"
            stats[why] = stats.get(why, 0) + 1
    # (host[5] = what the PREVIOUS frame on this workspace redid; the count lives at an offset that follows the grid, so the
    # first frame on another grid -- a bin size the rule has just switched to -- reads a word nobody has written)
    if not flags & 4 and counts_valid:
        stats["redo_tiles"] = stats.get("redo_tiles", 0) + (int(host[5]) & 0xffffffff)
        stats["cut_redo_tiles"] = stats.get("cut_redo_tiles", 0) + ((int(host[5]) >> 32) & 0x3fffffff)
        if (int(host[5]) >> 62) & 1:   # the clean-up launches of the previous frame disagreed with its count kernel (never seen)
            stats["regen_mismatch"] = stats.get("regen_mismatch", 0) + 1
    if flags & 64:
        stats["depth_cut"] = stats.get("depth_cut", 0) + 1
    if (flags & 4096) and not (flags & 4):   # the band pair left the clean-up launches to its finishing half (_band.py)
        stats["cleanup_deferred"] = stats.get("cleanup_deferred", 0) + 1
        if len(host) > 8 and int(host[8]) != 0:   # ... which found the rasteriser asking for them
            stats["cleanup_enqueued_late"] = stats.get("cleanup_enqueued_late", 0) + 1


WHOLE, RESUME, BEGIN, FINISH = 0, 1, 2, 3  # ms_render_fwd phases (include/mojosplat_hip.h)
RETRY_FULL_SORT = 64   # frames a lane stays on full sorts before it tries lazily sorted fronts again (doubling each time)
SWAP_FULL_SORT = 4     # ... after a scene swap (nearly every heavy bin regenerated at once): fixed, short
FULL_SORT = 0x100
FRONT_LEVEL = 0x200  # x level (0..3): deeper lazily sorted fronts
DEFER_CLEANUP = 0x1000  # the band pair only (_band.py): clean-up launches enqueued by the finishing half, if the rasteriser asks
ROWS16 = 0x800       # row_range counts rows of 16 px whatever the tile size (a band keeps its rows, the bins follow the scene)


# ---- what a DIFFERENTIABLE frame learns about its lazily sorted fronts (round 5, advisor) ----------------------------
# A frame on cached scratch is told by the NEXT frame's size record how many bins its clean-up pass had to redo, and the
# lane moves to deeper fronts or to full sorts (_Frame.finish).  A differentiable frame owns FRESH scratch, so that
# channel is garbage for it, and round 4 left such frames on the default fronts for ever: a scene whose fronts cannot
# saturate (low opacities after an opacity reset, fog) paid the clean-up pass and the backward's redo launch on every
# training step.  Now the backward asks the library for the count (ms_render_redo_counts: an 8-byte asynchronous copy into
# pinned memory behind ms_render_bwd), and the next differentiable frame of the same shape on this thread -- the first one
# that finds the copy complete: the host usually runs a step ahead of the GPU -- applies the lane's rule to it.
def own_mode(st, shape):
    """-> (full_sort, front_level) for the next lean differentiable frame of `shape` on this thread's lane."""
    pend = st.get("own_pending")
    if pend is not None and pend["ev"].query():
        st["own_pending"] = None
        redo = int(pend["buf"][0]) - int(pend["buf"][1])
        m = st.setdefault("own_learnt", {}).setdefault(pend["shape"], dict(full=False, level=0, frames=0, retry_after=RETRY_FULL_SORT))
        if FRAME_STATS is not None:
            FRAME_STATS["own_redo_tiles"] = FRAME_STATS.get("own_redo_tiles", 0) + max(redo, 0)
        if pend["lazy"] and not m["full"] and pend["level"] == m["level"]:
            if redo > 0:
                if redo > max(3, pend["heavy"] // 4) or m["level"] >= 2:
                    m["full"], m["frames"] = True, 0
                    if FRAME_STATS is not None:
                        FRAME_STATS["own_full_sort_on"] = FRAME_STATS.get("own_full_sort_on", 0) + 1
                else:
                    m["level"] += 1
                    if FRAME_STATS is not None:
                        FRAME_STATS["own_front_level_up"] = FRAME_STATS.get("own_front_level_up", 0) + 1
            else:
                m["retry_after"] = RETRY_FULL_SORT   # lazily sorted fronts hold: the next fall-back starts with short patience
    m = st.get("own_learnt", {}).get(shape)
    if m is None:
        return False, 0
    if m["full"]:
        m["frames"] += 1
        if m["frames"] > m["retry_after"]:   # a view through fog ends: try lazily sorted fronts again, with doubled patience
            m["full"], m["frames"] = False, 0
            m["retry_after"] = min(2 * m["retry_after"], 4096)
    return bool(m["full"]), int(m["level"])


def own_mirror(st):
    """The thread's pinned i32[2] that a differentiable frame's backward fills with its clean-up counts (ms_render_bwd_rows)."""
    buf = st.get("own_redo_buf")
    if buf is None:
        buf = st["own_redo_buf"] = torch.zeros(2, dtype=torch.int32).pin_memory()
        st["own_redo_np"] = buf.numpy()
        st["own_redo_ev"] = torch.cuda.Event()
    return buf


def own_report(st, shape, mode, level, heavy):
    """Behind a differentiable frame's backward (whose redo launch has written the frame's clean-up counts into own_mirror's
    words): an event on the current stream marks when they are valid; the next differentiable frame that finds it complete
    applies the lane's rule (own_mode)."""
    own_mirror(st)
    st["own_redo_ev"].record()
    st["own_pending"] = dict(ev=st["own_redo_ev"], buf=st["own_redo_np"], shape=shape, lazy=not (mode & FULL_SORT), level=level,
                             heavy=heavy)


def _after_frame(st, host, rc, grew, *, shape, level, mode, own, channels, frame=None):
    """What a finished frame teaches its lane (the size record `host` it left, the status `rc` of its finishing call): the
    lazy-sorting mode of the lane's next frames, whether they may speculate.  Shared by _Frame.finish and the band frames
    of _band.py."""
    # a frame whose tiles need the merge-fallback sort cannot run sync-free (the library redoes it
    # on the exact path); do not speculate on the next frame of such a scene.  Lazily sorted frames
    # (the library's default for plain forward frames of <= 4 channels) have no such tiles.
    # Lazily sorted fronts that turn out too short are made good by a clean-up pass that is slow by
    # design (one heavy bin costs more than lazy sorting saves on a whole frame).  host[5] = tiles the
    # PREVIOUS frame on this lane had to redo: when that frame ran at the lane's current level, the
    # lane moves on to fronts twice as deep, and after level 2 (or when many tiles fail at once) to
    # full sorts.  (The count lives in the lane's workspace, whose layout follows the frame's shape: it
    # only means something when the previous frame had the same shape, and a new shape starts afresh.)
    heavy = int(host[2]) + int(host[3]) + int(host[4])
    same_shape = (not own) and st.get("shape") == shape
    if FRAME_STATS is not None:
        _count_frame(FRAME_STATS, frame, host, grew, same_shape)
    if not own:
        memo = st.setdefault("learnt", {})   # shape -> (full_sort, front_level): a lane that alternates between
        if not same_shape:                   # shapes (render.py's race between grids) does not learn them anew
            st["full_sort"], st["front_level"] = memo.get(shape, (False, 0))
        # (low 32 bits: a bin redone because of its depth cut-off -- the high bits -- says nothing about the fronts)
        # (round 4: ... unless nearly every heavy bin had to be regenerated: that is a scene swap, the fronts of the new
        # scene's bins are as stale as the cut-offs were -- measured: the frame after such a one fails its fronts as
        # well, 70 ms at config 4 -- so the lane goes to full sorts one frame earlier)
        elif (rc == 0 and not (int(host[7]) & 4) and not st.get("full_sort") and ((int(host[5]) >> 32) & 0x3fffffff) > max(8, heavy // 2)):
            st["full_sort"] = True
            # (round 5, advisor: a swap is over after a frame -- the cut-offs and fronts the next lazily sorted frame
            # leaves are fresh -- so this fall-back has a short FIXED patience and does not lengthen the next one's)
            st["full_sort_frames"], st["full_sort_limit"] = 0, SWAP_FULL_SORT
            if FRAME_STATS is not None:
                FRAME_STATS["full_sort_on"] = FRAME_STATS.get("full_sort_on", 0) + 1
        elif (rc == 0 and not (int(host[7]) & 4) and (int(host[5]) & 0xffffffff) > 0 and not st.get("full_sort")
              and st.get("prev_level") == st.get("front_level", 0)):
            if (int(host[5]) & 0xffffffff) > max(3, heavy // 4) or st.get("front_level", 0) >= 2:
                st["full_sort"] = True
                st["full_sort_frames"], st["full_sort_limit"] = 0, None   # (patience: retry_after, doubling)
                if FRAME_STATS is not None:
                    FRAME_STATS["full_sort_on"] = FRAME_STATS.get("full_sort_on", 0) + 1
            else:
                st["front_level"] = st.get("front_level", 0) + 1
                if FRAME_STATS is not None:
                    FRAME_STATS["front_level_up"] = FRAME_STATS.get("front_level_up", 0) + 1
        # ... and giving up is not for ever: a view through fog ends.  After RETRY_FULL_SORT frames on full sorts the
        # lane tries lazily sorted fronts again, at the depth it last used; if they fail again (the library reports
        # it one frame later, above) it is back on full sorts with twice the patience, up to 4096 frames.  (A failed
        # retry costs one or two frames of the clean-up pass: 5-6 ms on the heaviest scenes since round 4's two-launch clean-up.)
        if st.get("full_sort") and same_shape:
            st["full_sort_frames"] = st.get("full_sort_frames", 0) + 1
            fixed = st.get("full_sort_limit")
            if st["full_sort_frames"] >= (fixed if fixed else st.get("retry_after", RETRY_FULL_SORT)):
                st["full_sort"], st["full_sort_frames"] = False, 0
                if not fixed:
                    st["retry_after"] = min(2 * st.get("retry_after", RETRY_FULL_SORT), 4096)
                st["retried"] = 2   # (the next two lazily sorted frames are on probation: the count comes a frame late)
                if FRAME_STATS is not None:
                    FRAME_STATS["lazy_sort_retry"] = FRAME_STATS.get("lazy_sort_retry", 0) + 1
        elif same_shape and st.get("retried") and rc == 0 and not (int(host[7]) & 4):
            # (round 5, advisor: a retried lazily sorted frame that reports no redone bin ends the doubling -- before, every
            # benign scene swap or camera cut lengthened the NEXT fall-back, up to 4096 frames on the slow path)
            if (int(host[5]) & 0xffffffff) == 0 and st.get("prev_level") is not None:
                st["retried"] -= 1
                if st["retried"] == 0:
                    st["retry_after"] = RETRY_FULL_SORT
        st["shape"], st["prev_level"] = shape, (level if not mode & FULL_SORT else None)
        memo[shape] = (bool(st.get("full_sort")), int(st.get("front_level", 0)))
        if len(memo) > 64:
            memo.pop(next(iter(memo)))
    lazy = LAZY_SORT and not own and channels <= 4 and not st.get("full_sort")
    st["speculate"] = lazy or int(host[4]) == 0


class _Frame:
    """One ms_render_fwd frame: its marshalled arguments and the lane whose scratch it occupies.
    `run(phase)` makes the library call; `finish()` completes a frame that was only begun."""

    def __init__(self, means3d, scales, quats, opacities, colors, camera, background, tile_size,
                 stage_events, row_range, out, lane, stream=None, own=False, rows16=False, out_y0=None, own_last=True):
        """own=True: the frame gets FRESH scratch (workspace, intersection buffer) and the per-pixel
        records of the backward pass instead of the lane's cached buffers -- a differentiable frame
        keeps them until its backward has run.  Only the size hint, the pinned record and the event
        come from the lane.  own_last=False: render_alphas only -- the frame is then binned and sorted like an
        inference frame (lazily sorted fronts, no projected arrays) and its backward is the quad-wave rasteriser,
        which needs no last_ids (3 channels)."""
        self.L = L = _hip.lib()
        self.dev = dev = means3d.device
        N = means3d.shape[0]
        means3d, scales, quats = _hip.f32c(means3d), _hip.f32c(scales), _hip.f32c(quats)
        op = _hip.f32c(opacities.reshape(-1))
        if colors.dtype == torch.float16:
            cdt, colors = 1, colors.contiguous()
        else:
            cdt, colors = 0, _hip.f32c(colors)
        C = colors.shape[1]
        assert means3d.shape == (N, 3) and scales.shape == (N, 3) and quats.shape == (N, 4)
        assert op.shape == (N,) and colors.shape == (N, C)
        bg = None if background is None else _hip.f32c(background.reshape(-1))
        H, W = camera.H, camera.W
        th, tw = -(-H // tile_size), -(-W // tile_size)
        vm = camera._viewmat_f32()
        if vm.device != dev:
            vm = vm.to(dev)
        self.st = st = _dev_state(dev, lane)
        assert not st.get("busy"), "a begun frame still occupies this lane: finish it first"
        self.own = own
        self.alphas = self.last = None
        if own:
            ws = torch.empty(L.ms_render_workspace_bytes(N, tw, th), dtype=torch.uint8, device=dev)
            hint = st.get("own_isect_bytes", 0)
            self.isect = torch.empty(hint, dtype=torch.uint8, device=dev) if hint else None
            self.alphas = torch.empty((H, W), dtype=torch.float32, device=dev)
            self.last = torch.empty((H, W), dtype=torch.int32, device=dev) if own_last else None
        else:
            ws = _grow(st, "ws", L.ms_render_workspace_bytes(N, tw, th), dev)
        self.ws, self.grid = ws, (N, tw, th)
        # rows16: row_range is in rows of 16 px although tile_size is 32 / 64 (MS_RENDER_ROWS16)
        self.rows16 = ROWS16 if (rows16 and row_range is not None and tile_size != 16) else 0
        r0, r1 = (0, th) if row_range is None else row_range
        # what the lane's sorting mode (below) was learnt on: scene size class (N to ~9 %: the clean-up count
        # the library reports one frame later lives at an offset that depends on the grid only), grid and band
        self.shape = (round(math.log2(N) * 8) if N > 0 else -1, tw, th, r0, r1)
        # sorting mode of this frame, fixed for both of its halves: lazily sorted fronts of the lane's
        # current depth level, or full sorts once the lane has given up on them
        self.level = int(st.get("front_level", 0))
        self.mode = (FULL_SORT if st.get("full_sort") else FRONT_LEVEL * self.level) | self.rows16
        if own and not own_last:
            # (round 5: a lean differentiable frame learns for itself -- own_mode / own_report below)
            full, self.level = own_mode(st, self.shape)
            self.mode = (FULL_SORT if full else FRONT_LEVEL * self.level) | self.rows16
        self.img_ptr = None
        if out is not None and out_y0 is not None:
            # `out` is a SLAB holding image rows [out_y0, out_y0 + out.shape[0]) (a rank's slot of a padded gather
            # buffer): the library addresses the full image, so it is handed the address image row 0 would have --
            # rows outside the band are never touched
            assert out.dtype == torch.float32 and out.is_contiguous() and out.device == dev and tuple(out.shape[1:]) == (W, C)
            px = tile_size if not self.rows16 else 16
            assert row_range is not None and out.shape[0] >= min(r1 * px, H) - min(r0 * px, H) and out_y0 == min(r0 * px, H)
            self.img = out
            self.img_ptr = ctypes.c_void_p(out.data_ptr() - out_y0 * W * C * 4)
        elif out is not None:
            assert out.dtype == torch.float32 and out.is_contiguous() and out.device == dev
            assert out.shape[0] >= H and tuple(out.shape[1:]) == (W, C)
            self.img = out
        else:
            self.img = torch.empty((H, W, C), dtype=torch.float32, device=dev)
        evs = None
        if stage_events is not None:
            evs = (ctypes.c_void_p * 4)(*[None if e is None else ctypes.c_void_p(e.cuda_event)
                                          for e in stage_events])  # entries may be None: not recorded
        self.keep = (means3d, scales, quats, op, colors, bg, vm, ws)  # keep the marshalled tensors alive
        self.head = (N, _hip.ptr(means3d), _hip.ptr(scales), 1, _hip.ptr(quats), _hip.ptr(op), _hip.ptr(colors),
                     cdt, C, _hip.ptr(vm), camera.fx, camera.fy, camera.cx, camera.cy, W, H, EPS2D, camera.near,
                     camera.far, tile_size, r0, r1, _hip.ptr(bg), _hip.ptr(ws), ws.numel())
        self.tail = (self.img_ptr if self.img_ptr is not None else _hip.ptr(self.img), _hip.ptr(self.alphas), _hip.ptr(self.last), evs,
                     ctypes.c_void_p(st["ev"].cuda_event) if st.get("speculate", True) else None,
                     _hip.stream(dev) if stream is None else ctypes.c_void_p(stream))
        self.host_ptr = ctypes.c_void_p(st["host"].data_ptr())

    def run(self, phase):
        phase |= self.mode
        isect = self.isect if self.own else self.st["isect"]
        return self.L.ms_render_fwd(*self.head, _hip.ptr(isect), 0 if isect is None else isect.numel(),
                                    self.host_ptr, phase, *self.tail)

    def begin(self):
        with _hip.on_device(self.dev):
            _hip.check(self.run(BEGIN), "ms_render_fwd(begin)")
        self.st["busy"] = True
        return self

    def finish(self, phase=FINISH, info=None):
        """-> (image, M) of the band.  Waits for the frame's size record, redoes the frame on the
        exact path if the speculation did not hold (growing the intersection buffer if needed)."""
        st, host = self.st, self.st["host_np"]
        grew = False
        with _hip.on_device(self.dev):
            rc = self.run(phase)
            if rc == 2:  # MS_ERR_WORKSPACE: the intersection buffer is too small for this frame's M
                need = int(host[5])
                grew = need > 0
                if self.own:
                    if need > 0:
                        # the speculative kernels may still be running on the old block
                        if self.isect is not None:
                            torch.cuda.synchronize(self.dev)
                        self.isect = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=self.dev)
                        st["own_isect_bytes"] = self.isect.numel()
                        rc = self.run(RESUME)
                elif need > 0 and (st["isect"] is None or st["isect"].numel() < need):
                    _grow(st, "isect", need, self.dev, slack=1.25)
                    rc = self.run(RESUME)
            st["busy"] = False
            _hip.check(rc, "ms_render_fwd")
        _after_frame(st, host, rc, grew, shape=self.shape, level=self.level, mode=self.mode, own=self.own,
                     channels=self.head[8], frame=self)
        if info is not None:
            info["on_grid"] = int(host[6])
            info["flags"] = int(host[7])
            info["n_xl"] = int(host[4])
        return self.img, int(host[0])

    def intermediates(self, M, flags, n_xl):
        """Views of what the frame left in its scratch: means2d, conics, radii, tile_ranges,
        flatten_ids (ms_render_workspace_layout / the isect_buf layout rules of the header)."""
        N, tw, th = self.grid
        off = (ctypes.c_size_t * 6)()
        _hip.check(self.L.ms_render_workspace_layout(N, tw, th, off), "ms_render_workspace_layout")

        def view(buf, o, count, dtype, shape):
            nbytes = count * torch.empty((), dtype=dtype).element_size()
            return buf[o:o + nbytes].view(dtype).view(shape)
        ws = self.ws
        means2d = view(ws, off[0], N * 2, torch.float32, (N, 2))
        conics = view(ws, off[1], N * 3, torch.float32, (N, 3))
        radii = view(ws, off[3], N * 2, torch.int32, (N, 2))
        ranges = view(ws, off[4], tw * th * 2, torch.int32, (th, tw, 2))
        isect = self.isect if self.own else self.st["isect"]
        align = lambda v: (v + 255) // 256 * 256
        if flags & 4:   # exact layout
            ids_off = align(8 * max(M, 1)) * (2 if n_xl > 0 else 1)
        else:           # sync-free layout: keys sized by the buffer's capacity
            cap = min((isect.numel() - 512) // 12, 0x7fffffff)
            ids_off = align(8 * cap)
        ids = view(isect, ids_off, M, torch.int32, (M,)) if M > 0 else torch.empty(0, dtype=torch.int32, device=self.dev)
        return means2d, conics, radii, ranges, ids


def render_fwd_hip(means3d, scales, quats, opacities, colors, camera, background, tile_size,
                   stage_events=None, row_range=None, out=None, lane=0, info=None, rows16=False, out_y0=None):
    """-> (image (H,W,C) f32, M).  `background` may be None.  stage_events: None or a list of 4
    torch.cuda.Event that have been recorded once (so their handles exist).
    row_range=(r0, r1) renders only tile rows [r0, r1) into `out` (a caller-owned framebuffer of
    at least H rows: the multi-GPU gather buffer); M is then the band's intersection count.
    `lane` selects an independent set of scratch buffers (frames in flight at the same time each
    need their own).  `info`: optional dict, receives `on_grid` = number of Gaussians touching the
    FULL tile grid (band-independent)."""
    if lane == 0:   # the calling thread's own scratch: no lock
        return _Frame(means3d, scales, quats, opacities, colors, camera, background, tile_size,
                      stage_events, row_range, out, lane, rows16=rows16, out_y0=out_y0).finish(WHOLE, info)
    with _frame_lock:
        return _Frame(means3d, scales, quats, opacities, colors, camera, background, tile_size,
                      stage_events, row_range, out, lane, rows16=rows16, out_y0=out_y0).finish(WHOLE, info)


def last_frame_list_entries(dev, N, tile_w, tile_h, lane=0):
    """Sum of the list lengths the rasteriser of the LAST frame on (dev, lane) was given, read back from the
    tile ranges the frame left in its workspace (for a split frame: the 16x16-block lists cut from its 32-px
    bins).  tile_w x tile_h = the grid of that frame's tile size.  Synchronises; for benchmarks / tests."""
    st = _state.get((dev, lane))
    if st is None or st.get("ws") is None:
        raise RuntimeError("no frame has been rendered on this lane")
    off = (ctypes.c_size_t * 6)()
    _hip.check(_hip.lib().ms_render_workspace_layout(N, tile_w, tile_h, off), "ms_render_workspace_layout")
    torch.cuda.synchronize(dev)
    nb = tile_w * tile_h * 8
    r = st["ws"][off[4]:off[4] + nb].view(torch.int32).view(tile_h, tile_w, 2)
    return int((r[..., 1] - r[..., 0]).clamp_min(0).sum())


def render_begin_hip(means3d, scales, quats, opacities, colors, camera, background, tile_size,
                     row_range=None, out=None, lane=0, stream=None, stage_events=None):
    """Enqueue a frame on the current stream (or on the raw hipStream_t handle `stream`; scratch
    that a redo has to grow is then allocated on the current stream and must only be used by
    streams the caller orders after it) WITHOUT waiting for its size record; -> a frame
    object whose .finish(info=None) -> (image, M) completes it.  The host can begin the next
    frame (on another lane) before finishing this one, which hides the count -> host -> emit
    hand-off latency entirely."""
    return _Frame(means3d, scales, quats, opacities, colors, camera, background, tile_size,
                  stage_events, row_range, out, lane, stream).begin()


@torch.no_grad()
def render_batch_hip(means3d, scales, quats, opacities, colors, cameras, background, tile_size, out=None):
    """Render the same Gaussians from several cameras -> (C, H, W, channels) f32: ONE library call,
    ms_render_fwd_batch (the camera dimension the reference's kernels carry and its wrappers pin to 1).

    Views are independent, so the library keeps two in flight on the two lane streams: view i+1's
    projection / counting (memory and latency bound) runs beside view i's rasteriser (VALU bound), and
    view i+1 is enqueued before the host waits for view i's size record.  Each lane has its own scratch;
    the caller's stream waits for both lanes before the batch is handed back."""
    L = _hip.lib()
    dev = means3d.device
    H, W = cameras[0].H, cameras[0].W
    assert all(c.H == H and c.W == W for c in cameras), "all cameras of a batch share one image size"
    near, far = cameras[0].near, cameras[0].far
    assert all(c.near == near and c.far == far for c in cameras), "all cameras of a batch share near / far planes"
    C = len(cameras)
    N = means3d.shape[0]
    means3d, scales, quats = _hip.f32c(means3d), _hip.f32c(scales), _hip.f32c(quats)
    op = _hip.f32c(opacities.reshape(-1))
    if colors.dtype == torch.float16:
        cdt, colors = 1, colors.contiguous()
    else:
        cdt, colors = 0, _hip.f32c(colors)
    CD = colors.shape[1]
    bg = None if background is None else _hip.f32c(background.reshape(-1))
    if out is None:
        out = torch.empty((C, H, W, CD), dtype=torch.float32, device=dev)
    else:   # a caller-owned slab (the view-sharded multi-GPU path renders straight into its gather buffer)
        assert out.dtype == torch.float32 and out.is_contiguous() and out.device == dev and tuple(out.shape) == (C, H, W, CD)
    if C == 0:
        return out, [], None
    vms = torch.stack([c._viewmat_f32().to(dev) for c in cameras]).contiguous()       # (C, 4, 4) on the device
    intr = (ctypes.c_float * (4 * C))(*[v for c in cameras for v in (c.fx, c.fy, c.cx, c.cy)])
    th, tw = -(-H // tile_size), -(-W // tile_size)
    cur = torch.cuda.current_stream(dev)
    streams = _lane_streams(dev)
    n_lanes = min(len(streams), 2, C)
    with _frame_lock:
        sts = [_dev_state(dev, 1 + k) for k in range(n_lanes)]
        for st in sts:
            assert not st.get("busy"), "a begun frame still occupies this lane: finish it first"
            _grow(st, "ws", L.ms_render_workspace_bytes(N, tw, th), dev)
        for s_ in streams[:n_lanes]:
            s_.wait_stream(cur)      # inputs (and the marshalled copies above) are ready
        counts = (ctypes.c_int64 * C)()
        done, need, lane_i = ctypes.c_int(0), ctypes.c_size_t(0), ctypes.c_int(0)
        with _hip.on_device(dev):
            while True:
                lanes = (_hip.ViewLane * n_lanes)()
                for k, st in enumerate(sts):
                    isect = st["isect"]
                    lanes[k] = _hip.ViewLane(st["ws"].data_ptr(), st["ws"].numel(),
                                             None if isect is None else isect.data_ptr(),
                                             0 if isect is None else isect.numel(), st["host"].data_ptr(),
                                             st["ev"].cuda_event, streams[k].cuda_stream)
                rc = L.ms_render_fwd_batch(C, N, _hip.ptr(means3d), _hip.ptr(scales), 1, _hip.ptr(quats), _hip.ptr(op),
                                           _hip.ptr(colors), cdt, CD, _hip.ptr(vms), ctypes.cast(intr, ctypes.c_void_p),
                                           W, H, EPS2D, near, far, tile_size, _hip.ptr(bg), n_lanes, lanes, 0,
                                           _hip.ptr(out), ctypes.cast(counts, ctypes.c_void_p), ctypes.byref(done),
                                           ctypes.byref(need), ctypes.byref(lane_i))
                if rc == 2 and need.value > 0:   # MS_ERR_WORKSPACE: that lane's intersection buffer is too small
                    _grow(sts[lane_i.value], "isect", need.value, dev, slack=1.25)
                    continue
                _hip.check(rc, "ms_render_fwd_batch")
                break
        # the batch ran on these lanes at the default sorting mode and on ITS grid: what the lanes had learnt about the
        # frames of another caller (the asynchronous band path shares them) no longer describes their workspaces
        for st in sts:
            st["shape"] = None
    for s_ in streams[:n_lanes]:
        cur.wait_stream(s_)
    for t in (out, means3d, scales, quats, op, colors, vms) + (() if bg is None else (bg,)):
        for s_ in streams[:n_lanes]:
            t.record_stream(s_)
    # the last view's size record on the first lane (for the binning rule): pairs, Gaussians on the grid, flags
    h = sts[0]["host_np"]
    return out, [int(c) for c in counts], dict(m=int(h[0]), on_grid=int(h[6]), flags=int(h[7]))


_lanes = {}
LANE_CALIBRATION = {}   # device -> what _lane_streams measured when it chose the two lane streams (diagnostics)


def _pick_lane_pair(pair_ratio, with_current):
    """The lane pair from a calibration: pair_ratio[(a, b)] = time of one spin kernel on each of streams a and b over one
    kernel's time (~1.1-1.3: independent, ~1.9: one hardware queue), with_current[a] likewise against the caller's stream.
    The pair's own independence first; a stream on the caller's QUEUE (> 1.6) is out -- the waits put on the caller's
    stream would stall it: measured slower than no overlap at all; ties (within 0.01) go to the pair met first."""
    best = None
    for (a, b), r in sorted(pair_ratio.items()):
        shared = max(with_current[a], with_current[b])
        score = r + (10.0 if shared > 1.6 else 0.1 * shared)
        if best is None or score < best[0] - 0.01:
            best = (score, (a, b))
    return best[1]


def _lane_streams_if_any(dev):
    """The lane streams of `dev` if a frame ever asked for them (no calibration is started here)."""
    return list(_lanes.get(dev) or ())


def _lane_streams(dev):
    """The two streams that frames in flight at the same time alternate between (multi-view batches, the asynchronous
    single-GPU and sharded entry points).

    Round 4: which two streams is not indifferent.  The runtime spreads a process's streams over a few hardware queues in
    the order of their first use, and two streams that land on one queue -- or on two queues of one pipe, whose wide
    kernels the dispatcher takes one after the other -- do not overlap at all: measured at config 3, the third and fourth
    stream a process uses render 0.166 ms a frame (the blocking call's rate), every other pair tried 0.144
    (`scripts/two_lane_probe.py`), and a wait put on the CALLER's stream stalls a lane that shares its queue.  So the
    lanes are chosen once per device from six candidates by a 2 ms calibration: a single-thread spin kernel
    (torch.cuda._sleep) on each of two streams takes ~1.1x one kernel's time when they are independent, 1.3x on the pair
    that did not overlap, 2.0x on one queue; the most independent pair wins, streams that share the caller's queue
    excluded.  MOJOSPLAT_LANE_CALIBRATION=0: the first two streams created, as before."""
    ls = _lanes.get(dev)
    if ls is not None:
        return ls
    with _frame_lock:
        ls = _lanes.get(dev)
        if ls is not None:
            return ls
        cand = [torch.cuda.Stream(device=dev) for _ in range(6)]
        pick, rec = (0, 1), {"calibrated": False}
        sleep = getattr(torch.cuda, "_sleep", None)
        if sleep is not None and os.environ.get("MOJOSPLAT_LANE_CALIBRATION", "1") != "0":
            try:
                cur = torch.cuda.current_stream(dev)
                cycles = 150000

                def spin(*streams):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for s_ in streams:
                        with torch.cuda.stream(s_):
                            sleep(cycles)
                    torch.cuda.synchronize(dev)
                    return time.perf_counter() - t0
                with torch.cuda.device(dev):
                    for s_ in cand:
                        spin(s_)          # (first use, in this order)
                    solo = min(spin(cand[0]) for _ in range(3))
                    with_cur = [min(spin(cur, s_) for _ in range(2)) / solo for s_ in cand]
                    pairs = {}
                    for a in range(len(cand)):
                        for b in range(a + 1, len(cand)):
                            pairs[(a, b)] = min(spin(cand[a], cand[b]) for _ in range(2)) / solo
                pick = _pick_lane_pair(pairs, with_cur)
                pairs = {f"{a},{b}": round(r, 2) for (a, b), r in pairs.items()}
                rec = {"calibrated": True, "solo_us": round(solo * 1e6, 1), "with_current_stream": [round(x, 2) for x in with_cur],
                       "pairs": pairs, "picked": list(pick), "ratio_of_the_pick": pairs[f"{pick[0]},{pick[1]}"]}
            except Exception as e:  # noqa: BLE001  (a calibration that cannot run must not take the renderer down)
                rec = {"calibrated": False, "error": repr(e)}
        LANE_CALIBRATION[dev] = rec
        ls = [cand[pick[0]], cand[pick[1]]]
        _lanes[dev] = ls
        return ls
